// micro-benchmark: streaming copy rate vs access width and wave geometry (what the OLS kernel's
// memory path can hope for).  usage: ./copy_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename V, int ROWS>
__global__ __launch_bounds__(64) void copyk(const V *__restrict__ x, V *__restrict__ y, long nblk)
{
  const int lane = threadIdx.x;
  for (long b = blockIdx.x; b < nblk; b += gridDim.x) {
    V v[ROWS];
    const V *xb = x + b * (64 * ROWS);
#pragma unroll
    for (int r = 0; r < ROWS; r++) v[r] = xb[64 * r + lane];
    V *yb = y + b * (64 * ROWS);
#pragma unroll
    for (int r = 0; r < ROWS; r++) yb[64 * r + lane] = v[r];
  }
}
template <typename V, int ROWS> void run(const char *name, int waves_per_cu)
{
  const size_t bytes = 1ull << 29;   // 512 MiB in, 512 MiB out
  V *x, *y;
  (void) hipMalloc(&x, bytes); (void) hipMalloc(&y, bytes);
  (void) hipMemset(x, 1, bytes);
  const long nblk = bytes / (sizeof(V) * 64 * ROWS);
  const int grid = 256 * waves_per_cu;
  hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
  copyk<V, ROWS><<<grid, 64>>>(x, y, nblk);
  (void) hipDeviceSynchronize();
  float best = 1e9;
  for (int it = 0; it < 5; it++) {
    (void) hipEventRecord(e0);
    copyk<V, ROWS><<<grid, 64>>>(x, y, nblk);
    (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("%-22s waves/CU=%2d  %.3f ms  %.2f TB/s\n", name, waves_per_cu, best, 2.0 * bytes / (best * 1e-3) / 1e12);
  (void) hipFree(x); (void) hipFree(y);
}
int main()
{
  for (int w : {8, 12, 16, 32}) {
    run<float2, 16>("8B/lane x16 rows", w);
    run<float4, 8>("16B/lane x8 rows", w);
    run<float4, 16>("16B/lane x16 rows", w);
  }
  return 0;
}
