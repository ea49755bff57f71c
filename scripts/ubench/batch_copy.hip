// micro-benchmark: does batching a wave's reads and writes (B blocks of 8 KiB read, then B blocks written) raise the
// streaming rate over the read-8-KiB / write-8-KiB alternation of copy_rate.hip?  (The operators' memory skeletons --
// FIR overlap-save, SOS, resampler -- all sit near 5.0 TB/s; this asks whether the fine read/write interleave is why.)
// usage: ./batch_copy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float vec4f __attribute__((ext_vector_type(4)));
template <int B, bool NT>
__global__ __launch_bounds__(64) void copyk(const vec4f *__restrict__ x, vec4f *__restrict__ y, long nblk)
{
  const int lane = threadIdx.x;
  // wave w takes blocks [w*B + i*G*B, ...): B consecutive 8-KiB blocks per turn
  for (long b = (long) blockIdx.x * B; b < nblk; b += (long) gridDim.x * B) {
    vec4f v[B][8];
#pragma unroll
    for (int q = 0; q < B; q++) {
      const vec4f *xb = x + (b + q) * 512;
#pragma unroll
      for (int r = 0; r < 8; r++) v[q][r] = xb[64 * r + lane];
    }
#pragma unroll
    for (int q = 0; q < B; q++) {
      vec4f *yb = y + (b + q) * 512;
#pragma unroll
      for (int r = 0; r < 8; r++) {
        if (NT) __builtin_nontemporal_store(v[q][r], yb + 64 * r + lane);
        else yb[64 * r + lane] = v[q][r];
      }
    }
  }
}
template <int B, bool NT> void run(int waves_per_cu)
{
  const size_t bytes = 1ull << 29;
  vec4f *x, *y;
  (void) hipMalloc(&x, bytes); (void) hipMalloc(&y, bytes);
  (void) hipMemset(x, 1, bytes);
  const long nblk = bytes / 8192;
  const int grid = 256 * waves_per_cu;
  hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
  copyk<B, NT><<<grid, 64>>>(x, y, nblk);
  (void) hipDeviceSynchronize();
  float best = 1e9;
  for (int it = 0; it < 7; it++) {
    (void) hipEventRecord(e0);
    copyk<B, NT><<<grid, 64>>>(x, y, nblk);
    (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("batch %d x 8 KiB%s  waves/CU=%2d  %.3f ms  %.2f TB/s\n", B, NT ? " nt" : "   ", waves_per_cu, best, 2.0 * bytes / (best * 1e-3) / 1e12);
  (void) hipFree(x); (void) hipFree(y);
}
int main()
{
  for (int w : {4, 8, 12, 16}) {
    run<1, false>(w);
    run<2, false>(w);
    run<4, false>(w);
    run<1, true>(w);
    run<4, true>(w);
  }
  return 0;
}
