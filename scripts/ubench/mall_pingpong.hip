// micro-benchmark: does the 256 MiB Infinity Cache absorb a write-then-read intermediate buffer?
// A long stream x (2 GiB) is copied to y through a small buffer T of S bytes, chunk by chunk:
// x_i -> T, then T -> y_i.  If T stays on die, the pair of copies costs about one HBM round trip;
// if not, two.  Streams use plain or non-temporal accesses.  usage: ./mall_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
template <bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(256) void copyk(const v4 *__restrict__ x, v4 *__restrict__ y, long n)
{
  const long stride = (long) gridDim.x * 256 * 4;
  for (long i = (long) blockIdx.x * 1024 + threadIdx.x; i < n; i += stride) {
    v4 v[4];
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (i + 256 * r < n) v[r] = NT_LD ? __builtin_nontemporal_load(x + i + 256 * r) : x[i + 256 * r];
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (i + 256 * r < n) {
        if (NT_ST) __builtin_nontemporal_store(v[r], y + i + 256 * r);
        else y[i + 256 * r] = v[r];
      }
  }
}
int main()
{
  const size_t total = 2ull << 30;
  v4 *x, *y, *T;
  (void) hipMalloc(&x, total); (void) hipMalloc(&y, total); (void) hipMalloc(&T, 1ull << 30);
  (void) hipMemset(x, 1, total); (void) hipMemset(T, 0, 1ull << 30);
  hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
  const int grid = 256 * 8;
  auto timeit = [&](auto fn) {
    fn(); (void) hipDeviceSynchronize();
    float best = 1e9;
    for (int it = 0; it < 3; it++) {
      (void) hipEventRecord(e0); fn(); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
      float ms; (void) hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    return best;
  };
  float t = timeit([&] { copyk<false, false><<<grid, 256>>>(x, y, total / 16); });
  printf("direct copy plain        : %.3f ms  %.2f TB/s (r+w)\n", t, 2.0 * total / t / 1e9);
  t = timeit([&] { copyk<true, true><<<grid, 256>>>(x, y, total / 16); });
  printf("direct copy nt/nt        : %.3f ms  %.2f TB/s (r+w)\n", t, 2.0 * total / t / 1e9);
  for (size_t S : {8ull << 20, 16ull << 20, 32ull << 20, 64ull << 20, 128ull << 20, 256ull << 20, 512ull << 20, 1024ull << 20}) {
    const long nchunk = total / S, n = S / 16;
    for (int pol = 0; pol < 2; pol++) {
      t = timeit([&] {
        for (long c = 0; c < nchunk; c++) {
          if (pol == 0) {
            copyk<false, false><<<grid, 256>>>(x + c * n, T, n);
            copyk<false, false><<<grid, 256>>>(T, y + c * n, n);
          } else {
            copyk<true, false><<<grid, 256>>>(x + c * n, T, n);
            copyk<false, true><<<grid, 256>>>(T, y + c * n, n);
          }
        }
      });
      // rate quoted like the FFT's roofline: algorithmic bytes = x read + y written
      printf("via T = %4zu MiB %s: %.3f ms  %.2f TB/s algorithmic (%ld launches, %.1f us each)\n", S >> 20,
             pol ? "nt streams   " : "plain streams", t, 2.0 * total / t / 1e9, 2 * nchunk, 1e3 * t / (2 * nchunk));
    }
  }
  // ring of G slots of 8 MiB (the real schedule reuses slots round-robin)
  for (int G : {2, 4, 8, 16}) {
    const size_t S = 8ull << 20;
    const long nchunk = total / S, n = S / 16;
    t = timeit([&] {
      for (long c = 0; c < nchunk + 1; c++) {
        if (c < nchunk) copyk<true, false><<<grid, 256>>>(x + c * n, T + (c % G) * n, n);
        if (c > 0) copyk<false, true><<<grid, 256>>>(T + ((c - 1) % G) * n, y + (c - 1) * n, n);
      }
    });
    printf("ring G = %2d x 8 MiB nt   : %.3f ms  %.2f TB/s algorithmic\n", G, t, 2.0 * total / t / 1e9);
  }
  return 0;
}
