// micro-benchmark: what a small host-buffer call costs -- H2D + kernel + D2H + sync on 16 KiB -- with pageable user
// memory handed to the runtime, and with a page-locked bounce buffer filled / drained by memcpy.  usage: ./small_copy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
__global__ void touch(const float *x, float *y, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] * 2.f;
}
static double us(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); }
int main()
{
  for (size_t bytes : {(size_t) 2048, (size_t) 16384, (size_t) 65536, (size_t) 262144}) {
    const int n = (int) (bytes / 4);
    float *hx = (float *) malloc(bytes), *hy = (float *) malloc(bytes), *px, *py, *dx, *dy;
    (void) hipHostMalloc((void **) &px, bytes); (void) hipHostMalloc((void **) &py, bytes);
    (void) hipMalloc((void **) &dx, bytes); (void) hipMalloc((void **) &dy, bytes);
    for (int i = 0; i < n; i++) hx[i] = (float) i;
    hipStream_t st; (void) hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    auto chain_pageable = [&] {
      (void) hipMemcpyAsync(dx, hx, bytes, hipMemcpyHostToDevice, st);
      touch<<<(n + 255) / 256, 256, 0, st>>>(dx, dy, n);
      (void) hipMemcpyAsync(hy, dy, bytes, hipMemcpyDeviceToHost, st);
      (void) hipStreamSynchronize(st);
    };
    auto chain_pinned = [&] {
      memcpy(px, hx, bytes);
      (void) hipMemcpyAsync(dx, px, bytes, hipMemcpyHostToDevice, st);
      touch<<<(n + 255) / 256, 256, 0, st>>>(dx, dy, n);
      (void) hipMemcpyAsync(py, dy, bytes, hipMemcpyDeviceToHost, st);
      (void) hipStreamSynchronize(st);
      memcpy(hy, py, bytes);
    };
    auto chain_zero = [&] {   // kernel reads / writes the pinned buffers directly
      memcpy(px, hx, bytes);
      touch<<<(n + 255) / 256, 256, 0, st>>>(px, py, n);
      (void) hipStreamSynchronize(st);
      memcpy(hy, py, bytes);
    };
    auto kernel_only = [&] {
      touch<<<(n + 255) / 256, 256, 0, st>>>(dx, dy, n);
      (void) hipStreamSynchronize(st);
    };
    auto med = [&](auto f) {
      double t[200];
      for (int i = 0; i < 20; i++) f();
      for (int i = 0; i < 200; i++) { auto a = std::chrono::steady_clock::now(); f(); t[i] = us(a); }
      std::sort(t, t + 200);
      return t[100];
    };
    printf("%7zu B: pageable chain %.1f us | pinned bounce chain %.1f us | zero-copy kernel %.1f us | kernel + sync alone %.1f us\n", bytes,
           med(chain_pageable), med(chain_pinned), med(chain_zero), med(kernel_only));
    free(hx); free(hy);
  }
  return 0;
}
