// micro-benchmark: issue rate of v_fma_f32 / v_add_f32 / v_pk_fma_f32 / v_pk_add_f32 on gfx950
// as a function of waves per SIMD.  usage: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters, float s)
{
  float a[16]; v2f p[16];
  for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 0.001f + i; p[i] = v2f{a[i], a[i] + 1}; }
  v2f sv = {s, s * 0.5f};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (MODE == 0) a[i] = __builtin_fmaf(a[i], s, 1.0f);
        if (MODE == 1) a[i] = a[i] + s;
        if (MODE == 2) p[i] = __builtin_elementwise_fma(p[i], sv, sv);
        if (MODE == 3) p[i] = p[i] + sv;
        if (MODE == 4) a[i] = a[i] * s;
      }
  }
  float r = 0;
  for (int i = 0; i < 16; i++) r += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int MODE> void run(const char *name, int flops_per_lane_inst)
{
  float *out; hipMalloc(&out, 256 * 64 * 64 * sizeof(float));
  const int iters = 4000;
  for (int wps = 1; wps <= 8; wps *= 2) {
    int blocks = 256 * 4 * wps;   // 64-thread blocks: one wave each
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(out, 10, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(out, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst = (double) blocks * iters * 64.0;   // wave-instructions
    double per_simd_per_s = inst / 1024.0 / (ms * 1e-3);
    printf("%-14s waves/SIMD=%d  %.3f ms  wave-inst/SIMD/s=%.3e  (=> %.2f cycles/inst at 2.4GHz)  %.1f TFLOP/s\n", name, wps, ms,
           per_simd_per_s, 2.4e9 / per_simd_per_s, inst * 64 * flops_per_lane_inst / (ms * 1e-3) / 1e12);
  }
  hipFree(out);
}
int main()
{
  run<0>("v_fma_f32", 2); run<1>("v_add_f32", 1); run<4>("v_mul_f32", 1); run<2>("v_pk_fma_f32", 4); run<3>("v_pk_add_f32", 2);
  return 0;
}
