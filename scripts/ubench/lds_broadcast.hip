// micro-benchmark: what a ds_read_b128 / ds_read_b64 costs when the lanes of a wave read (a) 64 different addresses without bank
// conflicts, (b) the SAME address (broadcast), (c) 16 lane groups of 4 with one address each, (d) random 512-B rows (the table-row
// gathers of the long interpolators).  One workgroup of 1024 threads per CU, LDS reads only, cycles per wave instruction from the
// wall clock (16 waves per CU share the LDS pipe: the figure is the LDS pipe's time per instruction).
// usage: ./lds_broadcast
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int B128>
__global__ __launch_bounds__(1024) void k(float *out, int iters)
{
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 1024) lds[i] = (float) i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned h = lane * 2654435761u;
  int off;                                                   // float index
  if (MODE == 0) off = lane * 4;                             // 64 distinct consecutive 16-B units
  else if (MODE == 1) off = 0;                               // one address
  else if (MODE == 2) off = (lane >> 2) * 4;                 // 16 addresses, 4 lanes each
  else if (MODE == 3) off = (int) ((h >> 20) & 127) * 124 + 4;              // random rows of pitch 31 x 16 B (odd), same column
  else if (MODE == 4) off = (int) (((lane % 21) * 2654435761u >> 20) & 127) * 124 + 4;   // 21 random rows, lanes l, l + 21, l + 42 share one
  else off = (int) (((lane / 3) * 2654435761u >> 20) & 127) * 124 + 4;                   // 22 random rows, three adjacent lanes share one
  float4 acc = make_float4(0, 0, 0, 0);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int o = (off + 8 * u + 64 * (it & 7)) & 16383;
      if (B128) {
        const float4 v = *reinterpret_cast<const float4 *>(&lds[o & ~3]);
        acc.x += v.x + v.w;
      } else {
        const float2 v = *reinterpret_cast<const float2 *>(&lds[o & ~1]);
        acc.x += v.y;
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
template <int MODE, int B128> void run(const char *name)
{
  float *out; (void) hipMalloc(&out, 64);
  const int iters = 20000, grid = 256;
  hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
  k<MODE, B128><<<grid, 1024>>>(out, 100);
  (void) hipDeviceSynchronize();
  float best = 1e9;
  for (int r = 0; r < 3; r++) {
    (void) hipEventRecord(e0);
    k<MODE, B128><<<grid, 1024>>>(out, iters);
    (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  // per CU: 16 waves x iters x 8 wave instructions through one LDS pipe
  const double instr = 16.0 * iters * 8, ns = best * 1e6 / instr;
  printf("%-44s %-5s %.3f ms  %.2f ns per wave instruction (= %.1f cycles at 2.1 GHz)\n", name, B128 ? "b128" : "b64", best, ns, ns * 2.1);
  (void) hipFree(out);
}
int main()
{
  run<0, 1>("64 distinct units, conflict-free");
  run<1, 1>("one address (broadcast)");
  run<2, 1>("16 addresses x 4 lanes");
  run<3, 1>("random rows (pitch 31 x 16 B)");
  run<4, 1>("21 random rows, lanes l / l+21 / l+42 share");
  run<5, 1>("22 random rows, adjacent triples share");
  run<0, 0>("64 distinct units, conflict-free");
  run<1, 0>("one address (broadcast)");
  run<3, 0>("random rows (pitch 31 x 16 B)");
  return 0;
}
