// micro-benchmark: what the access pattern of the four-step FFT passes costs against a linear copy.
// A batch of 1024 x 1024 complex matrices (8 MiB each, 2 GiB in all) is copied tile by tile by
// persistent 1024-thread workgroups, a tile = W adjacent columns x 1024 rows (row segments of 8 W
// bytes at an 8-KiB stride), double-buffered in registers like fft1m_cols_kernel.  Patterns:
//   SS: strided read, strided write (pass 2);  SL: strided read, linear 8-KiB runs written (pass 1)
//   LL: linear read and write.
// usage: ./strided_copy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
enum { SS = 0, SL = 1, LL = 2 };
// W columns of 8 B: TPR = W / 2 threads per row, RPP = 1024 / TPR rows per sweep, NS = 1024 / RPP sweeps
template <int W, int PAT, bool NT, int DEPTH>
__global__ __launch_bounds__(1024) void tilecopy(const char *__restrict__ x, char *__restrict__ y, int ntiles)
{
  constexpr int TPR = W / 2, RPP = 1024 / TPR, NS = 1024 / RPP;   // NS sweeps of 16 B per thread
  static_assert(NS % DEPTH == 0, "depth");
  const int t = threadIdx.x, rr = t / TPR, cc = t % TPR;
  constexpr int TPM = 1024 / W;   // tiles per matrix
  for (int id = blockIdx.x; id < ntiles; id += gridDim.x) {
    const size_t mat = (size_t) (id / TPM) << 23;
    const int ct = id % TPM;
    for (int s0 = 0; s0 < NS; s0 += DEPTH) {
      v4 q[DEPTH];
#pragma unroll
      for (int i = 0; i < DEPTH; i++) {
        const int row = rr + RPP * (s0 + i);
        const size_t off = PAT == LL ? mat + ((size_t) ct * 1024 * W * 8) + ((size_t) (s0 + i) * 1024 + t) * 16
                                     : mat + (size_t) row * 8192 + ct * W * 8 + cc * 16;
        const v4 *p = reinterpret_cast<const v4 *>(x + off);
        q[i] = NT ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
      for (int i = 0; i < DEPTH; i++) {
        const int row = rr + RPP * (s0 + i);
        const size_t off = PAT != SS ? mat + ((size_t) ct * 1024 * W * 8) + ((size_t) (s0 + i) * 1024 + t) * 16
                                     : mat + (size_t) row * 8192 + ct * W * 8 + cc * 16;
        v4 *p = reinterpret_cast<v4 *>(y + off);
        if (NT) __builtin_nontemporal_store(q[i], p); else *p = q[i];
      }
    }
  }
}
template <int W, int PAT, bool NT, int DEPTH> void run(const char *name, char *x, char *y, int nmat, int grid)
{
  hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
  const int ntiles = nmat * (1024 / W);
  tilecopy<W, PAT, NT, DEPTH><<<grid, 1024>>>(x, y, ntiles);
  (void) hipDeviceSynchronize();
  float best = 1e9;
  for (int it = 0; it < 5; it++) {
    (void) hipEventRecord(e0);
    tilecopy<W, PAT, NT, DEPTH><<<grid, 1024>>>(x, y, ntiles);
    (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
  }
  printf("%-34s grid %4d: %.3f ms  %.2f TB/s\n", name, grid, best, 2.0 * nmat * 8388608.0 / best / 1e9);
}
int main()
{
  const int nmat = 256;
  char *x, *y;
  (void) hipMalloc(&x, (size_t) nmat << 23); (void) hipMalloc(&y, (size_t) nmat << 23);
  (void) hipMemset(x, 1, (size_t) nmat << 23);
  for (int grid : {256, 512}) {
    run<16, LL, false, 8>("linear 16 B x8", x, y, nmat, grid);
    run<16, LL, true, 8>("linear nt", x, y, nmat, grid);
    run<16, SL, false, 8>("W=16 (128 B) strided rd, lin wr", x, y, nmat, grid);
    run<16, SS, false, 8>("W=16 (128 B) strided rd + wr", x, y, nmat, grid);
    run<16, SS, true, 8>("W=16 strided rd + wr nt", x, y, nmat, grid);
    run<32, SL, false, 8>("W=32 (256 B) strided rd, lin wr", x, y, nmat, grid);
    run<32, SS, false, 8>("W=32 (256 B) strided rd + wr", x, y, nmat, grid);
    run<64, SS, false, 8>("W=64 (512 B) strided rd + wr", x, y, nmat, grid);
    run<8, SS, false, 4>("W=8 (64 B) strided rd + wr", x, y, nmat, grid);
  }
  return 0;
}
