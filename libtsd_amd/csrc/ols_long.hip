// ols_long.hip -- overlap-save FIR for LONG filters (514 .. 12289 taps) on the radix-16
// Stockham engine: same contract as ols.hip / the direct kernel (FiltreRIF<T,Tc>::step,
// libtsd core/src/filtrage/filtre-rt.cc:53-109), block size N = 4096 .. 16384 (about 4 K).
//
// One workgroup of N/16 threads owns one block: it loads N input samples (the first K-1
// overlap the previous block; before the stream start they come from the handle's history),
// transforms them in LDS (stockham16.hpp), multiplies by the frequency response H (natural
// order, pre-divided by N), transforms back with conj(FFT(conj .)) -- the spectrum is already
// in the register layout the first pass reads, so the two transforms need no exchange in
// between -- and stores the N-(K-1) valid outputs.  Real data: two blocks per transform (re / im).
// HBM traffic per output sample:
// 8 B * N/L read + 8 B written (the direct kernel's cost grows with K instead: 4K flop/sample).
#include "fir_internal.hpp"
#include "stockham16.hpp"
#include <cmath>
#include <complex>
#include <cstdlib>
#include <vector>

namespace tsdgpu {

template <int R0, bool REAL>
__global__ __launch_bounds__(1024) void ols_long_kernel(const void *__restrict__ xin, const void *__restrict__ hist,
                                                        void *__restrict__ yout, const cpx *__restrict__ H,
                                                        const cpx *__restrict__ TW, int N, int tpt, int K, int HL, int L,
                                                        int64_t n)
{
  extern __shared__ __attribute__((aligned(16))) char olsl_raw[];
  cpx *lds = reinterpret_cast<cpx *>(olsl_raw);
  const int j = threadIdx.x;
  const int64_t b = blockIdx.x;
  // REAL data: two consecutive real blocks ride as the real and imaginary part of one complex
  // block (the taps being real, conv(h, a + j b) = conv(h, a) + j conv(h, b)): block index b then
  // addresses the real blocks 2b and 2b+1
  const int64_t in0 = (REAL ? 2 * b : b) * L - (K - 1);    // stream index of block position 0 (first block of the pair)
  cpx v[16];
#pragma unroll
  for (int m = 0; m < 16; m++) {
    const int64_t idx = in0 + j + m * tpt;
    cpx s = s16::c_mk(0.f, 0.f);
    if (REAL) {
      const float *xr = reinterpret_cast<const float *>(xin), *hr = reinterpret_cast<const float *>(hist);
      if (idx < 0) s.x = hr[HL + idx];                       // idx >= -(K-1) >= -HL
      else if (idx < n) s.x = xr[idx];
      const int64_t idx2 = idx + L;                          // same position in the second block of the pair
      if (idx2 < 0) s.y = hr[HL + idx2];
      else if (idx2 < n) s.y = xr[idx2];
    } else {
      if (idx < 0) s = reinterpret_cast<const cpx *>(hist)[HL + idx];
      else if (idx < n) s = reinterpret_cast<const cpx *>(xin)[idx];
    }
    v[m] = s;
  }
  auto sync = []() { __syncthreads(); };
  s16::transform<R0>(v, lds, TW, N, j, tpt, sync);
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const cpx t = s16::c_mul(v[q], H[j + q * tpt]);
    v[q] = s16::c_mk(t.x, -t.y);                            // conj: the inverse runs as conj(FFT(conj .))
  }
  sync();                                                    // the image is rewritten by the next transform
  s16::transform<R0>(v, lds, TW, N, j, tpt, sync);
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const int pos = j + q * tpt;
    if (pos < K - 1) continue;
    if (REAL) {
      const int64_t o = 2 * b * L + pos - (K - 1);
      float *yr = reinterpret_cast<float *>(yout);
      if (o < n) yr[o] = v[q].x;
      if (o + L < n) yr[o + L] = -v[q].y;                    // imaginary part of conj(...) restored
    } else {
      const int64_t o = b * L + pos - (K - 1);
      if (o < n) reinterpret_cast<cpx *>(yout)[o] = s16::c_mk(v[q].x, -v[q].y);
    }
  }
}

namespace {
// in-place radix-2 FFT in double (host, plan creation only)
void host_fft(std::vector<std::complex<double>> &a)
{
  const size_t n = a.size();
  for (size_t i = 1, jj = 0; i < n; i++) {
    size_t bit = n >> 1;
    for (; jj & bit; bit >>= 1) jj ^= bit;
    jj ^= bit;
    if (i < jj) std::swap(a[i], a[jj]);
  }
  const double PI = 3.14159265358979323846;
  for (size_t len = 2; len <= n; len <<= 1) {
    const std::complex<double> wl = std::polar(1.0, -2 * PI / (double) len);
    for (size_t i = 0; i < n; i += len) {
      std::complex<double> w = 1.0;
      for (size_t k = 0; k < len / 2; k++) {
        const std::complex<double> u = a[i + k], t = a[i + k + len / 2] * w;
        a[i + k] = u + t;
        a[i + k + len / 2] = u - t;
        // recompute the twiddle from the angle every 64 steps to stop the recurrence drifting
        w = ((k + 1) & 63) ? w * wl : std::polar(1.0, -2 * PI * (double) (k + 1) / (double) len);
      }
    }
  }
}
}  // namespace

bool ols_long_supported(const tsdgpu_fir *f)
{
  static const int KMIN = dev_switch("OLS_LONG_MIN") ? atoi(dev_switch("OLS_LONG_MIN")) : 514;
  return f->K >= KMIN && f->K <= 12289;
}

int ols_long_plan_create(tsdgpu_fir *f)
{
  const int K = f->K;
  int N = 2048;
  static const int RATIO = dev_switch("OLS_LONG_RATIO") ? atoi(dev_switch("OLS_LONG_RATIO")) : 4;
  // measured on 2^26 complex samples (scripts/perf_long_fir.py): blocks of about 4 K are the
  // optimum (K = 1024: N = 2048 / 4096 / 8192 -> 0.42 / 0.32 / 0.36 ms), i.e. overlap <= 25 % up to
  // K = 4097, growing to 75 % at the 12289-tap limit of the 16384-point block
  while (N < 16384 && N < RATIO * (K - 1)) N <<= 1;
  f->ols_N = N;
  f->ols_L = N - (K - 1);
  f->ols_long = true;
  std::vector<std::complex<double>> h((size_t) N, 0.0);
  const float *t = (const float *) f->taps_host.data();
  for (int i = 0; i < K; i++) h[i] = f->tap_type == TSDGPU_F32 ? std::complex<double>(t[i], 0.0) : std::complex<double>(t[2 * i], t[2 * i + 1]);
  host_fft(h);
  std::vector<cpx> tab((size_t) N + N / 16);
  for (int k = 0; k < N; k++) tab[k] = make_float2((float) (h[k].real() / N), (float) (h[k].imag() / N));
  const double PI = 3.14159265358979323846;
  for (int i = 0; i < N / 16; i++) tab[N + i] = make_float2((float) std::cos(-2 * PI * i / N), (float) std::sin(-2 * PI * i / N));
  if (hipMalloc(&f->d_H, tab.size() * sizeof(cpx)) != hipSuccess)
    return set_err(TSDGPU_ERR_HIP, "ols_long: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
  if (hipMemcpy(f->d_H, tab.data(), tab.size() * sizeof(cpx), hipMemcpyHostToDevice) != hipSuccess)
    return set_err(TSDGPU_ERR_HIP, "ols_long: upload failed: %s", hipGetErrorString(hipGetLastError()));
#define OLSL_ATTR(R, B) (void) hipFuncSetAttribute((const void *) ols_long_kernel<R, B>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
  OLSL_ATTR(16, false); OLSL_ATTR(8, false); OLSL_ATTR(4, false); OLSL_ATTR(2, false);
  OLSL_ATTR(16, true); OLSL_ATTR(8, true); OLSL_ATTR(4, true); OLSL_ATTR(2, true);
#undef OLSL_ATTR
  (void) hipGetLastError();
  return TSDGPU_OK;
}

int ols_long_step(tsdgpu_fir *f, const void *x, void *y, int64_t n, hipStream_t st)
{
  const int N = f->ols_N, L = f->ols_L, tpt = N / 16;
  int logn = 0;
  while ((1 << logn) < N) logn++;
  const int r0 = 1 << ((logn & 3) == 0 ? 4 : (logn & 3));
  const bool real = f->data_type == TSDGPU_F32;
  const int64_t nblocks = cdiv(n, real ? 2 * (int64_t) L : L);      // real data: one workgroup per pair of blocks
  TSD_CHECK(nblocks <= 0x7fffffff, "fir_step: too many blocks");
  const size_t lds = (size_t) (N + N / 16) * sizeof(cpx);
  const cpx *H = (const cpx *) f->d_H, *TW = H + N;
#define OLSL_LAUNCH(R, B)                                                                                         \
  hipLaunchKernelGGL((ols_long_kernel<R, B>), dim3((unsigned) nblocks), dim3(tpt), lds, st, x, (const void *) fir_hist_read(f), \
                     y, H, TW, N, tpt, f->K, f->HL, L, n)
  if (real) {
    if (r0 == 16) OLSL_LAUNCH(16, true); else if (r0 == 8) OLSL_LAUNCH(8, true); else if (r0 == 4) OLSL_LAUNCH(4, true); else OLSL_LAUNCH(2, true);
  } else {
    if (r0 == 16) OLSL_LAUNCH(16, false); else if (r0 == 8) OLSL_LAUNCH(8, false); else if (r0 == 4) OLSL_LAUNCH(4, false); else OLSL_LAUNCH(2, false);
  }
#undef OLSL_LAUNCH
  TSD_HIP(hipGetLastError());
  return fir_update_history(f, x, n, st);
}

}  // namespace tsdgpu
