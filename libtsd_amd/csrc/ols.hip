// ols.hip -- overlap-save FFT convolution: the HBM-roofline path for long FIR filters.
//
// Serves the same contract as the direct kernel (FiltreRIF<T,Tc>::step, libtsd
// core/src/filtrage/filtre-rt.cc:53-109): output n is sum_k h[k] x[n-k] with the history
// carried across steps -- NOT the reference's OLA filter contract (filtre_rif_fft,
// core/src/fourier/fourier.cc:946-990), whose output is delayed by Nz-M samples and whose
// complex instantiation drops the imaginary part.
//
// One wave64 owns one 1024-sample block: it loads 1024 input samples (the last K-1 of them
// overlap the previous block), transforms them with the in-wave FFT of fft1024_wave.hpp,
// multiplies by the precomputed frequency response H (already in the FFT's register order
// and pre-divided by N), transforms back and stores the N-(K-1) valid outputs.  Waves are
// persistent: twiddles and H (62 complex per lane) stay in registers across blocks, so the
// steady state touches HBM only for x and y: 8 B read (+ (K-1)/L re-read, normally an L2
// hit) and 8 B written per sample.
#include "fir_internal.hpp"
#include "fft1024_wave.hpp"

namespace tsdgpu {

using namespace w1024;
constexpr int OLS_N = 1024;

template <bool INTERIOR>
__device__ __forceinline__ void ols_load(cpx (&v)[16], const cpx *__restrict__ x, const cpx *__restrict__ hist,
                                         int histlen, int64_t g0, int64_t n, int lane)
{
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int64_t g = g0 + 64 * r + lane;
    if (INTERIOR)
      v[r] = x[g];
    else
      v[r] = g < 0 ? hist[histlen + g] : (g < n ? x[g] : mk(0.f, 0.f));
  }
}

__global__ __launch_bounds__(64) void ols_kernel(const cpx *__restrict__ x, const cpx *__restrict__ hist,
                                                 cpx *__restrict__ y, const cpx *__restrict__ Hreg,
                                                 const cpx *__restrict__ TW1, const cpx *__restrict__ TW2,
                                                 int Km1, int histlen, int L, int64_t n, int64_t nblocks)
{
  __shared__ cpx lds[LDS_ELEMS];
  const int lane = threadIdx.x;
  cpx tw1[16], tw2[16], H[16], v[16];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    tw1[r] = TW1[r * 64 + lane];
    tw2[r] = TW2[r * 64 + lane];
    H[r] = Hreg[r * 64 + lane];
  }
  auto sync = []() { __syncthreads(); };

  for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int64_t o0 = b * (int64_t) L;       // first output of the block
    const int64_t g0 = o0 - Km1;              // first input of the block
    const bool interior = g0 >= 0 && g0 + OLS_N <= n;
    if (interior)
      ols_load<true>(v, x, hist, histlen, g0, n, lane);
    else
      ols_load<false>(v, x, hist, histlen, g0, n, lane);

    forward(v, lds, lane, tw1, tw2, sync);
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = cmul(v[r], H[r]);
    inverse(v, lds, lane, tw1, tw2, sync);
    __syncthreads();   // LDS is reused by the next block

    // sample t = 64*r + lane of the circular convolution is output o0 + t - (K-1)
    if (interior) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int t = 64 * r + lane;
        if (t >= Km1) y[o0 + t - Km1] = v[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int t = 64 * r + lane;
        const int64_t o = o0 + t - Km1;
        if (t >= Km1 && o < n) y[o] = v[r];
      }
    }
  }
}

bool ols_preferred(const tsdgpu_fir *f)
{
  // complex data only for now; the direct kernel is HBM-bound below ~48 taps and the
  // 1024-point block needs L = 1025-K >= 512 to stay efficient
  return f->data_type == TSDGPU_C64 && f->K >= 48 && f->K <= 513;
}

int ols_plan_create(tsdgpu_fir *f)
{
  if (f->data_type != TSDGPU_C64 || f->K > OLS_N / 2 + 1) {
    // outside the block-FFT kernel's envelope: serve the request with the direct kernel
    f->method = TSDGPU_FIR_DIRECT;
    return TSDGPU_OK;
  }
  const int N = OLS_N, K = f->K;
  f->ols_N = N;
  f->ols_L = N - (K - 1);
  // H[k] = sum_n h[n] exp(-2 pi i k n / N), in double, then /N and register order
  std::vector<double> hr(K), hi(K, 0.0);
  if (f->tap_type == TSDGPU_F32) {
    const float *t = (const float *) f->taps_host.data();
    for (int i = 0; i < K; i++) hr[i] = t[i];
  } else {
    const float *t = (const float *) f->taps_host.data();
    for (int i = 0; i < K; i++) { hr[i] = t[2 * i]; hi[i] = t[2 * i + 1]; }
  }
  std::vector<double> c(N), s(N);
  const double PI = 3.14159265358979323846;
  for (int i = 0; i < N; i++) { c[i] = std::cos(2 * PI * i / N); s[i] = -std::sin(2 * PI * i / N); }
  std::vector<cpx> H(N), Hreg(N), tw1(N), tw2(N);
  for (int k = 0; k < N; k++) {
    double ar = 0, ai = 0;
    for (int i = 0; i < K; i++) {
      const int m = (int) (((int64_t) k * i) % N);
      ar += hr[i] * c[m] - hi[i] * s[m];
      ai += hr[i] * s[m] + hi[i] * c[m];
    }
    H[k] = mk((float) (ar / N), (float) (ai / N));
  }
  for (int lane = 0; lane < 64; lane++)
    for (int r = 0; r < 16; r++) Hreg[r * 64 + lane] = H[freq_index(lane, r)];
  fill_twiddles(tw1.data(), tw2.data());
  const size_t bytes = (size_t) N * sizeof(cpx);
  if (hipMalloc(&f->d_H, 3 * bytes) != hipSuccess)
    return set_err(TSDGPU_ERR_HIP, "ols: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
  char *d = (char *) f->d_H;
  if (hipMemcpy(d, Hreg.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d + bytes, tw1.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d + 2 * bytes, tw2.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
    return set_err(TSDGPU_ERR_HIP, "ols: upload failed: %s", hipGetErrorString(hipGetLastError()));
  // persistent grid: as many waves as the device keeps resident
  int dev = 0, cus = 256, per_cu = 8;
  (void) hipGetDevice(&dev);
  (void) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ols_kernel, 64, 0) != hipSuccess || per_cu < 1) {
    (void) hipGetLastError();
    per_cu = 8;
  }
  f->ols_grid = cus * per_cu;
  return TSDGPU_OK;
}

void ols_plan_destroy(tsdgpu_fir *f)
{
  if (f->d_H) (void) hipFree(f->d_H);
  f->d_H = nullptr;
}

int ols_step(tsdgpu_fir *f, const void *x, void *y, int64_t n, hipStream_t st)
{
  const int L = f->ols_L;
  const int64_t nblocks = cdiv(n, L);
  const int64_t grid = nblocks < f->ols_grid ? nblocks : f->ols_grid;
  const cpx *d = (const cpx *) f->d_H;
  hipLaunchKernelGGL(ols_kernel, dim3((unsigned) grid), dim3(64), 0, st, (const cpx *) x,
                     (const cpx *) f->hist[f->cur], (cpx *) y, d, d + OLS_N, d + 2 * OLS_N, f->K - 1, f->KP, L,
                     n, nblocks);
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}

}  // namespace tsdgpu
