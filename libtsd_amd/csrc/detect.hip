// detect.hip -- device side of the pattern detector and of the delay estimator (SURVEY.md section 8f rows
// 1 and 3; libtsd core/src/fourier/detection.cc:100-400, core/src/fourier/estimation-delais.cc:9-118,
// core/src/fourier/fourier.cc:489-597).
//
// Detector.  The normalised correlation score of a stream against a fixed pattern,
//     s[t] = sqrt(N / M) * |c[t]| / sqrt(e[t]),   c = x correlated with the unit-energy pattern,
//                                                  e = mean of |x|^2 over the pattern's M samples,
// and its peaks.  Everything per-sample runs here: |x|^2, the two filters (the OLA engine with the
// response conj(FFT(pattern)) or a FIR with the reversed conjugated pattern; an M-tap moving average),
// the alignment of the energy with the correlator's delay, the score, and the peak search.  A peak is a
// sample above the threshold that dominates the M - 1 samples on either side; it is decided M samples
// late, from a rolling device buffer that keeps the last 2M scores and correlation values of the stream,
// so a peak at a block border needs no special case.  What goes back to the host per block is the
// caller's score vector (when it is a host vector) and ONE small record list: for every peak its index,
// the three scores and the three complex correlation values around it -- what the host needs for the
// sub-sample interpolation, gain, phase and noise estimate.
//
// Delay estimation.  correlation_freq (X0 conj(X1) sqrt(n), lags reordered), the biased / unbiased
// cross-correlation around it, and the peak of |corr| / (e1 e2) with its two neighbours: one pass each,
// on the device; estimation_délais returns after one small D2H.
#include "common.hpp"
#include <vector>
#include <mutex>
#include <algorithm>
#include <cmath>
#include <complex>

namespace tsdgpu {

typedef float2 cpx;

__global__ void det_abs2_kernel(const cpx *__restrict__ x, float *__restrict__ e, int64_t n)
{
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) e[i] = x[i].x * x[i].x + x[i].y * x[i].y;
}

// score of the block's n correlation outputs into the rolling buffers at [P, P + n); the energy of output i
// is the moving average D samples earlier (D = correlator delay - (M - 1)): from this block or the history
struct DetHeader { int count, pad[3]; };

// (small blocks are launch-bound: the workgroups after the n / 256 scoring ones write the next energy history, and the very
// first thread clears the peak counter of the search that follows)
__global__ void det_score_kernel(const cpx *__restrict__ corr, const float *__restrict__ en, const float *__restrict__ ehist, int D,
                                 float ratio, float *__restrict__ sbuf, cpx *__restrict__ cbuf, int P, int64_t n,
                                 float *__restrict__ ehist_next, DetHeader *__restrict__ hdr, unsigned nb_score)
{
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr->count = 0;
  if (blockIdx.x >= nb_score) {
    // next history: the last D moving-average values of (ehist ++ en)
    const int k = (int) (blockIdx.x - nb_score) * blockDim.x + threadIdx.x;
    if (k < D) {
      const int64_t src = n - D + k;
      ehist_next[k] = src >= 0 ? en[src] : ehist[D + src];
    }
    return;
  }
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  cpx c = corr[i];
  float m2 = c.x * c.x + c.y * c.y;
  if (m2 <= 1e-12f) {                       // numerically empty correlation values count as zero
    c = make_float2(0.f, 0.f);
    m2 = 0.f;
  }
  const float e = i >= D ? en[i - D] : ehist[D + (i - D)];
  sbuf[P + i] = ratio * sqrtf(m2 / (e + 1e-20f));
  cbuf[P + i] = c;
}


// peaks among buffer positions j in [M + 1, M + n]: above the threshold, larger than the M - 1 later
// samples and not smaller than the M - 1 earlier ones.  One record per peak (unordered; sorted by the host).
// (the workgroups after the n / 256 searching ones move the last P samples of the rolling buffers to the front of the other pair)
__global__ void det_peak_kernel(const float *__restrict__ sbuf, const cpx *__restrict__ cbuf, int M, int P, int64_t n, float seuil,
                                DetHeader *__restrict__ hdr, tsdgpu_peak *__restrict__ recs, int max_recs, float *__restrict__ s_nxt,
                                cpx *__restrict__ c_nxt, unsigned nb_peak)
{
  if (blockIdx.x >= nb_peak) {
    const int i = (int) (blockIdx.x - nb_peak) * blockDim.x + threadIdx.x;
    if (i < P) {
      s_nxt[i] = sbuf[n + i];
      c_nxt[i] = cbuf[n + i];
    }
    return;
  }
  const int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int64_t j = M + 1 + k;
  const float s = sbuf[j];
  if (!(s > seuil)) return;
  for (int d = 1; d < M; d++)
    if (sbuf[j + d] >= s || sbuf[j - d] > s) return;
  const int slot = atomicAdd(&hdr->count, 1);
  if (slot >= max_recs) return;
  tsdgpu_peak r;
  r.index = (int) (j - P);
  r.s_m1 = sbuf[j - 1]; r.s0 = s; r.s_p1 = sbuf[j + 1];
  r.c_m1[0] = cbuf[j - 1].x; r.c_m1[1] = cbuf[j - 1].y;
  r.c0[0] = cbuf[j].x; r.c0[1] = cbuf[j].y;
  r.c_p1[0] = cbuf[j + 1].x; r.c_p1[1] = cbuf[j + 1].y;
  recs[slot] = r;
}


// ---- correlations ----------------------------------------------------------------------------------
// Y(0) = X0(0) conj(X1(0)), Y(i) = X0(n - i) conj(X1(n - i)), all times sqrt(n)   (fourier.cc:489-503)
__global__ void corr_freq_kernel(const cpx *__restrict__ X0, const cpx *__restrict__ X1, cpx *__restrict__ Y, int n, float g)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k = i == 0 ? 0 : n - i;
  const cpx a = X0[k], b = X1[k];
  Y[i] = make_float2((a.x * b.x + a.y * b.y) * g, (a.y * b.x - a.x * b.y) * g);
}
// out[0, 2m - 1): lags -(m-1) .. (m-1) taken from the circular correlation r of length L (positive lags at
// the head, negative ones at the tail), divided by n; unbiased: further divided by (n - |lag|) / n
__global__ void xcorr_extract_kernel(const cpx *__restrict__ r, cpx *__restrict__ out, int L, int n, int m, int unbiased)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * m - 1) return;
  const int lag = i - (m - 1);
  const cpx v = lag >= 0 ? r[lag] : r[L + lag];
  float w = 1.0f / (float) n;
  cpx o = make_float2(v.x * w, v.y * w);
  if (unbiased && lag != 0) {
    const float d = (float) (n - abs(lag)) / (float) n;
    // complex / real-valued complex, the reference's operation (fourier.cc:575-583)
    o = make_float2(o.x / d, o.y / d);
  }
  out[i] = o;
}
// sum of |x|^2 (double) and arg max of |c| with the two neighbours, one workgroup each (inputs of a delay
// estimate are a few thousand to a few million samples: a single 1024-lane pass is enough)
__global__ __launch_bounds__(1024) void energy_kernel(const cpx *__restrict__ x, int64_t n, double *__restrict__ out)
{
  __shared__ double sh[1024];
  double s = 0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double) x[i].x * x[i].x + (double) x[i].y * x[i].y;
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) {
    if ((int) threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}
struct PeakOut { int index; float v_m1, v0, v_p1; };
__global__ __launch_bounds__(1024) void absmax_kernel(const cpx *__restrict__ c, int n, PeakOut *__restrict__ out)
{
  __shared__ float sv[1024];
  __shared__ int si[1024];
  float best = -1.f;
  int bi = 0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float a = hypotf(c[i].x, c[i].y);
    if (a > best) { best = a; bi = i; }         // first maximum of this lane's stride
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) {
    if ((int) threadIdx.x < st) {
      const float o = sv[threadIdx.x + st];
      const int oi = si[threadIdx.x + st];
      if (o > sv[threadIdx.x] || (o == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = o; si[threadIdx.x] = oi; }   // lowest index wins ties, like index_max
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int k = si[0];
    out->index = k;
    out->v0 = sv[0];
    out->v_m1 = k > 0 ? hypotf(c[k - 1].x, c[k - 1].y) : 0.f;
    out->v_p1 = k + 1 < n ? hypotf(c[k + 1].x, c[k + 1].y) : 0.f;
  }
}

}  // namespace tsdgpu

using namespace tsdgpu;

struct tsdgpu_detector {
  int M = 0, Ne = 0, N = 1, mode = 0, delais = 0, D = 0, P = 0;
  float seuil = 0.5f, ratio = 1.f;
  tsdgpu_ola *ola = nullptr;
  tsdgpu_fir *fir_corr = nullptr, *fir_en = nullptr;
  DevBuf x_stage, corr, e2, en, recs;
  float *ehist[2] = {nullptr, nullptr};
  float *sbuf[2] = {nullptr, nullptr};
  cpx *cbuf[2] = {nullptr, nullptr};
  int64_t cap = 0;       // samples the rolling buffers can take per step
  int cur = 0, ecur = 0;
  int max_recs = 256;
};

namespace {
inline unsigned nb(int64_t n) { return (unsigned) cdiv(std::max<int64_t>(n, 1), 256); }

int det_reserve(tsdgpu_detector *d, int64_t n)
{
  if (n <= d->cap) return TSDGPU_OK;
  const int64_t want = n + n / 4 + 1024;
  for (int b = 0; b < 2; b++) {
    float *ns = nullptr;
    cpx *nc = nullptr;
    TSD_HIP(hipMalloc((void **) &ns, (size_t) (d->P + want + 2) * sizeof(float)));
    TSD_HIP(hipMalloc((void **) &nc, (size_t) (d->P + want + 2) * sizeof(cpx)));
    TSD_HIP(hipMemset(ns, 0, (size_t) (d->P + want + 2) * sizeof(float)));
    TSD_HIP(hipMemset(nc, 0, (size_t) (d->P + want + 2) * sizeof(cpx)));
    if (d->sbuf[b]) {
      TSD_HIP(hipMemcpy(ns, d->sbuf[b], (size_t) d->P * sizeof(float), hipMemcpyDeviceToDevice));
      TSD_HIP(hipMemcpy(nc, d->cbuf[b], (size_t) d->P * sizeof(cpx), hipMemcpyDeviceToDevice));
      (void) hipFree(d->sbuf[b]);
      (void) hipFree(d->cbuf[b]);
    }
    d->sbuf[b] = ns;
    d->cbuf[b] = nc;
  }
  TSD_HIP(hipStreamSynchronize(nullptr));
  d->cap = want;
  return TSDGPU_OK;
}
}  // namespace

extern "C" {

int tsdgpu_detector_create(tsdgpu_detector **out, const void *motif_host, int M, int Ne, int mode, float seuil)
{
  TSD_CHECK(out != nullptr, "detector_create: out is NULL");
  *out = nullptr;
  TSD_CHECK(motif_host != nullptr && M >= 3, "detector_create: a pattern of at least 3 samples is needed (M = %d)", M);
  TSD_CHECK(mode == 0 || mode == 1, "detector_create: mode %d (0 = OLA engine, 1 = FIR)", mode);
  TSD_CHECK(Ne >= 1, "detector_create: block length %d", Ne);
  tsdgpu_detector *d = new tsdgpu_detector();
  d->M = M;
  d->Ne = Ne;
  d->mode = mode;
  d->seuil = seuil;
  d->P = 2 * M;
  const std::complex<float> *mo = (const std::complex<float> *) motif_host;
  int rc = TSDGPU_OK;
  do {
    if (mode == 0) {
      // correlation through the OLA engine: X *= conj(FFT_N(pattern)) on the device (detection.cc:141-187)
      rc = tsdgpu_ola_create(&d->ola, Ne, M - 1, nullptr);
      if (rc) break;
      d->N = tsdgpu_ola_fft_size(d->ola);
      if (2 * M > d->N) { rc = set_err(TSDGPU_ERR_INVALID, "detector_create: a pattern of %d samples does not fit the %d-point OLA blocks", M, d->N); break; }
      std::vector<std::complex<float>> t((size_t) d->N, 0.f), T((size_t) d->N);
      std::copy(mo, mo + M, t.begin());
      tsdgpu_fft *p = nullptr;
      rc = tsdgpu_fft_create(&p, d->N, 1);
      if (!rc) rc = tsdgpu_fft_step(p, t.data(), T.data(), 1, 1, nullptr);
      tsdgpu_fft_destroy(p);
      if (rc) break;
      for (auto &v : T) v = std::conj(v);
      rc = tsdgpu_ola_set_response(d->ola, T.data());
      if (rc) break;
      d->delais = Ne;
    } else {
      // correlation as a FIR with the reversed conjugated pattern (real taps when the pattern is real): the reference's MODE_RIF
      // is a time-domain filter -- exact zeros in, exact zeros out -- so moderate patterns stay on the direct kernel
      const int methode_exacte = M <= 1024 ? TSDGPU_FIR_DIRECT : TSDGPU_FIR_AUTO;
      double im = 0, tot = 0;
      for (int i = 0; i < M; i++) { im += std::fabs(mo[i].imag()); tot += std::abs(mo[i]); }
      d->N = 1;
      if (im / std::max(tot, 1e-300) < 1e-7) {
        std::vector<float> h((size_t) M);
        for (int i = 0; i < M; i++) h[i] = mo[M - 1 - i].real();
        rc = tsdgpu_fir_create(&d->fir_corr, TSDGPU_C64, TSDGPU_F32, h.data(), M, methode_exacte);
      } else {
        std::vector<std::complex<float>> h((size_t) M);
        for (int i = 0; i < M; i++) h[i] = std::conj(mo[M - 1 - i]);
        rc = tsdgpu_fir_create(&d->fir_corr, TSDGPU_C64, TSDGPU_C64, h.data(), M, methode_exacte);
      }
      if (rc) break;
      d->delais = M - 1;
    }
    d->ratio = std::sqrt((float) d->N) / std::sqrt((float) M);
    d->D = d->delais - (M - 1);
    std::vector<float> mg((size_t) M, (float) (1.0 / (double) M));
    // the energy average in the TIME domain: a score divides by it, and where the stream is quiet (its start, a pause) an
    // FFT convolution leaves rounding noise of the block's loudest samples in place of a tiny positive average -- scores
    // of 3e5 at the first sample of a stream were found by the fuzz sweep.  A direct sum of non-negative terms cannot.
    rc = tsdgpu_fir_create(&d->fir_en, TSDGPU_F32, TSDGPU_F32, mg.data(), M, M <= 8192 ? TSDGPU_FIR_DIRECT : TSDGPU_FIR_AUTO);
    if (rc) break;
    const size_t eb = (size_t) std::max(d->D, 1) * sizeof(float);
    if (hipMalloc((void **) &d->ehist[0], eb) != hipSuccess || hipMalloc((void **) &d->ehist[1], eb) != hipSuccess ||
        hipMemset(d->ehist[0], 0, eb) != hipSuccess || hipMemset(d->ehist[1], 0, eb) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
      rc = set_err(TSDGPU_ERR_HIP, "detector_create: allocation failed: %s", hipGetErrorString(hipGetLastError()));
      break;
    }
    rc = d->recs.reserve(sizeof(DetHeader) + (size_t) d->max_recs * sizeof(tsdgpu_peak));
    if (!rc) rc = det_reserve(d, std::max(Ne, 4096));
  } while (0);
  if (rc) {
    tsdgpu_detector_destroy(d);
    return rc;
  }
  *out = d;
  return TSDGPU_OK;
}

int tsdgpu_detector_delay(const tsdgpu_detector *d) { return d ? d->delais : -1; }
int tsdgpu_detector_fft_size(const tsdgpu_detector *d) { return d ? d->N : -1; }

int tsdgpu_detector_step(tsdgpu_detector *d, const void *x, int64_t n, float *scores, tsdgpu_peak *peaks, int max_peaks, int *n_peaks,
                         void *stream)
{
  TSD_CHECK(d != nullptr, "detector_step: NULL handle");
  TSD_CHECK(n >= 2, "detector_step: blocks of at least 2 samples expected (got %lld)", (long long) n);
  TSD_CHECK(x != nullptr && (peaks != nullptr || max_peaks == 0) && n_peaks != nullptr, "detector_step: NULL argument");
  *n_peaks = 0;
  hipStream_t st = (hipStream_t) stream;
  const void *dx = nullptr;
  int rc = stage_in(x, (size_t) n * sizeof(cpx), d->x_stage, st, &dx);
  if (!rc) rc = d->corr.reserve((size_t) n * sizeof(cpx));
  if (!rc) rc = d->e2.reserve((size_t) n * sizeof(float));
  if (!rc) rc = d->en.reserve((size_t) n * sizeof(float));
  if (!rc) rc = det_reserve(d, n);
  if (rc) return rc;
  // energy: |x|^2 through the M-tap moving average
  hipLaunchKernelGGL(det_abs2_kernel, dim3(nb(n)), dim3(256), 0, st, (const cpx *) dx, d->e2.as<float>(), n);
  TSD_HIP(hipGetLastError());
  rc = tsdgpu_fir_step(d->fir_en, d->e2.p, d->en.p, n, st);
  if (rc) return rc;
  // correlation
  if (d->mode == 0) {
    int64_t got = 0;
    TSD_CHECK(tsdgpu_ola_max_out(d->ola, n) <= n + d->Ne, "detector_step: unexpected OLA output bound");
    rc = d->corr.reserve((size_t) (n + d->Ne) * sizeof(cpx));          // (the engine may hand out up to one more block)
    if (rc) return rc;
    rc = tsdgpu_ola_step(d->ola, dx, n, d->corr.p, &got, st);
    if (rc) return rc;
    if (got != n)
      return set_err(TSDGPU_ERR_INVALID, "detector_step: the OLA correlator returned %lld samples for %lld inputs (feed whole blocks of %d samples)",
                     (long long) got, (long long) n, d->Ne);
  } else {
    rc = tsdgpu_fir_step(d->fir_corr, dx, d->corr.p, n, st);
    if (rc) return rc;
  }
  const int cur = d->cur, nxt = cur ^ 1;
  DetHeader *hdr = (DetHeader *) d->recs.p;
  tsdgpu_peak *recs = (tsdgpu_peak *) ((char *) d->recs.p + sizeof(DetHeader));
  const unsigned nbn = nb(n);
  hipLaunchKernelGGL(det_score_kernel, dim3(nbn + (d->D > 0 ? nb(d->D) : 0)), dim3(256), 0, st, d->corr.as<cpx>(), d->en.as<float>(),
                     d->ehist[d->ecur], d->D, d->ratio, d->sbuf[cur], d->cbuf[cur], d->P, n, d->ehist[d->ecur ^ 1], hdr, nbn);
  TSD_HIP(hipGetLastError());
  if (d->D > 0) d->ecur ^= 1;
  hipLaunchKernelGGL(det_peak_kernel, dim3(nbn + nb(d->P)), dim3(256), 0, st, d->sbuf[cur], d->cbuf[cur], d->M, d->P, n, d->seuil, hdr, recs,
                     d->max_recs, d->sbuf[nxt], d->cbuf[nxt], nbn);
  TSD_HIP(hipGetLastError());
  d->cur = nxt;
  // results: the score vector (if asked for) and ONE small copy of the peak records
  if (scores) {
    if (is_device_ptr(scores)) {
      if ((rc = device_copy_small(scores, d->sbuf[cur] + d->P, (size_t) n * sizeof(float), st))) return rc;
    } else {
      TSD_HIP(hipMemcpyAsync(scores, d->sbuf[cur] + d->P, (size_t) n * sizeof(float), hipMemcpyDeviceToHost, st));
    }
  }
  std::vector<char> host(sizeof(DetHeader) + (size_t) d->max_recs * sizeof(tsdgpu_peak));
  {
    const int rc2 = finish_out(host.data(), host.size(), d->recs.p, true, st);     // (page-locked bounce + synchronisation)
    if (rc2) return rc2;
  }
  const DetHeader *hh = (const DetHeader *) host.data();
  const tsdgpu_peak *hr = (const tsdgpu_peak *) (host.data() + sizeof(DetHeader));
  TSD_CHECK(hh->count <= d->max_recs, "detector_step: %d peaks in one block (limit %d): raise the threshold", hh->count, d->max_recs);
  std::vector<tsdgpu_peak> v(hr, hr + hh->count);
  std::sort(v.begin(), v.end(), [](const tsdgpu_peak &a, const tsdgpu_peak &b) { return a.index < b.index; });
  TSD_CHECK((int) v.size() <= max_peaks, "detector_step: %d peaks, room for %d", (int) v.size(), max_peaks);
  std::copy(v.begin(), v.end(), peaks);
  *n_peaks = (int) v.size();
  return TSDGPU_OK;
}

int tsdgpu_detector_destroy(tsdgpu_detector *d)
{
  if (!d) return TSDGPU_OK;
  tsdgpu_ola_destroy(d->ola);
  tsdgpu_fir_destroy(d->fir_corr);
  tsdgpu_fir_destroy(d->fir_en);
  for (int b = 0; b < 2; b++) {
    if (d->ehist[b]) (void) hipFree(d->ehist[b]);
    if (d->sbuf[b]) (void) hipFree(d->sbuf[b]);
    if (d->cbuf[b]) (void) hipFree(d->cbuf[b]);
  }
  d->x_stage.release(); d->corr.release(); d->e2.release(); d->en.release(); d->recs.release();
  delete d;
  return TSDGPU_OK;
}

// ---- cross-correlation and delay estimate -------------------------------------------------------------
// xcorrb / xcorr (fourier.cc:534-597): both vectors of n samples zero-padded to L = n + 2m, circular
// correlation through two forward transforms (one batched call), correlation_freq and one inverse
// transform, lags -(m-1) .. (m-1) extracted and scaled.  x, y, out: host or device; y == NULL: autocorrelation.
int tsdgpu_xcorr(const void *x, const void *y, int n, int m, int unbiased, void *out, void *stream)
{
  TSD_CHECK(x != nullptr && out != nullptr && n >= 1, "xcorr: bad argument");
  if (m < 0) m = n;
  TSD_CHECK(m >= 1 && m <= n, "xcorr: m = %d outside [1, n = %d]", m, n);
  hipStream_t st = (hipStream_t) stream;
  const int L = n + 2 * m;
  // the plan of this length and the scratch buffers are borrowed from a small reserve for the call (a one-shot xcorr() of
  // 4096 samples spent 0.2 of its 0.3 ms building a 4224-point plan and in hipMalloc / hipFree)
  struct Ctx {
    int dev = 0, L = 0;
    tsdgpu_fft *p = nullptr;
    DevBuf pad, spec, res;
    void libere() { tsdgpu_fft_destroy(p); pad.release(); spec.release(); res.release(); }
    size_t octets() const { return pad.cap + spec.cap + res.cap; }
  };
  static CtxReserve<Ctx> *reserve = new CtxReserve<Ctx>();
  Ctx *c = reserve->prend([L](const Ctx &k) { return k.L == L; });
  auto rend = [&]() { reserve->rend(c); };
  DevBuf &pad = c->pad, &spec = c->spec, &res = c->res;
  int rc = pad.reserve((size_t) 2 * L * sizeof(cpx));
  if (!rc) rc = spec.reserve((size_t) 2 * L * sizeof(cpx));
  if (!rc) rc = res.reserve((size_t) (2 * m - 1) * sizeof(cpx));
  if (!rc && c->L != L) {
    tsdgpu_fft_destroy(c->p);
    c->p = nullptr;
    c->L = 0;
    rc = tsdgpu_fft_create(&c->p, L, 2);
    if (!rc) c->L = L;
  }
  if (rc) { rend(); return rc; }
  tsdgpu_fft *p = c->p;
  cpx *px = pad.as<cpx>(), *py = px + L;
  auto fin = [&](int code) {
    (void) hipStreamSynchronize(st);
    rend();
    return code;
  };
  if (hipMemsetAsync(px, 0, (size_t) 2 * L * sizeof(cpx), st) != hipSuccess ||
      hipMemcpyAsync(px + m, x, (size_t) n * sizeof(cpx), hipMemcpyDefault, st) != hipSuccess ||
      hipMemcpyAsync(py + m, y ? y : x, (size_t) n * sizeof(cpx), hipMemcpyDefault, st) != hipSuccess)
    return fin(set_err(TSDGPU_ERR_HIP, "xcorr: staging failed: %s", hipGetErrorString(hipGetLastError())));
  rc = tsdgpu_fft_step(p, px, spec.p, 2, 1, st);
  if (rc) return fin(rc);
  hipLaunchKernelGGL(corr_freq_kernel, dim3(nb(L)), dim3(256), 0, st, spec.as<cpx>(), spec.as<cpx>() + L, px, L, std::sqrt((float) L));
  rc = tsdgpu_fft_step(p, px, py, 1, 0, st);
  if (rc) return fin(rc);
  const bool out_dev = is_device_ptr(out);
  cpx *dst = out_dev ? (cpx *) out : res.as<cpx>();
  hipLaunchKernelGGL(xcorr_extract_kernel, dim3(nb(2 * m - 1)), dim3(256), 0, st, py, dst, L, n, m, unbiased ? 1 : 0);
  if (hipGetLastError() != hipSuccess) return fin(set_err(TSDGPU_ERR_HIP, "xcorr: launch failed"));
  if (!out_dev && hipMemcpyAsync(out, dst, (size_t) (2 * m - 1) * sizeof(cpx), hipMemcpyDeviceToHost, st) != hipSuccess)
    return fin(set_err(TSDGPU_ERR_HIP, "xcorr: download failed"));
  return fin(TSDGPU_OK);
}

// estimation_délais (estimation-delais.cc:100-118): both vectors (already of one length n) through the
// biased cross-correlation with m = n, |corr| normalised by the two RMS values, arg max and quadratic
// interpolation of the peak -> (delay in samples, score).  One small D2H.
int tsdgpu_delay_estimate(const void *x, const void *y, int n, float *delay, float *score, void *stream)
{
  TSD_CHECK(x != nullptr && y != nullptr && delay != nullptr && score != nullptr && n >= 1, "delay_estimate: bad argument");
  hipStream_t st = (hipStream_t) stream;
  struct Ctx {
    int dev = 0;
    DevBuf xs, ys, cr, sm;
    void libere() { xs.release(); ys.release(); cr.release(); sm.release(); }
    size_t octets() const { return xs.cap + ys.cap + cr.cap + sm.cap; }
  };
  static CtxReserve<Ctx> *reserve = new CtxReserve<Ctx>();
  Ctx *c = reserve->prend([](const Ctx &) { return true; });
  DevBuf &xs = c->xs, &ys = c->ys, &cr = c->cr, &sm = c->sm;
  int rc = cr.reserve((size_t) (2 * n - 1) * sizeof(cpx));
  if (!rc) rc = sm.reserve(2 * sizeof(double) + sizeof(PeakOut));
  const void *dx = nullptr, *dy = nullptr;
  if (!rc) rc = stage_in(x, (size_t) n * sizeof(cpx), xs, st, &dx);
  if (!rc) rc = stage_in(y, (size_t) n * sizeof(cpx), ys, st, &dy);
  auto fin = [&](int code) {
    (void) hipStreamSynchronize(st);
    reserve->rend(c);
    return code;
  };
  if (rc) return fin(rc);
  rc = tsdgpu_xcorr(dx, dy, n, n, 0, cr.p, st);
  if (rc) return fin(rc);
  double *e = (double *) sm.p;
  PeakOut *pk = (PeakOut *) (e + 2);
  hipLaunchKernelGGL(energy_kernel, dim3(1), dim3(1024), 0, st, (const cpx *) dx, (int64_t) n, e);
  hipLaunchKernelGGL(energy_kernel, dim3(1), dim3(1024), 0, st, (const cpx *) dy, (int64_t) n, e + 1);
  hipLaunchKernelGGL(absmax_kernel, dim3(1), dim3(1024), 0, st, cr.as<cpx>(), 2 * n - 1, pk);
  if (hipGetLastError() != hipSuccess) return fin(set_err(TSDGPU_ERR_HIP, "delay_estimate: launch failed"));
  struct { double e[2]; PeakOut p; } h;
  if (hipMemcpyAsync(&h, sm.p, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return fin(set_err(TSDGPU_ERR_HIP, "delay_estimate: download failed"));
  const float e1 = std::sqrt((float) (h.e[0] / n)), e2 = std::sqrt((float) (h.e[1] / n));
  const float den = std::max(e1 * e2, 1e-37f);              // (the reference adds 1e-50f, i.e. nothing, in float)
  const float v0 = h.p.v0 / den, vm = h.p.v_m1 / den, vp = h.p.v_p1 / den;
  float dl = 0.f;
  if (h.p.index > 0 && h.p.index + 1 < 2 * n - 1) {
    dl = (vp - vm) / (2 * (2 * v0 - vp - vm));               // vertex of the parabola through the three samples
    dl = std::min(0.5f, std::max(-0.5f, dl));
  }
  *delay = (float) (h.p.index - (n - 1)) + dl;
  *score = v0;
  return fin(TSDGPU_OK);
}

}  // extern "C"
