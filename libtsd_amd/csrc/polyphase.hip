// polyphase.hip -- the integer-rate stages around the resampler and the generic IIR:
//   Decimateur                (libtsd core/src/filtrage/filtre-rt.cc:127-169)   pure index pick
//   FiltreRIFDecim            (core/src/reechan/polyphase.cc:156-239)           FIR + keep 1 of R
//   FiltreRIFDemiBande        (polyphase.cc:54-149)                             half-band, R = 2
//   FiltreRIFUps              (polyphase.cc:246-341)                            polyphase x R
//   FiltreRII                 (filtre-rt.cc:177-289)                            direct form I
// They are compositions of the FIR kernels of fir.hip / ols.hip with small permutation
// kernels, all on device scratch (correctness-first: the decimators compute every
// full-rate output and keep one in R).
#include "common.hpp"
#include <cmath>

namespace tsdgpu {

// y[m] = x[start + m*R]
template <typename T>
__global__ void pick_kernel(const T *__restrict__ x, T *__restrict__ y, int64_t start, int R, int64_t nout)
{
  const int64_t m = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (m < nout) y[m] = x[start + m * R];
}
// y[a*R + i] = z[a]
template <typename T>
__global__ void interleave_kernel(const T *__restrict__ z, T *__restrict__ y, int R, int i, int64_t n)
{
  const int64_t a = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (a < n) y[a * R + i] = z[a];
}

// Recursive half of FiltreRII (filtre-rt.cc:251-279), literally sequential: one lane per
// real channel.  hist[c*Ky + k] = y[-1-k] (most recent first).
__global__ void rii_recursive_kernel(float *__restrict__ y, const float *__restrict__ denom, int Ky,
                                     float *__restrict__ hist, int nch, int64_t n)
{
  const int c = threadIdx.x;
  if (c >= nch) return;
  float *h = hist + c * Ky;
  const float d0 = denom[0];
  for (int64_t j = 0; j < n; j++) {
    float somme = y[j * nch + c];
    for (int k = 0; k < Ky; k++) somme -= h[k] * denom[k + 1];
    const float o = somme / d0;
    y[j * nch + c] = o;
    for (int k = Ky - 1; k > 0; k--) h[k] = h[k - 1];
    if (Ky > 0) h[0] = o;
  }
}

}  // namespace tsdgpu

using namespace tsdgpu;

struct tsdgpu_polyfir {
  int kind = 0, data_type = 0, R = 1, K = 0;
  std::vector<tsdgpu_fir *> fir;    // 1 (decimators) or R (upsampler phases)
  int cnt = 0;                      // inputs seen since the last kept output (decimators / pick)
  DevBuf z, in_stage, out_stage;
};

namespace tsdgpu {
int sos_create_ex(tsdgpu_sos **out, int data_type, const float *coefs_host, int nsec, float gain,
                  const float *rii1_host, int forme, int seeded);
}

struct tsdgpu_rii {
  int data_type = 0, Ky = 0;
  tsdgpu_sos *sos = nullptr;      // Kx <= 3 and Ky <= 2: one zero-seeded DF1 section on the block-parallel kernel
  tsdgpu_fir *fir = nullptr;
  float *d_denom = nullptr, *d_hist = nullptr;
  DevBuf in_stage, out_stage;
};

namespace {

template <typename T>
int launch_pick(const void *x, void *y, int64_t start, int R, int64_t nout, hipStream_t st)
{
  if (nout <= 0) return TSDGPU_OK;
  hipLaunchKernelGGL(pick_kernel<T>, dim3((unsigned) cdiv(nout, 256)), dim3(256), 0, st, (const T *) x, (T *) y, start, R, nout);
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}

int64_t poly_out_count(const tsdgpu_polyfir *p, int64_t n)
{
  switch (p->kind) {
    case TSDGPU_POLY_DECIM:
    case TSDGPU_POLY_HALFBAND: return (n + p->cnt) / p->R;
    case TSDGPU_POLY_UPS: return n * p->R;
    default: return (n + p->R - 1 - p->cnt) / p->R;      // Decimateur (filtre-rt.cc:139)
  }
}

}  // namespace

extern "C" {

int tsdgpu_polyfir_create(tsdgpu_polyfir **out, int kind, int data_type, const float *taps_host, int ntaps, int R)
{
  TSD_CHECK(out != nullptr, "polyfir_create: out is NULL");
  *out = nullptr;
  TSD_CHECK(kind >= TSDGPU_POLY_DECIM && kind <= TSDGPU_POLY_PICK, "polyfir_create: bad kind %d", kind);
  TSD_CHECK(data_type == TSDGPU_F32 || data_type == TSDGPU_C64, "polyfir_create: bad data_type %d", data_type);
  if (kind == TSDGPU_POLY_HALFBAND) R = 2;
  TSD_CHECK(R >= 1 && R <= 4096, "polyfir_create: bad rate %d", R);
  TSD_CHECK(kind == TSDGPU_POLY_PICK || (taps_host != nullptr && ntaps > 0), "polyfir_create: K > 0 required");
  tsdgpu_polyfir *p = new tsdgpu_polyfir();
  p->kind = kind;
  p->data_type = data_type;
  p->R = R;
  p->K = ntaps;
  int rc = TSDGPU_OK;
  if (kind == TSDGPU_POLY_DECIM || kind == TSDGPU_POLY_HALFBAND) {
    // window is correlated with the taps in forward order (polyphase.cc:223-229) == an FIR
    // with the taps reversed; half-band keeps the even taps and forces 0.5 on the centre sample
    std::vector<float> h((size_t) ntaps);
    for (int k = 0; k < ntaps; k++) {
      const int i = ntaps - 1 - k;
      float c = taps_host[i];
      if (kind == TSDGPU_POLY_HALFBAND) c = ((i & 1) == 0 ? c : 0.f) + (i == ntaps / 2 ? 0.5f : 0.f);
      h[k] = c;
    }
    tsdgpu_fir *f = nullptr;
    rc = tsdgpu_fir_create(&f, data_type, TSDGPU_F32, h.data(), ntaps, TSDGPU_FIR_AUTO);
    if (!rc) p->fir.push_back(f);
  } else if (kind == TSDGPU_POLY_UPS) {
    // coefs = c * R, zero-padded to a multiple of R (polyphase.cc:259-270); phase i correlates
    // the K/R-sample window with coefs[(R-1-i) + j*R]
    std::vector<float> c((size_t) ntaps);
    for (int i = 0; i < ntaps; i++) c[i] = taps_host[i] * (float) R;
    while (c.size() % (size_t) R) c.push_back(0.f);
    const int W = (int) c.size() / R;
    for (int i = 0; i < R && !rc; i++) {
      std::vector<float> g((size_t) W);
      for (int k = 0; k < W; k++) g[k] = c[(size_t) (R - 1 - i) + (size_t) (W - 1 - k) * R];
      tsdgpu_fir *f = nullptr;
      rc = tsdgpu_fir_create(&f, data_type, TSDGPU_F32, g.data(), W, TSDGPU_FIR_AUTO);
      if (!rc) p->fir.push_back(f);
    }
  }
  if (rc) {
    tsdgpu_polyfir_destroy(p);
    return rc;
  }
  *out = p;
  return TSDGPU_OK;
}

int64_t tsdgpu_polyfir_out_count(tsdgpu_polyfir *p, int64_t n) { return (!p || n < 0) ? -1 : poly_out_count(p, n); }

int tsdgpu_polyfir_step(tsdgpu_polyfir *p, const void *x, int64_t n, void *y, int64_t y_capacity, int64_t *n_out,
                        void *stream)
{
  TSD_CHECK(p != nullptr, "polyfir_step: NULL handle");
  TSD_CHECK(n >= 0, "polyfir_step: negative length");
  if (n_out) *n_out = 0;
  if (n == 0) return TSDGPU_OK;
  hipStream_t st = (hipStream_t) stream;
  const int64_t nout = poly_out_count(p, n);
  TSD_CHECK(nout <= y_capacity, "polyfir_step: output needs %lld samples, capacity is %lld", (long long) nout,
            (long long) y_capacity);
  TSD_CHECK(x != nullptr && (nout == 0 || y != nullptr), "polyfir_step: NULL buffer");
  const size_t sz = dtype_size(p->data_type);
  const void *dx = nullptr;
  void *dy = nullptr;
  bool staged = false;
  int rc = stage_in(x, (size_t) n * sz, p->in_stage, st, &dx);
  if (rc) return rc;
  rc = stage_out(y, (size_t) nout * sz, p->out_stage, &dy, &staged);
  if (rc) return rc;
  const bool cplx = p->data_type == TSDGPU_C64;
  if (p->kind == TSDGPU_POLY_PICK) {
    // Decimateur: picks x[cnt], x[cnt+R], ...; new cnt = (first index >= n) - n
    rc = cplx ? launch_pick<float2>(dx, dy, p->cnt, p->R, nout, st) : launch_pick<float>(dx, dy, p->cnt, p->R, nout, st);
    if (rc) return rc;
    p->cnt = (int) (p->cnt + nout * p->R - n);
  } else if (p->kind == TSDGPU_POLY_UPS) {
    rc = p->z.reserve((size_t) n * sz);
    if (rc) return rc;
    for (int i = 0; i < p->R; i++) {
      rc = tsdgpu_fir_step(p->fir[(size_t) i], dx, p->z.p, n, stream);
      if (rc) return rc;
      if (cplx)
        hipLaunchKernelGGL(interleave_kernel<float2>, dim3((unsigned) cdiv(n, 256)), dim3(256), 0, st,
                           (const float2 *) p->z.p, (float2 *) dy, p->R, i, n);
      else
        hipLaunchKernelGGL(interleave_kernel<float>, dim3((unsigned) cdiv(n, 256)), dim3(256), 0, st,
                           (const float *) p->z.p, (float *) dy, p->R, i, n);
      TSD_HIP(hipGetLastError());
    }
  } else {
    rc = p->z.reserve((size_t) n * sz);
    if (rc) return rc;
    rc = tsdgpu_fir_step(p->fir[0], dx, p->z.p, n, stream);
    if (rc) return rc;
    // outputs at stream inputs where the counter wraps: local index R-1-cnt, then every R
    const int64_t start = p->R - 1 - p->cnt;
    rc = cplx ? launch_pick<float2>(p->z.p, dy, start, p->R, nout, st) : launch_pick<float>(p->z.p, dy, start, p->R, nout, st);
    if (rc) return rc;
    p->cnt = (int) ((p->cnt + n) % p->R);
  }
  if (n_out) *n_out = nout;
  return finish_out(y, (size_t) nout * sz, dy, staged, st);
}

int tsdgpu_polyfir_reset(tsdgpu_polyfir *p)
{
  TSD_CHECK(p != nullptr, "polyfir_reset: NULL handle");
  p->cnt = 0;
  for (auto *f : p->fir) {
    const int rc = tsdgpu_fir_reset(f);
    if (rc) return rc;
  }
  return TSDGPU_OK;
}

int tsdgpu_polyfir_destroy(tsdgpu_polyfir *p)
{
  if (!p) return TSDGPU_OK;
  for (auto *f : p->fir) tsdgpu_fir_destroy(f);
  p->z.release();
  p->in_stage.release();
  p->out_stage.release();
  delete p;
  return TSDGPU_OK;
}

// ---- FiltreRII ------------------------------------------------------------------------------
int tsdgpu_rii_create(tsdgpu_rii **out, int data_type, const float *numer_host, int Kx, const float *denom_host, int Kd)
{
  TSD_CHECK(out != nullptr, "rii_create: out is NULL");
  *out = nullptr;
  TSD_CHECK(data_type == TSDGPU_F32 || data_type == TSDGPU_C64, "rii_create: bad data_type %d", data_type);
  TSD_CHECK(numer_host && Kx > 0 && denom_host && Kd > 0, "rii_create: numerator and denominator need >= 1 coefficient");
  TSD_CHECK(denom_host[0] != 0.f, "rii_create: denom[0] must be non zero");
  tsdgpu_rii *r = new tsdgpu_rii();
  r->data_type = data_type;
  r->Ky = Kd - 1;
  if (Kx <= 3 && Kd <= 3) {
    // y = (n0 x + n1 x1 + n2 x2 - d1 y1 - d2 y2) / d0 from zero memory == one FormeDirecte1
    // section with zero seed; coefficients pre-divided by d0 (the reference divides the sum)
    const float d0 = denom_host[0];
    float c[5] = {numer_host[0] / d0, Kx > 1 ? numer_host[1] / d0 : 0.f, Kx > 2 ? numer_host[2] / d0 : 0.f,
                  Kd > 1 ? denom_host[1] / d0 : 0.f, Kd > 2 ? denom_host[2] / d0 : 0.f};
    const int rc0 = tsdgpu::sos_create_ex(&r->sos, data_type, c, 1, 1.0f, nullptr, 1, 0);
    if (rc0) { delete r; return rc0; }
    *out = r;
    return TSDGPU_OK;
  }
  int rc = tsdgpu_fir_create(&r->fir, data_type, TSDGPU_F32, numer_host, Kx, TSDGPU_FIR_AUTO);
  const size_t hb = (size_t) std::max(r->Ky, 1) * 2 * sizeof(float);
  if (!rc && (hipMalloc((void **) &r->d_denom, (size_t) Kd * sizeof(float)) != hipSuccess ||
              hipMalloc((void **) &r->d_hist, hb) != hipSuccess))
    rc = set_err(TSDGPU_ERR_HIP, "rii_create: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
  if (!rc && (hipMemcpy(r->d_denom, denom_host, (size_t) Kd * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
              hipMemset(r->d_hist, 0, hb) != hipSuccess))
    rc = set_err(TSDGPU_ERR_HIP, "rii_create: upload failed: %s", hipGetErrorString(hipGetLastError()));
  if (rc) {
    tsdgpu_rii_destroy(r);
    return rc;
  }
  *out = r;
  return TSDGPU_OK;
}

int tsdgpu_rii_step(tsdgpu_rii *r, const void *x, void *y, int64_t n, void *stream)
{
  TSD_CHECK(r != nullptr, "rii_step: NULL handle");
  TSD_CHECK(n >= 0, "rii_step: negative length");
  if (n == 0) return TSDGPU_OK;
  TSD_CHECK(x != nullptr && y != nullptr, "rii_step: NULL buffer");
  if (r->sos) return tsdgpu_sos_step(r->sos, x, y, n, stream);
  hipStream_t st = (hipStream_t) stream;
  const size_t bytes = (size_t) n * dtype_size(r->data_type);
  const void *dx = nullptr;
  void *dy = nullptr;
  bool staged = false;
  int rc = stage_in(x, bytes, r->in_stage, st, &dx);
  if (rc) return rc;
  rc = stage_out(y, bytes, r->out_stage, &dy, &staged);
  if (rc) return rc;
  rc = tsdgpu_fir_step(r->fir, dx, dy, n, stream);                      // (1) non-recursive part
  if (rc) return rc;
  const int nch = r->data_type == TSDGPU_C64 ? 2 : 1;
  hipLaunchKernelGGL(rii_recursive_kernel, dim3(1), dim3(64), 0, st, (float *) dy, r->d_denom, r->Ky, r->d_hist, nch, n);
  TSD_HIP(hipGetLastError());                                           // (2) recursive part
  return finish_out(y, bytes, dy, staged, st);
}

int tsdgpu_rii_destroy(tsdgpu_rii *r)
{
  if (!r) return TSDGPU_OK;
  tsdgpu_sos_destroy(r->sos);
  tsdgpu_fir_destroy(r->fir);
  if (r->d_denom) (void) hipFree(r->d_denom);
  if (r->d_hist) (void) hipFree(r->d_hist);
  r->in_stage.release();
  r->out_stage.release();
  delete r;
  return TSDGPU_OK;
}

}  // extern "C"
