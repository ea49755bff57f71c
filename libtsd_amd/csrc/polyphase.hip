// polyphase.hip -- the integer-rate stages around the resampler and the generic IIR:
//   Decimateur                (libtsd core/src/filtrage/filtre-rt.cc:127-169)   pure index pick
//   FiltreRIFDecim            (core/src/reechan/polyphase.cc:156-239)           FIR + keep 1 of R
//   FiltreRIFDemiBande        (polyphase.cc:54-149)                             half-band, R = 2
//   FiltreRIFUps              (polyphase.cc:246-341)                            polyphase x R
//   FiltreRII                 (filtre-rt.cc:177-289)                            direct form I
// The FIR stages run one fused kernel (polyfir_fused_kernel: only the kept outputs of a decimator
// are computed, all R branches of the upsampler come from one pass over the input); very high
// rates or tap counts that do not fit its LDS tile fall back to compositions of the FIR
// kernels of fir.hip / ols.hip with small permutation kernels on device scratch.
#include "common.hpp"
#include <cmath>
#include <cstdlib>
#include <vector>

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

// Fused integer-rate FIR: output o belongs to group grp = o / NPH and phase ph = o % NPH and is
//     y[o] = sum_{k < W} g[ph][k] * x[grp * stride + start - k]          (x[i < 0] = history)
// Decimators: NPH = 1, stride = R (only the kept outputs are ever computed).  Upsampler: NPH = R,
// stride = 1, one W = ceil(K/R)-tap branch per output phase.  A workgroup stages the input
// span of its PF_TO outputs in LDS with coalesced loads; every thread then evaluates
// PF_TO/256 outputs from LDS (taps in LDS too).  Algorithmic bytes per input sample:
// 8 + 8/R (decimate) or 8 + 8 R (upsample) for complex data.
#ifndef PF_SPAN_TARGET
#define PF_SPAN_TARGET 2048
#endif
constexpr int PF_TO = 2048;          // outputs per workgroup (fewer when the decimation rate makes their input span too long)
constexpr int PF_MAX_SPAN = 16000;   // staged input samples per workgroup (129 KiB of complex data with the padding)
__device__ __forceinline__ float pf_mac(float acc, float g, float x) { return fmaf(g, x, acc); }
__device__ __forceinline__ float2 pf_mac(float2 acc, float g, float2 x) { return make_float2(fmaf(g, x.x, acc.x), fmaf(g, x.y, acc.y)); }
__device__ __forceinline__ float pf_zero(float) { return 0.f; }
__device__ __forceinline__ float2 pf_zero(float2) { return make_float2(0.f, 0.f); }
template <typename T>
__global__ __launch_bounds__(256) void polyfir_fused_kernel(const T *__restrict__ x, const T *__restrict__ hist, T *__restrict__ y,
                                                            const float *__restrict__ g, int NPH, int W, int stride,
                                                            int64_t start, int HW, int64_t n, int64_t nout, int TO)
{
  extern __shared__ __attribute__((aligned(16))) char pf_raw[];
  float *gs = reinterpret_cast<float *>(pf_raw);                         // NPH * W taps
  T *xs = reinterpret_cast<T *>(gs + ((NPH * W + 3) & ~3));              // staged inputs
  const int t = threadIdx.x;
  for (int i = t; i < NPH * W; i += 256) gs[i] = g[i];
  const int64_t o0 = (int64_t) blockIdx.x * TO;
  const int64_t o1 = min(o0 + TO, nout);                              // exclusive
  const int64_t grp0 = o0 / NPH, grp1 = (o1 - 1) / NPH;
  const int64_t i_lo = grp0 * stride + start - (W - 1), i_hi = grp1 * stride + start;   // inclusive input span
  const int span = (int) (i_hi - i_lo + 1);
  // 8 loads in flight per thread (a one-load-per-iteration loop exposed the HBM latency 16-32 times
  // per workgroup and ran the decimators at 0.30 ms per 2^26 samples)
  for (int i0 = t; i0 < span; i0 += 256 * 8) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = i0 + 256 * u;
      const int64_t idx = i_lo + i;
      v[u] = pf_zero(T{});
      if (i < span) {
        if (idx < 0) {
          if (idx >= -(int64_t) HW) v[u] = hist[HW + idx];
        } else if (idx < n) {
          v[u] = x[idx];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = i0 + 256 * u;
      if (i < span) xs[i] = v[u];
    }
  }
  __syncthreads();
  for (int64_t o = o0 + t; o < o1; o += 256) {
    const int64_t grp = o / NPH;
    const int ph = (int) (o - grp * NPH);
    const int b = (int) (grp * stride + start - i_lo);                   // staged index of the newest sample
    const float *gp = gs + ph * W;
    T acc = pf_zero(T{});
    for (int k = W - 1; k >= 0; k--) acc = pf_mac(acc, gp[k], xs[b - k]);   // oldest sample first, like the reference
    y[o] = acc;
  }
}
// new_hist = last HW samples of (old_hist ++ x[0..n))
template <typename T>
__global__ void pf_hist_update_kernel(const T *__restrict__ x, const T *__restrict__ old_hist, T *__restrict__ new_hist, int HW,
                                      int64_t n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HW) return;
  const int64_t src = n - HW + i;
  new_hist[i] = src >= 0 ? x[src] : old_hist[HW + src];
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

// The same recursion for Ky <= KMAX with the output memory in registers and the data staged
// through LDS in coalesced tiles (the kernel above walks global memory sample by sample:
// microseconds per sample).  Same operations in the same order: bit-identical results.
constexpr int RII_TILE = 4096;      // floats per tile (all channels)
template <int KMAX>
__global__ __launch_bounds__(256) void rii_recursive_tiled_kernel(float *__restrict__ y, const float *__restrict__ denom,
                                                                  int Ky, float *__restrict__ hist, int nch, int64_t n)
{
  __shared__ float tile[RII_TILE];
  const int tid = threadIdx.x;
  const int64_t nfl = n * nch;
  float h[KMAX], d[KMAX];
  const float d0 = denom[0];
#pragma unroll
  for (int k = 0; k < KMAX; k++) {
    d[k] = k < Ky ? denom[k + 1] : 0.f;
    h[k] = (tid < nch && k < Ky) ? hist[tid * Ky + k] : 0.f;
  }
  for (int64_t base = 0; base < nfl; base += RII_TILE) {
    const int cnt = (int) min((int64_t) RII_TILE, nfl - base);       // a multiple of nch
    for (int i = tid; i < cnt; i += 256) tile[i] = y[base + i];
    __syncthreads();
    if (tid < nch) {
      for (int j = tid; j < cnt; j += nch) {
        float somme = tile[j];
#pragma unroll
        for (int k = 0; k < KMAX; k++)
          if (k < Ky) somme -= h[k] * d[k];
        const float o = somme / d0;
        tile[j] = o;
#pragma unroll
        for (int k = KMAX - 1; k > 0; k--) h[k] = h[k - 1];
        h[0] = o;
      }
    }
    __syncthreads();
    for (int i = tid; i < cnt; i += 256) y[base + i] = tile[i];
    __syncthreads();
  }
  if (tid < nch) {
#pragma unroll
    for (int k = 0; k < KMAX; k++)
      if (k < Ky) hist[tid * Ky + k] = h[k];
  }
}

}  // namespace tsdgpu

using namespace tsdgpu;

struct tsdgpu_polyfir {
  int kind = 0, data_type = 0, R = 1, K = 0;
  std::vector<tsdgpu_fir *> fir;    // 1 (decimators) or R (upsampler phases)
  int cnt = 0;                      // inputs seen since the last kept output (decimators / pick)
  // fused path (polyfir_fused_kernel): taps [NPH][W] in FIR convention, history of the last HW inputs
  bool fused = false;
  int NPH = 1, W = 0, HW = 0, cur = 0, TO = 0;
  float *d_g = nullptr;
  void *d_hist[2] = {nullptr, nullptr};
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

// the fused kernel serves a stage when its taps and the input span of PF_TO outputs fit in LDS
int fused_setup(tsdgpu_polyfir *p, const std::vector<float> &g, int NPH, int W, int stride)
{
  // outputs per workgroup: as many as keep the staged input span within PF_MAX_SPAN samples
  // ... and preferably within ~PF_SPAN_TARGET samples (17 KiB: several workgroups per CU overlap their load and compute phases)
  int64_t to = std::min<int64_t>(PF_TO, ((int64_t) (PF_MAX_SPAN - W) / stride - 1) * NPH);
  to = std::min<int64_t>(to, std::max<int64_t>(256, (int64_t) (PF_SPAN_TARGET / stride) * NPH));
  if (getenv("TSDGPU_POLY_COMPOSED") || to < 256 || (size_t) NPH * W > 4096 || W < 1) return TSDGPU_OK;
  p->TO = (int) (to / 256 * 256);
  p->NPH = NPH;
  p->W = W;
  p->HW = std::max(W - 1, 1);
  const size_t hb = (size_t) p->HW * dtype_size(p->data_type);
  if (hipMalloc((void **) &p->d_g, g.size() * sizeof(float)) != hipSuccess || hipMalloc(&p->d_hist[0], hb) != hipSuccess ||
      hipMalloc(&p->d_hist[1], hb) != hipSuccess)
    return set_err(TSDGPU_ERR_HIP, "polyfir_create: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
  if (hipMemcpy(p->d_g, g.data(), g.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(p->d_hist[0], 0, hb) != hipSuccess || hipMemset(p->d_hist[1], 0, hb) != hipSuccess)
    return set_err(TSDGPU_ERR_HIP, "polyfir_create: upload failed: %s", hipGetErrorString(hipGetLastError()));
  (void) hipFuncSetAttribute((const void *) polyfir_fused_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void) hipFuncSetAttribute((const void *) polyfir_fused_kernel<float2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void) hipGetLastError();
  p->fused = true;
  return TSDGPU_OK;
}

template <typename T>
int fused_step(tsdgpu_polyfir *p, const void *dx, void *dy, int stride, int64_t start, int64_t n, int64_t nout, hipStream_t st)
{
  if (nout > 0) {
    const int64_t span = (int64_t) (p->TO / p->NPH + 1) * stride + p->W;
    const size_t lds = (size_t) ((p->NPH * p->W + 3) & ~3) * sizeof(float) + (size_t) span * sizeof(T);
    hipLaunchKernelGGL(polyfir_fused_kernel<T>, dim3((unsigned) cdiv(nout, p->TO)), dim3(256), lds, st, (const T *) dx,
                       (const T *) p->d_hist[p->cur], (T *) dy, p->d_g, p->NPH, p->W, stride, start, p->HW, n, nout, p->TO);
    TSD_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(pf_hist_update_kernel<T>, dim3((unsigned) cdiv(p->HW, 64)), dim3(64), 0, st, (const T *) dx,
                     (const T *) p->d_hist[p->cur], (T *) p->d_hist[p->cur ^ 1], p->HW, n);
  TSD_HIP(hipGetLastError());
  p->cur ^= 1;
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
    if (!rc) rc = fused_setup(p, h, 1, ntaps, R);
  } else if (kind == TSDGPU_POLY_UPS) {
    // coefs = c * R, zero-padded to a multiple of R (polyphase.cc:259-270); phase i correlates
    // the K/R-sample window with coefs[(R-1-i) + j*R]
    std::vector<float> c((size_t) ntaps);
    for (int i = 0; i < ntaps; i++) c[i] = taps_host[i] * (float) R;
    while (c.size() % (size_t) R) c.push_back(0.f);
    const int W = (int) c.size() / R;
    std::vector<float> gall;
    for (int i = 0; i < R && !rc; i++) {
      std::vector<float> g((size_t) W);
      for (int k = 0; k < W; k++) g[k] = c[(size_t) (R - 1 - i) + (size_t) (W - 1 - k) * R];
      tsdgpu_fir *f = nullptr;
      rc = tsdgpu_fir_create(&f, data_type, TSDGPU_F32, g.data(), W, TSDGPU_FIR_AUTO);
      if (!rc) p->fir.push_back(f);
      gall.insert(gall.end(), g.begin(), g.end());
    }
    if (!rc) rc = fused_setup(p, gall, R, W, 1);
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
  } else if (p->fused && p->kind == TSDGPU_POLY_UPS) {
    // output n*R + i = phase i of input n: group = input index, newest sample = that input
    rc = cplx ? fused_step<float2>(p, dx, dy, 1, 0, n, nout, st) : fused_step<float>(p, dx, dy, 1, 0, n, nout, st);
    if (rc) return rc;
  } else if (p->fused) {
    // kept outputs sit at local inputs R-1-cnt, then every R
    const int64_t start = p->R - 1 - p->cnt;
    rc = cplx ? fused_step<float2>(p, dx, dy, p->R, start, n, nout, st) : fused_step<float>(p, dx, dy, p->R, start, n, nout, st);
    if (rc) return rc;
    p->cnt = (int) ((p->cnt + n) % p->R);
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
  if (p->fused) TSD_HIP(hipMemset(p->d_hist[p->cur], 0, (size_t) p->HW * dtype_size(p->data_type)));
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
  if (p->d_g) (void) hipFree(p->d_g);
  for (void *h : p->d_hist)
    if (h) (void) hipFree(h);
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
#define RII_LAUNCH(KM) hipLaunchKernelGGL(rii_recursive_tiled_kernel<KM>, dim3(1), dim3(256), 0, st, (float *) dy, r->d_denom, r->Ky, r->d_hist, nch, n)
  if (r->Ky <= 4) RII_LAUNCH(4);
  else if (r->Ky <= 8) RII_LAUNCH(8);
  else if (r->Ky <= 16) RII_LAUNCH(16);
  else if (r->Ky <= 32) RII_LAUNCH(32);
  else hipLaunchKernelGGL(rii_recursive_kernel, dim3(1), dim3(64), 0, st, (float *) dy, r->d_denom, r->Ky, r->d_hist, nch, n);
#undef RII_LAUNCH
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
