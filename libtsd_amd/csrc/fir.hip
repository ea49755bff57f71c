// fir.hip -- streaming FIR for gfx950: direct sliding-dot-product kernel + C ABI.
//
// Stands behind FiltreRIF<T,Tc>::step / filtre_rif<Tc,T>  (libtsd core/src/filtrage/
// filtre-rt.cc:53-109,171-175).  y[n] = sum_k h[k] x[n-k]; the accumulation runs from the
// oldest sample (tap K-1) to the newest (tap 0), the reference's order (:84,98-104).
//
// Direct kernel layout (per workgroup of 256 lanes = 4 wave64):
//   * one tile = 256*R consecutive outputs; the tile's inputs plus a KP-sample halo are
//     staged ONCE in LDS with coalesced global loads (halo of the first tile comes from the
//     handle's history buffer = the reference's delay line);
//   * each lane owns R consecutive outputs and slides a 2R-sample register window over its
//     KP+R-1 inputs: every LDS sample read feeds R multiply-adds, taps are wave-uniform
//     scalar loads (SGPR operands);
//   * LDS rows are padded by one sample every R so that the lane stride (R+1 samples) is
//     odd in 4/8-byte bank units: conflict-free ds_read_b32/b64.
// The kernel is fp32-VALU bound (4*K flop per complex sample with real taps), not HBM
// bound; the overlap-save path (ols.hip) is the HBM-roofline candidate for long filters.
#include "common.hpp"
#include "fir_internal.hpp"
#include <cstdlib>

namespace tsdgpu {

// ------------------------------------------------------------------ arithmetic helpers
__device__ __forceinline__ float zero_of(float) { return 0.f; }
__device__ __forceinline__ float2 zero_of(float2) { return make_float2(0.f, 0.f); }

// acc += x * h for the three (data, tap) combinations of the reference
__device__ __forceinline__ float mac(float acc, float x, float h) { return fmaf(x, h, acc); }
__device__ __forceinline__ float2 mac(float2 acc, float2 x, float h)
{
  return make_float2(fmaf(x.x, h, acc.x), fmaf(x.y, h, acc.y));
}
__device__ __forceinline__ float2 mac(float2 acc, float2 x, float2 h)
{
  // (xr + j xi)(hr + j hi), limited-range product as in the reference build
  float re = fmaf(x.x, h.x, acc.x);
  re = fmaf(-x.y, h.y, re);
  float im = fmaf(x.x, h.y, acc.y);
  im = fmaf(x.y, h.x, im);
  return make_float2(re, im);
}

template <int R> __device__ __forceinline__ unsigned padded(unsigned s) { return s + s / R; }

// ------------------------------------------------------------------ direct kernel
template <typename T, typename TC, int R, int THREADS>
__global__ __launch_bounds__(THREADS) void fir_direct_kernel(
    const T *__restrict__ x, const T *__restrict__ hist, T *__restrict__ y,
    const TC *__restrict__ hrev, int KP, int64_t n)
{
  constexpr int TILE = THREADS * R;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *L = reinterpret_cast<T *>(smem_raw);

  const int64_t tile0 = (int64_t) blockIdx.x * TILE;
  const int H = KP;
  const int total = TILE + H;

  // stage tile + halo: L[padded(s)] = x_ext[tile0 - H + s]
  for (unsigned s = threadIdx.x; s < (unsigned) total; s += THREADS) {
    const int64_t g = tile0 - H + (int64_t) s;
    T v = zero_of(T{});
    if (g < 0)
      v = hist[H + g];
    else if (g < n)
      v = x[g];
    L[padded<R>(s)] = v;
  }
  __syncthreads();

  // lane window: w[i] = L[padded(t*R + 1 + i)], out[r] = sum_j hrev[j] * w[r + j].
  // With i = c*R + r:  padded(t*R + 1 + i) = t*(R+1) + c*(R+1) + 1 + r + (r == R-1),
  // i.e. one per-lane base plus compile-time offsets.
  const T *Lw = L + threadIdx.x * (R + 1);
  T acc[R], A[R], B[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    acc[r] = zero_of(T{});
    A[r] = Lw[1 + r + (r == R - 1)];
  }

  const int nchunk = KP / R;  // even by construction
  for (int c = 0; c < nchunk; c += 2) {
    const TC *h0 = hrev + c * R;
    const T *Lc = Lw + c * (R + 1);
#pragma unroll
    for (int r = 0; r < R; r++) B[r] = Lc[(R + 1) + 1 + r + (r == R - 1)];
#pragma unroll
    for (int jj = 0; jj < R; jj++) {
      const TC hv = h0[jj];
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int idx = r + jj;
        acc[r] = mac(acc[r], idx < R ? A[idx] : B[idx - R], hv);
      }
    }
    // the last refill reads past the lane's window; the staging area is over-allocated
    // by 2R samples so the read is in bounds and the values are never used
#pragma unroll
    for (int r = 0; r < R; r++) A[r] = Lc[2 * (R + 1) + 1 + r + (r == R - 1)];
#pragma unroll
    for (int jj = 0; jj < R; jj++) {
      const TC hv = h0[R + jj];
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int idx = r + jj;
        acc[r] = mac(acc[r], idx < R ? B[idx] : A[idx - R], hv);
      }
    }
  }

  const int64_t o0 = tile0 + (int64_t) threadIdx.x * R;
#pragma unroll
  for (int r = 0; r < R; r++)
    if (o0 + r < n) y[o0 + r] = acc[r];
}

// new_hist = last H samples of (old_hist ++ x[0..n))
template <typename T>
__global__ void fir_hist_update_kernel(const T *__restrict__ x, const T *__restrict__ old_hist,
                                       T *__restrict__ new_hist, int H, int64_t n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H) return;
  const int64_t g = n - H + i;
  new_hist[i] = g < 0 ? old_hist[H + g] : x[g];
}

template <typename T, typename TC, int R>
static int launch_direct(const tsdgpu_fir *f, const void *x, void *y, int64_t n, hipStream_t st)
{
  constexpr int THREADS = 256;
  constexpr int TILE = THREADS * R;
  const int KP = f->KP;
  const int total = TILE + KP + 2 * R;
  const size_t lds = (size_t) (total + total / R + 2) * sizeof(T);
  const int64_t tiles = cdiv(n, TILE);
  if (tiles > 0x7fffffff) return set_err(TSDGPU_ERR_UNSUPPORTED, "fir: n too large for one launch");
  hipLaunchKernelGGL((fir_direct_kernel<T, TC, R, THREADS>), dim3((unsigned) tiles), dim3(THREADS),
                     lds, st, (const T *) x, (const T *) f->hist[f->cur] + (f->HL - KP), (T *) y,
                     (const TC *) f->d_hrev, KP, n);
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}

int fir_direct_step(tsdgpu_fir *f, const void *x, void *y, int64_t n, hipStream_t st)
{
  if (f->data_type == TSDGPU_F32) return launch_direct<float, float, 16>(f, x, y, n, st);
  if (f->tap_type == TSDGPU_F32) return f->R == 16 ? launch_direct<float2, float, 16>(f, x, y, n, st) : launch_direct<float2, float, 8>(f, x, y, n, st);
  return launch_direct<float2, float2, 8>(f, x, y, n, st);
}

int fir_update_history(tsdgpu_fir *f, const void *x, int64_t n, hipStream_t st)
{
  const int H = f->HL;
  const int nxt = f->cur ^ 1;
  const int blocks = (int) cdiv(H, 256);
  if (f->data_type == TSDGPU_F32)
    hipLaunchKernelGGL(fir_hist_update_kernel<float>, dim3(blocks), dim3(256), 0, st,
                       (const float *) x, (const float *) f->hist[f->cur], (float *) f->hist[nxt], H, n);
  else
    hipLaunchKernelGGL(fir_hist_update_kernel<float2>, dim3(blocks), dim3(256), 0, st,
                       (const float2 *) x, (const float2 *) f->hist[f->cur], (float2 *) f->hist[nxt], H, n);
  TSD_HIP(hipGetLastError());
  f->cur = nxt;
  return TSDGPU_OK;
}

}  // namespace tsdgpu

using namespace tsdgpu;

extern "C" {

int tsdgpu_fir_create(tsdgpu_fir **out, int data_type, int tap_type, const void *taps_host,
                      int ntaps, int method)
{
  TSD_CHECK(out != nullptr, "fir_create: out is NULL");
  *out = nullptr;
  TSD_CHECK(taps_host != nullptr && ntaps > 0, "fir_create: need at least one tap (K > 0)");
  TSD_CHECK(data_type == TSDGPU_F32 || data_type == TSDGPU_C64, "fir_create: bad data_type %d", data_type);
  TSD_CHECK(tap_type == TSDGPU_F32 || tap_type == TSDGPU_C64, "fir_create: bad tap_type %d", tap_type);
  TSD_CHECK(!(data_type == TSDGPU_F32 && tap_type == TSDGPU_C64),
            "fir_create: complex taps on real data is not a libtsd instantiation");
  TSD_CHECK(method >= TSDGPU_FIR_AUTO && method <= TSDGPU_FIR_OVERLAP_SAVE, "fir_create: bad method %d", method);
  TSD_CHECK(ntaps <= (1 << 20), "fir_create: ntaps %d too large", ntaps);

  tsdgpu_fir *f = new tsdgpu_fir();
  f->data_type = data_type;
  f->tap_type = tap_type;
  f->K = ntaps;
  int R = data_type == TSDGPU_F32 ? 16 : 8;
  if (data_type == TSDGPU_C64 && tap_type == TSDGPU_F32 && getenv("TSDGPU_FIR_R16")) R = 16;
  f->R = R;
  f->KP = (int) (cdiv(ntaps, 2 * R) * 2 * R);
  f->HL = (int) (cdiv(f->KP, 64) * 64);

  // reversed, zero-padded taps: hrev[j] = h[KP-1-j]  (zeros on the old side)
  const size_t tsz = dtype_size(tap_type);
  std::vector<char> hrev((size_t) f->KP * tsz, 0);
  for (int k = 0; k < ntaps; k++)
    memcpy(&hrev[(size_t) (f->KP - 1 - k) * tsz], (const char *) taps_host + (size_t) k * tsz, tsz);
  f->taps_host.assign((const char *) taps_host, (const char *) taps_host + (size_t) ntaps * tsz);

  int rc = TSDGPU_OK;
  const size_t hbytes = (size_t) f->HL * dtype_size(data_type);
  do {
    if (hipMalloc(&f->d_hrev, hrev.size()) != hipSuccess ||
        hipMalloc(&f->hist[0], hbytes) != hipSuccess || hipMalloc(&f->hist[1], hbytes) != hipSuccess) {
      rc = set_err(TSDGPU_ERR_HIP, "fir_create: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
      break;
    }
    if (hipMemcpy(f->d_hrev, hrev.data(), hrev.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(f->hist[0], 0, hbytes) != hipSuccess || hipMemset(f->hist[1], 0, hbytes) != hipSuccess) {
      rc = set_err(TSDGPU_ERR_HIP, "fir_create: upload failed: %s", hipGetErrorString(hipGetLastError()));
      break;
    }
    f->method = method == TSDGPU_FIR_DIRECT ? TSDGPU_FIR_DIRECT
                : method == TSDGPU_FIR_OVERLAP_SAVE ? TSDGPU_FIR_OVERLAP_SAVE
                : (ols_preferred(f) ? TSDGPU_FIR_OVERLAP_SAVE : TSDGPU_FIR_DIRECT);
    if (f->method == TSDGPU_FIR_OVERLAP_SAVE) rc = ols_plan_create(f);
  } while (0);
  if (rc) {
    tsdgpu_fir_destroy(f);
    return rc;
  }
  *out = f;
  return TSDGPU_OK;
}

int tsdgpu_fir_step(tsdgpu_fir *f, const void *x, void *y, int64_t n, void *stream)
{
  TSD_CHECK(f != nullptr, "fir_step: NULL handle");
  TSD_CHECK(n >= 0, "fir_step: negative length");
  if (n == 0) return TSDGPU_OK;
  TSD_CHECK(x != nullptr && y != nullptr, "fir_step: NULL buffer");
  hipStream_t st = (hipStream_t) stream;
  const size_t bytes = (size_t) n * dtype_size(f->data_type);

  const void *dx = nullptr;
  void *dy = nullptr;
  bool staged = false;
  int rc = stage_in(x, bytes, f->in_stage, st, &dx);
  if (rc) return rc;
  rc = stage_out(y, bytes, f->out_stage, &dy, &staged);
  if (rc) return rc;
  if (dx == dy) {
    // in-place on device (allowed by the reference, filtre-rt.cc:76-80): tiles read their
    // neighbours' inputs, so filter from a private copy
    rc = f->in_stage.reserve(bytes);
    if (rc) return rc;
    TSD_HIP(hipMemcpyAsync(f->in_stage.p, dx, bytes, hipMemcpyDeviceToDevice, st));
    dx = f->in_stage.p;
  }
  if (f->method == TSDGPU_FIR_OVERLAP_SAVE) {
    rc = ols_step(f, dx, dy, n, st);             // history update folded into the launch
  } else {
    rc = fir_direct_step(f, dx, dy, n, st);
    if (!rc) rc = fir_update_history(f, dx, n, st);
  }
  if (rc) return rc;
  return finish_out(y, bytes, dy, staged, st);
}

int tsdgpu_fir_reset(tsdgpu_fir *f)
{
  TSD_CHECK(f != nullptr, "fir_reset: NULL handle");
  const size_t hbytes = (size_t) f->HL * dtype_size(f->data_type);
  TSD_HIP(hipMemset(f->hist[f->cur], 0, hbytes));
  return TSDGPU_OK;
}

int tsdgpu_fir_get_history(tsdgpu_fir *f, void *dst, void *stream)
{
  TSD_CHECK(f != nullptr && dst != nullptr, "fir_get_history: NULL argument");
  if (f->K < 2) return TSDGPU_OK;
  hipStream_t st = (hipStream_t) stream;
  const size_t sz = dtype_size(f->data_type);
  const char *src = (const char *) f->hist[f->cur] + (size_t) (f->HL - (f->K - 1)) * sz;
  const bool dev = is_device_ptr(dst);
  TSD_HIP(hipMemcpyAsync(dst, src, (size_t) (f->K - 1) * sz, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
  if (!dev) TSD_HIP(hipStreamSynchronize(st));
  return TSDGPU_OK;
}

int tsdgpu_fir_set_history(tsdgpu_fir *f, const void *src, void *stream)
{
  TSD_CHECK(f != nullptr && src != nullptr, "fir_set_history: NULL argument");
  if (f->K < 2) return TSDGPU_OK;
  hipStream_t st = (hipStream_t) stream;
  const size_t sz = dtype_size(f->data_type);
  char *dst = (char *) f->hist[f->cur] + (size_t) (f->HL - (f->K - 1)) * sz;
  const bool dev = is_device_ptr(src);
  TSD_HIP(hipMemcpyAsync(dst, src, (size_t) (f->K - 1) * sz, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  if (!dev) TSD_HIP(hipStreamSynchronize(st));
  return TSDGPU_OK;
}

int tsdgpu_fir_method_used(const tsdgpu_fir *f) { return f ? f->method : -1; }

int tsdgpu_fir_destroy(tsdgpu_fir *f)
{
  if (!f) return TSDGPU_OK;
  ols_plan_destroy(f);
  if (f->d_hrev) (void) hipFree(f->d_hrev);
  if (f->hist[0]) (void) hipFree(f->hist[0]);
  if (f->hist[1]) (void) hipFree(f->hist[1]);
  f->in_stage.release();
  f->out_stage.release();
  delete f;
  return TSDGPU_OK;
}

}  // extern "C"
