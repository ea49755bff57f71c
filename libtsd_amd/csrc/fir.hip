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
//   * every lane segment (64 B of samples) is followed by 16 B of padding: all LDS traffic is
//     conflict-free ds_read_b128 / ds_write_b128; interior tiles load and store global memory
//     with coalesced 16-B accesses (outputs are staged back through LDS).
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

// ------------------------------------------------------------------ direct kernel
// LDS image: the lane's R samples (64 B) form one segment, segments are 80 B apart (16 B of
// padding): 80 B = 5 x 16 B, so the 16 lanes a ds_read_b128 / ds_write_b128 services together
// hit 16 distinct 16-B slots -- every LDS access of the kernel is a conflict-free b128.
// Sample s of the tile (global index tile0 - KP + s) lives at pos(s) = q + (q / R) * P, q = s-1
// (sample 0 is never needed), which makes every lane window start on a segment boundary.
template <typename T, typename TC, int R, int THREADS, bool FAST>
__global__ __launch_bounds__(THREADS) void fir_direct_kernel(
    const T *__restrict__ x, const T *__restrict__ hist, T *__restrict__ y,
    const TC *__restrict__ hrev, int KP, int64_t n, const T *__restrict__ old_hist, T *__restrict__ new_hist, int HL)
{
  constexpr int TILE = THREADS * R;
  // the workgroup after the last tile writes the stream's new history (the last HL samples of old history ++ x) into the
  // handle's OTHER history buffer: one launch per step instead of two (small blocks are launch-bound)
  if (blockIdx.x == gridDim.x - 1) {
    for (int i = threadIdx.x; i < HL; i += THREADS) {
      const int64_t g = n - HL + i;
      new_hist[i] = g < 0 ? old_hist[HL + g] : x[g];
    }
    return;
  }
  constexpr int VEC = 16 / (int) sizeof(T);          // samples per 16 B
  constexpr int P = VEC;                             // pad samples per segment
  constexpr int SP = R + P;                          // segment pitch in samples (80 B)
  static_assert(R * sizeof(T) == 64, "one lane segment is 64 bytes");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *L = reinterpret_cast<T *>(smem_raw);

  const int64_t tile0 = (int64_t) blockIdx.x * TILE;
  const int H = KP;
  const int total = TILE + H;                        // samples 0 .. total-1 (sample 0 unused)
  const bool interior = FAST && tile0 - H >= 0 && tile0 + TILE + VEC <= n;      // (the chunked load reads up to VEC - 1 samples past the tile)

  if (interior) {
    // 16-B global loads that start ONE sample into the tile's range: chunk c = samples [c*VEC + 1, c*VEC + 1 + VEC), i.e.
    // positions q = c*VEC .. c*VEC + VEC - 1 -- a whole 16-B unit of one segment, so a chunk goes to LDS as one ds_write_b128
    // (loaded from the range's first sample the chunks straddled the units: two ds_write_b64 each, 2-way bank conflicts).
    // The loads are sizeof(T) off the 16-B grid, which global memory does not mind.
    struct __attribute__((aligned(4))) f4u { float x, y, z, w; };
    const f4u *xs = reinterpret_cast<const f4u *>(x + (tile0 - H + 1));
    // four loads in flight per thread (the tile is 4..5 chunks per thread: one load per
    // iteration exposed the HBM latency that many times per workgroup)
    const int nchunks = (total - 1 + VEC - 1) / VEC;
    for (int c0 = threadIdx.x; c0 < nchunks; c0 += 4 * THREADS) {
      f4u q4[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int c = c0 + u * THREADS;
        if (c < nchunks) q4[u] = xs[c];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int c = c0 + u * THREADS;
        if (c < nchunks) {
          const int q = c * VEC;
          *reinterpret_cast<float4 *>(L + q + (q / R) * P) = make_float4(q4[u].x, q4[u].y, q4[u].z, q4[u].w);
        }
      }
    }
  } else {
    for (int s = threadIdx.x + 1; s < total; s += THREADS) {
      const int64_t g = tile0 - H + (int64_t) s;
      T v = zero_of(T{});
      if (g < 0)
        v = hist[H + g];
      else if (g < n)
        v = x[g];
      const int q = s - 1;
      L[q + (q / R) * P] = v;
    }
  }
  __syncthreads();

  // lane window: w[i] = sample t*R + 1 + i = Lw[(i / R) * SP + i % R], out[r] = sum_j hrev[j] * w[r + j]
  const T *Lw = L + threadIdx.x * SP;
  T acc[R], A[R], B[R];
  auto load_seg = [&](T (&dst)[R], const T *seg) {
#pragma unroll
    for (int v4 = 0; v4 < R / VEC; v4++) {
      const float4 q4 = *reinterpret_cast<const float4 *>(seg + v4 * VEC);
      const T *e = reinterpret_cast<const T *>(&q4);
#pragma unroll
      for (int k = 0; k < VEC; k++) dst[v4 * VEC + k] = e[k];
    }
  };
#pragma unroll
  for (int r = 0; r < R; r++) acc[r] = zero_of(T{});
  load_seg(A, Lw);

  const int nchunk = KP / R;  // even by construction
#pragma unroll 2
  for (int c = 0; c < nchunk; c += 2) {
    const TC *h0 = hrev + c * R;
    load_seg(B, Lw + (c + 1) * SP);
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
    // by two segments so the read is in bounds and the values are never used
    load_seg(A, Lw + (c + 2) * SP);
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

  if (interior) {
    // outputs go back through LDS so that every global store is a coalesced 16-B access of whole lines.  (Round 4 stored a lane's
    // 64 contiguous bytes straight from its registers -- four 16-B stores, a wave instruction covering 32 lines half a line at a
    // time: nothing gained at 127 taps (0.3725 against 0.3727 ms), and the short filters this kernel actually serves lost a fifth:
    // 0.19 -> 0.22 ms at 31 taps, 0.10 -> 0.14 ms on real data.)
    __syncthreads();
    T *Lo = L + threadIdx.x * SP;
#pragma unroll
    for (int v4 = 0; v4 < R / VEC; v4++) {
      float4 q4;
      T *e = reinterpret_cast<T *>(&q4);
#pragma unroll
      for (int k = 0; k < VEC; k++) e[k] = acc[v4 * VEC + k];
      *reinterpret_cast<float4 *>(Lo + v4 * VEC) = q4;
    }
    __syncthreads();
    float4 *ys = reinterpret_cast<float4 *>(y + tile0);
    for (int c = threadIdx.x; c < TILE / VEC; c += THREADS)
      ys[c] = *reinterpret_cast<const float4 *>(L + (c / (R / VEC)) * SP + (c % (R / VEC)) * VEC);
  } else {
    const int64_t o0 = tile0 + (int64_t) threadIdx.x * R;
#pragma unroll
    for (int r = 0; r < R; r++)
      if (o0 + r < n) y[o0 + r] = acc[r];
  }
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
  const size_t nseg = (size_t) (TILE + KP) / R + 3;                 // + over-read segments
  const size_t lds = nseg * 80;
  const int64_t tiles = cdiv(n, TILE);
  if (tiles > 0x7fffffff) return set_err(TSDGPU_ERR_UNSUPPORTED, "fir: n too large for one launch");
  // 16-byte global accesses need 16-byte aligned buffers (tile starts are multiples of 64 B)
  const bool fast = (((uintptr_t) x | (uintptr_t) y) & 15) == 0;
  const T *hp = (const T *) fir_hist_read(f) + (f->HL - KP);
  const T *oldh = (const T *) fir_hist_read(f);
  T *newh = (T *) f->hist[f->cur ^ 1];
  if (fast)
    hipLaunchKernelGGL((fir_direct_kernel<T, TC, R, THREADS, true>), dim3((unsigned) tiles + 1), dim3(THREADS), lds, st,
                       (const T *) x, hp, (T *) y, (const TC *) f->d_hrev, KP, n, oldh, newh, f->HL);
  else
    hipLaunchKernelGGL((fir_direct_kernel<T, TC, R, THREADS, false>), dim3((unsigned) tiles + 1), dim3(THREADS), lds, st,
                       (const T *) x, hp, (T *) y, (const TC *) f->d_hrev, KP, n, oldh, newh, f->HL);
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}

// (the launch also writes the new history into the other buffer: the caller flips f->cur)
int fir_direct_step(tsdgpu_fir *f, const void *x, void *y, int64_t n, hipStream_t st)
{
  int rc;
  if (f->data_type == TSDGPU_F32) rc = launch_direct<float, float, 16>(f, x, y, n, st);
  else if (f->tap_type == TSDGPU_F32) rc = launch_direct<float2, float, 8>(f, x, y, n, st);
  else rc = launch_direct<float2, float2, 8>(f, x, y, n, st);
  if (!rc) f->cur ^= 1;
  return rc;
}

int fir_update_history(tsdgpu_fir *f, const void *x, int64_t n, hipStream_t st)
{
  const int H = f->HL;
  const int nxt = f->cur ^ 1;
  const int blocks = (int) cdiv(H, 256);
  if (f->data_type == TSDGPU_F32)
    hipLaunchKernelGGL(fir_hist_update_kernel<float>, dim3(blocks), dim3(256), 0, st,
                       (const float *) x, (const float *) fir_hist_read(f), (float *) f->hist[nxt], H, n);
  else
    hipLaunchKernelGGL(fir_hist_update_kernel<float2>, dim3(blocks), dim3(256), 0, st,
                       (const float2 *) x, (const float2 *) fir_hist_read(f), (float2 *) f->hist[nxt], H, n);
  TSD_HIP(hipGetLastError());
  f->cur = nxt;
  return TSDGPU_OK;
}

__global__ void fir_accumulate_kernel(float *__restrict__ y, const float *__restrict__ t, int64_t nfl)
{
  const int64_t i = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < nfl) {
    float4 a = *reinterpret_cast<const float4 *>(y + i);
    const float4 b = *reinterpret_cast<const float4 *>(t + i);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    *reinterpret_cast<float4 *>(y + i) = a;
  } else {
    for (int64_t k = i; k < nfl; k++) y[k] += t[k];
  }
}

// More than 12289 taps (the reference's FiltreRIF has no tap limit): partitioned convolution.  The taps are
// cut in segments of part_S; segment p is an ordinary filter (long-filter overlap-save plan) applied to the
// input delayed by p * part_S samples, which it reads straight out of [history ++ x]; the partial sums are
// added up in segment order.  dx, dy: device pointers (dy may be dx).
int fir_partitioned_step(tsdgpu_fir *f, const void *dx, void *dy, int64_t n, hipStream_t st)
{
  const size_t sz = dtype_size(f->data_type);
  const int64_t H = f->K - 1;
  const int64_t pad = (16 - H % 16) % 16;                       // x starts on a 128-B boundary inside X (part_S is a multiple of 16)
  int rc = f->part_x.reserve((size_t) (pad + H + n) * sz + 64);
  if (!rc && f->parts.size() > 1) rc = f->part_t.reserve((size_t) n * sz + 64);
  if (rc) return rc;
  char *X = (char *) f->part_x.p + (size_t) pad * sz;
  const char *hist = (const char *) f->hist[f->cur] + (size_t) (f->HL - H) * sz;
  TSD_HIP(hipMemcpyAsync(X, hist, (size_t) H * sz, hipMemcpyDeviceToDevice, st));
  TSD_HIP(hipMemcpyAsync(X + (size_t) H * sz, dx, (size_t) n * sz, hipMemcpyDeviceToDevice, st));
  for (size_t p = 0; p < f->parts.size(); p++) {
    tsdgpu_fir *c = f->parts[p];
    const int64_t start = H - (int64_t) p * f->part_S;          // element of X that is this segment's x[0]
    if (c->K > 1) {
      rc = tsdgpu_fir_set_history(c, X + (size_t) (start - (c->K - 1)) * sz, st);
      if (rc) return rc;
    }
    void *dst = p == 0 ? dy : f->part_t.p;
    rc = tsdgpu_fir_step(c, X + (size_t) start * sz, dst, n, st);
    if (rc) return rc;
    if (p > 0) {
      const int64_t nfl = n * (int64_t) (sz / 4);
      hipLaunchKernelGGL(fir_accumulate_kernel, dim3((unsigned) cdiv(cdiv(nfl, 4), 256)), dim3(256), 0, st, (float *) dy,
                         (const float *) f->part_t.p, nfl);
      TSD_HIP(hipGetLastError());
    }
  }
  // history <- the last K-1 samples of [history ++ x]
  const int nxt = f->cur ^ 1;
  TSD_HIP(hipMemcpyAsync((char *) f->hist[nxt] + (size_t) (f->HL - H) * sz, X + (size_t) n * sz, (size_t) H * sz, hipMemcpyDeviceToDevice, st));
  f->cur = nxt;
  return TSDGPU_OK;
}

// Capturable handles keep the stream's history in hist[0] at every step boundary: a step reads hist[0],
// its kernels write the new history to hist[1] as always, and this copies it back (HL samples, ~1 KB).
// Every launch of a step then has the same arguments as the previous one, so a step of fixed (x, y, n) can
// be captured into a hipGraph and replayed as a stream.
int fir_settle_history(tsdgpu_fir *f, hipStream_t st)
{
  if (!f->capturable || f->cur == 0) return TSDGPU_OK;
  const int rc = device_copy_small(f->hist[0], f->hist[1], (size_t) f->HL * dtype_size(f->data_type), st);
  if (rc) return rc;
  f->cur = 0;
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
  const int R = data_type == TSDGPU_F32 ? 16 : 8;
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
    // ONE allocation (the reversed taps, then the two zeroed histories, each on a 256-byte boundary) and ONE upload of its
    // host image: what a one-shot filtrer() pays per call (three allocations, a copy, two memsets and a synchronisation before)
    const size_t tb = (hrev.size() + 255) / 256 * 256, hb = (hbytes + 255) / 256 * 256;
    {
      std::vector<char> image(tb + 2 * hb, 0);
      memcpy(image.data(), hrev.data(), hrev.size());
      if (hipMalloc(&f->d_hrev, image.size()) != hipSuccess) {
        rc = set_err(TSDGPU_ERR_HIP, "fir_create: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
        break;
      }
      if (hipMemcpy(f->d_hrev, image.data(), image.size(), hipMemcpyHostToDevice) != hipSuccess) {
        rc = set_err(TSDGPU_ERR_HIP, "fir_create: upload failed: %s", hipGetErrorString(hipGetLastError()));
        break;
      }
    }
    f->hist[0] = (char *) f->d_hrev + tb;
    f->hist[1] = (char *) f->d_hrev + tb + hb;
    if (ntaps > 12289) {
      // beyond the long-filter plan (and beyond what the direct kernel can stage in LDS): partitioned
      f->method = TSDGPU_FIR_OVERLAP_SAVE;
      f->part_S = 8192;
      for (int o = 0; o < ntaps && !rc; o += f->part_S) {
        tsdgpu_fir *c = nullptr;
        rc = tsdgpu_fir_create(&c, data_type, tap_type, (const char *) taps_host + (size_t) o * tsz, std::min(f->part_S, ntaps - o), TSDGPU_FIR_AUTO);
        if (!rc) f->parts.push_back(c);
      }
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
  if (bytes >= PIPE_MIN_BYTES && host_pipe_enabled() && !is_device_ptr(x) && !is_device_ptr(y)) {
    // large host vectors: chunked H2D / kernel / D2H pipeline (the history carries from chunk to chunk)
    return pipelined_host_step(x, y, n, dtype_size(f->data_type), st,
                               [f](const void *cx, void *cy, int64_t cnt, hipStream_t s) { return tsdgpu_fir_step(f, cx, cy, cnt, s); });
  }

  const void *dx = nullptr;
  void *dy = nullptr;
  bool staged = false;
  int rc = stage_in(x, bytes, f->in_stage, st, &dx);
  if (rc) return rc;
  rc = stage_out(y, bytes, f->out_stage, &dy, &staged);
  if (rc) return rc;
  if (!f->parts.empty()) {
    rc = fir_partitioned_step(f, dx, dy, n, st);
    if (rc) return rc;
    rc = fir_settle_history(f, st);
    if (rc) return rc;
    return finish_out(y, bytes, dy, staged, st);
  }
  if (dx == dy) {
    // in-place on device (allowed by the reference, filtre-rt.cc:76-80): tiles read their
    // neighbours' inputs, so filter from a private copy
    rc = f->in_stage.reserve(bytes);
    if (rc) return rc;
    TSD_HIP(hipMemcpyAsync(f->in_stage.p, dx, bytes, hipMemcpyDeviceToDevice, st));
    dx = f->in_stage.p;
  }
  if (f->method == TSDGPU_FIR_OVERLAP_SAVE && f->ols_long) {
    rc = ols_long_step(f, dx, dy, n, st);
  } else if (f->method == TSDGPU_FIR_OVERLAP_SAVE) {
    rc = ols_step(f, dx, dy, n, st);             // history update folded into the launch
  } else {
    rc = fir_direct_step(f, dx, dy, n, st);      // history update folded into the launch
  }
  if (rc) return rc;
  rc = fir_settle_history(f, st);
  if (rc) return rc;
  return finish_out(y, bytes, dy, staged, st);
}

// (-1: no tsdgpu_fir_step_after on this handle -- NULL, or the partitioned plan of more than 12289 taps, whose partial
// filters each hold a delay line of their own: the callers fall back to set_history + step)
int tsdgpu_fir_lead(const tsdgpu_fir *f) { return (f && f->parts.empty()) ? f->HL : -1; }

int tsdgpu_fir_step_after(tsdgpu_fir *f, const void *x, void *y, int64_t n, int64_t lead, void *stream)
{
  TSD_CHECK(f != nullptr, "fir_step_after: NULL handle");
  TSD_CHECK(x != nullptr && y != nullptr && x != y, "fir_step_after: NULL or aliased buffers");
  TSD_CHECK(is_device_ptr(x) && is_device_ptr(y), "fir_step_after: device buffers expected");
  TSD_CHECK(f->parts.empty(), "fir_step_after: not available on the partitioned plan (more than 12289 taps)");
  TSD_CHECK(lead >= f->HL && lead <= n, "fir_step_after: lead %lld outside [%d, n = %lld]", (long long) lead, f->HL, (long long) n);
  const size_t sz = dtype_size(f->data_type);
  if (lead == n) return TSDGPU_OK;
  f->hist_ext = (const char *) x + (size_t) (lead - f->HL) * sz;
  const int rc = tsdgpu_fir_step(f, (const char *) x + (size_t) lead * sz, (char *) y + (size_t) lead * sz, n - lead, stream);
  f->hist_ext = nullptr;
  return rc;
}

int tsdgpu_fir_set_capturable(tsdgpu_fir *f, int on)
{
  TSD_CHECK(f != nullptr, "fir_set_capturable: NULL handle");
  hipStream_t st = nullptr;
  f->capturable = on != 0;
  const int rc = fir_settle_history(f, st);
  if (rc) return rc;
  TSD_HIP(hipStreamSynchronize(st));
  for (tsdgpu_fir *c : f->parts) {
    const int rc2 = tsdgpu_fir_set_capturable(c, on);
    if (rc2) return rc2;
  }
  return TSDGPU_OK;
}

int tsdgpu_fir_reset(tsdgpu_fir *f)
{
  TSD_CHECK(f != nullptr, "fir_reset: NULL handle");
  const size_t hbytes = (size_t) f->HL * dtype_size(f->data_type);
  TSD_HIP(hipMemset(f->hist[f->cur], 0, hbytes));
  TSD_HIP(hipStreamSynchronize(nullptr));      // see tsdgpu_sos_reset
  return TSDGPU_OK;
}

int tsdgpu_fir_reset_on(tsdgpu_fir *f, void *stream)
{
  TSD_CHECK(f != nullptr, "fir_reset: NULL handle");
  TSD_HIP(hipMemsetAsync(f->hist[f->cur], 0, (size_t) f->HL * dtype_size(f->data_type), (hipStream_t) stream));
  return TSDGPU_OK;
}

int tsdgpu_fir_get_history(tsdgpu_fir *f, void *dst, void *stream)
{
  TSD_CHECK(f != nullptr && dst != nullptr, "fir_get_history: NULL argument");
  if (f->K < 2) return TSDGPU_OK;
  hipStream_t st = (hipStream_t) stream;
  const size_t sz = dtype_size(f->data_type);
  const char *src = (const char *) f->hist[f->cur] + (size_t) (f->HL - (f->K - 1)) * sz;
  if (is_device_ptr(dst)) return device_copy_small(dst, src, (size_t) (f->K - 1) * sz, st);
  TSD_HIP(hipMemcpyAsync(dst, src, (size_t) (f->K - 1) * sz, hipMemcpyDeviceToHost, st));
  TSD_HIP(hipStreamSynchronize(st));
  return TSDGPU_OK;
}

int tsdgpu_fir_set_history(tsdgpu_fir *f, const void *src, void *stream)
{
  TSD_CHECK(f != nullptr && src != nullptr, "fir_set_history: NULL argument");
  if (f->K < 2) return TSDGPU_OK;
  hipStream_t st = (hipStream_t) stream;
  const size_t sz = dtype_size(f->data_type);
  char *dst = (char *) f->hist[f->cur] + (size_t) (f->HL - (f->K - 1)) * sz;
  if (is_device_ptr(src)) return device_copy_small(dst, src, (size_t) (f->K - 1) * sz, st);
  TSD_HIP(hipMemcpyAsync(dst, src, (size_t) (f->K - 1) * sz, hipMemcpyHostToDevice, st));
  TSD_HIP(hipStreamSynchronize(st));
  return TSDGPU_OK;
}

int tsdgpu_fir_method_used(const tsdgpu_fir *f) { return f ? f->method : -1; }

int tsdgpu_fir_destroy(tsdgpu_fir *f)
{
  if (!f) return TSDGPU_OK;
  ols_plan_destroy(f);
  for (tsdgpu_fir *c : f->parts) tsdgpu_fir_destroy(c);
  f->part_x.release();
  f->part_t.release();
  if (f->d_hrev) (void) hipFree(f->d_hrev);          // (the histories live in the same allocation)
  f->in_stage.release();
  f->out_stage.release();
  delete f;
  return TSDGPU_OK;
}

}  // extern "C"
