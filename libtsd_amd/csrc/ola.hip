// ola.hip -- the OLA frequency-domain engine behind filtre_fft() (libtsd
// core/src/fourier/fourier.cc:737-940): framing, batched FFTs, spectral processing and
// overlap-add all on the device.  The reference handles one block at a time (two FFTs of N
// per block, full-vector copies in between); here all the whole blocks of a call are framed by
// one kernel, transformed by ONE batched FFT each way, and overlap-added by one kernel.
#include "common.hpp"
#include "fft1024_wave.hpp"
#include "stockham16.hpp"
#include "fir_internal.hpp"
#include <vector>
#include <algorithm>
#include <cmath>

struct tsdgpu_ola {
  int Ne = 0, N = 0, Nz = 0;
  bool windowed = false;
  tsdgpu_fft *plan = nullptr;
  float *d_fen = nullptr;                  // Ne
  tsdgpu::cpx *d_H = nullptr;              // N, or null
  tsdgpu::cpx *d_svg = nullptr, *d_last = nullptr, *d_prev_half = nullptr, *d_rest = nullptr;   // Ne, Ne, Ne/2, Ne
  int nrest = 0;
  int64_t cnt_ech = 0;                     // fourier.cc:779: starts at -Ne/2
  int pending_blocks = -1;                 // >= 0 between analyse and synthese
  bool fuse_response = false;              // set by tsdgpu_ola_step: the forward transform applies H itself
  bool response_applied = false;
  tsdgpu::DevBuf frames, spectra, in_stage, out_stage;
  tsdgpu::cpx *d_fast = nullptr;           // in-wave paths: Ne = 512, N = 1024 without window (response in register order / N + twiddles, 3 x 1024); Ne = N = 512 windowed (512 + 2 x 1024)
  tsdgpu::cpx *d_svg_tmp = nullptr;        // Ne: the new tail, written by the last wave while the first one may still read d_svg
  tsdgpu::cpx *d_last_tmp = nullptr;       // Ne: the same for `last` (windowed run kernel)
  tsdgpu::cpx *d_bloc = nullptr;           // ONE allocation behind d_svg, d_svg_tmp, d_last, d_prev_half, d_rest
  tsdgpu::cpx *d_run = nullptr;            // any other geometry without window: response / N (N values), then W_N^i, i < N/16
};

namespace tsdgpu {
namespace {

// sample p of [rest ++ x]
__device__ __forceinline__ cpx ola_src(const cpx *rest, int nrest, const cpx *x, int64_t p)
{
  return p < nrest ? rest[p] : x[p - nrest];
}

// frames[f][i]: Nz zeros, then the block (fourier.cc:850) or, windowed, frame 2b = window *
// [second half of the previous block, first half of this one] (:885-886) and frame 2b+1 =
// window * block (:910)
__global__ void ola_frame_kernel(const cpx *__restrict__ rest, int nrest, const cpx *__restrict__ x,
                                 const cpx *__restrict__ prev_half, const float *__restrict__ fen, cpx *__restrict__ frames,
                                 int Ne, int N, int Nz, int windowed, int64_t total)
{
  const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t f = idx / N;
  const int i = (int) (idx - f * N), j = i - Nz;
  cpx v = make_float2(0.f, 0.f);
  if (j >= 0) {
    if (!windowed) {
      v = ola_src(rest, nrest, x, f * Ne + j);
    } else {
      const int64_t b = f >> 1;
      const int h = Ne / 2;
      if ((f & 1) == 0) {
        if (j < h) v = b == 0 ? prev_half[j] : ola_src(rest, nrest, x, (b - 1) * Ne + h + j);
        else v = ola_src(rest, nrest, x, b * Ne + (j - h));
      } else {
        v = ola_src(rest, nrest, x, b * Ne + j);
      }
      const float w = fen[j];
      v.x *= w;
      v.y *= w;
    }
  }
  frames[idx] = v;
}

// ---- framing fused into the forward transform (N = 16 .. 16384) ---------------------------------
// The frame kernel above writes 2N values per block and the FFT reads them back; here the
// Stockham transform of a frame (stockham16.hpp, N/16 threads) gathers its inputs straight from
// [rest ++ x] (zeros, window and the half-block overlap applied on the fly), and its store applies
// the optional response H or keeps only the power |X|^2 (psd_welch).
struct FrameSrc {
  const cpx *rest, *x, *prev_half;
  const float *fen;
  int nrest, Ne, N, Nz, mode;     // mode 0: simple OLA, 1: windowed OLA, 2: Welch segments (stride Ne, window fen)
  __device__ __forceinline__ cpx get(int64_t f, int i) const
  {
    if (mode == 2) {
      const cpx v = x[f * Ne + i];
      const float w = fen[i];
      return make_float2(v.x * w, v.y * w);
    }
    const int j = i - Nz;
    if (j < 0) return make_float2(0.f, 0.f);
    if (mode == 0) return ola_src(rest, nrest, x, f * Ne + j);
    const int64_t b = f >> 1;
    const int h = Ne / 2;
    cpx v;
    if ((f & 1) == 0) v = j < h ? (b == 0 ? prev_half[j] : ola_src(rest, nrest, x, (b - 1) * Ne + h + j)) : ola_src(rest, nrest, x, b * Ne + (j - h));
    else v = ola_src(rest, nrest, x, b * Ne + j);
    const float w = fen[j];
    return make_float2(v.x * w, v.y * w);
  }
};

template <int R0>
__global__ __launch_bounds__(1024) void framed_fft_kernel(FrameSrc S, const cpx *__restrict__ TW, int tpt, float scale, int64_t nfr,
                                                          const cpx *__restrict__ H, cpx *__restrict__ outc,
                                                          float *__restrict__ outp)
{
  extern __shared__ __attribute__((aligned(16))) char fr_raw[];
  cpx *lds = reinterpret_cast<cpx *>(fr_raw);
  const int N = S.N, t = threadIdx.x, T = blockDim.x / tpt;
  const int tl = t / tpt, j = t - tl * tpt;
  const int64_t f = (int64_t) blockIdx.x * T + tl;
  const bool live = f < nfr;
  cpx *s = lds + tl * (N + (N >> 4));
  cpx v[16];
#pragma unroll
  for (int m = 0; m < 16; m++) v[m] = live ? S.get(f, j + m * tpt) : make_float2(0.f, 0.f);
  s16::transform<R0>(v, s, TW, N, j, tpt, []() { __syncthreads(); });
  if (!live) return;
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const int k = j + q * tpt;
    cpx X = make_float2(v[q].x * scale, v[q].y * scale);
    if (H) {
      const cpx b = H[k];
      X = make_float2(X.x * b.x - X.y * b.y, X.x * b.y + X.y * b.x);
    }
    if (outc) outc[f * N + k] = X;
    else outp[f * N + k] = X.x * X.x + X.y * X.y;
  }
}

// launches the fused kernel when the plan runs on the radix-16 Stockham engine; false otherwise
bool framed_fft_launch(const tsdgpu_fft *plan, const FrameSrc &S, int64_t nfr, const cpx *H, cpx *outc, float *outp, hipStream_t st)
{
  const cpx *TW = fft_s16_twiddles(plan);
  static const bool off = dev_switch("OLA_UNFUSED") != nullptr;
  if (!TW || off || nfr <= 0) return false;
  const int N = S.N, tpt = N / 16, threads = std::max(256, tpt), T = threads / tpt;
  const size_t lds = (size_t) T * (N + N / 16) * sizeof(cpx);
  int logn = 0;
  while ((1 << logn) < N) logn++;
  const int r0 = 1 << ((logn & 3) == 0 ? 4 : (logn & 3));
  const float scale = 1.0f / std::sqrt((float) N);
  const unsigned grid = (unsigned) cdiv(nfr, T);
#define FR_LAUNCH(R)                                                                                                      \
  do {                                                                                                                    \
    (void) hipFuncSetAttribute((const void *) framed_fft_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((framed_fft_kernel<R>), dim3(grid), dim3(threads), lds, st, S, TW, tpt, scale, nfr, H, outc, outp);  \
  } while (0)
  if (r0 == 16) FR_LAUNCH(16); else if (r0 == 8) FR_LAUNCH(8); else if (r0 == 4) FR_LAUNCH(4); else FR_LAUNCH(2);
#undef FR_LAUNCH
  return true;
}

// ---- OLA without window as ONE kernel, any geometry whose frame fits the LDS ------------------------------------------
// OLA<cfloat>::step_interne (fourier.cc:846-872):  frame_b = [Nz zeros | block b] -> FFT -> * H -> IFFT = x2_b;
//   y_b = x2_{b-1}[Nz ..] with its last Nz samples += x2_b[0 .. Nz);   svg <- x2_b[Nz ..]
// The separate passes (framed transform, inverse transform, addition) move 80 B per sample through HBM.  Here tpt = N/16
// threads own a RUN of `per` consecutive blocks: both transforms in the LDS image of the frame (stockham16.hpp), the
// Ne carried samples in LDS beside it, 16 B of HBM traffic per sample.  A run starts from the block before it, which it
// recomputes (the first one takes the handle's svg).  Hs = H / N: the engine's transforms are unitary each way, the
// ones here are not scaled; the inverse is the conjugate of the forward transform of the conjugate.
// Every thread of the workgroup passes every barrier: runs past the end carry zeros.
template <int R0, int THREADS, bool HALF>
__global__ __launch_bounds__(THREADS, THREADS == 256 ? 3 : THREADS == 512 ? 2 : 4) void ola_run_kernel(const cpx *__restrict__ blk0, int nrest, const cpx *__restrict__ x,
                                                       cpx *__restrict__ y, const cpx *__restrict__ Hs, const cpx *__restrict__ TW,
                                                       const cpx *__restrict__ svg_in, cpx *__restrict__ svg_out, int Ne, int N, int tpt,
                                                       int64_t B, int per)
{
  extern __shared__ __attribute__((aligned(16))) char run_raw[];
  const int Nz = N - Ne, t = threadIdx.x, T = THREADS / tpt;
  const int tl = t / tpt, j0 = t - tl * tpt;
  // HALF (Ne = Nz = 8 tpt): registers 8..15 of a thread are the carried samples of its registers 0..7 -- no LDS copy
  cpx *s = reinterpret_cast<cpx *>(run_raw) + (size_t) tl * (N + (N >> 4) + (HALF ? 0 : Ne));
  cpx *tail = s + N + (N >> 4);
  const int64_t b_lo = ((int64_t) blockIdx.x * T + tl) * per, b_hi = min(B, b_lo + (int64_t) per);
  auto sync = []() { __syncthreads(); };
  cpx v[16], carry[HALF ? 8 : 1];
  for (int it = -1; it < per; it++) {
    const int64_t b = b_lo + it;
    // (opaque copy: keeps the compiler from carrying every address derived from the thread index across the loop --
    // 240 VGPRs and scratch otherwise)
    int j = j0;
    asm volatile("" : "+v"(j));
    {
      // block 0 = [rest ++ head of x] was made contiguous by the host (blk0); block b > 0 starts at x[b Ne - nrest]
      const bool live = b >= 0 && b < b_hi;
      const cpx *src = (b == 0 ? blk0 : x + (b * Ne - nrest)) - Nz;
#pragma unroll
      for (int m = 0; m < 16; m++) {
        const int p = j + m * tpt;
        v[m] = (live && (HALF ? m >= 8 : p >= Nz)) ? src[p] : make_float2(0.f, 0.f);
      }
    }
    if (it >= 0) sync();                            // the image is still being read by the last pass of the block before
    s16::transform<R0>(v, s, TW, N, j, tpt, sync);
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const cpx h = Hs[j + q * tpt];
      v[q] = make_float2(v[q].x * h.x - v[q].y * h.y, -(v[q].x * h.y + v[q].y * h.x));
    }
    sync();
    asm volatile("" : "+v"(j));
    s16::transform<R0>(v, s, TW, N, j, tpt, sync);
    asm volatile("" : "+v"(j));
    // v[q] = conj(x2[j + q tpt])
    const bool keep = it < 0 || b < b_hi;           // (a run cut short by the end keeps its last carried block)
    if (HALF) {
      if (it >= 0 && b < b_hi) {
#pragma unroll
        for (int q = 0; q < 8; q++) y[b * Ne + j + q * tpt] = make_float2(carry[q].x + v[q].x, carry[q].y - v[q].y);
      }
      if (it < 0 && b_lo == 0) {
#pragma unroll
        for (int q = 0; q < 8; q++) carry[q] = svg_in[j + q * tpt];
      } else if (keep) {
#pragma unroll
        for (int q = 0; q < 8; q++) carry[q] = make_float2(v[q + 8].x, -v[q + 8].y);
      }
    } else {
      if (it >= 0) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
          const int p = j + q * tpt;
          if (p < Nz) {
            const cpx a = tail[Ne - Nz + p];
            tail[Ne - Nz + p] = make_float2(a.x + v[q].x, a.y - v[q].y);
          }
        }
        sync();
        if (b < b_hi)
          for (int i = j; i < Ne; i += tpt) y[b * Ne + i] = tail[i];
        sync();
      }
      if (it < 0 && b_lo == 0) {
        for (int i = j; i < Ne; i += tpt) tail[i] = svg_in[i];
      } else if (keep) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
          const int p = j + q * tpt;
          if (p >= Nz) tail[p - Nz] = make_float2(v[q].x, -v[q].y);
        }
      }
    }
  }
  if (b_lo < B && b_hi == B) {
    if (HALF) {
#pragma unroll
      for (int q = 0; q < 8; q++) svg_out[j0 + q * tpt] = carry[q];
    }
  }
  if (!HALF) {
    sync();
    if (b_lo < B && b_hi == B)
      for (int i = j0; i < Ne; i += tpt) svg_out[i] = tail[i];
  }
}

struct OlaRunGeom {
  int tpt, threads, T;
  bool half;
  size_t lds;
};
OlaRunGeom ola_run_geom(int N, int Ne)
{
  OlaRunGeom g;
  g.tpt = std::max(N / 16, 1);
  g.threads = std::max(256, g.tpt);
  g.T = g.threads / g.tpt;
  g.half = 2 * Ne == N;
  g.lds = (size_t) g.T * (N + N / 16 + (g.half ? 0 : Ne)) * sizeof(cpx);
  return g;
}
// the fused kernel serves the frames that fit the LDS with at most 512 threads per transform
bool ola_run_fits(int N, int Ne)
{
  const OlaRunGeom g = ola_run_geom(N, Ne);
  // (N = 16384 -- 1024 threads, 128 registers each -- only with the carried block in registers)
  return N >= 16 && (g.threads <= 512 || (g.threads == 1024 && g.half)) && g.lds <= 160 * 1024;
}

// B >= 1 whole blocks of [rest ++ x] -> y; tables = Hs (N), TW (N/16).  y may not alias x (runs re-read their predecessor).
int ola_run_launch(const cpx *blk0, int nrest, const cpx *x, cpx *y, const cpx *tables, const cpx *svg_in, cpx *svg_out, int Ne, int N,
                   int64_t B, hipStream_t st)
{
  const OlaRunGeom g = ola_run_geom(N, Ne);
  const int tpt = g.tpt, T = g.T;
  int logn = 0;
  while ((1 << logn) < N) logn++;
  const int r0 = 1 << ((logn & 3) == 0 ? 4 : (logn & 3));
  // blocks per run: every run recomputes the block before it, so a run costs per + 1 blocks, and the workgroups pass over
  // the chip in rounds of (CUs x resident workgroups): the cheapest (rounds x (per + 1)) wins
  static const int cus = []() {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const int64_t slots = (int64_t) cus * std::max<int64_t>(1, std::min<int64_t>(g.threads == 256 ? 3 : 1, (int64_t) ((160 * 1024) / g.lds)));   // (registers: 3 waves per SIMD)
  int per = 1;
  int64_t best = -1;
  for (int c = 1; c <= 32; c++) {
    const int64_t wgs = cdiv(cdiv(B, c), T), cost = cdiv(wgs, slots) * (c + 1);
    if (best < 0 || cost < best) best = cost, per = c;
  }
  const int64_t runs = cdiv(B, per), grid = cdiv(runs, T);
  if (grid > 0x7fffffff) return set_err(TSDGPU_ERR_UNSUPPORTED, "ola: too many blocks in one call");
#define RUN_LAUNCH(R, TH, HF)                                                                                            \
  do {                                                                                                                   \
    (void) hipFuncSetAttribute((const void *) ola_run_kernel<R, TH, HF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((ola_run_kernel<R, TH, HF>), dim3((unsigned) grid), dim3(TH), g.lds, st, blk0, nrest, x, y, tables, tables + N, svg_in, svg_out, Ne, N, tpt, B, per); \
  } while (0)
#define RUN_PICK(TH, HF)                                                                                                 \
  do {                                                                                                                   \
    if (r0 == 16) RUN_LAUNCH(16, TH, HF); else if (r0 == 8) RUN_LAUNCH(8, TH, HF); else if (r0 == 4) RUN_LAUNCH(4, TH, HF); else RUN_LAUNCH(2, TH, HF); \
  } while (0)
  if (g.threads == 256 && g.half) RUN_PICK(256, true);
  else if (g.threads == 256) RUN_PICK(256, false);
  else if (g.threads == 512 && g.half) RUN_PICK(512, true);
  else if (g.threads == 512) RUN_PICK(512, false);
  else if (g.threads == 1024 && g.half) RUN_PICK(1024, true);
  else return set_err(TSDGPU_ERR_UNSUPPORTED, "ola: frame of %d points does not fit the fused kernel", N);   // (set_response does not select it)
#undef RUN_PICK
#undef RUN_LAUNCH
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}

// ---- the WINDOWED mode (fourier.cc:883-927) as ONE kernel: a run of consecutive blocks emulated statement by statement ----
// Per block two Hann-windowed frames half a block apart -- A: [second half of the block before | first half of this one],
// B: the block -- each [Nz zeros | Ne samples] -> FFT -> x H -> IFFT = x2, folded into the two carried vectors exactly as the
// reference does it:
//   A:  svg.segment(Ne - Nz, Nz) += x2.head(Nz);  last.tail(h) += svg.head(h) / 2;  y = last;
//       last.head(h) = svg.tail(h) / 2;  last.tail(h) = 0;  svg = x2.tail(Ne)
//   B:  svg.tail(Nz) += x2.head(Nz);  last += svg / 2;  svg = x2.segment(Nz, Ne)
// N/16 threads own a run; frame image, svg and last live in LDS (the reference's own order of additions, no closed form).
// The state entering block b depends on frames 2b - 3 .. 2b - 1 only (svg is overwritten by every frame, last by step A), so a
// run that starts at block b_lo >= 2 recomputes frame B of block b_lo - 2 and block b_lo - 1 from zero state and discards their
// output; the runs that start at block 0 or 1 begin from the handle's state.  16 B of HBM traffic per sample against the ~100 B of the framing /
// spectrum / inverse-frame / overlap-add passes.  Every thread of the workgroup passes every barrier: dead frames run on zeros.
// THREADS = 64 (frames of up to 1024 points: a run's N/16 threads lie inside ONE wave): no workgroup barrier at all -- a wave's
// LDS operations execute in order, wavefront-scope fences keep the compiler from moving them -- so the waves of a CU drift
// freely and hide each other's LDS latencies (with 256-thread workgroups the ~15 barriers per frame put 4 waves in lockstep:
// 0.404 ms per 2^24 samples at Ne = 512, slower than the multi-kernel engine).
template <int R0, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS <= 256 ? 2 : 1) void olaw_run_kernel(const cpx *__restrict__ blk0, int nrest, const cpx *__restrict__ x,
                                                        cpx *__restrict__ y, const cpx *__restrict__ Hs, const cpx *__restrict__ TW,
                                                        const float *__restrict__ fen, const cpx *__restrict__ svg_in,
                                                        const cpx *__restrict__ last_in, const cpx *__restrict__ prev_half_in,
                                                        cpx *__restrict__ svg_out, cpx *__restrict__ last_out, int Ne, int N, int tpt,
                                                        int64_t B, int per, int skip_first)
{
  extern __shared__ __attribute__((aligned(16))) char runw_raw[];
  const int Nz = N - Ne, h = Ne / 2, t = threadIdx.x, T = THREADS / tpt;
  const int tl = t / tpt, j0 = t - tl * tpt;
  cpx *s = reinterpret_cast<cpx *>(runw_raw) + (size_t) tl * (N + (N >> 4) + 2 * Ne);
  cpx *svg = s + N + (N >> 4), *last = svg + Ne;
  const int64_t b_lo = ((int64_t) blockIdx.x * T + tl) * per, b_hi = min(B, b_lo + (int64_t) per);
  auto sync = []() {
    if (THREADS == 64) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
      __syncthreads();
    }
  };
  // the runs that start at block 0 or 1 begin from the handle's state (block 0 is then the second run's whole warm-up); the others
  // from zeros, two blocks early
  const bool first_run = b_lo == 0, from_handle = b_lo <= 1 && b_lo < B;
  for (int i = j0; i < Ne; i += tpt) {
    svg[i] = from_handle ? svg_in[i] : make_float2(0.f, 0.f);
    last[i] = from_handle ? last_in[i] : make_float2(0.f, 0.f);
  }
  // sample i of block b of [rest ++ x] (block 0 was made contiguous by the host)
  auto blk = [&](int64_t b, int i) { return b == 0 ? blk0[i] : x[b * Ne - nrest + i]; };
  cpx v[16];
  bool first = true;
  for (int it = -2; it < per; it++) {
    const int64_t b = b_lo + it;
#pragma unroll 1
    for (int f = 0; f < 2; f++) {
      // frames that exist for this run: the warm-up (frame B of b_lo - 2, both of b_lo - 1) unless the run starts the call,
      // then the run's own blocks
      const bool live = b_lo < B && (it >= 0 ? b < b_hi : (!first_run && b >= 0 && (it == -1 || f == 1)));
      int j = j0;
      asm volatile("" : "+v"(j));      // (see ola_run_kernel: no addresses carried across the loop)
#pragma unroll
      for (int m = 0; m < 16; m++) {
        const int p = j + m * tpt, i = p - Nz;
        cpx a = make_float2(0.f, 0.f);
        if (live && i >= 0) {
          if (f == 0) a = i < h ? (b == 0 ? prev_half_in[i] : blk(b - 1, h + i)) : blk(b, i - h);      // :885
          else a = blk(b, i);                                                                          // :910
          const float w = fen[i];
          a.x *= w;
          a.y *= w;
        }
        v[m] = a;
      }
      if (!first) sync();               // the image is still being read by the last pass of the frame before
      first = false;
      s16::transform<R0>(v, s, TW, N, j, tpt, sync);
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const cpx hh = Hs[j + q * tpt];
        v[q] = make_float2(v[q].x * hh.x - v[q].y * hh.y, -(v[q].x * hh.y + v[q].y * hh.x));
      }
      sync();
      asm volatile("" : "+v"(j));
      s16::transform<R0>(v, s, TW, N, j, tpt, sync);
      asm volatile("" : "+v"(j));
      // v[q] = conj(x2[j + q tpt])
      if (live) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
          const int p = j + q * tpt;
          if (p < Nz) {                                          // svg.segment(Ne - Nz, Nz) += x2.head(Nz)   (:894 / :917)
            const cpx a = svg[Ne - Nz + p];
            svg[Ne - Nz + p] = make_float2(a.x + v[q].x, a.y - v[q].y);
          }
        }
      }
      sync();
      if (live) {
        if (f == 0) {
          for (int i = j; i < h; i += tpt) {                     // last.tail(h) += svg.head(h) / 2   (:897)
            const cpx a = last[h + i], c = svg[i];
            last[h + i] = make_float2(a.x + c.x / 2.0f, a.y + c.y / 2.0f);
          }
        } else {
          for (int i = j; i < Ne; i += tpt) {                    // last += svg / 2   (:918)
            const cpx a = last[i], c = svg[i];
            last[i] = make_float2(a.x + c.x / 2.0f, a.y + c.y / 2.0f);
          }
        }
      }
      sync();
      if (live && f == 0) {
        if (it >= 0 && !(skip_first && b == 0))                  // y = last   (:898-901; nothing for the very first block)
          for (int i = j; i < Ne; i += tpt) y[(b - skip_first) * Ne + i] = last[i];
        for (int i = j; i < h; i += tpt) {                       // last.head(h) = svg.tail(h) / 2; last.tail(h) = 0   (:903-904)
          const cpx c = svg[h + i];
          last[i] = make_float2(c.x / 2.0f, c.y / 2.0f);
          last[h + i] = make_float2(0.f, 0.f);
        }
      }
      sync();
      if (live) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
          const int p = j + q * tpt;
          if (p >= Nz) svg[p - Nz] = make_float2(v[q].x, -v[q].y);      // svg = x2.tail(Ne)   (:907 / :919)
        }
      }
    }
  }
  sync();
  if (b_lo < B && b_hi == B)
    for (int i = j0; i < Ne; i += tpt) {
      svg_out[i] = svg[i];
      last_out[i] = last[i];
    }
}

struct OlawGeom {
  int tpt, threads, T;
  size_t lds;
};
OlawGeom olaw_geom(int N, int Ne)
{
  OlawGeom g;
  g.tpt = std::max(N / 16, 1);
  g.threads = g.tpt <= 64 ? 64 : std::max(256, g.tpt);       // (a run inside one wave: one-wave workgroups, no barriers)
  g.T = g.threads / g.tpt;
  g.lds = (size_t) g.T * (N + N / 16 + 2 * Ne) * sizeof(cpx);
  return g;
}
bool olaw_run_fits(int N, int Ne)
{
  const OlawGeom g = olaw_geom(N, Ne);
  // (only frames of up to 1024 points, whose runs lie inside one wave: with workgroup barriers the kernel is slower than the
  // multi-kernel engine -- 0.416 against 0.373 ms per 2^24 samples at Ne = N = 4096; 0.317 against 0.375 at Ne = N = 512)
  return N >= 16 && g.threads == 64 && g.lds <= 160 * 1024 && (Ne & 1) == 0;
}
// B >= 1 whole blocks of [rest ++ x] -> y (B - skip_first blocks of output); tables = Hs (N), TW (N/16)
int olaw_run_launch(const cpx *blk0, int nrest, const cpx *x, cpx *y, const cpx *tables, const float *fen, const cpx *svg_in, const cpx *last_in,
                    const cpx *prev_half_in, cpx *svg_out, cpx *last_out, int Ne, int N, int64_t B, int skip_first, hipStream_t st)
{
  const OlawGeom g = olaw_geom(N, Ne);
  int logn = 0;
  while ((1 << logn) < N) logn++;
  const int r0 = 1 << ((logn & 3) == 0 ? 4 : (logn & 3));
  static const int cus = []() {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  // blocks per run: a run costs per + 1.5 blocks (its warm-up), the workgroups pass over the chip in rounds
  const int64_t slots = (int64_t) cus * std::max<int64_t>(1, std::min<int64_t>(g.threads == 64 ? 16 : g.threads == 256 ? 2 : 1, (int64_t) ((160 * 1024) / g.lds)));
  int per = 1;
  int64_t best = -1;
  for (int c = 1; c <= 64; c++) {
    const int64_t wgs = cdiv(cdiv(B, c), g.T), cost = cdiv(wgs, slots) * (2 * c + 3);
    if (best < 0 || cost < best) best = cost, per = c;
  }
  const int64_t grid = cdiv(cdiv(B, per), g.T);
  if (grid > 0x7fffffff) return set_err(TSDGPU_ERR_UNSUPPORTED, "ola: too many blocks in one call");
#define RUNW_LAUNCH(R, TH)                                                                                                \
  do {                                                                                                                   \
    (void) hipFuncSetAttribute((const void *) olaw_run_kernel<R, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((olaw_run_kernel<R, TH>), dim3((unsigned) grid), dim3(TH), g.lds, st, blk0, nrest, x, y, tables, tables + N, fen, svg_in, last_in, prev_half_in, svg_out, last_out, Ne, N, g.tpt, B, per, skip_first); \
  } while (0)
#define RUNW_PICK(TH)                                                                                                    \
  do {                                                                                                                   \
    if (r0 == 16) RUNW_LAUNCH(16, TH); else if (r0 == 8) RUNW_LAUNCH(8, TH); else if (r0 == 4) RUNW_LAUNCH(4, TH); else RUNW_LAUNCH(2, TH); \
  } while (0)
  if (g.threads == 64) RUNW_PICK(64);
  else return set_err(TSDGPU_ERR_UNSUPPORTED, "ola: frame of %d points does not fit the fused windowed kernel", N);
#undef RUNW_PICK
#undef RUNW_LAUNCH
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}

// ---- Welch sums as ONE kernel, 16 <= N <= 8192 (N = 1024 has its in-wave flavour in ols.hip) -------------------------
// psd_welch (freqestim.cc:7-20): segments of N samples every N/2, windowed, |FFT|^2 summed.  N/16 threads walk a run of
// `per` consecutive segments: window, transform in the LDS image, the 16 running sums of a thread in registers; the run's
// row part[run][.] gets them in fftshift order, scaled to the engine's unitary transform (|X|^2 / N).  8 B of HBM
// traffic per sample and a row per run, against the |X|^2 plane (4 B per segment sample written, then read) of the
// framed transform + summation passes.
// Which segments a run sums (the run's row of `part` gets the sum):
//   mode 0  psd_welch: run r = segments [r per, (r + 1) per) of nseg, a segment every N/2 samples
//   mode 1  rt_spectrum (fourier.cc:1228-1335), every sub-block of a group of blocks into one spectrum: segment s = sample
//           s N; the call's B blocks of nsubs segments fall into groups -- group 0 completes the cnt0 blocks the handle
//           has accumulated already (up to nmeans), the others hold nmeans blocks, the last one what is left -- and a group's
//           segments are cut in rpg runs of per
//   mode 2  rt_spectrum in sweep mode: sub-block i of every block of a group is a class of its own (segments b nsubs + i,
//           stride nsubs); class (g, i) is cut in rpg runs of per blocks
struct SegRuns {
  int mode, per, hop, nsubs, nmeans, cnt0, rpg;
  int64_t nseg, B;
};
__device__ __forceinline__ void seg_run(const SegRuns &M, int64_t run, int64_t &seg0, int &count, int &stride)
{
  if (M.mode == 0) {
    seg0 = run * M.per;
    count = (int) max((int64_t) 0, min((int64_t) M.per, M.nseg - seg0));
    stride = 1;
    return;
  }
  const int64_t cls = run / M.rpg;
  const int r = (int) (run - cls * M.rpg);
  const int64_t g = M.mode == 1 ? cls : cls / M.nsubs;
  const int i = M.mode == 1 ? 0 : (int) (cls - g * M.nsubs);
  const int64_t be0 = min(M.B, (int64_t) (M.nmeans - M.cnt0));
  const int64_t bs = g == 0 ? 0 : be0 + (g - 1) * M.nmeans, be = g == 0 ? be0 : min(M.B, bs + M.nmeans);
  if (M.mode == 1) {
    seg0 = bs * M.nsubs + (int64_t) r * M.per;
    count = (int) max((int64_t) 0, min((int64_t) M.per, be * M.nsubs - seg0));
    stride = 1;
  } else {
    const int64_t b0 = bs + (int64_t) r * M.per;
    count = (int) max((int64_t) 0, min((int64_t) M.per, be - b0));
    seg0 = b0 * M.nsubs + i;
    stride = M.nsubs;
  }
}
template <int R0, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS == 256 ? 3 : THREADS == 512 ? 2 : 4) void welch_run_kernel(const cpx *__restrict__ x, const float *__restrict__ w,
                                                                                    const cpx *__restrict__ TW, float *__restrict__ part,
                                                                                    int N, int tpt, SegRuns M, int64_t nruns)
{
  extern __shared__ __attribute__((aligned(16))) char wr_raw[];
  const int t = threadIdx.x, T = THREADS / tpt, pas = M.hop, per = M.per;
  const int tl = t / tpt, j0 = t - tl * tpt;
  cpx *s = reinterpret_cast<cpx *>(wr_raw) + (size_t) tl * (N + (N >> 4));
  const int64_t run = (int64_t) blockIdx.x * T + tl;
  int64_t k_lo = 0;
  int count = 0, stride = 1;
  if (run < nruns) seg_run(M, run, k_lo, count, stride);
  auto sync = []() { __syncthreads(); };
  float win[16], acc[16];
#pragma unroll
  for (int m = 0; m < 16; m++) {
    win[m] = w[j0 + m * tpt];
    acc[m] = 0.f;
  }
  cpx v[16];
  for (int it = 0; it < per; it++) {
    const int64_t k = k_lo + (int64_t) it * stride;
    int j = j0;
    asm volatile("" : "+v"(j));        // (see ola_run_kernel: no addresses carried across the loop)
    const bool live = it < count;
    const cpx *src = x + k * pas;
#pragma unroll
    for (int m = 0; m < 16; m++) {
      const cpx a = live ? src[j + m * tpt] : make_float2(0.f, 0.f);
      v[m] = make_float2(a.x * win[m], a.y * win[m]);                          // x.segment(i, N) * f  (:15)
    }
    if (it > 0) sync();                // the image is still being read by the last pass of the segment before
    s16::transform<R0>(v, s, TW, N, j, tpt, sync);
#pragma unroll
    for (int q = 0; q < 16; q++) acc[q] += v[q].x * v[q].x + v[q].y * v[q].y;    // abs2 (:16)
  }
  if (run < nruns && (M.mode != 0 || count > 0)) {
    const float g = 1.0f / (float) N;
#pragma unroll
    for (int q = 0; q < 16; q++) part[(size_t) run * N + ((j0 + q * tpt + N / 2) & (N - 1))] = acc[q] * g;
  }
}

// rows of partial sums written: part must hold nruns x N floats
int seg_runs_launch(const cpx *x, const float *w, const cpx *TW, float *part, int N, const SegRuns &M, int64_t nruns, hipStream_t st)
{
  const OlaRunGeom g = ola_run_geom(N, N / 2);
  int logn = 0;
  while ((1 << logn) < N) logn++;
  const int r0 = 1 << ((logn & 3) == 0 ? 4 : (logn & 3));
  const int64_t grid = cdiv(nruns, g.T);
#define WR_LAUNCH(R, TH)                                                                                                 \
  do {                                                                                                                   \
    (void) hipFuncSetAttribute((const void *) welch_run_kernel<R, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((welch_run_kernel<R, TH>), dim3((unsigned) grid), dim3(TH), g.lds, st, x, w, TW, part, N, g.tpt, M, nruns); \
  } while (0)
#define WR_PICK(TH)                                                                                                      \
  do {                                                                                                                   \
    if (r0 == 16) WR_LAUNCH(16, TH); else if (r0 == 8) WR_LAUNCH(8, TH); else if (r0 == 4) WR_LAUNCH(4, TH); else WR_LAUNCH(2, TH); \
  } while (0)
  if (g.threads == 256) WR_PICK(256);
  else if (g.threads == 512) WR_PICK(512);
  else if (g.threads == 1024) WR_PICK(1024);
  else return set_err(TSDGPU_ERR_UNSUPPORTED, "welch: N = %d does not fit the fused kernel", N);
#undef WR_PICK
#undef WR_LAUNCH
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}
int welch_run_launch(const cpx *x, const float *w, const cpx *TW, float *part, int N, int64_t nseg, int per, hipStream_t st)
{
  const SegRuns M = {0, per, N / 2, 1, 1, 0, 1, nseg, 0};
  return seg_runs_launch(x, w, TW, part, N, M, cdiv(nseg, per), st);
}

// dst[i] = sample (p0 + i) of [rest ++ x]   (new prev_half / new rest)
__global__ void ola_gather_kernel(const cpx *__restrict__ rest, int nrest, const cpx *__restrict__ x, int64_t p0,
                                  cpx *__restrict__ dst, int count)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] = ola_src(rest, nrest, x, p0 + i);
}

__global__ void ola_mul_kernel(cpx *__restrict__ X, const cpx *__restrict__ H, int N, int64_t total)
{
  const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const cpx a = X[idx], b = H[idx % N];
  X[idx] = make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// simple OLA (fourier.cc:868-872): out_b = svg with its last Nz samples += x2_b.head(Nz),
// svg <- x2_b.tail(Ne); the svg of block b > 0 is the tail of frame b-1.
__global__ void ola_add_kernel(const cpx *__restrict__ fr, const cpx *__restrict__ svg, cpx *__restrict__ out, int Ne, int N,
                               int Nz, int64_t total)
{
  const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t b = idx / Ne;
  const int i = (int) (idx - b * Ne);
  cpx v = b == 0 ? svg[i] : fr[(b - 1) * N + Nz + i];
  if (i >= Ne - Nz) {
    const cpx a = fr[b * N + (i - (Ne - Nz))];
    v.x += a.x;
    v.y += a.y;
  }
  out[idx] = v;
}

// windowed OLA (fourier.cc:883-923).  The reference walks the blocks with two carried vectors,
// svg and last; unrolled, block b only needs the frames of blocks b-2 .. b:
//   S_b[i]    = xb_b[Nz+i]                                   svg after the block        (:919)
//   svgA_b[i] = S_{b-1}[i] + T(xa_b)[i]                      svg after the first add    (:892)
//   L_b[i]    = (i < h ? svgA_b[h+i]/2 : 0) + (xa_b[Nz+i] + T(xb_b)[i])/2   last after the block (:901-918)
//   out_b[i]  = L_{b-1}[i] + (i >= h ? svgA_b[i-h]/2 : 0)                                (:895-896)
// with T(v)[i] = v[i-(Ne-Nz)] for i >= Ne-Nz, else 0; S_{-1}, L_{-1} = the handle's state.
// The additions are made in the reference's order.
struct OlaW {
  const cpx *fr, *svg0;
  int Ne, N, Nz;
  __device__ __forceinline__ cpx T(const cpx *v, int i) const { return i >= Ne - Nz ? v[i - (Ne - Nz)] : make_float2(0.f, 0.f); }
  __device__ __forceinline__ cpx svgA(int64_t b, int i) const
  {
    const cpx s = b == 0 ? svg0[i] : fr[(size_t) (2 * b - 1) * N + Nz + i];
    const cpx t = T(fr + (size_t) (2 * b) * N, i);
    return make_float2(s.x + t.x, s.y + t.y);
  }
  __device__ __forceinline__ cpx L(int64_t b, int i) const
  {
    const int h = Ne / 2;
    cpx l = make_float2(0.f, 0.f);
    if (i < h) {
      const cpx a = svgA(b, h + i);
      l = make_float2(a.x / 2.0f, a.y / 2.0f);
    }
    const cpx xa = fr[(size_t) (2 * b) * N + Nz + i], t = T(fr + (size_t) (2 * b + 1) * N, i);
    const cpx sb = make_float2(xa.x + t.x, xa.y + t.y);
    return make_float2(l.x + sb.x / 2.0f, l.y + sb.y / 2.0f);
  }
};

__global__ void ola_add_windowed_kernel(OlaW w, const cpx *__restrict__ last0, cpx *__restrict__ out, int skip_first,
                                        int64_t total)
{
  const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t b = idx / w.Ne + skip_first;
  const int i = (int) (idx % w.Ne), h = w.Ne / 2;
  cpx v = b == 0 ? last0[i] : w.L(b - 1, i);
  if (i >= h) {
    const cpx a = w.svgA(b, i - h);
    v.x += a.x / 2.0f;
    v.y += a.y / 2.0f;
  }
  out[idx] = v;
}

// new state: last <- L_{B-1} (reads the OLD svg when B == 1), then svg <- S_{B-1} in a later launch
__global__ void ola_state_windowed_kernel(OlaW w, cpx *__restrict__ last, int64_t B)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < w.Ne) last[i] = w.L(B - 1, i);
}

// ---- psd_welch ------------------------------------------------------------------------------------
__global__ void welch_frame_kernel(const cpx *__restrict__ x, const float *__restrict__ w, cpx *__restrict__ seg, int N, int pas,
                                   int64_t total)
{
  const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t k = idx / N;
  const int j = (int) (idx - k * N);
  const cpx v = x[k * pas + j];
  const float f = w[j];
  seg[idx] = make_float2(v.x * f, v.y * f);                       // x.segment(i, N) * f  (:15)
}

// part[g][i] = sum over the segments of group g of |X[seg][src(i)]|^2, src = the fftshift map
// (res.head(N/2) = X.tail(N/2), fourier.hpp:232-248); consecutive threads = consecutive bins
// the same reduction over segment powers already formed by the fused transform
__global__ void welch_power_sum_kernel(const float *__restrict__ P, float *__restrict__ part, int N, int64_t nseg, int64_t per_group)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int h = N / 2;
  const int src = i < h ? N - h + i : i - h;
  const int64_t k0 = (int64_t) blockIdx.y * per_group, k1 = min(k0 + per_group, nseg);
  float acc = 0.f;
  for (int64_t k = k0; k < k1; k++) acc += P[k * N + src];
  part[(size_t) blockIdx.y * N + i] = acc;
}
__global__ void welch_power_kernel(const cpx *__restrict__ X, float *__restrict__ part, int N, int64_t nseg, int64_t per_group)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int h = N / 2;
  const int src = i < h ? N - h + i : i - h;
  const int64_t k0 = (int64_t) blockIdx.y * per_group, k1 = min(k0 + per_group, nseg);
  float acc = 0.f;
  for (int64_t k = k0; k < k1; k++) {
    const cpx v = X[k * N + src];
    acc += v.x * v.x + v.y * v.y;                                 // abs2 (:16)
  }
  part[(size_t) blockIdx.y * N + i] = acc;
}
// first stage of a two-stage sum over many partial rows: out[y][i] = sum of the rows of group y (rows_per_group each)
__global__ void welch_sum_groups_kernel(const float *__restrict__ part, float *__restrict__ out, int N, int rows, int rows_per_group)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int g0 = blockIdx.y * rows_per_group, g1 = min(g0 + rows_per_group, rows);
  float acc = 0.f;
  for (int g = g0; g < g1; g++) acc += part[(size_t) g * N + i];
  out[(size_t) blockIdx.y * N + i] = acc;
}
__global__ void welch_sum_kernel(const float *__restrict__ part, float *__restrict__ S, int N, int groups)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float acc = 0.f;
  for (int g = 0; g < groups; g++) acc += part[(size_t) g * N + i];
  S[i] = acc;
}

// ... of rows in transform order: bin i of the result is bin src(i) of the rows (the fftshift map above).  64 bins x 4 row lanes per
// workgroup (a row lane adds every fourth row, the four sums are added in lane order): N / 64 workgroups instead of N / 256
__global__ __launch_bounds__(256) void welch_sum_shift_kernel(const float *__restrict__ part, float *__restrict__ S, int N, int groups, int shift)
{
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, l = threadIdx.x >> 6, i = blockIdx.x * 64 + c;
  float acc = 0.f;
  if (i < N) {
    const int h = N / 2;
    const int src = !shift ? i : i < h ? N - h + i : i - h;       // (shift = 0: rows already in fftshift order)
    for (int g = l; g < groups; g += 4) acc += part[(size_t) g * N + src];
  }
  red[l][c] = acc;
  __syncthreads();
  if (l == 0 && i < N) S[i] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
}

inline unsigned nblk(int64_t total) { return (unsigned) cdiv(total, 256); }

// ---- rt_spectrum (fourier.cc:1162-1342) -----------------------------------------------------------------------------
// sizes the run kernel does not serve (not a power of two, or no room in LDS): the windowed sub-blocks go through the
// batched plan, then one thread per bin sums |X|^2 over the segments of its run (the plan is unitary: no 1/N here)
__global__ void spec_frame_kernel(const cpx *__restrict__ x, const float *__restrict__ w, cpx *__restrict__ seg, int N, int64_t total)
{
  const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = (int) (idx % N);
  const cpx v = x[idx];
  const float f = w[j];
  seg[idx] = make_float2(v.x * f, v.y * f);                       // x.segment(i * Nf, Nf) * f  (:1248)
}
__global__ void spec_power_rows_kernel(const cpx *__restrict__ X, float *__restrict__ part, int N, SegRuns M, int64_t nruns)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t run = blockIdx.y;
  if (i >= N || run >= nruns) return;
  int64_t seg0;
  int count, stride;
  seg_run(M, run, seg0, count, stride);
  const int h = N / 2;
  const int src = i < h ? N - h + i : i - h;                      // fftshift (fourier.hpp:232-248)
  float acc = 0.f;
  for (int it = 0; it < count; it++) {
    const cpx v = X[(seg0 + (int64_t) it * stride) * N + src];
    acc += v.x * v.x + v.y * v.y;                                 // abs2 (:1249)
  }
  part[(size_t) run * N + i] = acc;
}
// The groups of a call -> spectra.  blockIdx.y = group.  A group that reaches nmeans blocks gives a row of y:
//   plain   mag(k) = sum over the group's rows, / (nmeans nsubs Nf), 10 log10(. + FLT_MIN)                 (:1271-1283)
//   sweep   mag(i step + k) += sum_i(k) masque(k); / (nmeans nsubs Nf); / mag_cnt; the same dB             (:1262-1266)
// the last group, if incomplete, leaves its sums (per class, unmasked) in acc_out for the next call; group 0 starts from
// acc_in (two buffers: group 0 and the last group run concurrently).
struct SpecFin {
  int Nf, Ns, nsubs, nmeans, cnt0, rpg, sweep, step, G;
  int64_t B;
};
__global__ void spec_finish_kernel(const float *__restrict__ part, const float *__restrict__ acc_in, float *__restrict__ acc_out,
                                   const float *__restrict__ masque, const float *__restrict__ mag_cnt, float *__restrict__ y, SpecFin F)
{
  const int g = blockIdx.y;
  const int64_t be0 = min(F.B, (int64_t) (F.nmeans - F.cnt0));
  const int64_t bs = g == 0 ? 0 : be0 + (int64_t) (g - 1) * F.nmeans, be = g == 0 ? be0 : min(F.B, bs + F.nmeans);
  const bool complete = (g == 0 ? F.cnt0 : 0) + (be - bs) == F.nmeans;
  const bool last = g == F.G - 1;
  const int ncls = F.sweep ? F.nsubs : 1;
  // sum of class c (of this group) at bin k
  auto total = [&](int c, int k) {
    float t = g == 0 ? acc_in[(size_t) c * F.Nf + k] : 0.f;
    const float *rows = part + ((size_t) g * ncls + c) * F.rpg * F.Nf;
    for (int r = 0; r < F.rpg; r++) t += rows[(size_t) r * F.Nf + k];
    return t;
  };
  const float scale = (float) ((int64_t) F.nmeans * F.nsubs * F.Nf);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < max(F.Ns, F.Nf); idx += gridDim.x * blockDim.x) {
    if (complete) {
      if (idx < F.Ns) {
        float m = 0.f;
        if (!F.sweep) m = total(0, idx);
        else
          for (int i = 0; i < F.nsubs; i++) {
            const int k = idx - i * F.step;
            if (k >= 0 && k < F.Nf) m += total(i, k) * masque[k];
          }
        m = m / scale;
        if (F.sweep) m = m / mag_cnt[idx];
        y[(size_t) g * F.Ns + idx] = 10.f * log10f(m + 1.17549435e-38f);        // pow2db(mag_moy + numeric_limits<float>::min())
      }
      if (last && idx < F.Nf)
        for (int c = 0; c < ncls; c++) acc_out[(size_t) c * F.Nf + idx] = 0.f;    // mag_moy.setZero()
    } else if (idx < F.Nf) {                                                    // (only the last group can be incomplete)
      for (int c = 0; c < ncls; c++) acc_out[(size_t) c * F.Nf + idx] = total(c, idx);
    }
  }
}

// a few hundred samples from one device buffer to another: a kernel (a hipMemcpyAsync of that size spends 15-40 us in the
// runtime -- the buffering calls of a stream fed in blocks shorter than Ne cost 52 us on average with it)
int copy_small(cpx *dst, const cpx *src, int64_t count, hipStream_t st)
{
  if (count <= 0) return TSDGPU_OK;
  hipLaunchKernelGGL(ola_gather_kernel, dim3(nblk(count)), dim3(256), 0, st, (const cpx *) nullptr, 0, src, (int64_t) 0, dst, (int) count);
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}

int ola_alloc(cpx **p, size_t count)
{
  TSD_HIP(hipMalloc((void **) p, std::max<size_t>(count, 1) * sizeof(cpx)));
  TSD_HIP(hipMemset(*p, 0, std::max<size_t>(count, 1) * sizeof(cpx)));
  TSD_HIP(hipStreamSynchronize(nullptr));   // the memset is not ordered with the caller's non-blocking streams
  return TSDGPU_OK;
}

}  // namespace
}  // namespace tsdgpu

using namespace tsdgpu;

extern "C" {

int tsdgpu_ola_create(tsdgpu_ola **out, int block_len, int min_zeros, const float *window)
{
  TSD_CHECK(out != nullptr, "ola_create: out is NULL");
  *out = nullptr;
  TSD_CHECK(min_zeros >= 0, "ola_create: negative zero count");
  const int Ne = block_len > 0 ? block_len : 512;                          // fourier.cc:770-771
  int64_t N = 1;
  while (N < (int64_t) Ne + min_zeros) N <<= 1;                              // prochaine_puissance_de_2 (:776)
  TSD_CHECK(N <= (1 << 24), "ola_create: FFT size %lld too large", (long long) N);
  const int Nz = (int) N - Ne;
  // the overlap of a frame is carried in ONE block-sized buffer (svg.tail(Nz), :870)
  TSD_CHECK(Nz <= Ne, "ola_create: Nz = %d zeros exceed the block size Ne = %d", Nz, Ne);
  TSD_CHECK(window == nullptr || (Ne & 1) == 0, "ola_create: the windowed mode needs an even block size (Ne = %d)", Ne);
  tsdgpu_ola *h = new tsdgpu_ola();
  h->Ne = Ne;
  h->N = (int) N;
  h->Nz = Nz;
  h->windowed = window != nullptr;
  h->cnt_ech = -(Ne / 2);
  int rc = tsdgpu_fft_create(&h->plan, h->N, 1);
  if (!rc) {
    // the carried vectors in ONE zeroed allocation (five allocations, five memsets and five synchronisations before)
    const size_t q = ((size_t) Ne + 1) / 2 * 2;          // 16-byte slots
    rc = ola_alloc(&h->d_bloc, 6 * q);
    if (!rc) {
      h->d_svg = h->d_bloc;
      h->d_svg_tmp = h->d_bloc + q;
      h->d_last = h->d_bloc + 2 * q;
      h->d_rest = h->d_bloc + 3 * q;
      h->d_prev_half = h->d_bloc + 4 * q;
      h->d_last_tmp = h->d_bloc + 5 * q;
    }
  }
  if (!rc && window) {
    if (hipMalloc((void **) &h->d_fen, Ne * sizeof(float)) != hipSuccess ||
        hipMemcpy(h->d_fen, window, Ne * sizeof(float), hipMemcpyDefault) != hipSuccess)
      rc = set_err(TSDGPU_ERR_HIP, "ola_create: window upload failed: %s", hipGetErrorString(hipGetLastError()));
  }
  if (rc) {
    tsdgpu_ola_destroy(h);
    return rc;
  }
  *out = h;
  return TSDGPU_OK;
}

int tsdgpu_ola_fft_size(const tsdgpu_ola *h) { return h ? h->N : -1; }
int tsdgpu_ola_block_len(const tsdgpu_ola *h) { return h ? h->Ne : -1; }

int tsdgpu_ola_set_response(tsdgpu_ola *h, const void *H)
{
  TSD_CHECK(h != nullptr, "ola_set_response: NULL handle");
  if (!H) {
    if (h->d_H) (void) hipFree(h->d_H);
    h->d_H = nullptr;
    if (h->d_fast) (void) hipFree(h->d_fast);
    h->d_fast = nullptr;
    if (h->d_run) (void) hipFree(h->d_run);
    h->d_run = nullptr;
    return TSDGPU_OK;
  }
  if (!h->d_H) TSD_HIP(hipMalloc((void **) &h->d_H, (size_t) h->N * sizeof(cpx)));
  TSD_HIP(hipMemcpy(h->d_H, H, (size_t) h->N * sizeof(cpx), hipMemcpyDefault));
  // the reference's default geometry without window: one fused kernel (ols.hip, ola1024_kernel) serves whole-block calls
  static const bool unfused = dev_switch("OLA_UNFUSED") != nullptr;
  if (!unfused && !h->windowed && h->N == 1024 && h->Ne == 512) {
    std::vector<cpx> Hh(1024), t3(3 * 1024);
    TSD_HIP(hipMemcpy(Hh.data(), h->d_H, 1024 * sizeof(cpx), hipMemcpyDeviceToHost));
    ola1024_tables(Hh.data(), t3.data());
    if (!h->d_fast) TSD_HIP(hipMalloc((void **) &h->d_fast, t3.size() * sizeof(cpx)));
    TSD_HIP(hipMemcpy(h->d_fast, t3.data(), t3.size() * sizeof(cpx), hipMemcpyHostToDevice));
  }
  // the windowed mode at Ne = N = 512 (the engine's defaults with a window): the in-wave pair transform (ols.hip, olaw512_kernel)
  static const bool no_w512 = dev_switch("OLAW512") != nullptr && atoi(dev_switch("OLAW512")) == 0;
  if (!unfused && !no_w512 && h->windowed && h->N == 512 && h->Ne == 512) {
    std::vector<cpx> Hh(512), tb(512 + 2 * 1024);
    TSD_HIP(hipMemcpy(Hh.data(), h->d_H, 512 * sizeof(cpx), hipMemcpyDeviceToHost));
    olaw512_tables(Hh.data(), tb.data());
    if (!h->d_fast) TSD_HIP(hipMalloc((void **) &h->d_fast, tb.size() * sizeof(cpx)));
    TSD_HIP(hipMemcpy(h->d_fast, tb.data(), tb.size() * sizeof(cpx), hipMemcpyHostToDevice));
  }
  if (!unfused && (h->windowed ? olaw_run_fits(h->N, h->Ne) : ola_run_fits(h->N, h->Ne))) {
    // the other geometries whose frame and carried block fit the LDS: one kernel too (ola_run_kernel) -- which also serves the
    // ragged calls of the default geometry (waiting samples in front of x), the in-wave kernel taking the whole-block ones
    const int N = h->N;
    std::vector<cpx> tb((size_t) N + N / 16);
    TSD_HIP(hipMemcpy(tb.data(), h->d_H, (size_t) N * sizeof(cpx), hipMemcpyDeviceToHost));
    for (int i = 0; i < N; i++) tb[i] = make_float2(tb[i].x / (float) N, tb[i].y / (float) N);
    const double PI = 3.14159265358979323846;
    for (int i = 0; i < N / 16; i++) {
      const double a = -2.0 * PI * (double) i / (double) N;
      tb[(size_t) N + i] = make_float2((float) std::cos(a), (float) std::sin(a));
    }
    if (!h->d_run) TSD_HIP(hipMalloc((void **) &h->d_run, tb.size() * sizeof(cpx)));
    TSD_HIP(hipMemcpy(h->d_run, tb.data(), tb.size() * sizeof(cpx), hipMemcpyHostToDevice));
  }
  return TSDGPU_OK;
}

int64_t tsdgpu_ola_max_out(const tsdgpu_ola *h, int64_t n)
{
  if (!h || n < 0) return -1;
  return ((h->nrest + n) / h->Ne) * h->Ne;
}

int tsdgpu_ola_analyse(tsdgpu_ola *h, const void *x, int64_t n, void **spectra, int *frames, void *stream)
{
  TSD_CHECK(h != nullptr, "ola_analyse: NULL handle");
  TSD_CHECK(n >= 0 && (n == 0 || x != nullptr), "ola_analyse: bad input");
  TSD_CHECK(h->pending_blocks < 0, "ola_analyse: the previous analyse has not been completed by ola_synthese");
  hipStream_t st = (hipStream_t) stream;
  const int Ne = h->Ne, N = h->N, per = h->windowed ? 2 : 1;
  const void *dxv = nullptr;
  int rc = stage_in(x, (size_t) n * sizeof(cpx), h->in_stage, st, &dxv);
  if (rc) return rc;
  const cpx *dx = (const cpx *) dxv;
  const int64_t tot = (int64_t) h->nrest + n, B = tot / Ne;
  TSD_CHECK(B * per <= (1 << 24), "ola_analyse: %lld blocks in one call", (long long) B);
  if (B > 0) {
    const int64_t fe = B * per * N;
    if ((rc = h->frames.reserve((size_t) fe * sizeof(cpx)))) return rc;
    if ((rc = h->spectra.reserve((size_t) fe * sizeof(cpx)))) return rc;
    const FrameSrc S{h->d_rest, dx, h->d_prev_half, h->d_fen, h->nrest, Ne, N, h->Nz, h->windowed ? 1 : 0};
    const cpx *Hf = h->fuse_response ? h->d_H : nullptr;
    h->response_applied = false;
    if (framed_fft_launch(h->plan, S, B * per, Hf, h->spectra.as<cpx>(), nullptr, st)) {
      TSD_HIP(hipGetLastError());
      h->response_applied = Hf != nullptr;
    } else {
      hipLaunchKernelGGL(ola_frame_kernel, dim3(nblk(fe)), dim3(256), 0, st, h->d_rest, h->nrest, dx, h->d_prev_half, h->d_fen,
                         h->frames.as<cpx>(), Ne, N, h->Nz, h->windowed ? 1 : 0, fe);
      TSD_HIP(hipGetLastError());
      if ((rc = tsdgpu_fft_step(h->plan, h->frames.p, h->spectra.p, (int) (B * per), 1, st))) return rc;
    }
    if (h->windowed) {
      hipLaunchKernelGGL(ola_gather_kernel, dim3(nblk(Ne / 2)), dim3(256), 0, st, h->d_rest, h->nrest, dx,
                         (B - 1) * Ne + Ne / 2, h->d_prev_half, Ne / 2);                                    // :926
      TSD_HIP(hipGetLastError());
    }
  }
  // the samples after the last whole block wait for the next call
  const int left = (int) (tot - B * Ne);
  if (B > 0) {
    if (left > 0) {
      hipLaunchKernelGGL(ola_gather_kernel, dim3(nblk(left)), dim3(256), 0, st, h->d_rest, 0, dx, B * Ne - h->nrest, h->d_rest,
                         left);                                    // lies inside x (B >= 1): rest is only written
      TSD_HIP(hipGetLastError());
    }
  } else if (n > 0) {
    if ((rc = copy_small(h->d_rest + h->nrest, dx, n, st))) return rc;
  }
  h->nrest = left;
  h->pending_blocks = (int) B;
  if (dxv != x) TSD_HIP(hipStreamSynchronize(st));   // the staging buffer is reused by the next call
  if (spectra) *spectra = B > 0 ? h->spectra.p : nullptr;
  if (frames) *frames = (int) (B * per);
  return TSDGPU_OK;
}

int tsdgpu_ola_synthese(tsdgpu_ola *h, void *y, int64_t *n_out, void *stream)
{
  TSD_CHECK(h != nullptr, "ola_synthese: NULL handle");
  TSD_CHECK(h->pending_blocks >= 0, "ola_synthese: nothing analysed");
  hipStream_t st = (hipStream_t) stream;
  const int Ne = h->Ne, N = h->N, per = h->windowed ? 2 : 1;
  // (the handle's state -- pending_blocks, cnt_ech -- is committed only once every launch of this call has
  // been accepted: a failed call leaves the analysed blocks pending, and synthese can be called again)
  const int64_t B = h->pending_blocks;
  if (n_out) *n_out = 0;
  if (B == 0) {
    h->pending_blocks = -1;
    return TSDGPU_OK;
  }
  const bool skip_first = h->windowed && h->cnt_ech < 0;
  const int64_t nout = (B - (skip_first ? 1 : 0)) * Ne;
  TSD_CHECK(y != nullptr || nout == 0, "ola_synthese: NULL output");
  int rc = tsdgpu_fft_step(h->plan, h->spectra.p, h->frames.p, (int) (B * per), 0, st);
  if (rc) return rc;
  void *dyv = nullptr;
  bool staged = false;
  if ((rc = stage_out(y, (size_t) nout * sizeof(cpx), h->out_stage, &dyv, &staged))) return rc;
  const cpx *fr = h->frames.as<cpx>();
  if (!h->windowed) {
    hipLaunchKernelGGL(ola_add_kernel, dim3(nblk(nout)), dim3(256), 0, st, fr, h->d_svg, (cpx *) dyv, Ne, N, h->Nz, nout);
    TSD_HIP(hipGetLastError());
    TSD_HIP(hipMemcpyAsync(h->d_svg, fr + (size_t) (B - 1) * N + h->Nz, (size_t) Ne * sizeof(cpx), hipMemcpyDeviceToDevice, st));
  } else {
    const OlaW w{fr, h->d_svg, Ne, N, h->Nz};
    if (nout > 0) {
      hipLaunchKernelGGL(ola_add_windowed_kernel, dim3(nblk(nout)), dim3(256), 0, st, w, h->d_last, (cpx *) dyv, skip_first ? 1 : 0,
                         nout);
      TSD_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(ola_state_windowed_kernel, dim3(nblk(Ne)), dim3(256), 0, st, w, h->d_last, B);
    TSD_HIP(hipGetLastError());
    TSD_HIP(hipMemcpyAsync(h->d_svg, fr + (size_t) (2 * B - 1) * N + h->Nz, (size_t) Ne * sizeof(cpx), hipMemcpyDeviceToDevice, st));
  }
  rc = finish_out(y, (size_t) nout * sizeof(cpx), dyv, staged, st);
  if (rc) return rc;
  h->pending_blocks = -1;
  h->cnt_ech += B * Ne;
  if (n_out) *n_out = nout;
  return TSDGPU_OK;
}

int tsdgpu_ola_apply_response(tsdgpu_ola *h, void *stream)
{
  TSD_CHECK(h != nullptr && h->pending_blocks >= 0, "ola_apply_response: nothing analysed");
  const int64_t tot = (int64_t) h->pending_blocks * (h->windowed ? 2 : 1) * h->N;
  if (!h->d_H || tot == 0 || h->response_applied) return TSDGPU_OK;
  h->response_applied = true;
  hipLaunchKernelGGL(ola_mul_kernel, dim3(nblk(tot)), dim3(256), 0, (hipStream_t) stream, h->spectra.as<cpx>(), h->d_H, h->N, tot);
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}

int tsdgpu_ola_read_spectra(tsdgpu_ola *h, void *host_dst, void *stream)
{
  TSD_CHECK(h != nullptr && h->pending_blocks >= 0, "ola_read_spectra: nothing analysed");
  const size_t bytes = (size_t) h->pending_blocks * (h->windowed ? 2 : 1) * h->N * sizeof(cpx);
  if (bytes == 0) return TSDGPU_OK;
  TSD_CHECK(host_dst != nullptr, "ola_read_spectra: NULL destination");
  TSD_HIP(hipMemcpyAsync(host_dst, h->spectra.p, bytes, hipMemcpyDefault, (hipStream_t) stream));
  TSD_HIP(hipStreamSynchronize((hipStream_t) stream));
  return TSDGPU_OK;
}

int tsdgpu_ola_write_spectra(tsdgpu_ola *h, const void *host_src, void *stream)
{
  TSD_CHECK(h != nullptr && h->pending_blocks >= 0, "ola_write_spectra: nothing analysed");
  const size_t bytes = (size_t) h->pending_blocks * (h->windowed ? 2 : 1) * h->N * sizeof(cpx);
  if (bytes == 0) return TSDGPU_OK;
  TSD_CHECK(host_src != nullptr, "ola_write_spectra: NULL source");
  TSD_HIP(hipMemcpyAsync(h->spectra.p, host_src, bytes, hipMemcpyDefault, (hipStream_t) stream));
  TSD_HIP(hipStreamSynchronize((hipStream_t) stream));
  return TSDGPU_OK;
}

int tsdgpu_ola_step(tsdgpu_ola *h, const void *x, int64_t n, void *y, int64_t *n_out, void *stream)
{
  TSD_CHECK(h != nullptr, "ola_step: NULL handle");
  // a large HOST vector goes through in chunks of whole blocks: H2D of chunk i + 1, the engine on chunk i and D2H of chunk
  // i - 1 overlap; the engine's state (waiting samples, carried block, sample counter) goes from chunk to chunk as between calls
  if (x != nullptr && y != nullptr && h->pending_blocks < 0 && (size_t) n * sizeof(cpx) >= PIPE_MIN_BYTES && host_pipe_enabled() &&
      !is_device_ptr(x) && !is_device_ptr(y) && !host_ranges_overlap(x, (size_t) n * sizeof(cpx), y, (size_t) tsdgpu_ola_max_out(h, n) * sizeof(cpx))) {
    const int Ne = h->Ne;
    return pipelined_host_step_var(
        x, n, sizeof(cpx), y, sizeof(cpx), n_out, Ne, (hipStream_t) stream, [Ne](int64_t c) { return c + Ne; },
        [h](const void *cx, void *cy, int64_t cnt, int64_t, int64_t *got, hipStream_t q) { return tsdgpu_ola_step(h, cx, cnt, cy, got, q); });
  }
  if (h->d_fast && !h->windowed && h->nrest == 0 && n >= h->Ne && h->pending_blocks < 0 && x != nullptr && y != nullptr) {
    // fast path: B whole blocks through ONE kernel, 16 B of HBM traffic per sample (see ols.hip)
    hipStream_t st = (hipStream_t) stream;
    const int Ne = h->Ne;
    const int64_t B = n / Ne, nout = B * Ne, left = n - nout;
    if (n_out) *n_out = 0;
    const void *dxv = nullptr;
    void *dyv = nullptr;
    bool staged = false;
    int rc = stage_in(x, (size_t) n * sizeof(cpx), h->in_stage, st, &dxv);
    if (rc) return rc;
    if ((rc = stage_out(y, (size_t) nout * sizeof(cpx), h->out_stage, &dyv, &staged))) return rc;
    if (dxv == dyv) {
      // in place on the device: a run re-reads the block before it, which the run before may already have overwritten
      if ((rc = h->in_stage.reserve((size_t) n * sizeof(cpx)))) return rc;
      TSD_HIP(hipMemcpyAsync(h->in_stage.p, dxv, (size_t) n * sizeof(cpx), hipMemcpyDeviceToDevice, st));
      dxv = h->in_stage.p;
    }
    const cpx *dx = (const cpx *) dxv;
    if ((rc = copy_small(h->d_rest, dx + nout, left, st))) return rc;
    if ((rc = ola1024_launch(dx, (cpx *) dyv, h->d_fast, h->d_svg, h->d_svg_tmp, B, st))) return rc;
    std::swap(h->d_svg, h->d_svg_tmp);            // (the last wave wrote the new tail beside the one the first wave read)
    if ((rc = finish_out(y, (size_t) nout * sizeof(cpx), dyv, staged, st))) return rc;
    if (dxv != x && !staged) TSD_HIP(hipStreamSynchronize(st));    // (a staged input must outlive its kernels)
    h->nrest = (int) left;
    h->cnt_ech += nout;
    if (n_out) *n_out = nout;
    return TSDGPU_OK;
  }
  if (h->d_run && !h->windowed && h->pending_blocks < 0 && n >= 0 && (n == 0 || x != nullptr) && ((int64_t) h->nrest + n) / h->Ne >= 1 && y != nullptr) {
    // any other geometry without window: B whole blocks of [rest ++ x] through ONE kernel (ola_run_kernel)
    hipStream_t st = (hipStream_t) stream;
    const int Ne = h->Ne;
    const int64_t tot = (int64_t) h->nrest + n, B = tot / Ne, nout = B * Ne, left = tot - nout;
    if (n_out) *n_out = 0;
    TSD_CHECK(B <= (1 << 24), "ola_step: %lld blocks in one call", (long long) B);
    const void *dxv = nullptr;
    void *dyv = nullptr;
    bool staged = false;
    int rc = stage_in(x, (size_t) n * sizeof(cpx), h->in_stage, st, &dxv);
    if (rc) return rc;
    if ((rc = stage_out(y, (size_t) nout * sizeof(cpx), h->out_stage, &dyv, &staged))) return rc;
    if (host_ranges_overlap(dxv, (size_t) n * sizeof(cpx), dyv, (size_t) nout * sizeof(cpx))) {
      // in place on the device: a run re-reads the block before it, which the run before may already have overwritten
      if ((rc = h->in_stage.reserve((size_t) n * sizeof(cpx)))) return rc;
      TSD_HIP(hipMemcpyAsync(h->in_stage.p, dxv, (size_t) n * sizeof(cpx), hipMemcpyDeviceToDevice, st));
      dxv = h->in_stage.p;
    }
    const cpx *dx = (const cpx *) dxv;
    // block 0 made contiguous: the waiting samples, then the head of x (rest has room for a whole block)
    if (h->nrest > 0 && (rc = copy_small(h->d_rest + h->nrest, dx, Ne - h->nrest, st))) return rc;
    if ((rc = ola_run_launch(h->nrest > 0 ? h->d_rest : dx, h->nrest, dx, (cpx *) dyv, h->d_run, h->d_svg, h->d_svg_tmp, Ne, h->N, B, st))) return rc;
    std::swap(h->d_svg, h->d_svg_tmp);
    if (left > 0) {
      // the samples after the last whole block lie inside x (B >= 1): rest is only written
      hipLaunchKernelGGL(ola_gather_kernel, dim3(nblk(left)), dim3(256), 0, st, h->d_rest, 0, dx, nout - h->nrest, h->d_rest, (int) left);
      TSD_HIP(hipGetLastError());
    }
    if ((rc = finish_out(y, (size_t) nout * sizeof(cpx), dyv, staged, st))) return rc;
    if (dxv != x && !staged) TSD_HIP(hipStreamSynchronize(st));    // (a staged input must outlive its kernels)
    h->nrest = (int) left;
    h->cnt_ech += nout;
    if (n_out) *n_out = nout;
    return TSDGPU_OK;
  }
  if (h->d_run && h->windowed && h->pending_blocks < 0 && n >= 0 && (n == 0 || x != nullptr) && ((int64_t) h->nrest + n) / h->Ne >= 1 && y != nullptr) {
    // the windowed mode: B whole blocks of [rest ++ x] through ONE kernel (olaw_run_kernel)
    hipStream_t st = (hipStream_t) stream;
    const int Ne = h->Ne;
    const int64_t tot = (int64_t) h->nrest + n, B = tot / Ne, left = tot - B * Ne;
    const int skip = h->cnt_ech < 0 ? 1 : 0;                       // (:899-901: the very first block gives no output)
    const int64_t nout = (B - skip) * Ne;
    if (n_out) *n_out = 0;
    TSD_CHECK(B <= (1 << 24), "ola_step: %lld blocks in one call", (long long) B);
    const void *dxv = nullptr;
    void *dyv = nullptr;
    bool staged = false;
    int rc = stage_in(x, (size_t) n * sizeof(cpx), h->in_stage, st, &dxv);
    if (rc) return rc;
    if ((rc = stage_out(y, (size_t) nout * sizeof(cpx), h->out_stage, &dyv, &staged))) return rc;
    if (nout > 0 && host_ranges_overlap(dxv, (size_t) n * sizeof(cpx), dyv, (size_t) nout * sizeof(cpx))) {
      // in place on the device: a run re-reads the blocks before it, which the run before may already have overwritten
      if ((rc = h->in_stage.reserve((size_t) n * sizeof(cpx)))) return rc;
      TSD_HIP(hipMemcpyAsync(h->in_stage.p, dxv, (size_t) n * sizeof(cpx), hipMemcpyDeviceToDevice, st));
      dxv = h->in_stage.p;
    }
    const cpx *dx = (const cpx *) dxv;
    // block 0 made contiguous: the waiting samples, then the head of x (rest has room for a whole block)
    if (h->nrest > 0 && (rc = copy_small(h->d_rest + h->nrest, dx, Ne - h->nrest, st))) return rc;
    if (h->d_fast && h->N == 512 && Ne == 512)
      rc = olaw512_launch(h->nrest > 0 ? h->d_rest : dx, h->nrest, dx, (cpx *) dyv, h->d_fast, h->d_fen, h->d_svg, h->d_last, h->d_prev_half,
                          h->d_svg_tmp, h->d_last_tmp, B, skip, st);
    else
      rc = olaw_run_launch(h->nrest > 0 ? h->d_rest : dx, h->nrest, dx, (cpx *) dyv, h->d_run, h->d_fen, h->d_svg, h->d_last, h->d_prev_half,
                           h->d_svg_tmp, h->d_last_tmp, Ne, h->N, B, skip, st);
    if (rc) return rc;
    std::swap(h->d_svg, h->d_svg_tmp);
    std::swap(h->d_last, h->d_last_tmp);
    // the second half of the last block waits for the next call's first frame (:926); then the samples after the last whole block
    hipLaunchKernelGGL(ola_gather_kernel, dim3(nblk(Ne / 2)), dim3(256), 0, st, h->d_rest, h->nrest, dx, (B - 1) * Ne + Ne / 2, h->d_prev_half, Ne / 2);
    TSD_HIP(hipGetLastError());
    if (left > 0) {
      hipLaunchKernelGGL(ola_gather_kernel, dim3(nblk(left)), dim3(256), 0, st, h->d_rest, 0, dx, B * Ne - h->nrest, h->d_rest, (int) left);
      TSD_HIP(hipGetLastError());
    }
    if ((rc = finish_out(y, (size_t) nout * sizeof(cpx), dyv, staged, st))) return rc;
    if (dxv != x && !staged) TSD_HIP(hipStreamSynchronize(st));    // (a staged input must outlive its kernels)
    h->nrest = (int) left;
    h->cnt_ech += B * Ne;
    if (n_out) *n_out = nout;
    return TSDGPU_OK;
  }
  void *sp = nullptr;
  int nf = 0;
  h->fuse_response = true;
  int rc = tsdgpu_ola_analyse(h, x, n, &sp, &nf, stream);
  h->fuse_response = false;
  if (rc) return rc;
  (void) sp;
  if ((rc = tsdgpu_ola_apply_response(h, stream))) return rc;
  return tsdgpu_ola_synthese(h, y, n_out, stream);
}

int tsdgpu_welch(const void *x, int64_t n, int N, const float *window, float *S, int64_t *n_segments, void *stream)
{
  TSD_CHECK(N >= 1 && N <= (1 << 24), "welch: N = %d", N);
  TSD_CHECK(n >= 0 && (n == 0 || x != nullptr) && window != nullptr && S != nullptr, "welch: bad arguments");
  hipStream_t st = (hipStream_t) stream;
  const int pas = std::max(N / 2, 1);
  int64_t nseg = 0;
  if (n > N) nseg = (n - N - 1) / pas + 1;                         // i = 0, pas, ... while i + N < n  (:13)
  if (n_segments) *n_segments = nseg;
  TSD_CHECK(nseg <= 0x7fffffff, "welch: %lld segments in one call", (long long) nseg);
  // scratch and plan borrowed for the call (the one-shot API used to spend most of its time in hipMalloc / hipFree and in
  // building the plan)
  struct Ctx {
    int dev = 0, N = 0, lots = 0;
    tsdgpu_fft *plan = nullptr;
    DevBuf xin, seg, part, wbuf, sout, tw;
    int tw_N = 0;                     // the size the transform tables in tw were made for (0: none)
    void libere() { if (plan) tsdgpu_fft_destroy(plan); xin.release(); seg.release(); part.release(); wbuf.release(); sout.release(); tw.release(); }
    size_t octets() const { return xin.cap + seg.cap + part.cap + wbuf.cap + sout.cap; }
  };
  static CtxReserve<Ctx> *reserve = new CtxReserve<Ctx>(4);
  Ctx *c = reserve->prend([N](const Ctx &k) { return k.N == N || k.tw_N == N; });
  DevBuf &xin = c->xin, &seg = c->seg, &part = c->part, &wbuf = c->wbuf, &sout = c->sout;
  const void *dxv = nullptr, *dwv = nullptr;
  void *dS = nullptr;
  bool staged = false;
  int rc = stage_in(x, (size_t) n * sizeof(cpx), xin, st, &dxv);
  if (!rc) rc = stage_in(window, (size_t) N * sizeof(float), wbuf, st, &dwv);
  if (!rc) rc = stage_out(S, (size_t) N * sizeof(float), sout, &dS, &staged);
  tsdgpu_fft *plan = nullptr;
  static const bool multi = dev_switch("OLA_UNFUSED") != nullptr;
  if (!rc && nseg > 0 && N == 1024 && !multi) {
    // N = 1024: ONE kernel on the in-wave transform keeps the running sums in registers (ols.hip, welch1024_kernel)
    if (c->tw_N != 1024) {
      std::vector<cpx> t2(2048);
      welch1024_tables(t2.data());
      c->tw_N = 0;
      rc = c->tw.reserve(t2.size() * sizeof(cpx));
      if (!rc && hipMemcpy(c->tw.p, t2.data(), t2.size() * sizeof(cpx), hipMemcpyHostToDevice) != hipSuccess)
        rc = set_err(TSDGPU_ERR_HIP, "welch: table upload failed");
      if (!rc) c->tw_N = 1024;
    }
    const int per = (int) std::min<int64_t>(64, std::max<int64_t>(1, cdiv(nseg, 2048)));       // (rounded UP: 2048 waves are resident at once; one more is a second, nearly empty round)
    const int64_t rows = cdiv(nseg, per);
    // the rows of the waves are summed in two deterministic stages (a single stage would walk thousands of rows from
    // four workgroups)
    // (groups of 8 rows over many workgroups, then 4 row lanes per bin: the 64-group form -- 35 rows per thread from 256 workgroups, then
    // 64 rows per thread from four -- took 27 + 16 us of the call's 118)
    const int rpg = (int) std::min<int64_t>(64, std::max<int64_t>(8, cdiv(rows, 96))), ngr = (int) cdiv(rows, rpg);
    if (!rc) rc = part.reserve((size_t) (rows + ngr) * N * sizeof(float));
    float *p1 = part.as<float>(), *p2 = p1 + (size_t) rows * N;
    if (!rc) rc = welch1024_launch((const cpx *) dxv, (const float *) dwv, c->tw.as<cpx>(), p1, nseg, per, st);
    if (!rc) {
      hipLaunchKernelGGL(welch_sum_groups_kernel, dim3(nblk(N), (unsigned) ngr), dim3(256), 0, st, p1, p2, N, (int) rows, rpg);
      hipLaunchKernelGGL(welch_sum_shift_kernel, dim3((unsigned) cdiv(N, 64)), dim3(256), 0, st, p2, (float *) dS, N, ngr, 0);
      if (hipGetLastError() != hipSuccess) rc = set_err(TSDGPU_ERR_HIP, "welch: launch failed");
    }
    if (!rc) rc = finish_out(S, (size_t) N * sizeof(float), dS, staged, st);
    (void) hipStreamSynchronize(st);
    reserve->rend(c);
    return rc;
  }
  if (!rc && nseg > 0 && !multi && N >= 16 && (N & (N - 1)) == 0 && ola_run_fits(N, N / 2)) {
    // the other powers of two up to 16384: ONE kernel on the LDS transform keeps the running sums in registers (welch_run_kernel)
    if (c->tw_N != N) {
      std::vector<cpx> tw((size_t) N / 16);
      const double PI = 3.14159265358979323846;
      for (int i = 0; i < N / 16; i++) {
        const double a = -2.0 * PI * (double) i / (double) N;
        tw[i] = make_float2((float) std::cos(a), (float) std::sin(a));
      }
      c->tw_N = 0;
      rc = c->tw.reserve(tw.size() * sizeof(cpx));
      if (!rc && hipMemcpy(c->tw.p, tw.data(), tw.size() * sizeof(cpx), hipMemcpyHostToDevice) != hipSuccess)
        rc = set_err(TSDGPU_ERR_HIP, "welch: table upload failed");
      if (!rc) c->tw_N = N;
    }
    const OlaRunGeom g = ola_run_geom(N, N / 2);
    static const int cus = []() {
      int dev = 0, n = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
      return n > 0 ? n : 256;
    }();
    // one pass of the chip: a run per (resident workgroup x transform)
    const int64_t places = (int64_t) cus * (g.threads == 256 ? 3 : 1) * g.T;
    const int per = (int) std::min<int64_t>(64, std::max<int64_t>(1, cdiv(nseg, places)));
    const int64_t rows = cdiv(nseg, per);
    const int rpg = (int) std::min<int64_t>(64, std::max<int64_t>(8, cdiv(rows, 96))), ngr = (int) cdiv(rows, rpg);
    if (!rc) rc = part.reserve((size_t) (rows + ngr) * N * sizeof(float));
    float *p1 = part.as<float>(), *p2 = p1 + (size_t) rows * N;
    if (!rc) rc = welch_run_launch((const cpx *) dxv, (const float *) dwv, c->tw.as<cpx>(), p1, N, nseg, per, st);
    if (!rc) {
      hipLaunchKernelGGL(welch_sum_groups_kernel, dim3(nblk(N), (unsigned) ngr), dim3(256), 0, st, p1, p2, N, (int) rows, rpg);
      hipLaunchKernelGGL(welch_sum_shift_kernel, dim3((unsigned) cdiv(N, 64)), dim3(256), 0, st, p2, (float *) dS, N, ngr, 0);
      if (hipGetLastError() != hipSuccess) rc = set_err(TSDGPU_ERR_HIP, "welch: launch failed");
    }
    if (!rc) rc = finish_out(S, (size_t) N * sizeof(float), dS, staged, st);
    (void) hipStreamSynchronize(st);
    reserve->rend(c);
    return rc;
  }
  if (!rc && nseg > 0 && !(c->plan && c->N == N)) {        // (the batch count given at creation is only a hint)
    if (c->plan) tsdgpu_fft_destroy(c->plan);
    c->plan = nullptr;
    c->N = c->lots = 0;
    rc = tsdgpu_fft_create(&c->plan, N, (int) nseg);
    if (!rc) { c->N = N; c->lots = (int) nseg; }
  }
  plan = c->plan;
  if (!rc && nseg == 0) {
    if (hipMemsetAsync(dS, 0, (size_t) N * sizeof(float), st) != hipSuccess) rc = set_err(TSDGPU_ERR_HIP, "welch: memset failed");
  } else if (!rc) {
    // sizes whose plan is the wave-level Bluestein (N = 1000 = 8 x 125, odd N <= 511 ...): that kernel frames, windows, transforms and
    // sums |X|^2 per workgroup; what is left is the reduction of its partial rows
    int64_t rows = 0;
    if (!multi) rc = fft_blu_framed_launch(plan, (const cpx *) dxv, pas, (const float *) dwv, nseg, nullptr, 0, &rows, st);
    if (!rc && rows > 0) {
      // (one row per workgroup of the persistent grid: a few hundred; groups of 8 rows, then 4 row lanes per bin)
      const int rpg = (int) std::min<int64_t>(64, std::max<int64_t>(8, cdiv(rows, 96))), ngr = (int) cdiv(rows, rpg);
      rc = part.reserve((size_t) (rows + ngr) * N * sizeof(float));
      float *p1 = part.as<float>(), *p2 = p1 + (size_t) rows * N;
      if (!rc) rc = fft_blu_framed_launch(plan, (const cpx *) dxv, pas, (const float *) dwv, nseg, p1, rows, &rows, st);
      if (!rc) {
        hipLaunchKernelGGL(welch_sum_groups_kernel, dim3(nblk(N), (unsigned) ngr), dim3(256), 0, st, p1, p2, N, (int) rows, rpg);
        hipLaunchKernelGGL(welch_sum_shift_kernel, dim3((unsigned) cdiv(N, 64)), dim3(256), 0, st, p2, (float *) dS, N, ngr, 1);
        if (hipGetLastError() != hipSuccess) rc = set_err(TSDGPU_ERR_HIP, "welch: launch failed");
      }
      if (!rc) rc = finish_out(S, (size_t) N * sizeof(float), dS, staged, st);
      (void) hipStreamSynchronize(st);
      reserve->rend(c);
      return rc;
    }
    if (rc) {
      (void) hipStreamSynchronize(st);
      reserve->rend(c);
      return rc;
    }
    const int64_t total = nseg * N;
    // enough groups to fill the chip whatever N: N/256 x groups workgroups, <= 512 partial sums per bin
    const int groups = (int) std::max<int64_t>(1, std::min<int64_t>(512, std::min<int64_t>(cdiv(nseg, 16), cdiv(262144, N))));
    const int64_t per_group = cdiv(nseg, groups);
    rc = seg.reserve((size_t) total * sizeof(cpx));
    if (!rc) rc = part.reserve((size_t) (groups + 64) * N * sizeof(float));       // + the rows of the second summation stage
    bool fused = false;
    if (!rc) {
      // fused: segments gathered and windowed by the transform itself, which stores |X|^2 only
      const FrameSrc S{nullptr, (const cpx *) dxv, nullptr, (const float *) dwv, 0, pas, N, 0, 2};
      fused = framed_fft_launch(plan, S, nseg, nullptr, nullptr, seg.as<float>(), st);
    }
    if (!rc && !fused) {
      hipLaunchKernelGGL(welch_frame_kernel, dim3(nblk(total)), dim3(256), 0, st, (const cpx *) dxv, (const float *) dwv,
                         seg.as<cpx>(), N, pas, total);
      if (hipGetLastError() != hipSuccess) rc = set_err(TSDGPU_ERR_HIP, "welch: launch failed");
      if (!rc) rc = tsdgpu_fft_step(plan, seg.p, seg.p, (int) nseg, 1, st);
    }
    if (!rc) {
      if (fused)
        hipLaunchKernelGGL(welch_power_sum_kernel, dim3(nblk(N), (unsigned) groups), dim3(256), 0, st, seg.as<float>(),
                           part.as<float>(), N, nseg, per_group);
      else
        hipLaunchKernelGGL(welch_power_kernel, dim3(nblk(N), (unsigned) groups), dim3(256), 0, st, seg.as<cpx>(), part.as<float>(), N,
                           nseg, per_group);
      if (groups > 64) {
        // (two deterministic stages: one stage would walk hundreds of rows from N / 256 workgroups)
        const int rpg = (int) cdiv(groups, 64), ngr = (int) cdiv(groups, rpg);
        float *p2 = part.as<float>() + (size_t) groups * N;
        hipLaunchKernelGGL(welch_sum_groups_kernel, dim3(nblk(N), (unsigned) ngr), dim3(256), 0, st, part.as<float>(), p2, N, groups, rpg);
        hipLaunchKernelGGL(welch_sum_kernel, dim3(nblk(N)), dim3(256), 0, st, p2, (float *) dS, N, ngr);
      } else {
        hipLaunchKernelGGL(welch_sum_kernel, dim3(nblk(N)), dim3(256), 0, st, part.as<float>(), (float *) dS, N, groups);
      }
      if (hipGetLastError() != hipSuccess) rc = set_err(TSDGPU_ERR_HIP, "welch: launch failed");
    }
  }
  if (!rc) rc = finish_out(S, (size_t) N * sizeof(float), dS, staged, st);
  // the scratch buffers die with the call: wait for the work that uses them
  (void) hipStreamSynchronize(st);
  reserve->rend(c);
  return rc;
}

// ---- rt_spectrum on the device (fourier.cc:1162-1342) ------------------------------------------------
// Blocks of BS = nsubs x Nf samples: window x sub-block -> transform -> |X|^2 (fftshift order) summed over the sub-blocks and
// over nmeans blocks -> dB.  Only x goes up; Ns floats come back per nmeans blocks.  The sums of a group of blocks that is
// not complete at the end of a call stay on the device (d_acc) until the next call.
struct tsdgpu_spectrum {
  int BS = 0, nsubs = 1, nmeans = 1, Nf = 0, Ns = 0, sweep = 0, step = 0;
  int cnt = 0;                          // blocks accumulated since the last spectrum
  bool run_kernel = false;              // power-of-two Nf that fits the LDS transform: ONE kernel (welch_run_kernel)
  float *d_tab = nullptr;               // window (Nf) | masque (Nf) | mag_cnt (Ns) | acc A | acc B (ncls x Nf each) | TW (Nf / 16 complex)
  float *d_win = nullptr, *d_mask = nullptr, *d_cnt = nullptr, *d_acc[2] = {nullptr, nullptr};
  cpx *d_tw = nullptr;
  int cur = 0;
  tsdgpu_fft *plan = nullptr;           // the other sizes: batched plan
  DevBuf in_stage, out_stage, part, seg;
};

int tsdgpu_spectrum_create(tsdgpu_spectrum **out, int BS, int nsubs, int nmeans, const float *window_host, int sweep_active,
                           int sweep_step, const float *mask_host)
{
  TSD_CHECK(out != nullptr, "spectrum_create: out is NULL");
  *out = nullptr;
  // (BS need not be a multiple of nsubs: Nf = BS / nsubs like SpectrumConfig::Nf, :1150-1153, and the trailing BS - nsubs Nf
  // samples of every block are never read -- `x.segment(i * Nf, Nf)`, :1254)
  TSD_CHECK(BS >= 1 && nsubs >= 1 && nmeans >= 1 && BS >= nsubs, "spectrum_create: BS = %d, nsubs = %d, nmeans = %d", BS, nsubs, nmeans);
  TSD_CHECK(window_host != nullptr, "spectrum_create: NULL window");
  TSD_CHECK(!sweep_active || sweep_step >= 0, "spectrum_create: sweep step %d", sweep_step);
  tsdgpu_spectrum *h = new tsdgpu_spectrum();
  h->BS = BS; h->nsubs = nsubs; h->nmeans = nmeans;
  const int Nf = h->Nf = BS / nsubs;
  // One sub-block per block takes the reference's `sinon` branch (:1272-1277): no masque, no shifted accumulation, and
  // mag_cnt = max(masque, 1) = 1 everywhere -- the plain spectrum, sweep.active or not.
  h->sweep = (sweep_active && nsubs > 1) ? 1 : 0;
  h->step = sweep_step;
  const int Ns = h->Ns = h->sweep ? Nf + (nsubs - 1) * sweep_step : Nf;        // SpectrumConfig::Ns (:1157-1161)
  const int ncls = h->sweep ? nsubs : 1;
  h->run_kernel = Nf >= 16 && (Nf & (Nf - 1)) == 0 && ola_run_fits(Nf, Nf / 2) && dev_switch("OLA_UNFUSED") == nullptr;
  // ONE allocation, ONE upload of its host image
  const size_t ntw = h->run_kernel ? (size_t) Nf / 16 : 0;
  std::vector<float> img((size_t) 2 * Nf + Ns + 2 * (size_t) ncls * Nf + 2 * ntw, 0.f);
  float *win = img.data(), *mask = win + Nf, *cnt = mask + Nf;
  for (int k = 0; k < Nf; k++) {
    win[k] = window_host[k];
    mask[k] = mask_host ? mask_host[k] : 1.f;
  }
  if (h->sweep) {
    // mag_cnt (:1196-1202): how many unmasked sub-block bins fall on each bin of the swept spectrum, at least 1
    for (int i = 0; i < nsubs; i++)
      for (int k = 0; k < Nf; k++) cnt[(size_t) i * sweep_step + k] += mask[k];
    for (int j = 0; j < Ns; j++) cnt[j] = std::max(cnt[j], 1.0f);
  } else {
    for (int j = 0; j < Ns; j++) cnt[j] = 1.f;
  }
  float *tw = cnt + Ns + 2 * (size_t) ncls * Nf;
  const double PI = 3.14159265358979323846;
  for (size_t i = 0; i < ntw; i++) {
    const double a = -2.0 * PI * (double) i / (double) Nf;
    tw[2 * i] = (float) std::cos(a);
    tw[2 * i + 1] = (float) std::sin(a);
  }
  int rc = TSDGPU_OK;
  if (hipMalloc((void **) &h->d_tab, img.size() * sizeof(float)) != hipSuccess)
    rc = set_err(TSDGPU_ERR_HIP, "spectrum_create: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
  else if (hipMemcpy(h->d_tab, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
    rc = set_err(TSDGPU_ERR_HIP, "spectrum_create: upload failed: %s", hipGetErrorString(hipGetLastError()));
  if (!rc) {
    h->d_win = h->d_tab;
    h->d_mask = h->d_win + Nf;
    h->d_cnt = h->d_mask + Nf;
    h->d_acc[0] = h->d_cnt + Ns;
    h->d_acc[1] = h->d_acc[0] + (size_t) ncls * Nf;
    h->d_tw = (cpx *) (h->d_acc[1] + (size_t) ncls * Nf);
    if (!h->run_kernel) rc = tsdgpu_fft_create(&h->plan, Nf, nsubs * std::min(nmeans, 64));
  }
  if (rc) {
    tsdgpu_spectrum_destroy(h);
    return rc;
  }
  *out = h;
  return TSDGPU_OK;
}

int tsdgpu_spectrum_bins(const tsdgpu_spectrum *h) { return h ? h->Ns : -1; }
int tsdgpu_spectrum_pending(const tsdgpu_spectrum *h) { return h ? h->cnt : -1; }

static int spectrum_step_blocks(tsdgpu_spectrum *h, const void *x, int64_t nblocks, float *y, int64_t y_capacity, int64_t *n_spectra, void *stream);

int tsdgpu_spectrum_step(tsdgpu_spectrum *h, const void *x, int64_t nblocks, float *y, int64_t y_capacity, int64_t *n_spectra, void *stream)
{
  TSD_CHECK(h != nullptr, "spectrum_step: NULL handle");
  TSD_CHECK(nblocks >= 0, "spectrum_step: negative block count");
  if (n_spectra) *n_spectra = 0;
  if (nblocks == 0) return TSDGPU_OK;
  TSD_CHECK(x != nullptr, "spectrum_step: NULL input");
  if (nblocks == 1 || h->BS == h->nsubs * h->Nf) return spectrum_step_blocks(h, x, nblocks, y, y_capacity, n_spectra, stream);
  // BS is not a multiple of nsubs: the segments of a block are contiguous, the blocks are BS apart -- block by block
  const int64_t nout = (h->cnt + nblocks) / h->nmeans;
  TSD_CHECK(nout <= y_capacity, "spectrum_step: %lld spectra completed, room for %lld", (long long) nout, (long long) y_capacity);
  int64_t done = 0;
  for (int64_t b = 0; b < nblocks; b++) {
    int64_t got = 0;
    const int rc = spectrum_step_blocks(h, (const cpx *) x + (size_t) b * h->BS, 1, y ? y + (size_t) done * h->Ns : nullptr, y_capacity - done, &got, stream);
    if (rc) return rc;
    done += got;
  }
  if (n_spectra) *n_spectra = done;
  return TSDGPU_OK;
}

static int spectrum_step_blocks(tsdgpu_spectrum *h, const void *x, int64_t nblocks, float *y, int64_t y_capacity, int64_t *n_spectra, void *stream)
{
  hipStream_t st = (hipStream_t) stream;
  const int Nf = h->Nf, Ns = h->Ns, nsubs = h->nsubs, nmeans = h->nmeans;
  const int64_t B = nblocks, S = B * nsubs;
  const int64_t nout = (h->cnt + B) / nmeans;                  // spectra completed by this call
  TSD_CHECK(nout <= y_capacity, "spectrum_step: %lld spectra completed, room for %lld", (long long) nout, (long long) y_capacity);
  TSD_CHECK(nout == 0 || y != nullptr, "spectrum_step: NULL output");
  const int64_t be0 = std::min<int64_t>(B, nmeans - h->cnt);
  const int64_t G = 1 + (B > be0 ? cdiv(B - be0, nmeans) : 0);
  TSD_CHECK(G <= 65535 && S <= 0x7fffffff, "spectrum_step: %lld blocks in one call", (long long) B);
  const void *dxv = nullptr;
  void *dyv = nullptr;
  bool staged = false;
  int rc = stage_in(x, (size_t) B * h->BS * sizeof(cpx), h->in_stage, st, &dxv);
  if (rc) return rc;
  rc = stage_out(y, (size_t) nout * Ns * sizeof(float), h->out_stage, &dyv, &staged);
  if (rc) return rc;
  // runs: a class = the segments that are summed into one spectrum (sweep: one per sub-block index)
  const int ncls = h->sweep ? nsubs : 1;
  const int64_t per_class = h->sweep ? nmeans : (int64_t) nmeans * nsubs;         // segments of a full class
  static const int cus = []() {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  int64_t places = 4096;
  if (h->run_kernel) {
    const OlaRunGeom g = ola_run_geom(Nf, Nf / 2);
    places = (int64_t) cus * (g.threads == 256 ? 3 : 1) * g.T;
  }
  const int per = (int) std::min<int64_t>(std::min<int64_t>(64, per_class), std::max<int64_t>(1, cdiv(S, places)));
  const int rpg = (int) cdiv(per_class, per);
  const int64_t nruns = G * ncls * rpg;
  rc = h->part.reserve((size_t) nruns * Nf * sizeof(float));
  if (rc) return rc;
  const SegRuns M = {h->sweep ? 2 : 1, per, Nf, nsubs, nmeans, h->cnt, rpg, S, B};
  if (h->run_kernel) {
    rc = seg_runs_launch((const cpx *) dxv, h->d_win, h->d_tw, h->part.as<float>(), Nf, M, nruns, st);
    if (rc) return rc;
  } else {
    rc = h->seg.reserve((size_t) S * Nf * sizeof(cpx));
    if (rc) return rc;
    const int64_t total = S * Nf;
    hipLaunchKernelGGL(spec_frame_kernel, dim3(nblk(total)), dim3(256), 0, st, (const cpx *) dxv, h->d_win, h->seg.as<cpx>(), Nf, total);
    TSD_HIP(hipGetLastError());
    rc = tsdgpu_fft_step(h->plan, h->seg.p, h->seg.p, (int) S, 1, st);
    if (rc) return rc;
    hipLaunchKernelGGL(spec_power_rows_kernel, dim3(nblk(Nf), (unsigned) nruns), dim3(256), 0, st, h->seg.as<cpx>(), h->part.as<float>(), Nf, M, nruns);
    TSD_HIP(hipGetLastError());
  }
  const SpecFin F = {Nf, Ns, nsubs, nmeans, h->cnt, rpg, h->sweep, h->step, (int) G, B};
  hipLaunchKernelGGL(spec_finish_kernel, dim3(nblk(std::max(Ns, Nf)), (unsigned) G), dim3(256), 0, st, h->part.as<float>(), h->d_acc[h->cur],
                     h->d_acc[h->cur ^ 1], h->d_mask, h->d_cnt, (float *) dyv, F);
  TSD_HIP(hipGetLastError());
  h->cur ^= 1;
  h->cnt = (int) ((h->cnt + B) % nmeans);
  if (n_spectra) *n_spectra = nout;
  rc = finish_out(y, (size_t) nout * Ns * sizeof(float), dyv, staged, st);
  if (rc) return rc;
  if (dxv != x && !staged) TSD_HIP(hipStreamSynchronize(st));       // (a staged input must outlive its kernels)
  return TSDGPU_OK;
}

int tsdgpu_spectrum_reset(tsdgpu_spectrum *h, void *stream)
{
  TSD_CHECK(h != nullptr, "spectrum_reset: NULL handle");
  const int ncls = h->sweep ? h->nsubs : 1;
  TSD_HIP(hipMemsetAsync(h->d_acc[0], 0, 2 * (size_t) ncls * h->Nf * sizeof(float), (hipStream_t) stream));
  h->cnt = 0;
  return TSDGPU_OK;
}

int tsdgpu_spectrum_destroy(tsdgpu_spectrum *h)
{
  if (!h) return TSDGPU_OK;
  if (h->plan) tsdgpu_fft_destroy(h->plan);
  if (h->d_tab) (void) hipFree(h->d_tab);
  h->in_stage.release();
  h->out_stage.release();
  h->part.release();
  h->seg.release();
  delete h;
  return TSDGPU_OK;
}

int tsdgpu_ola_destroy(tsdgpu_ola *h)
{
  if (!h) return TSDGPU_OK;
  if (h->plan) tsdgpu_fft_destroy(h->plan);
  for (void *q : {(void *) h->d_fen, (void *) h->d_H, (void *) h->d_bloc, (void *) h->d_fast, (void *) h->d_run})
    if (q) (void) hipFree(q);
  h->frames.release();
  h->spectra.release();
  h->in_stage.release();
  h->out_stage.release();
  delete h;
  return TSDGPU_OK;
}

}  // extern "C"
