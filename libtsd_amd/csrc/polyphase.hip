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
#include <algorithm>
#include <array>
#include <cmath>
#include <complex>
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
// Decimators of small rate (2 <= R <= PF_ROWS_MAXR; the half-band stages of filtre_reechan are R = 2): the same sums by
// POLYPHASE ROWS.  The fused kernel above lets lane t read the staged sample (t R + c - k) for tap k: a lane stride of R
// samples on the LDS banks (2-way conflicts at R = 2 -- decimation by 2 ran slower than by 3) and one tap read per product.
// Here the samples are staged phase-major -- row p = (index mod R), column = index / R -- and the taps regrouped by row
// (row r: the taps k = r, r + R, ..., oldest first; they meet the samples of phase (c - r) mod R), so that
//     y[o0 + t] = sum_r sum_m  gr[r][m] * X[phase_r][t + col0_r + m]
// reads consecutive columns in consecutive lanes (conflict-free), four taps per broadcast 16-byte read, and skips the rows
// and quadruples of taps that are zero: the half-band filter, half of whose taps vanish, costs a quarter of the LDS cycles.
// (The products are summed class by class instead of oldest sample first: same values to rounding, and the same bits
// however the stream is cut in calls.)
constexpr int PF_ROWS_MAXR = 16;
template <typename T>
__global__ __launch_bounds__(256) void decim_rows_kernel(const T *__restrict__ x, const T *__restrict__ hist, T *__restrict__ y,
                                                         const float *__restrict__ g, int R, int W, int64_t start, int HW, int64_t n,
                                                         int64_t nout, int TO, int Lr, int pitch)
{
  extern __shared__ __attribute__((aligned(16))) char dr_raw[];
  float *gr = reinterpret_cast<float *>(dr_raw);                          // R rows of Lr taps (Lr a multiple of 4)
  int *range = reinterpret_cast<int *>(gr + R * Lr);                      // per row: first and one-past-last non-zero quadruple
  T *X = reinterpret_cast<T *>(range + 2 * PF_ROWS_MAXR);                 // R rows of `pitch` samples
  const int t = threadIdx.x;
  const int64_t o0 = (int64_t) blockIdx.x * TO, o1 = min(o0 + TO, nout);
  const int64_t oldest = o0 * R + start - (W - 1);                        // oldest sample of output o0 (may be negative: history)
  const int64_t i_lo = (oldest >= 0 ? oldest / R : -((-oldest + R - 1) / R)) * R;      // floored to a multiple of R
  const int cc = (int) (o0 * R + start - i_lo);                           // staged index of output o0's newest sample
  // taps by class r = k mod R, oldest first: gr[r][mm] = g[r + (jmax_r - mm) R] -- the same table whatever the call's
  // alignment, so that the order of the sums, hence every output bit, does not depend on how a stream is cut in calls
  for (int idx = t; idx < R * Lr; idx += 256) {
    const int r = idx / Lr, mm = idx - r * Lr;
    const int jmax = r < W ? (W - 1 - r) / R : -1;
    gr[idx] = mm <= jmax ? g[r + (jmax - mm) * R] : 0.f;
  }
  // samples, phase-major; everything past the data is zero (the padded taps multiply it)
  for (int i0 = t; i0 < R * pitch; i0 += 256 * 8) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = i0 + 256 * u;
      const int64_t idx = i_lo + i;
      v[u] = pf_zero(T{});
      if (i < R * pitch) {
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
      if (i < R * pitch) X[(i % R) * pitch + i / R] = v[u];
    }
  }
  __syncthreads();
  if (t < R) {
    int a = Lr / 4, b = 0;
    for (int q = 0; q < Lr / 4; q++) {
      const float *w = gr + t * Lr + 4 * q;
      if (w[0] != 0.f || w[1] != 0.f || w[2] != 0.f || w[3] != 0.f) {
        a = min(a, q);
        b = q + 1;
      }
    }
    range[2 * t] = a;
    range[2 * t + 1] = b;
  }
  __syncthreads();
  for (int tt = t; tt < (int) (o1 - o0); tt += 256) {
    T acc = pf_zero(T{});
    for (int r = 0; r < R; r++) {
      // class r meets the samples of phase (cc - r) mod R, its oldest one in column (cc - r - phase) / R - jmax_r
      const int jmax = r < W ? (W - 1 - r) / R : 0;
      const int ph = (cc - r) % R, col0 = (cc - r - ph) / R - jmax;
      const T *xr = X + ph * pitch + tt + col0;
      const float *gp = gr + r * Lr;
      const int qa = range[2 * r], qb = range[2 * r + 1];
      for (int q = qa; q < qb; q++) {
        const float4 w = *reinterpret_cast<const float4 *>(gp + 4 * q);
        acc = pf_mac(acc, w.x, xr[4 * q]);
        acc = pf_mac(acc, w.y, xr[4 * q + 1]);
        acc = pf_mac(acc, w.z, xr[4 * q + 2]);
        acc = pf_mac(acc, w.w, xr[4 * q + 3]);
      }
    }
    y[o0 + tt] = acc;
  }
}

// ---- decimators of rate 2, 4, 8 and up to 64 taps (the half-band stages of filtre_reechan, filtre_rif_decim): the direct FIR kernel's
// scheme (fir.hip) evaluated on the kept positions only.  polyfir_fused_kernel spends ~8 instructions per tap (a run-time tap loop, one
// LDS sample read at a lane stride of R samples -- 2-way conflicts at R = 2 -- and one LDS tap read per product) and runs the 15-tap
// stages at 0.51-0.60 of 8 TB/s where the same taps at full rate reach 0.73.  Here, as in fir_direct_kernel: a lane owns a 64-B segment
// of consecutive input positions (segments 80 B apart: conflict-free ds_read_b128), slides a two-segment register window over the taps
// (wave-uniform scalar loads) and keeps the RS / DEC positions of its segment that survive the decimation -- the tile starts on a kept
// position, so those are r = 0, DEC, 2 DEC ... at compile time; 16-B global loads, outputs back through LDS for coalesced stores.
// Same products in the same order (oldest sample first) as the fused kernel.  hrev: the taps reversed and zero-padded to KP.
template <typename T, int RS, int DEC>
__global__ __launch_bounds__(256) void decim_direct_kernel(const T *__restrict__ x, const T *__restrict__ hist, T *__restrict__ y,
                                                           const float *__restrict__ hrev, int KP, int start, int HW, int64_t n, int64_t nout)
{
  constexpr int THREADS = 256, TILE = THREADS * RS, VEC = 16 / (int) sizeof(T), P = VEC, SP = RS + P, RO = RS / DEC;
  static_assert(RS * sizeof(T) == 64 && RS % DEC == 0, "one lane segment is 64 bytes, a whole number of kept positions");
  extern __shared__ __attribute__((aligned(16))) char dd_raw[];
  T *L = reinterpret_cast<T *>(dd_raw);
  const int64_t tile0 = (int64_t) start + (int64_t) blockIdx.x * TILE;      // first position of the tile (a kept one)
  const int H = KP, total = TILE + H;                                       // staged samples 1 .. total - 1: position tile0 - H + s
  const bool fast = (((uintptr_t) x) & (sizeof(T) - 1)) == 0 && tile0 - H >= 0 && tile0 + TILE + VEC <= n;
  if (fast) {
    struct __attribute__((aligned(4))) f4u { float x, y, z, w; };
    const f4u *xs = reinterpret_cast<const f4u *>(x + (tile0 - H + 1));
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
          *reinterpret_cast<float4 *>(L + q + (q / RS) * P) = make_float4(q4[u].x, q4[u].y, q4[u].z, q4[u].w);
        }
      }
    }
  } else {
    for (int s = threadIdx.x + 1; s < total; s += THREADS) {
      const int64_t g = tile0 - H + (int64_t) s;
      T v = pf_zero(T{});
      if (g < 0) {
        if (g >= -(int64_t) HW) v = hist[HW + g];
      } else if (g < n) {
        v = x[g];
      }
      const int q = s - 1;
      L[q + (q / RS) * P] = v;
    }
  }
  __syncthreads();
  // lane window: w[i] = sample t RS + 1 + i; kept position r (a multiple of DEC): out = sum_j hrev[j] w[r + j]
  const T *Lw = L + threadIdx.x * SP;
  T acc[RO], A[RS], B[RS];
  auto load_seg = [&](T (&dst)[RS], const T *seg) {
#pragma unroll
    for (int v4 = 0; v4 < RS / VEC; v4++) {
      const float4 q4 = *reinterpret_cast<const float4 *>(seg + v4 * VEC);
      const T *e = reinterpret_cast<const T *>(&q4);
#pragma unroll
      for (int k = 0; k < VEC; k++) dst[v4 * VEC + k] = e[k];
    }
  };
#pragma unroll
  for (int r = 0; r < RO; r++) acc[r] = pf_zero(T{});
  load_seg(A, Lw);
  const int nchunk = KP / RS;                                // even by construction
  for (int c = 0; c < nchunk; c += 2) {
    const float *h0 = hrev + c * RS;
    load_seg(B, Lw + (c + 1) * SP);
#pragma unroll
    for (int jj = 0; jj < RS; jj++) {
      const float hv = h0[jj];
#pragma unroll
      for (int r = 0; r < RO; r++) {
        const int idx = r * DEC + jj;
        acc[r] = pf_mac(acc[r], hv, idx < RS ? A[idx] : B[idx - RS]);
      }
    }
    load_seg(A, Lw + (c + 2) * SP);                          // (the last refill reads two over-allocated segments, never used)
#pragma unroll
    for (int jj = 0; jj < RS; jj++) {
      const float hv = h0[RS + jj];
#pragma unroll
      for (int r = 0; r < RO; r++) {
        const int idx = r * DEC + jj;
        acc[r] = pf_mac(acc[r], hv, idx < RS ? B[idx] : A[idx - RS]);
      }
    }
  }
  // outputs of the tile: positions tile0 + DEC q, q < TILE / DEC -> y[(tile0 - start) / DEC + q]; back through LDS, coalesced
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RO; r++) L[threadIdx.x * SP + r] = acc[r];
  __syncthreads();
  const int64_t ob = (int64_t) blockIdx.x * (TILE / DEC);
  for (int q = threadIdx.x; q < TILE / DEC; q += THREADS)
    if (ob + q < nout) y[ob + q] = L[(q / RO) * SP + (q % RO)];
}

// ---- upsamplers of rate 2, 4 (filtre_rif_ups: the interpolating stages of filtre_reechan), branches of up to 32 taps: the same scheme with
// NPH = RU accumulators per position -- y[a RU + i] = sum_k g[i][k] x[a - k]: the RU branches share the lane's register window.  The
// RS x RU outputs of a lane (contiguous) return through LDS segments of an odd number of 16-B units for coalesced stores.
// hrev: RU rows of KP reversed, zero-padded branch taps.
template <typename T, int RS, int RU>
__global__ __launch_bounds__(256) void ups_direct_kernel(const T *__restrict__ x, const T *__restrict__ hist, T *__restrict__ y,
                                                         const float *__restrict__ hrev, int KP, int HW, int64_t n, int64_t nout)
{
  constexpr int THREADS = 256, TILE = THREADS * RS, VEC = 16 / (int) sizeof(T), P = VEC, SP = RS + P, OS = RS * RU + VEC;
  extern __shared__ __attribute__((aligned(16))) char du_raw[];
  T *L = reinterpret_cast<T *>(du_raw);
  const int64_t tile0 = (int64_t) blockIdx.x * TILE;
  const int H = KP, total = TILE + H;
  const bool fast = tile0 - H >= 0 && tile0 + TILE + VEC <= n && ((((uintptr_t) x) & 15) == 0);
  if (fast) {
    struct __attribute__((aligned(4))) f4u { float x, y, z, w; };
    const f4u *xs = reinterpret_cast<const f4u *>(x + (tile0 - H + 1));
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
          *reinterpret_cast<float4 *>(L + q + (q / RS) * P) = make_float4(q4[u].x, q4[u].y, q4[u].z, q4[u].w);
        }
      }
    }
  } else {
    for (int s = threadIdx.x + 1; s < total; s += THREADS) {
      const int64_t g = tile0 - H + (int64_t) s;
      T v = pf_zero(T{});
      if (g < 0) {
        if (g >= -(int64_t) HW) v = hist[HW + g];
      } else if (g < n) {
        v = x[g];
      }
      const int q = s - 1;
      L[q + (q / RS) * P] = v;
    }
  }
  __syncthreads();
  const T *Lw = L + threadIdx.x * SP;
  T acc[RU][RS], A[RS], B[RS];
  auto load_seg = [&](T (&dst)[RS], const T *seg) {
#pragma unroll
    for (int v4 = 0; v4 < RS / VEC; v4++) {
      const float4 q4 = *reinterpret_cast<const float4 *>(seg + v4 * VEC);
      const T *e = reinterpret_cast<const T *>(&q4);
#pragma unroll
      for (int k = 0; k < VEC; k++) dst[v4 * VEC + k] = e[k];
    }
  };
#pragma unroll
  for (int i = 0; i < RU; i++)
#pragma unroll
    for (int r = 0; r < RS; r++) acc[i][r] = pf_zero(T{});
  load_seg(A, Lw);
  const int nchunk = KP / RS;                                // even by construction
  for (int c = 0; c < nchunk; c += 2) {
    load_seg(B, Lw + (c + 1) * SP);
#pragma unroll
    for (int jj = 0; jj < RS; jj++)
#pragma unroll
      for (int i = 0; i < RU; i++) {
        const float hv = hrev[i * KP + c * RS + jj];
#pragma unroll
        for (int r = 0; r < RS; r++) {
          const int idx = r + jj;
          acc[i][r] = pf_mac(acc[i][r], hv, idx < RS ? A[idx] : B[idx - RS]);
        }
      }
    load_seg(A, Lw + (c + 2) * SP);
#pragma unroll
    for (int jj = 0; jj < RS; jj++)
#pragma unroll
      for (int i = 0; i < RU; i++) {
        const float hv = hrev[i * KP + (c + 1) * RS + jj];
#pragma unroll
        for (int r = 0; r < RS; r++) {
          const int idx = r + jj;
          acc[i][r] = pf_mac(acc[i][r], hv, idx < RS ? B[idx] : A[idx - RS]);
        }
      }
  }
  // the lane's RS x RU outputs, position-major, through LDS
  __syncthreads();
  T *Lo = L + threadIdx.x * OS;
#pragma unroll
  for (int r = 0; r < RS; r++)
#pragma unroll
    for (int i = 0; i < RU; i++) Lo[r * RU + i] = acc[i][r];
  __syncthreads();
  const int64_t ob = (int64_t) blockIdx.x * TILE * RU;
  for (int q = threadIdx.x; q < TILE * RU; q += THREADS)
    if (ob + q < nout) y[ob + q] = L[(q / (RS * RU)) * OS + (q % (RS * RU))];
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

// Complex-coefficient flavour of the literal recursion (filtre_rii<cfloat, cfloat>): one lane walks
// the complex stream; tiles staged through LDS like above, the output memory (most recent first) in LDS.
constexpr int RII_KC_MAX = 256;
__global__ __launch_bounds__(256) void rii_recursive_cplx_kernel(float2 *__restrict__ y, const float2 *__restrict__ denom, int Ky,
                                                                 float2 *__restrict__ hist, int64_t n)
{
  __shared__ float2 tile[RII_TILE / 2];
  __shared__ float2 h[RII_KC_MAX], d[RII_KC_MAX];
  const int tid = threadIdx.x;
  for (int k = tid; k < Ky; k += 256) {
    h[k] = hist[k];
    d[k] = denom[k + 1];
  }
  const float2 d0 = denom[0];
  const float nd0 = d0.x * d0.x + d0.y * d0.y;
  int head = 0;   // h is circular: h[(head + k) % Ky] = y[-1-k]
  __syncthreads();
  for (int64_t base = 0; base < n; base += RII_TILE / 2) {
    const int cnt = (int) min((int64_t) (RII_TILE / 2), n - base);
    for (int i = tid; i < cnt; i += 256) tile[i] = y[base + i];
    __syncthreads();
    if (tid == 0) {
      for (int j = 0; j < cnt; j++) {
        float2 somme = tile[j];
        for (int k = 0; k < Ky; k++) {
          int idx = head + k;
          if (idx >= Ky) idx -= Ky;
          const float2 w = h[idx], c = d[k];
          somme.x -= w.x * c.x - w.y * c.y;
          somme.y -= w.x * c.y + w.y * c.x;
        }
        // somme / d0 (limited-range complex division, as -fcx-limited-range compiles it)
        const float2 o = make_float2((somme.x * d0.x + somme.y * d0.y) / nd0, (somme.y * d0.x - somme.x * d0.y) / nd0);
        tile[j] = o;
        if (Ky > 0) {
          head = head == 0 ? Ky - 1 : head - 1;
          h[head] = o;
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < cnt; i += 256) y[base + i] = tile[i];
    __syncthreads();
  }
  if (tid == 0)
    for (int k = 0; k < Ky; k++) {
      int idx = head + k;
      if (idx >= Ky) idx -= Ky;
      hist[k] = h[idx];
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
  float *d_hrev = nullptr;          // decim_direct_kernel: the taps reversed, zero-padded to KPd (inside d_g's allocation)
  int KPd = 0;
  void *d_hist[2] = {nullptr, nullptr};
  DevBuf z, in_stage, out_stage;
};

namespace tsdgpu {
int sos_create_ex(tsdgpu_sos **out, int data_type, const float *coefs_host, int nsec, float gain,
                  const float *rii1_host, int forme, int seeded);
}


// ---- filtre_rii<cfloat, cfloat> (filtre-rt.cc:177-289, :795), block-parallel --------------------------------------------
// 1 / D(z^-1) with a COMPLEX denominator D = d0 prod_s (1 - r_s z^-1) runs as a cascade of first-order complex sections
// y[n] = v[n] + r_s y[n-1] (complex poles need no conjugate partner), each by the scheme of the SOS kernel: a lane owns 16
// consecutive samples of a 1024-sample sub-tile and runs the section from zero state; the lanes' end values go through a
// Kogge-Stone scan of the affine maps y -> a y + e, a = r^16 the same for every lane, so only e travels (6 shuffle steps
// against the host-made powers a^(2^k)); every sample then gets its carry: c <- r c, y += c.  A wave walks the sub-tiles of
// its chunk carrying the sections' states exactly; chunk 0 starts from the stream's state, the others `warm` sub-tiles
// early from zero state, the host having found the warm-up after which the cascade's memory is below 1e-9.
// tab[s][8] = r, a, a^2, a^4, a^8, a^16, a^32, (pad).
constexpr int C1_LANE = 16, C1_SUB = 64 * C1_LANE, C1_PITCH = C1_LANE + 2;     // pitch 18 complex = 144 B: conflict-free b128
constexpr int C1_MAX_SEC = 16;
__device__ __forceinline__ float2 c1_mul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 c1_fma(float2 a, float2 b, float2 c) { return make_float2(fmaf(a.x, b.x, fmaf(-a.y, b.y, c.x)), fmaf(a.x, b.y, fmaf(a.y, b.x, c.y))); }
__global__ __launch_bounds__(64) void rii_c1_kernel(const float2 *__restrict__ u, float2 *__restrict__ y, const float2 *__restrict__ tab, int p,
                                                    float2 gain, const float2 *__restrict__ st_in, float2 *__restrict__ st_out, int64_t n,
                                                    int64_t n_sub, int spc, int warm)
{
  __shared__ __attribute__((aligned(16))) float2 lds[64 * C1_PITCH];
  __shared__ float2 sst[C1_MAX_SEC];                         // the sections' states: y_s at the last sample done
  const int lane = threadIdx.x;
  const int64_t chunk = blockIdx.x, t_first = chunk * spc, t_last = min(t_first + spc, n_sub);
  int64_t t = max((int64_t) 0, t_first - warm);
  if (lane < p) sst[lane] = t == 0 ? st_in[lane] : make_float2(0.f, 0.f);
  auto wsync = []() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  wsync();
  for (; t < t_last; t++) {
    const int64_t base = t * C1_SUB;
    const int m = (int) min((int64_t) C1_SUB, n - base);      // samples of this sub-tile inside the call (the rest reads as zero)
    // ---- load 16 B per lane (two samples), transpose through LDS: lane gets samples [16 lane, 16 lane + 16)
    float2 v[C1_LANE];
#pragma unroll
    for (int i = 0; i < C1_LANE / 2; i++) {
      const int e = 2 * (i * 64 + lane);
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e + 1 < m) q = *reinterpret_cast<const float4 *>(u + base + e);
      else if (e < m) { const float2 a = u[base + e]; q.x = a.x; q.y = a.y; }
      *reinterpret_cast<float4 *>(&lds[(e / C1_LANE) * C1_PITCH + (e % C1_LANE)]) = q;
    }
    wsync();
#pragma unroll
    for (int i = 0; i < C1_LANE / 2; i++) {
      const float4 q = *reinterpret_cast<const float4 *>(&lds[lane * C1_PITCH + 2 * i]);
      v[2 * i] = make_float2(q.x, q.y);
      v[2 * i + 1] = make_float2(q.z, q.w);
    }
    wsync();
    // (lane, index) of the sub-tile's last real sample: where the sections' states are read
    const int lstar = (m - 1) / C1_LANE, istar = (m - 1) % C1_LANE;
    for (int s = 0; s < p; s++) {
      const float2 r = tab[s * 8], yprev = sst[s];
      // zero-state run of the lane
      float2 acc = make_float2(0.f, 0.f);
#pragma unroll
      for (int i = 0; i < C1_LANE; i++) {
        acc = c1_fma(r, acc, v[i]);
        v[i] = acc;
      }
      // true end values of the lanes: lane 0 takes the section's state in, then the scan
      float2 e = acc;
      if (lane == 0) e = c1_fma(tab[s * 8 + 1], yprev, e);
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const float2 ak = tab[s * 8 + 1 + k];
        float2 o;
        o.x = __shfl_up(e.x, 1 << k);
        o.y = __shfl_up(e.y, 1 << k);
        if (lane >= (1 << k)) e = c1_fma(ak, o, e);
      }
      // the value entering the lane, carried through its samples
      float2 c;
      c.x = __shfl_up(e.x, 1);
      c.y = __shfl_up(e.y, 1);
      if (lane == 0) c = yprev;
      float2 fin = make_float2(0.f, 0.f);
#pragma unroll
      for (int i = 0; i < C1_LANE; i++) {
        c = c1_mul(r, c);
        v[i].x += c.x;
        v[i].y += c.y;
        if (i == istar) fin = v[i];
      }
      fin.x = __shfl(fin.x, lstar);
      fin.y = __shfl(fin.y, lstar);
      if (lane == 0) sst[s] = fin;
      wsync();
    }
    if (t >= t_first) {
      // ---- x 1 / d0, transpose back, store 16 B per lane
#pragma unroll
      for (int i = 0; i < C1_LANE / 2; i++) {
        const float2 a = c1_mul(v[2 * i], gain), b = c1_mul(v[2 * i + 1], gain);
        *reinterpret_cast<float4 *>(&lds[lane * C1_PITCH + 2 * i]) = make_float4(a.x, a.y, b.x, b.y);
      }
      wsync();
#pragma unroll
      for (int i = 0; i < C1_LANE / 2; i++) {
        const int e2 = 2 * (i * 64 + lane);
        const float4 q = *reinterpret_cast<const float4 *>(&lds[(e2 / C1_LANE) * C1_PITCH + (e2 % C1_LANE)]);
        if (e2 + 1 < m) *reinterpret_cast<float4 *>(y + base + e2) = q;
        else if (e2 < m) y[base + e2] = make_float2(q.x, q.y);
      }
      wsync();
    }
  }
  if (t_last == n_sub && t_last > t_first && lane < p) st_out[lane] = sst[lane];     // the wave of the last sub-tile publishes the stream state
}

struct tsdgpu_rii {
  int data_type = 0, coef_type = 0, Ky = 0, Kx = 0;
  // path 0: the whole H(z) as zero-seeded FormeDirecte1 sections on the block-parallel SOS kernel
  //         (numerator of <= 3 taps folded into the first section)
  // path 1: numerator on the FIR kernel, then the all-pole part 1/D as sections on the SOS kernel
  // path 2: numerator on the FIR kernel, then the literal sequential recursion (reference operation order)
  // path 3: complex coefficients: numerator on the FIR kernel, then 1 / D as first-order complex sections (rii_c1_kernel)
  int path = 2;
  int c1_p = 0, c1_cur = 0;                        // path 3: sections, current state buffer
  int64_t c1_W = 0;                                // ... warm-up in samples
  float2 c1_gain = {1.f, 0.f};                     // ... 1 / d0
  float2 *d_c1 = nullptr;                          // ... tab [p][8] | state A [16] | state B [16]
  DevBuf z;                                        // ... the non-recursive part's output
  tsdgpu_sos *sos = nullptr;
  tsdgpu_fir *fir = nullptr;
  float *d_denom = nullptr, *d_hist = nullptr;     // literal path: denominator (float or float2) and output memory
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
  if (dev_switch("POLY_COMPOSED") || to < 256 || (size_t) NPH * W > 4096 || W < 1) return TSDGPU_OK;
  p->TO = (int) (to / 256 * 256);
  p->NPH = NPH;
  p->W = W;
  p->HW = std::max(W - 1, 1);
  // ONE allocation -- the taps, then the two (zero) histories, 16-byte aligned -- and ONE upload of its host image (three
  // allocations, a copy, two memsets and a synchronisation before: a third of a one-shot rééchan(x, 4))
  const size_t hb = ((size_t) p->HW * dtype_size(p->data_type) + 15) / 16 * 16, gb = (g.size() * sizeof(float) + 15) / 16 * 16;
  // decimators of rate 2 / 4 / 8 up to 64 taps: the reversed, padded taps of decim_direct_kernel behind the histories
  const int RSd = p->data_type == TSDGPU_F32 ? 16 : 8;
  // ... and upsamplers of rate 2 / 4 with branches of up to 32 taps (ups_direct_kernel): one such row per branch
  const bool updir = stride == 1 && (NPH == 2 || NPH == 4) && W <= 32 && dev_switch("POLY_NO_DIRECT") == nullptr;
  const bool direct = (NPH == 1 && (stride == 2 || stride == 4 || stride == 8) && W <= 64 && dev_switch("POLY_NO_DIRECT") == nullptr) || updir;
  p->KPd = direct ? (int) (cdiv(W, 2 * RSd) * 2 * RSd) : 0;
  const size_t rb = (size_t) p->KPd * NPH * sizeof(float);
  std::vector<char> image(gb + 2 * hb + rb, 0);
  std::memcpy(image.data(), g.data(), g.size() * sizeof(float));
  if (direct) {
    float *hr = reinterpret_cast<float *>(image.data() + gb + 2 * hb);
    for (int i = 0; i < NPH; i++)
      for (int k = 0; k < W; k++) hr[(size_t) i * p->KPd + p->KPd - 1 - k] = g[(size_t) i * W + k];   // hrev[j] = g[KP - 1 - j]: g[0] meets the newest sample
  }
  if (hipMalloc((void **) &p->d_g, image.size()) != hipSuccess)
    return set_err(TSDGPU_ERR_HIP, "polyfir_create: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
  p->d_hist[0] = (char *) p->d_g + gb;
  p->d_hist[1] = (char *) p->d_g + gb + hb;
  p->d_hrev = direct ? reinterpret_cast<float *>((char *) p->d_g + gb + 2 * hb) : nullptr;
  if (hipMemcpy(p->d_g, image.data(), image.size(), hipMemcpyHostToDevice) != hipSuccess)
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
  if (nout > 0 && p->d_hrev && p->NPH > 1 && stride == 1) {
    constexpr int RS = 64 / (int) sizeof(T);
    const int64_t tiles = cdiv(n, (int64_t) 256 * RS);
    const int VECs = 16 / (int) sizeof(T);
    const size_t lds = std::max(((size_t) (256 * RS + p->KPd) / RS + 3) * 80, (size_t) 256 * (RS * p->NPH + VECs) * sizeof(T));
    if (tiles <= 0x7fffffff) {
      if (p->NPH == 2) {
        (void) hipFuncSetAttribute((const void *) ups_direct_kernel<T, RS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((ups_direct_kernel<T, RS, 2>), dim3((unsigned) tiles), dim3(256), lds, st, (const T *) dx, (const T *) p->d_hist[p->cur], (T *) dy,
                           p->d_hrev, p->KPd, p->HW, n, nout);
      } else {
        (void) hipFuncSetAttribute((const void *) ups_direct_kernel<T, RS, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((ups_direct_kernel<T, RS, 4>), dim3((unsigned) tiles), dim3(256), lds, st, (const T *) dx, (const T *) p->d_hist[p->cur], (T *) dy,
                           p->d_hrev, p->KPd, p->HW, n, nout);
      }
      TSD_HIP(hipGetLastError());
      nout = 0;
    }
  }
  if (nout > 0 && p->d_hrev && p->NPH == 1) {
    constexpr int RS = 64 / (int) sizeof(T);
    const int64_t tiles = cdiv(nout, (int64_t) 256 * RS / stride);
    const size_t lds = ((size_t) (256 * RS + p->KPd) / RS + 3) * 80;
    if (tiles <= 0x7fffffff) {
#define DD_LAUNCH(D) hipLaunchKernelGGL((decim_direct_kernel<T, RS, D>), dim3((unsigned) tiles), dim3(256), lds, st, (const T *) dx,     \
                                        (const T *) p->d_hist[p->cur], (T *) dy, p->d_hrev, p->KPd, (int) start, p->HW, n, nout)
      if (stride == 2) DD_LAUNCH(2); else if (stride == 4) DD_LAUNCH(4); else DD_LAUNCH(8);
#undef DD_LAUNCH
      TSD_HIP(hipGetLastError());
      nout = 0;                                  // (served; the history update below still runs)
    }
  }
  static const bool sans_rangs = dev_switch("POLY_NO_ROWS") != nullptr;
  if (nout > 0 && !sans_rangs && p->NPH == 1 && stride >= 2 && stride <= PF_ROWS_MAXR && p->W >= 32) {
    // decimators of small rate and at least 32 taps: polyphase rows (decim_rows_kernel).  (Shorter filters are bound by the
    // staging, which the phase-major scatter makes dearer: 15 taps at R = 2, 2^26 samples: 0.22 against 0.18 ms.)
    const int Lr = ((p->W + stride - 1) / stride + 1 + 3) / 4 * 4, pitch = p->TO + Lr + 4;
    const size_t lds = (size_t) stride * Lr * sizeof(float) + 2 * PF_ROWS_MAXR * sizeof(int) + (size_t) stride * pitch * sizeof(T);
    if (lds <= 150 * 1024) {
      (void) hipFuncSetAttribute((const void *) decim_rows_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(decim_rows_kernel<T>, dim3((unsigned) cdiv(nout, p->TO)), dim3(256), lds, st, (const T *) dx, (const T *) p->d_hist[p->cur],
                         (T *) dy, p->d_g, stride, p->W, start, p->HW, n, nout, p->TO, Lr, pitch);
      TSD_HIP(hipGetLastError());
      nout = 0;                                  // (served; the history update below still runs)
    }
  }
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
    // (the composition -- a full-rate FIR, then the pick -- only where the fused kernel does not serve: building its
    // handle costs a one-shot rééchan() 30 us per stage)
    rc = fused_setup(p, h, 1, ntaps, R);
    if (!rc && !p->fused) {
      tsdgpu_fir *f = nullptr;
      rc = tsdgpu_fir_create(&f, data_type, TSDGPU_F32, h.data(), ntaps, TSDGPU_FIR_AUTO);
      if (!rc) p->fir.push_back(f);
    }
  } else if (kind == TSDGPU_POLY_UPS) {
    // coefs = c * R, zero-padded to a multiple of R (polyphase.cc:259-270); phase i correlates
    // the K/R-sample window with coefs[(R-1-i) + j*R]
    std::vector<float> c((size_t) ntaps);
    for (int i = 0; i < ntaps; i++) c[i] = taps_host[i] * (float) R;
    while (c.size() % (size_t) R) c.push_back(0.f);
    const int W = (int) c.size() / R;
    std::vector<float> gall;
    for (int i = 0; i < R; i++)
      for (int k = 0; k < W; k++) gall.push_back(c[(size_t) (R - 1 - i) + (size_t) (W - 1 - k) * R]);
    rc = fused_setup(p, gall, R, W, 1);
    for (int i = 0; i < R && !rc && !p->fused; i++) {      // the composition (R branch FIRs + interleave) only without the fused kernel
      tsdgpu_fir *f = nullptr;
      rc = tsdgpu_fir_create(&f, data_type, TSDGPU_F32, gall.data() + (size_t) i * W, W, TSDGPU_FIR_AUTO);
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
  // large HOST vectors: chunked H2D / kernel / D2H pipeline (chunks are whole multiples of the rate, so a chunk of a
  // decimating stage never leaves a partial group to the next one beyond what the stage's own counter carries)
  if ((size_t) n * sz >= PIPE_MIN_BYTES && host_pipe_enabled() && y != nullptr && !is_device_ptr(x) && !is_device_ptr(y) &&
      !host_ranges_overlap(x, (size_t) n * sz, y, (size_t) nout * sz)) {
    const bool up = p->kind == TSDGPU_POLY_UPS;
    const int R = std::max(p->R, 1);
    return pipelined_host_step_var(
        x, n, sz, y, sz, n_out, R, st, [up, R](int64_t c) { return up ? c * R : c / R + 2; },
        [p](const void *cx, void *cy, int64_t cnt, int64_t cap, int64_t *got, hipStream_t q) {
          return tsdgpu_polyfir_step(p, cx, cnt, cy, cap, got, q);
        });
  }
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
  if (p->fused) {
    TSD_HIP(hipMemset(p->d_hist[p->cur], 0, (size_t) p->HW * dtype_size(p->data_type)));
    TSD_HIP(hipStreamSynchronize(nullptr));
  }
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
  if (p->d_g) (void) hipFree(p->d_g);            // (the histories live in the same allocation)
  p->z.release();
  p->in_stage.release();
  p->out_stage.release();
  delete p;
  return TSDGPU_OK;
}

// ---- FiltreRII ------------------------------------------------------------------------------
// Block-parallel plan: the denominator D(z^-1) = d0 prod (1 + a1 z^-1 + a2 z^-2) is factored on the host
// (Aberth iteration in double, conjugate / real roots paired) and the recursion runs as a cascade of
// zero-seeded FormeDirecte1 sections on the SOS kernel -- the same transfer function from the same
// zero initial state, computed in one HBM pass at a few hundred Gsamples/s instead of one sample per
// dependent-FMA latency.  A cascade rounds differently from the direct form, so every filter is
// checked at creation: the float direct-form recursion (the reference's own operation order) and the
// float cascade are both run on the host over 8192 noise samples, and the cascade is used only when
// they agree to 4e-6 of the peak; otherwise (clustered roots, unstable or ill-conditioned direct forms)
// the literal sequential kernels above serve, bit-faithful to the reference's order.
}  // extern "C" (host helpers below have C++ linkage)

namespace {

typedef std::complex<double> cd;

// roots of c[0] z^p + c[1] z^(p-1) + ... + c[p] (c[0] != 0; real or complex coefficients): Aberth-Ehrlich, then a coefficient check
template <typename TC> bool poly_roots(const std::vector<TC> &c, std::vector<cd> &r)
{
  const int p = (int) c.size() - 1;
  r.assign((size_t) p, cd(0, 0));
  if (p <= 0) return true;
  double rad = 0;
  for (int k = 1; k <= p; k++) rad = std::max(rad, std::pow(std::abs(c[k] / c[0]), 1.0 / k));
  rad = std::max(2 * rad, 1e-3);
  for (int k = 0; k < p; k++) r[k] = std::polar(rad * (0.5 + 0.5 * (k + 1) / p), 2 * M_PI * k / p + 0.4);
  auto eval = [&](cd z, cd &dv) {
    cd v = c[0];
    dv = 0;
    for (int k = 1; k <= p; k++) {
      dv = dv * z + v;
      v = v * z + c[k];
    }
    return v;
  };
  for (int it = 0; it < 400; it++) {
    double delta = 0;
    for (int i = 0; i < p; i++) {
      cd dv;
      const cd v = eval(r[i], dv);
      if (v == cd(0, 0)) continue;
      const cd nw = v / dv;
      cd sum = 0;
      for (int j = 0; j < p; j++)
        if (j != i) sum += 1.0 / (r[i] - r[j]);
      const cd w = nw / (1.0 - nw * sum);
      r[i] -= w;
      delta = std::max(delta, std::abs(w) / std::max(1e-30, std::abs(r[i])));
    }
    if (delta < 1e-15) break;
  }
  // expand prod (z - r_i) and compare with the monic coefficients
  std::vector<cd> e(1, cd(1, 0));
  for (int i = 0; i < p; i++) {
    e.push_back(0);
    for (int k = (int) e.size() - 1; k > 0; k--) e[k] -= r[i] * e[k - 1];
  }
  double err = 0, nrm = 0;
  for (int k = 0; k <= p; k++) {
    err = std::max(err, std::abs(e[k] - cd(c[k] / c[0])));
    nrm = std::max(nrm, std::abs(c[k] / c[0]));
  }
  return std::isfinite(err) && err <= 1e-9 * nrm;
}

// all-pole sections (a1, a2) of D(z^-1) / d0 from its coefficient list; false when the factoring is unreliable
bool factor_denominator(const float *den, int Kd, std::vector<std::array<float, 2>> &sec)
{
  int p = Kd - 1;
  while (p > 0 && den[p] == 0.f) p--;                      // trailing zeros: poles at the origin
  sec.clear();
  if (p == 0) return true;
  std::vector<double> c((size_t) p + 1);
  for (int k = 0; k <= p; k++) c[k] = den[k];
  std::vector<cd> r;
  if (!poly_roots(c, r)) return false;
  std::vector<cd> cplx;
  std::vector<double> reel;
  for (const cd &z : r) {
    if (!(std::abs(z) < 1.0)) return false;                // unstable or marginal: leave it to the literal kernel
    if (std::fabs(z.imag()) <= 1e-12 * std::max(1.0, std::abs(z))) reel.push_back(z.real());
    else cplx.push_back(z);
  }
  // conjugate pairs: every root of positive imaginary part takes the closest remaining conjugate
  std::vector<bool> pris(cplx.size(), false);
  for (size_t i = 0; i < cplx.size(); i++) {
    if (pris[i] || cplx[i].imag() < 0) continue;
    pris[i] = true;
    int best = -1;
    double bd = 1e300;
    for (size_t j = 0; j < cplx.size(); j++)
      if (!pris[j] && cplx[j].imag() < 0 && std::abs(cplx[j] - std::conj(cplx[i])) < bd) {
        bd = std::abs(cplx[j] - std::conj(cplx[i]));
        best = (int) j;
      }
    if (best < 0 || bd > 1e-7 * std::max(1.0, std::abs(cplx[i]))) return false;
    pris[(size_t) best] = true;
    const cd m = 0.5 * (cplx[i] + std::conj(cplx[(size_t) best]));
    sec.push_back({(float) (-2 * m.real()), (float) std::norm(m)});
  }
  for (size_t i = 0; i < cplx.size(); i++)
    if (!pris[i]) return false;
  std::sort(reel.begin(), reel.end(), [](double a, double b) { return std::fabs(a) > std::fabs(b); });
  for (size_t i = 0; i + 1 < reel.size(); i += 2) sec.push_back({(float) (-(reel[i] + reel[i + 1])), (float) (reel[i] * reel[i + 1])});
  if (reel.size() & 1) sec.push_back({(float) (-reel.back()), 0.f});
  return true;
}

// create-time check of the cascade against the reference's direct form, both in float, zero memory
bool cascade_matches_direct_form(const float *num, int Kx, const float *den, int Kd, const std::vector<float> &coefs5, float gain,
                                 bool num_in_cascade)
{
  const int N = 8192, Ky = Kd - 1, ns = (int) coefs5.size() / 5;
  std::vector<float> x((size_t) N), u((size_t) N), yd((size_t) N), yc((size_t) N);
  uint32_t lcg = 12345u;
  for (int i = 0; i < N; i++) {
    lcg = lcg * 1664525u + 1013904223u;
    x[i] = (float) ((int32_t) lcg) * (1.0f / 2147483648.0f);
  }
  for (int j = 0; j < N; j++) {                            // (1) non-recursive part, oldest sample first
    float s = 0;
    for (int i = Kx - 1; i >= 0; i--)
      if (j - i >= 0) s += x[j - i] * num[i];
    u[j] = s;
  }
  for (int j = 0; j < N; j++) {                            // (2) recursive part (filtre-rt.cc:251-279)
    float s = u[j];
    for (int k = 1; k <= Ky; k++)
      if (j - k >= 0) s -= yd[j - k] * den[k];
    yd[j] = s / den[0];
  }
  const std::vector<float> &in = num_in_cascade ? x : u;
  std::vector<float> st((size_t) ns * 4, 0.f);             // x1, x2, y1, y2 per section
  for (int j = 0; j < N; j++) {
    float v = in[j];
    for (int q = 0; q < ns; q++) {
      const float *c = &coefs5[(size_t) q * 5];
      float *m = &st[(size_t) q * 4];
      const float o = c[0] * v + c[1] * m[0] + c[2] * m[1] - c[3] * m[2] - c[4] * m[3];
      m[1] = m[0]; m[0] = v; m[3] = m[2]; m[2] = o;
      v = o;
    }
    yc[j] = v * gain;
  }
  float peak = 0, err = 0;
  for (int j = 0; j < N; j++) {
    if (!std::isfinite(yd[j]) || !std::isfinite(yc[j])) return false;
    peak = std::max(peak, std::fabs(yd[j]));
    err = std::max(err, std::fabs(yd[j] - yc[j]));
  }
  return peak > 0 && err <= 4e-6f * peak;   // bar: 1e-5 of the peak on any stream; 2.5x margin for the extremes of long streams
}


// ---- complex denominators: first-order sections (rii_c1_kernel) -------------------------------------------------------
// poles r_s of D(z^-1) = d0 prod (1 - r_s z^-1); false when the factoring is unreliable or a pole is not inside the circle
bool factor_denominator_c(const float *den2, int Kd, std::vector<cd> &poles)
{
  int p = Kd - 1;
  while (p > 0 && den2[2 * p] == 0.f && den2[2 * p + 1] == 0.f) p--;
  poles.clear();
  if (p == 0) return true;
  std::vector<cd> c((size_t) p + 1);
  for (int k = 0; k <= p; k++) c[k] = cd(den2[2 * k], den2[2 * k + 1]);
  if (!poly_roots(c, poles)) return false;
  for (const cd &z : poles)
    if (!(std::abs(z) < 1.0)) return false;
  // the largest poles first: the sections' order does not change the transfer function, only the rounding
  std::sort(poles.begin(), poles.end(), [](const cd &a, const cd &b) { return std::abs(a) > std::abs(b); });
  return true;
}
// samples after which the cascade's memory is below 1e-9: the zero-input response from every unit state, in double
int64_t cascade_c_warmup(const std::vector<cd> &poles, int64_t limit)
{
  const int p = (int) poles.size();
  int64_t W = 0;
  for (int j = 0; j < p; j++) {
    std::vector<cd> st((size_t) p, cd(0, 0));
    st[(size_t) j] = 1.0;
    int64_t n = 0, calme = 0;
    while (n < limit && calme < 64) {
      cd v = 0;                                            // zero input
      double mx = 0;
      for (int s = 0; s < p; s++) {
        v = v + poles[(size_t) s] * st[(size_t) s];
        st[(size_t) s] = v;
        mx = std::max(mx, std::abs(v));
      }
      n++;
      calme = mx < 1e-9 ? calme + 1 : 0;
    }
    if (n >= limit) return -1;
    W = std::max(W, n);
  }
  return W;
}
// create-time check of the float cascade against the reference's direct form (filtre-rt.cc:251-279, complex arithmetic
// as -fcx-limited-range compiles it), zero memory, 8192 noise samples
bool cascade_c_matches_direct_form(const float *num2, int Kx, const float *den2, int Kd, const std::vector<cd> &poles)
{
  typedef std::complex<float> cf;
  const int N = 8192, Ky = Kd - 1, p = (int) poles.size();
  std::vector<cf> x((size_t) N), u((size_t) N), yd((size_t) N), yc((size_t) N);
  uint32_t lcg = 12345u;
  auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float) ((int32_t) lcg) * (1.0f / 2147483648.0f); };
  for (int i = 0; i < N; i++) x[i] = cf(rnd(), rnd());
  auto mul = [](cf a, cf b) { return cf(a.real() * b.real() - a.imag() * b.imag(), a.real() * b.imag() + a.imag() * b.real()); };
  const cf d0(den2[0], den2[1]);
  const float nd0 = d0.real() * d0.real() + d0.imag() * d0.imag();
  for (int j = 0; j < N; j++) {
    cf sacc(0, 0);
    for (int i = Kx - 1; i >= 0; i--)
      if (j - i >= 0) sacc += mul(x[j - i], cf(num2[2 * i], num2[2 * i + 1]));
    u[j] = sacc;
  }
  for (int j = 0; j < N; j++) {
    cf sacc = u[j];
    for (int k = 1; k <= Ky; k++)
      if (j - k >= 0) sacc -= mul(yd[j - k], cf(den2[2 * k], den2[2 * k + 1]));
    yd[j] = cf((sacc.real() * d0.real() + sacc.imag() * d0.imag()) / nd0, (sacc.imag() * d0.real() - sacc.real() * d0.imag()) / nd0);
  }
  const cd g = 1.0 / cd(den2[0], den2[1]);
  const cf gf((float) g.real(), (float) g.imag());
  std::vector<cf> st((size_t) p, cf(0, 0)), rf((size_t) p);
  for (int s = 0; s < p; s++) rf[(size_t) s] = cf((float) poles[(size_t) s].real(), (float) poles[(size_t) s].imag());
  for (int j = 0; j < N; j++) {
    cf v = u[j];
    for (int s = 0; s < p; s++) {
      v = v + mul(rf[(size_t) s], st[(size_t) s]);
      st[(size_t) s] = v;
    }
    yc[j] = mul(v, gf);
  }
  float peak = 0, err = 0;
  for (int j = 0; j < N; j++) {
    if (!std::isfinite(yd[j].real()) || !std::isfinite(yc[j].real()) || !std::isfinite(yd[j].imag()) || !std::isfinite(yc[j].imag())) return false;
    peak = std::max(peak, std::abs(yd[j]));
    err = std::max(err, std::abs(yd[j] - yc[j]));
  }
  return peak > 0 && err <= 4e-6f * peak;
}

}  // namespace

extern "C" {

int tsdgpu_rii_create2(tsdgpu_rii **out, int data_type, int coef_type, const void *numer_host, int Kx, const void *denom_host, int Kd)
{
  TSD_CHECK(out != nullptr, "rii_create: out is NULL");
  *out = nullptr;
  TSD_CHECK(data_type == TSDGPU_F32 || data_type == TSDGPU_C64, "rii_create: bad data_type %d", data_type);
  TSD_CHECK(coef_type == TSDGPU_F32 || coef_type == TSDGPU_C64, "rii_create: bad coef_type %d", coef_type);
  TSD_CHECK(coef_type == TSDGPU_F32 || data_type == TSDGPU_C64, "rii_create: complex coefficients need complex data");
  TSD_CHECK(numer_host && Kx > 0 && denom_host && Kd > 0, "rii_create: numerator and denominator need >= 1 coefficient");
  // complex coefficients whose imaginary parts all vanish are real coefficients
  std::vector<float> nr((size_t) Kx), dr((size_t) Kd);
  bool reel = true;
  if (coef_type == TSDGPU_C64) {
    const float *a = (const float *) numer_host, *b = (const float *) denom_host;
    for (int i = 0; i < Kx; i++) { nr[i] = a[2 * i]; reel = reel && a[2 * i + 1] == 0.f; }
    for (int i = 0; i < Kd; i++) { dr[i] = b[2 * i]; reel = reel && b[2 * i + 1] == 0.f; }
    TSD_CHECK(b[0] != 0.f || b[1] != 0.f, "rii_create: denom[0] must be non zero");
    TSD_CHECK(reel || Kd - 1 <= RII_KC_MAX, "rii_create: complex denominators of more than %d coefficients are not built", RII_KC_MAX + 1);
  } else {
    std::copy((const float *) numer_host, (const float *) numer_host + Kx, nr.begin());
    std::copy((const float *) denom_host, (const float *) denom_host + Kd, dr.begin());
  }
  if (reel) TSD_CHECK(dr[0] != 0.f, "rii_create: denom[0] must be non zero");
  tsdgpu_rii *r = new tsdgpu_rii();
  r->data_type = data_type;
  r->coef_type = reel ? TSDGPU_F32 : TSDGPU_C64;
  r->Ky = Kd - 1;
  r->Kx = Kx;
  int rc = TSDGPU_OK;
  static const bool litteral = dev_switch("RII_LITERAL") != nullptr;
  if (reel && !litteral) {
    std::vector<std::array<float, 2>> poles;
    if (factor_denominator(dr.data(), Kd, poles) && (int) poles.size() <= 32) {
      const bool plie = Kx <= 3;                            // numerator folded into the first section
      const float d0 = dr[0];
      std::vector<float> c5;
      if (poles.empty()) poles.push_back({0.f, 0.f});
      for (size_t q = 0; q < poles.size(); q++) {
        if (q == 0 && plie) { c5.push_back(nr[0] / d0); c5.push_back(Kx > 1 ? nr[1] / d0 : 0.f); c5.push_back(Kx > 2 ? nr[2] / d0 : 0.f); }
        else { c5.push_back(1.f); c5.push_back(0.f); c5.push_back(0.f); }
        c5.push_back(poles[q][0]);
        c5.push_back(poles[q][1]);
      }
      const float gain = plie ? 1.0f : 1.0f / d0;
      if (cascade_matches_direct_form(nr.data(), Kx, dr.data(), Kd, c5, gain, plie)) {
        rc = tsdgpu::sos_create_ex(&r->sos, data_type, c5.data(), (int) poles.size(), gain, nullptr, 1, 0);
        if (!rc && !plie) rc = tsdgpu_fir_create(&r->fir, data_type, TSDGPU_F32, nr.data(), Kx, TSDGPU_FIR_AUTO);
        if (rc) { tsdgpu_rii_destroy(r); return rc; }
        r->path = plie ? 0 : 1;
        *out = r;
        return TSDGPU_OK;
      }
    }
  }
  if (!reel && !litteral && data_type == TSDGPU_C64) {
    // complex coefficients: first-order complex sections when every pole lies inside the circle, the memory dies within
    // 2^20 samples and the float cascade reproduces the direct form on the create-time check
    std::vector<cd> poles;
    const float *den2 = (const float *) denom_host, *num2 = (const float *) numer_host;
    if (factor_denominator_c(den2, Kd, poles) && !poles.empty() && (int) poles.size() <= C1_MAX_SEC) {
      const int64_t W = cascade_c_warmup(poles, (int64_t) 1 << 20);
      if (W >= 0 && cascade_c_matches_direct_form(num2, Kx, den2, Kd, poles)) {
        const int p = (int) poles.size();
        std::vector<float2> img((size_t) p * 8 + 2 * C1_MAX_SEC, make_float2(0.f, 0.f));
        for (int sct = 0; sct < p; sct++) {
          const cd rq = poles[(size_t) sct];
          cd a = 1.0;
          for (int i = 0; i < C1_LANE; i++) a *= rq;                  // a = r^16
          img[(size_t) sct * 8] = make_float2((float) rq.real(), (float) rq.imag());
          for (int k = 0; k < 6; k++) {
            img[(size_t) sct * 8 + 1 + k] = make_float2((float) a.real(), (float) a.imag());
            a *= a;
          }
        }
        const cd g = 1.0 / cd(den2[0], den2[1]);
        r->c1_gain = make_float2((float) g.real(), (float) g.imag());
        r->c1_p = p;
        r->c1_W = W;
        rc = tsdgpu_fir_create(&r->fir, data_type, TSDGPU_C64, numer_host, Kx, TSDGPU_FIR_AUTO);
        if (!rc && (hipMalloc((void **) &r->d_c1, img.size() * sizeof(float2)) != hipSuccess ||
                    hipMemcpy(r->d_c1, img.data(), img.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess))
          rc = set_err(TSDGPU_ERR_HIP, "rii_create: upload failed: %s", hipGetErrorString(hipGetLastError()));
        if (rc) { tsdgpu_rii_destroy(r); return rc; }
        r->path = 3;
        *out = r;
        return TSDGPU_OK;
      }
    }
  }
  // literal path
  r->path = 2;
  const size_t cw = reel ? sizeof(float) : sizeof(float2);
  rc = tsdgpu_fir_create(&r->fir, data_type, reel ? TSDGPU_F32 : TSDGPU_C64, reel ? (const void *) nr.data() : numer_host, Kx, TSDGPU_FIR_AUTO);
  const size_t hb = (size_t) std::max(r->Ky, 1) * 2 * sizeof(float);
  if (!rc && (hipMalloc((void **) &r->d_denom, (size_t) Kd * cw) != hipSuccess || hipMalloc((void **) &r->d_hist, hb) != hipSuccess))
    rc = set_err(TSDGPU_ERR_HIP, "rii_create: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
  if (!rc && (hipMemcpy(r->d_denom, reel ? (const void *) dr.data() : denom_host, (size_t) Kd * cw, hipMemcpyHostToDevice) != hipSuccess ||
              hipMemset(r->d_hist, 0, hb) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess))
    rc = set_err(TSDGPU_ERR_HIP, "rii_create: upload failed: %s", hipGetErrorString(hipGetLastError()));
  if (rc) {
    tsdgpu_rii_destroy(r);
    return rc;
  }
  *out = r;
  return TSDGPU_OK;
}

int tsdgpu_rii_create(tsdgpu_rii **out, int data_type, const float *numer_host, int Kx, const float *denom_host, int Kd)
{
  return tsdgpu_rii_create2(out, data_type, TSDGPU_F32, numer_host, Kx, denom_host, Kd);
}

int tsdgpu_rii_path(const tsdgpu_rii *r) { return r ? r->path : -1; }

int tsdgpu_rii_step(tsdgpu_rii *r, const void *x, void *y, int64_t n, void *stream)
{
  TSD_CHECK(r != nullptr, "rii_step: NULL handle");
  TSD_CHECK(n >= 0, "rii_step: negative length");
  if (n == 0) return TSDGPU_OK;
  TSD_CHECK(x != nullptr && y != nullptr, "rii_step: NULL buffer");
  if (r->path == 0) return tsdgpu_sos_step(r->sos, x, y, n, stream);
  hipStream_t st = (hipStream_t) stream;
  const size_t bytes = (size_t) n * dtype_size(r->data_type);
  if (bytes >= PIPE_MIN_BYTES && host_pipe_enabled() && !is_device_ptr(x) && !is_device_ptr(y))
    return pipelined_host_step(x, y, n, dtype_size(r->data_type), st,
                               [r](const void *cx, void *cy, int64_t cnt, hipStream_t q) { return tsdgpu_rii_step(r, cx, cy, cnt, q); });
  const void *dx = nullptr;
  void *dy = nullptr;
  bool staged = false;
  int rc = stage_in(x, bytes, r->in_stage, st, &dx);
  if (rc) return rc;
  rc = stage_out(y, bytes, r->out_stage, &dy, &staged);
  if (rc) return rc;
  if (r->path == 3) {
    // (1) the non-recursive part into an aligned scratch vector (the chunks' warm-ups re-read their predecessors' inputs: never
    // in place), (2) the first-order sections from there
    rc = r->z.reserve(bytes + ((uintptr_t) dy & 15 ? bytes : 0));
    if (rc) return rc;
    float2 *zu = r->z.as<float2>(), *zy = ((uintptr_t) dy & 15) ? zu + n : (float2 *) dy;
    rc = tsdgpu_fir_step(r->fir, dx, zu, n, stream);
    if (rc) return rc;
    const int64_t n_sub = cdiv(n, C1_SUB);
    const int warm = (int) cdiv(r->c1_W, C1_SUB);
    // chunks: enough of them to fill the chip, the warm-up never more than a quarter of a chunk's work
    int64_t spc = std::max<int64_t>(std::max<int64_t>(1, 4 * (int64_t) warm), cdiv(n_sub, 8192));
    const int64_t nchunks = cdiv(n_sub, spc);
    float2 *st0 = r->d_c1 + (size_t) r->c1_p * 8 + (size_t) r->c1_cur * C1_MAX_SEC, *st1 = r->d_c1 + (size_t) r->c1_p * 8 + (size_t) (r->c1_cur ^ 1) * C1_MAX_SEC;
    hipLaunchKernelGGL(rii_c1_kernel, dim3((unsigned) nchunks), dim3(64), 0, st, (const float2 *) zu, zy, (const float2 *) r->d_c1, r->c1_p, r->c1_gain,
                       (const float2 *) st0, st1, n, n_sub, (int) spc, warm);
    TSD_HIP(hipGetLastError());
    r->c1_cur ^= 1;
    if (zy != (float2 *) dy) TSD_HIP(hipMemcpyAsync(dy, zy, bytes, hipMemcpyDeviceToDevice, st));
    return finish_out(y, bytes, dy, staged, st);
  }
  rc = tsdgpu_fir_step(r->fir, dx, dy, n, stream);                      // (1) non-recursive part
  if (rc) return rc;
  if (r->path == 1) {
    rc = tsdgpu_sos_step(r->sos, dy, dy, n, stream);                    // (2) all-pole cascade, in place
    if (rc) return rc;
    return finish_out(y, bytes, dy, staged, st);
  }
  if (r->coef_type == TSDGPU_C64) {
    hipLaunchKernelGGL(rii_recursive_cplx_kernel, dim3(1), dim3(256), 0, st, (float2 *) dy, (const float2 *) r->d_denom, r->Ky,
                       (float2 *) r->d_hist, n);
    TSD_HIP(hipGetLastError());
    return finish_out(y, bytes, dy, staged, st);
  }
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
  if (r->d_c1) (void) hipFree(r->d_c1);
  r->z.release();
  r->in_stage.release();
  r->out_stage.release();
  delete r;
  return TSDGPU_OK;
}

}  // extern "C"
