// fft.hip -- complex FFT plans for gfx950 behind FFTPlan / TFRPlanDefaut
// (libtsd core/include/tsd/fourier.hpp:19-35; core/src/fourier/fourier.cc:61-121,360-486).
//
// Contract kept from the reference: unitary scaling 1/sqrt(n) in BOTH directions, natural
// order output, any n >= 1, plan reconfigures itself per size (here: one plan per size).
//   n = 2^p            -> Stockham-free design: in-LDS decimation-in-frequency passes with the
//                         bit-reversal folded into the (coalesced) store; n > 4096 goes through
//                         a four-step split n = N1*N2 (column FFTs staged through LDS in
//                         16-column tiles so every global access is a full 128-B line)
//   n even, not 2^p    -> even/odd split recursion (fourier.cc:438-463)
//   n odd              -> Bluestein chirp-z on n2 = next pow2 >= 2n-1 (fourier.cc:237-255,391-400)
// Twiddles are produced on the host in double precision and rounded once to float.
#include "common.hpp"
#include "fft1024_wave.hpp"
#include "stockham16.hpp"
#include <cmath>
#include <memory>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

namespace tsdgpu {

typedef float2 cpx;
__device__ __forceinline__ cpx cmk(float a, float b) { return make_float2(a, b); }
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return cmk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return cmk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cpx cmul(cpx a, cpx b) { return cmk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cpx cconj(cpx a) { return cmk(a.x, -a.y); }
__device__ __forceinline__ cpx cscale(cpx a, float s) { return cmk(a.x * s, a.y * s); }

constexpr int FFT_THREADS = 256;
constexpr int LDS_MAX_N = 4096;      // largest transform done in one workgroup's LDS (radix-2 fallback kernel)
constexpr int S16_MAX_N = 16384;     // largest transform of the radix-16 Stockham kernel (139 KiB of LDS)
// threads per transform from which fft_s16_kernel loads / stores straight from registers (measured:
// even 32-B runs per transform beat the LDS staging: n = 64: 0.193 vs 0.236 ms per 2^26 points;
// n = 32 and 16 are better staged)
#ifndef S16_DIRECT_TPT
#define S16_DIRECT_TPT 4
#endif
constexpr int COL_TILE = 16;         // columns per tile: 16 * 8 B = one 128-B line per row

__device__ __forceinline__ unsigned bitrev(unsigned i, int logn) { return logn == 0 ? 0u : (__brev(i) >> (32 - logn)); }

// In-place radix-2 decimation in frequency on `cols` independent length-n sequences held in
// LDS at s[c * pitch + i].  tw[k] = exp(-2 pi i k / n), k < n/2.  Result is bit-reversed.
__device__ __forceinline__ void dif_passes(cpx *s, int pitch, int cols, int n, int logn,
                                           const cpx *__restrict__ tw, bool inverse)
{
  const int half_total = (n >> 1) * cols;
  for (int st = 0; st < logn; st++) {
    const int half = n >> (st + 1);           // butterfly span
    const int tstride = 1 << st;              // twiddle index stride: W_{2*half}^j = tw[j * tstride]
    for (int q = threadIdx.x; q < half_total; q += blockDim.x) {
      const int c = q >> (logn - 1), b = q & ((n >> 1) - 1);
      const int j = b & (half - 1);
      const int i0 = ((b - j) << 1) + j;
      cpx *p = s + c * pitch;
      const cpx a = p[i0], d = p[i0 + half];
      cpx w = tw[j * tstride];
      if (inverse) w = cconj(w);
      p[i0] = cadd(a, d);
      p[i0 + half] = cmul(csub(a, d), w);
    }
    __syncthreads();
  }
}

// One workgroup per transform (n <= 4096), contiguous in and out.
__global__ __launch_bounds__(FFT_THREADS) void fft_rows_kernel(const cpx *__restrict__ in, cpx *__restrict__ out,
                                                               const cpx *__restrict__ tw, int n, int logn,
                                                               int inverse, float scale)
{
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  cpx *s = reinterpret_cast<cpx *>(smem_raw);
  const cpx *x = in + (size_t) blockIdx.x * n;
  cpx *y = out + (size_t) blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = x[i];
  __syncthreads();
  dif_passes(s, n, 1, n, logn, tw, inverse != 0);
  for (int i = threadIdx.x; i < n; i += blockDim.x) y[i] = cscale(s[bitrev(i, logn)], scale);
}

// Column FFTs of a row-major [R][C] matrix (FFT length R along the row index), COL_TILE
// columns per workgroup staged through LDS.  blockIdx.y = batch.
//   TRANSPOSE = true : out is [C][R]: out[c][k] = FFT_c[k] * W_N^(c*k)   (four-step pass 1)
//   TRANSPOSE = false: out is [R][C]: out[k][c] = FFT_c[k] * scale       (four-step pass 2)
// W_N^m is looked up as thi[m >> 12] * tlo[m & 4095].
template <bool TRANSPOSE>
__global__ __launch_bounds__(FFT_THREADS) void fft_cols_kernel(const cpx *__restrict__ in, cpx *__restrict__ out,
                                                               const cpx *__restrict__ tw, int R, int logR, int C,
                                                               const cpx *__restrict__ thi, const cpx *__restrict__ tlo,
                                                               int inverse, float scale, int tile)
{
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  cpx *s = reinterpret_cast<cpx *>(smem_raw);
  const int pitch = R + 1;                         // odd pitch: column-strided LDS access stays conflict-light
  const size_t boff = (size_t) blockIdx.y * (size_t) R * C;
  const cpx *x = in + boff;
  cpx *y = out + boff;
  const int c0 = blockIdx.x * tile;
  const int ncol = min(tile, C - c0);
  // load: consecutive threads walk the tile's columns first (contiguous in memory)
  for (int q = threadIdx.x; q < R * tile; q += blockDim.x) {
    const int r = q / tile, c = q - r * tile;
    if (c < ncol) s[c * pitch + r] = x[(size_t) r * C + c0 + c];
  }
  __syncthreads();
  dif_passes(s, pitch, ncol, R, logR, tw, inverse != 0);
  if (TRANSPOSE) {
    for (int q = threadIdx.x; q < R * ncol; q += blockDim.x) {
      const int c = q / R, k = q - c * R;
      cpx v = s[c * pitch + bitrev(k, logR)];
      const unsigned m = (unsigned) (c0 + c) * (unsigned) k;     // < N <= 2^24
      cpx w = cmul(thi[m >> 12], tlo[m & 4095]);
      if (inverse) w = cconj(w);
      y[(size_t) (c0 + c) * R + k] = cmul(v, w);
    }
  } else {
    for (int q = threadIdx.x; q < R * tile; q += blockDim.x) {
      const int k = q / tile, c = q - k * tile;
      if (c < ncol) y[(size_t) k * C + c0 + c] = cscale(s[c * pitch + bitrev(k, logR)], scale);
    }
  }
}

// ---- fast paths on the in-wave 1024-point FFT (fft1024_wave.hpp) ---------------------------
// Inverse transforms use conj(FFT(conj(x))): one forward code path, conjugation at the first
// load and the last store.
__device__ __forceinline__ void wave_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// n = 1024: one wave per transform, 4 transforms per workgroup, contiguous in and out.
__global__ __launch_bounds__(256) void fft1024_rows_kernel(const cpx *__restrict__ in, cpx *__restrict__ out,
                                                           const cpx *__restrict__ TW1, const cpx *__restrict__ TW2,
                                                           int inverse, float scale, int ntr)
{
  __shared__ cpx lds[4 * w1024::LDS_ELEMS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tr = blockIdx.x * 4 + wv;
  if (tr >= ntr) return;
  cpx tw1[16], tw2[16], v[16];
  const cpx *x = in + (size_t) tr * 1024;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    v[r] = x[64 * r + lane];
    if (inverse) v[r].y = -v[r].y;
    tw1[r] = TW1[r * 64 + lane];
    tw2[r] = TW2[r * 64 + lane];
  }
  w1024::forward<1>(v, lds + wv * w1024::LDS_ELEMS, lane, tw1, tw2, wave_fence);
  cpx *y = out + (size_t) tr * 1024;
  const int k0 = (lane >> 2) + 16 * (lane & 3);            // freq_index(lane, r) = k0 + 64*(r>>2) + 256*(r&3)
#pragma unroll
  for (int r = 0; r < 16; r++) {
    cpx o = cscale(v[r], scale);
    if (inverse) o.y = -o.y;
    y[k0 + 64 * (r >> 2) + 256 * (r & 3)] = o;
  }
}

// ---- n = 16 .. 16384: Stockham autosort in LDS, radix 16 ---------------------------------------
// n = R0 * 16^a with R0 in {16, 8, 4, 2}: the first pass has radix R0 and needs no twiddles,
// every further pass is a 16-point in-register DFT (w1024::dft16).  Thread j of the n/16
// threads of a transform owns butterfly j of every pass (Stockham indexing: pass with sub-
// transform length Ns reads x[j + q n/16], multiplies by W_{16 Ns}^{q k}, k = j mod Ns, and writes
// y[(j - k) 16 + k + q Ns]), so log16(n) passes replace the log2(n) of the radix-2 kernel and
// the LDS sees each point once per pass.  The twiddle of a butterfly is ONE table value
// (W_{16 Ns}^k) raised to q = 2..15 by a depth-4 product tree in registers.  LDS index i is
// stored at i + i/16 (keeps the stride-16 writes of the passes conflict-free).
// Workgroup = max(256, n/16) threads = 4096/n transforms (n < 4096) or one.  From n = 64 the
// first pass loads and the last pass stores straight from registers (runs of n/2 contiguous
// bytes per transform, 512 B per wave instruction from n = 1024); n = 16 and 32 stage their
// global accesses through LDS to keep them 16-B coalesced.
template <int R0>
__global__ __launch_bounds__(1024) void fft_s16_kernel(const cpx *__restrict__ in, cpx *__restrict__ out,
                                                       const cpx *__restrict__ TW, int n, int tpt, int inverse,
                                                       float scale, int ntr, const cpx *__restrict__ rrot)
{
  // rrot != nullptr: REAL FFT of 2n points per transform (RTFRPlan::step, fourier.cc:315-347): the n
  // complex points are the packed pairs, and the untangling + forced conjugate symmetry below
  // replace the plain store -- the half-size spectrum never goes to HBM.
  extern __shared__ __attribute__((aligned(16))) char s16_raw[];
  cpx *lds = reinterpret_cast<cpx *>(s16_raw);
  const int t = threadIdx.x;
  const int tl = t / tpt, j = t - tl * tpt;
  const int T = blockDim.x / tpt;                          // transforms per workgroup
  const int tr = blockIdx.x * T + tl;
  const bool live = tr < ntr;
  const int pn = n + (n >> 4);
  cpx *s = lds + tl * pn;
  const bool staged = tpt < S16_DIRECT_TPT;
  const int64_t g0 = (int64_t) blockIdx.x * T * n, gend = (int64_t) ntr * n;   // this workgroup's points
  cpx v[16];
  if (staged) {
    // coalesced 16-B loads of the workgroup's T*n points into the padded per-transform images
    const int tot2 = (T * n) >> 1;
    for (int i0 = t; i0 < tot2; i0 += 8 * blockDim.x) {      // 8 loads in flight per thread
      float4 q[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int e = 2 * (i0 + u * (int) blockDim.x);
        q[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < 2 * tot2 && g0 + e < gend) q[u] = *reinterpret_cast<const float4 *>(in + g0 + e);
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int e = 2 * (i0 + u * (int) blockDim.x);
        if (e < 2 * tot2) {
          const int tq = e / n, eq = e - tq * n;
          cpx *d = lds + tq * pn + s16::pad(eq);
          d[0] = cmk(q[u].x, inverse ? -q[u].y : q[u].y);
          d[1] = cmk(q[u].z, inverse ? -q[u].w : q[u].w);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 16; m++) v[m] = s[s16::pad(j + m * tpt)];
    __syncthreads();
  } else {
    const cpx *x = in + (size_t) (live ? tr : 0) * n;
#pragma unroll
    for (int m = 0; m < 16; m++) {
      v[m] = x[j + m * tpt];
      if (inverse) v[m].y = -v[m].y;
    }
  }
  // ---- pass 0: radix R0, no twiddles
  s16::pass0<R0>(v);
  // v[i + q * (16 / R0)] = output q of butterfly jb = j + i * tpt  ->  y[jb * R0 + q]
  if (n == R0) {                                           // n = 16: a single pass (always staged: tpt = 1)
#pragma unroll
    for (int q = 0; q < 16; q++) s[s16::pad(q)] = v[q];
  } else {
    s16::pass0_store<R0>(s, v, j, tpt);
    // ---- radix-16 passes
    for (int Ns = R0;; Ns <<= 4) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 16; q++) v[q] = s[s16::pad(j + q * tpt)];
      const int k = j & (Ns - 1);
      s16::twiddle_powers(v, TW[k * (tpt / Ns)]);
      w1024::dft16<false>(v);
      const int base = (j - k) * 16 + k;
      const bool last = Ns * 16 == n;
      if (last && !staged && !rrot) {
        if (live) {
          cpx *y = out + (size_t) tr * n;
#pragma unroll
          for (int q = 0; q < 16; q++) {
            cpx o = cscale(v[q], scale);
            if (inverse) o.y = -o.y;
            y[base + q * Ns] = o;
          }
        }
        return;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 16; q++) s[s16::pad(base + q * Ns)] = v[q];
      if (last) break;
    }
  }
  __syncthreads();
  if (rrot) {
    // y(i) = r2 (Xt(i) + conj Xt(h-i)) - j2 (Xt(i) - conj Xt(h-i)) rot(i), i = 0..h, h = n; y(2h-i) = conj y(i)
    // (the arithmetic of rfft_untangle_kernel, on the transform still in LDS)
    const int h = n, per = h + 1, tot = T * per;
    const float c = 0.35355339059327373f;                    // (float) (0.5 / sqrt(2.0))
    int tq = t / per, i = t - tq * per;
    const int dq = (int) blockDim.x / per, dr = (int) blockDim.x - dq * per;
    for (int e = t; e < tot; e += blockDim.x) {
      const int trq = blockIdx.x * T + tq;
      if (trq < ntr) {
        const cpx *X = lds + tq * pn;
        const cpx X1 = cscale(X[s16::pad(i == h ? 0 : i)], scale), X2c = cconj(cscale(X[s16::pad(i > 0 ? h - i : 0)], scale));
        const cpx a = cscale(cadd(X1, X2c), c);
        const cpx d = csub(X1, X2c);
        const cpx jd = cmk(-d.y * c, d.x * c);
        cpx v = csub(a, cmul(jd, rrot[i]));
        if (i == 0 || i == h) v.y = 0.f;
        cpx *Y = out + (size_t) trq * 2 * h;
        Y[i] = v;
        if (i > 0 && i < h) Y[2 * h - i] = cconj(v);
      }
      tq += dq; i += dr;
      if (i >= per) { i -= per; tq++; }
    }
    return;
  }
  // staged store: coalesced 16 B per lane
  const int tot2 = (T * n) >> 1;
  for (int i = t; i < tot2; i += blockDim.x) {
    const int e = 2 * i;
    if (g0 + e >= gend) break;
    const int tq = e / n, eq = e - tq * n;
    const cpx *d = lds + tq * pn + s16::pad(eq);
    const cpx a = d[0], b = d[1];
    *reinterpret_cast<float4 *>(out + g0 + e) = make_float4(a.x * scale, inverse ? -a.y * scale : a.y * scale, b.x * scale,
                                                            inverse ? -b.y * scale : b.y * scale);
  }
}

// ---- smooth sizes n = 2^a 3^b 5^c 7^d 11^e 13^f <= 16384: mixed-radix Stockham in LDS ---------
// The reference reaches such sizes by even/odd splits down to an odd length and Bluestein
// there (fourier.cc:391-464); the two-pass plan above does the same in two HBM passes.  When
// every prime factor is small the transform is ONE kernel with one pass over HBM: the n points
// sit in LDS and go through one autosort pass per factor (radix 16/8/4/2 for the power of two,
// then 3, 5, 7, 11, 13): butterfly j of a radix-R pass with sub-transform length Ns reads
// x[j + q n/R], multiplies by W_{R Ns}^(q k), k = j mod Ns (one table value W_n^(k n/(R Ns)) and
// its successive powers), takes the R-point DFT and writes y[(j-k) R + k + q Ns] -- the same
// indexing as fft_s16_kernel, any radix.  A thread keeps the inputs of all its butterflies of a
// pass in registers (at most MR_PTS points), so one LDS image per transform is enough.
// tpt threads per transform, blockDim/tpt transforms per workgroup; global accesses staged
// through LDS, coalesced.
constexpr int MR_PTS = 16;
constexpr int MR_MAXF = 14;
struct MrFactors { int nf; int r[MR_MAXF]; };
template <int R> struct OddW;
template <> struct OddW<3> { static constexpr float c[3] = {1.000000000e+00f, -5.000000000e-01f, -5.000000000e-01f}; static constexpr float s[3] = {-0.000000000e+00f, -8.660254038e-01f, 8.660254038e-01f}; };
template <> struct OddW<5> { static constexpr float c[5] = {1.000000000e+00f, 3.090169944e-01f, -8.090169944e-01f, -8.090169944e-01f, 3.090169944e-01f}; static constexpr float s[5] = {-0.000000000e+00f, -9.510565163e-01f, -5.877852523e-01f, 5.877852523e-01f, 9.510565163e-01f}; };
template <> struct OddW<7> { static constexpr float c[7] = {1.000000000e+00f, 6.234898019e-01f, -2.225209340e-01f, -9.009688679e-01f, -9.009688679e-01f, -2.225209340e-01f, 6.234898019e-01f}; static constexpr float s[7] = {-0.000000000e+00f, -7.818314825e-01f, -9.749279122e-01f, -4.338837391e-01f, 4.338837391e-01f, 9.749279122e-01f, 7.818314825e-01f}; };
template <> struct OddW<11> { static constexpr float c[11] = {1.000000000e+00f, 8.412535328e-01f, 4.154150130e-01f, -1.423148383e-01f, -6.548607339e-01f, -9.594929736e-01f, -9.594929736e-01f, -6.548607339e-01f, -1.423148383e-01f, 4.154150130e-01f, 8.412535328e-01f}; static constexpr float s[11] = {-0.000000000e+00f, -5.406408175e-01f, -9.096319954e-01f, -9.898214419e-01f, -7.557495744e-01f, -2.817325568e-01f, 2.817325568e-01f, 7.557495744e-01f, 9.898214419e-01f, 9.096319954e-01f, 5.406408175e-01f}; };
template <> struct OddW<13> { static constexpr float c[13] = {1.000000000e+00f, 8.854560257e-01f, 5.680647467e-01f, 1.205366803e-01f, -3.546048870e-01f, -7.485107482e-01f, -9.709418174e-01f, -9.709418174e-01f, -7.485107482e-01f, -3.546048870e-01f, 1.205366803e-01f, 5.680647467e-01f, 8.854560257e-01f}; static constexpr float s[13] = {-0.000000000e+00f, -4.647231720e-01f, -8.229838659e-01f, -9.927088741e-01f, -9.350162427e-01f, -6.631226582e-01f, -2.393156643e-01f, 2.393156643e-01f, 6.631226582e-01f, 9.350162427e-01f, 9.927088741e-01f, 8.229838659e-01f, 4.647231720e-01f}; };

template <int R> __device__ __forceinline__ void mr_dft(cpx (&a)[R])
{
  if constexpr (R == 2) s16::dft2(a[0], a[1]);
  else if constexpr (R == 4) w1024::dft4<false>(a[0], a[1], a[2], a[3]);
  else if constexpr (R == 8) {
    // s16::dft8 takes the samples in natural order and leaves X[q] at e[q]
    s16::dft8(a);
  } else if constexpr (R == 16) w1024::dft16<false>(a);
  else {
    cpx o[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
      cpx acc = a[0];
#pragma unroll
      for (int p = 1; p < R; p++) {
        const int m = (p * q) % R;
        acc = cadd(acc, cmul(a[p], cmk(OddW<R>::c[m], OddW<R>::s[m])));
      }
      o[q] = acc;
    }
#pragma unroll
    for (int q = 0; q < R; q++) a[q] = o[q];
  }
}

template <int R>
__device__ __attribute__((noinline)) void mr_pass(int s_off, const cpx *__restrict__ TWn, int n, int Ns, int j0, int tpt)
{
  // (not inlined: every radix gets its own register allocation; the LDS image is named by its
  // offset so that the accesses stay ds_read / ds_write -- a pointer argument would be generic)
  extern __shared__ __attribute__((aligned(16))) char s16_raw[];
  cpx *s = reinterpret_cast<cpx *>(s16_raw) + s_off;
  constexpr int U = MR_PTS / R;
  const int nb = n / R;
  cpx v[U][R];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int j = j0 + u * tpt;
    if (j < nb) {
#pragma unroll
      for (int q = 0; q < R; q++) v[u][q] = s[s16::pad(j + q * nb)];
    }
  }
  __syncthreads();
  const int tstep = n / (R * Ns);
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int j = j0 + u * tpt;
    if (j < nb) {
      const int k = (Ns & (Ns - 1)) == 0 ? (j & (Ns - 1)) : j % Ns;
      if (Ns > 1) {
        // (the table is global memory: say so, or the non-inlined function issues a flat load)
        typedef float v2f_t __attribute__((ext_vector_type(2)));
        const v2f_t wv = *(const __attribute__((address_space(1))) v2f_t *) (TWn + k * tstep);
        const cpx w1 = cmk(wv.x, wv.y);
        if constexpr (R == 16) s16::twiddle_powers(v[u], w1);
        else {
          cpx w = w1;
#pragma unroll
          for (int q = 1; q < R; q++) {
            v[u][q] = cmul(v[u][q], w);
            w = cmul(w, w1);
          }
        }
      }
      mr_dft<R>(v[u]);
      const int base = (j - k) * R + k;
#pragma unroll
      for (int q = 0; q < R; q++) s[s16::pad(base + q * Ns)] = v[u][q];
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void fft_mr_kernel(const cpx *__restrict__ in, cpx *__restrict__ out,
                                                      const cpx *__restrict__ TWn, MrFactors F, int n, int tpt, int inverse,
                                                      float scale, int ntr)
{
  extern __shared__ __attribute__((aligned(16))) char s16_raw[];
  cpx *lds = reinterpret_cast<cpx *>(s16_raw);
  const int t = threadIdx.x, T = blockDim.x / tpt;
  const int tl = t / tpt, j0 = t - tl * tpt;
  const int pn = s16::pad(n) + 1;
  const int64_t g0 = (int64_t) blockIdx.x * T * n, gend = (int64_t) ntr * n;
  const int tot = T * n;
  const int nthr = blockDim.x;
  // element e of the workgroup's T*n points lives at image e / n, index e mod n: tracked
  // incrementally (e advances by nthr), no division per element
  const int dq = nthr / n, dr = nthr - dq * n;
  int tq = t / n, eq = t - tq * n;
  for (int e0 = t; e0 < tot; e0 += 8 * nthr) {           // 8 loads in flight per thread
    cpx a[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int e = e0 + u * nthr;
      a[u] = cmk(0.f, 0.f);
      if (e < tot && g0 + e < gend) a[u] = in[g0 + e];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int e = e0 + u * nthr;
      if (e < tot) {
        if (inverse) a[u].y = -a[u].y;
        lds[tq * pn + s16::pad(eq)] = a[u];
      }
      tq += dq; eq += dr;
      if (eq >= n) { eq -= n; tq++; }
    }
  }
  __syncthreads();
  int Ns = 1;
  for (int f = 0; f < F.nf; f++) {
    const int R = F.r[f];
    const int so = tl * pn;
    switch (R) {
      case 2: mr_pass<2>(so, TWn, n, Ns, j0, tpt); break;
      case 3: mr_pass<3>(so, TWn, n, Ns, j0, tpt); break;
      case 4: mr_pass<4>(so, TWn, n, Ns, j0, tpt); break;
      case 5: mr_pass<5>(so, TWn, n, Ns, j0, tpt); break;
      case 7: mr_pass<7>(so, TWn, n, Ns, j0, tpt); break;
      case 8: mr_pass<8>(so, TWn, n, Ns, j0, tpt); break;
      case 11: mr_pass<11>(so, TWn, n, Ns, j0, tpt); break;
      case 13: mr_pass<13>(so, TWn, n, Ns, j0, tpt); break;
      default: mr_pass<16>(so, TWn, n, Ns, j0, tpt); break;
    }
    Ns *= R;
  }
  tq = t / n;
  eq = t - tq * n;
#pragma unroll 4
  for (int e = t; e < tot; e += nthr) {
    cpx a = cscale(lds[tq * pn + s16::pad(eq)], scale);
    if (inverse) a.y = -a.y;
    if (g0 + e < gend) out[g0 + e] = a;
    tq += dq; eq += dr;
    if (eq >= n) { eq -= n; tq++; }
  }
}

// ---- four-step passes on the radix-16 Stockham engine (n = 2^15 .. 2^24, except 2^20) ---------
// Column FFTs of a row-major [L][C] matrix (length L along the row index), CT adjacent
// columns per workgroup (CT * 8 B row segments, 16-B accesses), each column transformed in LDS
// by L/16 threads exactly like a row of fft_s16_kernel (column c of the tile lives at
// s[c * pn + pad(i)]).  blockIdx.y = batch.
//   PASS 1: out is [C][L]: out[c][k] = FFT_c[k] * W_N^(c*k)   (transposed: row c contiguous)
//   PASS 2: out is [L][C]: out[k][c] = FFT_c[k] * scale       (natural order)
// W_N^m is looked up as thi[m >> 12] * tlo[m & 4095].  Inverse: conj at the load of pass 1 and
// at the store of pass 2 (one forward code path).
template <int PASS, int R0>
__global__ __launch_bounds__(1024) void fft_cols16_kernel(const cpx *__restrict__ in, cpx *__restrict__ out,
                                                          const cpx *__restrict__ TW, int L, int tpt, int C, int CT,
                                                          const cpx *__restrict__ thi, const cpx *__restrict__ tlo,
                                                          int inverse, float scale, int ragged, int ipitch)
{
  // (ipitch: elements between the rows of `in` -- C, or the padded pitch of an intermediate written by fft1m_cols_kernel<1>)
  extern __shared__ __attribute__((aligned(16))) char s16_raw[];
  cpx *lds = reinterpret_cast<cpx *>(s16_raw);
  const int t = threadIdx.x, nthr = blockDim.x;
  const int pn = L + (L >> 4) + 1;
  const size_t boff = (size_t) blockIdx.y * (size_t) L * C;
  const cpx *x = in + (size_t) blockIdx.y * (size_t) L * ipitch;
  cpx *y = out + boff;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs, so blockIdx.x, +8, +16 ...
  // share an L2.  Tiles narrower than a 128-B line (CT < 16) split cache lines with their
  // neighbours: give neighbouring tiles to the SAME XCD, back to back, so that the second one
  // finds the line in L2 instead of fetching it again from HBM.
  int tile = blockIdx.x;
  if (CT < 16 && (gridDim.x & 7) == 0) tile = (int) (blockIdx.x & 7) * (int) (gridDim.x >> 3) + (int) (blockIdx.x >> 3);
  const int c0 = tile * CT, h = CT >> 1;
  if (ragged) {
    // C is not a multiple of CT (or odd): 8-B accesses, columns beyond C read as zero
    for (int q = t; q < L * CT; q += nthr) {
      const int r = q / CT, c = q - r * CT;
      cpx f = cmk(0.f, 0.f);
      if (c0 + c < C) f = x[(size_t) r * ipitch + c0 + c];
      if (PASS == 1 && inverse) f.y = -f.y;
      lds[c * pn + s16::pad(r)] = f;
    }
  } else
  {
    // L*CT/2 float4 over CT*L/16 threads = exactly 8 per thread: all 8 loads in flight at once
    float4 f[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int q = t + i * nthr, r = q / h, c = 2 * (q - r * h);
      f[i] = *reinterpret_cast<const float4 *>(x + (size_t) r * ipitch + c0 + c);
    }
    const float sg = (PASS == 1 && inverse) ? -1.f : 1.f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int q = t + i * nthr, r = q / h, c = 2 * (q - r * h);
      lds[c * pn + s16::pad(r)] = cmk(f[i].x, sg * f[i].y);
      lds[(c + 1) * pn + s16::pad(r)] = cmk(f[i].z, sg * f[i].w);
    }
  }
  __syncthreads();
  const int cl = t / tpt, j = t - cl * tpt;
  cpx *s = lds + cl * pn;
  cpx v[16];
  // pass 1: the four-step twiddle W_N^(c*k) of this thread's 16 outputs k = j + q*tpt is
  // W_N^(c*j) * (W_N^(c*tpt))^q: two table look-ups and a product tree, applied in registers
  // before the last write to LDS (a look-up per point in the store loop cost 2x the pass)
  auto four_step_twiddle = [&](cpx (&u)[16]) {
    const unsigned c = (unsigned) (c0 + cl);
    const unsigned mb = c * (unsigned) j, md = c * (unsigned) tpt;            // < N <= 2^24
    const cpx b = cmul(thi[mb >> 12], tlo[mb & 4095]), d1 = cmul(thi[md >> 12], tlo[md & 4095]);
    const cpx d2 = cmul(d1, d1), d3 = cmul(d2, d1), d4 = cmul(d2, d2);
    const cpx d5 = cmul(d4, d1), d6 = cmul(d4, d2), d7 = cmul(d4, d3), d8 = cmul(d4, d4);
    const cpx b8 = cmul(b, d8);
    u[0] = cmul(u[0], b);
    u[1] = cmul(u[1], cmul(b, d1)); u[2] = cmul(u[2], cmul(b, d2)); u[3] = cmul(u[3], cmul(b, d3));
    u[4] = cmul(u[4], cmul(b, d4)); u[5] = cmul(u[5], cmul(b, d5)); u[6] = cmul(u[6], cmul(b, d6));
    u[7] = cmul(u[7], cmul(b, d7)); u[8] = cmul(u[8], b8);
    u[9] = cmul(u[9], cmul(b8, d1)); u[10] = cmul(u[10], cmul(b8, d2)); u[11] = cmul(u[11], cmul(b8, d3));
    u[12] = cmul(u[12], cmul(b8, d4)); u[13] = cmul(u[13], cmul(b8, d5)); u[14] = cmul(u[14], cmul(b8, d6));
    u[15] = cmul(u[15], cmul(b8, d7));
  };
#pragma unroll
  for (int m = 0; m < 16; m++) v[m] = s[s16::pad(j + m * tpt)];
  __syncthreads();
  s16::pass0<R0>(v);
  if (PASS == 1 && R0 == 16 && L == 16) four_step_twiddle(v);
  s16::pass0_store<R0>(s, v, j, tpt);
  for (int Ns = R0; Ns < L; Ns <<= 4) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = s[s16::pad(j + q * tpt)];
    const int k = j & (Ns - 1);
    s16::twiddle_powers(v, TW[k * (tpt / Ns)]);
    w1024::dft16<false>(v);
    const int base = (j - k) * 16 + k;
    if (PASS == 1 && Ns * 16 == L) four_step_twiddle(v);     // last pass: outputs k = j + q * tpt
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; q++) s[s16::pad(base + q * Ns)] = v[q];
  }
  __syncthreads();
  if (PASS == 1) {
    const int hl = L >> 1;
    for (int q = t; q < CT * hl; q += nthr) {
      const int c = q / hl, k = 2 * (q - c * hl);
      const cpx *d = lds + c * pn + s16::pad(k);
      const cpx a = d[0], b = d[1];
      *reinterpret_cast<float4 *>(y + (size_t) (c0 + c) * L + k) = make_float4(a.x, a.y, b.x, b.y);
    }
  } else if (ragged) {
    for (int q = t; q < L * CT; q += nthr) {
      const int k = q / CT, c = q - k * CT;
      if (c0 + c < C) {
        const cpx a = lds[c * pn + s16::pad(k)];
        y[(size_t) k * C + c0 + c] = cmk(a.x * scale, inverse ? -a.y * scale : a.y * scale);
      }
    }
  } else {
    for (int q = t; q < L * h; q += nthr) {
      const int k = q / h, c = 2 * (q - k * h);
      const cpx a = lds[c * pn + s16::pad(k)], b = lds[(c + 1) * pn + s16::pad(k)];
      const float si = inverse ? -scale : scale;
      *reinterpret_cast<float4 *>(y + (size_t) k * C + c0 + c) = make_float4(a.x * scale, a.y * si, b.x * scale, b.y * si);
    }
  }
}

// ---- n = m * P, m odd <= 31, P = 2^p in 16 .. 4096 (e.g. 15360 = 15 * 1024) --------------------
// The reference reaches such sizes by p levels of even/odd split down to m-point Bluestein
// transforms (fourier.cc:438-464), i.e. a radix-2 decimation in time over the 2^p factor.  The
// same transform as two passes: (1) for every residue r = index mod P, the m-point DFT of the
// decimated sequence x[r + P i] -- evaluated directly, one thread per r with the m samples
// in registers -- times the four-step twiddle W_n^(r k1), stored as Z[r][k1]; (2) P-point
// column FFTs of Z over r (fft_cols16_kernel<2>, ragged columns), which leave
// X[k1 + m k2] at [k2][k1] = natural order.  For m <= 31 the direct m-point DFT differs from
// the reference's float32-chirp Bluestein by < 5e-6 (larger odd parts keep the recursion).
template <int MMAX>
__global__ __launch_bounds__(256) void fft_odd_dft_kernel(const cpx *__restrict__ in, cpx *__restrict__ z,
                                                          const cpx *__restrict__ Wm, const cpx *__restrict__ Wn, int m,
                                                          int P, int inverse, float scale, int64_t total)
{
  extern __shared__ __attribute__((aligned(16))) char odd_raw[];
  cpx *wm = reinterpret_cast<cpx *>(odd_raw);                // W_m^j, j < m
  cpx *stage = wm + 32;                                      // 256 x m outputs of the workgroup
  const int t = threadIdx.x;
  if (t < m) wm[t] = Wm[t];
  __syncthreads();
  const int64_t g0 = (int64_t) blockIdx.x * 256, g = g0 + t;
  if (g < total) {
    const int64_t b = g / P;
    const int r = (int) (g - b * P);
    const cpx *x = in + (size_t) b * m * P + r;
    cpx u[MMAX];
#pragma unroll
    for (int i = 0; i < MMAX; i++) {
      u[i] = cmk(0.f, 0.f);
      if (i < m) {
        u[i] = x[(size_t) P * i];
        if (inverse) u[i].y = -u[i].y;
      }
    }
    for (int k1 = 0; k1 < m; k1++) {
      cpx acc = u[0];
      int idx = 0;
#pragma unroll
      for (int i = 1; i < MMAX; i++) {
        idx += k1;
        if (idx >= m) idx -= m;
        if (i < m) {
          const cpx w = wm[idx];
          acc.x = fmaf(u[i].x, w.x, fmaf(-u[i].y, w.y, acc.x));
          acc.y = fmaf(u[i].x, w.y, fmaf(u[i].y, w.x, acc.y));
        }
      }
      stage[t * m + k1] = cscale(cmul(acc, Wn[r * k1]), scale);   // odd m: the lane stride 2m dwords is conflict-free
    }
  }
  __syncthreads();
  // Z[(b P + r) m + k1]: the workgroup's 256 residues are one contiguous run of 256 m values
  const int64_t zbase = g0 * m, zend = total * m;
  for (int i = t; i < 256 * m; i += 256)
    if (zbase + i < zend) z[zbase + i] = stage[i];
}

// n = 2^20 = 1024 x 1024, four-step.  One workgroup = 16 waves = 16 adjacent columns of the
// row-major [1024][1024] matrix; the tile [1024 rows][16 cols] is loaded with 16-B accesses
// (128-B row segments), staged in LDS at pitch 17 (odd -> the column each wave reads, and the
// exchange image it then reuses in place, are bank-conflict free; the 16-B fills and read-backs of
// the tile rows are not -- the hardware serves a 128-bit access in lane groups that straddle two
// rows: PMC counts 24-28 % of the LDS-active cycles as conflicts, in a kernel that runs at the
// rate of a copy), transformed by the wave.
//   PASS 1: out is the transposed matrix, out[c][k] = FFT_c[k] * W_N^(c*k)  (row c contiguous)
//   PASS 2: out[k][c] = FFT_c[k] * scale (natural order; staged back through LDS)
// Twiddle W_N^(c*k), k = k0(lane) + 64*(r>>2) + 256*(r&3): TA[c][lane] * TD[c][r] (host tables).
//
// The 148 KiB tile allows one workgroup per CU, so a workgroup is persistent (tiles id, id +
// grid, ...) and software-pipelined: as soon as the 8 float4 of a tile have been written to
// LDS the same registers receive the loads of the NEXT tile, which stay in flight through the
// transform and the store phase of the current one.  That needs (a) barriers that only wait
// for LDS (a __syncthreads() drains vmcnt and with it the prefetch), and (b) the wave-FFT
// twiddles out of registers: they sit in the last 8 KiB of LDS (tw1 rows 1..15; tw2 depends on
// lane & 3 only).  Without the pipeline the load, transform and store phases of a CU ran
// back to back (measured: pass 1 = 0.65 ms without stores + 0.49 ms without loads = 0.99 ms).
constexpr int F1M_ROWS = 1088, F1M_PITCH = 17;
constexpr int F1M_TILE_ELEMS = F1M_ROWS * F1M_PITCH;
constexpr size_t F1M_LDS = (size_t) (F1M_TILE_ELEMS + 15 * 64 + 16 * 4 + 256 + 2) * sizeof(cpx);   // 158,224 B (the last 16: the pulled tile ids)

__device__ __forceinline__ void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// ---- n = 4096 .. 16384 in batches: the Stockham kernel above with PERSISTENT workgroups -------
// One transform per workgroup (n/16 threads) and few workgroups per CU (one for n = 16384): the
// load, transform and store phases of fft_s16_kernel then run back to back on a CU.  Here a
// workgroup walks transforms tr, tr + grid, ... and loads the next one into a second register
// set before transforming the current one; the barriers wait for LDS only (lds_barrier), so
// the loads -- and the previous transform's stores -- stay in flight through the passes.
template <int R0>
__device__ __forceinline__ void s16p_one(cpx (&v)[16], cpx *__restrict__ s, const cpx *__restrict__ TW, cpx *__restrict__ y,
                                         int n, int tpt, int j, int inverse, float scale)
{
  if (inverse) {
#pragma unroll
    for (int m = 0; m < 16; m++) v[m].y = -v[m].y;
  }
  s16::pass0<R0>(v);
  s16::pass0_store<R0>(s, v, j, tpt);
  for (int Ns = R0;; Ns <<= 4) {
    lds_barrier();
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = s[s16::pad(j + q * tpt)];
    const int k = j & (Ns - 1);
    s16::twiddle_powers(v, TW[k * (tpt / Ns)]);
    w1024::dft16<false>(v);
    const int base = (j - k) * 16 + k;
    if (Ns * 16 == n) {
#pragma unroll
      for (int q = 0; q < 16; q++) {
        cpx o = cscale(v[q], scale);
        if (inverse) o.y = -o.y;
        y[base + q * Ns] = o;
      }
      break;
    }
    lds_barrier();
#pragma unroll
    for (int q = 0; q < 16; q++) s[s16::pad(base + q * Ns)] = v[q];
  }
  lds_barrier();     // the next transform's pass 0 rewrites the image
}

// (Measured in round 3 and not kept: the transforms handed out dynamically like the tiles of fft1m_cols_kernel --
// 0.2765 against 0.2728 ms per 2^26 points at n = 16384: this kernel is not limited by its static partition.)
template <int R0>
__global__ __launch_bounds__(1024) void fft_s16_persistent_kernel(const cpx *__restrict__ in, cpx *__restrict__ out,
                                                                  const cpx *__restrict__ TW, int n, int tpt, int inverse,
                                                                  float scale, int ntr)
{
  extern __shared__ __attribute__((aligned(16))) char s16_raw[];
  cpx *s = reinterpret_cast<cpx *>(s16_raw);
  const int j = threadIdx.x, G = gridDim.x;
  int tr = blockIdx.x;
  if (tr >= ntr) return;
  cpx A[16], B[16];
  auto fetch = [&](cpx (&v)[16], int t_) {
    const cpx *x = in + (size_t) t_ * n + j;
#pragma unroll
    for (int m = 0; m < 16; m++) v[m] = x[m * tpt];
  };
  fetch(A, tr);
  for (;;) {
    if (tr + G < ntr) fetch(B, tr + G);
    s16p_one<R0>(A, s, TW, out + (size_t) tr * n, n, tpt, j, inverse, scale);
    tr += G;
    if (tr >= ntr) break;
    if (tr + G < ntr) fetch(A, tr + G);
    s16p_one<R0>(B, s, TW, out + (size_t) tr * n, n, tpt, j, inverse, scale);
    tr += G;
    if (tr >= ntr) break;
  }
}

// complex type of the 1M-point column kernel: -DF1M_PACKED=1 selects the packed (VOP3P) flavour of
// fft1024_wave.hpp (a complex = one 64-bit VGPR pair, half the VALU instructions)
#ifndef F1M_PACKED
#define F1M_PACKED 0
#endif
#if F1M_PACKED
using f1c = w1024::v2f;
__device__ __forceinline__ f1c f1mul(f1c a, cpx w) { return w1024::cmul(a, (f1c){w.x, w.y}); }
__device__ __forceinline__ f1c f1out(f1c a, float s, int inverse) { a = a * s; if (inverse) a.y = -a.y; return a; }
#else
using f1c = cpx;
__device__ __forceinline__ f1c f1mul(f1c a, cpx w) { return cmul(a, w); }
__device__ __forceinline__ f1c f1out(f1c a, float s, int inverse) { a = cscale(a, s); if (inverse) a.y = -a.y; return a; }
#endif
struct LdsTw1 {   // [r - 1][lane]
  const f1c *p;
  __device__ __forceinline__ f1c operator[](int r) const { return p[(r - 1) * 64]; }
};
struct LdsTw2 {   // [r][lane & 3]
  const f1c *p;
  __device__ __forceinline__ f1c operator[](int r) const { return p[r * 4]; }
};

struct F1mCtx {
  const cpx *in;
  cpx *out;
  const cpx *TA, *TD;
  cpx *tile, *col, *ltd;
  LdsTw1 tw1;
  LdsTw2 tw2;
  int ipitch, opitch, inverse, t, lane, wv, rr, cc, k0, grid;
  int tshift;          // log2 of the column tiles per transform (6: 1024 columns; pass 1 of a 1024 x C plan: log2(C / 16))
  int orows;           // rows of a transform's output block (pass 1: its columns, i.e. C; pass 2: 1024)
  int lc1;             // pass 2 as the LAST pass of a three-pass plan (n = 1024 x C1 x 1024): transform T' = (b, p), p < C1 = 2^lc1,
                       // stores its row q at y[b n + (C1 q + p) 1024 ..]: row pitch C1 x 1024, base p x 1024 (0: the two-pass plans)
  float scale;
};

// tile id -> (transform, column tile): consecutive ids are adjacent 128-B column segments of one transform (measured on the three-pass
// plan, whose rows are 128 KiB apart: spreading the column tiles -- times 17 mod 2^tshift -- or interleaving the transforms changes
// nothing or costs 2-3 %)
__device__ __forceinline__ void f1m_decode(const F1mCtx &k, int id, int &tq, int &ct)
{
  ct = id & ((1 << k.tshift) - 1);
  tq = id >> k.tshift;
}
// loads of tile `id` into registers: 8 x 16 B of the tile + this tile's share of the TA/TD tables
// Addresses are formed as (uniform tile base) + (32-bit per-thread byte offset), and the offset
// is made opaque once per tile: otherwise the 8 row addresses of the loads and of the stores
// are hoisted out of the tile loop as 64-bit loop invariants (32 VGPRs) and the prefetch spills.
__device__ __forceinline__ unsigned opaque(unsigned v)
{
  asm volatile("" : "+v"(v));
  return v;
}
template <int PASS>
__device__ __forceinline__ void f1m_issue(const F1mCtx &k, int id, float4 (&q)[8], cpx &ta, cpx &td)
{
  int tq_, ct;
  f1m_decode(k, id, tq_, ct);
  const char *x = reinterpret_cast<const char *>(k.in + (size_t) tq_ * 1024 * k.ipitch + ct * 16);
  const unsigned o = opaque(((unsigned) k.rr * k.ipitch + k.cc) * 8u), step = 128u * 8u * k.ipitch;
#pragma unroll
  for (int i = 0; i < 8; i++) q[i] = *reinterpret_cast<const float4 *>(x + (o + step * i));
  if (PASS == 1) {
    ta = k.TA[(ct * 16 + k.wv) * 64 + k.lane];
    td = k.TD[ct * 256 + (k.t & 255)];                     // [column][r] of the 16 columns
  }
}

// one tile: registers -> LDS, (prefetch of the next tile), transform, store
// nid: the tile whose loads are issued while this one is transformed (PREFETCH); publish: a tile id thread 0 pulled at the
// start of this tile, left in LDS for everybody before the tile's last barrier (see fft1m_cols_kernel)
template <int PASS, bool PREFETCH>
__device__ __forceinline__ void f1m_tile(const F1mCtx &k, int id, int nid, float4 (&q)[8], cpx &ta, cpx &td, int publish = -1, int *lnext = nullptr)
{
  constexpr int P = F1M_PITCH;
  cpx *tile = k.tile, *col = k.col;
  const int lane = k.lane, rr = k.rr, cc = k.cc, k0 = k.k0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int row = rr + 128 * i;
    const float sg = (PASS == 1 && k.inverse) ? -1.f : 1.f;
    tile[row * P + cc] = cmk(q[i].x, sg * q[i].y);
    tile[row * P + cc + 1] = cmk(q[i].z, sg * q[i].w);
  }
  const cpx ta_cur = ta;
  if (PASS == 1 && k.t < 256) k.ltd[k.t] = td;
  if (PREFETCH) f1m_issue<PASS>(k, nid, q, ta, td);   // in flight until the next tile's LDS write
  lds_barrier();
  f1c v[16];
  f1c *colc = reinterpret_cast<f1c *>(col);
#pragma unroll
  for (int r = 0; r < 16; r++) v[r] = colc[(64 * r + lane) * P];
  w1024::forward<P>(v, colc, lane, k.tw1, k.tw2, wave_fence);
  int tq_, ct_;
  f1m_decode(k, id, tq_, ct_);
  const int c0 = ct_ * 16, c = c0 + k.wv;
  const size_t tq = (size_t) tq_;
  cpx *y = PASS == 2 ? k.out + ((tq >> k.lc1) << (20 + k.lc1)) + (tq & ((1u << k.lc1) - 1)) * 1024 : k.out + tq * k.orows * k.opitch;
  wave_fence();
  if (PASS == 1) {
    // row c of the transposed intermediate.  The spectrum leaves the wave through its own LDS
    // column so that every lane stores 16 B (two adjacent bins): 8-B-per-lane stores from
    // the register order ran the write phase at 2.8 TB/s.
    cpx *z = y + (size_t) c * k.opitch;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const cpx w = cmul(ta_cur, k.ltd[k.wv * 16 + r]);
      colc[(k0 + 64 * (r >> 2) + 256 * (r & 3)) * P] = f1mul(v[r], w);
    }
    wave_fence();
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const cpx e0 = col[(128 * j + 2 * lane) * P], e1 = col[(128 * j + 2 * lane + 1) * P];
      *reinterpret_cast<float4 *>(z + 128 * j + 2 * lane) = make_float4(e0.x, e0.y, e1.x, e1.y);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      colc[(k0 + 64 * (r >> 2) + 256 * (r & 3)) * P] = f1out(v[r], k.scale, k.inverse);
    }
    lds_barrier();
    char *yb = reinterpret_cast<char *>(y + c0);
    const unsigned o = opaque(((unsigned) rr * k.opitch + cc) * 8u), step = 128u * 8u * k.opitch;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int row = rr + 128 * i;
      const cpx a = tile[row * P + cc], b = tile[row * P + cc + 1];
      *reinterpret_cast<float4 *>(yb + (o + step * i)) = make_float4(a.x, a.y, b.x, b.y);
    }
  }
  if (lnext && k.t == 0) lnext[0] = publish;
  lds_barrier();                                            // the next tile overwrites the LDS image
}

// ctr != NULL: the tiles are handed out DYNAMICALLY (one counter, never reset: the launch starts from `base` and leaves it
// at base + ntiles + workgroups -- one failing pull per workgroup) instead of tile = workgroup + k * grid: a static
// partition of a persistent grid streams 6-9 % below the same traffic handed out in order (scripts/ubench/copy_shapes.hip).
// Thread 0 pulls the id of the tile after next at the start of a tile and leaves it in LDS before the tile's last barrier.
template <int PASS, bool DYN>
__global__ __launch_bounds__(1024) void fft1m_cols_kernel(const cpx *__restrict__ in, cpx *__restrict__ out,
                                                          const cpx *__restrict__ TW1, const cpx *__restrict__ TW2,
                                                          const cpx *__restrict__ TA, const cpx *__restrict__ TD,
                                                          int inverse, float scale, int zp, int ntiles, unsigned *ctr, unsigned base,
                                                          int tshift, int lc1)
{
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  cpx *tile = reinterpret_cast<cpx *>(smem_raw);
  cpx *ltw1 = tile + F1M_TILE_ELEMS, *ltw2 = ltw1 + 15 * 64, *ltd = ltw2 + 16 * 4;
  int *lnext = reinterpret_cast<int *>(ltd + 256);
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  if (t < 15 * 64) ltw1[t] = TW1[64 + t];
  if (t < 64) ltw2[t] = TW2[(t >> 2) * 64 + (t & 3)];
  F1mCtx k;
  k.in = in; k.out = out; k.TA = TA; k.TD = TD;
  k.tile = tile; k.col = tile + wv; k.ltd = ltd;           // this wave's column: slots e*P + wv
  k.tw1 = LdsTw1{reinterpret_cast<const f1c *>(ltw1) + lane}; k.tw2 = LdsTw2{reinterpret_cast<const f1c *>(ltw2) + (lane & 3)};
  // (pass 1 of a 1024 x C plan: C = 16 << tshift columns of 1024 points, input rows C apart, C transposed output rows of pitch zp)
  k.ipitch = PASS == 1 ? (16 << tshift) : zp; k.opitch = PASS == 1 ? zp : (1024 << lc1);
  k.tshift = tshift; k.orows = PASS == 1 ? (16 << tshift) : 1024; k.lc1 = lc1;
  k.inverse = inverse; k.t = t; k.lane = lane; k.wv = wv;
  k.rr = t >> 3; k.cc = 2 * (t & 7);                       // 8 threads x 16 B per 128-B row segment
  k.k0 = (lane >> 2) + 16 * (lane & 3);
  k.grid = gridDim.x; k.scale = scale;

  // tiles id, id + grid, ...: tile = (transform id / 64, column tile id % 64).  The first tile is
  // peeled so that the loop is entered in the same memory-counter state as its back edge
  // (prefetch loads followed by 8 stores): the compiler's wait before the LDS write is then
  // vmcnt(8) -- the stores of the previous tile keep draining -- rather than vmcnt(0).
  float4 q[8];
  cpx ta = cmk(1.f, 0.f), td = cmk(1.f, 0.f);
  if (DYN) {
    auto pull = [&]() -> int {
      const unsigned v = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
      return v < (unsigned) ntiles ? (int) v : -1;
    };
    if (t == 0) {
      const int a = pull();
      lnext[0] = a;
      lnext[1] = a >= 0 ? pull() : -1;
    }
    __syncthreads();
    int id = lnext[0], nid = lnext[1];
    __syncthreads();                                         // (lnext[0] is rewritten at the end of the first tile)
    if (id < 0) return;
    f1m_issue<PASS>(k, id, q, ta, td);
    if (nid >= 0) {                                          // (first tile peeled, as below: the loop's memory-counter state)
      int pulled = -1;
      if (t == 0) pulled = pull();                           // its latency hides under the tile
      f1m_tile<PASS, true>(k, id, nid, q, ta, td, pulled, lnext);
      id = nid;
      nid = lnext[0];
      while (nid >= 0) {
        pulled = -1;
        if (t == 0) pulled = pull();
        f1m_tile<PASS, true>(k, id, nid, q, ta, td, pulled, lnext);
        id = nid;
        nid = lnext[0];
      }
    }
    f1m_tile<PASS, false>(k, id, -1, q, ta, td);
    return;
  }
  int id = blockIdx.x;
  if (id >= ntiles) return;
  f1m_issue<PASS>(k, id, q, ta, td);
  if (id + k.grid < ntiles) {
    f1m_tile<PASS, true>(k, id, id + k.grid, q, ta, td);
    id += k.grid;
    while (id + k.grid < ntiles) {
      f1m_tile<PASS, true>(k, id, id + k.grid, q, ta, td);
      id += k.grid;
    }
  }
  f1m_tile<PASS, false>(k, id, -1, q, ta, td);
}

// ---- n = 2^23 .. 2^25 in THREE passes: n = 1024 x C1 x 1024 (C1 = 8, 16, 32) -----------------------------------------------------------
// Two passes need 4096-point columns from 2^23 on: a 128-B row segment of them is 512 KiB, so fft_cols16_kernel takes 32-B
// segments and runs at a quarter of the 2^20 plan's rate (0.18 of 8 TB/s at 2^24).  Three passes that all move 128-B segments:
//   pass 1   fft1m_cols_kernel<1>: 1024-point columns of x viewed [1024][C], C = C1 x 1024, four-step twiddle W_n^(c k1), stored
//            transposed: Z[c][k1], rows of pitch 1040
//   pass 2a  fft_planes_kernel (here): with c = 1024 a + b, a C1-point DFT over a -- the C1 PLANES of 1024 rows each, element-wise in
//            (b, k1): pure streaming, in place -- times W_C^(b p):  U[p][b][k1]
//   pass 2b  fft1m_cols_kernel<2>: for every (p, k1) the 1024-point FFT over b (the rows of plane p), bin q stored at
//            y[(C1 q + p) 1024 + k1]: X[k1 + 1024 (p + C1 q)]
// (index algebra: W_C^(c k2) with k2 = p + C1 q is W_C1^(a p) W_C^(b p) W_1024^(b q))
template <int C1, int VEC>
__global__ __launch_bounds__(256) void fft_planes_kernel(cpx *__restrict__ z, const cpx *__restrict__ TP, int zp, int64_t total)
{
  const int64_t g = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (g >= total) return;
  constexpr int PER_ROW = 1024 / VEC;                       // threads of a row: a wave stays inside one row b
  const int64_t rowid = g / PER_ROW;                        // batch * 1024 + b
  const int kq = (int) (g - rowid * PER_ROW) * VEC;
  const int b = __builtin_amdgcn_readfirstlane((int) (rowid & 1023));
  const int64_t bt = rowid >> 10;
  const size_t ps = (size_t) 1024 * zp;                     // plane stride
  cpx *base = z + ((size_t) bt * C1 * 1024 + b) * zp + kq;
  cpx e[VEC][C1];
#pragma unroll
  for (int a = 0; a < C1; a++) {
    if (VEC == 2) {
      const float4 q = *reinterpret_cast<const float4 *>(base + a * ps);
      e[0][a] = cmk(q.x, q.y);
      e[VEC - 1][a] = cmk(q.z, q.w);
    } else {
      e[0][a] = base[a * ps];
    }
  }
  const cpx *tw = TP + b * C1;                              // W_C^(b p), p < C1: wave-uniform
#pragma unroll
  for (int v = 0; v < VEC; v++) {
    if (C1 == 8) {
      cpx (&a8)[8] = reinterpret_cast<cpx (&)[8]>(e[v]);
      s16::dft8(a8);
    } else if (C1 == 16) {
      cpx (&a16)[16] = reinterpret_cast<cpx (&)[16]>(e[v]);
      w1024::dft16<false>(a16);
    } else {
      // 32 = 2 x 16, decimation in time: X[k] = E[k] + W_32^k O[k], X[k + 16] = E[k] - W_32^k O[k]
      cpx ev[16], od[16];
#pragma unroll
      for (int i = 0; i < 16; i++) { ev[i] = e[v][2 * i]; od[i] = e[v][2 * i + 1]; }
      w1024::dft16<false>(ev);
      w1024::dft16<false>(od);
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const float ang = -6.28318530717958647692f * (float) k / 32.0f;
        const cpx o = cmul(od[k], cmk(__builtin_cosf(ang), __builtin_sinf(ang)));
        e[v][k] = cadd(ev[k], o);
        e[v][k + 16] = csub(ev[k], o);
      }
    }
#pragma unroll
    for (int pq = 1; pq < C1; pq++) e[v][pq] = cmul(e[v][pq], tw[pq]);
  }
#pragma unroll
  for (int a = 0; a < C1; a++) {
    if (VEC == 2) *reinterpret_cast<float4 *>(base + a * ps) = make_float4(e[0][a].x, e[0][a].y, e[VEC - 1][a].x, e[VEC - 1][a].y);
    else base[a * ps] = e[0][a];
  }
}

// ---- 2048-point columns, sixteen per tile: fft2k_cols_kernel (pass 2 of n = 2^21 and 2^22) ----------------------------------
// A column of 2048 points times a 128-B row segment (16 columns) is 256 KiB: more than the LDS, so fft_cols16_kernel takes 4-8
// columns per tile -- 32- and 64-B row segments, and the pass runs at half the rate of the 1024-point plans.  Here the tile lives
// in the REGISTER FILE and goes through the 2^20 kernel's LDS image (1024 rows x 16 columns) as two half tiles, each transformed
// by the in-wave 1024-point FFT, one column per wave at a time; a radix-2 step joins the halves -- in time for pass 1 (even / odd
// rows, the combination in registers), in frequency for pass 2 (top / bottom rows, the butterfly at the row loads).
// EIGHT waves per workgroup, TWO columns per wave: a 512-thread workgroup may hold 256 registers per lane,
// which is what lets the loads of the next half tile (64 registers) stay in flight beside E (64) and the transform -- with sixteen
// one-column waves (128 registers) nothing fits beside E and O, and a tile's 46 k cycles of memory time and 35 k of arithmetic ran
// one after the other (0.300 ms per pass; profiles/EXPERIMENTS.md).  One persistent workgroup per CU (the image is 148 KiB), tiles
// handed out from a counter like the 2^20 kernel's.  Inverse transforms: conj at the last store (pass 1 conjugated its input).
template <int PASS, bool DYN>
__global__ __launch_bounds__(512) void fft2k_cols_kernel(const cpx *__restrict__ in, cpx *__restrict__ out,
                                                         const cpx *__restrict__ TW1, const cpx *__restrict__ TW2,
                                                         const cpx *__restrict__ W2K, int C, int ipitch, int opitch,
                                                         int inverse, float scale, int ntiles, unsigned *ctr, unsigned base,
                                                         const cpx *__restrict__ TA, const cpx *__restrict__ TD)
{
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int P = F1M_PITCH;
  cpx *tile = reinterpret_cast<cpx *>(smem_raw);
  cpx *ltw1 = tile + F1M_TILE_ELEMS, *ltw2 = ltw1 + 15 * 64;
  int *lnext = reinterpret_cast<int *>(ltw2 + 16 * 4 + 256);
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  for (int i = t; i < 15 * 64; i += 512) ltw1[i] = TW1[64 + i];
  if (t < 64) ltw2[t] = TW2[(t >> 2) * 64 + (t & 3)];
  const LdsTw1 tw1{reinterpret_cast<const f1c *>(ltw1) + lane};
  const LdsTw2 tw2{reinterpret_cast<const f1c *>(ltw2) + (lane & 3)};
  f1c *colc = reinterpret_cast<f1c *>(tile + 2 * wv);      // this wave's two columns: slots e * P + 2 wv + u
  const int rr = t >> 3, cc = 2 * (t & 7);                 // 8 threads x 16 B per 128-B row segment, rows rr + 64 i
  const int k0 = (lane >> 2) + 16 * (lane & 3);            // the engine's frequency of (lane, r): k0 + 64 (r >> 2) + 256 (r & 3)
  const cpx wl = W2K[k0];                                  // W_2048^k0
  const int tpt = C >> 4;                                  // column tiles per transform
  const size_t tstride_i = (size_t) 2048 * ipitch, tstride_o = (size_t) (PASS == 1 ? C : 2048) * opitch;      // (pass 1 writes C transposed rows)

  float4 q[16];
  // loads of half `h` (rows 2 j + h, j = rr + 64 i) of tile `id`
  auto issue = [&](int id, int h) {
    const int b = id / tpt, ct = id - b * tpt;
    const char *x = reinterpret_cast<const char *>(in + (size_t) b * tstride_i + (size_t) ct * 16);
    const unsigned o = opaque(((unsigned) (2 * rr + h) * (unsigned) ipitch + (unsigned) cc) * 8u), step = 128u * 8u * (unsigned) ipitch;
#pragma unroll
    for (int i = 0; i < 16; i++) q[i] = *reinterpret_cast<const float4 *>(x + (o + step * i));      // (< 2^32 bytes inside a transform)
  };
  auto rows_to_lds = [&]() {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int row = rr + 64 * i;
      const float sg = (PASS == 1 && inverse) ? -1.f : 1.f;      // (inverse = conj o FFT o conj: pass 1 conjugates its input)
      tile[row * P + cc] = cmk(q[i].x, sg * q[i].y);
      tile[row * P + cc + 1] = cmk(q[i].z, sg * q[i].w);
    }
  };
  auto column_fft = [&](f1c (&v)[16], f1c *cu) {
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = cu[(64 * r + lane) * P];
    w1024::forward<P>(v, cu, lane, tw1, tw2, wave_fence);
    wave_fence();
  };
  auto pull = [&]() -> int {
    const unsigned v = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
    return v < (unsigned) ntiles ? (int) v : -1;
  };
  int id, nid;
  if (DYN) {
    if (t == 0) {
      const int a = pull();
      lnext[0] = a;
      lnext[1] = a >= 0 ? pull() : -1;
    }
    __syncthreads();
    id = lnext[0];
    nid = lnext[1];
    __syncthreads();
  } else {
    __syncthreads();
    id = blockIdx.x < (unsigned) ntiles ? (int) blockIdx.x : -1;
    nid = id >= 0 && id + (int) gridDim.x < ntiles ? id + (int) gridDim.x : -1;
  }
  if (id < 0) return;
  if (PASS == 2) {
    // Pass 2 (natural store) by decimation in FREQUENCY: s(j) = x(j) + x(j + 1024) gives the even bins, d(j) = (x(j) - x(j + 1024))
    // W_2048^j the odd ones.  The butterfly is done by the threads that loaded the rows (top and bottom half tile in registers), the
    // even bins leave right after the first pair of transforms -- so each of the tile's two transform phases has a half tile of
    // stores and a half tile of loads of the NEXT tile in flight beside it (the decimation-in-time form above has all 256 KiB of
    // stores and a half tile of loads on the first phase and waits for them there).
    float4 qb[16];
    auto issue_dif = [&](int id_, int h, float4 (&dst)[16]) {
      const int b = id_ / tpt, ct = id_ - b * tpt;
      const char *x = reinterpret_cast<const char *>(in + (size_t) b * tstride_i + (size_t) ct * 16);
      const unsigned o = opaque(((unsigned) (1024 * h + rr) * (unsigned) ipitch + (unsigned) cc) * 8u), step = 64u * 8u * (unsigned) ipitch;
#pragma unroll
      for (int i = 0; i < 16; i++) dst[i] = *reinterpret_cast<const float4 *>(x + (o + step * i));
    };
    auto store_dif = [&](int id_, int parity) {
      const int b = id_ / tpt, ct = id_ - b * tpt;
      char *yb = reinterpret_cast<char *>(out + (size_t) b * tstride_o + (size_t) ct * 16);
      const unsigned o = opaque(((unsigned) (2 * rr + parity) * (unsigned) opitch + (unsigned) cc) * 8u), step = 128u * 8u * (unsigned) opitch;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int row = rr + 64 * i;
        const cpx a = tile[row * P + cc], bb = tile[row * P + cc + 1];
        *reinterpret_cast<float4 *>(yb + (o + step * i)) = make_float4(a.x, a.y, bb.x, bb.y);
      }
    };
    const cpx wr = W2K[rr];                                  // W_2048^rr; row j = rr + 64 i: W_2048^j = wr * W2K[80 + i]
    issue_dif(id, 0, q);
    issue_dif(id, 1, qb);
    while (id >= 0) {
      int pulled = -1;
      if (DYN && t == 0 && nid >= 0) pulled = pull();
      // butterfly: s -> the image, d stays in qb
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int row = rr + 64 * i;
        const cpx w = cmul(wr, W2K[80 + i]);
        const cpx t0 = cmk(q[i].x, q[i].y), t1 = cmk(q[i].z, q[i].w), b0 = cmk(qb[i].x, qb[i].y), b1 = cmk(qb[i].z, qb[i].w);
        tile[row * P + cc] = cadd(t0, b0);
        tile[row * P + cc + 1] = cadd(t1, b1);
        const cpx d0 = cmul(csub(t0, b0), w), d1 = cmul(csub(t1, b1), w);
        qb[i] = make_float4(d0.x, d0.y, d1.x, d1.y);
      }
      if (nid >= 0) issue_dif(nid, 0, q);                     // the next tile's top half: in flight through the first transforms
      lds_barrier();
      // (one column at a time: its spectrum goes back into its own column of the image before the next one is read; both
      // side by side -- two instruction streams for a SIMD that has only two waves -- spilled 37 registers and measured slower)
#pragma unroll
      for (int u = 0; u < 2; u++) {
        f1c V[16];
        column_fft(V, colc + u);
#pragma unroll
        for (int r = 0; r < 16; r++) colc[(k0 + 64 * (r >> 2) + 256 * (r & 3)) * P + u] = f1out(V[r], scale, inverse);
        wave_fence();
      }
      lds_barrier();
      store_dif(id, 0);                                       // bins 2 k
      lds_barrier();
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int row = rr + 64 * i;
        tile[row * P + cc] = cmk(qb[i].x, qb[i].y);
        tile[row * P + cc + 1] = cmk(qb[i].z, qb[i].w);
      }
      if (nid >= 0) issue_dif(nid, 1, qb);                    // the next tile's bottom half: in flight through the second transforms
      lds_barrier();
#pragma unroll
      for (int u = 0; u < 2; u++) {
        f1c V[16];
        column_fft(V, colc + u);
#pragma unroll
        for (int r = 0; r < 16; r++) colc[(k0 + 64 * (r >> 2) + 256 * (r & 3)) * P + u] = f1out(V[r], scale, inverse);
        wave_fence();
      }
      lds_barrier();
      store_dif(id, 1);                                       // bins 2 k + 1
      if (DYN) {
        if (t == 0) lnext[0] = pulled;
        lds_barrier();
        id = nid;
        nid = id >= 0 ? lnext[0] : -1;
        lds_barrier();
      } else {
        lds_barrier();
        id = nid;
        nid = id >= 0 && id + (int) gridDim.x < ntiles ? id + (int) gridDim.x : -1;
      }
    }
    return;
  }
  // Pass 1 (transposed store) by decimation in TIME: the even rows give E, the odd rows O, X(k) = E(k) + W_2048^k O(k) and
  // X(k + 1024) = E(k) - W_2048^k O(k) come out as two contiguous halves of the column's spectrum -- what a row of the
  // transposed intermediate wants (decimation in frequency would interleave them).
  issue(id, 0);
  while (id >= 0) {
    int pulled = -1;
    if (DYN && t == 0 && nid >= 0) pulled = pull();         // the tile after next: its latency hides under this tile
    rows_to_lds();                                          // even rows
    issue(id, 1);                                           // in flight through the first transforms
    lds_barrier();
    f1c E[2][16], H[2][16];
    column_fft(E[0], colc);
    column_fft(E[1], colc + 1);
    lds_barrier();                                          // every wave is done with its columns of the image
    rows_to_lds();                                          // odd rows
    if (nid >= 0) issue(nid, 0);                            // the next tile's even rows: in flight through the second transforms
    lds_barrier();
    const int ct_cur = id - (id / tpt) * tpt;
#pragma unroll
    for (int u = 0; u < 2; u++) {
      column_fft(H[u], colc + u);
      // X(k) = E + w O, X(k + 1024) = E - w O, w = W_2048^k = wl * W_2048^(64 a + 256 b), k = k0 + 64 a + 256 b, a = r >> 2, b = r & 3
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const f1c tw = f1mul(H[u][r], cmul(wl, W2K[64 + r]));     // (W2K[64 + r] = W_2048^(64 a + 256 b): wave-uniform)
        H[u][r] = w1024::csub(E[u][r], tw);
        E[u][r] = w1024::cadd(E[u][r], tw);
      }
      {
        // four-step twiddle W_n^(c k), c = this column, k = k0 + 64 a + 256 b (+ 1024): TA[c][lane] = W_n^(c k0), TD[c][r] =
        // W_n^(c (64 a + 256 b)), TD[c][16] = W_n^(1024 c); then row c of the transposed intermediate, 2048 contiguous values:
        // the spectrum leaves the wave through its own LDS column so that every lane stores 16 B (two adjacent bins)
        const int c = ct_cur * 16 + 2 * wv + u;
        const cpx ta = TA[(size_t) c * 64 + lane], th = TD[(size_t) c * 17 + 16];
        f1c *cu = colc + u;
        const cpx *cur = tile + 2 * wv + u;
        cpx *zrow = out + (size_t) (id / tpt) * tstride_o + (size_t) c * opitch;
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
          for (int r = 0; r < 16; r++) {
            cpx w = cmul(ta, TD[(size_t) c * 17 + r]);
            if (half) w = cmul(w, th);
            cu[(k0 + 64 * (r >> 2) + 256 * (r & 3)) * P] = f1mul(half ? H[u][r] : E[u][r], w);
          }
          wave_fence();
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const cpx e0 = cur[(128 * j + 2 * lane) * P], e1 = cur[(128 * j + 2 * lane + 1) * P];
            *reinterpret_cast<float4 *>(zrow + 1024 * half + 128 * j + 2 * lane) = make_float4(e0.x, e0.y, e1.x, e1.y);
          }
          wave_fence();
        }
      }
    }
    if (DYN) {
      if (t == 0) lnext[0] = pulled;
      lds_barrier();
      id = nid;
      nid = id >= 0 ? lnext[0] : -1;
      lds_barrier();                                        // (lnext[0] is rewritten at the next tile's end)
    } else {
      lds_barrier();
      id = nid;
      nid = id >= 0 && id + (int) gridDim.x < ntiles ? id + (int) gridDim.x : -1;
    }
  }
}

// ---- helpers for the non power-of-two paths ------------------------------------------------
// even split: tmp[b][0..h) = x[b][0::2], tmp[b][h..n) = x[b][1::2]
__global__ void fft_split_eo_kernel(const cpx *__restrict__ x, cpx *__restrict__ tmp, int n, int64_t total)
{
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t b = i / n;
  const int k = (int) (i - b * n), h = n >> 1;
  tmp[i] = x[b * n + (k < h ? 2 * k : 2 * (k - h) + 1)];
}
// y[i] = (E[i % h] + rot[i] * O[i % h]) / sqrt(2)      (fourier.cc:451-462)
__global__ void fft_combine_eo_kernel(const cpx *__restrict__ eo, cpx *__restrict__ y, const cpx *__restrict__ rot,
                                      int n, int inverse, int64_t total)
{
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t b = i / n;
  const int k = (int) (i - b * n), h = n >> 1;
  const cpx E = eo[b * n + (k % h)], O = eo[b * n + h + (k % h)];
  cpx r = rot[k];
  if (inverse) r = cconj(r);
  y[i] = cscale(cadd(E, cmul(r, O)), 0.70710678118654752f);
}
// Bluestein: xp[b][i] = x[b][i] * chirp[n-1+i] (i < n), 0 up to n2
__global__ void czt_pre_kernel(const cpx *__restrict__ x, cpx *__restrict__ xp, const cpx *__restrict__ chirp,
                               int n, int n2, int64_t total)
{
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t b = i / n2;
  const int k = (int) (i - b * n2);
  xp[i] = k < n ? cmul(x[b * n + k], chirp[n - 1 + k]) : cmk(0.f, 0.f);
}
__global__ void czt_mul_kernel(cpx *__restrict__ a, const cpx *__restrict__ xc, int n2, int64_t total)
{
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  a[i] = cmul(a[i], xc[i % n2]);
}
// y[b][k] = y2[b][n-1+k'] * chirp[n-1+k'] * g, with k' = k (forward) or (n-k)%n (inverse, tfr2itfr)
__global__ void czt_post_kernel(const cpx *__restrict__ y2, cpx *__restrict__ y, const cpx *__restrict__ chirp,
                                int n, int n2, float g, int inverse, int64_t total)
{
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t b = i / n;
  const int k = (int) (i - b * n);
  const int kk = inverse ? (n - k) % n : k;
  y[i] = cscale(cmul(y2[b * n2 + n - 1 + kk], chirp[n - 1 + kk]), g);
}
// ---- odd n = 5 .. 8191: Bluestein in ONE kernel ------------------------------------------------------
// Same arithmetic as the five-kernel path above (pre-multiply by the reference's float32 chirp,
// zero-pad to n2 = pp2(2n-1) <= 16384, FFT, times the transformed conjugate chirp, inverse FFT,
// post-multiply), but n2/16 threads keep a whole transform in registers + LDS (stockham16.hpp):
// 16 B of HBM traffic per point instead of ~150.  The spectrum leaves the forward transform in
// the register layout the next one reads, as in ols_long.hip.  max(256, n2/16) threads per
// workgroup = several transforms when n2 < 4096.
// The same kernel is pass 1 of the mixed-radix plan for n = m * P with a large odd part m:
// transform tr = (b, r) then reads the decimated sequence x[b][r + P i] (in_stride = P) and
// writes Z[b][r][k1] * W_n^(r k1) (Wn != nullptr).  For the inverse of that plan the odd part is
// inverted like the reference does it (forward transform + index reversal, so the float32-chirp
// rounding enters identically) and conjugated (conj_out): pass 2 then runs forward and conjugates
// at its store, i.e. the power-of-two part uses conj(FFT(conj .)).
template <int R0>
__global__ __launch_bounds__(1024) void fft_bluestein_kernel(const cpx *__restrict__ in, cpx *__restrict__ out,
                                                             const cpx *__restrict__ chirp, const cpx *__restrict__ xc,
                                                             const cpx *__restrict__ TW, int n, int n2, int tpt, int reverse,
                                                             float g, int P, const cpx *__restrict__ Wn, int conj_out,
                                                             int64_t ntr, int fuse, float s2)
{
  extern __shared__ __attribute__((aligned(16))) char blu_raw[];
  const int t = threadIdx.x;
  const int tl = t / tpt, j = t - tl * tpt, T = blockDim.x / tpt;
  const int64_t tr = (int64_t) blockIdx.x * T + tl;
  const bool live = tr < ntr;
  cpx *lds = reinterpret_cast<cpx *>(blu_raw) + (size_t) tl * (n2 + (n2 >> 4));
  const int64_t bb = tr / P;
  const int r = (int) (tr - bb * P);
  const cpx *x = in + (size_t) bb * n * P + r;
  cpx *y = out + (size_t) tr * n;
  cpx v[16];
#pragma unroll
  for (int m = 0; m < 16; m++) {
    const int pos = j + m * tpt;
    v[m] = cmk(0.f, 0.f);
    if (live && pos < n) {
      v[m] = cmul(x[(size_t) pos * P], chirp[n - 1 + pos]);
    }
  }
  auto sync = []() { __syncthreads(); };
  s16::transform<R0>(v, lds, TW, n2, j, tpt, sync);          // sqrt(n2) * unitary FFT
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const cpx p2 = cmul(v[q], xc[j + q * tpt]);
    v[q] = cmk(p2.x, -p2.y);
  }
  sync();
  s16::transform<R0>(v, lds, TW, n2, j, tpt, sync);          // conj of n2 * (unitary inverse of the product)
  if (!fuse && !live) return;
  // fuse = P in {2, 4, 8, 16} (the whole mixed-radix plan in this kernel, round 3): the workgroup holds whole groups of P residues; their
  // Z[r][k1] stay in the transforms' LDS images and pass 2 -- a P-point DFT over r per column k1, fft_smallcols_kernel's -- follows
  // a barrier later: 16 B of HBM traffic per point instead of 48
  if (fuse) sync();                                            // (the image is still being read by the transform's last pass)
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const int pos = j + q * tpt, kk = pos - (n - 1);
    if (live && kk >= 0 && kk < n) {
      const int k = reverse ? (kk == 0 ? 0 : n - kk) : kk;    // tfr2itfr: X^-1[k] = X[(n - k) % n]
      cpx o = cscale(cmul(cmk(v[q].x, -v[q].y), chirp[pos]), g);
      if (conj_out) o.y = -o.y;
      if (Wn) o = cmul(o, Wn[(size_t) r * k]);
      if (fuse) lds[k] = o;
      else y[k] = o;
    }
  }
  if (!fuse) return;
  sync();
  const int G = T / P;
  const size_t pitch = (size_t) n2 + (n2 >> 4);
  for (int idx = t; idx < G * n; idx += (int) blockDim.x) {
    const int gq = idx / n, k1 = idx - gq * n;
    const int64_t tr0 = (int64_t) blockIdx.x * T + (int64_t) gq * P;
    if (tr0 >= ntr) continue;
    const cpx *zb = reinterpret_cast<const cpx *>(blu_raw) + (size_t) gq * P * pitch + k1;
    cpx *yb = out + (size_t) (tr0 / P) * n * P + k1;
    if (P == 16) {
      cpx e[16];
#pragma unroll
      for (int rr = 0; rr < 16; rr++) e[rr] = zb[(size_t) rr * pitch];
      w1024::dft16<false>(e);
#pragma unroll
      for (int k2 = 0; k2 < 16; k2++) yb[(size_t) k2 * n] = cmk(e[k2].x * s2, conj_out ? -e[k2].y * s2 : e[k2].y * s2);
      continue;
    }
    cpx e[8];
#pragma unroll
    for (int rr = 0; rr < 8; rr++) e[rr] = rr < P ? zb[(size_t) rr * pitch] : cmk(0.f, 0.f);
    if (P == 2) {
      s16::dft2(e[0], e[1]);
    } else if (P == 4) {
      w1024::dft4<false>(e[0], e[1], e[2], e[3]);
    } else {
      s16::dft8(e);
    }
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++)
      if (k2 < P) yb[(size_t) k2 * n] = cmk(e[k2].x * s2, conj_out ? -e[k2].y * s2 : e[k2].y * s2);
  }
}

// ---- the same Bluestein for transforms that fit a WAVE (n2 <= 1024, i.e. odd n <= 512: the odd parts of 1000, 3000, 6000 ...) --------
// fft_bluestein_kernel spends 72 % of its wave cycles waiting (PMC, n = 1000): every thread fetches its 16 chirp and 16 spectrum
// values per transform from L2 next to its 16 samples, four waves meet at five barriers per transform, and nothing is in
// flight while a wave computes.  Here a workgroup is PERSISTENT: chirp, transformed chirp and twiddles are staged in LDS once,
// the tpt <= 64 threads of a transform synchronise inside their wave only (LDS is in order per wave), and the samples of
// the next slot are requested before the current one is transformed.  Same arithmetic, same order of operations.
// FRAME (psd_welch, ola.hip): transform (b, r) reads segment b of a stream -- x[b bstride + r + P i] times the window
// w[r + P i] -- and the kernel stores |X|^2 only.
struct BluArgs {
  const cpx *in;
  cpx *out;
  float *pw;
  const cpx *chirp, *xc, *TW, *Wn;
  const float *win;
  int n, n2, tpt, reverse, logP, conj_out, fuse;
  float g, s2;
  int64_t ntr, bstride;
};
// TPT = threads per transform = n2 / 16, a template parameter: the LDS offsets of the exchanges are then immediates (with a run-time tpt
// half of the kernel's vector instructions were index arithmetic, or -- hoisted -- 256 registers of addresses)
constexpr int blu_r0(int tpt) { return tpt == 1 || tpt == 16 || tpt == 256 ? 16 : tpt == 2 || tpt == 32 ? 2 : tpt == 4 || tpt == 64 ? 4 : 8; }
// Arithmetic: PACKED (w1024::v2f -- a complex number in a 64-bit register pair, one or two VOP3P instructions per primitive): the
// kernel is bound by its vector instructions (PMC, n = 125: 72 % VALU busy in the scalar flavour), unlike the memory-bound
// kernels where the packed flavour changed nothing.
using bwc = w1024::v2f;
__device__ __forceinline__ bwc bw_conj_mul(bwc a, bwc w)      // conj(a w) = (a.x w.x - a.y w.y, -a.x w.y - a.y w.x)
{
  bwc t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));                      // (a.x w.x, -a.x w.y)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
// pass 0 of a transform whose inputs v[8 .. 15] are ZERO (the padding of the chirp product: n <= n2 / 2 always) -- the first
// butterfly stage of every radix loses half of its additions; v[8 .. 15] are not read
template <int R0, typename C> __device__ __forceinline__ void bw_pass0_half(C (&v)[16])
{
  using namespace w1024;
  if (R0 == 16) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
    for (int q = 0; q < 4; q++) {                          // dft4(a, b, 0, 0) = (a + b, a - i b, a - b, a + i b)
      const C a = v[q], b = v[4 + q];
      v[q] = cadd(a, b);
      v[4 + q] = caddrot<false>(a, b);
      v[8 + q] = csub(a, b);
      v[12 + q] = csubrot<false>(a, b);
    }
    v[5] = cmul(v[5], Make<C>::of(C1, -S1));
    v[6] = cmul(v[6], Make<C>::of(R2, -R2));
    v[7] = cmul(v[7], Make<C>::of(S1, -C1));
    v[9] = cmul(v[9], Make<C>::of(R2, -R2));
    v[11] = cmul(v[11], Make<C>::of(-R2, -R2));
    v[13] = cmul(v[13], Make<C>::of(S1, -C1));
    v[14] = cmul(v[14], Make<C>::of(-R2, -R2));
    v[15] = cmul(v[15], Make<C>::of(-C1, S1));
    dft4<false>(v[0], v[1], v[2], v[3]);
    dft4<false>(v[4], v[5], v[6], v[7]);
    dft4_crot<false>(v[8], v[9], v[10], v[11]);
    dft4<false>(v[12], v[13], v[14], v[15]);
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
      for (int y = x + 1; y < 4; y++) {
        const C t = v[4 * x + y];
        v[4 * x + y] = v[4 * y + x];
        v[4 * y + x] = t;
      }
  } else if (R0 == 8) {
    constexpr float R2 = 0.70710678118654752f;
#pragma unroll
    for (int i = 0; i < 2; i++) {                          // dft8 of (e0, e1, e2, e3, 0, 0, 0, 0), e[q] = v[i + 2 q]
      const C e0 = v[i], e1 = v[i + 2], e2 = v[i + 4], e3 = v[i + 6];
      const C a0 = cadd(e0, e2), a1 = caddrot<false>(e0, e2), a2 = csub(e0, e2), a3 = csubrot<false>(e0, e2);
      const C b0 = cadd(e1, e3), b1 = caddrot<false>(e1, e3), b2 = csub(e1, e3), b3 = csubrot<false>(e1, e3);
      const C o1 = cmul(b1, Make<C>::of(R2, -R2)), o2 = rot90<false>(b2), o3 = cmul(b3, Make<C>::of(-R2, -R2));
      v[i] = cadd(a0, b0); v[i + 8] = csub(a0, b0);
      v[i + 2] = cadd(a1, o1); v[i + 10] = csub(a1, o1);
      v[i + 4] = cadd(a2, o2); v[i + 12] = csub(a2, o2);
      v[i + 6] = cadd(a3, o3); v[i + 14] = csub(a3, o3);
    }
  } else if (R0 == 4) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const C a = v[i], b = v[i + 4];
      v[i] = cadd(a, b);
      v[i + 4] = caddrot<false>(a, b);
      v[i + 8] = csub(a, b);
      v[i + 12] = csubrot<false>(a, b);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) v[i + 8] = v[i];
  }
}
// s16::transform with (HALF) the pruned first pass and the twiddles of the radix-16 passes READ instead of generated: W[i][q] =
// W_n^(q i), i < n / 16, in rows of 17 (s16::twiddle_powers spends 11 complex products per pass on the powers of one table value)
constexpr int BW_TWROW = 17, BW_TAB_MAX_TPT = 32, BW_MAX_TPT = 128;
// (TAB = false: W[i] = W_n^i, i < n / 16, and generated powers -- where the rows would cost a resident workgroup: n = 1024)
template <int R0, bool HALF, bool TAB, typename C, typename SYNC>
__device__ __forceinline__ void bw_transform(C (&v)[16], C *s, const C *__restrict__ W, int n, int j, int tpt, SYNC sync)
{
  if (HALF) bw_pass0_half<R0>(v);
  else s16::pass0<R0>(v);
  if (n == R0) return;
  s16::pass0_store<R0>(s, v, j, tpt);
  for (int Ns = R0;; Ns <<= 4) {
    sync();
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = s[s16::pad(j + q * tpt)];
    const int k = j & (Ns - 1);
    if (TAB) {
      const C *w = W + k * (tpt / Ns) * BW_TWROW;
#pragma unroll
      for (int q = 1; q < 16; q++) v[q] = w1024::cmul(v[q], w[q]);
    } else {
      s16::twiddle_powers(v, W[k * (tpt / Ns)]);
    }
    w1024::dft16<false>(v);
    if (Ns * 16 == n) return;                                // natural order in registers
    const int base = (j - k) * 16 + k;
    sync();
#pragma unroll
    for (int q = 0; q < 16; q++) s[s16::pad(base + q * Ns)] = v[q];
  }
}
// a transform of up to 64 threads lives in one wave (LDS is in order per wave: a fence); 128 threads: the workgroup's barrier
template <int TPT> __device__ __forceinline__ void bw_sync()
{
  if (TPT <= 64) wave_fence();
  else lds_barrier();
}
template <int TPT, bool FRAME>
__global__ __launch_bounds__(512, 3) void fft_blu_wave_kernel(const BluArgs A)
{
  extern __shared__ __attribute__((aligned(16))) char bw_raw[];
  constexpr int tpt = TPT, n2 = 16 * TPT, R0 = blu_r0(TPT);
  constexpr bool TAB = TPT <= BW_TAB_MAX_TPT;

  constexpr int TWN = TAB ? TPT * BW_TWROW + (TPT & 1) : TPT;
  const int n = A.n, P = 1 << A.logP;
  const int t = threadIdx.x, NT = blockDim.x, T = NT / tpt;
  const int tl = t / tpt, j0 = t - tl * tpt;
  // LDS: chirp[n - 1 + i] (i < n) | transformed chirp (n2) | twiddle rows (17 n2 / 16) | post-multiplier table (R n) | T images
  const int R = A.fuse ? P : 1;
  bwc *chL = reinterpret_cast<bwc *>(bw_raw);
  bwc *xcL = chL + n + 1;
  bwc *twL = xcL + n2;
  bwc *poL = twL + TWN;
  bwc *img0 = poL + R * n + 1;
  constexpr int pitch = n2 + (n2 >> 4);
  const bwc *gch = reinterpret_cast<const bwc *>(A.chirp), *gxc = reinterpret_cast<const bwc *>(A.xc), *gtw = reinterpret_cast<const bwc *>(A.TW);
  const bwc *gwn = reinterpret_cast<const bwc *>(A.Wn);
  for (int i = t; i < n; i += NT) chL[i] = gch[n - 1 + i];
  for (int i = t; i < n2; i += NT) xcL[i] = gxc[i];
  for (int i = t; i < TWN; i += NT) twL[i] = gtw[i];
  // post-multiplier of output kk of residue r: chirp (conjugated for the inverse of a mixed plan, whose odd part is conjugated
  // after its forward transform: conj(conj(v) c) = v conj(c)) times g times, when pass 2 is fused (the residue of a thread is
  // then the same in every slot), the four-step twiddle W_N^(r k) of the bin k the output lands on
  for (int r = 0; r < R; r++)
    for (int kk = t; kk < n; kk += NT) {
      bwc c = gch[n - 1 + kk] * A.g;
      if (A.conj_out) c.y = -c.y;
      const int k = A.reverse ? (kk == 0 ? 0 : n - kk) : kk;           // tfr2itfr: X^-1[k] = X[(n - k) % n]
      if (A.fuse && gwn) c = w1024::cmul(c, gwn[r * k]);
      poL[r * n + kk] = c;
    }
  // FRAME: the powers are summed over the slots of the workgroup (an output bin always belongs to the same thread) and leave as ONE row
  // of N = n P floats per workgroup: partial sums for the caller's row reduction
  float *accL = reinterpret_cast<float *>(img0 + T * pitch);         // fused: ONE row of N = n P sums (a thread owns its bins k1 in every group)
  float accR[FRAME ? 12 : 1];                                          // not fused: a thread's twelve outputs are the same bins in every slot
  if (FRAME) {
    if (A.fuse)
      for (int i = t; i < n * P; i += NT) accL[i] = 0.f;
#pragma unroll
    for (int i = 0; i < (FRAME ? 12 : 1); i++) accR[i] = 0.f;
  }
  __syncthreads();
  const int64_t nslots = (A.ntr + T - 1) / T;
  // the padding of the chirp product is at least half of n2: registers 8 .. 15 of a thread are zeros, never loaded.  Loads are
  // unconditional: positions past n (and transforms past the last one) are clamped onto valid elements and masked at the product
  bwc u[8];
  float uw[FRAME ? 8 : 1];
  auto fetch = [&](int64_t slot_) {
    int j = j0;
    asm volatile("" : "+v"(j));                               // (opaque: keeps sixteen hoisted addresses and predicates per stage out of the loop's registers)
    const int64_t tr = min(slot_ * T + tl, A.ntr - 1);
    const int64_t bb = tr >> A.logP;
    const int r = (int) (tr - (bb << A.logP));
    const bwc *x = reinterpret_cast<const bwc *>(A.in + bb * A.bstride + r);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int pos = min(j + m * tpt, n - 1);
      u[m] = x[(size_t) ((unsigned) pos << A.logP)];
      if (FRAME) uw[m] = A.win[r + (pos << A.logP)];
    }
  };
  int64_t slot = blockIdx.x;
  if (slot < nslots) fetch(slot);
  const float rn = 1.0f / (float) n;
  const float sgn = A.conj_out ? 1.f : -1.f;                   // the post-multiplier takes conj(v), or v for the conjugated output
  const bwc *poT = poL + (A.fuse ? (tl & (P - 1)) * n : 0);
  for (; slot < nslots; slot += gridDim.x) {
    int j = j0;
    asm volatile("" : "+v"(j));
    j &= tpt - 1;                                                      // (the range is what turns the padded LDS indices into immediates)
    const int64_t tr = slot * T + tl;
    const bool live = tr < A.ntr;
    bwc *img = img0 + tl * pitch;
    bwc v[16];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int pos = j + m * tpt;
      bwc a = u[m];
      if (FRAME) a = a * uw[m];
      const bwc c = w1024::cmul(a, chL[pos]);                          // (past n: a value of the next table, masked)
      v[m] = pos < n ? c : (bwc){0.f, 0.f};
    }
    if (slot + gridDim.x < nslots) fetch(slot + gridDim.x);
    bw_transform<R0, true, TAB>(v, img, twL, n2, j, tpt, bw_sync<TPT>);       // sqrt(n2) * unitary FFT
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = bw_conj_mul(v[q], xcL[j + q * tpt]);
    bw_sync<TPT>();
    bw_transform<R0, false, TAB>(v, img, twL, n2, j, tpt, bw_sync<TPT>);     // conj of n2 * (unitary inverse of the product)
    // outputs n - 1 .. 2 n - 2 of the convolution: positions below 4 tpt never qualify (n - 1 >= n2 / 4).  o = conj(v) chirp g [W]
    bwc o[12];
    int kc[12];
    bool ok[12];
#pragma unroll
    for (int q = 4; q < 16; q++) {
      const int kk = j + q * tpt - (n - 1);
      ok[q - 4] = live && (unsigned) kk < (unsigned) n;
      kc[q - 4] = ok[q - 4] ? kk : 0;
    }
#pragma unroll
    for (int q = 4; q < 16; q++) {
      bwc w = v[q];
      w.y *= sgn;
      o[q - 4] = w1024::cmul(w, poT[kc[q - 4]]);
    }
    if (!A.fuse) {
      if (gwn) {                                                       // pass 1 of a two-kernel mixed plan: the residue changes with the slot
        const int r = (int) (tr & (P - 1));
        bwc wn[12];
#pragma unroll
        for (int i = 0; i < 12; i++) {
          const int k = A.reverse ? (kc[i] == 0 ? 0 : n - kc[i]) : kc[i];
          wn[i] = gwn[r * k];
        }
#pragma unroll
        for (int i = 0; i < 12; i++) o[i] = w1024::cmul(o[i], wn[i]);
      }
      bwc *y = reinterpret_cast<bwc *>(A.out) + (size_t) tr * n;
#pragma unroll
      for (int i = 0; i < 12; i++) {
        const int k = A.reverse ? (kc[i] == 0 ? 0 : n - kc[i]) : kc[i];
        if (FRAME) {
          if (ok[i]) accR[i] += o[i].x * o[i].x + o[i].y * o[i].y;
        } else if (ok[i]) {
          y[k] = o[i];
        }
      }
      bw_sync<TPT>();                                                         // (the next slot's pass 0 rewrites the image)
      continue;
    }
    // fused pass 2: the outputs go to the image at their convolution index kk (slot n: a dump nobody reads); the P-point column DFT
    // over the residues of every group of this slot (fft_bluestein_kernel's pass 2) fetches bin k1 from kk = (n - k1) % n when reversed
    bw_sync<TPT>();                                                           // (the image is still being read by the transform's last pass)
#pragma unroll
    for (int i = 0; i < 12; i++) img[ok[i] ? kc[i] : n] = o[i];
    lds_barrier();
    const int G = T >> A.logP;
    const bwc sv = {A.s2, A.conj_out ? -A.s2 : A.s2};
    // (FRAME: a thread takes bin k1 of every group in turn -- idx = gq n + k1 with k1 = t, t + NT ... -- so that the sums of a bin have one owner)
    for (int idx0 = t; idx0 < (FRAME ? n : G * n); idx0 += NT)
    for (int idx = idx0; idx < G * n; idx += (FRAME ? n : G * n)) {
      const int gq = (int) (((float) idx + 0.5f) * rn), k1 = idx - gq * n;       // (idx < 2^15, n <= 511: the quotient is exact)
      const int64_t tr0 = slot * T + ((int64_t) gq << A.logP);
      if (tr0 >= A.ntr) continue;
      const int kk1 = A.reverse ? (k1 == 0 ? 0 : n - k1) : k1;
      const bwc *zb = img0 + (gq << A.logP) * pitch + kk1;
      const size_t ob = (size_t) (tr0 >> A.logP) * n * P + k1;
      bwc e[16];
      if (P == 16) {
#pragma unroll
        for (int rr = 0; rr < 16; rr++) e[rr] = zb[rr * pitch];
        w1024::dft16<false>(e);
      } else {
#pragma unroll
        for (int rr = 0; rr < 8; rr++) e[rr] = rr < P ? zb[rr * pitch] : (bwc){0.f, 0.f};
        if (P == 2) {
          s16::dft2(e[0], e[1]);
        } else if (P == 4) {
          w1024::dft4<false>(e[0], e[1], e[2], e[3]);
        } else {
          bwc e8[8];
#pragma unroll
          for (int rr = 0; rr < 8; rr++) e8[rr] = e[rr];
          s16::dft8(e8);
#pragma unroll
          for (int rr = 0; rr < 8; rr++) e[rr] = e8[rr];
        }
      }
      bwc *yo = reinterpret_cast<bwc *>(A.out) + ob;
#pragma unroll
      for (int k2 = 0; k2 < 16; k2++)
        if (k2 < P) {
          const bwc oo = e[k2] * sv;
          if (FRAME) accL[k2 * n + k1] += oo.x * oo.x + oo.y * oo.y;
          else yo[(size_t) k2 * n] = oo;
        }
    }
    lds_barrier();                                                     // the next slot rewrites the images
  }
  if (FRAME) {
    __syncthreads();
    if (A.fuse) {
      float *row = A.pw + (size_t) blockIdx.x * n * P;
      for (int i = t; i < n * P; i += NT) row[i] = accL[i];
    } else {
      // the T transforms' register sums meet in LDS (the images are free now) and are added in order
      float *sub = reinterpret_cast<float *>(img0);
      for (int i = t; i < T * n; i += NT) sub[i] = 0.f;
      __syncthreads();
#pragma unroll
      for (int q = 4; q < 16; q++) {
        const int kk = j0 + q * tpt - (n - 1);
        if ((unsigned) kk < (unsigned) n) sub[tl * n + kk] = accR[q - 4];
      }
      __syncthreads();
      float *row = A.pw + (size_t) blockIdx.x * n;
      for (int i = t; i < n; i += NT) {
        float sacc = sub[i];
        for (int g = 1; g < T; g++) sacc += sub[g * n + i];
        row[i] = sacc;
      }
    }
  }
}

// ---- n = m * P, m odd <= 15, P = 2^p >= 16, n <= 16384 (48, 96 ... 1536, 3072, 5120, 7168, 12288, 15360): ONE kernel, the power of two on
// the radix-16 Stockham engine (round 3).  Decimation in time over the odd factor: the m sub-sequences x[m i + r] are P-point
// transforms F_r -- s16::transform, compile-time radix-16 passes, one LDS image each, all m of a transform in one workgroup --
// and X[k + P q] = sum_r W_m^(r q) (W_n^(r k) F_r[k]) is an m-point DFT over r per column k, evaluated directly a barrier later
// from the images (the twiddled F_r written back in natural order).  fft_mr_kernel serves the same sizes with one autosort pass per
// factor and run-time index arithmetic (~180 VALU instructions per point: 28-36 % of 8 TB/s at 1536 ... 15360); here the
// power-of-two part costs what it costs the power-of-two plans.  Same accuracy class (direct m-point DFT against the
// reference's float32-chirp Bluestein of m <= 31 points: < 5e-6).  Inverse = conj o forward o conj.
template <int R0, int MMAX>
__global__ __launch_bounds__(1024) void fft_oddpow2_kernel(const cpx *__restrict__ in, cpx *__restrict__ out, const cpx *__restrict__ TW,
                                                           const cpx *__restrict__ Wm, const cpx *__restrict__ Wn, int m, int P, int tpt,
                                                           int inverse, float scale, int batch)
{
  extern __shared__ __attribute__((aligned(16))) char op_raw[];
  const int t = threadIdx.x, per = m * tpt, T = (int) blockDim.x / per;       // threads per transform, transforms per workgroup
  const int tl = t / per, u = t - tl * per, r = u / tpt, j = u - r * tpt;
  const int n = m * P;
  const size_t pitch = (size_t) P + (P >> 4);
  cpx *img0 = reinterpret_cast<cpx *>(op_raw) + (size_t) tl * m * pitch;     // the m images of this thread's transform
  cpx *img = img0 + (size_t) r * pitch;
  const int64_t b = (int64_t) blockIdx.x * T + tl;
  const bool live = tl < T && b < batch;
  const cpx *x = in + (size_t) b * n;
  cpx v[16];
#pragma unroll
  for (int q = 0; q < 16; q++) {
    v[q] = cmk(0.f, 0.f);
    if (live) {
      v[q] = x[(size_t) (j + q * tpt) * m + r];
      if (inverse) v[q].y = -v[q].y;
    }
  }
  auto sync = []() { __syncthreads(); };
  if (tl < T) s16::transform<R0>(v, img, TW, P, j, tpt, sync);
  else {
    // (threads past the last whole transform of the workgroup only keep the barriers company -- never the case: blockDim = T per)
  }
  sync();                                                    // the image is still being read by the transform's last pass
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const int k = j + q * tpt;
    img[k] = cmul(v[q], Wn[(size_t) r * k]);                 // W_n^(r k) F_r[k], natural order
  }
  sync();
  if (!live) return;
  cpx *y = out + (size_t) b * n;
  for (int k = u; k < P; k += per) {
    cpx e[MMAX];
#pragma unroll
    for (int i = 0; i < MMAX; i++) e[i] = i < m ? img0[(size_t) i * pitch + k] : cmk(0.f, 0.f);
    for (int qq = 0; qq < m; qq++) {
      cpx acc = e[0];
      int idx = 0;
#pragma unroll
      for (int i = 1; i < MMAX; i++) {
        idx += qq;
        if (idx >= m) idx -= m;
        if (i < m) {
          const cpx w = Wm[idx];
          acc.x = fmaf(e[i].x, w.x, fmaf(-e[i].y, w.y, acc.x));
          acc.y = fmaf(e[i].x, w.y, fmaf(e[i].y, w.x, acc.y));
        }
      }
      y[k + (size_t) P * qq] = cmk(acc.x * scale, inverse ? -acc.y * scale : acc.y * scale);
    }
  }
}

// Pass 2 of the mixed-radix plan when the power-of-two factor is only 2, 4 or 8: one thread per
// column k1 combines the PP residues, X[k1 + m k2] at [k2][k1] (natural order).
template <int PP>
__global__ __launch_bounds__(256) void fft_smallcols_kernel(const cpx *__restrict__ z, cpx *__restrict__ out, int m,
                                                            int inverse, float scale, int64_t total)
{
  const int64_t gidx = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (gidx >= total) return;
  const int64_t b = gidx / m;
  const int k1 = (int) (gidx - b * m);
  const cpx *zb = z + (size_t) b * m * PP + k1;
  cpx *yb = out + (size_t) b * m * PP + k1;
  cpx e[8];
#pragma unroll
  for (int r = 0; r < PP; r++) e[r] = zb[(size_t) r * m];
  if (PP == 2) {
    s16::dft2(e[0], e[1]);
  } else if (PP == 4) {
    w1024::dft4<false>(e[0], e[1], e[2], e[3]);
  } else {
    s16::dft8(e);
  }
#pragma unroll
  for (int k2 = 0; k2 < PP; k2++) yb[(size_t) k2 * m] = cmk(e[k2].x * scale, inverse ? -e[k2].y * scale : e[k2].y * scale);
}

// ---- real FFT (RTFRPlan::step, fourier.cc:311-354), even n ---------------------------------------
// The n real samples ARE the n/2 packed complex samples; after their FFT Xt the spectrum is
//   y(i) = r2 (Xt(i) + conj Xt(h-i)) - j2 (Xt(i) - conj Xt(h-i)) rot(i),  i = 0..h,  h = n/2
// (Xt(h) := Xt(0)), with r2 = j2/i = 0.5/sqrt(2) on top of the unitary half-size FFT, and the upper
// half is the forced conjugate symmetry csym_forçage(): y(0), y(h) real, y(n-i) = conj y(i).
__global__ void rfft_untangle_kernel(const cpx *__restrict__ Xt, cpx *__restrict__ y, const cpx *__restrict__ rot, int n,
                                     int64_t total)
{
  const int h = n >> 1;
  const int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;     // one thread per (transform, i in 0..h)
  if (g >= total) return;
  const int64_t b = g / (h + 1);
  const int i = (int) (g - b * (h + 1));
  const cpx *X = Xt + (size_t) b * h;
  cpx *Y = y + (size_t) b * n;
  const cpx X1 = X[i == h ? 0 : i], X2c = cconj(X[i > 0 ? h - i : 0]);
  const float c = 0.35355339059327373f;                                  // (float) (0.5 / sqrt(2.0))
  const cpx a = cscale(cadd(X1, X2c), c);
  const cpx d = csub(X1, X2c);
  const cpx jd = cmk(-d.y * c, d.x * c);                                 // j2 * d
  cpx v = csub(a, cmul(jd, rot[i]));
  if (i == 0 || i == h) v.y = 0.f;
  Y[i] = v;
  if (i > 0 && i < h) Y[n - i] = cconj(v);
}
__global__ void real_to_complex_kernel(const float *__restrict__ x, cpx *__restrict__ y, int64_t total)
{
  const int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (g < total) y[g] = cmk(x[g], 0.f);
}

template <typename T>
__global__ void fftshift_kernel(const T *__restrict__ x, T *__restrict__ y, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // res.tail(n - n/2) = X.head(n - n/2); res.head(n/2) = X.tail(n/2)   (fourier.hpp:232-248)
  const int h = n / 2;
  y[i] = i < h ? x[n - h + i] : x[i - h];
}

}  // namespace tsdgpu

using namespace tsdgpu;

struct tsdgpu_fft {
  int n = 0;
  enum Kind { ONE, POW2_LDS, POW2_S16, POW2_W1024, POW2_W1M, POW2_4STEP, SMOOTH, ODDPOW2, MIXED, EVEN, ODD } kind = ONE;
  // pow2
  int logn = 0;
  cpx *d_tw = nullptr;        // W_n^k, k < n/2 (LDS path) ...
  // four-step n = N1 * N2
  int N1 = 0, N2 = 0, logN1 = 0, logN2 = 0;
  cpx *d_tw1 = nullptr, *d_tw2 = nullptr, *d_thi = nullptr, *d_tlo = nullptr;
  // in-wave 1024-point FFT paths: lane twiddles (2 x 1024) and, for n = 2^20, TA [1024][64] / TD [1024][16]
  cpx *d_w1 = nullptr, *d_w2 = nullptr, *d_ta = nullptr, *d_td = nullptr, *d_w2k = nullptr;
  unsigned *d_ctr = nullptr;  // n = 2^20: work counter of the dynamic tile hand-out (fft1m_cols_kernel), never reset
  bool cols2k = false;        // four-step plan whose pass 2 (2048-point columns, N2 = 2048) runs on fft2k_cols_kernel
  bool cols2k_p1 = false;     // ... whose pass 1 (N1 = 2048) does
  bool p1k1 = false;          // n = 2^15 .. 2^19 ALSO planned as 1024 x C (calls of 2^21 points and more): pass 1 on fft1m_cols_kernel<1>
                              // (padded intermediate), pass 2 = C-point columns on fft_cols16_kernel<2> with the tables d_tw2a
  int N2a = 0, logN2a = 0;
  cpx *d_tw2a = nullptr;
  int c3 = 0;                 // three-pass plan n = 1024 x c3 x 1024 (c3 = 8, 16, 32; 0: not this plan); d_tp: W_C^(b p) [1024][c3]
  cpx *d_tp = nullptr;
  unsigned ctr_base = 0;      // its value before the next launch (advanced once a launch pair has been accepted)
  bool ctr_stale = false;     // a launch that used the counter failed: zero it again before the next use
  // even / odd
  tsdgpu_fft *sub = nullptr;
  cpx *d_rot = nullptr;       // W_n^k, k < n (even split)
  int n2 = 0;
  int mix_m = 0, mix_P = 0;   // MIXED: n = mix_m * mix_P
  MrFactors mr{};             // SMOOTH: radix sequence; mr_tpt threads per transform
  int mr_tpt = 0;
  bool blu_fused = false;     // ODD: the one-kernel Bluestein (n2 = 1024 .. 16384)
  cpx *d_wm = nullptr;        // W_m^j, j < m
  cpx *d_chirp = nullptr, *d_xc = nullptr;   // Bluestein chirp (2n-1) and FFT of its conjugate (n2)
  cpx *d_twf = nullptr;       // n2 <= 1024: W_n2^(q i) in rows [i][17], i < n2 / 16 (fft_blu_wave_kernel)
  DevBuf work, work2, in_stage, out_stage;
  StepOrder order;            // top-level plans only (sub-plans run under their owner's)
  // grouped schedule of the 2^20 plan (TSDGPU_FFT_GROUP): two side streams and their events
};

namespace {

int log2_exact(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

int upload(cpx **dst, const std::vector<cpx> &v)
{
  if (hipMalloc((void **) dst, v.size() * sizeof(cpx)) != hipSuccess)
    return set_err(TSDGPU_ERR_HIP, "fft: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
  if (hipMemcpy(*dst, v.data(), v.size() * sizeof(cpx), hipMemcpyHostToDevice) != hipSuccess)
    return set_err(TSDGPU_ERR_HIP, "fft: upload failed: %s", hipGetErrorString(hipGetLastError()));
  return TSDGPU_OK;
}

std::vector<cpx> twiddle_table(int n, int count)   // exp(-2 pi i k / n), k < count
{
  std::vector<cpx> t((size_t) (count > 0 ? count : 1));
  const double PI = 3.14159265358979323846;
  for (int k = 0; k < count; k++) {
    const double a = -2.0 * PI * (double) k / (double) n;
    t[k] = make_float2((float) std::cos(a), (float) std::sin(a));
  }
  if (count == 0) t[0] = make_float2(1.f, 0.f);
  return t;
}

// prochaine_puissance_de_2 (libtsd core/src/tsd.cc:287-291), float-log based like the reference
int ref_next_pow2(int i)
{
  const int lg2 = (int) std::ceil(std::log((float) i) / std::log(2.0f));
  return (int) (1l << lg2);
}

int plan_create(tsdgpu_fft **out, int n);
void plan_destroy(tsdgpu_fft *p);

// n = m * P with m odd in 3..8191 and P = 2^p in 2..4096: the two-pass mixed-radix plan
bool mixed_split(int n, int *m, int *P)
{
  int q = n, pw = 1;
  while ((q & 1) == 0) { q >>= 1; pw <<= 1; }
  if (q < 3 || q > 8191 || pw < 2 || pw > 4096 || n > (1 << 22)) return false;   // (W_n table: 8 n bytes)
  *m = q;
  *P = pw;
  return true;
}

// n = 2^a 3^b 5^c 7^d 11^e 13^f, not a power of two, <= 16384: the one-kernel mixed-radix plan.
// Radix sequence: 16s and the remaining 8 / 4 / 2, then the odd primes.  Threads per transform:
// enough for every pass to keep its butterflies' inputs in MR_PTS registers.
// n = m * P with m odd in 3 .. 15, P a power of two >= 16, m P / 16 <= 1024 threads and m LDS images within the CU: fft_oddpow2_kernel
bool oddpow2_split(int n, int *m, int *P)
{
  if (n < 48 || n > 16384) return false;
  int pw = n & -n, odd = n / pw;
  if (odd < 3 || odd > 15 || pw < 16) return false;
  if ((size_t) odd * (pw + pw / 16) * sizeof(cpx) > 150 * 1024) return false;
  // where it beats fft_mr_kernel (profiles/r3_perf_fft_oddpow2.txt; TSDGPU_FFT_ODDPOW2_ALL=1: wherever it fits): the direct
  // m-point combination grows with m^2 and short transforms leave a workgroup little to do per barrier
  if (dev_switch("FFT_ODDPOW2_ALL") == nullptr) {
    const bool wins = (odd == 3 && pw >= 32) || (odd == 5 && pw >= 32 && pw <= 1024) || (odd == 7 && pw >= 128 && pw <= 512) ||
                      (odd == 9 && pw >= 256 && pw <= 512);
    if (!wins) return false;
  }
  *m = odd;
  *P = pw;
  return true;
}
bool smooth_plan(int n, MrFactors *F, int *tpt_out)
{
  if (n < 3 || n > 16384 || (n & (n - 1)) == 0) return false;
  int q = n, nf = 0, tpt = 1;
  MrFactors f{};
  auto push = [&](int r) {
    if (nf < MR_MAXF) f.r[nf] = r;
    nf++;
    const int U = MR_PTS / r, nb = n / r;
    tpt = std::max(tpt, (nb + U - 1) / U);
  };
  int odd = n;
  while ((odd & 1) == 0) odd >>= 1;
  {
    int o = odd;
    for (int pr : {3, 5, 7, 11, 13})
      while (o % pr == 0) o /= pr;
    if (o != 1) return false;                           // a larger prime factor remains
  }
  q = n / odd;                                          // the power of two goes first: Ns stays a power of
  while (q % 16 == 0) { push(16); q /= 16; }            // two through its passes and k = j mod Ns is a mask
  if (q > 1) push(q);                                   // 8, 4 or 2
  q = odd;
  for (int pr : {13, 11, 7, 5, 3})
    while (q % pr == 0) { push(pr); q /= pr; }
  q = n / odd;
  // Parity: beyond an odd part of 31 the reference's result carries the rounding of its float32
  // Bluestein chirp (1e-5 of the maximum and more); a more accurate transform would leave the 1e-5
  // band around it, so those sizes stay on the plan that reproduces the chirp (MIXED / ODD).
  if (n / q > 31) return false;
  if (nf > MR_MAXF) return false;
  tpt = (tpt + 7) / 8 * 8;
  if (tpt > 1024) return false;
  f.nf = nf;
  *F = f;
  *tpt_out = tpt;
  return true;
}

int plan_init(tsdgpu_fft *p, int n)
{
  p->n = n;
  int rc = TSDGPU_OK;
  if (n == 1) {
    p->kind = tsdgpu_fft::ONE;
  } else if ((n & (n - 1)) == 0) {
    p->logn = log2_exact(n);
    const bool fast = dev_switch("FFT_GENERIC") == nullptr;
    if (fast && n == (1 << 20)) {
      p->kind = n == 1024 ? tsdgpu_fft::POW2_W1024 : tsdgpu_fft::POW2_W1M;
      std::vector<cpx> t1(1024), t2(1024);
      w1024::fill_twiddles(t1.data(), t2.data());
      if ((rc = upload(&p->d_w1, t1))) return rc;
      if ((rc = upload(&p->d_w2, t2))) return rc;
      if (n != 1024) {
        std::vector<cpx> ta((size_t) 1024 * 64), td((size_t) 1024 * 16);
        const double PI = 3.14159265358979323846;
        for (int c = 0; c < 1024; c++) {
          for (int lane = 0; lane < 64; lane++) {
            const int64_t m = ((int64_t) c * ((lane >> 2) + 16 * (lane & 3))) % n;
            const double a = -2.0 * PI * (double) m / (double) n;
            ta[(size_t) c * 64 + lane] = make_float2((float) std::cos(a), (float) std::sin(a));
          }
          for (int r = 0; r < 16; r++) {
            const int64_t m = ((int64_t) c * (64 * (r >> 2) + 256 * (r & 3))) % n;
            const double a = -2.0 * PI * (double) m / (double) n;
            td[(size_t) c * 16 + r] = make_float2((float) std::cos(a), (float) std::sin(a));
          }
        }
        if ((rc = upload(&p->d_ta, ta))) return rc;
        if ((rc = upload(&p->d_td, td))) return rc;
        if (hipMalloc((void **) &p->d_ctr, 256) != hipSuccess || hipMemset(p->d_ctr, 0, 256) != hipSuccess) {
          (void) hipGetLastError();
          p->d_ctr = nullptr;                                // (the static partition then)
        }
        (void) hipFuncSetAttribute((const void *) fft1m_cols_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void) hipFuncSetAttribute((const void *) fft1m_cols_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void) hipFuncSetAttribute((const void *) fft1m_cols_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void) hipFuncSetAttribute((const void *) fft1m_cols_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void) hipGetLastError();
      }
    } else if (fast && n >= 16 && n <= S16_MAX_N) {
      p->kind = tsdgpu_fft::POW2_S16;
      rc = upload(&p->d_tw, twiddle_table(n, n / 16));       // W_n^i, i < n/16: the base twiddles of every pass
#define S16_ATTR(R) (void) hipFuncSetAttribute((const void *) fft_s16_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
      S16_ATTR(16); S16_ATTR(8); S16_ATTR(4); S16_ATTR(2);
#undef S16_ATTR
#define S16P_ATTR(R) (void) hipFuncSetAttribute((const void *) fft_s16_persistent_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
      S16P_ATTR(16); S16P_ATTR(8); S16P_ATTR(4); S16P_ATTR(2);
#undef S16P_ATTR
      (void) hipGetLastError();
    } else if (n <= LDS_MAX_N) {
      p->kind = tsdgpu_fft::POW2_LDS;
      rc = upload(&p->d_tw, twiddle_table(n, n / 2));
    } else {
      TSD_CHECK(p->logn <= 28, "fft: n = %d exceeds the four-step limit 2^28 (two passes of at most 16384-point columns)", n);
      p->kind = tsdgpu_fft::POW2_4STEP;
      p->logN1 = p->logn / 2;
      // n = 2^15 .. 2^19: planned a second time as 1024 x C, so that pass 1 of LARGE calls runs on the 2^20 plan's column kernel
      // (in-wave 1024-point FFT, prefetched tiles, dynamic hand-out: 0.73 ms per 2^28 points against 0.88 for a fft_cols16_kernel
      // pass); small calls keep the square split (a 128-KiB tile per workgroup costs a single 2^16-point transform 4 us of its 18);
      // TSDGPU_FFT_NO_1K_P1=1: the square split always
      const bool use1k = fast && p->logn >= 15 && p->logn <= 19 && dev_switch("FFT_NO_1K_P1") == nullptr;
      p->logN2 = p->logn - p->logN1;
      p->N1 = 1 << p->logN1;
      p->N2 = 1 << p->logN2;
      if ((rc = upload(&p->d_tw1, twiddle_table(p->N1, p->N1 / 2)))) return rc;
      if ((rc = upload(&p->d_tw2, twiddle_table(p->N2, p->N2 / 2)))) return rc;
      // W_n^m = thi[m >> 12] * tlo[m & 4095]
      std::vector<cpx> hi((size_t) (n >> 12)), lo(4096);
      const double PI = 3.14159265358979323846;
      for (int a = 0; a < (n >> 12); a++) {
        const double ang = -2.0 * PI * ((double) a * 4096.0) / (double) n;
        hi[a] = make_float2((float) std::cos(ang), (float) std::sin(ang));
      }
      for (int b = 0; b < 4096; b++) {
        const double ang = -2.0 * PI * (double) b / (double) n;
        lo[b] = make_float2((float) std::cos(ang), (float) std::sin(ang));
      }
      if ((rc = upload(&p->d_thi, hi))) return rc;
      if ((rc = upload(&p->d_tlo, lo))) return rc;
      if (use1k) {
        p->logN2a = p->logn - 10;
        p->N2a = 1 << p->logN2a;
        if ((rc = upload(&p->d_tw2a, twiddle_table(p->N2a, p->N2a / 2)))) return rc;
        std::vector<cpx> t1(1024), t2(1024);
        w1024::fill_twiddles(t1.data(), t2.data());
        if ((rc = upload(&p->d_w1, t1))) return rc;
        if ((rc = upload(&p->d_w2, t2))) return rc;
        std::vector<cpx> ta((size_t) p->N2a * 64), td((size_t) p->N2a * 16);
        for (int c = 0; c < p->N2a; c++) {
          for (int lane = 0; lane < 64; lane++) {
            const double a = -2.0 * PI * (double) (((int64_t) c * ((lane >> 2) + 16 * (lane & 3))) % n) / (double) n;
            ta[(size_t) c * 64 + lane] = make_float2((float) std::cos(a), (float) std::sin(a));
          }
          for (int r = 0; r < 16; r++) {
            const double a = -2.0 * PI * (double) (((int64_t) c * (64 * (r >> 2) + 256 * (r & 3))) % n) / (double) n;
            td[(size_t) c * 16 + r] = make_float2((float) std::cos(a), (float) std::sin(a));
          }
        }
        if ((rc = upload(&p->d_ta, ta))) return rc;
        if ((rc = upload(&p->d_td, td))) return rc;
        if (hipMalloc((void **) &p->d_ctr, 256) != hipSuccess || hipMemset(p->d_ctr, 0, 256) != hipSuccess) {
          (void) hipGetLastError();
          p->d_ctr = nullptr;
        }
        (void) hipFuncSetAttribute((const void *) fft1m_cols_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void) hipFuncSetAttribute((const void *) fft1m_cols_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        p->p1k1 = true;
      }
      // 2^24, 2^25: the three-pass plan; 2^23 keeps the 2048 x 4096 plan unless TSDGPU_FFT_3PASS_23=1 (measured: see DESIGN.md 3.3)
      const bool use3 = p->logn >= 23 && p->logn <= 25 && dev_switch("FFT_NO_3PASS") == nullptr &&
                        (p->logn > 23 || dev_switch_int("FFT_3PASS_23", 0) != 0);
      if (!use3 && (p->N2 == 2048 || p->N1 == 2048) && (p->N1 & 15) == 0 && dev_switch("FFT_NO_2K") == nullptr) {
        // 2048-point column passes on fft2k_cols_kernel: the in-wave engine's tables, W_2048^k0 (64 entries) and the 16 constants
        // W_2048^(64 a + 256 b)
        std::vector<cpx> t1(1024), t2(1024), w2k(96);       // (... and, for the decimation-in-frequency form, W_2048^(64 i), i < 16)
        w1024::fill_twiddles(t1.data(), t2.data());
        for (int i = 0; i < 96; i++) {
          const int m = i < 64 ? i : (i < 80 ? 64 * ((i - 64) >> 2) + 256 * ((i - 64) & 3) : 64 * (i - 80));
          const double ang = -2.0 * PI * (double) m / 2048.0;
          w2k[i] = make_float2((float) std::cos(ang), (float) std::sin(ang));
        }
        if ((rc = upload(&p->d_w1, t1))) return rc;
        if ((rc = upload(&p->d_w2, t2))) return rc;
        if ((rc = upload(&p->d_w2k, w2k))) return rc;
        auto Wn = [&](int64_t m) {
          const double a = -2.0 * PI * (double) (m % n) / (double) n;
          return make_float2((float) std::cos(a), (float) std::sin(a));
        };
        if (p->N1 == 1024) {
          // pass 1 on fft1m_cols_kernel<1>: 1024-point columns of the N2 = 2048 columns, four-step twiddle W_n^(c k) as TA[c][lane] TD[c][r]
          std::vector<cpx> ta((size_t) p->N2 * 64), td((size_t) p->N2 * 16);
          for (int c = 0; c < p->N2; c++) {
            for (int lane = 0; lane < 64; lane++) ta[(size_t) c * 64 + lane] = Wn((int64_t) c * ((lane >> 2) + 16 * (lane & 3)));
            for (int r = 0; r < 16; r++) td[(size_t) c * 16 + r] = Wn((int64_t) c * (64 * (r >> 2) + 256 * (r & 3)));
          }
          if ((rc = upload(&p->d_ta, ta))) return rc;
          if ((rc = upload(&p->d_td, td))) return rc;
          (void) hipFuncSetAttribute((const void *) fft1m_cols_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          (void) hipFuncSetAttribute((const void *) fft1m_cols_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        } else if (p->N1 == 2048) {
          // pass 1 on fft2k_cols_kernel<1>: TA[c][lane] = W_n^(c k0), TD[c][r] = W_n^(c (64 a + 256 b)), TD[c][16] = W_n^(1024 c)
          std::vector<cpx> ta((size_t) p->N2 * 64), td((size_t) p->N2 * 17);
          for (int c = 0; c < p->N2; c++) {
            for (int lane = 0; lane < 64; lane++) ta[(size_t) c * 64 + lane] = Wn((int64_t) c * ((lane >> 2) + 16 * (lane & 3)));
            for (int r = 0; r < 16; r++) td[(size_t) c * 17 + r] = Wn((int64_t) c * (64 * (r >> 2) + 256 * (r & 3)));
            td[(size_t) c * 17 + 16] = Wn((int64_t) c * 1024);
          }
          if ((rc = upload(&p->d_ta, ta))) return rc;
          if ((rc = upload(&p->d_td, td))) return rc;
          p->cols2k_p1 = true;
        }
        if (hipMalloc((void **) &p->d_ctr, 256) != hipSuccess || hipMemset(p->d_ctr, 0, 256) != hipSuccess) {
          (void) hipGetLastError();
          p->d_ctr = nullptr;                                // (the static partition then)
        }
        (void) hipFuncSetAttribute((const void *) fft2k_cols_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void) hipFuncSetAttribute((const void *) fft2k_cols_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void) hipFuncSetAttribute((const void *) fft2k_cols_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void) hipFuncSetAttribute((const void *) fft2k_cols_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        p->cols2k = p->N2 == 2048;
      }
      if (use3) {
        // three passes of 128-B row segments (fft_planes_kernel): n = 1024 x C1 x 1024
        const int C1 = 1 << (p->logn - 20), C = C1 * 1024;
        std::vector<cpx> t1(1024), t2(1024);
        w1024::fill_twiddles(t1.data(), t2.data());
        if ((rc = upload(&p->d_w1, t1))) return rc;
        if ((rc = upload(&p->d_w2, t2))) return rc;
        auto Wd = [&](int64_t m, int64_t mod) {
          const double a = -2.0 * PI * (double) (m % mod) / (double) mod;
          return make_float2((float) std::cos(a), (float) std::sin(a));
        };
        {
          std::vector<cpx> ta((size_t) C * 64), td((size_t) C * 16), tp((size_t) 1024 * C1);
          for (int c = 0; c < C; c++) {
            for (int lane = 0; lane < 64; lane++) ta[(size_t) c * 64 + lane] = Wd((int64_t) c * ((lane >> 2) + 16 * (lane & 3)), n);
            for (int r = 0; r < 16; r++) td[(size_t) c * 16 + r] = Wd((int64_t) c * (64 * (r >> 2) + 256 * (r & 3)), n);
          }
          for (int b = 0; b < 1024; b++)
            for (int q = 0; q < C1; q++) tp[(size_t) b * C1 + q] = Wd((int64_t) b * q, C);
          if ((rc = upload(&p->d_ta, ta))) return rc;
          if ((rc = upload(&p->d_td, td))) return rc;
          if ((rc = upload(&p->d_tp, tp))) return rc;
        }
        if (hipMalloc((void **) &p->d_ctr, 256) != hipSuccess || hipMemset(p->d_ctr, 0, 256) != hipSuccess) {
          (void) hipGetLastError();
          p->d_ctr = nullptr;
        }
        (void) hipFuncSetAttribute((const void *) fft1m_cols_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void) hipFuncSetAttribute((const void *) fft1m_cols_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void) hipFuncSetAttribute((const void *) fft1m_cols_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void) hipFuncSetAttribute((const void *) fft1m_cols_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        p->c3 = C1;
      }
      // the column tiles use up to ~150 KiB of the CU's 160 KiB LDS
#define C16_ATTR(P, R) (void) hipFuncSetAttribute((const void *) fft_cols16_kernel<P, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
      C16_ATTR(1, 16); C16_ATTR(1, 8); C16_ATTR(1, 4); C16_ATTR(1, 2); C16_ATTR(2, 16); C16_ATTR(2, 8); C16_ATTR(2, 4); C16_ATTR(2, 2);
#undef C16_ATTR
      (void) hipFuncSetAttribute((const void *) fft_cols_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void) hipFuncSetAttribute((const void *) fft_cols_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void) hipGetLastError();
    }
  } else if (oddpow2_split(n, &p->mix_m, &p->mix_P) && dev_switch("FFT_GENERIC") == nullptr && dev_switch("FFT_NO_SMOOTH") == nullptr &&
             dev_switch("FFT_NO_ODDPOW2") == nullptr) {
    p->kind = tsdgpu_fft::ODDPOW2;
    const int m = p->mix_m, P = p->mix_P;
    p->logn = log2_exact(P);
    if ((rc = upload(&p->d_wm, twiddle_table(m, m)))) return rc;             // W_m^j
    if ((rc = upload(&p->d_rot, twiddle_table(n, n)))) return rc;            // W_n^j
    if ((rc = upload(&p->d_tw, twiddle_table(P, std::max(1, P / 16))))) return rc;
#define OP_ATTR(R) (void) hipFuncSetAttribute((const void *) fft_oddpow2_kernel<R, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
  (void) hipFuncSetAttribute((const void *) fft_oddpow2_kernel<R, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
  (void) hipFuncSetAttribute((const void *) fft_oddpow2_kernel<R, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    OP_ATTR(16); OP_ATTR(8); OP_ATTR(4); OP_ATTR(2);
#undef OP_ATTR
    (void) hipGetLastError();
  } else if (smooth_plan(n, &p->mr, &p->mr_tpt) && dev_switch("FFT_GENERIC") == nullptr && dev_switch("FFT_NO_SMOOTH") == nullptr) {
    p->kind = tsdgpu_fft::SMOOTH;
    if ((rc = upload(&p->d_rot, twiddle_table(n, n)))) return rc;            // W_n^j
    (void) hipFuncSetAttribute((const void *) fft_mr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void) hipGetLastError();
  } else if ((n & 1) == 0 && mixed_split(n, &p->mix_m, &p->mix_P) && dev_switch("FFT_GENERIC") == nullptr) {
    p->kind = tsdgpu_fft::MIXED;
    const int m = p->mix_m, P = p->mix_P;
    p->logn = log2_exact(P);
    if (m <= 31) {
      if ((rc = upload(&p->d_wm, twiddle_table(m, m)))) return rc;           // pass 1 = direct m-point DFT
    } else {
      if ((rc = plan_create(&p->sub, m))) return rc;                           // pass 1 = one-kernel Bluestein
      TSD_CHECK(p->sub->kind == tsdgpu_fft::ODD && p->sub->blu_fused, "fft: no fused Bluestein plan for m = %d", m);
    }
    if ((rc = upload(&p->d_rot, twiddle_table(n, n)))) return rc;            // W_n^j: the four-step twiddles
    if ((rc = upload(&p->d_tw, twiddle_table(P, std::max(1, P / 16))))) return rc;
#define C16_ATTR(R) (void) hipFuncSetAttribute((const void *) fft_cols16_kernel<2, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    C16_ATTR(16); C16_ATTR(8); C16_ATTR(4); C16_ATTR(2);
#undef C16_ATTR
    (void) hipFuncSetAttribute((const void *) fft_odd_dft_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void) hipFuncSetAttribute((const void *) fft_odd_dft_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void) hipFuncSetAttribute((const void *) fft_odd_dft_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void) hipGetLastError();
  } else if ((n & 1) == 0) {
    p->kind = tsdgpu_fft::EVEN;
    if ((rc = plan_create(&p->sub, n / 2))) return rc;
    // tfr_rotation_rapide (fourier.cc:32-46): double-precision recurrence r *= w0, rounded to float
    std::vector<cpx> rot((size_t) n);
    const double PI = 3.14159265358979323846;
    double rr = 1.0, ri = 0.0;
    const double wr = std::cos(-2 * PI / n), wi = std::sin(-2 * PI / n);
    for (int i = 0; i < n; i++) {
      rot[i] = make_float2((float) rr, (float) ri);
      const double tr = rr * wr - ri * wi, ti = rr * wi + ri * wr;
      rr = tr; ri = ti;
    }
    rc = upload(&p->d_rot, rot);
  } else {
    p->kind = tsdgpu_fft::ODD;
    p->n2 = ref_next_pow2(2 * n - 1);
    if ((rc = plan_create(&p->sub, p->n2))) return rc;
    // chirp = polar(square(linspace(-(n-1), n-1, 2n-1)) / 2 * (-2 pi / n)), all in float like
    // the reference (fourier.cc:396-399): the float rounding of the angle is part of its result
    std::vector<cpx> chirp((size_t) (2 * n - 1)), icp((size_t) p->n2, make_float2(0.f, 0.f));
    const double step = ((double) (float) (n - 1) - (double) (float) -(n - 1)) / (2 * n - 2);
    const float mul = (float) (-2 * 3.14159265358979323846 / n);
    for (int i = 0; i < 2 * n - 1; i++) {
      const float t = i == 0 ? (float) -(n - 1) : (float) ((double) (float) -(n - 1) + step * i);
      float v = (t * t) / 2;
      v *= mul;
      chirp[i] = make_float2(std::cos(v), std::sin(v));
      icp[i] = make_float2(chirp[i].x, -chirp[i].y);
    }
    if ((rc = upload(&p->d_chirp, chirp))) return rc;
    cpx *d_icp = nullptr;
    if ((rc = upload(&d_icp, icp))) return rc;
    if (hipMalloc((void **) &p->d_xc, (size_t) p->n2 * sizeof(cpx)) != hipSuccess) {
      (void) hipFree(d_icp);
      return set_err(TSDGPU_ERR_HIP, "fft: hipMalloc failed");
    }
    rc = tsdgpu_fft_step(p->sub, d_icp, p->d_xc, 1, 1, nullptr);
    (void) hipDeviceSynchronize();
    (void) hipFree(d_icp);
    if (!rc && p->n2 >= 16 && p->n2 <= S16_MAX_N && dev_switch("FFT_GENERIC") == nullptr) {
      p->blu_fused = true;
      rc = upload(&p->d_tw, twiddle_table(p->n2, p->n2 / 16));
      if (!rc && p->n2 <= 1024) {
        const int tp = p->n2 / 16;
        std::vector<cpx> twf((size_t) tp * BW_TWROW, make_float2(1.f, 0.f));
        const double PI = 3.14159265358979323846;
        for (int i = 0; i < tp; i++)
          for (int q = 0; q < 16; q++) {
            const double a = -2.0 * PI * (double) ((q * i) % p->n2) / (double) p->n2;
            twf[(size_t) i * BW_TWROW + q] = make_float2((float) std::cos(a), (float) std::sin(a));
          }
        rc = upload(&p->d_twf, twf);
      }
      (void) hipFuncSetAttribute((const void *) fft_bluestein_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void) hipFuncSetAttribute((const void *) fft_bluestein_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void) hipFuncSetAttribute((const void *) fft_bluestein_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void) hipFuncSetAttribute((const void *) fft_bluestein_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define BW_ATTR(R)                                                                                                                   \
  (void) hipFuncSetAttribute((const void *) fft_blu_wave_kernel<R, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
  (void) hipFuncSetAttribute((const void *) fft_blu_wave_kernel<R, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
      BW_ATTR(1); BW_ATTR(2); BW_ATTR(4); BW_ATTR(8); BW_ATTR(16); BW_ATTR(32); BW_ATTR(64); BW_ATTR(128);
#undef BW_ATTR
      (void) hipGetLastError();
    }
  }
  return rc;
}

int plan_create(tsdgpu_fft **out, int n)
{
  tsdgpu_fft *p = new tsdgpu_fft();
  const int rc = plan_init(p, n);
  if (rc) {
    plan_destroy(p);
    return rc;
  }
  *out = p;
  return TSDGPU_OK;
}

void plan_destroy(tsdgpu_fft *p)
{
  if (!p) return;
  if (p->sub) plan_destroy(p->sub);
  if (p->d_ctr) (void) hipFree(p->d_ctr);
  for (cpx *q : {p->d_tw, p->d_tw1, p->d_tw2, p->d_thi, p->d_tlo, p->d_rot, p->d_chirp, p->d_xc, p->d_twf, p->d_tp, p->d_tw2a, p->d_w1, p->d_w2, p->d_ta, p->d_td, p->d_wm, p->d_w2k})
    if (q) (void) hipFree(q);
  p->work.release();
  p->work2.release();
  p->in_stage.release();
  p->out_stage.release();
  p->order.release();
  delete p;
}

inline unsigned blocks_for(int64_t total) { return (unsigned) cdiv(total, 256); }

// One-kernel Bluestein of `p` (an ODD plan with blu_fused) over ntr = batch * P transforms;
// P > 1 / Wn: pass 1 of the mixed-radix plan (see fft_bluestein_kernel)
// fuse: pass 2 of a mixed-radix plan with P = 2, 4, 8, 16 in the same kernel (bluestein_fusable)
bool bluestein_fusable(const tsdgpu_fft *sub, int P)
{
  static const bool off = dev_switch("FFT_MIXED_UNFUSED") != nullptr;
  const int n2 = sub->n2, tpt = n2 / 16;
  return !off && sub->blu_fused && (P == 2 || P == 4 || P == 8 || P == 16) && P * tpt <= 1024 &&
         (size_t) std::max(256 / tpt, P) * (n2 + n2 / 16) * sizeof(cpx) <= 158 * 1024;
}
int cu_count()
{
  static const int NCU = [] {
    int dev = 0, c = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void) hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
    return c > 0 ? c : 256;
  }();
  return NCU;
}
// the wave-level kernel serves transforms of at most 64 threads (n2 <= 1024); fused groups of P residues within 512 threads
// (TSDGPU_FFT_BLU_OLD=1: fft_bluestein_kernel everywhere)
bool blu_wave_fits(const tsdgpu_fft *p, int P, bool fuse)
{
  const int tpt = p->n2 / 16;
  return p->blu_fused && (p->d_twf || tpt > BW_TAB_MAX_TPT) && tpt >= 1 && tpt <= BW_MAX_TPT && (P & (P - 1)) == 0 && (!fuse || P * tpt <= 512) && dev_switch("FFT_BLU_OLD") == nullptr;
}
// x: transform (b, r) starts at x + b * bstride + r (element stride P); pw != NULL: |X|^2 out instead of X, win = the window
// pw_rows != NULL (psd_welch): the kernel sums |X|^2 over the transforms of each workgroup; pw receives *pw_rows partial rows of
// n P floats, one per workgroup (at most pw_cap rows, else TSDGPU_ERR_INVALID); pw == NULL with pw_rows: only the row count is computed
int launch_blu_wave(const tsdgpu_fft *p, const cpx *x, cpx *y, float *pw, const float *win, int64_t bstride, int64_t ntr, int reverse,
                    int conj_out, int P, const cpx *Wn, hipStream_t st, bool fuse, int64_t *pw_rows = nullptr, int64_t pw_cap = 0)
{
  const int n = p->n, n2 = p->n2, tpt = n2 / 16;
  const float gf = std::sqrt((float) n2) / std::sqrt((float) n) / (float) n2;
  int logP = 0;
  while ((1 << logP) < P) logP++;
  const int NT = fuse ? std::max(256, P * tpt) : 256, T = NT / tpt;
  const bool tab = tpt <= BW_TAB_MAX_TPT;
  const size_t lds = (size_t) ((n + 1) + n2 + (tab ? tpt * BW_TWROW + (tpt & 1) : tpt) + (size_t) (fuse ? P : 1) * n + 1 + (size_t) T * (n2 + n2 / 16)) * sizeof(cpx) +
                     (pw_rows && fuse ? (size_t) n * P * sizeof(float) : 0);
  TSD_CHECK(lds <= 160 * 1024, "fft_step: Bluestein tables of n = %d do not fit the LDS", n);
  const int64_t nslots = cdiv(ntr, T);
  // persistent grid = the workgroups that are resident at once (registers and LDS: asked from the runtime, remembered per shape)
  auto resident = [&](const void *fn) {
    static std::mutex mtx;
    static std::unordered_map<uint64_t, int> memo;
    const uint64_t key = ((uint64_t) (uintptr_t) fn << 20) ^ ((uint64_t) (lds / 256) << 2) ^ (uint64_t) (NT / 256);
    std::lock_guard<std::mutex> lock(mtx);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, NT, lds) != hipSuccess || nb < 1) {
      (void) hipGetLastError();
      nb = 1;
    }
    memo[key] = nb;
    return nb;
  };
  const BluArgs A{x, y, pw, p->d_chirp, p->d_xc, tab ? p->d_twf : p->d_tw, Wn, win, n, n2, tpt, reverse, logP, conj_out, fuse ? P : 0, gf,
                  1.0f / std::sqrt((float) P), ntr, bstride};
#define BW_LAUNCH(TP)                                                                                                          \
  do {                                                                                                                         \
    const void *fn = pw_rows ? (const void *) fft_blu_wave_kernel<TP, true> : (const void *) fft_blu_wave_kernel<TP, false>;   \
    const int64_t grid = std::min<int64_t>(nslots, (int64_t) cu_count() * resident(fn));                                        \
    if (pw_rows) {                                                                                                             \
      *pw_rows = grid;                                                                                                         \
      if (!pw) return TSDGPU_OK;                                                                                               \
      TSD_CHECK(*pw_rows <= pw_cap, "welch: %lld partial rows, room for %lld", (long long) *pw_rows, (long long) pw_cap);      \
    }                                                                                                                          \
    if (pw) hipLaunchKernelGGL((fft_blu_wave_kernel<TP, true>), dim3((unsigned) grid), dim3(NT), lds, st, A);                  \
    else hipLaunchKernelGGL((fft_blu_wave_kernel<TP, false>), dim3((unsigned) grid), dim3(NT), lds, st, A);                    \
  } while (0)
  switch (tpt) {
    case 1: BW_LAUNCH(1); break;
    case 2: BW_LAUNCH(2); break;
    case 4: BW_LAUNCH(4); break;
    case 8: BW_LAUNCH(8); break;
    case 16: BW_LAUNCH(16); break;
    case 32: BW_LAUNCH(32); break;
    case 64: BW_LAUNCH(64); break;
    default: BW_LAUNCH(128); break;
  }
#undef BW_LAUNCH
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}
int launch_bluestein(const tsdgpu_fft *p, const cpx *x, cpx *y, int64_t ntr, int reverse, int conj_out, int P, const cpx *Wn,
                     hipStream_t st, bool fuse = false)
{
  const int n = p->n, n2 = p->n2;
  // unitary FFT, product with the unitary xc, unitary inverse, times sqrt(n2)/sqrt(n): the two
  // unnormalised transforms carry n2, so the output factor is sqrt(n2)/sqrt(n) / n2
  const float gf = std::sqrt((float) n2) / std::sqrt((float) n) / (float) n2;
  int l2 = 0;
  while ((1 << l2) < n2) l2++;
  const int r0 = 1 << ((l2 & 3) == 0 ? 4 : (l2 & 3)), tpt = n2 / 16;
  if (blu_wave_fits(p, P, fuse))
    return launch_blu_wave(p, x, y, nullptr, nullptr, (int64_t) n * P, ntr, reverse, conj_out, P, Wn, st, fuse);
  const int threads = std::max(256, fuse ? P * tpt : tpt), T = threads / tpt;       // (fused: whole groups of P residues per workgroup)
  const size_t lds = (size_t) T * (n2 + n2 / 16) * sizeof(cpx);
  const int64_t grid = cdiv(ntr, T);
  TSD_CHECK(grid <= 0x7fffffff, "fft_step: too many transforms");
  const float s2 = 1.0f / std::sqrt((float) P);
#define BLU_LAUNCH(R) hipLaunchKernelGGL((fft_bluestein_kernel<R>), dim3((unsigned) grid), dim3(threads), lds, st, x, y, p->d_chirp, p->d_xc, p->d_tw, n, n2, tpt, reverse, gf, P, Wn, conj_out, ntr, fuse ? P : 0, s2)
  if (r0 == 16) BLU_LAUNCH(16); else if (r0 == 8) BLU_LAUNCH(8); else if (r0 == 4) BLU_LAUNCH(4); else BLU_LAUNCH(2);
#undef BLU_LAUNCH
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}

// device pointers in, device pointers out; x == y allowed
int step_device(tsdgpu_fft *p, const cpx *x, cpx *y, int batch, int forward, hipStream_t st)
{
  const int n = p->n;
  const int inverse = forward ? 0 : 1;
  const int64_t total = (int64_t) n * batch;
  if (batch > 65535 && (p->kind == tsdgpu_fft::MIXED || p->kind == tsdgpu_fft::POW2_4STEP)) {
    // these plans put the batch index in gridDim.y: larger batches go in slices
    for (int b0 = 0; b0 < batch; b0 += 65535) {
      const int rc = step_device(p, x + (size_t) b0 * n, y + (size_t) b0 * n, std::min(65535, batch - b0), forward, st);
      if (rc) return rc;
    }
    return TSDGPU_OK;
  }
  switch (p->kind) {
    case tsdgpu_fft::ONE:
      if (x != y) TSD_HIP(hipMemcpyAsync(y, x, (size_t) total * sizeof(cpx), hipMemcpyDeviceToDevice, st));
      return TSDGPU_OK;
    case tsdgpu_fft::POW2_LDS: {
      const float scale = 1.0f / std::sqrt((float) n);
      hipLaunchKernelGGL(fft_rows_kernel, dim3((unsigned) batch), dim3(FFT_THREADS), (size_t) n * sizeof(cpx), st, x,
                         y, p->d_tw, n, p->logn, inverse, scale);
      TSD_HIP(hipGetLastError());
      return TSDGPU_OK;
    }
    case tsdgpu_fft::POW2_S16: {
      const float scale = 1.0f / std::sqrt((float) n);
      const int tpt = n / 16, threads = std::max(256, tpt), T = threads / tpt;
      const size_t lds = (size_t) T * (n + n / 16) * sizeof(cpx);
      const unsigned grid = (unsigned) cdiv(batch, T);
      const int r0 = 1 << ((p->logn & 3) == 0 ? 4 : (p->logn & 3));
      // one transform per workgroup and more transforms than the chip holds at once: persistent
      // workgroups that prefetch their next transform
      constexpr int PERSIST_MIN = 16384;   // measured: 16384 0.292 -> 0.252 ms per 2^26 points; 8192 and 4096 lose 1-10 %
      if (n >= PERSIST_MIN && T == 1) {
        static const int NCU = [] {
          int dev = 0, c = 256;
          if (hipGetDevice(&dev) == hipSuccess) (void) hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
          return c > 0 ? c : 256;
        }();
        const int per_cu = std::max(1, std::min((int) (160 * 1024 / lds), 2048 / threads));
        if (batch > 2 * NCU * per_cu) {
          const unsigned g = (unsigned) (NCU * per_cu);
#define S16P_LAUNCH(R) hipLaunchKernelGGL((fft_s16_persistent_kernel<R>), dim3(g), dim3(threads), lds, st, x, y, p->d_tw, n, tpt, inverse, scale, batch)
          if (r0 == 16) S16P_LAUNCH(16); else if (r0 == 8) S16P_LAUNCH(8); else if (r0 == 4) S16P_LAUNCH(4); else S16P_LAUNCH(2);
#undef S16P_LAUNCH
          TSD_HIP(hipGetLastError());
          return TSDGPU_OK;
        }
      }
#define S16_LAUNCH(R) hipLaunchKernelGGL((fft_s16_kernel<R>), dim3(grid), dim3(threads), lds, st, x, y, p->d_tw, n, tpt, inverse, scale, batch, (const cpx *) nullptr)
      if (r0 == 16) S16_LAUNCH(16); else if (r0 == 8) S16_LAUNCH(8); else if (r0 == 4) S16_LAUNCH(4); else S16_LAUNCH(2);
#undef S16_LAUNCH
      TSD_HIP(hipGetLastError());
      return TSDGPU_OK;
    }
    case tsdgpu_fft::POW2_W1024: {
      const float scale = 1.0f / 32.0f;
      hipLaunchKernelGGL(fft1024_rows_kernel, dim3((unsigned) cdiv(batch, 4)), dim3(256), 0, st, x, y, p->d_w1, p->d_w2,
                         inverse, scale, batch);
      TSD_HIP(hipGetLastError());
      return TSDGPU_OK;
    }
    case tsdgpu_fft::POW2_W1M: {
      // the transposed intermediate gets a padded row pitch: its columns are read back by
      // pass 2 with a stride that is no longer a power of two (spreads the HBM channels)
      constexpr int ZP = 1024 + 16;
      int rc = p->work.reserve((size_t) batch * 1024 * ZP * sizeof(cpx));
      if (rc) return rc;
      cpx *z = p->work.as<cpx>();
      // persistent workgroups, one per CU (148 KiB of LDS each); tiles are dealt round-robin
      const int ntiles = 64 * batch;
      static const int NCU = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
      }();
      const int GRID = NCU;
      const int grid = std::min(ntiles, GRID);
      // dynamic tile hand-out when every workgroup gets several tiles (TSDGPU_FFT_DYN=0: the static partition)
      const char *dyn_s = dev_switch("FFT_DYN");
      // (not while `st` records a graph: the base is a launch argument, a replay would see a spent counter and write nothing)
      unsigned *ctr = (p->d_ctr && ntiles >= 4 * grid && !(dyn_s && atoi(dyn_s) == 0) && !stream_is_capturing(st)) ? p->d_ctr : nullptr;
      if (ctr && p->ctr_stale) {
        TSD_HIP(hipMemsetAsync(p->d_ctr, 0, 256, st));
        p->ctr_base = 0;
        p->ctr_stale = false;
      }
      const unsigned b1 = p->ctr_base, b2 = b1 + (unsigned) ntiles + (unsigned) grid;
      if (ctr) {
        hipLaunchKernelGGL((fft1m_cols_kernel<1, true>), dim3(grid), dim3(1024), F1M_LDS, st, x, z, p->d_w1, p->d_w2, p->d_ta,
                           p->d_td, inverse, 1.0f, ZP, ntiles, ctr, b1, 6, 0);
        hipLaunchKernelGGL((fft1m_cols_kernel<2, true>), dim3(grid), dim3(1024), F1M_LDS, st, z, y, p->d_w1, p->d_w2, p->d_ta,
                           p->d_td, inverse, 1.0f / 1024.0f, ZP, ntiles, ctr, b2, 6, 0);
      } else {
        hipLaunchKernelGGL((fft1m_cols_kernel<1, false>), dim3(grid), dim3(1024), F1M_LDS, st, x, z, p->d_w1, p->d_w2, p->d_ta,
                           p->d_td, inverse, 1.0f, ZP, ntiles, ctr, b1, 6, 0);
        hipLaunchKernelGGL((fft1m_cols_kernel<2, false>), dim3(grid), dim3(1024), F1M_LDS, st, z, y, p->d_w1, p->d_w2, p->d_ta,
                           p->d_td, inverse, 1.0f / 1024.0f, ZP, ntiles, ctr, b2, 6, 0);
      }
      if (const hipError_t le = hipGetLastError(); le != hipSuccess) {
        if (ctr) p->ctr_stale = true;                    // (the device counter and the host base may have parted)
        return set_err(TSDGPU_ERR_HIP, "fft_step: launch failed: %s", hipGetErrorString(le));
      }
      if (ctr) p->ctr_base = b2 + (unsigned) ntiles + (unsigned) grid;
      return TSDGPU_OK;
    }
    case tsdgpu_fft::POW2_4STEP: {
      TSD_CHECK(batch <= 65535, "fft_step: batch %d too large for the four-step path (max 65535 per call)", batch);
      int rc = p->work.reserve((size_t) total * sizeof(cpx));
      if (rc) return rc;
      cpx *z = p->work.as<cpx>();
      const float scale = 1.0f / std::sqrt((float) n);
      // x viewed [N1][N2]: pass 1 = column FFTs (length N1) + twiddle, stored transposed [N2][N1];
      // z viewed [N2][N1]: pass 2 = column FFTs (length N2), natural store y[k2 * N1 + k1]
      static const bool generic = dev_switch("FFT_GENERIC") != nullptr;
      if (!generic) {
        auto launch = [&](int pass, const cpx *src, cpx *dst, const cpx *tw, int L, int logL, int C, float sc, int ipitch = 0) {
          const int tpt = L / 16;
          int CT = std::min(std::max(16, 256 / tpt), C);
          // 2048-point columns: 8 of them fill the LDS of a CU with ONE workgroup (139 KiB) whose load, transform and store phases
          // nothing overlaps; two workgroups of 4 columns measure 3-5 % faster (profiles/r3_fft_large_ab.txt)
          if (L == 2048) CT = std::min(CT, 4);
          while ((size_t) CT * (L + L / 16 + 1) * sizeof(cpx) > 150 * 1024) CT >>= 1;
          while (CT * tpt > 1024) CT >>= 1;                                    // L = 2048 -> 8 columns, 4096 -> 4
          const size_t lds = (size_t) CT * (L + L / 16 + 1) * sizeof(cpx);
          const dim3 grid((unsigned) (C / CT), (unsigned) batch), blk((unsigned) (CT * tpt));
          const int r0 = 1 << ((logL & 3) == 0 ? 4 : (logL & 3));
#define C16_LAUNCH(P, R) hipLaunchKernelGGL((fft_cols16_kernel<P, R>), grid, blk, lds, st, src, dst, tw, L, tpt, C, CT, p->d_thi, p->d_tlo, inverse, sc, 0, ipitch > 0 ? ipitch : C)
          if (pass == 1) {
            if (r0 == 16) C16_LAUNCH(1, 16); else if (r0 == 8) C16_LAUNCH(1, 8); else if (r0 == 4) C16_LAUNCH(1, 4); else C16_LAUNCH(1, 2);
          } else {
            if (r0 == 16) C16_LAUNCH(2, 16); else if (r0 == 8) C16_LAUNCH(2, 8); else if (r0 == 4) C16_LAUNCH(2, 4); else C16_LAUNCH(2, 2);
          }
#undef C16_LAUNCH
        };
        if (p->p1k1 && total >= (p->logn >= 18 ? (int64_t) 1 << 19 : (int64_t) 1 << 21)) {
          const int C = p->N2a, zp = 1024 + 16;
          rc = p->work.reserve((size_t) batch * C * zp * sizeof(cpx));
          if (rc) return rc;
          z = p->work.as<cpx>();
          const int ncu = cu_count();
          const int64_t nt1 = (int64_t) (C / 16) * batch;
          const int g1 = (int) std::min<int64_t>(nt1, ncu);
          const bool dyn_ok = p->d_ctr && dev_switch_int("FFT_DYN", 1) != 0 && !stream_is_capturing(st);
          unsigned *c1 = (dyn_ok && nt1 >= 4 * g1) ? p->d_ctr : nullptr;
          if (c1 && p->ctr_stale) {
            TSD_HIP(hipMemsetAsync(p->d_ctr, 0, 256, st));
            p->ctr_base = 0;
            p->ctr_stale = false;
          }
          const unsigned b1 = p->ctr_base;
          int tshift = 0;
          while ((16 << tshift) < C) tshift++;
          if (c1)
            hipLaunchKernelGGL((fft1m_cols_kernel<1, true>), dim3(g1), dim3(1024), F1M_LDS, st, x, z, p->d_w1, p->d_w2, p->d_ta, p->d_td, inverse, 1.0f, zp,
                               (int) nt1, c1, b1, tshift, 0);
          else
            hipLaunchKernelGGL((fft1m_cols_kernel<1, false>), dim3(g1), dim3(1024), F1M_LDS, st, x, z, p->d_w1, p->d_w2, p->d_ta, p->d_td, inverse, 1.0f, zp,
                               (int) nt1, (unsigned *) nullptr, 0u, tshift, 0);
          if (const hipError_t le = hipGetLastError(); le != hipSuccess) {
            if (c1) p->ctr_stale = true;
            return set_err(TSDGPU_ERR_HIP, "fft_step: launch failed: %s", hipGetErrorString(le));
          }
          if (c1) p->ctr_base = b1 + (unsigned) nt1 + (unsigned) g1;
          launch(2, z, y, p->d_tw2a, p->N2a, p->logN2a, 1024, scale, zp);
          TSD_HIP(hipGetLastError());
          return TSDGPU_OK;
        }
        if (p->c3) {
          const int C1 = p->c3, C = C1 * 1024, zp = 1024 + 16, lc1 = p->logn - 20;
          rc = p->work.reserve((size_t) batch * C * zp * sizeof(cpx));
          if (rc) return rc;
          z = p->work.as<cpx>();
          const int ncu = cu_count();
          const int64_t nt1 = (int64_t) (C / 16) * batch, nt2 = (int64_t) 64 * C1 * batch;
          TSD_CHECK(nt1 <= 0x3fffffff && nt2 <= 0x3fffffff, "fft_step: batch %d too large for n = 2^%d", batch, p->logn);
          const int g1 = (int) std::min<int64_t>(nt1, ncu), g2 = (int) std::min<int64_t>(nt2, ncu);
          const bool dyn_ok = p->d_ctr && dev_switch_int("FFT_DYN", 1) != 0 && !stream_is_capturing(st);
          unsigned *c1 = (dyn_ok && nt1 >= 4 * g1) ? p->d_ctr : nullptr, *c2 = (dyn_ok && nt2 >= 4 * g2) ? p->d_ctr : nullptr;
          if ((c1 || c2) && p->ctr_stale) {
            TSD_HIP(hipMemsetAsync(p->d_ctr, 0, 256, st));
            p->ctr_base = 0;
            p->ctr_stale = false;
          }
          const unsigned b1 = p->ctr_base, b2 = b1 + (c1 ? (unsigned) nt1 + (unsigned) g1 : 0u);
          int tshift = 0;
          while ((16 << tshift) < C) tshift++;
          if (c1)
            hipLaunchKernelGGL((fft1m_cols_kernel<1, true>), dim3(g1), dim3(1024), F1M_LDS, st, x, z, p->d_w1, p->d_w2, p->d_ta, p->d_td, inverse, 1.0f, zp,
                               (int) nt1, c1, b1, tshift, 0);
          else
            hipLaunchKernelGGL((fft1m_cols_kernel<1, false>), dim3(g1), dim3(1024), F1M_LDS, st, x, z, p->d_w1, p->d_w2, p->d_ta, p->d_td, inverse, 1.0f, zp,
                               (int) nt1, (unsigned *) nullptr, 0u, tshift, 0);
          {
            const int vec = C1 == 32 ? 1 : 2;
            const int64_t threads = (int64_t) batch * 1024 * (1024 / vec);
            const unsigned gp = (unsigned) cdiv(threads, 256);
            if (C1 == 8) hipLaunchKernelGGL((fft_planes_kernel<8, 2>), dim3(gp), dim3(256), 0, st, z, p->d_tp, zp, threads);
            else if (C1 == 16) hipLaunchKernelGGL((fft_planes_kernel<16, 2>), dim3(gp), dim3(256), 0, st, z, p->d_tp, zp, threads);
            else hipLaunchKernelGGL((fft_planes_kernel<32, 1>), dim3(gp), dim3(256), 0, st, z, p->d_tp, zp, threads);
          }
          if (c2)
            hipLaunchKernelGGL((fft1m_cols_kernel<2, true>), dim3(g2), dim3(1024), F1M_LDS, st, z, y, p->d_w1, p->d_w2, p->d_ta, p->d_td, inverse, scale, zp,
                               (int) nt2, c2, b2, 6, lc1);
          else
            hipLaunchKernelGGL((fft1m_cols_kernel<2, false>), dim3(g2), dim3(1024), F1M_LDS, st, z, y, p->d_w1, p->d_w2, p->d_ta, p->d_td, inverse, scale, zp,
                               (int) nt2, (unsigned *) nullptr, 0u, 6, lc1);
          if (const hipError_t le = hipGetLastError(); le != hipSuccess) {
            if (c1 || c2) p->ctr_stale = true;
            return set_err(TSDGPU_ERR_HIP, "fft_step: launch failed: %s", hipGetErrorString(le));
          }
          p->ctr_base = b2 + (c2 ? (unsigned) nt2 + (unsigned) g2 : 0u);
          return TSDGPU_OK;
        }
        if (p->cols2k || p->cols2k_p1) {
          // 2048-point column passes in sixteen-column tiles held in the register file (fft2k_cols_kernel), persistent grids; the
          // 1024-point pass of 2^21 on the 2^20 plan's column kernel; a padded intermediate when both passes can take one
          static const int NCU2 = [] {
            int dev = 0, nn = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void) hipDeviceGetAttribute(&nn, hipDeviceAttributeMultiprocessorCount, dev);
            return nn > 0 ? nn : 256;
          }();
          const bool p1w = p->cols2k && p->N1 == 1024, p1k = p->cols2k_p1, p2k = p->cols2k;
          const int zp = (p2k && (p1w || p1k)) ? p->N1 + 16 : p->N1;
          if (zp != p->N1) {
            rc = p->work.reserve((size_t) batch * p->N2 * zp * sizeof(cpx));
            if (rc) return rc;
            z = p->work.as<cpx>();
          }
          const int nt1 = (p->N2 / 16) * batch, g1 = std::min(nt1, NCU2), nt2 = (p->N1 / 16) * batch, g2 = std::min(nt2, NCU2);
          const bool dyn_ok = p->d_ctr && dev_switch_int("FFT_DYN", 1) != 0 && !stream_is_capturing(st);
          unsigned *c1 = ((p1w || p1k) && dyn_ok && nt1 >= 4 * g1) ? p->d_ctr : nullptr, *c2 = (p2k && dyn_ok && nt2 >= 4 * g2) ? p->d_ctr : nullptr;
          if ((c1 || c2) && p->ctr_stale) {
            TSD_HIP(hipMemsetAsync(p->d_ctr, 0, 256, st));
            p->ctr_base = 0;
            p->ctr_stale = false;
          }
          const unsigned b1 = p->ctr_base, b2 = b1 + (c1 ? (unsigned) nt1 + (unsigned) g1 : 0u);
          if (p1w) {
            int tshift = 0;
            while ((16 << tshift) < p->N2) tshift++;
            if (c1)
              hipLaunchKernelGGL((fft1m_cols_kernel<1, true>), dim3(g1), dim3(1024), F1M_LDS, st, x, z, p->d_w1, p->d_w2, p->d_ta, p->d_td, inverse, 1.0f, zp,
                                 nt1, c1, b1, tshift, 0);
            else
              hipLaunchKernelGGL((fft1m_cols_kernel<1, false>), dim3(g1), dim3(1024), F1M_LDS, st, x, z, p->d_w1, p->d_w2, p->d_ta, p->d_td, inverse, 1.0f, zp,
                                 nt1, (unsigned *) nullptr, 0u, tshift, 0);
          } else if (p1k) {
            if (c1)
              hipLaunchKernelGGL((fft2k_cols_kernel<1, true>), dim3(g1), dim3(512), F1M_LDS, st, x, z, p->d_w1, p->d_w2, p->d_w2k, p->N2, p->N2, zp,
                                 inverse, 1.0f, nt1, c1, b1, p->d_ta, p->d_td);
            else
              hipLaunchKernelGGL((fft2k_cols_kernel<1, false>), dim3(g1), dim3(512), F1M_LDS, st, x, z, p->d_w1, p->d_w2, p->d_w2k, p->N2, p->N2, zp,
                                 inverse, 1.0f, nt1, (unsigned *) nullptr, 0u, p->d_ta, p->d_td);
          } else {
            launch(1, x, z, p->d_tw1, p->N1, p->logN1, p->N2, 1.0f);
          }
          if (p2k) {
            if (c2)
              hipLaunchKernelGGL((fft2k_cols_kernel<2, true>), dim3(g2), dim3(512), F1M_LDS, st, z, y, p->d_w1, p->d_w2, p->d_w2k, p->N1, zp, p->N1,
                                 inverse, scale, nt2, c2, b2, (const cpx *) nullptr, (const cpx *) nullptr);
            else
              hipLaunchKernelGGL((fft2k_cols_kernel<2, false>), dim3(g2), dim3(512), F1M_LDS, st, z, y, p->d_w1, p->d_w2, p->d_w2k, p->N1, zp, p->N1,
                                 inverse, scale, nt2, (unsigned *) nullptr, 0u, (const cpx *) nullptr, (const cpx *) nullptr);
          } else {
            launch(2, z, y, p->d_tw2, p->N2, p->logN2, p->N1, scale);
          }
          if (const hipError_t le = hipGetLastError(); le != hipSuccess) {
            if (c1 || c2) p->ctr_stale = true;
            return set_err(TSDGPU_ERR_HIP, "fft_step: launch failed: %s", hipGetErrorString(le));
          }
          p->ctr_base = b2 + (c2 ? (unsigned) nt2 + (unsigned) g2 : 0u);
          return TSDGPU_OK;
        }
        launch(1, x, z, p->d_tw1, p->N1, p->logN1, p->N2, 1.0f);
        TSD_HIP(hipGetLastError());
        launch(2, z, y, p->d_tw2, p->N2, p->logN2, p->N1, scale);
        TSD_HIP(hipGetLastError());
        return TSDGPU_OK;
      }
      // x viewed [N1][N2]: pass 1 = column FFTs (length N1) + twiddle, stored transposed [N2][N1]
      int tile1 = COL_TILE, tile2 = COL_TILE;
      while ((size_t) tile1 * (p->N1 + 1) * sizeof(cpx) > 150 * 1024) tile1 >>= 1;
      while ((size_t) tile2 * (p->N2 + 1) * sizeof(cpx) > 150 * 1024) tile2 >>= 1;
      hipLaunchKernelGGL(fft_cols_kernel<true>, dim3((unsigned) cdiv(p->N2, tile1), (unsigned) batch), dim3(FFT_THREADS),
                         (size_t) tile1 * (p->N1 + 1) * sizeof(cpx), st, x, z, p->d_tw1, p->N1, p->logN1, p->N2,
                         p->d_thi, p->d_tlo, inverse, 1.0f, tile1);
      TSD_HIP(hipGetLastError());
      // z viewed [N2][N1]: pass 2 = column FFTs (length N2), natural store: y[k2 * N1 + k1]
      hipLaunchKernelGGL(fft_cols_kernel<false>, dim3((unsigned) cdiv(p->N1, tile2), (unsigned) batch), dim3(FFT_THREADS),
                         (size_t) tile2 * (p->N2 + 1) * sizeof(cpx), st, z, y, p->d_tw2, p->N2, p->logN2, p->N1,
                         p->d_thi, p->d_tlo, inverse, scale, tile2);
      TSD_HIP(hipGetLastError());
      return TSDGPU_OK;
    }
    case tsdgpu_fft::SMOOTH: {
      const int tpt = p->mr_tpt, T = std::max(1, 256 / tpt), threads = T * tpt;
      const size_t lds = (size_t) T * (n + n / 16 + 1) * sizeof(cpx);
      hipLaunchKernelGGL(fft_mr_kernel, dim3((unsigned) cdiv(batch, T)), dim3(threads), lds, st, x, y, p->d_rot, p->mr, n, tpt,
                         inverse, 1.0f / std::sqrt((float) n), batch);
      TSD_HIP(hipGetLastError());
      return TSDGPU_OK;
    }
    case tsdgpu_fft::ODDPOW2: {
      const int m = p->mix_m, P = p->mix_P, tpt = P / 16, per = m * tpt;
      const int T = std::max(1, 256 / per), threads = T * per;
      const size_t lds = (size_t) T * m * (P + P / 16) * sizeof(cpx);
      const int r0 = 1 << ((p->logn & 3) == 0 ? 4 : (p->logn & 3));
      const unsigned grid = (unsigned) cdiv(batch, T);
      const float sc = 1.0f / std::sqrt((float) n);
#define OP_LAUNCH(R, MM) hipLaunchKernelGGL((fft_oddpow2_kernel<R, MM>), dim3(grid), dim3(threads), lds, st, x, y, p->d_tw, p->d_wm, p->d_rot, m, P, tpt, inverse, sc, batch)
#define OP_PICK(R) do { if (m <= 3) OP_LAUNCH(R, 4); else if (m <= 7) OP_LAUNCH(R, 8); else OP_LAUNCH(R, 16); } while (0)
      if (r0 == 16) OP_PICK(16); else if (r0 == 8) OP_PICK(8); else if (r0 == 4) OP_PICK(4); else OP_PICK(2);
#undef OP_PICK
#undef OP_LAUNCH
      TSD_HIP(hipGetLastError());
      return TSDGPU_OK;
    }
    case tsdgpu_fft::MIXED: {
      const int m = p->mix_m, P = p->mix_P;
      TSD_CHECK(batch <= 65535, "fft_step: batch %d too large for the mixed-radix path (max 65535 per call)", batch);
      int rc = p->work.reserve((size_t) total * sizeof(cpx));
      if (rc) return rc;
      cpx *z = p->work.as<cpx>();
      const int64_t tot1 = (int64_t) batch * P;
      const float s2 = 1.0f / std::sqrt((float) P);
      // pass 1: Z[b][r][k1] = W_n^(r k1) * (unitary m-point DFT of x[b][r + P i])
      if (m <= 31) {
        const unsigned g1 = (unsigned) cdiv(tot1, 256);
        const float s1 = 1.0f / std::sqrt((float) m);
#define ODD_LAUNCH(M) hipLaunchKernelGGL((fft_odd_dft_kernel<M>), dim3(g1), dim3(256), (size_t) (32 + 256 * m) * sizeof(cpx), st, x, z, p->d_wm, p->d_rot, m, P, inverse, s1, tot1)
        if (m <= 8) ODD_LAUNCH(8); else if (m <= 16) ODD_LAUNCH(16); else ODD_LAUNCH(32);
#undef ODD_LAUNCH
        TSD_HIP(hipGetLastError());
      } else if (bluestein_fusable(p->sub, P)) {
        return launch_bluestein(p->sub, x, y, tot1, inverse, inverse, P, p->d_rot, st, true);      // both passes in one kernel
      } else {
        rc = launch_bluestein(p->sub, x, z, tot1, inverse, inverse, P, p->d_rot, st);
        if (rc) return rc;
      }
      // pass 2: P-point column FFTs over r, natural-order store
      if (P < 16) {
        const int64_t tot2 = (int64_t) batch * m;
        const unsigned g2 = (unsigned) cdiv(tot2, 256);
        if (P == 2) hipLaunchKernelGGL((fft_smallcols_kernel<2>), dim3(g2), dim3(256), 0, st, z, y, m, inverse, s2, tot2);
        else if (P == 4) hipLaunchKernelGGL((fft_smallcols_kernel<4>), dim3(g2), dim3(256), 0, st, z, y, m, inverse, s2, tot2);
        else hipLaunchKernelGGL((fft_smallcols_kernel<8>), dim3(g2), dim3(256), 0, st, z, y, m, inverse, s2, tot2);
        TSD_HIP(hipGetLastError());
        return TSDGPU_OK;
      }
      const int tpt = P / 16;
      // ragged tiles: any column count works; no wider than the m live columns
      int CT = std::min(std::min(16, m), 1024 / tpt);
      while ((size_t) CT * (P + P / 16 + 1) * sizeof(cpx) > 150 * 1024) CT--;
      const size_t lds = (size_t) CT * (P + P / 16 + 1) * sizeof(cpx);
      const dim3 grid((unsigned) cdiv(m, CT), (unsigned) batch), blk((unsigned) (CT * tpt));
      const int r0 = 1 << ((p->logn & 3) == 0 ? 4 : (p->logn & 3));
#define C16_LAUNCH(R) hipLaunchKernelGGL((fft_cols16_kernel<2, R>), grid, blk, lds, st, z, y, p->d_tw, P, tpt, m, CT, nullptr, nullptr, inverse, s2, 1, m)
      if (r0 == 16) C16_LAUNCH(16); else if (r0 == 8) C16_LAUNCH(8); else if (r0 == 4) C16_LAUNCH(4); else C16_LAUNCH(2);
#undef C16_LAUNCH
      TSD_HIP(hipGetLastError());
      return TSDGPU_OK;
    }
    case tsdgpu_fft::EVEN: {
      int rc = p->work.reserve((size_t) total * sizeof(cpx));
      if (rc) return rc;
      rc = p->work2.reserve((size_t) total * sizeof(cpx));
      if (rc) return rc;
      cpx *t1 = p->work.as<cpx>(), *t2 = p->work2.as<cpx>();
      hipLaunchKernelGGL(fft_split_eo_kernel, dim3(blocks_for(total)), dim3(256), 0, st, x, t1, n, total);
      TSD_HIP(hipGetLastError());
      rc = step_device(p->sub, t1, t2, 2 * batch, forward, st);
      if (rc) return rc;
      hipLaunchKernelGGL(fft_combine_eo_kernel, dim3(blocks_for(total)), dim3(256), 0, st, t2, y, p->d_rot, n, inverse,
                         total);
      TSD_HIP(hipGetLastError());
      return TSDGPU_OK;
    }
    case tsdgpu_fft::ODD: {
      const int n2 = p->n2;
      if (p->blu_fused) {                                       // (in place is fine: a workgroup loads its whole transforms before it stores)
        const int rc2 = launch_bluestein(p, x, y, batch, inverse, 0, 1, nullptr, st);
        return rc2;
      }
      const int64_t tot2 = (int64_t) n2 * batch;
      int rc = p->work.reserve((size_t) tot2 * sizeof(cpx));
      if (rc) return rc;
      rc = p->work2.reserve((size_t) tot2 * sizeof(cpx));
      if (rc) return rc;
      cpx *a = p->work.as<cpx>(), *b = p->work2.as<cpx>();
      hipLaunchKernelGGL(czt_pre_kernel, dim3(blocks_for(tot2)), dim3(256), 0, st, x, a, p->d_chirp, n, n2, tot2);
      TSD_HIP(hipGetLastError());
      rc = step_device(p->sub, a, b, batch, 1, st);
      if (rc) return rc;
      hipLaunchKernelGGL(czt_mul_kernel, dim3(blocks_for(tot2)), dim3(256), 0, st, b, p->d_xc, n2, tot2);
      TSD_HIP(hipGetLastError());
      rc = step_device(p->sub, b, a, batch, 0, st);
      if (rc) return rc;
      const float g = std::sqrt((float) n2) / std::sqrt((float) n);
      hipLaunchKernelGGL(czt_post_kernel, dim3(blocks_for(total)), dim3(256), 0, st, a, y, p->d_chirp, n, n2, g, inverse,
                         total);
      TSD_HIP(hipGetLastError());
      return TSDGPU_OK;
    }
  }
  return TSDGPU_OK;
}

}  // namespace

namespace tsdgpu {
const float2 *fft_s16_twiddles(const tsdgpu_fft *p) { return (p && p->kind == tsdgpu_fft::POW2_S16) ? p->d_tw : nullptr; }
// psd_welch at sizes whose plan is the wave-level Bluestein (odd N <= 511, or N = m 2^p with such an odd part and p <= 4): segment s
// = x[s pas .. s pas + N) times the window goes through the transform where it is, and the |X|^2 (unitary scaling) come out summed per
// workgroup: `rows` partial rows of N floats in pw (pw == NULL: only the count, for the caller's allocation).  0: not this plan.
int fft_blu_framed_launch(const tsdgpu_fft *p, const float2 *x, int64_t pas, const float *win, int64_t nseg, float *pw, int64_t cap_rows,
                          int64_t *rows, void *stream)
{
  *rows = 0;
  if (!p || nseg <= 0) return TSDGPU_OK;
  hipStream_t st = (hipStream_t) stream;
  if (p->kind == tsdgpu_fft::ODD && blu_wave_fits(p, 1, false))
    return launch_blu_wave(p, x, nullptr, pw, win, pas, nseg, 0, 0, 1, nullptr, st, false, rows, cap_rows);
  if (p->kind == tsdgpu_fft::MIXED && p->sub && bluestein_fusable(p->sub, p->mix_P) && blu_wave_fits(p->sub, p->mix_P, true))
    return launch_blu_wave(p->sub, x, nullptr, pw, win, pas, nseg * p->mix_P, 0, 0, p->mix_P, p->d_rot, st, true, rows, cap_rows);
  return TSDGPU_OK;
}

}

extern "C" {

int tsdgpu_fft_create(tsdgpu_fft **out, int n, int batch_hint)
{
  (void) batch_hint;
  TSD_CHECK(out != nullptr, "fft_create: out is NULL");
  *out = nullptr;
  TSD_CHECK(n >= 1, "fft_create: n must be >= 1 (got %d)", n);
  TSD_CHECK(n <= (1 << 28), "fft_create: n = %d too large (limit 2^28)", n);
  return plan_create(out, n);
}

int tsdgpu_fft_step(tsdgpu_fft *p, const void *x, void *y, int batch, int forward, void *stream)
{
  TSD_CHECK(p != nullptr, "fft_step: NULL plan");
  TSD_CHECK(batch >= 0, "fft_step: negative batch");
  if (batch == 0) return TSDGPU_OK;
  TSD_CHECK(x != nullptr && y != nullptr, "fft_step: NULL buffer");
  hipStream_t st = (hipStream_t) stream;
  const size_t bytes = (size_t) p->n * batch * sizeof(cpx);
  const void *dx = nullptr;
  void *dy = nullptr;
  bool staged = false;
  std::lock_guard<std::mutex> lock(p->order.mu);
  int rc = p->order.enter(st);
  if (rc) return rc;
  // a large batch on HOST vectors goes through in chunks of whole transforms: H2D of chunk i + 1, the transforms of chunk i
  // and D2H of chunk i - 1 overlap (common.hpp: pipelined_host_step_var); y may be x
  if (batch >= 2 && bytes >= PIPE_MIN_BYTES && host_pipe_enabled() && !is_device_ptr(x) && !is_device_ptr(y) &&
      (x == y || !host_ranges_overlap(x, bytes, y, bytes))) {
    const int n = p->n;
    rc = pipelined_host_step_var(
        x, (int64_t) n * batch, sizeof(cpx), y, sizeof(cpx), nullptr, n, st, [](int64_t c) { return c; },
        [p, n, forward](const void *cx, void *cy, int64_t cnt, int64_t, int64_t *got, hipStream_t q) {
          *got = cnt;
          return step_device(p, (const cpx *) cx, (cpx *) cy, (int) (cnt / n), forward, q);
        });
    if (rc) return rc;
    return p->order.leave(st);
  }
  rc = stage_in(x, bytes, p->in_stage, st, &dx);
  if (rc) return rc;
  rc = stage_out(y, bytes, p->out_stage, &dy, &staged);
  if (rc) return rc;
  rc = step_device(p, (const cpx *) dx, (cpx *) dy, batch, forward, st);
  if (rc) return rc;
  rc = finish_out(y, bytes, dy, staged, st);
  if (rc) return rc;
  return p->order.leave(st);
}

int tsdgpu_fft_size(const tsdgpu_fft *p) { return p ? p->n : -1; }

int tsdgpu_fft_destroy(tsdgpu_fft *p)
{
  plan_destroy(p);
  return TSDGPU_OK;
}

int tsdgpu_fftshift(const void *x, void *y, int n, int data_type, void *stream)
{
  TSD_CHECK(n >= 0, "fftshift: negative length");
  if (n == 0) return TSDGPU_OK;
  TSD_CHECK(x != nullptr && y != nullptr && x != y, "fftshift: needs distinct non-NULL buffers");
  hipStream_t st = (hipStream_t) stream;
  const size_t bytes = (size_t) n * dtype_size(data_type);
  DevBuf a, b;
  const void *dx = nullptr;
  void *dy = nullptr;
  bool staged = false;
  int rc = stage_in(x, bytes, a, st, &dx);
  if (!rc) rc = stage_out(y, bytes, b, &dy, &staged);
  if (!rc) {
    if (data_type == TSDGPU_C64)
      hipLaunchKernelGGL(fftshift_kernel<float2>, dim3((unsigned) cdiv(n, 256)), dim3(256), 0, st, (const float2 *) dx,
                         (float2 *) dy, n);
    else
      hipLaunchKernelGGL(fftshift_kernel<float>, dim3((unsigned) cdiv(n, 256)), dim3(256), 0, st, (const float *) dx,
                         (float *) dy, n);
    if (hipGetLastError() != hipSuccess) rc = set_err(TSDGPU_ERR_HIP, "fftshift launch failed");
  }
  if (!rc) rc = finish_out(y, bytes, dy, staged, st);
  if (rc == TSDGPU_OK && !staged) { /* async on the caller's stream */ }
  else (void) hipStreamSynchronize(st);
  a.release();
  b.release();
  return rc;
}

/* ---- real FFT ---------------------------------------------------------------------------- */
struct tsdgpu_rfft {
  int n = 0;
  tsdgpu_fft *sub = nullptr;      // n/2-point complex plan (even n) or n-point plan (odd n)
  cpx *d_rot = nullptr;           // tfr_rotation(n): W_n^i, double recurrence rounded to float
  DevBuf work, in_stage, out_stage;
  StepOrder order;
};

int tsdgpu_rfft_create(tsdgpu_rfft **out, int n)
{
  TSD_CHECK(out != nullptr, "rfft_create: out is NULL");
  *out = nullptr;
  TSD_CHECK(n >= 1, "rfft_create: n must be >= 1 (got %d)", n);
  tsdgpu_rfft *p = new tsdgpu_rfft();
  p->n = n;
  int rc = TSDGPU_OK;
  if ((n & 1) == 0) {
    rc = plan_create(&p->sub, n / 2);
    if (!rc) {
      std::vector<cpx> rot((size_t) n);
      const double PI = 3.14159265358979323846;
      double rr = 1.0, ri = 0.0;
      const double wr = std::cos(-2 * PI / n), wi = std::sin(-2 * PI / n);
      for (int i = 0; i < n; i++) {
        rot[i] = make_float2((float) rr, (float) ri);
        const double tr = rr * wr - ri * wi, ti = rr * wi + ri * wr;
        rr = tr; ri = ti;
      }
      rc = upload(&p->d_rot, rot);
    }
  } else {
    rc = plan_create(&p->sub, n);
  }
  if (rc) {
    tsdgpu_rfft_destroy(p);
    return rc;
  }
  *out = p;
  return TSDGPU_OK;
}

}  // extern "C"
// the transform of `batch` real vectors on device pointers (dx: n floats each, dy: n complex each)
static int rfft_device(tsdgpu_rfft *p, const void *dx, void *dy, int batch, hipStream_t st)
{
  const int n = p->n;
  int rc = TSDGPU_OK;
  if ((n & 1) == 0 && p->sub->kind == tsdgpu_fft::POW2_S16 && dev_switch("RFFT_TWO_PASS") == nullptr) {
    // half-size Stockham transform with the untangling fused into its store: one pass over HBM
    const int h = n / 2, tpt = h / 16, threads = std::max(256, tpt), T = threads / tpt;
    const size_t lds = (size_t) T * (h + h / 16) * sizeof(cpx);
    const unsigned grid = (unsigned) cdiv(batch, T);
    const int r0 = 1 << ((p->sub->logn & 3) == 0 ? 4 : (p->sub->logn & 3));
    const float scale = 1.0f / std::sqrt((float) h);
#define S16R_LAUNCH(R) hipLaunchKernelGGL((fft_s16_kernel<R>), dim3(grid), dim3(threads), lds, st, (const cpx *) dx, (cpx *) dy, p->sub->d_tw, h, tpt, 0, scale, batch, (const cpx *) p->d_rot)
    if (r0 == 16) S16R_LAUNCH(16); else if (r0 == 8) S16R_LAUNCH(8); else if (r0 == 4) S16R_LAUNCH(4); else S16R_LAUNCH(2);
#undef S16R_LAUNCH
    TSD_HIP(hipGetLastError());
  } else if ((n & 1) == 0) {
    const int h = n / 2;
    rc = p->work.reserve((size_t) h * batch * sizeof(cpx));
    if (rc) return rc;
    cpx *xt = p->work.as<cpx>();
    rc = step_device(p->sub, (const cpx *) dx, xt, batch, 1, st);     // the real pairs, read as complex
    if (rc) return rc;
    const int64_t total = (int64_t) batch * (h + 1);
    hipLaunchKernelGGL(rfft_untangle_kernel, dim3((unsigned) cdiv(total, 256)), dim3(256), 0, st, xt, (cpx *) dy, p->d_rot, n,
                       total);
    TSD_HIP(hipGetLastError());
  } else {
    // odd n: the reference simply transforms x.as_complex() (fourier.cc:348-352)
    const int64_t total = (int64_t) batch * n;
    hipLaunchKernelGGL(real_to_complex_kernel, dim3((unsigned) cdiv(total, 256)), dim3(256), 0, st, (const float *) dx,
                       (cpx *) dy, total);
    TSD_HIP(hipGetLastError());
    rc = step_device(p->sub, (const cpx *) dy, (cpx *) dy, batch, 1, st);
    if (rc) return rc;
  }
  return TSDGPU_OK;
}
extern "C" {

int tsdgpu_rfft_step(tsdgpu_rfft *p, const void *x, void *y, int batch, void *stream)
{
  TSD_CHECK(p != nullptr, "rfft_step: NULL plan");
  TSD_CHECK(batch >= 0, "rfft_step: negative batch");
  if (batch == 0) return TSDGPU_OK;
  TSD_CHECK(x != nullptr && y != nullptr && x != y, "rfft_step: needs distinct non-NULL buffers");
  hipStream_t st = (hipStream_t) stream;
  const int n = p->n;
  const size_t in_bytes = (size_t) n * batch * sizeof(float), out_bytes = (size_t) n * batch * sizeof(cpx);
  const void *dx = nullptr;
  void *dy = nullptr;
  bool staged = false;
  std::lock_guard<std::mutex> lock(p->order.mu);
  int rc = p->order.enter(st);
  if (rc) return rc;
  // a large batch on HOST vectors: chunks of whole transforms through the staging pipeline (see tsdgpu_fft_step)
  if (batch >= 2 && out_bytes >= PIPE_MIN_BYTES && host_pipe_enabled() && !is_device_ptr(x) && !is_device_ptr(y) &&
      !host_ranges_overlap(x, in_bytes, y, out_bytes)) {
    rc = pipelined_host_step_var(
        x, (int64_t) n * batch, sizeof(float), y, sizeof(cpx), nullptr, n, st, [](int64_t c) { return c; },
        [p, n](const void *cx, void *cy, int64_t cnt, int64_t, int64_t *got, hipStream_t q) {
          *got = cnt;
          return rfft_device(p, cx, cy, (int) (cnt / n), q);
        });
    if (rc) return rc;
    return p->order.leave(st);
  }
  rc = stage_in(x, in_bytes, p->in_stage, st, &dx);
  if (rc) return rc;
  rc = stage_out(y, out_bytes, p->out_stage, &dy, &staged);
  if (rc) return rc;
  rc = rfft_device(p, dx, dy, batch, st);
  if (rc) return rc;
  rc = finish_out(y, out_bytes, dy, staged, st);
  if (rc) return rc;
  return p->order.leave(st);
}

int tsdgpu_rfft_destroy(tsdgpu_rfft *p)
{
  if (!p) return TSDGPU_OK;
  if (p->sub) plan_destroy(p->sub);
  if (p->d_rot) (void) hipFree(p->d_rot);
  p->work.release();
  p->in_stage.release();
  p->out_stage.release();
  p->order.release();
  delete p;
  return TSDGPU_OK;
}

}  // extern "C"
