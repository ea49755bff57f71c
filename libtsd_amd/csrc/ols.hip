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
#include <cstdlib>

namespace tsdgpu {

using namespace w1024;
constexpr int OLS_N = 1024;
// complex values of the kernel: float2 with scalar fp32 arithmetic; -DOLS_SCALAR=0 selects the
// packed (VOP3P) flavour of fft1024_wave.hpp -- half the VALU instructions, parity-green, but
// measured 2 % SLOWER (0.2313 vs 0.2260 ms, three interleaved runs): the kernel runs at the
// speed of its memory skeleton, so the arithmetic is not what it waits for
#ifndef OLS_RUN_DEFAULT   // blocks per run of a wave (ols_body): 1 = every block loads its whole overlap again
#define OLS_RUN_DEFAULT 2
#endif
#ifndef OLS_DYN_DEFAULT   // counters of the dynamic hand-out (OlsDyn); 0 = static partition
#define OLS_DYN_DEFAULT 16
#endif
constexpr int OLS_MAX_CTR = 32;
// One wave per workgroup, the block's 46 per-lane constants (twiddles of the two radix-16 stages, the response H) in registers for
// the wave's lifetime: 200 VGPRs, 2 waves per SIMD.  Measured against it and not kept (profiles/EXPERIMENTS.md, round 3): four
// waves per workgroup with the tables in an LDS image (0.2081-0.2087 against 0.2017-0.2032 ms), twiddles generated from four
// table entries per stage (0.2154-0.267), the packed VOP3P arithmetic of fft1024_wave.hpp (2 % slower), 16-B and non-temporal
// global accesses.
using cv = cpx;
__device__ __forceinline__ cv mkv(float a, float b) { return Make<cv>::of(a, b); }

// EDGE = false: block fully inside [0, n) on both the input and the output side -- no guards,
// straight-line code (lets hipcc use counted vmcnt waits so the prefetch and the previous
// block's stores stay in flight).  EDGE = true: guarded loads (history / zero fill) and stores.
constexpr bool NT = false;      // (non-temporal accesses: measured slower, the helpers below stay for the ragged paths' signature)
__device__ __forceinline__ cv ntload(const cv *p)
{
  v2f t = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(p));
  return mkv(t.x, t.y);
}
__device__ __forceinline__ void ntstore(cv *p, cv v)
{
  v2f t = {v.x, v.y};
  __builtin_nontemporal_store(t, reinterpret_cast<v2f *>(p));
}
// REAL data: two consecutive real blocks are packed as the real and imaginary part of one
// complex block (the taps being real, conv(h, a + j b) = conv(h, a) + j conv(h, b)), so block
// index b then addresses the pair of real blocks (2b, 2b+1).
template <bool EDGE>
__device__ __forceinline__ void ols_fetch_real(cv (&v)[16], const float *__restrict__ x, const float *__restrict__ hist,
                                               int histlen, int Km1, int L, int64_t n, int64_t b, int lane)
{
  const int64_t ga = 2 * b * (int64_t) L - Km1, gb = ga + L;
  if (!EDGE) {
    const float *xa = x + ga, *xb = x + gb;
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = mkv(xa[64 * r + lane], xb[64 * r + lane]);
  } else {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int64_t g1 = ga + 64 * r + lane, g2 = gb + 64 * r + lane;
      v[r] = mkv(g1 < 0 ? hist[histlen + g1] : (g1 < n ? x[g1] : 0.f), g2 < 0 ? hist[histlen + g2] : (g2 < n ? x[g2] : 0.f));
    }
  }
}
template <bool EDGE>
__device__ __forceinline__ void ols_fetch(cv (&v)[16], const cv *__restrict__ x, const cv *__restrict__ hist,
                                          int histlen, int Km1, int L, int64_t n, int64_t b, int lane)
{
  const int64_t g0 = b * (int64_t) L - Km1;   // first input of block b
  if (!EDGE) {
    const cv *xb = x + g0;
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = NT ? ntload(xb + 64 * r + lane) : xb[64 * r + lane];
  } else {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int64_t g = g0 + 64 * r + lane;
      v[r] = g < 0 ? hist[histlen + g] : (g < n ? x[g] : mkv(0.f, 0.f));
    }
  }
}

// 2 waves per SIMD (256-register budget): twiddles + H resident (96 VGPRs), the block being
// transformed (32) and the NEXT block's samples already in flight (32) -- the wave's own
// prefetch, not occupancy, hides the HBM latency.
//
// Blocks [b_lo, b_hi) are processed with block b = b_lo + slot + i*G.  The EDGE variant is
// launched with G = 1 per edge block.
// R0 > 0 (interior launches, overlap of exactly R0 rows of 64 samples): a wave walks RUNS of R consecutive blocks and the
// R0 overlap rows of a block inside a run are the last R0 input rows of the block before it -- copied from registers
// before the transform overwrites them instead of being loaded again.  Every sample then crosses the L2 1 + R0/(16 - R0)/R
// times instead of 1 + R0/(16 - R0) (K = 127: 1.036 with R = 4 against 1.143), while the grid still sweeps one contiguous
// span of G*R blocks per round (fully chunked streams -- one per wave for the whole call -- measured slower).
// R0 = 0: every block loads its 16 rows (any overlap; R is 1).
//
// DYN: the runs are handed out DYNAMICALLY instead of run = slot + round * G.  A persistent grid with a static partition
// streams 6-9 % below a non-persistent launch of the same copy (scripts/ubench/copy_shapes.hip: 5.5 against 6.05 TB/s --
// nothing balances the waves, they drift apart and the slowest one ends the launch); pulling the next unit from a few
// counters brings the copy to 5.9.  NC counters on their own 128-B lines; wave w pulls from counter (w / 8) % NC, so the
// pullers of one counter sit on all 8 XCDs; a pulled value v stands for unit v * NC + c.  The counters are never reset:
// every launch starts from `base` (host-tracked) and advances each counter by exactly Q + G / NC -- Q = ceil(units / NC)
// values inside the quota plus one failing pull per wave -- so the host knows the next launch's base without a memset.
// The pull for the NEXT run is issued at the start of the current one: its latency hides under R blocks.
struct OlsDyn {
  unsigned *ctr;
  unsigned base, Q;
  int NC;
  int64_t nunits;
};
struct RegTab {
  cv v[16];
  __device__ __forceinline__ const cv &operator[](int r) const { return v[r]; }
};

template <bool EDGE, bool REAL, int R0, bool DYN>
__device__ __forceinline__ void ols_body(cv *lds, const void *__restrict__ xv, const void *__restrict__ histv,
                                         void *__restrict__ yv, const cv *__restrict__ Hreg,
                                         const cv *__restrict__ TW1, const cv *__restrict__ TW2, int Km1,
                                         int histlen, int L, int64_t n, int64_t b_lo, int64_t b_hi, int64_t G,
                                         int64_t w, int64_t cgrp, int R, OlsDyn dyn)
{
  const int lane = threadIdx.x & 63;
  const cv *x = (const cv *) xv, *hist = (const cv *) histv;
  cv *y = (cv *) yv;
  const float *xr = (const float *) xv, *histr = (const float *) histv;
  float *yr = (float *) yv;
  RegTab tw1r, tw2r, Hr;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    tw1r.v[r] = TW1[r * 64 + lane];
    tw2r.v[r] = TW2[r * 64 + lane];
    Hr.v[r] = Hreg[r * 64 + lane];
  }
  // One wave per workgroup: its LDS operations execute in order, so exchanging data between
  // lanes needs no s_barrier and -- crucially -- no vmcnt(0) drain (a __syncthreads() would
  // wait for the prefetch loads and the previous block's stores).  Wavefront-scope fences
  // only stop the compiler from moving LDS accesses across the exchange points.
  auto sync = []() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  // Block schedule: at iteration i the grid covers the contiguous span [i*G, (i+1)*G) of
  // blocks (sequential HBM streams), and inside the span workgroup w -- dispatched round-robin
  // over the 8 XCDs, so w % 8 labels its XCD -- takes block (w % 8) * G/8 + w / 8: blocks that
  // share their K-1 overlap samples run on the same XCD at the same time and the second
  // reader hits that XCD's L2.  Placement only affects speed, never results.
  const int64_t slot = (G % 8 == 0) ? (w % 8) * (G / 8) + w / 8 : w;
  const int ctr_c = DYN ? (int) (cgrp % dyn.NC) : 0;     // cgrp: rounds of 8 workgroups (one per XCD)
  // -> the next unit of this wave's counter, or -1 once its quota is spent (exactly one failing pull per wave)
  // (measured, round 4: issuing the pull a block earlier than its value is taken -- what pays in the resampler, whose loop
  // otherwise drains its LDS-DMA at the pull -- costs 2 % here: 0.2057 against 0.2015 ms, three interleaved pairs)
  auto pull = [&]() -> int64_t {
    for (;;) {
      unsigned v = 0;
      if (lane == 0) v = __hip_atomic_fetch_add(dyn.ctr + ctr_c * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v = (unsigned) __builtin_amdgcn_readfirstlane((int) v) - dyn.base;
      if (v >= dyn.Q) return -1;
      const int64_t u = (int64_t) v * dyn.NC + ctr_c;
      if (u < dyn.nunits) return u;                  // (the last row of units may be partial: pull again)
    }
  };
  int64_t unit = EDGE ? 0 : (DYN ? pull() : slot);
  if (unit < 0) return;
  int64_t run0 = EDGE ? b_lo : b_lo + unit * R;      // first block of the wave's current run
  int64_t b = run0;
  if (b >= b_hi) return;

  // One block: prefetch the next block into `nxt`, transform `cur` in place, then store it.
  // Memory-queue discipline (vmcnt retires in issue order and hipcc waits vmcnt(0) around the
  // predicated stores): the wait for the prefetch is forced BEFORE this block's stores are
  // issued -- the loads are a whole block old by then, so it costs nothing -- and afterwards
  // nothing waits on the stores: they drain while the next block is being transformed.
  // nb: the block prefetched while `blk` is transformed; inrun: nb = blk + 1 inside the same run
  auto process = [&](cv (&cur)[16], cv (&nxt)[16], int64_t blk, int64_t nb, bool inrun) {
    const bool more = !EDGE && nb < b_hi;
    // the prefetch of block nb into `nxt`: the rows reused from this block are copied NOW (cur still holds the raw samples),
    // the loads are issued at `prefetch_loads()` -- before the forward transform
    if (more && R0 > 0 && inrun) {
#pragma unroll
      for (int r = 0; r < R0; r++) nxt[r] = REAL ? mkv(cur[(16 - R0 + r) & 15].y, 0.f) : cur[(16 - R0 + r) & 15];
    }
    auto prefetch_loads = [&]() {
      if (!more) return;
      if (R0 > 0 && inrun) {
        if (!REAL) {
          const cv *xb = x + (nb * (int64_t) L - Km1);
#pragma unroll
          for (int r = R0; r < 16; r++) nxt[r] = NT ? ntload(xb + 64 * r + lane) : xb[64 * r + lane];
        } else {
          // pair (2 nb, 2 nb + 1): the first block's overlap is the tail of this pair's second block; the second block's
          // overlap is the tail of the first block being loaded now -- filled in once the loads have landed (below)
          const float *xa = xr + (2 * nb * (int64_t) L - Km1), *xb = xa + L;
#pragma unroll
          for (int r = R0; r < 16; r++) nxt[r] = mkv(xa[64 * r + lane], xb[64 * r + lane]);
        }
      } else if (REAL) ols_fetch_real<EDGE>(nxt, xr, histr, histlen, Km1, L, n, nb, lane);
      else ols_fetch<EDGE>(nxt, x, hist, histlen, Km1, L, n, nb, lane);
    };
    prefetch_loads();
    forward(cur, lds, lane, tw1r, tw2r, sync);
#pragma unroll
    for (int r = 0; r < 16; r++) cur[r] = cmul(cur[r], Hr[r]);
    inverse(cur, lds, lane, tw1r, tw2r, sync);
    sync();   // LDS is reused by the next block
    if (more) {
#pragma unroll
      for (int r = 0; r < 16; r++) asm volatile("" ::"v"(nxt[r]));
      if (REAL && R0 > 0 && inrun) {
#pragma unroll
        for (int r = 0; r < R0; r++) nxt[r].y = nxt[(16 - R0 + r) & 15].x;
      }
    }
    // sample t = 64*r + lane of the circular convolution is output o0 + t - (K-1)
    const int r0 = Km1 >> 6;   // Km1 is a multiple of 64: rows below r0 are overlap, the rest whole
    if (!REAL) {
      const int64_t o0 = blk * (int64_t) L;
      cv *yb = y + (o0 - Km1);
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int t = 64 * r + lane;
        if (!EDGE) {
          if (r >= r0) { if (NT) ntstore(yb + t, cur[r]); else yb[t] = cur[r]; }
        } else {
          if (r >= r0 && o0 + t - Km1 < n) yb[t] = cur[r];
        }
      }
    } else {
      const int64_t oa = 2 * blk * (int64_t) L, ob = oa + L;
      float *ya = yr + (oa - Km1), *yb = yr + (ob - Km1);
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int t = 64 * r + lane;
        if (!EDGE) {
          if (r >= r0) { ya[t] = cur[r].x; yb[t] = cur[r].y; }
        } else {
          if (r >= r0 && oa + t - Km1 < n) ya[t] = cur[r].x;
          if (r >= r0 && ob + t - Km1 < n) yb[t] = cur[r].y;
        }
      }
    }
  };

  cv A[16], B[16];
  if (REAL) ols_fetch_real<EDGE>(A, xr, histr, histlen, Km1, L, n, b, lane);
  else ols_fetch<EDGE>(A, x, hist, histlen, Km1, L, n, b, lane);
  if (EDGE) {
    process(A, B, b, b_hi, false);   // edge launches: one block per wave, no prefetch
    return;
  }
  int i = 0;                        // position of b inside its run
  constexpr int64_t END = INT64_MAX;
  // the run after the current one (requested one run ahead)
  auto unit_after = [&](int64_t u) -> int64_t {
    if (DYN) return pull();
    return b_lo + (u + G) * R < b_hi ? u + G : -1;
  };
  int64_t unit_next = unit_after(unit);
  auto next_of = [&](int64_t &nb, bool &inrun) {
    inrun = i + 1 < R && b + 1 < b_hi;
    nb = inrun ? b + 1 : (unit_next >= 0 ? b_lo + unit_next * R : END);
  };
  auto advance = [&](int64_t nb, bool inrun) {
    if (inrun) i++;
    else {
      i = 0;
      run0 = nb;
      unit = unit_next;
      unit_next = unit_after(unit);
    }
    b = nb;
  };
  for (;;) {
    int64_t nb;
    bool inrun;
    next_of(nb, inrun);
    process(A, B, b, nb, inrun);
    if (nb >= b_hi) break;
    advance(nb, inrun);
    next_of(nb, inrun);
    process(B, A, b, nb, inrun);
    if (nb >= b_hi) break;
    advance(nb, inrun);
  }
}

// One launch per step: waves [0, G) walk the interior blocks, the next `ne` waves take one edge block each (block 0 with the
// history halo, the ragged last block), and the wave after them writes the new history (the last `histlen` samples of
// history ++ x) into the handle's other history buffer.  One wave per workgroup (tables in registers, 2 waves per SIMD).
template <bool REAL, int R0, bool DYN>
__global__ __launch_bounds__(64, 2) void ols_kernel(const void *__restrict__ x, const void *__restrict__ hist,
                                                    void *__restrict__ hist_next, void *__restrict__ y,
                                                    const cpx *__restrict__ Hreg, const cpx *__restrict__ TW1,
                                                    const cpx *__restrict__ TW2, int Km1, int histlen, int L,
                                                    int64_t n, int64_t b_lo, int64_t b_hi, int64_t nblocks, int G,
                                                    int ne, int64_t n_lo, int64_t b_tail, int R, OlsDyn dyn)
{
  __shared__ cv lds[LDS_ELEMS];
  // wave index: workgroups are dealt round-robin over the XCDs, so the waves of workgroups g, g + 8, ... share an XCD; w / 8
  // numbers the rounds of 8 workgroups (the counter choice of the dynamic hand-out)
  const int64_t w = (int64_t) blockIdx.x;
  const int lane = threadIdx.x & 63;
  if (w < G) {
    ols_body<false, REAL, R0, DYN>(lds, x, hist, y, (const cv *) Hreg, (const cv *) TW1, (const cv *) TW2, Km1, histlen, L, n, b_lo, b_hi, G,
                                       w, (int64_t) blockIdx.x / 8, R, dyn);
  } else if (w < G + ne) {
    // edge items: [0, n_lo) need the history halo, [b_tail, nblocks) are ragged at the end
    const int64_t b = (w - G) < n_lo ? (int64_t) (w - G) : b_tail + (w - G - n_lo);
    ols_body<true, REAL, 0, false>(lds, x, hist, y, (const cv *) Hreg, (const cv *) TW1, (const cv *) TW2, Km1, histlen, L, n, b, nblocks, 1, 0, 0, 1, dyn);
  } else if (w == G + ne) {
    for (int i = lane; i < histlen; i += 64) {
      const int64_t g = n - histlen + i;
      if (REAL) ((float *) hist_next)[i] = g < 0 ? ((const float *) hist)[histlen + g] : ((const float *) x)[g];
      else ((cpx *) hist_next)[i] = g < 0 ? ((const cpx *) hist)[histlen + g] : ((const cpx *) x)[g];
    }
  }
}

bool ols_preferred(const tsdgpu_fir *f)
{
  // the direct kernel is HBM-bound below ~48 taps (complex data; measured crossover for real data,
  // whose blocks are packed two per complex FFT: ~40 taps); the 1024-point wave block loses efficiency as
  // L = 1024 - roundup(K-1, 64) shrinks, so from L = 512 on larger blocks take over:
  // K = 514 .. 12289: the long-filter plan (ols_long.hip, blocks of 4096..16384 on the Stockham engine;
  // measured crossover with the 1024-point wave blocks: K = 513 -> 0.313 ms here, 0.294 ms there)
  if (ols_long_supported(f)) return true;
  // (complex data, real taps, 2^26 samples, round 2: direct 0.186 / 0.192 / 0.234 ms at 16 / 48 / 64 taps against 0.219 ms
  // overlap-save whatever the count: the crossover sits at ~57 taps)
  if (f->data_type == TSDGPU_C64) return f->K >= (f->tap_type == TSDGPU_F32 ? 57 : 48) && f->K <= 513;
  return f->tap_type == TSDGPU_F32 && f->K >= 40 && f->K <= 513;
}

int ols_plan_create(tsdgpu_fir *f)
{
  if (ols_long_supported(f)) return ols_long_plan_create(f);
  if (f->K > OLS_N - 63) {
    // outside the block-FFT kernel's envelope: serve the request with the direct kernel
    f->method = TSDGPU_FIR_DIRECT;
    return TSDGPU_OK;
  }
  const int N = OLS_N, K = f->K;
  f->ols_N = N;
  f->ols_L = N - (int) (cdiv(K - 1, 64) * 64);   // see ols_kernel: overlap rounded up to whole 64-sample rows
  // H[k] = sum_n h[n] exp(-2 pi i k n / N), in double, then /N and register order
  std::vector<double> hr(K), hi(K, 0.0);
  if (f->tap_type == TSDGPU_F32) {
    const float *t = (const float *) f->taps_host.data();
    for (int i = 0; i < K; i++) hr[i] = t[i];
  } else {
    const float *t = (const float *) f->taps_host.data();
    for (int i = 0; i < K; i++) { hr[i] = t[2 * i]; hi[i] = t[2 * i + 1]; }
  }
  // a radix-2 transform in double (the plan of a one-shot filtrer() is built per call: the K x N sums of the definition
  // took 100 us of its 185), the unit roots and the engine's twiddles made once per process, ONE upload
  struct Tables {
    std::vector<double> c, s;
    std::vector<cpx> tw12;
    Tables() : c(OLS_N / 2), s(OLS_N / 2), tw12(2 * OLS_N)
    {
      const double PI = 3.14159265358979323846;
      for (int i = 0; i < OLS_N / 2; i++) { c[i] = std::cos(2 * PI * i / OLS_N); s[i] = -std::sin(2 * PI * i / OLS_N); }
      fill_twiddles(tw12.data(), tw12.data() + OLS_N);
    }
  };
  static const Tables *tb = new Tables();
  std::vector<double> ar(N, 0.0), ai(N, 0.0);
  int logn = 0;
  while ((1 << logn) < N) logn++;
  for (int i = 0; i < K; i++) {                         // bit-reversed placement of the zero-padded taps
    int rv = 0;
    for (int b = 0; b < logn; b++) rv |= ((i >> b) & 1) << (logn - 1 - b);
    ar[rv] = hr[i];
    ai[rv] = hi[i];
  }
  for (int len = 2; len <= N; len <<= 1) {
    const int half = len / 2, step = N / len;
    for (int base = 0; base < N; base += len)
      for (int j = 0; j < half; j++) {
        const double wr = tb->c[(size_t) j * step], wi = tb->s[(size_t) j * step];
        const int a = base + j, b = a + half;
        const double tr = ar[b] * wr - ai[b] * wi, ti = ar[b] * wi + ai[b] * wr;
        ar[b] = ar[a] - tr; ai[b] = ai[a] - ti;
        ar[a] += tr; ai[a] += ti;
      }
  }
  std::vector<cpx> img(3 * (size_t) N);
  for (int lane = 0; lane < 64; lane++)
    for (int r = 0; r < 16; r++) {
      const int k = freq_index(lane, r);
      img[r * 64 + lane] = mk((float) (ar[k] / N), (float) (ai[k] / N));
    }
  std::copy(tb->tw12.begin(), tb->tw12.end(), img.begin() + N);
  const size_t bytes = (size_t) N * sizeof(cpx);
  // (the work counters of the dynamic hand-out live behind the tables: zero once, never reset -- see OlsDyn)
  const size_t ctr_bytes = (size_t) OLS_MAX_CTR * 128;
  img.resize(3 * (size_t) N + ctr_bytes / sizeof(cpx), mk(0.f, 0.f));
  if (hipMalloc(&f->d_H, 3 * bytes + ctr_bytes) != hipSuccess)
    return set_err(TSDGPU_ERR_HIP, "ols: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
  if (hipMemcpy(f->d_H, img.data(), 3 * bytes + ctr_bytes, hipMemcpyHostToDevice) != hipSuccess)
    return set_err(TSDGPU_ERR_HIP, "ols: upload failed: %s", hipGetErrorString(hipGetLastError()));
  f->d_ctr = (unsigned *) ((char *) f->d_H + 3 * bytes);
  f->ctr_base = 0;
  // persistent grid: as many waves as the device keeps resident (asked once per process: the devices of a node are alike)
  static const std::pair<int, int> occ = []() {
    int dev = 0, cus = 256, per_cu = 8;
    (void) hipGetDevice(&dev);
    (void) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ols_kernel<false, 2, true>, 64, 0) != hipSuccess || per_cu < 1) {
      (void) hipGetLastError();
      per_cu = 8;
    }
    return std::make_pair(cus, per_cu);
  }();
  const int cus = occ.first, per_cu = occ.second;      // waves per CU
  f->ols_grid = cus * per_cu;
  if (getenv("TSDGPU_DEBUG")) fprintf(stderr, "[tsdgpu] ols plan: N=%d K=%d L=%d cus=%d waves/CU=%d grid=%d waves\n", N, K, f->ols_L, cus, per_cu, f->ols_grid);
  return TSDGPU_OK;
}

void ols_plan_destroy(tsdgpu_fir *f)
{
  if (f->d_H) (void) hipFree(f->d_H);
  f->d_H = nullptr;
  f->d_ctr = nullptr;
}

int ols_step(tsdgpu_fir *f, const void *x, void *y, int64_t n, hipStream_t st)
{
  const bool real = f->data_type == TSDGPU_F32;
  const int L = f->ols_L;
  const int64_t LB = real ? 2 * (int64_t) L : L;          // outputs per work item (block or block pair)
  const int64_t nblocks = cdiv(n, LB);
  const cpx *d = (const cpx *) f->d_H;
  // interior items: inputs [b*LB-overlap, b*LB+LB) and outputs [b*LB, b*LB+LB) all inside [0, n)
  const int64_t ovl = OLS_N - L;
  const int64_t b_lo = std::min<int64_t>(nblocks, f->K > 1 ? cdiv(ovl, LB) : 0);
  const int64_t b_hi = std::max<int64_t>(b_lo, n / LB);
  int64_t grid = 0;
  // runs of R consecutive blocks per wave (ols_body): only with whole-row overlaps the kernel is specialised for, and only
  // when there are blocks enough for every wave to get several runs
  const int r0 = (int) (ovl / 64);
  const char *run_s = dev_switch("OLS_RUN");      // (read per step: scripts/perf_ols_run.py interleaves values in one process)
  const int run_env = run_s ? atoi(run_s) : OLS_RUN_DEFAULT;
  // dynamic hand-out of the runs (OlsDyn) unless the step must be capturable in a graph (frozen kernel arguments)
  const char *nc_s = dev_switch("OLS_DYN");       // number of counters; 0 = the static partition
  int NC = nc_s ? atoi(nc_s) : OLS_DYN_DEFAULT;
  if (NC < 0 || NC > OLS_MAX_CTR || f->capturable || !f->d_ctr || stream_is_capturing(st)) NC = 0;
  int R = 1;
  unsigned ctr_add = 0;
  OlsDyn dyn = {f->d_ctr, f->ctr_base, 0u, NC > 0 ? NC : 1, 0};
  if (b_hi > b_lo) {
    const int64_t nint = b_hi - b_lo;
    if (r0 >= 1 && r0 <= 4 && run_env > 1 && nint >= (int64_t) f->ols_grid * run_env * 2) R = run_env;
    const int64_t nruns = cdiv(nint, R);
    const char *min_s = dev_switch("OLS_DYN_MIN");        // runs per wave from which the hand-out is dynamic (tests: 0)
    const int64_t dyn_min = min_s ? atoi(min_s) : 4;
    if (NC > 0 && nruns >= dyn_min * (int64_t) f->ols_grid) {
      // whole groups of 8 * NC workgroups, so that every counter has the same number of pullers
      const int64_t grp = (int64_t) 8 * NC;
      grid = std::max<int64_t>(grp, (f->ols_grid / grp) * grp);
      if (NC != f->ctr_nc) {
        // another counter count than the launches before (a tuning switch flipped mid-stream): the counters beyond the
        // old count lag behind the base -- start over from zero
        TSD_HIP(hipMemsetAsync(f->d_ctr, 0, (size_t) OLS_MAX_CTR * 128, st));
        f->ctr_base = 0;
        f->ctr_nc = NC;
        dyn.base = 0;
      }
      dyn.nunits = nruns;
      dyn.Q = (unsigned) cdiv(nruns, NC);
      ctr_add = dyn.Q + (unsigned) (grid / NC);               // what this launch adds to every counter (booked once it is accepted)
    } else {
      NC = 0;
      // balance the rounds: every wave gets ceil(nruns/grid) or one fewer runs, no tail round
      grid = nruns < f->ols_grid ? nruns : f->ols_grid;
      if (nruns > grid) {
        const int64_t rounds = cdiv(nruns, grid);
        grid = cdiv(cdiv(nruns, rounds), 8) * 8;
      }
    }
  } else NC = 0;
  const int64_t b_lo_w = b_lo;
  // edge items, one wave each: [0, b_lo) read the history halo, [b_hi, nblocks) are ragged
  const int64_t n_lo = b_lo, b_tail = b_hi;
  const int ne = (int) (n_lo + (nblocks - b_tail));
  int64_t e[2] = {n_lo, b_tail};
  const int nxt = f->cur ^ 1;
  const unsigned nwg = (unsigned) (grid + ne + 1);
#define OLS_LAUNCH(REAL, R0, DYN)                                                                                                     \
  hipLaunchKernelGGL((ols_kernel<REAL, R0, DYN>), dim3(nwg), dim3(64), 0, st, x, (const void *) fir_hist_read(f),                        \
                     f->hist[nxt], y, d, d + OLS_N, d + 2 * OLS_N, OLS_N - L, f->HL, L, n, b_lo_w, b_hi, nblocks, (int) grid, ne,   \
                     e[0], e[1], R, dyn)
#define OLS_LAUNCH_R0(REAL, DYN)                                        \
  switch (R > 1 ? r0 : 0) {                                             \
    case 1: OLS_LAUNCH(REAL, 1, DYN); break;                            \
    case 2: OLS_LAUNCH(REAL, 2, DYN); break;                            \
    case 3: OLS_LAUNCH(REAL, 3, DYN); break;                            \
    case 4: OLS_LAUNCH(REAL, 4, DYN); break;                            \
    default: OLS_LAUNCH(REAL, 0, DYN); break;                           \
  }
  if (real) {
    if (NC > 0) { OLS_LAUNCH_R0(true, true) } else { OLS_LAUNCH_R0(true, false) }
  } else {
    if (NC > 0) { OLS_LAUNCH_R0(false, true) } else { OLS_LAUNCH_R0(false, false) }
  }
#undef OLS_LAUNCH_R0
#undef OLS_LAUNCH
  if (const hipError_t le = hipGetLastError(); le != hipSuccess) {
    if (NC > 0) f->ctr_nc = 0;       // (device counters and host base may have parted: the next dynamic launch zeroes them)
    return set_err(TSDGPU_ERR_HIP, "fir_step: overlap-save launch failed: %s", hipGetErrorString(le));
  }
  f->ctr_base += ctr_add;
  f->cur = nxt;      // the history update is part of the launch
  return TSDGPU_OK;
}

// ---- overlap-ADD on the same in-wave transform: the fast path of the OLA engine (ola.hip) ----------------------------
// OLA<cfloat>::step_interne without window (fourier.cc:846-872) in its default geometry Ne = 512, N = 1024 (= Nz = Ne):
//   frame = [512 zeros | block b]  ->  X = FFT(frame)  ->  X *= H (the caller's response)  ->  x2 = IFFT(X)
//   y_b = x2_{b-1}[512..1023] + x2_b[0..511]          (svg.tail(Nz) += x2.head(Nz); y = svg; svg = x2.tail(Ne))
// The multi-kernel engine moves 80 B per sample through HBM for that (frames, spectra, inverse frames); here a wave keeps
// a RUN of consecutive blocks in registers end to end: 16 B per sample.  The tail of the block before the run is needed to
// start it: the first wave takes it from the handle's state (svg), the others recompute that one block (1 / per more work).
__global__ __launch_bounds__(64, 2) void ola1024_kernel(const cpx *__restrict__ x, cpx *__restrict__ y, const cpx *__restrict__ Hreg,
                                                        const cpx *__restrict__ TW1, const cpx *__restrict__ TW2,
                                                        const cpx *__restrict__ svg_in, cpx *__restrict__ svg_out, int64_t B, int per)
{
  __shared__ cv lds[LDS_ELEMS];
  const int lane = threadIdx.x;
  const int64_t b_lo = (int64_t) blockIdx.x * per, b_hi = min(B, b_lo + (int64_t) per);
  if (b_lo >= B) return;
  cv tw1[16], tw2[16], H[16];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    tw1[r] = ((const cv *) TW1)[r * 64 + lane];
    tw2[r] = ((const cv *) TW2)[r * 64 + lane];
    H[r] = ((const cv *) Hreg)[r * 64 + lane];
  }
  auto sync = []() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const cv *xs = (const cv *) x;
  cv *ys = (cv *) y;
  cv cur[16], carry[8], nxt[8];
  auto transform = [&](cv (&v)[16]) {
    forward(v, lds, lane, tw1, tw2, sync);
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = cmul(v[r], H[r]);
    inverse(v, lds, lane, tw1, tw2, sync);
    sync();
  };
  if (b_lo == 0) {
#pragma unroll
    for (int r = 0; r < 8; r++) carry[r] = ((const cv *) svg_in)[64 * r + lane];
  } else {
#pragma unroll
    for (int r = 0; r < 8; r++) { cur[r] = mkv(0.f, 0.f); cur[8 + r] = xs[(b_lo - 1) * 512 + 64 * r + lane]; }
    transform(cur);
#pragma unroll
    for (int r = 0; r < 8; r++) carry[r] = cur[8 + r];
  }
#pragma unroll
  for (int r = 0; r < 8; r++) nxt[r] = xs[b_lo * 512 + 64 * r + lane];
  for (int64_t b = b_lo; b < b_hi; b++) {
#pragma unroll
    for (int r = 0; r < 8; r++) { cur[r] = mkv(0.f, 0.f); cur[8 + r] = nxt[r]; }
    const bool more = b + 1 < b_hi;
    if (more) {
#pragma unroll
      for (int r = 0; r < 8; r++) nxt[r] = xs[(b + 1) * 512 + 64 * r + lane];
    }
    transform(cur);
    if (more) {
      // (vmcnt retires in order: wait for the prefetch, a whole transform old, BEFORE the stores are queued behind it)
#pragma unroll
      for (int r = 0; r < 8; r++) asm volatile("" ::"v"(nxt[r]));
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      ys[b * 512 + 64 * r + lane] = mkv(cur[r].x + carry[r].x, cur[r].y + carry[r].y);
      carry[r] = cur[8 + r];
    }
  }
  if (b_hi == B) {
#pragma unroll
    for (int r = 0; r < 8; r++) ((cv *) svg_out)[64 * r + lane] = carry[r];
  }
}

// tables of the fast path: the response in the transform's register order, scaled by 1/N (the engine's transforms are
// unitary each way, the in-wave ones are not scaled), and the two twiddle sets -> 3 x 1024 complex values
void ola1024_tables(const cpx *H_host, cpx *out3)
{
  for (int lane = 0; lane < 64; lane++)
    for (int r = 0; r < 16; r++) {
      const cpx h = H_host[freq_index(lane, r)];
      out3[r * 64 + lane] = mk(h.x / 1024.f, h.y / 1024.f);
    }
  fill_twiddles(out3 + 1024, out3 + 2048);
}

// B whole blocks of 512 samples: x, y, svg device pointers (y may not alias x: blocks are re-read by the next run)
int ola1024_launch(const cpx *x, cpx *y, const cpx *tables3, const cpx *svg_in, cpx *svg_out, int64_t B, hipStream_t st)
{
  if (B <= 0) return TSDGPU_OK;
  // blocks per wave: short calls spread one block per wave (each recomputes its predecessor: twice the arithmetic, all of it
  // parallel); long ones amortise the recomputed block over a run of up to 16
  const int per = (int) std::min<int64_t>(16, std::max<int64_t>(1, cdiv(B, 2048)));      // (rounded up: 2048 waves are resident at once)
  const int64_t grid = cdiv(B, per);
  if (grid > 0x7fffffff) return set_err(TSDGPU_ERR_UNSUPPORTED, "ola: too many blocks in one call");
  hipLaunchKernelGGL(ola1024_kernel, dim3((unsigned) grid), dim3(64), 0, st, x, y, tables3, tables3 + 1024, tables3 + 2048, svg_in, svg_out, B, per);
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}

// ---- the WINDOWED overlap-add engine at its default geometry (Ne = N = 512, fourier.cc:883-927) on the in-wave transform ----
// Per block b two Hann-windowed frames half a block apart, A_b = [second half of block b-1 | first half of block b] and
// B_b = block b, each -> FFT -> x H -> IFFT.  Without zero padding the reference's bookkeeping (svg, last; statement by
// statement in olaw_run_kernel, ola.hip) closes to
//     y_b = [ B_{b-2}.tail / 2 + A_{b-1}.head / 2  |  A_{b-1}.tail / 2 + B_{b-1}.head / 2 ]
// (every sum has two terms: the order of the reference's additions does not matter), with the handle's state
// svg = B_{b-1}, last = [ B_{b-2}.tail / 2 + A_{b-1}.head / 2 | A_{b-1}.tail / 2 ] between calls.
// A wave transforms the PAIR (A_p, B_p) in one pass of the 1024-point engine (forward_pair512: lanes with bit 1 clear hold A_p,
// the others B_p; 16 samples per lane) and writes block p + 1 after it: the lanes of A_p the first half -- with the upper half of
// B_{p-1}, kept from the pass before --, the lanes of B_p the second half with the upper half of A_p; the halves travel between a
// lane and its partner (lane ^ 2) by DPP.  A run of `per` blocks starts two passes early (the first one only for its B); the
// runs that start at block 0 or 1 take the handle's state instead, and the last run makes the next state.  16 B of HBM traffic
// per sample; the same work per sample as the unwindowed default geometry (one 1024-point pair of transforms per 512 samples).
__global__ __launch_bounds__(64, 2) void olaw512_kernel(const cpx *__restrict__ blk0, int nrest, const cpx *__restrict__ x, cpx *__restrict__ y,
                                                        const cpx *__restrict__ Hreg, const cpx *__restrict__ TW1, const cpx *__restrict__ TW2,
                                                        const float *__restrict__ fen, const cpx *__restrict__ svg_in,
                                                        const cpx *__restrict__ last_in, const cpx *__restrict__ prev_half_in,
                                                        cpx *__restrict__ svg_out, cpx *__restrict__ last_out, int64_t B, int per,
                                                        int skip_first)
{
  __shared__ cv lds[LDS_ELEMS];
  const int lane = threadIdx.x;
  const int64_t b_lo = (int64_t) blockIdx.x * per, b_hi = min(B, b_lo + (int64_t) per);
  if (b_lo >= B) return;
  const int sig = (lane >> 1) & 1, c = 2 * (lane >> 2) + (lane & 1);      // this lane's sequence (0: A, 1: B) and sample column
  cv tw1[16], tw2[16], H[8];
  float win[16];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    tw1[r] = ((const cv *) TW1)[r * 64 + lane];
    tw2[r] = ((const cv *) TW2)[r * 64 + lane];
    win[r] = fen[32 * r + c];
  }
#pragma unroll
  for (int q = 0; q < 8; q++) H[q] = ((const cv *) Hreg)[q * 64 + lane];         // bin of reg 4 j1hi + 2 sigma + j2: q = 2 j1hi + j2
  auto sync = []() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // sample i (-256 <= i) of the stream [rest ++ x]; block 0 was made contiguous by the host, the half block before it is the handle's
  auto sample = [&](int64_t i) -> cv {
    if (i < 0) return ((const cv *) prev_half_in)[i + 256];
    if (i < 512) return ((const cv *) blk0)[i];
    return ((const cv *) x)[i - nrest];
  };
  auto xchg = [&](cv a) -> cv {        // the partner's value (lane ^ 2)
    return mkv(__shfl_xor(a.x, 2), __shfl_xor(a.y, 2));
  };
  cv v[16], keep[8];                   // keep: on the lanes of A the upper half of the B before
  // the first pass: block 0 or 1 starts from the state (B_{-1} = svg), later runs two blocks early
  int64_t p = b_lo <= 1 ? 0 : b_lo - 2;
  if (b_lo <= 1) {
#pragma unroll
    for (int r = 0; r < 8; r++) keep[r] = ((const cv *) svg_in)[256 + 32 * r + c];
    if (b_lo == 0 && !skip_first) {
      // y_0 = [ last.head | last.tail + svg.head / 2 ]   (:895-899)
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int i = 32 * r + c;
        cv o = ((const cv *) last_in)[256 * sig + i];
        if (sig) {
          const cv g = ((const cv *) svg_in)[i];
          o = mkv(o.x + g.x * 0.5f, o.y + g.y * 0.5f);
        }
        ((cv *) y)[256 * sig + i] = o;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 8; r++) keep[r] = mkv(0.f, 0.f);
  }
  // passes p = first .. b_hi - 2 write block p + 1; the last run adds pass B - 1 for the state
  const int64_t p_end = b_hi == B ? B : b_hi - 1;      // exclusive
  for (; p < p_end; p++) {
    const int64_t base = 512 * p - 256 + 256 * sig;    // first sample of this lane's frame
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const cv a = sample(base + 32 * r + c);
      v[r] = mkv(a.x * win[r], a.y * win[r]);          // :886 / :910
    }
    forward_pair512(v, lds, lane, tw1, tw2, sync);
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = cmul(v[r], H[2 * (r >> 2) + (r & 1)]);
    inverse_pair512(v, lds, lane, tw1, tw2, sync);
    sync();
    cv got[8];
#pragma unroll
    for (int r = 0; r < 8; r++) got[r] = xchg(v[8 + r]);
    if (p + 1 < B) {
      if (p + 1 >= b_lo) {
        // lanes of A: y.head = B_{p-1}.tail / 2 + A_p.head / 2; lanes of B: y.tail = A_p.tail / 2 + B_p.head / 2
        cv *yo = (cv *) y + (p + 1 - skip_first) * 512 + 256 * sig + c;
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const cv o = sig ? got[r] : keep[r];
          yo[32 * r] = mkv(o.x * 0.5f + v[r].x * 0.5f, o.y * 0.5f + v[r].y * 0.5f);
        }
      }
    } else {
      // the state after the call's last block: svg = B_p, last = [ B_{p-1}.tail / 2 + A_p.head / 2 | A_p.tail / 2 ]
      if (sig) {
#pragma unroll
        for (int r = 0; r < 16; r++) ((cv *) svg_out)[32 * r + c] = v[r];
      } else {
#pragma unroll
        for (int r = 0; r < 8; r++) {
          ((cv *) last_out)[32 * r + c] = mkv(keep[r].x * 0.5f + v[r].x * 0.5f, keep[r].y * 0.5f + v[r].y * 0.5f);
          ((cv *) last_out)[256 + 32 * r + c] = mkv(v[8 + r].x * 0.5f, v[8 + r].y * 0.5f);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 8; r++) keep[r] = got[r];      // (on the lanes of A: the upper half of B_p)
  }
}

// tables of that kernel: the response of the 8 bins a lane holds per sequence, scaled by 1/512, and the two twiddle sets of the
// pair transform -> 512 + 2 x 1024 complex values
void olaw512_tables(const cpx *H_host, cpx *out)
{
  for (int lane = 0; lane < 64; lane++)
    for (int q = 0; q < 8; q++) {
      const cpx h = H_host[pair512_freq_index(lane, 4 * (q >> 1) + (q & 1))];
      out[q * 64 + lane] = mk(h.x / 512.f, h.y / 512.f);
    }
  fill_twiddles_pair512(out + 512, out + 512 + 1024);
}

int olaw512_launch(const cpx *blk0, int nrest, const cpx *x, cpx *y, const cpx *tables, const float *fen, const cpx *svg_in, const cpx *last_in,
                   const cpx *prev_half_in, cpx *svg_out, cpx *last_out, int64_t B, int skip_first, hipStream_t st)
{
  if (B <= 0) return TSDGPU_OK;
  // blocks per wave: a run costs per + 2 passes; short calls spread out, long ones amortise the two warm-up passes over 16 blocks
  const int per = (int) std::min<int64_t>(16, std::max<int64_t>(2, cdiv(B, 2048)));
  const int64_t grid = cdiv(B, per);
  if (grid > 0x7fffffff) return set_err(TSDGPU_ERR_UNSUPPORTED, "ola: too many blocks in one call");
  hipLaunchKernelGGL(olaw512_kernel, dim3((unsigned) grid), dim3(64), 0, st, blk0, nrest, x, y, tables, tables + 512, tables + 512 + 1024, fen, svg_in,
                     last_in, prev_half_in, svg_out, last_out, B, per, skip_first);
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}

// ---- Welch periodogram sums, N = 1024, on the same in-wave transform (ola.hip: tsdgpu_welch's fast path) -------------
// psd_welch (freqestim.cc:7-20): segments of N samples every N/2, windowed, |FFT|^2 summed.  A wave runs through `per`
// consecutive segments: loads (each sample is read by two segments: 16 B of HBM traffic per sample, against the 50 B of
// the framing / transform / power passes), window, transform, and the 1024 running sums stay in registers until the
// end of the run; part[wave][i] gets them in fftshift order (the reduction over the waves is welch_sum_kernel's).
__global__ __launch_bounds__(64, 2) void welch1024_kernel(const cpx *__restrict__ x, const float *__restrict__ w,
                                                          const cpx *__restrict__ TW1, const cpx *__restrict__ TW2,
                                                          float *__restrict__ part, int64_t nseg, int per)
{
  __shared__ cv lds[LDS_ELEMS];
  const int lane = threadIdx.x;
  const int64_t k_lo = (int64_t) blockIdx.x * per, k_hi = min(nseg, k_lo + (int64_t) per);
  cv tw1[16], tw2[16];
  float win[16], acc[16];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    tw1[r] = ((const cv *) TW1)[r * 64 + lane];
    tw2[r] = ((const cv *) TW2)[r * 64 + lane];
    win[r] = w[64 * r + lane];
    acc[r] = 0.f;
  }
  auto sync = []() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const cv *xs = (const cv *) x;
  cv cur[16], nxt[16];
  if (k_lo < k_hi) {
#pragma unroll
    for (int r = 0; r < 16; r++) nxt[r] = xs[k_lo * 512 + 64 * r + lane];
  }
  for (int64_t k = k_lo; k < k_hi; k++) {
#pragma unroll
    for (int r = 0; r < 16; r++) cur[r] = mkv(nxt[r].x * win[r], nxt[r].y * win[r]);      // x.segment(i, N) * f  (:15)
    if (k + 1 < k_hi) {
#pragma unroll
      for (int r = 0; r < 16; r++) nxt[r] = xs[(k + 1) * 512 + 64 * r + lane];
    }
    forward(cur, lds, lane, tw1, tw2, sync);
    sync();
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] += cur[r].x * cur[r].x + cur[r].y * cur[r].y;     // abs2 (:16)
  }
  // the engine's transform is unitary (1 / sqrt(N)): |X|^2 / N
#pragma unroll
  for (int r = 0; r < 16; r++) part[(size_t) blockIdx.x * 1024 + ((freq_index(lane, r) + 512) & 1023)] = acc[r] * (1.0f / 1024.0f);
}

int welch1024_launch(const cpx *x, const float *w, const cpx *tw2x1024, float *part, int64_t nseg, int per, hipStream_t st)
{
  const int64_t grid = cdiv(nseg, per);
  hipLaunchKernelGGL(welch1024_kernel, dim3((unsigned) grid), dim3(64), 0, st, x, w, tw2x1024, tw2x1024 + 1024, part, nseg, per);
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}

void welch1024_tables(cpx *out2) { fill_twiddles(out2, out2 + 1024); }

}  // namespace tsdgpu
