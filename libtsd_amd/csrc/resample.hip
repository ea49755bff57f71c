// resample.hip -- fractional resampler: LUT-sinc interpolator driven by libtsd's float32
// phase accumulator, reproduced bit-exactly and evaluated tile-parallel.
//
// Stands behind AdaptationRythmeSimple<T>::step / filtre_itrp<T> (libtsd core/src/reechan/
// ra.cc:13-79) with InterpolateurRIF::step (core/include/tsd/filtrage.hpp:1873-1881) and the
// sinc LUT of itrp_sinc (core/src/reechan/itrp.cc:10-55), i.e. what filtre_reechan builds for
// a ratio in [0.5,2) (ra.cc:104-156).  Per input sample the reference does
//     window <- shift in x;  while (phase < 1) { emit sum_i lut[i][(int)(phase*nphases)] * window[i];
//                                               phase += 1/ratio; }  phase -= 1;
// with `phase` a float32: the number of outputs and the (input index, LUT column) of each one
// are an index/permute result and are matched BIT-EXACTLY (same float additions, in the same
// order, on host and device).
//
// The recurrence is sequential, so it is cut with checkpoints: the host simulates it once per
// handle (lazily, only as far as the stream has advanced), records (phase, outputs so far)
// every 8 inputs and runs Brent's cycle detection on the phase at input boundaries -- the
// state space is finite, so the sequence becomes periodic (ratio 160/147: period 3 853 516
// inputs <-> 4 194 303 outputs) and from then on any stream position is a table lookup.
// On the device every lane restarts from the checkpoint at or below its 8-input segment,
// replays at most 15+8 inputs, and writes one (input, column) record per output into LDS;
// then the workgroup evaluates all outputs of the tile in parallel (15 taps x window from
// LDS) and stores them coalesced.  HBM traffic: 8 B in + 8*ratio B out per complex sample.
#include "common.hpp"
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>

namespace tsdgpu {

constexpr int RS_WAVES = 8;                     // waves per workgroup; each wave owns its tiles
constexpr int RS_THREADS = 64 * RS_WAVES;
constexpr int RS_SEG = 8;                       // inputs per lane
constexpr int RS_TI = 64 * RS_SEG;              // inputs per (wave) tile: 512
constexpr int RS_CK = 8;                        // checkpoint spacing (inputs)

struct RsParams {
  int64_t pos;        // absolute index of x[0]
  int64_t n;          // inputs in this call
  int64_t tile0;      // absolute start of the first tile (multiple of RS_TI, <= pos)
  int64_t mu, lambda; // cycle of the phase sequence (lambda = 0: not found yet, table covers the call)
  int64_t opp;        // outputs per period
  int64_t cum_pos;    // outputs emitted before `pos`
  float inc;          // 1/ratio as float (ra.cc:29)
  int K, nph, lstride, rec_cap;
  int lut_in_lds;     // generic kernel: LUT staged in LDS (1) or read from global memory (0: large K)
  int mode;           // 0: table; 1: InterpolateurLineaire; 2: InterpolateurLagrange of degree K-1 (itrp.cc:80-133)
  int gl;             // row pitch of the table in global memory: K rounded up to a multiple of 4 (16-B row reads)
  float inv_den[32];  // Lagrange: 1 / prod_{k != j} (j - k)
};

struct RsCk { uint32_t phase_bits; uint32_t cum; };
constexpr int RS_KMAX = 256;                   // longest interpolator (taps); LUTs beyond RS_LUT_LDS_BYTES stay in global memory
constexpr int RS_LUT_LDS_BYTES = 48 * 1024;      // tables up to this size leave room for RS_WAVES waves per workgroup
constexpr size_t RS_LDS_LIMIT = 158 * 1024;
__host__ __device__ inline int rs_tile_elems(int K) { return RS_TI + (K <= 32 ? 32 : (K + 31) / 32 * 32) + 2; }   // cum = outputs before this input (canonical index space)

__host__ __device__ inline float bits2f(uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(b);
#else
  float f; memcpy(&f, &b, 4); return f;
#endif
}
__host__ __device__ inline uint32_t f2bits(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_uint(f);
#else
  uint32_t b; memcpy(&b, &f, 4); return b;
#endif
}

__device__ __forceinline__ float zero_of(float) { return 0.f; }
__device__ __forceinline__ float2 zero_of(float2) { return make_float2(0.f, 0.f); }
__device__ __forceinline__ float tap_mac(float acc, float h, float x) { return fmaf(h, x, acc); }
__device__ __forceinline__ float2 tap_mac(float2 acc, float h, float2 x)
{
  return make_float2(fmaf(h, x.x, acc.x), fmaf(h, x.y, acc.y));
}

// the same two fused multiply-adds as ONE v_pk_fma_f32 (long interpolators: a wave per SIMD, VALU issue counts)
typedef float rs_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float tap_mac_pk(float acc, float h, float x) { return fmaf(h, x, acc); }
__device__ __forceinline__ float2 tap_mac_pk(float2 acc, float h, float2 x)
{
  const rs_v2f r = __builtin_elementwise_fma(rs_v2f{h, h}, rs_v2f{x.x, x.y}, rs_v2f{acc.x, acc.y});
  return make_float2(r.x, r.y);
}

// canonical index of the periodic phase sequence: ic -> mu + (ic - mu) mod lambda, q += the number of periods removed.
// A few subtractions when the index is a few periods out (the usual case: lambda in the millions, 160/147: 3.85 M); a
// division otherwise -- ratios whose float recurrence has a SHORT period (1: lambda = 1, 0.5: 2, 2.5: 5 ...) put millions
// of periods between the table and a stream position, and the subtraction loop they used to go through cost
// 178 ms per 4 M samples at ratio 1.
__device__ __forceinline__ void rs_wrap(int64_t &ic, int64_t &q, int64_t mu, int64_t lambda)
{
  if (lambda <= 0) return;
  const int64_t lim = mu + lambda;
  if (ic < lim) return;
  if (ic - lim < 4 * lambda) {
    do { ic -= lambda; q++; } while (ic >= lim);
  } else {
    const int64_t d = ic - mu, k = d / lambda;
    q += k;
    ic = mu + (d - k * lambda);
  }
}

// KT = 15: the filtre_reechan interpolator -- taps padded to 16 and read as four ds_read_b128
// from LUT rows of pitch 20 floats (80 B: 16-B aligned, and 5 mod 16 in 16-B units, so the
// rows selected by 16 lanes whose columns step regularly fall on distinct bank groups).
// KT = 0: generic K <= 32, scalar tap reads at an odd pitch.
//
// Work decomposition: the workgroup (8 waves) shares the LUT in LDS; every WAVE owns its own
// 512-input tiles (64 lanes x 8 inputs) with a private sample window and record list, so the
// tile loop contains no workgroup barrier at all -- waves drift freely and hide each other's
// latencies.  The next tile's samples and checkpoint are prefetched into registers.
template <typename T, int KT>
__global__ __launch_bounds__(RS_THREADS) void resample_kernel(const T *__restrict__ x, const T *__restrict__ hist,
                                                              T *__restrict__ y, const float *__restrict__ lut,
                                                              const RsCk *__restrict__ ck, RsParams P, int ntiles,
                                                              T *__restrict__ hist_next)
{
  // the workgroup after the persistent ones writes the stream's next window history (the last K - 1 inputs of
  // history ++ x) into the handle's other buffer: one launch per step
  // waves per workgroup: RS_WAVES, or fewer when a long interpolator's table takes most of the LDS (rs_geometry)
  const int NT = blockDim.x, NW = NT >> 6;
  if (blockIdx.x == gridDim.x - 1) {
    const int H = P.K - 1;
    for (int i = threadIdx.x; i < H; i += NT) {
      const int64_t g = P.n - H + i;
      hist_next[i] = g < 0 ? (hist ? hist[H + g] : zero_of(T{})) : x[g];
    }
    return;
  }
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int K = KT > 0 ? KT : P.K;
  const int lstride = KT == 15 ? 20 : P.lstride;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float *lut_s = reinterpret_cast<float *>(smem_raw);                       // (nph+1) x lstride, 16-B aligned
  char *wbase = reinterpret_cast<char *>(lut_s + (P.lut_in_lds ? ((P.nph + 1) * lstride + 3) / 4 * 4 : 4));
  const int tile_elems = rs_tile_elems(K);
  const size_t wbytes = ((size_t) tile_elems * sizeof(T) + (size_t) P.rec_cap * (P.mode ? 8 : 4) + 15) / 16 * 16;
  T *tile = reinterpret_cast<T *>(wbase + wv * wbytes);
  uint32_t *rec = reinterpret_cast<uint32_t *>(tile + tile_elems);
  float *rec_tau = reinterpret_cast<float *>(rec + P.rec_cap);      // analytic interpolators: the phase itself

  // ---- stage the LUT once per (persistent) workgroup (a LUT too large for LDS is read in place)
  const int lut_n = P.lut_in_lds ? (P.nph + 1) * K : 0;
  for (int i = threadIdx.x; i < lut_n; i += NT) {
    const int c = i / K, k = i - c * K;
    lut_s[c * lstride + k] = lut[c * P.gl + k];
  }
  const float *lutp = P.lut_in_lds ? lut_s : lut;
  const int lrow = P.lut_in_lds ? lstride : P.gl;         // both multiples of 4: rows are read 16 B at a time
  if (KT == 15)
    for (int c = threadIdx.x; c <= P.nph; c += NT) lut_s[c * lstride + 15] = 0.f;   // 16th tap
  __syncthreads();

  auto wave_sync = []() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  // canonical (cycle-folded) index of this wave's first tile: ONE 64-bit division per wave;
  // afterwards the index advances by a constant and is folded by subtraction
  const int wtile0 = blockIdx.x * NW + wv;
  const int wstep = (gridDim.x - 1) * NW;
  int64_t icT = P.tile0 + (int64_t) wtile0 * RS_TI, qT = 0;
  if (P.lambda > 0 && icT >= P.mu + P.lambda) {
    const int64_t d = icT - P.mu;
    qT = d / P.lambda;
    icT = P.mu + (d - qT * P.lambda);
  }
  const int64_t icStep = (int64_t) wstep * RS_TI;

  constexpr int NPF = (RS_TI + RS_KMAX + 63) / 64;   // samples per lane per tile (K <= RS_KMAX)
  T pf[NPF];
  RsCk cpf;
  auto fetch = [&](int tix_, int64_t icT_) {
    const int64_t T0_ = P.tile0 + (int64_t) tix_ * RS_TI;
    const int64_t rel0 = T0_ - (K - 1) - P.pos;                     // x index of tile[0]
    if (rel0 >= 0 && rel0 + RS_TI + K <= P.n) {                     // interior tile: no guards
      const T *xs = x + rel0;
#pragma unroll
      for (int j = 0; j < NPF; j++) {
        const int s_ = lane + j * 64;
        pf[j] = s_ < RS_TI + K ? xs[s_] : zero_of(T{});
      }
    } else {
#pragma unroll
      for (int j = 0; j < NPF; j++) {
        const int64_t rel = rel0 + lane + j * 64;
        T v = zero_of(T{});
        if (lane + j * 64 < RS_TI + K) {
          if (rel < 0) {
            if (hist && rel >= -(int64_t) (K - 1)) v = hist[(K - 1) + rel];
          } else if (rel < P.n) {
            v = x[rel];
          }
        }
        pf[j] = v;
      }
    }
    int64_t ic_ = icT_ + lane * RS_SEG;
    {
      int64_t q_ = 0;
      rs_wrap(ic_, q_, P.mu, P.lambda);
    }
    cpf.phase_bits = 0x40000000u;                                   // 2.0f: "emit nothing"
    cpf.cum = 0;
    if (T0_ + (int64_t) lane * RS_SEG < P.pos + P.n) cpf = ck[ic_ / RS_CK];
  };

  if (wtile0 < ntiles) fetch(wtile0, icT);
  for (int tix = wtile0; tix < ntiles; tix += wstep) {
    const int64_t T0 = P.tile0 + (int64_t) tix * RS_TI;
    // ---- consume the prefetched tile: samples to LDS, checkpoint to registers
#pragma unroll
    for (int j = 0; j < NPF; j++) {
      const int s_ = lane + j * 64;
      if (s_ < RS_TI + K) tile[s_] = pf[j];
    }
    const int64_t i_abs = T0 + (int64_t) lane * RS_SEG;
    int64_t ic = icT + lane * RS_SEG, q = qT;
    rs_wrap(ic, q, P.mu, P.lambda);
    const bool in_call = i_abs < P.pos + P.n;              // lanes past the end of the call do nothing
    const float inc = P.inc;
    float phase = bits2f(cpf.phase_bits);
    int64_t cum = (int64_t) cpf.cum + q * P.opp;
    // ---- issue the next tile's loads
    icT += icStep;
    rs_wrap(icT, qT, P.mu, P.lambda);
    if (tix + wstep < ntiles) fetch(tix + wstep, icT);

    if (in_call) {
      for (int s = (int) (ic % RS_CK); s > 0; s--) {      // catch up from the checkpoint
        while (phase < 1.f) { phase = phase + inc; cum++; }
        phase = phase - 1.f;
      }
    }
    // outputs before the tile = lane 0's count (lane 0 is always inside the call)
    const int64_t cum_t0 = ((int64_t) __shfl((int) (cum >> 32), 0) << 32) | (uint32_t) __shfl((int) (uint32_t) cum, 0);
    const float fnph = (float) P.nph;
    int last = 0;
    for (int s = 0; s < RS_SEG && in_call; s++) {
      const int64_t i = i_abs + s;
      const bool live = i >= P.pos && i < P.pos + P.n;
      while (phase < 1.f) {
        if (live) {
          const int o = (int) (cum - cum_t0);
          rec[o] = ((uint32_t) (lane * RS_SEG + s) << 13) | (uint32_t) (int) (phase * fnph);   // itrp.cc:19
          if (P.mode) rec_tau[o] = phase;
          last = o + 1;
        }
        phase = phase + inc;                                // ra.cc:71, float32 add
        cum++;
      }
      phase = phase - 1.f;                                  // ra.cc:73
    }
    // wave-wide maximum of `last`
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) last = max(last, __shfl_xor(last, d));
    wave_sync();

    // ---- evaluate the tile's outputs in parallel
    const int o_begin = (int) max((int64_t) 0, P.cum_pos - cum_t0);
    const int o_end = last;
    T *yt = y + (cum_t0 - P.cum_pos);
    for (int o = o_begin + lane; o < o_end; o += 64) {
      const uint32_t r = rec[o];
      const T *w = tile + (r >> 13);                         // window: x[i-K+1 .. i], oldest first
      const float *h = lutp + (r & 8191u) * lrow;
      T acc = zero_of(T{});
      if (KT == 15) {
        float hh[16];
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
          const float4 q4 = *reinterpret_cast<const float4 *>(h + 4 * k4);
          hh[4 * k4] = q4.x; hh[4 * k4 + 1] = q4.y; hh[4 * k4 + 2] = q4.z; hh[4 * k4 + 3] = q4.w;
        }
#pragma unroll
        for (int k = 0; k < 15; k++) acc = tap_mac(acc, hh[k], w[k]);   // filtrage.hpp:1877-1879 order
      } else if (P.mode == 0) {
        // taps 16 B at a time (a row per lane: scalar reads cost one cache line or LDS access per tap)
        const int K4 = K & ~3;
#pragma unroll 4
        for (int k = 0; k < K4; k += 4) {
          const float4 q4 = *reinterpret_cast<const float4 *>(h + k);
          acc = tap_mac(acc, q4.x, w[k]);
          acc = tap_mac(acc, q4.y, w[k + 1]);
          acc = tap_mac(acc, q4.z, w[k + 2]);
          acc = tap_mac(acc, q4.w, w[k + 3]);
        }
        for (int k = K4; k < K; k++) acc = tap_mac(acc, h[k], w[k]);
      } else if (P.mode == 1) {
        const float tau = rec_tau[o];                          // coefs(tau) = {1 - tau, tau}  (itrp.cc:83-86)
        acc = tap_mac(acc, 1.f - tau, w[0]);
        acc = tap_mac(acc, tau, w[1]);
      } else {
        // Lagrange of degree d through the points 0..d, evaluated at (d-1)/2 + tau (itrp.cc:112-132)
        const int d = K - 1;
        const float t = ((d - 1.0f) / 2) + rec_tau[o];
        // p_j = prod_{k != j} (t - k) / (j - k): the constant denominators are folded into one
        // host-computed reciprocal per tap (the K^2 correctly-rounded divisions of the literal form
        // cost 10x the rest of the kernel; the weights move by a few ulp)
        for (int j = 0; j <= d; j++) {
          float p = P.inv_den[j];
          for (int k = 0; k <= d; k++)
            if (k != j) p *= (t - k);
          acc = tap_mac(acc, p, w[j]);
        }
      }
      yt[o] = acc;
    }
    wave_sync();          // tile / rec are rewritten by the next iteration
  }
}

// ---- long table-driven interpolators (K >= 24 taps, ratio <= 2; e.g. the 127-tap sinc of the reference's test_ra_unit) --------
// resample_kernel evaluates an output per lane and reads, per tap, a sample (8 B) and a coefficient (4 B) from LDS: 12 B of LDS
// traffic per tap and output -- at 127 taps the LDS floor alone is 1.4 ms per 2^26 inputs, and with the 130-KiB table in LDS two
// waves per CU are left to reach it (4.6 ms).  Here the SAMPLES come from registers: a lane owns 8 consecutive inputs (as in
// the K = 15 kernel), walks the taps in chunks of 16 with the chunk's 23-sample window in registers (statically indexed: the
// input index s and the tap k are unrolled), and evaluates the FIRST output of each of its inputs -- every input has one when
// ratio >= 1, most have when it is below.  An input's SECOND output (ratio > 1: 9 % of the outputs at 160/147) would cost a
// second, mostly idle copy of the whole evaluation: those go to a list and are evaluated one per lane in the old way afterwards.
// 5.4 B of LDS traffic per tap and output instead of 12; the accumulation order (tap 0 .. K-1, filtrage.hpp:1877-1879) is kept.
// Outputs are stored straight from registers (a lane's outputs are consecutive; the lines are completed in L2).
#ifndef RSL_CHUNK
#define RSL_CHUNK 8
#endif
constexpr int RSL_CH = RSL_CHUNK;                           // taps per chunk (8: 64 + 30 registers of rows and window; 16 spills at two waves per SIMD)
constexpr int RSL_WIN = RSL_CH + RS_SEG - 1;                // samples of a chunk's window
__host__ __device__ inline int rsl_slot(int p) { return p + (p >> 3); }       // padded sample image: a lane's window starts 9 slots after its neighbour's
// (the last sample a window read touches: lane 63, last chunk, window entry 22)
__host__ __device__ inline int rsl_tile_slots(int K) { return rsl_slot(RS_TI - RS_SEG + (K + RSL_CH - 1) / RSL_CH * RSL_CH - RSL_CH + RSL_WIN - 1) + 2; }

// TAP PHASES: a 257 x 127 table takes 133 KiB of the 160 and leaves room for four waves -- one per SIMD, with nobody to
// fill the latency of its LDS reads and dependent multiply-adds (PMC: VALU 34 % busy, LDS 48 %).  The table is therefore staged
// in `nphase` slices of `kp` taps (a multiple of 16); a workgroup walks ALL its tiles per slice: phase 0 stores the partial sums
// of taps [0, kp) as the outputs, every later phase reloads them, continues the same accumulation in the same order (a float
// stored and reloaded is the same float: the result does not depend on the number of phases) and stores them back.  The static
// partition gives a tile to the same wave -- and an output to the same lane -- in every phase, so a thread re-reads only what
// it wrote itself.  The extra traffic (x once more and the outputs out and back per extra phase) is noise at 3-10 % of the
// HBM roofline; the replay is repeated per phase.
template <typename T>
__global__ __launch_bounds__(512) void resample_long_kernel(const T *__restrict__ x, const T *__restrict__ hist, T *__restrict__ y,
                                                            const float *__restrict__ lut, const RsCk *__restrict__ ck, RsParams P,
                                                            int ntiles, T *__restrict__ hist_next, int cap2, int kp, int nphase)
{
  const int NT = blockDim.x, NW = NT >> 6;
  if (blockIdx.x == gridDim.x - 1) {
    const int H = P.K - 1;
    for (int i = threadIdx.x; i < H; i += NT) {
      const int64_t g = P.n - H + i;
      hist_next[i] = g < 0 ? (hist ? hist[H + g] : zero_of(T{})) : x[g];
    }
    return;
  }
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int K = P.K;
  const int lsp = kp + 4;                                  // slice pitch: a multiple of 4 floats with an odd number of 16-B units
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float *lut_s = reinterpret_cast<float *>(smem_raw);
  char *wbase = reinterpret_cast<char *>(lut_s + (P.nph + 1) * lsp);
  const int tile_slots = rsl_tile_slots(K);
  const size_t wbytes = ((size_t) tile_slots * sizeof(T) + (size_t) cap2 * 8 + 15) / 16 * 16;
  T *tile = reinterpret_cast<T *>(wbase + wv * wbytes);
  uint2 *rec2 = reinterpret_cast<uint2 *>(tile + tile_slots);       // second outputs: (input << 13 | column, output index)

  auto wave_sync = []() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const int wtile0 = blockIdx.x * NW + wv;
  const int wstep = (gridDim.x - 1) * NW;
  const int64_t icStep = (int64_t) wstep * RS_TI;
  const float fnph = (float) P.nph;

  constexpr int NPF = (RS_TI + RS_KMAX + 63) / 64;
  T pf[NPF];
  RsCk cpf;
  auto fetch = [&](int tix_, int64_t icT_) {
    const int64_t T0_ = P.tile0 + (int64_t) tix_ * RS_TI;
    const int64_t rel0 = T0_ - (K - 1) - P.pos;                     // x index of tile sample 0
    if (rel0 >= 0 && rel0 + RS_TI + K <= P.n) {
      const T *xs = x + rel0;
#pragma unroll
      for (int j = 0; j < NPF; j++) {
        const int s_ = lane + j * 64;
        pf[j] = s_ < RS_TI + K ? xs[s_] : zero_of(T{});
      }
    } else {
#pragma unroll
      for (int j = 0; j < NPF; j++) {
        const int64_t rel = rel0 + lane + j * 64;
        T v = zero_of(T{});
        if (lane + j * 64 < RS_TI + K) {
          if (rel < 0) {
            if (hist && rel >= -(int64_t) (K - 1)) v = hist[(K - 1) + rel];
          } else if (rel < P.n) {
            v = x[rel];
          }
        }
        pf[j] = v;
      }
    }
    int64_t ic_ = icT_ + lane * RS_SEG, q_ = 0;
    rs_wrap(ic_, q_, P.mu, P.lambda);
    cpf.phase_bits = 0x40000000u;                                   // 2.0f: "emit nothing"
    cpf.cum = 0;
    if (T0_ + (int64_t) lane * RS_SEG < P.pos + P.n) cpf = ck[ic_ / RS_CK];
  };

  for (int ph = 0; ph < nphase; ph++) {
    // ---- this phase's taps [ka, kb) of every row
    const int ka = ph * kp, kb = min(K, ka + kp), kn = kb - ka;
    if (ph > 0) __syncthreads();                             // everybody has left the previous slice
    for (int i = threadIdx.x; i < (P.nph + 1) * kn; i += NT) {
      const int c = i / kn, k = i - c * kn;
      lut_s[c * lsp + k] = lut[c * P.gl + ka + k];
    }
    __syncthreads();
    const int nfull = kn / RSL_CH * RSL_CH, ktail = kn - nfull;       // whole chunks of this phase, taps of its last partial one

    int64_t icT = P.tile0 + (int64_t) wtile0 * RS_TI, qT = 0;
    rs_wrap(icT, qT, P.mu, P.lambda);
    if (wtile0 < ntiles) fetch(wtile0, icT);
    for (int tix = wtile0; tix < ntiles; tix += wstep) {
      const int64_t T0 = P.tile0 + (int64_t) tix * RS_TI;
#pragma unroll
      for (int j = 0; j < NPF; j++) {
        const int s_ = lane + j * 64;
        if (s_ < RS_TI + K) tile[rsl_slot(s_)] = pf[j];
      }
      const int64_t i_abs = T0 + (int64_t) lane * RS_SEG;
      int64_t ic = icT + lane * RS_SEG, q = qT;
      rs_wrap(ic, q, P.mu, P.lambda);
      const bool in_call = i_abs < P.pos + P.n;
      const float inc = P.inc;
      float phase = bits2f(cpf.phase_bits);
      int64_t cum = (int64_t) cpf.cum + q * P.opp;
      icT += icStep;
      rs_wrap(icT, qT, P.mu, P.lambda);
      if (tix + wstep < ntiles) fetch(tix + wstep, icT);

      if (in_call) {
        for (int s = (int) (ic % RS_CK); s > 0; s--) {
          while (phase < 1.f) { phase = phase + inc; cum++; }
          phase = phase - 1.f;
        }
      }
      const int64_t cum_t0 = ((int64_t) __shfl((int) (cum >> 32), 0) << 32) | (uint32_t) __shfl((int) (uint32_t) cum, 0);
      // ---- replay: the first output of every input stays with the lane, second ones go to the wave's list.  Branch-free: an
      // input has at most two outputs here (1/ratio >= 0.5, host), so the reference's while loop (ra.cc:64-73) is its two
      // possible turns written out -- the same float additions in the same order; counts relative to the tile in 32 bits.
      int col0[RS_SEG], o0[RS_SEG];
      unsigned has0 = 0;
      int n2 = 0;                                            // entries of rec2 so far (wave-uniform)
      int crel = (int) (cum - cum_t0);                       // outputs of the tile before the lane's next one
      // the lane's inputs [s_lo, s_hi) belong to this call
      const int s_lo = in_call ? (int) max((int64_t) 0, min((int64_t) RS_SEG, P.pos - i_abs)) : RS_SEG;
      const int s_hi = in_call ? (int) min((int64_t) RS_SEG, P.pos + P.n - i_abs) : 0;
#pragma unroll
      for (int s = 0; s < RS_SEG; s++) {
        const bool live = s >= s_lo && s < s_hi;
        const bool first = in_call && phase < 1.f;
        const float ph1 = phase + inc;                       // ra.cc:71, float32 add
        const bool second = first && ph1 < 1.f;
        const float ph2 = ph1 + inc;
        col0[s] = first ? (int) (phase * fnph) : 0;          // itrp.cc:19
        const int col1 = (int) (ph1 * fnph);
        o0[s] = crel;
        if (in_call) phase = (second ? ph2 : first ? ph1 : phase) - 1.f;       // ra.cc:73
        crel += (first ? 1 : 0) + (second ? 1 : 0);
        if (live && first) has0 |= 1u << s;
        const bool sec = live && second;
        const uint64_t m = __ballot(sec);
        if (sec) {
          const int at = n2 + __popcll(m & ((1ull << lane) - 1ull));
          if (at < cap2) rec2[at] = make_uint2(((uint32_t) (lane * RS_SEG + s) << 13) | (uint32_t) col1, (uint32_t) (o0[s] + 1));
        }
        n2 += __popcll(m);
      }
      n2 = min(n2, cap2);
      wave_sync();

      // ---- first outputs: taps in chunks of 16, the chunk's window in registers
      T *yt = y + (cum_t0 - P.cum_pos);
      T acc[RS_SEG];
#pragma unroll
      for (int s = 0; s < RS_SEG; s++) {
        acc[s] = zero_of(T{});
        if (ph > 0 && (has0 & (1u << s))) acc[s] = yt[o0[s]];
      }
      const T *wl = tile + 9 * lane + ka + (ka >> 3);        // slot of sample 8 lane + ka (ka is a multiple of 16)
      // (every input is evaluated, with column 0 where it has no output: straight-line code; all eight rows first, then tap-major
      // over the eight inputs -- eight independent accumulation chains and one exposed LDS latency per chunk)
      for (int k0 = 0; k0 < nfull; k0 += RSL_CH) {
        T X[RSL_WIN];
        const T *wc = wl + (k0 + (k0 >> 3));
#pragma unroll
        for (int j = 0; j < RSL_WIN; j++) X[j] = wc[j + (j >> 3)];
        float hh[RS_SEG][RSL_CH];
#pragma unroll
        for (int s = 0; s < RS_SEG; s++) {
          const float *h = lut_s + col0[s] * lsp + k0;
#pragma unroll
          for (int k4 = 0; k4 < RSL_CH / 4; k4++) {
            const float4 q4 = *reinterpret_cast<const float4 *>(h + 4 * k4);
            hh[s][4 * k4] = q4.x; hh[s][4 * k4 + 1] = q4.y; hh[s][4 * k4 + 2] = q4.z; hh[s][4 * k4 + 3] = q4.w;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < RSL_CH; k++) {
#pragma unroll
          for (int s = 0; s < RS_SEG; s++) acc[s] = tap_mac_pk(acc[s], hh[s][k], X[s + k]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ktail > 0) {
        T X[RSL_WIN];
        const T *wc = wl + nfull + (nfull >> 3);
#pragma unroll
        for (int j = 0; j < RSL_WIN; j++) X[j] = wc[j + (j >> 3)];
#pragma unroll
        for (int s = 0; s < RS_SEG; s++) {
          const float *h = lut_s + col0[s] * lsp + nfull;       // (the slice pitch is a multiple of 4 floats)
          float hh[RSL_CH];
#pragma unroll
          for (int k4 = 0; k4 < RSL_CH / 4; k4++) {
            float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * k4 < ktail) q4 = *reinterpret_cast<const float4 *>(h + 4 * k4);
            hh[4 * k4] = q4.x; hh[4 * k4 + 1] = q4.y; hh[4 * k4 + 2] = q4.z; hh[4 * k4 + 3] = q4.w;
          }
#pragma unroll
          for (int k = 0; k < RSL_CH; k++)
            if (k < ktail) acc[s] = tap_mac_pk(acc[s], hh[k], X[s + k]);
        }
      }
#pragma unroll
      for (int s = 0; s < RS_SEG; s++)
        if (has0 & (1u << s)) yt[o0[s]] = acc[s];

      // ---- second outputs: one per lane, samples and taps from LDS
      // (sample i0 + k sits in slot A + k + ((r + k) >> 3), A = slot of i0, r = i0 mod 8: with k = 8 m + e the run-time part is one of
      // eight base addresses per output, the rest an immediate offset -- no address arithmetic per tap)
      for (int e = lane; e < n2; e += 64) {
        const uint2 r = rec2[e];
        const int i0 = (int) (r.x >> 13) + ka, r8 = i0 & 7;
        const float *h = lut_s + (r.x & 8191u) * lsp;
        const T *be[8];
#pragma unroll
        for (int j = 0; j < 8; j++) be[j] = tile + rsl_slot(i0) + ((r8 + j) >> 3);
        T a = zero_of(T{});
        if (ph > 0) a = yt[r.y];
        const int K8 = kn & ~7;
        for (int k = 0; k < K8; k += 8) {
          const float4 qa = *reinterpret_cast<const float4 *>(h + k), qb = *reinterpret_cast<const float4 *>(h + k + 4);
          const int o9 = k + (k >> 3);
          a = tap_mac_pk(a, qa.x, be[0][o9]);
          a = tap_mac_pk(a, qa.y, be[1][o9 + 1]);
          a = tap_mac_pk(a, qa.z, be[2][o9 + 2]);
          a = tap_mac_pk(a, qa.w, be[3][o9 + 3]);
          a = tap_mac_pk(a, qb.x, be[4][o9 + 4]);
          a = tap_mac_pk(a, qb.y, be[5][o9 + 5]);
          a = tap_mac_pk(a, qb.z, be[6][o9 + 6]);
          a = tap_mac_pk(a, qb.w, be[7][o9 + 7]);
        }
        for (int k = K8; k < kn; k++) a = tap_mac_pk(a, h[k], tile[rsl_slot(i0 + k)]);
        yt[r.y] = a;
      }
      wave_sync();          // the sample image and the list are rewritten by the next iteration
    }
  }
}

// Dynamic hand-out of the tiles (the scheme of the overlap-save FIR, ols.hip: OlsDyn): a persistent grid with a static
// partition streams 6-9 % below the same bytes handed out in order (scripts/ubench/copy_shapes.hip).  NC counters on their
// own 128-B lines, workgroup g pulls from counter (g / 8) % NC (its pullers sit on all 8 XCDs), a pulled value v stands for
// tile v * NC + c; never reset: each launch starts from `base` and advances every counter by Q + pullers-per-counter.
constexpr int RS_MAX_CTR = 32;
struct RsDyn {
  unsigned *ctr;
  unsigned base, Q;
  int NC;            // 0: static partition (tile = wave + k * waves)
};
constexpr int RS15_TILE_PAD = (RS_TI + 16) + (RS_TI + 16) / 8 + 2;      // padded sample slots per wave (lane stride 9: sample s at s + (s >> 3))

// ---- K = 15 (the filtre_reechan interpolator), at most two outputs per input (ratio <= 2) ----------------------------------------
// Every wave owns 512-input tiles: a lane replays the phase recurrence for its 8 inputs, evaluates the FIRST output of each
// from a 22-sample register window (taps: four ds_read_b128 of a table row of pitch 20 floats) and hands the second outputs to
// a per-wave list.  (Rounds 1-3 ran the reference's `while (phase < 1)` loop per lane: whenever ONE of the 64 lanes has a second
// output on an input -- ratio 160/147: 9 % of the inputs, i.e. practically always somewhere in the wave -- the whole wave took
// the second turn: 16 table rows and 16 x 15 multiply-adds per lane for 8.7 outputs; 0.527 ms per 2^27 inputs against 0.462 now.)
// The replay is branch-free and runs first (1/ratio > 0.5:
// an input has 0, 1 or 2 outputs), a wave scan of the per-lane counts gives every output its place in the tile -- so the per-lane
// checkpoint is the PHASE alone (4 B per 8 inputs instead of 8: the schedule table is the kernel's only traffic beyond its
// samples) --, the FIRST output of each input is evaluated straight-line from the 22-sample register window, and the second
// outputs go to a per-wave list and are evaluated one per lane from the sample image.  Same arithmetic per output (taps 0 .. 14
// in the reference's order), so the outputs are bit for bit the other kernel's.
// The samples come in by LDS-DMA (global_load_lds): no prefetch registers, no ds_write pass.  The padded image (lane stride 9
// slots: sample s at slot s + (s >> 3)) is filled by giving every lane the SOURCE address of its 16 image bytes (the destination
// of an LDS-DMA is lane-linear): a chunk that starts on a pad slot starts one sample early, and whatever lands on a pad is never
// read.  The image is free again once the window is in registers and the second outputs are done: the next tile's DMA is issued
// there and lands under the first-output pass.  Tiles that touch the ends of the call (history before it, nothing behind it) are
// loaded through registers with bounds checks.
template <typename T> struct Rs15sDma;
template <> struct Rs15sDma<float2> {
  static constexpr int SPC = 2;       // image slots per DMA lane (16 B)
  static __device__ __forceinline__ void issue(const float2 *src, unsigned lds_dst)
  {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
  }
};
template <> struct Rs15sDma<float> {
  static constexpr int SPC = 1;       // 4 B
  static __device__ __forceinline__ void issue(const float *src, unsigned lds_dst)
  {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
  }
};
constexpr int RS15S_CKB = 272;        // bytes of a wave's checkpoint area in LDS: 64 phases + the output count, 16-B granular
constexpr int RS15S_SRC_SPAN = 8 * ((RS15_TILE_PAD - 1) / 9) + 8 + 1;      // samples a full image fill may read (from the tile's first)

template <typename T>
__global__ __launch_bounds__(1024) void resample15s_kernel(const T *__restrict__ x, const T *__restrict__ hist, T *__restrict__ y,
                                                           const float *__restrict__ lut, const RsCk *__restrict__ ck,
                                                           const uint32_t *__restrict__ phs, RsParams P, int ntiles,
                                                           T *__restrict__ hist_next, RsDyn dyn, int lcap)
{
  const int NW = (int) (blockDim.x >> 6);
  if (blockIdx.x == gridDim.x - 1) {      // (see resample_kernel: the next window history rides in this launch)
    for (int i = threadIdx.x; i < 14; i += (int) blockDim.x) {
      const int64_t g = P.n - 14 + i;
      hist_next[i] = g < 0 ? (hist ? hist[14 + g] : zero_of(T{})) : x[g];
    }
    return;
  }
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int K = 15, LS = 20, IMG = RS15_TILE_PAD, SPC = Rs15sDma<T>::SPC;
  constexpr int NCH = (IMG + SPC - 1) / SPC, NDMA = (NCH + 63) / 64;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));     // (the wave index in an SGPR: what follows from it is scalar)
  float *lut_s = reinterpret_cast<float *>(smem_raw);
  char *wbase = reinterpret_cast<char *>(lut_s + ((P.nph + 1) * LS + 3) / 4 * 4);
  const size_t wbytes = ((size_t) (IMG + P.rec_cap) * sizeof(T) + (size_t) lcap * sizeof(uint32_t) + RS15S_CKB + 15) / 16 * 16;
  uint32_t *ckl = reinterpret_cast<uint32_t *>(wbase + wv * wbytes);     // the tile's checkpoint: 64 phases, then the outputs before it
  T *img = reinterpret_cast<T *>(ckl + RS15S_CKB / 4);       // padded: sample s at s + (s >> 3)
  T *obuf = img + IMG;                                       // the tile's outputs, in order
  uint32_t *list = reinterpret_cast<uint32_t *>(obuf + P.rec_cap);       // second outputs: image sample (9 bits) | column << 9 (9) | place << 18
  const unsigned img_lds = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (size_t) img);
  const unsigned ckl_lds = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (size_t) ckl);

  for (int i = threadIdx.x; i < (P.nph + 1) * K; i += (int) blockDim.x) {
    const int c = i / K, k = i - c * K;
    lut_s[c * LS + k] = lut[c * P.gl + k];
  }
  for (int c = threadIdx.x; c <= P.nph; c += (int) blockDim.x) lut_s[c * LS + 15] = 0.f;
  __syncthreads();

  auto wave_sync = []() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const int wtile0 = blockIdx.x * NW + wv;
  const int wstep = (gridDim.x - 1) * NW;
  int64_t ic0 = P.tile0, q0 = 0;
  rs_wrap(ic0, q0, P.mu, P.lambda);
  auto tile_ic = [&](int tix_, int64_t &ic_, int64_t &q_) {
    ic_ = ic0 + (int64_t) tix_ * RS_TI;
    q_ = q0;
    if (P.lambda > 0) {
      const int64_t lim = P.mu + P.lambda;
      if (ic_ >= lim) {
        if (ic_ - lim < 64 * P.lambda) {
          do { ic_ -= P.lambda; q_++; } while (ic_ >= lim);       // (wave-uniform: scalar instructions)
        } else {
          const int64_t d = ic_ - P.mu, k = d / P.lambda;
          q_ += k;
          ic_ = P.mu + (d - k * P.lambda);
        }
      }
    }
  };
  const int ctr_c = dyn.NC > 0 ? (int) ((blockIdx.x / 8) % dyn.NC) : 0;
  // the next tile of this wave: the pull is ISSUED early and its value TAKEN a tile's evaluation later (dynamic: from the wave's
  // counter, -1 once the quota is spent -- exactly one failing pull per wave; static: prev + waves)
  auto pull_issue = [&]() -> unsigned {
    unsigned v = 0;
    if (dyn.NC > 0 && lane == 0) v = __hip_atomic_fetch_add(dyn.ctr + ctr_c * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
  };
  auto pull_take = [&](unsigned v, int prev) -> int {
    if (dyn.NC == 0) {
      const int t = prev < 0 ? wtile0 : prev + wstep;
      return t < ntiles ? t : -1;
    }
    for (;;) {
      v = (unsigned) __builtin_amdgcn_readfirstlane((int) v) - dyn.base;
      if (v >= dyn.Q) return -1;
      const int64_t t = (int64_t) v * dyn.NC + ctr_c;
      if (t < ntiles) return (int) t;
      v = pull_issue();
    }
  };

  // what the tile being loaded brings besides its samples: the lane's phase at its first input, the outputs before the tile
  // (the checkpoint comes in by LDS-DMA like the samples: a register load would be counted by hipcc, whose waits for it --
  // wherever it schedules a copy of the register -- also wait for the DMA in flight)
  bool pf_in = false;       // the lane has inputs inside the call
  int pf_lead = 0;          // inputs between the lane's checkpoint and its first input (a period that is no multiple of 8 shifts the grid)
  int64_t pf_cum0 = 0;      // whole periods' outputs before the tile (the checkpoint's own count is in ckl[64])
  // per-lane element offsets of the DMA sources inside a tile's sample range (loop-invariant: 32-bit, one register each)
  unsigned doff[NDMA];
#pragma unroll
  for (int d = 0; d < NDMA; d++) {
    const int c = d * 64 + lane, p0 = c * SPC, g = p0 / 9, r = p0 - 9 * g;
    doff[d] = (unsigned) (8 * g + min(r, 7));
  }
  auto fetch = [&](int tix_) {
    int64_t icT_, qT_;
    tile_ic(tix_, icT_, qT_);
    const int64_t T0_ = P.tile0 + (int64_t) tix_ * RS_TI;
    const int64_t rel0 = T0_ - (K - 1) - P.pos;              // image sample 0, relative to x
    if (rel0 >= 0 && rel0 + RS15S_SRC_SPAN <= P.n) {
      const T *xs = x + rel0;                                // (wave-uniform)
#pragma unroll
      for (int d = 0; d < NDMA; d++) {
        unsigned off = doff[d];
        asm volatile("" : "+v"(off));                          // (keeps the hoisted form 32 bits wide: one register per DMA, not two)
        if (d * 64 + lane < NCH) Rs15sDma<T>::issue(xs + off, img_lds + (unsigned) (d * 64 * SPC * (int) sizeof(T)));
      }
    } else {
      // (an end of the call: one or two tiles per launch -- a plain loop, two live registers)
#pragma unroll 1
      for (int s_ = lane; s_ < RS_TI + K; s_ += 64) {
        const int64_t rel = rel0 + s_;
        T v = zero_of(T{});
        if (rel < 0) {
          if (hist && rel >= -(int64_t) (K - 1)) v = hist[(K - 1) + rel];
        } else if (rel < P.n) {
          v = x[rel];
        }
        img[s_ + (s_ >> 3)] = v;
      }
    }
    // the lane's phase at its first input: table index = canonical index / 8 -- the tile's plus the lane, unless the period of
    // the phase sequence ends inside the tile (then lane by lane)
    const int64_t left_ = P.pos + P.n - T0_;                  // inputs from the tile's first to the call's end (wave-uniform)
    const int lefti = left_ > RS_TI ? RS_TI : (int) left_;
    pf_in = RS_SEG * lane < lefti;
    const uint32_t *psrc = phs;                               // (lanes behind the call's end read entry 0 and ignore it)
    pf_lead = 0;
    if (pf_in) {
      if (P.lambda <= 0 || icT_ + RS_TI <= P.mu + P.lambda) {
        psrc = phs + (icT_ / RS_CK + lane);
        pf_lead = (int) (icT_ % RS_CK);
      } else {
        int64_t ic_ = icT_ + lane * RS_SEG, q_ = 0;
        rs_wrap(ic_, q_, P.mu, P.lambda);
        psrc = phs + ic_ / RS_CK;
        pf_lead = (int) (ic_ % RS_CK);
      }
    }
    Rs15sDma<float>::issue(reinterpret_cast<const float *>(psrc), ckl_lds);
    if (lane == 0) Rs15sDma<float>::issue(reinterpret_cast<const float *>(&ck[icT_ / RS_CK].cum), ckl_lds + 256u);
    pf_cum0 = qT_ * P.opp;
  };

  // three tiles are known at any time: the one evaluated, the one being loaded, and the one after -- whose pull is issued at
  // the top of an iteration and taken at the top of the next, behind the wait for the samples that comes there anyway
  int tix = pull_take(pull_issue(), -1);
  int ntix = tix >= 0 ? pull_take(pull_issue(), tix) : -1;
  unsigned pulled = ntix >= 0 ? pull_issue() : 0u;
  if (tix >= 0) fetch(tix);
  const float inc = P.inc, fnph = (float) P.nph;
  // The outputs of a tile are stored at the START of the next iteration (before its second outputs go to the staging buffer):
  // vmcnt retires in issue order, so stores issued behind the next tile's DMA would stand between it and the wait at the loop
  // top -- every tile would pay a store round trip.  Issued here they are a whole first-output pass old when that wait comes.
  int pend_begin = 0, pend_end = 0;
  T *pend_y = y;
  auto flush = [&]() {
    for (int oo = pend_begin + lane; oo < pend_end; oo += 64) pend_y[oo] = obuf[oo];
    pend_end = 0;
  };
  while (tix >= 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the tile's DMA has landed (hipcc does not count asm loads)
    wave_sync();
    const int64_t T0 = P.tile0 + (int64_t) tix * RS_TI;
    int64_t cum_t0 = pf_cum0 + (int64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) ckl[64]);
    // inputs of this lane inside the call's end (an input behind it produces nothing here: it belongs to the next call)
    const int64_t left = P.pos + P.n - T0;                   // (wave-uniform)
    const int nlive = min(RS_SEG, max(0, (left > RS_TI ? RS_TI : (int) left) - RS_SEG * lane));
    const int n2tix = ntix >= 0 ? pull_take(pulled, ntix) : -1;
    pulled = n2tix >= 0 ? pull_issue() : 0u;

    // ---- branch-free replay (ra.cc:64-73 with at most two turns of the while loop: inc > 0.5 and phase >= 0)
    float ph[RS_SEG];
    unsigned m1 = 0, m2 = 0;
    {
      float p = pf_in ? bits2f(ckl[lane]) : 2.0f;             // (2.0f: no output)
      // from the checkpoint to the lane's first input (nothing when the tile sits on the checkpoint grid); lane 0's outputs
      // on the way come before the tile
      int led = 0;
      for (int r = pf_lead; r > 0; r--) {
        const bool a0 = p < 1.f;
        const float p1 = p + inc;
        const bool a1 = a0 && (p1 < 1.f);
        led += (a0 ? 1 : 0) + (a1 ? 1 : 0);
        p = a0 ? (a1 ? p1 + inc : p1) : p;
        p = p - 1.f;
      }
      cum_t0 += __builtin_amdgcn_readfirstlane(led);
#pragma unroll
      for (int s = 0; s < RS_SEG; s++) {
        ph[s] = p;
        const bool e0 = (p < 1.f) && (s < nlive);
        const float p1 = p + inc;
        const bool e1 = e0 && (p1 < 1.f);
        const float p2 = p1 + inc;
        m1 |= e0 ? (1u << s) : 0u;
        m2 |= e1 ? (1u << s) : 0u;
        p = (p < 1.f) ? ((p1 < 1.f) ? p2 : p1) : p;           // phase after the loop ...
        p = p - 1.f;                                         // ... ra.cc:73
      }
    }
    // places: exclusive wave scan of (outputs, second outputs) per lane, packed
    const unsigned mine = ((unsigned) (__builtin_popcount(m1) + __builtin_popcount(m2)) << 16) | (unsigned) __builtin_popcount(m2);
    unsigned incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o_ = (unsigned) __shfl_up((int) incl, d);
      if (lane >= d) incl += o_;
    }
    const unsigned tot = (unsigned) __builtin_amdgcn_readlane((int) incl, 63);
    const int total = (int) (tot >> 16), nsec = (int) (tot & 0xffffu);
    const unsigned excl = incl - mine;
    const int o_base = (int) (excl >> 16);
    // the list of second outputs
    {
      int o_run = o_base, e_run = (int) (excl & 0xffffu);
#pragma unroll
      for (int s = 0; s < RS_SEG; s++) {
        o_run += (m1 >> s) & 1u;
        if ((m2 >> s) & 1u) {
          const int col2 = (int) ((ph[s] + inc) * fnph);
          list[e_run] = (unsigned) (RS_SEG * lane + s) | ((unsigned) col2 << 9) | ((unsigned) o_run << 18);
          e_run++;
          o_run++;
        }
      }
    }
    wave_sync();
    flush();                                                 // the previous tile's outputs (see above)
    // ---- second outputs, one per lane: window from the image (sample m + k at slot(m + e) + 9 * (k >> 3), e = k & 7)
    for (int e = lane; e < nsec; e += 64) {
      const uint32_t rc = list[e];
      const int m = (int) (rc & 511u);
      const float *h = lut_s + (int) ((rc >> 9) & 511u) * LS;
      float hh[16];
#pragma unroll
      for (int k4 = 0; k4 < 4; k4++) {
        const float4 q4 = *reinterpret_cast<const float4 *>(h + 4 * k4);
        hh[4 * k4] = q4.x; hh[4 * k4 + 1] = q4.y; hh[4 * k4 + 2] = q4.z; hh[4 * k4 + 3] = q4.w;
      }
      asm volatile("" :: "v"(hh[15]));
      const T *be[8];
#pragma unroll
      for (int j = 0; j < 8; j++) be[j] = img + (m + j) + ((m + j) >> 3);
      T sv[15];
      if constexpr (sizeof(T) == 8) {
        // (one statement for the same reason as the window's: no ds_read2_b64)
        asm volatile(
            "ds_read_b64 %0, %15\n\tds_read_b64 %1, %16\n\tds_read_b64 %2, %17\n\tds_read_b64 %3, %18\n\t"
            "ds_read_b64 %4, %19\n\tds_read_b64 %5, %20\n\tds_read_b64 %6, %21\n\tds_read_b64 %7, %22\n\t"
            "ds_read_b64 %8, %15 offset:72\n\tds_read_b64 %9, %16 offset:72\n\tds_read_b64 %10, %17 offset:72\n\tds_read_b64 %11, %18 offset:72\n\t"
            "ds_read_b64 %12, %19 offset:72\n\tds_read_b64 %13, %20 offset:72\n\tds_read_b64 %14, %21 offset:72\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(sv[0]), "=&v"(sv[1]), "=&v"(sv[2]), "=&v"(sv[3]), "=&v"(sv[4]), "=&v"(sv[5]), "=&v"(sv[6]), "=&v"(sv[7]),
              "=&v"(sv[8]), "=&v"(sv[9]), "=&v"(sv[10]), "=&v"(sv[11]), "=&v"(sv[12]), "=&v"(sv[13]), "=&v"(sv[14])
            : "v"((unsigned) (size_t) be[0]), "v"((unsigned) (size_t) be[1]), "v"((unsigned) (size_t) be[2]), "v"((unsigned) (size_t) be[3]),
              "v"((unsigned) (size_t) be[4]), "v"((unsigned) (size_t) be[5]), "v"((unsigned) (size_t) be[6]), "v"((unsigned) (size_t) be[7])
            : "memory");
      } else {
#pragma unroll
        for (int k = 0; k < 15; k++) sv[k] = be[k & 7][9 * (k >> 3)];
      }
      T acc = zero_of(T{});
#pragma unroll
      for (int k = 0; k < 15; k++) acc = tap_mac(acc, hh[k], sv[k]);
      obuf[rc >> 18] = acc;
    }
    // register window (read AFTER the second outputs: their 15 sample reads and the window need not be live together): W[j] = sample 8*lane + j of the image (slot 9*lane + j + (j >> 3))
    T W[22];
    if constexpr (sizeof(T) == 8) {
      // 22 ds_read_b64 and their wait in ONE statement: hipcc pairs adjacent 8-byte reads into ds_read2_b64, which moves half
      // the bytes per LDS cycle (MI355X_MICROARCH.md, LDS table: 8 cycles for 16 B per lane against 2 x 2)
      const unsigned wa = (unsigned) (size_t) (img + 9 * lane);
#define RS_W(j) "ds_read_b64 %" #j ", %22 offset:%c" 
      asm volatile(
          "ds_read_b64 %0, %22 offset:0\n\tds_read_b64 %1, %22 offset:8\n\tds_read_b64 %2, %22 offset:16\n\tds_read_b64 %3, %22 offset:24\n\t"
          "ds_read_b64 %4, %22 offset:32\n\tds_read_b64 %5, %22 offset:40\n\tds_read_b64 %6, %22 offset:48\n\tds_read_b64 %7, %22 offset:56\n\t"
          "ds_read_b64 %8, %22 offset:72\n\tds_read_b64 %9, %22 offset:80\n\tds_read_b64 %10, %22 offset:88\n\tds_read_b64 %11, %22 offset:96\n\t"
          "ds_read_b64 %12, %22 offset:104\n\tds_read_b64 %13, %22 offset:112\n\tds_read_b64 %14, %22 offset:120\n\tds_read_b64 %15, %22 offset:128\n\t"
          "ds_read_b64 %16, %22 offset:144\n\tds_read_b64 %17, %22 offset:152\n\tds_read_b64 %18, %22 offset:160\n\tds_read_b64 %19, %22 offset:168\n\t"
          "ds_read_b64 %20, %22 offset:176\n\tds_read_b64 %21, %22 offset:184\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(W[0]), "=&v"(W[1]), "=&v"(W[2]), "=&v"(W[3]), "=&v"(W[4]), "=&v"(W[5]), "=&v"(W[6]), "=&v"(W[7]), "=&v"(W[8]),
            "=&v"(W[9]), "=&v"(W[10]), "=&v"(W[11]), "=&v"(W[12]), "=&v"(W[13]), "=&v"(W[14]), "=&v"(W[15]), "=&v"(W[16]),
            "=&v"(W[17]), "=&v"(W[18]), "=&v"(W[19]), "=&v"(W[20]), "=&v"(W[21])
          : "v"(wa)
          : "memory");
#undef RS_W
    } else {
      const T *wl = img + 9 * lane;
#pragma unroll
      for (int j = 0; j < 22; j++) W[j] = wl[j + (j >> 3)];
    }
    // ---- the image is free: the next tile's samples start coming in
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wave_sync();
    if (ntix >= 0) fetch(ntix);
    // ---- first outputs from the register window (filtrage.hpp:1877-1879 order)
    {
      int o_run = o_base;
#pragma unroll
      for (int s = 0; s < RS_SEG; s++) {
        if ((m1 >> s) & 1u) {
          const float *h = lut_s + (int) (ph[s] * fnph) * LS;                    // itrp.cc:19
          float hh[16];
#pragma unroll
          for (int k4 = 0; k4 < 4; k4++) {
            const float4 q4 = *reinterpret_cast<const float4 *>(h + 4 * k4);
            hh[4 * k4] = q4.x; hh[4 * k4 + 1] = q4.y; hh[4 * k4 + 2] = q4.z; hh[4 * k4 + 3] = q4.w;
          }
          asm volatile("" :: "v"(hh[15]));                    // (keeps the row four ds_read_b128: a b96 takes 8 LDS cycles, a b128 4)
          T acc = zero_of(T{});
#pragma unroll
          for (int k = 0; k < 15; k++) acc = tap_mac(acc, hh[k], W[s + k]);
          obuf[o_run] = acc;
        }
        o_run += ((m1 >> s) & 1u) + ((m2 >> s) & 1u);
        __builtin_amdgcn_sched_barrier(0);       // (one table row in flight: hoisting all eight costs 128 registers)
      }
    }
    pend_begin = (int) max((int64_t) 0, P.cum_pos - cum_t0);
    pend_end = total;
    pend_y = y + (cum_t0 - P.cum_pos);
    tix = ntix;
    ntix = n2tix;
  }
  wave_sync();
  flush();
}


}  // namespace tsdgpu

using namespace tsdgpu;

// The schedule of the float32 phase recurrence depends on the increment alone: it is simulated
// ONCE per ratio and per process -- checkpoints every RS_CK inputs up to the end of the first
// period -- and shared by all the handles of that ratio (a rééchan()-style one-shot call creates
// a handle per call: without the cache every call replayed up to 8 M sequential steps, tens of ms).
struct RsSched {
  std::mutex mtx;
  std::vector<RsCk> ck;       // checkpoints every RS_CK inputs of the canonical sequence
  int64_t sim_i = 0;          // inputs simulated so far (canonical)
  float sim_phase = 0.f;
  int64_t sim_cum = 0;
  // Brent cycle detection on the phase at input boundaries
  uint32_t tort_bits = 0;
  int64_t brent_power = 1, brent_lam = 0;
  bool cyc = false;
  int64_t mu = 0, lambda = 0, opp = 0;
  // the table grows 8 B per RS_CK inputs until the period is found, and counts outputs in 32 bits: a
  // recurrence whose period were not found within RS_SIM_MAX inputs is refused instead of growing on
  bool failed = false;
};
constexpr int64_t RS_SIM_MAX = (int64_t) 1 << 29;      // inputs (a 256 MiB table); 160/147 needs 8 M
static std::shared_ptr<RsSched> sched_for(float inc)
{
  static std::mutex m;
  static std::map<uint32_t, std::shared_ptr<RsSched>> cache;
  std::lock_guard<std::mutex> lock(m);
  uint32_t key;
  memcpy(&key, &inc, 4);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  if (cache.size() >= 32) cache.erase(cache.begin());        // (handles keep their own reference)
  auto sp = std::make_shared<RsSched>();
  cache[key] = sp;
  return sp;
}

struct tsdgpu_resampler {
  int data_type = 0, K = 0, nph = 0, lstride = 0, gl = 0, mode = 0;
  float ratio = 1.f, inc = 1.f;
  float inv_den[32] = {0};
  float *d_lut = nullptr;
  void *d_hist[2] = {nullptr, nullptr};
  bool hist_zero = true;       // the window history is all zeros (start, reset, seek without samples): the kernels get a null pointer instead of a cleared buffer
  int cur = 0;
  // stream position
  int64_t pos = 0, cum_pos = 0;
  // host schedule (shared by every handle of the same ratio, see RsSched)
  std::shared_ptr<RsSched> sch;
  std::vector<RsCk> &ck;
  int64_t &sim_i;
  float &sim_phase;
  int64_t &sim_cum;
  uint32_t &tort_bits;
  int64_t &brent_power, &brent_lam;
  bool &cyc;
  int64_t &mu, &lambda, &opp;
  explicit tsdgpu_resampler(std::shared_ptr<RsSched> s_)
      : sch(std::move(s_)), ck(sch->ck), sim_i(sch->sim_i), sim_phase(sch->sim_phase), sim_cum(sch->sim_cum),
        tort_bits(sch->tort_bits), brent_power(sch->brent_power), brent_lam(sch->brent_lam), cyc(sch->cyc), mu(sch->mu),
        lambda(sch->lambda), opp(sch->opp) {}
  RsCk *d_ck = nullptr;
  uint32_t *d_ph = nullptr;     // the phases alone (resample15s_kernel: 4 B per checkpoint), behind d_ck in its allocation
  size_t d_ck_cap = 0, d_ck_n = 0;
  DevBuf in_stage, out_stage;
  // work counters of the fused kernel's dynamic tile hand-out (RsDyn), behind the histories in d_lut's allocation
  unsigned *d_ctr = nullptr;
  unsigned ctr_base = 0;
  int ctr_nc = 0;
};

namespace {

inline void sim_one(float &phase, int64_t &cum, float inc)
{
  // plain binary32 adds (SSE scalar ops; this file is compiled without fast-math or contraction)
  float p = phase;
  while (p < 1.f) { p = p + inc; cum++; }
  p = p - 1.f;
  phase = p;
}

// state (phase, outputs before) at canonical input index ic, ic <= sim_i
void state_at_canonical(const tsdgpu_resampler *r, int64_t ic, float *phase, int64_t *cum)
{
  const RsCk &c = r->ck[(size_t) (ic / RS_CK)];
  float p = bits2f(c.phase_bits);
  int64_t cu = c.cum;
  for (int64_t s = ic % RS_CK; s > 0; s--) sim_one(p, cu, r->inc);
  *phase = p;
  *cum = cu;
}

// advance the host simulation until it covers canonical index `upto` or the cycle is known
void extend(tsdgpu_resampler *r, int64_t upto)
{
  while (!r->cyc && !r->sch->failed && r->sim_i < upto + RS_CK) {
    if (r->sim_i >= RS_SIM_MAX || r->sim_cum >= (int64_t) 0xFFFF0000ll) {
      r->sch->failed = true;
      return;
    }
    if (r->sim_i % RS_CK == 0) r->ck.push_back(RsCk{f2bits(r->sim_phase), (uint32_t) r->sim_cum});
    // Brent: compare the state at this input boundary with the tortoise
    const uint32_t bits = f2bits(r->sim_phase);
    if (r->sim_i == 0) {
      r->tort_bits = bits;
    } else {
      r->brent_lam++;
      if (bits == r->tort_bits) {
        r->lambda = r->brent_lam;
        // mu = first index whose state recurs lambda inputs later
        float pa = 0.f, pb;
        int64_t ca = 0, cb;
        state_at_canonical(r, r->lambda, &pb, &cb);
        int64_t m = 0;
        while (f2bits(pa) != f2bits(pb)) {
          sim_one(pa, ca, r->inc);
          sim_one(pb, cb, r->inc);
          m++;
        }
        r->mu = m;
        r->opp = cb - ca;
        r->cyc = true;
        // make sure the table covers [0, mu + lambda + RS_CK)
        while (r->sim_i < r->mu + r->lambda + 2 * RS_CK) {
          sim_one(r->sim_phase, r->sim_cum, r->inc);
          r->sim_i++;
          if (r->sim_i % RS_CK == 0) r->ck.push_back(RsCk{f2bits(r->sim_phase), (uint32_t) r->sim_cum});
        }
        return;
      }
      if (r->brent_lam == r->brent_power) {
        r->tort_bits = bits;
        r->brent_power *= 2;
        r->brent_lam = 0;
      }
    }
    sim_one(r->sim_phase, r->sim_cum, r->inc);
    r->sim_i++;
  }
}

// outputs emitted before absolute input index i (and the phase there)
void state_at(tsdgpu_resampler *r, int64_t i, float *phase, int64_t *cum)
{
  extend(r, i);
  int64_t ic = i, q = 0;
  if (r->cyc && i >= r->mu + r->lambda) {
    const int64_t d = i - r->mu;
    q = d / r->lambda;
    ic = r->mu + d % r->lambda;
  }
  state_at_canonical(r, ic, phase, cum);
  *cum += q * r->opp;
}

int sync_table(tsdgpu_resampler *r, hipStream_t st)
{
  if (r->d_ck_n == r->ck.size()) return TSDGPU_OK;
  if (r->ck.size() > r->d_ck_cap) {
    RsCk *nd = nullptr;
    const size_t cap = r->ck.size() + r->ck.size() / 2 + 1024;
    if (hipMalloc((void **) &nd, cap * (sizeof(RsCk) + sizeof(uint32_t))) != hipSuccess)
      return set_err(TSDGPU_ERR_ALLOC, "resampler: schedule table alloc failed");
    TSD_HIP(hipStreamSynchronize(st));
    if (r->d_ck) (void) hipFree(r->d_ck);
    r->d_ck = nd;
    r->d_ph = reinterpret_cast<uint32_t *>(nd + cap);
    r->d_ck_cap = cap;
    r->d_ck_n = 0;
  }
  const size_t fresh = r->ck.size() - r->d_ck_n;
  std::vector<uint32_t> phases(fresh);
  for (size_t i = 0; i < fresh; i++) phases[i] = r->ck[r->d_ck_n + i].phase_bits;
  TSD_HIP(hipMemcpyAsync(r->d_ck + r->d_ck_n, r->ck.data() + r->d_ck_n, fresh * sizeof(RsCk), hipMemcpyHostToDevice, st));
  TSD_HIP(hipMemcpyAsync(r->d_ph + r->d_ck_n, phases.data(), fresh * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  TSD_HIP(hipStreamSynchronize(st));    // the vectors may grow (reallocate) / go away before the copies have run
  r->d_ck_n = r->ck.size();
  return TSDGPU_OK;
}

}  // namespace

// LDS bytes per workgroup of the generic kernel for this handle (the arithmetic of tsdgpu_resampler_step):
// checked at creation so that a configuration no step could launch -- a large ratio with wide complex
// tiles -- is refused there instead of failing at every step
// Geometry of the generic kernel for this handle: is the table staged in LDS, and how many waves share it.  A table of up to
// 48 KiB leaves room for RS_WAVES waves; a LONG interpolator's table (itrp_sinc{127, 256, ...} of the reference's test_ra_unit:
// 257 rows of 132 floats = 133 KiB) is staged too when at least two waves' tiles still fit beside it -- read through L2
// instead (a row per lane per tap quadruple: 64 cache lines per wave instruction) it cost 5.7 ms per 2^26 inputs.
struct RsGeom {
  bool lut_in_lds;
  int waves;
  size_t lds;
};
static RsGeom rs_geometry(const tsdgpu_resampler *r, int mode)
{
  const size_t sz = dtype_size(r->data_type);
  const int rec_cap = ((int) ((double) RS_TI * (double) r->ratio * 1.0001) + 32 + 3) / 4 * 4;
  const size_t wbytes = ((size_t) rs_tile_elems(r->K) * sz + (size_t) rec_cap * (mode ? 8 : 4) + 15) / 16 * 16;
  const int lstride = r->K == 15 ? 20 : r->lstride;
  const size_t lut = ((size_t) (r->nph + 1) * lstride + 4) * 4;
  RsGeom g;
  g.lut_in_lds = lut - 16 <= (size_t) RS_LUT_LDS_BYTES;
  g.waves = RS_WAVES;
  if (!g.lut_in_lds && mode == 0 && lut + 64 + 2 * wbytes <= RS_LDS_LIMIT) {
    g.lut_in_lds = true;
    g.waves = (int) std::min<size_t>(RS_WAVES, (RS_LDS_LIMIT - lut - 64) / wbytes);
  }
  g.lds = (g.lut_in_lds ? lut : 32) + (size_t) g.waves * wbytes + 64;
  return g;
}
static size_t rs_lds_need(const tsdgpu_resampler *r, int mode) { return rs_geometry(r, mode).lds; }
// geometry of resample_long_kernel, or waves = 0 when the handle does not qualify (table-driven, 24 taps and more, at most two
// outputs per input, table + one wave within the LDS); TSDGPU_RS_LONG=0 keeps such handles on resample_kernel
struct RsLongGeom {
  int waves, cap2, kp, nphase;
  size_t lds;
};
static RsLongGeom rs_long_geometry(const tsdgpu_resampler *r)
{
  RsLongGeom g = {0, 0, 0, 0, 0};
  // (read per call: the tests flip them between handles)
  const char *e_off = dev_switch("RS_LONG"), *e_kmin = dev_switch("RS_LONG_KMIN"), *e_w = dev_switch("RS_LONG_WAVES"),
             *e_ph = dev_switch("RS_LONG_PHASES");
  const bool off = e_off && atoi(e_off) == 0;
  const int kmin = e_kmin ? atoi(e_kmin) : 24;
  const int want = e_w ? atoi(e_w) : 8;                      // waves per workgroup aimed at
  const int force_ph = e_ph ? atoi(e_ph) : 0;
  if (off || r->mode != 0 || r->K < kmin || !(r->inc >= 0.5f)) return g;
  const size_t sz = dtype_size(r->data_type);
  g.cap2 = r->ratio > 1.f ? ((int) ((double) RS_TI * ((double) r->ratio - 1.0) * 1.0001) + 40 + 1) / 2 * 2 : 8;
  const size_t wbytes = ((size_t) rsl_tile_slots(r->K) * sz + (size_t) g.cap2 * 8 + 15) / 16 * 16;
  // as few tap phases as leave room for `want` waves (at most 8: the kernel's bound); a table too large for that takes the phase
  // count that leaves the most waves, down to one
  const int kc = (r->K + RSL_CH - 1) / RSL_CH;               // chunks of taps
  int best_w = 0;
  for (int nph = 1; nph <= kc; nph++) {
    if (force_ph > 0 && nph != std::min(force_ph, kc)) continue;
    const int kp = (kc + nph - 1) / nph * RSL_CH;
    if ((int64_t) kp * (nph - 1) >= r->K) continue;          // (an empty last phase: the same slices as a smaller count)
    const size_t lut = (size_t) (r->nph + 1) * (kp + 4) * 4;
    if (lut + wbytes + 64 > RS_LDS_LIMIT) continue;
    const int w = (int) std::min<size_t>(8, (RS_LDS_LIMIT - lut - 64) / wbytes);
    if (w > best_w) {
      best_w = w;
      g.waves = w;
      g.kp = kp;
      g.nphase = nph;
      g.lds = lut + (size_t) w * wbytes + 64;
    }
    if (w >= std::min(want, 8)) break;
  }
  return g;
}

extern "C" {

int tsdgpu_resampler_create(tsdgpu_resampler **out, int data_type, float ratio, const float *lut_host, int K,
                            int nphases)
{
  TSD_CHECK(out != nullptr, "resampler_create: out is NULL");
  *out = nullptr;
  TSD_CHECK(data_type == TSDGPU_F32 || data_type == TSDGPU_C64, "resampler_create: bad data_type %d", data_type);
  TSD_CHECK(lut_host != nullptr, "resampler_create: NULL lut");
  TSD_CHECK(std::isfinite(ratio) && ratio > 0.f, "resampler_create: invalid ratio %g", (double) ratio);
  if (ratio > 8.f || ratio < 1.f / 64.f)
    return set_err(TSDGPU_ERR_UNSUPPORTED, "resampler_create: ratio %g outside [1/64, 8] (filtre_reechan folds "
                   "ratios into [0.5,2) with half-band stages first)", (double) ratio);
  if (K < 1 || K > RS_KMAX || nphases < 1 || nphases > 8191)
    return set_err(TSDGPU_ERR_UNSUPPORTED, "resampler_create: K=%d nphases=%d unsupported (K <= %d, nphases <= 8191)", K, nphases, RS_KMAX);
  tsdgpu_resampler *r = new tsdgpu_resampler(sched_for(1.f / ratio));
  r->data_type = data_type;
  r->K = K;
  r->nph = nphases;
  r->gl = (K + 3) / 4 * 4;
  r->lstride = r->gl + (((r->gl / 4) & 1) ? 0 : 4);      // LDS pitch: a multiple of 4 floats with an odd number of 16-B units
  r->ratio = ratio;
  r->inc = 1.f / ratio;                      // ra.cc:29
  const size_t lut_bytes = (size_t) (nphases + 1) * ((K + 3) / 4 * 4) * sizeof(float);
  // ONE allocation (the table with its rows padded to 16 bytes, then the two window histories) and ONE upload of its host
  // image: a one-shot rééchan() pays for every creation (three allocations, a memset, a strided copy, two memsets and a
  // synchronisation before)
  const size_t hb = ((size_t) std::max(K - 1, 1) * dtype_size(data_type) + 15) / 16 * 16, lb = (lut_bytes + 15) / 16 * 16;
  int rc = TSDGPU_OK;
  const size_t ctr_bytes = (size_t) RS_MAX_CTR * 128;
  std::vector<char> image(lb + 2 * hb + ctr_bytes, 0);
  for (int c = 0; c <= nphases; c++)
    std::memcpy(image.data() + (size_t) c * r->gl * sizeof(float), lut_host + (size_t) c * K, (size_t) K * sizeof(float));
  if (hipMalloc((void **) &r->d_lut, image.size()) != hipSuccess)
    rc = set_err(TSDGPU_ERR_HIP, "resampler_create: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
  else if (hipMemcpy(r->d_lut, image.data(), image.size(), hipMemcpyHostToDevice) != hipSuccess)
    rc = set_err(TSDGPU_ERR_HIP, "resampler_create: upload failed: %s", hipGetErrorString(hipGetLastError()));
  if (!rc) {
    r->d_hist[0] = (char *) r->d_lut + lb;
    r->d_hist[1] = (char *) r->d_lut + lb + hb;
    r->d_ctr = (unsigned *) ((char *) r->d_lut + lb + 2 * hb);      // (lb and hb are multiples of 16 bytes)
  }
  if (rc) {
    tsdgpu_resampler_destroy(r);
    return rc;
  }
  {
    const size_t lds = rs_lds_need(r, 0);
    if (lds > 158 * 1024) {
      tsdgpu_resampler_destroy(r);
      return set_err(TSDGPU_ERR_UNSUPPORTED, "resampler_create: ratio %g with %d taps on %s data needs %zu bytes of LDS per workgroup (limit 158 KiB): "
                     "fold the ratio with half-band stages first (filtre_reechan does)", (double) ratio, K, data_type == TSDGPU_C64 ? "complex" : "real", lds);
    }
  }
  (void) hipFuncSetAttribute((const void *) resample_kernel<float, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void) hipFuncSetAttribute((const void *) resample_kernel<float2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void) hipFuncSetAttribute((const void *) resample_long_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void) hipFuncSetAttribute((const void *) resample_long_kernel<float2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void) hipFuncSetAttribute((const void *) resample15s_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void) hipFuncSetAttribute((const void *) resample15s_kernel<float2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void) hipGetLastError();
  *out = r;
  return TSDGPU_OK;
}

int tsdgpu_resampler_create_analytic(tsdgpu_resampler **out, int data_type, float ratio, int kind, int degree)
{
  TSD_CHECK(out != nullptr, "resampler_create_analytic: out is NULL");
  *out = nullptr;
  TSD_CHECK(kind == TSDGPU_ITRP_LINEAR || kind == TSDGPU_ITRP_LAGRANGE, "resampler_create_analytic: kind %d", kind);
  TSD_CHECK(kind == TSDGPU_ITRP_LINEAR || (degree >= 1 && degree <= 31), "resampler_create_analytic: Lagrange degree %d (1..31)", degree);
  const int K = kind == TSDGPU_ITRP_LINEAR ? 2 : degree + 1;
  std::vector<float> dummy((size_t) 2 * K, 0.f);       // no table: the taps come from the phase
  int rc = tsdgpu_resampler_create(out, data_type, ratio, dummy.data(), K, 1);
  if (rc) return rc;
  (*out)->mode = kind;
  if (rs_lds_need(*out, kind) > 158 * 1024) {
    tsdgpu_resampler_destroy(*out);
    *out = nullptr;
    return set_err(TSDGPU_ERR_UNSUPPORTED, "resampler_create_analytic: ratio %g needs more LDS than a workgroup has: fold the ratio first", (double) ratio);
  }
  for (int j = 0; j < K; j++) {
    double den = 1.0;
    for (int k = 0; k < K; k++)
      if (k != j) den *= (double) (j - k);
    (*out)->inv_den[j] = (float) (1.0 / den);
  }
  return TSDGPU_OK;
}

int64_t tsdgpu_resampler_out_count(tsdgpu_resampler *r, int64_t n)
{
  if (!r || n < 0) return -1;
  float ph;
  int64_t c;
  std::lock_guard<std::mutex> lock(r->sch->mtx);
  state_at(r, r->pos + n, &ph, &c);
  if (r->sch->failed) {
    set_err(TSDGPU_ERR_UNSUPPORTED, "resampler: the float32 phase recurrence of ratio %g shows no period within 2^29 inputs", (double) r->ratio);
    return -1;
  }
  return c - r->cum_pos;
}

int tsdgpu_resampler_step(tsdgpu_resampler *r, const void *x, int64_t n, void *y, int64_t y_capacity, int64_t *n_out,
                          void *stream)
{
  TSD_CHECK(r != nullptr, "resampler_step: NULL handle");
  TSD_CHECK(n >= 0, "resampler_step: negative length");
  if (n_out) *n_out = 0;
  if (n == 0) return TSDGPU_OK;
  TSD_CHECK(x != nullptr, "resampler_step: NULL input");
  hipStream_t st = (hipStream_t) stream;
  // a large HOST vector goes through in chunks: H2D of chunk i+1, the kernel of chunk i and D2H of chunk i-1 overlap
  // (position, window and output offset are carried from chunk to chunk exactly as between two calls)
  if ((size_t) n * dtype_size(r->data_type) >= PIPE_MIN_BYTES && host_pipe_enabled() && y != nullptr && !is_device_ptr(x) && !is_device_ptr(y)) {
    const size_t esz = dtype_size(r->data_type);
    const int64_t total = tsdgpu_resampler_out_count(r, n);
    if (total < 0) return TSDGPU_ERR_INVALID;
    TSD_CHECK(total <= y_capacity, "resampler_step: output needs %lld samples, capacity is %lld", (long long) total, (long long) y_capacity);
    if (!host_ranges_overlap(x, (size_t) n * esz, y, (size_t) total * esz)) {
      const double ratio = (double) r->ratio;
      return pipelined_host_step_var(
          x, n, esz, y, esz, n_out, 1, st, [ratio](int64_t c) { return (int64_t) ((double) c * ratio * 1.0001) + 64; },
          [r](const void *cx, void *cy, int64_t cnt, int64_t cap, int64_t *got, hipStream_t q) {
            return tsdgpu_resampler_step(r, cx, cnt, cy, cap, got, q);
          });
    }
  }
  float ph_end;
  int64_t cum_end;
  // (the schedule is shared between the handles of a ratio: held while this call reads or extends it)
  std::lock_guard<std::mutex> lock(r->sch->mtx);
  state_at(r, r->pos + n, &ph_end, &cum_end);
  if (r->sch->failed)
    return set_err(TSDGPU_ERR_UNSUPPORTED, "resampler_step: the float32 phase recurrence of ratio %g shows no period within 2^29 inputs",
                   (double) r->ratio);
  const int64_t nout = cum_end - r->cum_pos;
  TSD_CHECK(nout <= y_capacity, "resampler_step: output needs %lld samples, capacity is %lld", (long long) nout,
            (long long) y_capacity);
  TSD_CHECK(nout == 0 || y != nullptr, "resampler_step: NULL output");
  int rc = sync_table(r, st);
  if (rc) return rc;
  const size_t sz = dtype_size(r->data_type);
  const void *dx = nullptr;
  void *dy = nullptr;
  bool staged = false;
  rc = stage_in(x, (size_t) n * sz, r->in_stage, st, &dx);
  if (rc) return rc;
  rc = stage_out(y, (size_t) nout * sz, r->out_stage, &dy, &staged);
  if (rc) return rc;

  RsParams P;
  P.pos = r->pos;
  P.n = n;
  P.tile0 = (r->pos / RS_TI) * RS_TI;
  P.mu = r->cyc ? r->mu : 0;
  P.lambda = r->cyc ? r->lambda : 0;
  P.opp = r->opp;
  P.cum_pos = r->cum_pos;
  P.inc = r->inc;
  P.K = r->K;
  P.nph = r->nph;
  P.lstride = r->lstride;
  P.mode = r->mode;
  P.gl = r->gl;
  for (int j = 0; j < 32; j++) P.inv_den[j] = r->inv_den[j];
  // at most floor(1/inc)+1 outputs per input
  // outputs are spaced 1/ratio apart in input time (up to float32 rounding of the adds)
  P.rec_cap = ((int) ((double) RS_TI * (double) r->ratio * 1.0001) + 32 + 3) / 4 * 4;
  const int64_t tiles = cdiv(r->pos + n - P.tile0, RS_TI);
  TSD_CHECK(tiles <= 0x7fffffff, "resampler_step: n too large for one launch");
  const RsGeom geo = rs_geometry(r, r->mode);
  P.lut_in_lds = geo.lut_in_lds ? 1 : 0;
  const size_t lds = geo.lds;
  TSD_CHECK(lds <= RS_LDS_LIMIT, "resampler_step: configuration needs %zu bytes of LDS", lds);
  // persistent workgroups (the LUT is staged once per workgroup): as many as stay resident
  int per_cu = (int) std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (lds + 1024)));
  const int64_t pgrid = std::min<int64_t>(cdiv(tiles, geo.waves), (int64_t) 256 * per_cu);
  const void *hcur = r->hist_zero ? nullptr : r->d_hist[r->cur];
  unsigned ctr_add = 0;
  const char *e15 = dev_switch("RS15");                     // =0: the K = 15 interpolator through the kernels of the other lengths (parity tests)
  // K = 15 at a ratio below 2 (at most two outputs per input): the split kernel, as many waves per workgroup as its LDS allows
  const int rec15 = ((int) ((double) RS_TI * (double) r->ratio * 1.0001) + 8 + 3) / 4 * 4;
  const int lcap15 = (int) ((double) RS_TI * std::max(0.0, (double) r->ratio - 1.0) * 1.0001) + 8;
  const size_t wb15s = ((size_t) (RS15_TILE_PAD + rec15) * sz + (size_t) lcap15 * 4 + RS15S_CKB + 15) / 16 * 16;
  const size_t lut15 = (size_t) ((r->nph + 1) * 20 + 4) * 4;
  const int nw15s = (int) std::min<size_t>(16, (RS_LDS_LIMIT - lut15 - 64) / wb15s);
  if (r->K == 15 && r->mode == 0 && r->inc >= 0.5f && r->nph <= 511 && nw15s >= 4 && !(e15 && atoi(e15) == 0)) {
    const int NW = nw15s;
    P.rec_cap = rec15;
    int64_t g15 = std::min<int64_t>(cdiv(tiles, NW), 256);
    const char *nc_s = dev_switch("RS_DYN");
    int NC = nc_s ? atoi(nc_s) : 16;
    if (NC < 0 || NC > RS_MAX_CTR || !r->d_ctr || 256 % (8 * std::max(NC, 1)) != 0 || stream_is_capturing(st)) NC = 0;
    const char *min_s = dev_switch("RS_DYN_MIN");
    if (tiles < (int64_t) (min_s ? atoi(min_s) : 4) * 256 * NW) NC = 0;
    RsDyn dyn = {r->d_ctr, r->ctr_base, 0u, NC};
    if (NC > 0) {
      g15 = 256;
      if (NC != r->ctr_nc) {
        TSD_HIP(hipMemsetAsync(r->d_ctr, 0, (size_t) RS_MAX_CTR * 128, st));
        r->ctr_base = 0;
        r->ctr_nc = NC;
        dyn.base = 0;
      }
      dyn.Q = (unsigned) cdiv(tiles, NC);
      ctr_add = dyn.Q + (unsigned) (g15 / NC * NW);
    }
    const size_t lds15s = lut15 + (size_t) NW * wb15s + 64;
    if (r->data_type == TSDGPU_C64)
      hipLaunchKernelGGL(resample15s_kernel<float2>, dim3((unsigned) g15 + 1), dim3(64 * NW), lds15s, st, (const float2 *) dx,
                         (const float2 *) hcur, (float2 *) dy, r->d_lut, r->d_ck, r->d_ph, P, (int) tiles, (float2 *) r->d_hist[r->cur ^ 1], dyn, lcap15);
    else
      hipLaunchKernelGGL(resample15s_kernel<float>, dim3((unsigned) g15 + 1), dim3(64 * NW), lds15s, st, (const float *) dx,
                         (const float *) hcur, (float *) dy, r->d_lut, r->d_ck, r->d_ph, P, (int) tiles, (float *) r->d_hist[r->cur ^ 1], dyn, lcap15);
  } else if (const RsLongGeom lg = rs_long_geometry(r); lg.waves > 0) {
    const int64_t lgrid = std::min<int64_t>(cdiv(tiles, lg.waves), 256);
    if (r->data_type == TSDGPU_C64)
      hipLaunchKernelGGL(resample_long_kernel<float2>, dim3((unsigned) lgrid + 1), dim3(64 * lg.waves), lg.lds, st, (const float2 *) dx,
                         (const float2 *) hcur, (float2 *) dy, r->d_lut, r->d_ck, P, (int) tiles, (float2 *) r->d_hist[r->cur ^ 1], lg.cap2, lg.kp, lg.nphase);
    else
      hipLaunchKernelGGL(resample_long_kernel<float>, dim3((unsigned) lgrid + 1), dim3(64 * lg.waves), lg.lds, st, (const float *) dx,
                         (const float *) hcur, (float *) dy, r->d_lut, r->d_ck, P, (int) tiles, (float *) r->d_hist[r->cur ^ 1], lg.cap2, lg.kp, lg.nphase);
  } else {
#define RS_LAUNCH(T, KT)                                                                                            \
  hipLaunchKernelGGL((resample_kernel<T, KT>), dim3((unsigned) pgrid + 1), dim3(64 * geo.waves), lds, st, (const T *) dx, \
                     (const T *) hcur, (T *) dy, r->d_lut, r->d_ck, P, (int) tiles, (T *) r->d_hist[r->cur ^ 1])
  if (r->data_type == TSDGPU_C64) RS_LAUNCH(float2, 0); else RS_LAUNCH(float, 0);
#undef RS_LAUNCH
  }
  if (const hipError_t le = hipGetLastError(); le != hipSuccess) {
    if (ctr_add) r->ctr_nc = 0;   // (device counters and host base may have parted: the next dynamic launch zeroes them)
    return set_err(TSDGPU_ERR_HIP, "resampler_step: launch failed: %s", hipGetErrorString(le));
  }
  r->ctr_base += ctr_add;
  if (r->K > 1) r->cur ^= 1;      // (the launch wrote the next window history into the other buffer)
  r->hist_zero = false;
  r->pos += n;
  r->cum_pos = cum_end;
  if (n_out) *n_out = nout;
  return finish_out(y, (size_t) nout * sz, dy, staged, st);
}

int tsdgpu_resampler_reset(tsdgpu_resampler *r)
{
  TSD_CHECK(r != nullptr, "resampler_reset: NULL handle");
  r->pos = 0;
  r->cum_pos = 0;
  r->hist_zero = true;
  return TSDGPU_OK;
}

int tsdgpu_resampler_seek(tsdgpu_resampler *r, int64_t pos, const void *hist, void *stream)
{
  TSD_CHECK(r != nullptr && pos >= 0, "resampler_seek: bad argument");
  hipStream_t st = (hipStream_t) stream;
  float ph;
  int64_t c;
  {
    std::lock_guard<std::mutex> lock(r->sch->mtx);
    state_at(r, pos, &ph, &c);
    if (r->sch->failed)
      return set_err(TSDGPU_ERR_UNSUPPORTED, "resampler_seek: the float32 phase recurrence of ratio %g shows no period within 2^29 inputs",
                     (double) r->ratio);
  }
  r->pos = pos;
  r->cum_pos = c;
  if (hist && r->K > 1) {
    const size_t hb = (size_t) (r->K - 1) * dtype_size(r->data_type);
    if (is_device_ptr(hist)) {
      const int rc = device_copy_small(r->d_hist[r->cur], hist, hb, st);
      if (rc) return rc;
    } else {
      TSD_HIP(hipMemcpyAsync(r->d_hist[r->cur], hist, hb, hipMemcpyHostToDevice, st));
      TSD_HIP(hipStreamSynchronize(st));
    }
    r->hist_zero = false;
  } else {
    r->hist_zero = true;          // (no launch: the kernels take a null history as zeros)
  }
  return TSDGPU_OK;
}

int64_t tsdgpu_resampler_out_offset(const tsdgpu_resampler *r) { return r ? r->cum_pos : -1; }

int tsdgpu_resampler_destroy(tsdgpu_resampler *r)
{
  if (!r) return TSDGPU_OK;
  if (r->d_lut) (void) hipFree(r->d_lut);            // (the window histories live in the same allocation)
  if (r->d_ck) (void) hipFree(r->d_ck);
  r->in_stage.release();
  r->out_stage.release();
  delete r;
  return TSDGPU_OK;
}

}  // extern "C"
