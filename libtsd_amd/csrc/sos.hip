// sos.hip -- cascade of second-order IIR sections (biquads) as a block-parallel recursion.
//
// Stands behind ChaineSOIS<T,T,T>::step / filtre_sois<T>  (libtsd core/src/filtrage/
// filtre-rt.cc:303-400,407-437,440-602): direct-form-II biquads
//     d = x - a1*d1 - a2*d2;  y = b0*d + b1*d1 + b2*d2        (:373-379)
// chained section after section, each section seeded on the very first sample of the stream
// with d1 = d2 = (its own first input) (:361-365), then `y *= gain` or a trailing first-order
// section (:567-570).  The reference makes one sequential pass over the vector per section;
// here ALL sections are applied in ONE pass over HBM (8 B/sample for float data).
//
// Parallel formulation (exact linear algebra, not an approximation of the recurrence):
//  * a wave64 owns a sub-tile of 2048 floats; each lane holds 32 consecutive floats in
//    registers (complex data = two interleaved real channels, the coefficients being real);
//  * per section the lane runs the recurrence from ZERO state over its samples, the 64
//    zero-state end states are combined by a Kogge-Stone scan of affine maps over the wave
//    (wave shuffles; all maps share the linear part M^L, so the scan only carries the two
//    state values), and every sample is corrected by the zero-input response of the lane's
//    true start state (two FMAs against tabulated responses).  The corrected outputs are the
//    next section's inputs, still in registers;
//  * a wave walks a chunk of consecutive sub-tiles carrying the state exactly; chunk 0 starts
//    from the stream state carried by the handle, every other chunk starts W samples early
//    from zero state, W chosen on the host so that the cascade's state-transition matrix
//    satisfies ||Phi^W||_inf <= 1e-9 (chunks are >= 4 W long; a filter that decays slowly
//    simply gets fewer, longer chunks -- down to a single sequential one);
//  * global accesses are 16 B per lane, transposed lane<->sample through padded LDS.
#include "common.hpp"
#include <cmath>
#include <cstdlib>
#include <vector>

namespace tsdgpu {

constexpr int SOS_MAX_SEC = 32;
constexpr int LANE_FLOATS = 32;                   // floats a lane holds per sub-tile (16 was measured: twice the scans and LDS work per sample)
constexpr int SUB_FLOATS = 64 * LANE_FLOATS;      // floats per sub-tile (2048)
constexpr int LDS_LANE_PITCH = LANE_FLOATS + 4;   // floats: + 4 pad -> conflict-free b128 both ways (36: 9 x 16 B, odd; 20: 5 x 16 B, odd)
constexpr int LANE_QUADS = LANE_FLOATS / 4;
// float offset of float p (a multiple of 4) of a sub-tile in the wave's LDS image; row = the lane that owns it
__device__ __forceinline__ int sos_img(int p)
{
  return (p / LANE_FLOATS) * LDS_LANE_PITCH + (p % LANE_FLOATS);
}

constexpr int SOS_WARM_FACTOR = 4;    // a chunk is at least this many times its warm-up's cost long
constexpr int NARROW_FLOATS = 4;      // floats per lane of a warm-up step (one 16-B load, no transposition)

struct SosSection {
  float b0, b1, b2, a1, a2;
  float seed;                 // 1: first-sample seed (SOIS), 0: zero start (RIIFoS)
  float df1;                  // 1: FormeDirecte1 (filtre-rt.cc:384-393), 0: FormeDirecte2 (:369-380)
  float sg;                   // +1 / -1: the scans carry (d1, delta = d1 - sg d2) instead of (d1, d2), see sos_cascade
  float A[6][4];              // T (M^L)^(2^k) T^-1, k = 0..5, row-major 2x2, M = [[-a1,-a2],[1,0]], T = [[1,0],[1,-sg]]
  float c1[LANE_FLOATS];      // output response to start state (d1, delta) = (1, 0) (per in-lane sample index)
  float c2[LANE_FLOATS];      // output response to start state (d1, delta) = (0, 1)
  // the same tables for the narrow warm-up steps (L = NARROW_FLOATS / channels samples per lane)
  float An[6][4];
  float c1n[NARROW_FLOATS], c2n[NARROW_FLOATS];
  // Levels of the Kogge-Stone scan that matter: after K levels a lane's sum holds the terms of the 2^K lanes before it, and
  // the first term left out is (M^L)^(2^K) Z = A[K] Z -- below 1e-9 of the states for a damped section long before the sixth
  // level (pole radius 0.88: three levels; 0.67: two; 0.5 and less: one).  The bound of the chunk warm-ups (||Phi^W|| <= 1e-9).
  int nlev, nlevn;
};

// state buffer layout (floats): [0] = seeded flag, then per (section, channel) four values:
// DF2: (d1, d2, -, -);  DF1: (y1, y2, x1, x2)
__host__ __device__ inline int state_index(int sec, int ch) { return 1 + (sec * 2 + ch) * 4; }
constexpr int STATE_FLOATS = 1 + SOS_MAX_SEC * 8;

__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The stream state of the handle is the reference's: per (section, channel) (d1, d2, x1, x2).  The running image a wave keeps
// in LDS holds (d1, delta = d1 - sg d2, x1, x2) -- the coordinates of the scans, see sos_cascade.
__device__ __forceinline__ void state_load(float *sst, const float *__restrict__ st, const SosSection *__restrict__ sec, int nsec, int lane)
{
  for (int i = lane; i < nsec * 8; i += 64) {
    float v = st[1 + i];
    if ((i & 3) == 1) v = st[i] - sec[i >> 3].sg * v;
    sst[i] = v;
  }
}
__device__ __forceinline__ void state_store(float *__restrict__ st, const float *sst, const SosSection *__restrict__ sec, int nsec, int lane)
{
  if (lane == 0) st[0] = 1.f;
  for (int i = lane; i < nsec * 8; i += 64) st[1 + i] = (i & 3) == 1 ? sec[i >> 3].sg * (sst[i - 1] - sst[i]) : sst[i];
}

// One cascade pass over the wave's samples held in registers: lane l owns LF consecutive floats
// (LF / NCH samples per channel), v in, v out.  Per section: zero-state run, Kogge-Stone scan of the
// end states over the 64 lanes, zero-input correction of every sample; the running state of every
// (section, channel) lives in sst (LDS) and is advanced to the end of these 64 * LF floats.
// NARROW selects the tables of the NARROW_FLOATS-per-lane warm-up steps.
// Coordinates of the carried state: (d1, delta = d1 - sg d2), sg = the sign of the poles' real part.  A narrow-band
// section has its poles next to +1 (or -1): M^n ~ [[n+1, -n], [n, -(n-1)]], and a DC level of 10^6 in (d1, d2) -- what
// a cut-off of 1e-4 makes of an offset of 0.5 -- went through the scan as the difference of products of 10^9: the
// states came out with an absolute error of ~100 where the sequential recurrence has 0.1 (outputs 10-25 x noisier than
// the reference's own float32 run against float64).  In (level, slope) coordinates the same maps are
// ~[[1, n], [~0, 1]]: no cancellation.  The zero-state run of a lane produces moderate values, so its delta is exact.
template <int NCH, int LF, bool NARROW>
__device__ __forceinline__ void sos_cascade(float (&v)[LF], const SosSection *__restrict__ sec, int nsec, float *sst, int lane,
                                            bool do_seed, int last = 63)
{
  // `last`: the lane whose end state is carried on (63; a partial sub-tile of the ragged end stops at an earlier lane)
  constexpr int L = LF / NCH;
#pragma unroll 1
  for (int s = 0; s < nsec; s++) {
    const SosSection &k = sec[s];
    const float b0 = k.b0, b1 = k.b1, b2 = k.b2, a1 = k.a1, a2 = k.a2;
    // (the scan's tables loaded here, beside the coefficients -- one scalar-cache round trip for both; read level by level
    // inside the scan, each level waited for its own)
    float A[6][4];
    {
      const float(*Ag)[4] = NARROW ? k.An : k.A;
#pragma unroll
      for (int q = 0; q < 6; q++)
#pragma unroll
        for (int j = 0; j < 4; j++) A[q][j] = Ag[q][j];
    }
    const float *c1 = NARROW ? k.c1n : k.c1, *c2 = NARROW ? k.c2n : k.c2;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      float *ss = &sst[(s * 2 + c) * 4];
      float sin1 = ss[0], sin0 = ss[1];
      float xin1 = ss[2], xin2 = ss[3];                      // DF1 only: previous two inputs
      if (do_seed && k.seed != 0.f) {
        // premier_appel: every memory of the section = its own first input (filtre-rt.cc:361-365)
        const float x0 = __shfl(v[c], 0);
        sin1 = xin1 = xin2 = x0;
        sin0 = x0 - k.sg * x0;
      }
      float d1 = 0.f, d2 = 0.f;
      if (k.df1 == 0.f) {
        // DF2 zero-state run over the lane's L samples (b0 = 1 -- what the pole / zero pairing of
        // filtre_sois always produces -- saves the multiply; same value, 1.0f * d being exact)
        if (b0 == 1.f) {
#pragma unroll
          for (int i = 0; i < L; i++) {
            const float xin = v[i * NCH + c];
            const float d = fmaf(-a2, d2, fmaf(-a1, d1, xin));
            v[i * NCH + c] = fmaf(b2, d2, fmaf(b1, d1, d));
            d2 = d1;
            d1 = d;
          }
        } else {
#pragma unroll
          for (int i = 0; i < L; i++) {
            const float xin = v[i * NCH + c];
            const float d = fmaf(-a2, d2, fmaf(-a1, d1, xin));
            v[i * NCH + c] = fmaf(b2, d2, fmaf(b1, d1, b0 * d));
            d2 = d1;
            d1 = d;
          }
        }
      } else {
        // DF1: v = b0 x + b1 x[-1] + b2 x[-2] (previous lane's last two inputs for i < 2),
        // then the all-pole recursion y = v - a1 y1 - a2 y2 from zero state
        static_assert(L >= 2, "a lane holds at least two samples per channel");
        const float my1 = v[(L - 1) * NCH + c], my2 = v[(L - 2) * NCH + c];
        float xp1 = __shfl_up(my1, 1), xp2 = __shfl_up(my2, 1);
        if (lane == 0) { xp1 = xin1; xp2 = xin2; }
        if (lane == last) { ss[2] = my1; ss[3] = my2; }
#pragma unroll
        for (int i = 0; i < L; i++) {
          const float xin = v[i * NCH + c];
          const float fir = fmaf(b2, xp2, fmaf(b1, xp1, b0 * xin));
          const float yv = fmaf(-a2, d2, fmaf(-a1, d1, fir));
          v[i * NCH + c] = yv;
          xp2 = xp1;
          xp1 = xin;
          d2 = d1;
          d1 = yv;
        }
      }
      // lane 0 absorbs the start state: P = M^L * S_in + Z
      float p1 = d1, p0 = fmaf(-k.sg, d2, d1);
      if (lane == 0) {
        p1 = fmaf(A[0][0], sin1, fmaf(A[0][1], sin0, p1));
        p0 = fmaf(A[0][2], sin1, fmaf(A[0][3], sin0, p0));
      }
      // inclusive Kogge-Stone scan over the 64 lanes: P_l = sum_j (M^L)^(l-j) Z_j, cut where the powers have died out (nlev)
      const int nlev = NARROW ? k.nlevn : k.nlev;
#pragma unroll
      for (int kk = 0; kk < 6; kk++) {
        if (kk >= nlev) break;
        const int dd = 1 << kk;
        const float q1 = __shfl_up(p1, dd), q0 = __shfl_up(p0, dd);
        if (lane >= dd) {
          p1 = fmaf(A[kk][0], q1, fmaf(A[kk][1], q0, p1));
          p0 = fmaf(A[kk][2], q1, fmaf(A[kk][3], q0, p0));
        }
      }
      // true start state of this lane = end state of the previous lane
      float s1 = __shfl_up(p1, 1), s0 = __shfl_up(p0, 1);
      if (lane == 0) { s1 = sin1; s0 = sin0; }
      // zero-input correction of every output of the lane
#pragma unroll
      for (int i = 0; i < L; i++) v[i * NCH + c] = fmaf(c1[i], s1, fmaf(c2[i], s0, v[i * NCH + c]));
      // state after the last sample, carried on
      if (lane == last) {
        ss[0] = p1;
        ss[1] = p0;
      }
    }
    wave_sync();
  }
}

// NCH = 1: real samples; NCH = 2: interleaved complex (two real channels).
// Chunk c owns sub-tiles [c spc, (c+1) spc).  Chunk 0 starts from the stream state; every other chunk
// starts from zero state `warm_sub` whole sub-tiles plus `warm_nar` narrow steps (256 floats each, one
// 16-B load per lane, a 4-float recurrence per lane and section: a fifth of a sub-tile's work) before
// its first sample -- the host picks them so that the state transition over the warm-up is below 1e-9.
//
// MODE 0 is that scheme.  A filter whose memory is long against the call (a DC blocker, a smoother with a cut-off of
// 1e-4: warm-ups of 10^5 samples and more, or no decay at all) would leave it a handful of chunks -- down to ONE wave
// walking the whole vector.  Such calls carry the state EXACTLY instead (tsdgpu_sos_step):
//   MODE 1  every chunk but the last runs from zero state (chunk 0: from the stream state) WITHOUT storing outputs and
//           publishes its end state E_c = carry[c];
//   (sos_carry_scan_kernel turns the E_c into the true start states: S_{c+1} = Phi^L S_c + E_c, S_1 = E_0)
//   MODE 2  every chunk starts from carry[c] = S_c and stores its outputs: no warm-up.
// 12 B per sample instead of 8, whatever the pole radius.
constexpr int SOS_WPE = 4;            // waves per SIMD the kernel is compiled for (a fifth costs spills: profiles/EXPERIMENTS.md)
template <int NCH, int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SOS_WPE, SOS_WPE))) void sos_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                 const SosSection *__restrict__ sec, int nsec, float gain,
                                                 const float *__restrict__ st_in, float *__restrict__ st_out,
                                                 int64_t n_sub, int spc, int warm_sub, int warm_nar, float *__restrict__ carry,
                                                 int64_t skip_f)
{
  if (MODE == 1 && blockIdx.x + 1 == gridDim.x) return;       // nobody starts from the last chunk's end state
  __shared__ __attribute__((aligned(16))) float lds[64 * LDS_LANE_PITCH];
  __shared__ float sst[SOS_MAX_SEC * 8];          // running state per (section, channel): 4 floats
  const int lane = threadIdx.x;
  const int64_t chunk = blockIdx.x;
  const int64_t t_first = chunk * spc;                         // first sub-tile whose output we own
  const int64_t t_last = min(t_first + spc, n_sub);            // exclusive
  const bool first_chunk = chunk == 0;
  const bool seeded = st_in[0] != 0.f;

  // running state per section and channel: wave-uniform, kept in LDS (indexed by the
  // runtime section number; a register array would go to scratch)
  if (first_chunk) state_load(sst, st_in, sec, nsec, lane);
  else
    for (int i = lane; i < nsec * 8; i += 64) sst[i] = MODE == 2 ? carry[(size_t) chunk * (nsec * 8) + i] : 0.f;
  wave_sync();

  int64_t t = t_first;
  if (MODE == 0 && !first_chunk) {
    t = t_first - warm_sub;
    // narrow warm-up steps come first in time: they cover the floats just before sub-tile t
    const float *xw = x + t * SUB_FLOATS - (int64_t) warm_nar * (64 * NARROW_FLOATS);
    for (int w = 0; w < warm_nar; w++) {
      const float4 q = *reinterpret_cast<const float4 *>(xw + (int64_t) w * (64 * NARROW_FLOATS) + 4 * lane);
      float v4[NARROW_FLOATS] = {q.x, q.y, q.z, q.w};
      sos_cascade<NCH, NARROW_FLOATS, true>(v4, sec, nsec, sst, lane, false);
    }
  }

  for (; t < t_last; t++) {
    const float *xt = x + t * SUB_FLOATS;
    // ---- load 16 B per lane, transpose through LDS: lane gets floats [32*lane, 32*lane+32)
    float v[LANE_FLOATS];
#pragma unroll
    for (int i = 0; i < LANE_QUADS; i++) {
      const int p = 4 * (i * 64 + lane);                       // float index in the sub-tile
      const float4 q = *reinterpret_cast<const float4 *>(xt + p);
      *reinterpret_cast<float4 *>(&lds[sos_img(p)]) = q;
    }
    wave_sync();
#pragma unroll
    for (int i = 0; i < LANE_QUADS; i++) {
      const float4 q = *reinterpret_cast<const float4 *>(&lds[sos_img(lane * LANE_FLOATS + 4 * i)]);
      v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
    }
    wave_sync();

    sos_cascade<NCH, LANE_FLOATS, false>(v, sec, nsec, sst, lane, first_chunk && !seeded && t == 0);

    if (MODE != 1 && t >= t_first) {
      // ---- gain, transpose back, store 16 B per lane
#pragma unroll
      for (int i = 0; i < LANE_QUADS; i++) {
        const float4 q = make_float4(v[4 * i] * gain, v[4 * i + 1] * gain, v[4 * i + 2] * gain, v[4 * i + 3] * gain);
        *reinterpret_cast<float4 *>(&lds[sos_img(lane * LANE_FLOATS + 4 * i)]) = q;
      }
      wave_sync();
      float *yt = y + t * SUB_FLOATS;
      // (tsdgpu_sos_step_skip: the first skip_f floats of the call -- a multiple of 4 -- are filtered but not stored)
      const int64_t keep_from = skip_f - t * SUB_FLOATS;
#pragma unroll
      for (int i = 0; i < LANE_QUADS; i++) {
        const int p = 4 * (i * 64 + lane);
        if (p >= keep_from) *reinterpret_cast<float4 *>(yt + p) = *reinterpret_cast<const float4 *>(&lds[sos_img(p)]);
      }
      wave_sync();
    }
  }

  if (MODE == 1) {
    for (int i = lane; i < nsec * 8; i += 64) carry[(size_t) chunk * (nsec * 8) + i] = sst[i];
    return;
  }
  // the wave that owns the last sub-tile publishes the stream state
  if (t_last == n_sub && t_last > t_first) state_store(st_out, sst, sec, nsec, lane);
}

// The chunks' zero-state end states E_c (carry[c], the layout of the running state image: per (section, channel) four
// floats, the first two being (d1, delta)) -> the chunks' true start states, in place: carry[c] <- S_c, c >= 1, with
//   S_1 = E_0 (chunk 0 ran from the stream state),  S_{c+1} = Phi^L S_c + E_c,
// i.e. S_{c+1} = T_c, the running sums T_c = sum_{i <= c} (Phi^L)^(c-i) E_i.  A sequential recurrence of C matrix-vector
// products would take a millisecond, a Kogge-Stone scan from one workgroup half of one (measured: each of its levels
// waits on global memory): the chunks go in blocks of CARRY_BLOCK instead, a wave per block walking it with the vector
// and the matrix in LDS, everything in double:
//   phase 1  (a wave per block)   local sums from zero:  loc_c = P loc_{c-1} + E_c  inside the block       -> ws[c]
//   phase 2  (one wave)           block starts:          G_{g+1} = PB G_g + loc_{last of g},  G_0 = 0       -> gs[g]
//   phase 3  (a wave per block)   V = G_g;  per chunk:   V = P V,  T_c = V + loc_c,  carry[c + 1] = T_c
// with P = Phi^L and PB = Phi^(L CARRY_BLOCK) (m x m, row-major; m = comps x nsec, comps = 2 values per section -- (d1, delta) --
// for a DF2 chain, 4 -- (y1, delta, x1, x2) -- when a section is in FormeDirecte1; host).  2 x CARRY_BLOCK + C / CARRY_BLOCK
// dependent steps of ~100 cycles.
constexpr int CARRY_BLOCK = 64;

template <int MP>
__global__ __launch_bounds__(64) void sos_carry_kernel(float *__restrict__ carry, const double *__restrict__ Pg, double *__restrict__ ws,
                                                       double *__restrict__ gs, int C, int nsec, int nch, int comps, int phase)
{
  extern __shared__ double cl[];                 // P (m x m), V (per), W (per), In (CARRY_BLOCK x per)
  const int m = comps * nsec, per = nch * m, lane = threadIdx.x, rec = nsec * 8;
  double *P = cl, *V = cl + m * m, *W = V + per, *In = W + per;
  const double *src = phase == 2 ? Pg + (size_t) m * m : Pg;
  for (int i = lane; i < m * m; i += 64) P[i] = src[i];
  auto slot = [&](int r) {     // r = ch * m + i  ->  float index inside a chunk's carry record
    const int ch = r / m, i = r - ch * m, sc = i / comps;
    return (sc * 2 + ch) * 4 + (i - sc * comps);
  };
  // the inputs of the block's steps come into LDS first (a load inside the recurrence costs its full latency per step:
  // 0.6 us per step measured, against ~50 ns)
  const int G = (C + CARRY_BLOCK - 1) / CARRY_BLOCK;
  const int g = blockIdx.x, c0 = g * CARRY_BLOCK, c1 = min(C, c0 + CARRY_BLOCK);
  const int nb = phase == 2 ? G - 1 : c1 - c0;
  for (int e = lane; e < nb * per; e += 64) {
    const int q = e / per, r = e - q * per;
    if (phase == 1) In[e] = (double) carry[(size_t) (c0 + q) * rec + slot(r)];
    else if (phase == 2) In[e] = ws[(size_t) (min(C, (q + 1) * CARRY_BLOCK) - 1) * per + r];
    else In[e] = ws[(size_t) c0 * per + e];
  }
  for (int r = lane; r < per; r += 64) V[r] = phase == 3 ? gs[(size_t) g * per + r] : 0.0;
  if (phase == 2)
    for (int r = lane; r < per; r += 64) gs[r] = 0.0;
  wave_sync();
  if (MP > 0) {
    // m <= MP <= 16 (up to 8 sections) and a row per lane: the lane's row of P and its component of the vector stay in
    // registers, the other components come by wave shuffles -- no LDS round trip and no barrier inside the recurrence.
    // (A lone wave issues an instruction every 4-8 cycles: the step is as long as its instruction count -- hence MP,
    // the compile-time bound of m; the rolled loop below takes ~1000 cycles per step.)
    const bool actif = lane < per;
    const int ch = actif ? lane / m : 0, i = actif ? lane - ch * m : 0, chm = ch * m;
    constexpr int MQ = MP > 0 ? MP : 1;
    double prow[MQ];
#pragma unroll
    for (int j = 0; j < MQ; j++) prow[j] = (actif && j < m) ? P[i * m + j] : 0.0;
    double v = actif ? V[lane] : 0.0;
    for (int q = 0; q < nb; q++) {
      const double inq = actif ? In[q * per + lane] : 0.0;
      // (prow is zero beyond m; the lanes beyond `per` hold zeros: a shuffle past the vector brings a finite value)
      double a0 = phase == 3 ? 0.0 : inq, a1 = 0.0;
#pragma unroll
      for (int j = 0; j < MQ; j += 2) {
        a0 = fma(prow[j], __shfl(v, (chm + j) & 63), a0);
        if (j + 1 < MQ) a1 = fma(prow[j + 1], __shfl(v, (chm + j + 1) & 63), a1);
      }
      v = a0 + a1;
      if (actif) In[q * per + lane] = phase == 3 ? v + inq : v;
    }
    wave_sync();
  } else {
    for (int q = 0; q < nb; q++) {
      // W = P V (+ the step's input); every lane handles rows lane, lane + 64
      for (int r = lane; r < per; r += 64) {
        const int ch = r / m, i = r - ch * m;
        double acc = phase == 3 ? 0.0 : In[q * per + r];
        for (int j = 0; j < m; j++) acc = fma(P[i * m + j], V[ch * m + j], acc);
        W[r] = acc;
      }
      wave_sync();
      // (results wait in LDS, over the inputs they replace: a global store inside the loop makes the next barrier wait
      // for its acknowledgement -- 0.5 us per step)
      for (int r = lane; r < per; r += 64) {
        const double w = W[r];
        V[r] = w;
        In[q * per + r] = phase == 3 ? w + In[q * per + r] : w;
      }
      wave_sync();
    }
  }
  for (int e = lane; e < nb * per; e += 64) {
    const int q = e / per, r = e - q * per;
    if (phase == 1) ws[(size_t) c0 * per + e] = In[e];
    else if (phase == 2) gs[(size_t) per + e] = In[e];
    else if (c0 + q + 1 < C) carry[(size_t) (c0 + q + 1) * rec + slot(r)] = (float) In[e];
  }
}

// Ragged end (the samples after the last whole sub-tile; a whole call of fewer than 2048 floats): ONE wave.
//  (1) its whole lanes (32 floats each) run as ONE pass of the block-parallel cascade over a partial sub-tile: the lanes past
//      the data run on zeros and the state carried on is taken at the last lane with data (a first version stepped through
//      groups of 256 floats with the narrow warm-up tables, up to 7 passes: 46 us for a 1000-sample complex block);
//  (2) the last < 32 floats go through the sections as a systolic pipeline: lane (section, channel) applies the reference
//      recurrence literally to sample t - section at step t and hands its output to the next section's lane by a shuffle,
//      the samples sitting in LDS -- m + nsec steps of a few tens of cycles.
// (The first version walked the samples section after section from global memory, one dependent load per sample: 0.9 us per
// sample of a 6-section chain -- 466 us for a 512-sample block, up to 1.8 ms added to ANY call whose length is not a multiple
// of 2048 floats.)  x must be 16-B aligned at float f0 (it is: f0 is a multiple of 2048 floats of an aligned buffer).
template <int NCH>
__global__ __launch_bounds__(64) void sos_tail_kernel(const float *__restrict__ x, float *__restrict__ y, const SosSection *__restrict__ sec,
                                                      int nsec, float gain, float *__restrict__ st, int64_t f0, int64_t f1, int64_t skip_f)
{
  __shared__ float sst[SOS_MAX_SEC * 8];
  __shared__ float buf[LANE_FLOATS];
  __shared__ __attribute__((aligned(16))) float lds[64 * LDS_LANE_PITCH];
  const int lane = threadIdx.x;
  const bool seeded = st[0] != 0.f;
  state_load(sst, st, sec, nsec, lane);
  wave_sync();
  int64_t f = f0;
  // (1) the whole lanes of the partial sub-tile (32 floats each) as ONE pass of the block-parallel cascade: the lanes past
  // the data run on zeros, the state carried on is the one of the last lane with data
  const int nl = (int) ((f1 - f0) / LANE_FLOATS);
  if (nl > 0) {
    const int nf = nl * LANE_FLOATS;
    float v[LANE_FLOATS];
#pragma unroll
    for (int i = 0; i < LANE_QUADS; i++) {
      const int p = 4 * (i * 64 + lane);
      const float4 q = p < nf ? *reinterpret_cast<const float4 *>(x + f + p) : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4 *>(&lds[(p / LANE_FLOATS) * LDS_LANE_PITCH + (p % LANE_FLOATS)]) = q;
    }
    wave_sync();
#pragma unroll
    for (int i = 0; i < LANE_QUADS; i++) {
      const float4 q = *reinterpret_cast<const float4 *>(&lds[lane * LDS_LANE_PITCH + 4 * i]);
      v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
    }
    wave_sync();
    sos_cascade<NCH, LANE_FLOATS, false>(v, sec, nsec, sst, lane, !seeded && f == 0, nl - 1);
#pragma unroll
    for (int i = 0; i < LANE_QUADS; i++)
      *reinterpret_cast<float4 *>(&lds[lane * LDS_LANE_PITCH + 4 * i]) =
          make_float4(v[4 * i] * gain, v[4 * i + 1] * gain, v[4 * i + 2] * gain, v[4 * i + 3] * gain);
    wave_sync();
#pragma unroll
    for (int i = 0; i < LANE_QUADS; i++) {
      const int p = 4 * (i * 64 + lane);
      if (p < nf && f + p >= skip_f) *reinterpret_cast<float4 *>(y + f + p) = *reinterpret_cast<const float4 *>(&lds[(p / LANE_FLOATS) * LDS_LANE_PITCH + (p % LANE_FLOATS)]);
    }
    wave_sync();
    f += nf;
  }
  // (2) systolic tail over the m = f1 - f < 32 floats left
  const int m = (int) (f1 - f), ms = m / NCH;
  if (ms > 0) {
    for (int i = lane; i < m; i += 64) buf[i] = x[f + i];
    wave_sync();
    const int sidx = lane / NCH, c = lane - sidx * NCH;
    const bool actif = sidx < nsec;
    const SosSection &k = sec[actif ? sidx : 0];
    const float b0 = k.b0, b1 = k.b1, b2 = k.b2, a1 = k.a1, a2 = k.a2;
    const bool df1 = k.df1 != 0.f, graine = k.seed != 0.f && !seeded && f == 0;    // first sample of the stream: seeded sections
    float *ss = &sst[(sidx * 2 + c) * 4];
    float d1 = 0.f, d2 = 0.f, x1 = 0.f, x2 = 0.f, out = 0.f;
    if (actif) { d1 = ss[0]; d2 = k.sg * (ss[0] - ss[1]); x1 = ss[2]; x2 = ss[3]; }
    for (int t = 0; t < ms + nsec - 1; t++) {
      const float amont = __shfl_up(out, NCH);             // what the previous section produced at the previous step
      const int i = t - sidx;
      if (actif && i >= 0 && i < ms) {
        const float xin = sidx == 0 ? buf[i * NCH + c] : amont;
        if (i == 0 && graine) d1 = d2 = x1 = x2 = xin;     // filtre-rt.cc:361-365
        float o;
        if (!df1) {
          const float d = fmaf(-a2, d2, fmaf(-a1, d1, xin));
          o = fmaf(b2, d2, fmaf(b1, d1, b0 * d));
          d2 = d1;
          d1 = d;
        } else {
          o = fmaf(-a2, d2, fmaf(-a1, d1, fmaf(b2, x2, fmaf(b1, x1, b0 * xin))));
          x2 = x1;
          x1 = xin;
          d2 = d1;
          d1 = o;
        }
        out = o;
        if (sidx == nsec - 1) buf[i * NCH + c] = o * gain;  // (index i <= t: section 0 has read it already)
      }
    }
    wave_sync();
    for (int i = lane; i < m; i += 64)
      if (f + i >= skip_f) y[f + i] = buf[i];
    if (actif) { ss[0] = d1; ss[1] = d1 - k.sg * d2; ss[2] = x1; ss[3] = x2; }
    wave_sync();
  }
  if (f1 > f0) state_store(st, sst, sec, nsec, lane);
}

}  // namespace tsdgpu

using namespace tsdgpu;

namespace tsdgpu {
int sos_create_ex(tsdgpu_sos **out, int data_type, const float *coefs_host, int nsec, float gain,
                  const float *rii1_host, int forme, int seeded);
}

struct tsdgpu_sos {
  int data_type = 0, nsec = 0, nch = 1;
  float gain = 1.f;
  SosSection *d_sec = nullptr;
  float *d_state[2] = {nullptr, nullptr};
  int cur = 0;
  bool capturable = false;      // tsdgpu_sos_set_capturable: the state is back in d_state[0] after every step
  int64_t halo = 0;             // W: samples after which the state transition is below 1e-9
  int64_t skip_f = 0;           // tsdgpu_sos_step_skip: floats at the start of the current call that are filtered but not stored
  DevBuf in_stage, out_stage;
  // exact carry of the state from chunk to chunk (long-memory filters, see sos_kernel)
  int comps = 2;                // state values per section and channel in the carry: 2 (DF2 chain) or 4 (a DF1 section: + its last two inputs)
  std::vector<double> phi;      // one-step zero-input transition of the whole cascade, m x m, m = comps x nsec
  std::vector<float> sg_host;   // per section: the sign of the scans' coordinates (SosSection::sg)
  DevBuf carry, scan_ws, scan_P;
  int64_t scan_L = 0;           // the chunk length (samples) scan_P was made for (< 0: no tables for that length)
};

namespace {

// one step of the whole cascade in double on a state vector (2 per section), input u
double cascade_step(const std::vector<SosSection> &sec, std::vector<double> &st, double u)
{
  double v = u;
  for (size_t s = 0; s < sec.size(); s++) {
    const double d1 = st[2 * s], d2 = st[2 * s + 1];
    const double d = v - (double) sec[s].a1 * d1 - (double) sec[s].a2 * d2;
    v = (double) sec[s].b0 * d + (double) sec[s].b1 * d1 + (double) sec[s].b2 * d2;
    st[2 * s + 1] = d1;
    st[2 * s] = d;
  }
  return v;
}

int64_t compute_halo(const std::vector<SosSection> &sec)
{
  const int m = 2 * (int) sec.size();
  if (m == 0) return 0;
  // Phi = one-step zero-input state transition, column by column
  std::vector<double> P((size_t) m * m, 0.0);
  for (int j = 0; j < m; j++) {
    std::vector<double> st((size_t) m, 0.0);
    st[j] = 1.0;
    cascade_step(sec, st, 0.0);
    for (int i = 0; i < m; i++) P[(size_t) i * m + j] = st[i];
  }
  int64_t W = 1;
  for (int it = 0; it < 40; it++) {
    double nrm = 0;
    for (int i = 0; i < m; i++) {
      double r = 0;
      for (int j = 0; j < m; j++) r += std::fabs(P[(size_t) i * m + j]);
      nrm = std::max(nrm, r);
    }
    if (!(nrm > 1e-9)) return W;
    if (!std::isfinite(nrm)) break;
    std::vector<double> Q((size_t) m * m, 0.0);
    for (int i = 0; i < m; i++)
      for (int k = 0; k < m; k++) {
        const double a = P[(size_t) i * m + k];
        if (a == 0) continue;
        for (int j = 0; j < m; j++) Q[(size_t) i * m + j] += a * P[(size_t) k * m + j];
      }
    P.swap(Q);
    W *= 2;
  }
  return -1;   // does not decay (unstable or marginal filter)
}

// Phi: the one-step zero-input transition of the cascade in the coordinates of the waves' running state image, m x m
// row-major, in double; m = comps per section: (d1 | y1, delta = that - sg * (d2 | y2)) and, with comps = 4, (x1, x2), the last
// two inputs of a FormeDirecte1 section (zero rows and columns for a DF2 section of a mixed chain)
std::vector<double> cascade_transition(const std::vector<SosSection> &sec, int comps)
{
  const int ns = (int) sec.size(), m = comps * ns;
  std::vector<double> P((size_t) m * m, 0.0);
  for (int j = 0; j < m; j++) {
    // unit vector j of the image -> natural states (a, b, x1, x2) per section
    std::vector<double> a((size_t) ns, 0.0), b((size_t) ns, 0.0), x1((size_t) ns, 0.0), x2((size_t) ns, 0.0);
    const int sj = j / comps, cj = j - sj * comps;
    const double sgj = sec[sj].sg;
    if (cj == 0) { a[sj] = 1.0; b[sj] = sgj; }        // level 1, slope 0
    else if (cj == 1) b[sj] = -sgj;                   // slope 1
    else if (cj == 2) x1[sj] = 1.0;
    else x2[sj] = 1.0;
    // one step of the cascade with a zero input
    double v = 0.0;
    for (int q = 0; q < ns; q++) {
      const SosSection &k = sec[q];
      if (k.df1 == 0.f) {
        const double d = v - (double) k.a1 * a[q] - (double) k.a2 * b[q];
        const double o = (double) k.b0 * d + (double) k.b1 * a[q] + (double) k.b2 * b[q];
        b[q] = a[q];
        a[q] = d;
        v = o;
      } else {
        const double fir = (double) k.b0 * v + (double) k.b1 * x1[q] + (double) k.b2 * x2[q];
        const double o = fir - (double) k.a1 * a[q] - (double) k.a2 * b[q];
        x2[q] = x1[q];
        x1[q] = v;
        b[q] = a[q];
        a[q] = o;
        v = o;
      }
    }
    for (int q = 0; q < ns; q++) {
      P[(size_t) (q * comps) * m + j] = a[q];
      P[(size_t) (q * comps + 1) * m + j] = a[q] - (double) sec[q].sg * b[q];
      if (comps == 4) {
        P[(size_t) (q * comps + 2) * m + j] = sec[q].df1 != 0.f ? x1[q] : 0.0;
        P[(size_t) (q * comps + 3) * m + j] = sec[q].df1 != 0.f ? x2[q] : 0.0;
      }
    }
  }
  return P;
}
void matmul(const std::vector<double> &A, const std::vector<double> &B, std::vector<double> &C, int m)
{
  C.assign((size_t) m * m, 0.0);
  for (int i = 0; i < m; i++)
    for (int k = 0; k < m; k++) {
      const double a = A[(size_t) i * m + k];
      if (a == 0) continue;
      for (int j = 0; j < m; j++) C[(size_t) i * m + j] += a * B[(size_t) k * m + j];
    }
}
// Phi^e by squaring
void matpow(const std::vector<double> &phi, int m, int64_t e, std::vector<double> &R)
{
  std::vector<double> B = phi, T;
  R.assign((size_t) m * m, 0.0);
  for (int i = 0; i < m; i++) R[(size_t) i * m + i] = 1.0;
  for (; e > 0; e >>= 1) {
    if (e & 1) { matmul(R, B, T, m); R.swap(T); }
    if (e > 1) { matmul(B, B, T, m); B.swap(T); }
  }
}
// P = Phi^L, PB = Phi^(L CARRY_BLOCK); false when a power leaves the float range (a filter that blows up)
bool carry_tables(const std::vector<double> &phi, int m, int64_t L, std::vector<double> &out)
{
  std::vector<double> P, PB;
  matpow(phi, m, L, P);
  matpow(P, m, CARRY_BLOCK, PB);
  out = P;
  out.insert(out.end(), PB.begin(), PB.end());
  for (double v : out)
    if (!std::isfinite(v) || std::fabs(v) > 1e30) return false;
  return true;
}

void fill_tables_for(SosSection &k, int L, int NF, float *c1, float *c2, float (*Aout)[4]);
int scan_levels(const float (*A)[4]);
void fill_tables(SosSection &k, int L)
{
  k.sg = k.a1 <= 0.f ? 1.f : -1.f;          // poles' real part = -a1 / 2
  fill_tables_for(k, L, LANE_FLOATS, k.c1, k.c2, k.A);
  // narrow warm-up steps: NARROW_FLOATS floats per lane = L * NARROW_FLOATS / LANE_FLOATS samples per channel
  fill_tables_for(k, L * NARROW_FLOATS / LANE_FLOATS, NARROW_FLOATS, k.c1n, k.c2n, k.An);
  k.nlev = scan_levels(k.A);
  k.nlevn = scan_levels(k.An);
}
void fill_tables_for(SosSection &k, int L, int NF, float *c1o, float *c2o, float (*Aout)[4])
{
  const double a1 = k.a1, a2 = k.a2;
  // DF1 carries (y1, y2): the correction is the all-pole zero-input response itself
  const double b0 = k.df1 != 0.f ? 1.0 : k.b0, b1 = k.df1 != 0.f ? 0.0 : k.b1, b2 = k.df1 != 0.f ? 0.0 : k.b2;
  // zero-input responses from unit start states (d1, delta) = (1,0) and (0,1), i.e. (d1, d2) = (1, sg) and (0, -sg)
  const double sg = k.sg;
  for (int which = 0; which < 2; which++) {
    double d1 = which == 0 ? 1.0 : 0.0, d2 = which == 0 ? sg : -sg;
    for (int i = 0; i < NF; i++) {
      double o = 0;
      if (i < L) {
        const double d = -a1 * d1 - a2 * d2;
        o = b0 * d + b1 * d1 + b2 * d2;
        d2 = d1;
        d1 = d;
      }
      (which == 0 ? c1o : c2o)[i] = (float) o;
    }
  }
  // M^L by L-fold application, then repeated squaring for the scan
  double A[4] = {1, 0, 0, 1};
  const double M[4] = {-a1, -a2, 1, 0};
  for (int i = 0; i < L; i++) {
    const double t[4] = {M[0] * A[0] + M[1] * A[2], M[0] * A[1] + M[1] * A[3], M[2] * A[0] + M[3] * A[2],
                         M[2] * A[1] + M[3] * A[3]};
    for (int j = 0; j < 4; j++) A[j] = t[j];
  }
  for (int kk = 0; kk < 6; kk++) {
    // T A T^-1 with T = [[1,0],[1,-sg]], T^-1 = [[1,0],[sg,-sg]]
    const double B[4] = {A[0] + sg * A[1], -sg * A[1], A[2] + sg * A[3], -sg * A[3]};        // A T^-1
    const double Tm[4] = {B[0], B[1], B[0] - sg * B[2], B[1] - sg * B[3]};                    // T (A T^-1)
    for (int j = 0; j < 4; j++) Aout[kk][j] = (float) Tm[j];
    const double t[4] = {A[0] * A[0] + A[1] * A[2], A[0] * A[1] + A[1] * A[3], A[2] * A[0] + A[3] * A[2],
                         A[2] * A[1] + A[3] * A[3]};
    for (int j = 0; j < 4; j++) A[j] = t[j];
  }
}
// scan levels that matter for this table (see SosSection::nlev): the first K whose power is below 1e-9 in the row-sum norm
int scan_levels(const float (*A)[4])
{
  static const bool full = dev_switch("SOS_FULL_SCAN") != nullptr;      // A/B and test switch: all six levels
  if (full) return 6;
  for (int K = 0; K < 6; K++) {
    const double n0 = std::fabs((double) A[K][0]) + std::fabs((double) A[K][1]), n1 = std::fabs((double) A[K][2]) + std::fabs((double) A[K][3]);
    if (std::isfinite(n0) && std::isfinite(n1) && std::max(n0, n1) <= 1e-9) return K;
  }
  return 6;
}

}  // namespace

extern "C" {

int tsdgpu_sos_create(tsdgpu_sos **out, int data_type, const float *coefs_host, int nsec, float gain,
                      const float *rii1_host, int forme)
{
  return tsdgpu::sos_create_ex(out, data_type, coefs_host, nsec, gain, rii1_host, forme, 1);
}

}  // extern "C"

// `seeded` = 0 builds sections that start from zero memory (what FiltreRII does) instead of
// SOIS' first-sample seed; used by the low-order fast path of tsdgpu_rii.
int tsdgpu::sos_create_ex(tsdgpu_sos **out, int data_type, const float *coefs_host, int nsec, float gain,
                          const float *rii1_host, int forme, int seeded)
{
  TSD_CHECK(out != nullptr, "sos_create: out is NULL");
  *out = nullptr;
  TSD_CHECK(data_type == TSDGPU_F32 || data_type == TSDGPU_C64, "sos_create: bad data_type %d", data_type);
  TSD_CHECK(nsec >= 0 && nsec + (rii1_host ? 1 : 0) <= SOS_MAX_SEC, "sos_create: %d sections unsupported (max %d)",
            nsec, SOS_MAX_SEC);
  TSD_CHECK(nsec == 0 || coefs_host != nullptr, "sos_create: NULL coefficients");
  TSD_CHECK(forme == 1 || forme == 2, "sos_create: forme must be 1 (FormeDirecte1) or 2 (FormeDirecte2), got %d", forme);

  tsdgpu_sos *s = new tsdgpu_sos();
  s->data_type = data_type;
  s->nch = data_type == TSDGPU_C64 ? 2 : 1;
  const int L = LANE_FLOATS / s->nch;
  std::vector<SosSection> sec;
  for (int i = 0; i < nsec; i++) {
    SosSection k{};
    k.b0 = coefs_host[5 * i]; k.b1 = coefs_host[5 * i + 1]; k.b2 = coefs_host[5 * i + 2];
    k.a1 = coefs_host[5 * i + 3]; k.a2 = coefs_host[5 * i + 4];
    k.seed = seeded ? 1.f : 0.f;
    k.df1 = forme == 1 ? 1.f : 0.f;
    fill_tables(k, L);
    sec.push_back(k);
  }
  if (rii1_host) {
    // RIIFoS (filtre-rt.cc:407-437): y = -a1*y1 + b0*x0 + b1*x1 from zero memory == the
    // biquad (b0,b1,0 ; a1,0) from zero state; it carries the gain, so `gain` is not applied
    SosSection k{};
    k.b0 = rii1_host[0]; k.b1 = rii1_host[1]; k.b2 = 0.f; k.a1 = rii1_host[2]; k.a2 = 0.f;
    k.seed = 0.f;
    fill_tables(k, L);
    sec.push_back(k);
    s->gain = 1.f;
  } else {
    s->gain = gain;
  }
  if (sec.empty()) {
    // order-0 chain: y = x * gain, expressed as an identity section with zero state
    SosSection k{};
    k.b0 = 1.f;
    fill_tables(k, L);
    sec.push_back(k);
  }
  s->nsec = (int) sec.size();
  s->halo = compute_halo(sec);
  s->comps = 2;
  for (const SosSection &k : sec)
    if (k.df1 != 0.f) s->comps = 4;
  s->phi = cascade_transition(sec, s->comps);
  for (const SosSection &k : sec) s->sg_host.push_back(k.sg);

  int rc = TSDGPU_OK;
  do {
    // ONE allocation (the section tables, then the two zeroed state buffers) and ONE upload of its host image
    const size_t sb = (std::max<size_t>(1, sec.size()) * sizeof(SosSection) + 15) / 16 * 16, stb = ((size_t) STATE_FLOATS * 4 + 15) / 16 * 16;
    std::vector<char> image(sb + 2 * stb, 0);
    if (!sec.empty()) std::memcpy(image.data(), sec.data(), sec.size() * sizeof(SosSection));
    if (hipMalloc((void **) &s->d_sec, image.size()) != hipSuccess) {
      rc = set_err(TSDGPU_ERR_HIP, "sos_create: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
      break;
    }
    if (hipMemcpy(s->d_sec, image.data(), image.size(), hipMemcpyHostToDevice) != hipSuccess) {
      rc = set_err(TSDGPU_ERR_HIP, "sos_create: upload failed: %s", hipGetErrorString(hipGetLastError()));
      break;
    }
    s->d_state[0] = reinterpret_cast<float *>(reinterpret_cast<char *>(s->d_sec) + sb);
    s->d_state[1] = reinterpret_cast<float *>(reinterpret_cast<char *>(s->d_sec) + sb + stb);
  } while (0);
  if (rc) {
    tsdgpu_sos_destroy(s);
    return rc;
  }
  *out = s;
  return TSDGPU_OK;
}

extern "C" {

int tsdgpu_sos_step(tsdgpu_sos *s, const void *x, void *y, int64_t n, void *stream)
{
  TSD_CHECK(s != nullptr, "sos_step: NULL handle");
  TSD_CHECK(n >= 0, "sos_step: negative length");
  if (n == 0) return TSDGPU_OK;
  TSD_CHECK(x != nullptr && y != nullptr, "sos_step: NULL buffer");
  hipStream_t st = (hipStream_t) stream;
  const size_t bytes = (size_t) n * dtype_size(s->data_type);
  if (bytes >= PIPE_MIN_BYTES && host_pipe_enabled() && !is_device_ptr(x) && !is_device_ptr(y)) {
    // large host vectors: chunked H2D / kernel / D2H pipeline (the section states carry from chunk to chunk)
    return pipelined_host_step(x, y, n, dtype_size(s->data_type), st,
                               [s](const void *cx, void *cy, int64_t cnt, hipStream_t q) { return tsdgpu_sos_step(s, cx, cy, cnt, q); });
  }
  const void *dx = nullptr;
  void *dy = nullptr;
  bool staged = false;
  int rc = stage_in(x, bytes, s->in_stage, st, &dx);
  if (rc) return rc;
  rc = stage_out(y, bytes, s->out_stage, &dy, &staged);
  if (rc) return rc;
  if (dx == dy) {
    // in place (filtre-rt.cc:354-355): chunks re-read their predecessors' inputs for the warm-up
    rc = s->in_stage.reserve(bytes);
    if (rc) return rc;
    TSD_HIP(hipMemcpyAsync(s->in_stage.p, dx, bytes, hipMemcpyDeviceToDevice, st));
    dx = s->in_stage.p;
  }
  // the wave kernel moves 16 B per lane: a device pointer that is not 16-B aligned (a view into
  // a larger vector) is bounced through an aligned buffer -- one extra copy, instead of leaving
  // the whole vector to the sequential tail kernel
  if (((uintptr_t) dx & 15) != 0) {
    rc = s->in_stage.reserve(bytes);
    if (rc) return rc;
    TSD_HIP(hipMemcpyAsync(s->in_stage.p, dx, bytes, hipMemcpyDeviceToDevice, st));
    dx = s->in_stage.p;
  }
  void *dy_user = nullptr;
  if (((uintptr_t) dy & 15) != 0) {
    rc = s->out_stage.reserve(bytes);
    if (rc) return rc;
    dy_user = dy;
    dy = s->out_stage.p;
  }
  const int nch = s->nch;
  const int64_t nfl = n * nch;                              // floats
  const int64_t n_sub = nfl / SUB_FLOATS;                   // whole sub-tiles go to the wave kernel
  float *st_in = s->d_state[s->cur], *st_out = s->d_state[s->cur ^ 1];
  if (n_sub > 0) {
    const int64_t sub_samples = SUB_FLOATS / nch;
    int64_t warm_sub, warm_nar = 0, spc;
    if (s->halo < 0) {
      warm_sub = 0;
      spc = n_sub;                                          // no decay: one sequential chunk
    } else {
      // warm-up of `halo` samples: whole sub-tiles for the bulk of a long halo, then narrow steps of
      // 64 * NARROW_FLOATS floats (a fifth of a sub-tile's work each) for the rest
      const int64_t nar_samples = 64 * NARROW_FLOATS / nch;
      warm_sub = s->halo / sub_samples;
      warm_nar = cdiv(s->halo - warm_sub * sub_samples, nar_samples);
      if (warm_nar * nar_samples >= sub_samples) { warm_sub++; warm_nar = 0; }
      // chunk length: enough chunks to fill the chip, but the warm-up never more than a quarter of the
      // chunk's work (a narrow step counts as a fifth of a sub-tile)
      // (with the quarter rule above: 8192 chunks of four sub-tiles at 2^26 samples, 0.1267 ms against 0.1317 with 4096 chunks of
      // eight -- measured twice, interleaved)
      constexpr int64_t TARGET = 16384;                 // (flat from 8192 up: profiles/EXPERIMENTS.md, round 4)
      const int64_t warm_cost = warm_sub + cdiv(warm_nar, 5);
      spc = std::max<int64_t>({2, SOS_WARM_FACTOR * warm_cost, warm_sub + 1, n_sub / TARGET});
      // ... unless the call is short of filling the chip anyway: then the shortest chunks finish first (a wave alone takes
      // ~4 us per sub-tile: 4097 complex samples in one chunk of 4 sub-tiles cost 38 us)
      const int64_t spc_min = std::max<int64_t>(1, warm_sub + 1);
      if (cdiv(n_sub, spc_min) <= 2048) spc = spc_min;
    }
    int64_t nchunks = cdiv(n_sub, spc);
    TSD_CHECK(nchunks <= 0x7fffffff, "sos_step: too many chunks");
    // a memory that is long against the call leaves few chunks (one, without decay): carry the state exactly from chunk
    // to chunk instead -- two passes over x and a scan of the chunks' end states (see sos_kernel)
    static const bool no_exact = dev_switch("SOS_NO_EXACT_CARRY") != nullptr;
    static const int64_t EX_TARGET = 4096;
    const int64_t spc_ex = std::max<int64_t>(1, n_sub / EX_TARGET), nch_ex = cdiv(n_sub, spc_ex);
    const size_t sos_dyn_lds = 0;
    bool exact = false;
    if (!no_exact && !s->capturable && nch_ex >= 8) {
      // which is cheaper (microseconds, rough): a wave alone takes tw per sub-tile (latency-bound: 1.5 + 0.4 per section,
      // measured on the sequential chunk), the chip as a whole moves a sub-tile's 16 KB at ~4 TB/s; the carry kernels
      // cost three short launches of 64 dependent steps (12 us each at m <= 4, 40 at m = 16, more on the generic loop)
      const int m = s->comps * s->nsec, per = nch * m;
      const double tw = 1.5 + 0.4 * s->nsec, bw = (double) n_sub * (SUB_FLOATS * 8.0) / 4e6;
      const double warm_cost = s->halo < 0 ? 0.0 : (double) warm_sub + (double) warm_nar / 5.0;
      const double t_norm = std::max((spc + warm_cost) * tw, bw * (1.0 + warm_cost / (double) spc));
      const double carry_us = (per > 64 || m > 16) ? 100.0 + 3.0 * per : 9.0 + 7.0 * (m <= 2 ? 2 : m <= 4 ? 4 : m <= 8 ? 8 : 16);
      const double t_ex = 2.0 * std::max(spc_ex * tw, 0.75 * bw) + carry_us;
      exact = t_ex < 0.8 * t_norm;
    }
    if (exact) {
      const int m = s->comps * s->nsec;
      const int64_t L = spc_ex * sub_samples;
      if (s->scan_L != L && s->scan_L != -L) {
        std::vector<double> P;
        if (carry_tables(s->phi, m, L, P)) {
          if ((rc = s->scan_P.reserve(P.size() * sizeof(double)))) return rc;
          TSD_HIP(hipMemcpyAsync(s->scan_P.p, P.data(), P.size() * sizeof(double), hipMemcpyHostToDevice, st));
          TSD_HIP(hipStreamSynchronize(st));             // (P dies with this scope)
          s->scan_L = L;
        } else {
          s->scan_L = -L;                                // the powers leave the float range: the sequential chunk it is
        }
      }
      if (s->scan_L == -L) exact = false;
    }
    if (exact) {
      const int m = s->comps * s->nsec, per = nch * m, G = (int) cdiv(nch_ex, CARRY_BLOCK);
      const size_t rec = (size_t) s->nsec * 8, img = (size_t) nch_ex * per;
      if ((rc = s->carry.reserve((size_t) nch_ex * rec * sizeof(float)))) return rc;
      if ((rc = s->scan_ws.reserve((img + (size_t) G * per) * sizeof(double)))) return rc;
      float *carry = s->carry.as<float>();
      double *ws = s->scan_ws.as<double>(), *gs = ws + img;
      const size_t cl = ((size_t) m * m + 2 * per + (size_t) CARRY_BLOCK * per) * sizeof(double);
#define SOS_LAUNCH(NCH, MODE)                                                                                              \
  hipLaunchKernelGGL((sos_kernel<NCH, MODE>), dim3((unsigned) nch_ex), dim3(64), sos_dyn_lds, st, (const float *) dx, (float *) dy, s->d_sec, \
                     s->nsec, s->gain, st_in, st_out, n_sub, (int) spc_ex, 0, 0, carry, s->skip_f)
      if (nch == 1) SOS_LAUNCH(1, 1); else SOS_LAUNCH(2, 1);
      const int mp = per > 64 || m > 16 ? 0 : m <= 2 ? 2 : m <= 4 ? 4 : m <= 8 ? 8 : 16;
#define CARRY_LAUNCH(MP)                                                                                                   \
  do {                                                                                                                     \
    (void) hipFuncSetAttribute((const void *) sos_carry_kernel<MP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    for (int phase = 1; phase <= 3; phase++)                                                                               \
      hipLaunchKernelGGL(sos_carry_kernel<MP>, dim3(phase == 2 ? 1u : (unsigned) G), dim3(64), cl, st, carry, s->scan_P.as<double>(), \
                         ws, gs, (int) nch_ex, s->nsec, nch, s->comps, phase);                                                     \
  } while (0)
      if (mp == 2) CARRY_LAUNCH(2); else if (mp == 4) CARRY_LAUNCH(4); else if (mp == 8) CARRY_LAUNCH(8);
      else if (mp == 16) CARRY_LAUNCH(16); else CARRY_LAUNCH(0);
#undef CARRY_LAUNCH
      if (nch == 1) SOS_LAUNCH(1, 2); else SOS_LAUNCH(2, 2);
#undef SOS_LAUNCH
      nchunks = nch_ex;
    } else if (nch == 1) {
      hipLaunchKernelGGL((sos_kernel<1, 0>), dim3((unsigned) nchunks), dim3(64), sos_dyn_lds, st, (const float *) dx, (float *) dy,
                         s->d_sec, s->nsec, s->gain, st_in, st_out, n_sub, (int) spc, (int) warm_sub, (int) warm_nar, (float *) nullptr, s->skip_f);
    } else {
      hipLaunchKernelGGL((sos_kernel<2, 0>), dim3((unsigned) nchunks), dim3(64), sos_dyn_lds, st, (const float *) dx, (float *) dy,
                         s->d_sec, s->nsec, s->gain, st_in, st_out, n_sub, (int) spc, (int) warm_sub, (int) warm_nar, (float *) nullptr, s->skip_f);
    }
    TSD_HIP(hipGetLastError());
    s->cur ^= 1;
  }
  const int64_t n0 = n_sub * (SUB_FLOATS / nch);
  if (n0 < n) {
    // the ragged end: narrow parallel steps + a systolic pipeline over the last < 256 floats, one wave (float indices)
    float *stc = s->d_state[s->cur];
    if (nch == 1)
      hipLaunchKernelGGL(sos_tail_kernel<1>, dim3(1), dim3(64), 0, st, (const float *) dx, (float *) dy, s->d_sec, s->nsec,
                         s->gain, stc, n_sub * SUB_FLOATS, nfl, s->skip_f);
    else
      hipLaunchKernelGGL(sos_tail_kernel<2>, dim3(1), dim3(64), 0, st, (const float *) dx, (float *) dy, s->d_sec, s->nsec,
                         s->gain, stc, n_sub * SUB_FLOATS, nfl, s->skip_f);
    TSD_HIP(hipGetLastError());
  }
  if (dy_user) {
    const size_t sk = (size_t) s->skip_f * sizeof(float);      // (tsdgpu_sos_step_skip: nothing before the first stored float)
    TSD_HIP(hipMemcpyAsync((char *) dy_user + sk, (const char *) dy + sk, bytes - sk, hipMemcpyDeviceToDevice, st));
    dy = dy_user;
  }
  if (s->capturable && s->cur == 1) {
    // same launch arguments at every step (see tsdgpu_fir_set_capturable): the state goes back to buffer 0
    if ((rc = device_copy_small(s->d_state[0], s->d_state[1], STATE_FLOATS * 4, st))) return rc;
    s->cur = 0;
  }
  return finish_out(y, bytes, dy, staged, st);
}

int tsdgpu_sos_step_skip(tsdgpu_sos *s, const void *x, void *y, int64_t n, int64_t skip, void *stream)
{
  TSD_CHECK(s != nullptr, "sos_step_skip: NULL handle");
  TSD_CHECK(skip >= 0 && skip <= n, "sos_step_skip: skip %lld outside [0, n = %lld]", (long long) skip, (long long) n);
  TSD_CHECK(n == 0 || (x != nullptr && y != nullptr && x != y && is_device_ptr(x) && is_device_ptr(y)), "sos_step_skip: distinct device buffers expected");
  const int64_t sf = skip * s->nch;
  if (sf % 4 != 0)
    return set_err(TSDGPU_ERR_UNSUPPORTED, "sos_step_skip: skip * channels = %lld is not a multiple of 4 floats", (long long) sf);
  s->skip_f = sf;
  const int rc = tsdgpu_sos_step(s, x, y, n, stream);
  s->skip_f = 0;
  return rc;
}

int tsdgpu_sos_set_capturable(tsdgpu_sos *s, int on)
{
  TSD_CHECK(s != nullptr, "sos_set_capturable: NULL handle");
  s->capturable = on != 0;
  if (s->capturable && s->cur == 1) {
    TSD_HIP(hipMemcpy(s->d_state[0], s->d_state[1], STATE_FLOATS * 4, hipMemcpyDeviceToDevice));
    TSD_HIP(hipStreamSynchronize(nullptr));
    s->cur = 0;
  }
  return TSDGPU_OK;
}

int tsdgpu_sos_reset(tsdgpu_sos *s)
{
  TSD_CHECK(s != nullptr, "sos_reset: NULL handle");
  // (a memset of device memory may return before it has run: wait, so that a step enqueued on ANY
  // stream afterwards -- non-blocking ones are not ordered with the null stream -- sees the zeros)
  TSD_HIP(hipMemset(s->d_state[s->cur], 0, STATE_FLOATS * 4));
  TSD_HIP(hipStreamSynchronize(nullptr));
  return TSDGPU_OK;
}
int tsdgpu_sos_reset_on(tsdgpu_sos *s, void *stream)
{
  TSD_CHECK(s != nullptr, "sos_reset: NULL handle");
  TSD_HIP(hipMemsetAsync(s->d_state[s->cur], 0, STATE_FLOATS * 4, (hipStream_t) stream));
  return TSDGPU_OK;
}

int64_t tsdgpu_sos_halo(const tsdgpu_sos *s) { return s ? s->halo : -1; }

}  // extern "C"

// ---- the stream state as a host vector (internal: the exact sharding of long-memory cascades, sharded.hip) ------------
// layout: [0] = "first sample seen" flag, then per (section, channel) (d1, d2, x1, x2) -- what the kernels publish
namespace tsdgpu {
int sos_state_floats() { return STATE_FLOATS; }
int sos_state_get(tsdgpu_sos *s, float *host, hipStream_t st)
{
  TSD_HIP(hipMemcpyAsync(host, s->d_state[s->cur], STATE_FLOATS * sizeof(float), hipMemcpyDeviceToHost, st));
  TSD_HIP(hipStreamSynchronize(st));
  return TSDGPU_OK;
}
int sos_state_set(tsdgpu_sos *s, const float *host, hipStream_t st)
{
  TSD_HIP(hipMemcpyAsync(s->d_state[s->cur], host, STATE_FLOATS * sizeof(float), hipMemcpyHostToDevice, st));
  TSD_HIP(hipStreamSynchronize(st));            // (`host` may die with the caller's scope)
  return TSDGPU_OK;
}
// out = Phi^L in + add: the state `in` after L samples of zero input (Phi^L in the scans' coordinates, double) plus the end
// state `add` of a zero-state run over those L samples (null: none); the flag of `out` is set when either has it.
// false when the powers leave the float range
bool sos_state_propagate(const tsdgpu_sos *s, int64_t L, const float *in, const float *add, float *out)
{
  const int ns = s->nsec, comps = s->comps, m = comps * ns;
  for (int i = 0; i < STATE_FLOATS; i++) out[i] = in[i];
  if (add && add[0] != 0.f) out[0] = 1.f;
  if (m == 0) return true;
  std::vector<double> P;
  matpow(s->phi, m, std::max<int64_t>(L, 0), P);
  for (double v : P)
    if (!std::isfinite(v) || std::fabs(v) > 1e30) return false;
  for (int ch = 0; ch < s->nch; ch++) {
    std::vector<double> v((size_t) m, 0.0), w((size_t) m, 0.0);
    for (int q = 0; q < ns; q++) {
      const float *r = in + state_index(q, ch);
      const double sg = s->sg_host[(size_t) q];
      v[(size_t) q * comps] = r[0];
      v[(size_t) q * comps + 1] = (double) r[0] - sg * (double) r[1];
      if (comps == 4) { v[(size_t) q * comps + 2] = r[2]; v[(size_t) q * comps + 3] = r[3]; }
    }
    for (int i = 0; i < m; i++) {
      double a = 0;
      for (int j = 0; j < m; j++) a += P[(size_t) i * m + j] * v[j];
      w[i] = a;
    }
    if (add) {
      for (int q = 0; q < ns; q++) {
        const float *r = add + state_index(q, ch);
        const double sg = s->sg_host[(size_t) q];
        w[(size_t) q * comps] += r[0];
        w[(size_t) q * comps + 1] += (double) r[0] - sg * (double) r[1];
        if (comps == 4) { w[(size_t) q * comps + 2] += r[2]; w[(size_t) q * comps + 3] += r[3]; }
      }
    }
    for (int q = 0; q < ns; q++) {
      float *r = out + state_index(q, ch);
      const double sg = s->sg_host[(size_t) q];
      r[0] = (float) w[(size_t) q * comps];
      r[1] = (float) (sg * (w[(size_t) q * comps] - w[(size_t) q * comps + 1]));
      if (comps == 4) { r[2] = (float) w[(size_t) q * comps + 2]; r[3] = (float) w[(size_t) q * comps + 3]; }
    }
  }
  return true;
}
}  // namespace tsdgpu

extern "C" {

int tsdgpu_sos_state_floats(void) { return tsdgpu::sos_state_floats(); }
int tsdgpu_sos_get_state(tsdgpu_sos *s, float *state_host, void *stream)
{
  TSD_CHECK(s != nullptr && state_host != nullptr, "sos_get_state: NULL argument");
  return tsdgpu::sos_state_get(s, state_host, (hipStream_t) stream);
}
int tsdgpu_sos_set_state(tsdgpu_sos *s, const float *state_host, void *stream)
{
  TSD_CHECK(s != nullptr && state_host != nullptr, "sos_set_state: NULL argument");
  return tsdgpu::sos_state_set(s, state_host, (hipStream_t) stream);
}
int tsdgpu_sos_propagate_state(const tsdgpu_sos *s, int64_t n_samples, const float *state_in, const float *end_state, float *state_out)
{
  TSD_CHECK(s != nullptr && state_in != nullptr && state_out != nullptr && n_samples >= 0, "sos_propagate_state: bad argument");
  if (!tsdgpu::sos_state_propagate(s, n_samples, state_in, end_state, state_out))
    return set_err(TSDGPU_ERR_UNSUPPORTED, "sos_propagate_state: the transition over %lld samples leaves the float range", (long long) n_samples);
  return TSDGPU_OK;
}


int tsdgpu_sos_destroy(tsdgpu_sos *s)
{
  if (!s) return TSDGPU_OK;
  if (s->d_sec) (void) hipFree(s->d_sec);           // (the state buffers live in the same allocation)
  s->in_stage.release();
  s->out_stage.release();
  s->carry.release();
  s->scan_ws.release();
  s->scan_P.release();
  delete s;
  return TSDGPU_OK;
}

}  // extern "C"
