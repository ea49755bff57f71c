// fft1024_wave.hpp -- 1024-point complex FFT held by ONE wave64 (16 points per lane).
//
// 1024 = 16 x 16 x 4, decimation in frequency for the forward direction and the exact
// mirror (decimation in time) for the inverse, so that forward followed by inverse needs
// no reordering at all: the spectrum lives in a "register order" that only this file
// knows (freq_index()).  Used by the overlap-save FIR (ols.hip) and as the row/column
// transform of the large four-step FFT (fft.hip).
//
//   stage A  lane = n2 (0..63), reg = n1 (0..15): time index t = 64*n1 + n2.
//            radix-16 over n1 -> k1, times W_1024^(n2*k1)
//   xchg 1   through LDS: element (k1, n2 = 4*m1 + m2)  ->  lane (k1, m2), reg m1
//   stage B  radix-16 over m1 -> j1, times W_64^(m2*j1)
//   xchg 2   element (k1, j1 = 4*j1hi + j1lo, m2) -> lane (k1, j1lo), reg (j1hi, m2)
//   stage C  radix-4 over m2 -> j2.   frequency k = k1 + 16*j1 + 256*j2.
//
// LDS image: 16 rows of 68 complex (64 + 4 pad) = 8704 B per wave; with the address maps
// below every ds_write_b64 / ds_read_b64 of both exchanges, in both directions, is
// bank-conflict free (rows shift by 4 slots, m2 planes by 17).
//
// Everything is plain C++ on float pairs: the same code is compiled by g++ for the
// CPU-side structural test (tests/cpu/test_fft1024_wave.cc emulates the 64 lanes).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TSD_HD __host__ __device__ __forceinline__
namespace tsdgpu { using cpx = float2; }
#else
#include <cmath>
#define TSD_HD inline
namespace tsdgpu { struct cpx { float x, y; }; }
#endif

namespace tsdgpu {
namespace w1024 {

constexpr int LDS_ROW = 68;                 // complex elements per k1 row
constexpr int LDS_ELEMS = 16 * LDS_ROW;     // 1088 complex = 8704 bytes per wave

TSD_HD cpx mk(float a, float b) { cpx r; r.x = a; r.y = b; return r; }
TSD_HD cpx cadd(cpx a, cpx b) { return mk(a.x + b.x, a.y + b.y); }
TSD_HD cpx csub(cpx a, cpx b) { return mk(a.x - b.x, a.y - b.y); }
TSD_HD cpx cmul(cpx a, cpx b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
TSD_HD cpx cmulc(cpx a, cpx b) { return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)
// multiply by -i (forward) / +i (inverse)
template <bool INV> TSD_HD cpx rot90(cpx a) { return INV ? mk(-a.y, a.x) : mk(a.y, -a.x); }
// a +- rot90(t) (one instruction each in the packed flavour below)
template <bool INV> TSD_HD cpx caddrot(cpx a, cpx t) { return cadd(a, rot90<INV>(t)); }
template <bool INV> TSD_HD cpx csubrot(cpx a, cpx t) { return csub(a, rot90<INV>(t)); }
template <typename C> struct Make;
template <> struct Make<cpx> { static TSD_HD cpx of(float a, float b) { return mk(a, b); } };

#if defined(__HIPCC__)
// Packed flavour: a complex number is a 64-bit VGPR pair and every primitive is ONE or TWO
// VOP3P instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two fp32 lanes per issue
// slot -- the only way to the full fp32 VALU rate of gfx950).  The swaps and sign flips of
// complex arithmetic ride on the op_sel / neg modifiers (op_sel[i]: half of source i read by
// the LOW result lane, op_sel_hi[i]: by the HIGH lane), which hipcc does not form by itself
// (it builds (-w.y, w.y) with two extra moves), hence the inline assembly.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f cadd(v2f a, v2f b) { return a + b; }
__device__ __forceinline__ v2f csub(v2f a, v2f b) { return a - b; }
__device__ __forceinline__ v2f cmul(v2f a, v2f w)
{
  v2f t, r;   // t = (a.x w.x, a.y w.x);  r = (t.x - a.y w.y, t.y + a.x w.y)
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
__device__ __forceinline__ v2f cmulc(v2f a, v2f w)   // a * conj(w)
{
  v2f t, r;   // r = (t.x + a.y w.y, t.y - a.x w.y)
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
template <bool INV> __device__ __forceinline__ v2f caddrot(v2f a, v2f t)
{
  v2f r;      // forward: (a.x + t.y, a.y - t.x);  inverse: (a.x - t.y, a.y + t.x)
  if (!INV) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(t));
  else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(t));
  return r;
}
template <bool INV> __device__ __forceinline__ v2f csubrot(v2f a, v2f t) { return caddrot<!INV>(a, t); }
template <bool INV> __device__ __forceinline__ v2f rot90(v2f a)
{
  v2f r;
  const v2f z = {0.f, 0.f};
  if (!INV) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(z), "v"(a));
  else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(z), "v"(a));
  return r;
}
template <> struct Make<v2f> { static __device__ __forceinline__ v2f of(float a, float b) { return (v2f){a, b}; } };
#endif

template <bool INV, typename C> TSD_HD C ctw(C a, C w) { return INV ? cmulc(a, w) : cmul(a, w); }

// 4-point DFT, natural order in and out. forward: W4 = -i.
template <bool INV, typename C> TSD_HD void dft4(C &a, C &b, C &c, C &d)
{
  const C t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
  a = cadd(t0, t2);
  b = caddrot<INV>(t1, t3);
  c = csub(t0, t2);
  d = csubrot<INV>(t1, t3);
}
// the same with input c still owing a factor W4 (-i forward, +i inverse)
template <bool INV, typename C> TSD_HD void dft4_crot(C &a, C &b, C &c, C &d)
{
  const C t0 = caddrot<INV>(a, c), t1 = csubrot<INV>(a, c), t2 = cadd(b, d), t3 = csub(b, d);
  a = cadd(t0, t2);
  b = caddrot<INV>(t1, t3);
  c = csub(t0, t2);
  d = csubrot<INV>(t1, t3);
}

// 16-point DFT in registers, natural order in and out.
template <bool INV, typename C> TSD_HD void dft16(C (&v)[16])
{
  constexpr float C1 = 0.92387953251128674f;   // cos(pi/8)
  constexpr float S1 = 0.38268343236508977f;   // sin(pi/8)
  constexpr float R2 = 0.70710678118654752f;   // sqrt(1/2)
  // stage 1: over n1 (stride 4) for each n2:  v[4*k1 + n2] <- sum_n1 x[4*n1 + n2] W4^(n1 k1)
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) dft4<INV>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
  // twiddles W16^(n2*k1), forward value (conjugated by ctw<INV> for the inverse)
  v[5] = ctw<INV>(v[5], Make<C>::of(C1, -S1));      // k1=1,n2=1: W16^1
  v[6] = ctw<INV>(v[6], Make<C>::of(R2, -R2));      // k1=1,n2=2: W16^2
  v[7] = ctw<INV>(v[7], Make<C>::of(S1, -C1));      // k1=1,n2=3: W16^3
  v[9] = ctw<INV>(v[9], Make<C>::of(R2, -R2));      // k1=2,n2=1: W16^2
  //  v[10]: k1=2,n2=2: W16^4 = -i, applied inside dft4_crot below
  v[11] = ctw<INV>(v[11], Make<C>::of(-R2, -R2));   // k1=2,n2=3: W16^6
  v[13] = ctw<INV>(v[13], Make<C>::of(S1, -C1));    // k1=3,n2=1: W16^3
  v[14] = ctw<INV>(v[14], Make<C>::of(-R2, -R2));   // k1=3,n2=2: W16^6
  v[15] = ctw<INV>(v[15], Make<C>::of(-C1, S1));    // k1=3,n2=3: W16^9
  // stage 2: over n2 for each k1: v[4*k1 + k2] <- X[k1 + 4*k2]
  dft4<INV>(v[0], v[1], v[2], v[3]);
  dft4<INV>(v[4], v[5], v[6], v[7]);
  dft4_crot<INV>(v[8], v[9], v[10], v[11]);
  dft4<INV>(v[12], v[13], v[14], v[15]);
  // transpose to natural order: out[k1 + 4*k2] = v[4*k1 + k2]
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = a + 1; b < 4; b++) {
      const C t = v[4 * a + b];
      v[4 * a + b] = v[4 * b + a];
      v[4 * b + a] = t;
    }
}

// ---- per-lane constants ------------------------------------------------------------------
// tw1[k1] = W_1024^(lane*k1)            (stage A, lane = n2)
// tw2[j1] = W_64^((lane&3)*j1)          (stage B, lane = 4*k1 + m2)
// Tables are produced on the host in double precision (fill_twiddles) and laid out
// [16][64] so a wave reads them with coalesced loads.
inline void fill_twiddles(cpx *tw1, cpx *tw2)
{
  const double PI = 3.14159265358979323846;
  for (int r = 0; r < 16; r++)
    for (int lane = 0; lane < 64; lane++) {
      double a1 = -2.0 * PI * (double) (lane * r) / 1024.0;
      tw1[r * 64 + lane] = mk((float) std::cos(a1), (float) std::sin(a1));
      double a2 = -2.0 * PI * (double) ((lane & 3) * r) / 64.0;
      tw2[r * 64 + lane] = mk((float) std::cos(a2), (float) std::sin(a2));
    }
}

// frequency bin held by (lane, reg) after forward(): lane = 4*k1 + j1lo, reg = 4*j1hi + j2
TSD_HD int freq_index(int lane, int reg)
{
  const int k1 = lane >> 2, j1lo = lane & 3, j1hi = reg >> 2, j2 = reg & 3;
  return k1 + 16 * (4 * j1hi + j1lo) + 256 * j2;
}
// time sample held by (lane, reg) before forward() / after inverse()
TSD_HD int time_index(int lane, int reg) { return 64 * reg + lane; }

// Twiddles GENERATED instead of held: PowGen{w1, w2, w4, w8} stands for the table tw[r] = w^r, r = 1..15, of which only the
// entries 1, 2, 4, 8 are kept (the table's own correctly rounded values); the others are products of the kept ones along
// the binary expansion of r (w3 = w2 w1, w5 = w4 w1, w6 = w4 w2, w7 = w6 w1 ... w15 = w8 w7: 11 complex products per stage).
// A generated entry carries the roundings of at most four table values and three products -- raising ONE base value to the
// r-th power would multiply its rounding by r (measured on the CPU emulation: 5 x the table's error; this form: see
// tests/cpu/test_fft1024_wave.cc) -- and the 2 x 15 complex table entries of a lane (60 VGPRs) become 2 x 4 (16).
template <typename C> struct PowGen {
  C w1, w2, w4, w8;
};
template <typename T> struct IsPowGen {
  static constexpr bool v = false;
};
template <typename C> struct IsPowGen<PowGen<C>> {
  static constexpr bool v = true;
};
// v[r] <- v[r] w^r (forward) or v[r] conj(w^r) (inverse), r = 1..15
template <bool INV, typename C> TSD_HD void mul_powers(C (&v)[16], const PowGen<C> &g)
{
  const C p3 = cmul(g.w2, g.w1), p5 = cmul(g.w4, g.w1), p6 = cmul(g.w4, g.w2), p7 = cmul(p6, g.w1);
  v[1] = ctw<INV>(v[1], g.w1);
  v[2] = ctw<INV>(v[2], g.w2);
  v[3] = ctw<INV>(v[3], p3);
  v[4] = ctw<INV>(v[4], g.w4);
  v[5] = ctw<INV>(v[5], p5);
  v[6] = ctw<INV>(v[6], p6);
  v[7] = ctw<INV>(v[7], p7);
  v[8] = ctw<INV>(v[8], g.w8);
  v[9] = ctw<INV>(v[9], cmul(g.w8, g.w1));
  v[10] = ctw<INV>(v[10], cmul(g.w8, g.w2));
  v[11] = ctw<INV>(v[11], cmul(g.w8, p3));
  v[12] = ctw<INV>(v[12], cmul(g.w8, g.w4));
  v[13] = ctw<INV>(v[13], cmul(g.w8, p5));
  v[14] = ctw<INV>(v[14], cmul(g.w8, p6));
  v[15] = ctw<INV>(v[15], cmul(g.w8, p7));
}

// ---- the phases.  SYNC() must order LDS writes before the following LDS reads of the
// same wave (a __syncthreads() in a 64-lane workgroup; a no-op per-phase loop on the CPU).
template <bool INV, typename C, typename TW> TSD_HD void stage_twiddles(C (&v)[16], const TW &tw)
{
  if constexpr (IsPowGen<TW>::v) {
    mul_powers<INV>(v, tw);
  } else {
#pragma unroll
    for (int r = 1; r < 16; r++) v[r] = ctw<INV>(v[r], tw[r]);
  }
}
template <bool INV, typename C, typename TW> TSD_HD void stageA(C (&v)[16], const TW &tw1)
{
  if (!INV) {
    dft16<false>(v);
    stage_twiddles<false>(v, tw1);
  } else {
    stage_twiddles<true>(v, tw1);
    dft16<true>(v);
  }
}
template <bool INV, typename C, typename TW> TSD_HD void stageB(C (&v)[16], const TW &tw2)
{
  if (!INV) {
    dft16<false>(v);
    stage_twiddles<false>(v, tw2);
  } else {
    stage_twiddles<true>(v, tw2);
    dft16<true>(v);
  }
}
template <bool INV, typename C> TSD_HD void stageC(C (&v)[16])
{
#pragma unroll
  for (int h = 0; h < 4; h++) dft4<INV>(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
}

// exchange 1, forward direction: lane n2 / reg k1  ->  lane (k1,m2) / reg m1
template <int S = 1, typename C> TSD_HD void x1_write_rows(const C (&v)[16], C *lds, int lane)
{
#pragma unroll
  for (int r = 0; r < 16; r++) lds[(LDS_ROW * r + lane) * S] = v[r];
}
template <int S = 1, typename C> TSD_HD void x1_read_rows(C (&v)[16], const C *lds, int lane)
{
#pragma unroll
  for (int r = 0; r < 16; r++) v[r] = lds[(LDS_ROW * r + lane) * S];
}
template <int S = 1, typename C> TSD_HD void x1_write_cols(const C (&v)[16], C *lds, int lane)
{
  const int base = LDS_ROW * (lane >> 2) + (lane & 3);
#pragma unroll
  for (int m1 = 0; m1 < 16; m1++) lds[(base + 4 * m1) * S] = v[m1];
}
template <int S = 1, typename C> TSD_HD void x1_read_cols(C (&v)[16], const C *lds, int lane)
{
  const int base = LDS_ROW * (lane >> 2) + (lane & 3);
#pragma unroll
  for (int m1 = 0; m1 < 16; m1++) v[m1] = lds[(base + 4 * m1) * S];
}
// exchange 2: image [k1][m2][j1] with plane stride 17
template <int S = 1, typename C> TSD_HD void x2_write_j1(const C (&v)[16], C *lds, int lane)     // lane (k1,m2), reg j1
{
  const int base = LDS_ROW * (lane >> 2) + 17 * (lane & 3);
#pragma unroll
  for (int j1 = 0; j1 < 16; j1++) lds[(base + j1) * S] = v[j1];
}
template <int S = 1, typename C> TSD_HD void x2_read_j1(C (&v)[16], const C *lds, int lane)
{
  const int base = LDS_ROW * (lane >> 2) + 17 * (lane & 3);
#pragma unroll
  for (int j1 = 0; j1 < 16; j1++) v[j1] = lds[(base + j1) * S];
}
template <int S = 1, typename C> TSD_HD void x2_write_m2(const C (&v)[16], C *lds, int lane)     // lane (k1,j1lo), reg (j1hi,m2)
{
  const int base = LDS_ROW * (lane >> 2) + (lane & 3);
#pragma unroll
  for (int h = 0; h < 4; h++)
#pragma unroll
    for (int m2 = 0; m2 < 4; m2++) lds[(base + 17 * m2 + 4 * h) * S] = v[4 * h + m2];
}
template <int S = 1, typename C> TSD_HD void x2_read_m2(C (&v)[16], const C *lds, int lane)
{
  const int base = LDS_ROW * (lane >> 2) + (lane & 3);
#pragma unroll
  for (int h = 0; h < 4; h++)
#pragma unroll
    for (int m2 = 0; m2 < 4; m2++) v[4 * h + m2] = lds[(base + 17 * m2 + 4 * h) * S];
}

// Whole transforms for one lane; SYNC is a callable.  Unnormalised (the caller folds 1/N or
// 1/sqrt(N) into a later multiply).
// S = element stride of the wave's LDS image (1: private contiguous buffer; an ODD stride such
// as 9 interleaves the images of several waves -- multiplying slot numbers by an odd constant
// permutes the banks, so the conflict-free property of the maps above is preserved).
// TW1/TW2: anything indexable with [r], r = 1..15 (register arrays, or an accessor over LDS).
template <int S = 1, typename C, typename TW1, typename TW2, typename SYNC>
TSD_HD void forward(C (&v)[16], C *lds, int lane, const TW1 &tw1, const TW2 &tw2, SYNC sync)
{
  stageA<false>(v, tw1);
  x1_write_rows<S>(v, lds, lane);
  sync();
  x1_read_cols<S>(v, lds, lane);
  stageB<false>(v, tw2);
  sync();
  x2_write_j1<S>(v, lds, lane);
  sync();
  x2_read_m2<S>(v, lds, lane);
  stageC<false>(v);
}
template <int S = 1, typename C, typename TW1, typename TW2, typename SYNC>
TSD_HD void inverse(C (&v)[16], C *lds, int lane, const TW1 &tw1, const TW2 &tw2, SYNC sync)
{
  stageC<true>(v);
  sync();
  x2_write_m2<S>(v, lds, lane);
  sync();
  x2_read_j1<S>(v, lds, lane);
  stageB<true>(v, tw2);
  sync();
  x1_write_cols<S>(v, lds, lane);
  sync();
  x1_read_rows<S>(v, lds, lane);
  stageA<true>(v, tw1);
}

// ---- TWO 512-point transforms in one wave (round 3: the windowed overlap-add engine at its default geometry) ----------------
// The same three stages and LDS exchanges with one bit of the engine's last digit m2 = 2 sigma + eps standing for the SEQUENCE:
// 512 = 16 x 16 x 2, v = 32 n1 + 2 m1 + eps, k = k1 + 16 j1 + 256 j2 --
//   time:  lane n2 = 4 m1 + 2 sigma + eps, reg n1   holds sample v = 32 n1 + 2 m1 + eps of sequence sigma
//   stage A twiddle W_512^((2 m1 + eps) k1), stage B twiddle W_32^(eps j1), stage C a radix-2 over eps per sequence
//   freq:  lane 4 k1 + j1lo, reg 4 j1hi + 2 sigma + j2   holds bin k of sequence sigma (both sequences' bin k in one lane)
// i.e. forward() / inverse() with the tables of fill_twiddles_pair512 and stageC2 in place of stageC.
inline void fill_twiddles_pair512(cpx *tw1, cpx *tw2)
{
  const double PI = 3.14159265358979323846;
  for (int r = 0; r < 16; r++)
    for (int lane = 0; lane < 64; lane++) {
      const int w = 2 * (lane >> 2) + (lane & 1);
      double a1 = -2.0 * PI * (double) (w * r) / 512.0;
      tw1[r * 64 + lane] = mk((float) std::cos(a1), (float) std::sin(a1));
      double a2 = -2.0 * PI * (double) ((lane & 1) * r) / 32.0;
      tw2[r * 64 + lane] = mk((float) std::cos(a2), (float) std::sin(a2));
    }
}
TSD_HD int pair512_seq_of_lane(int lane) { return (lane >> 1) & 1; }
TSD_HD int pair512_time_index(int lane, int reg) { return 32 * reg + 2 * (lane >> 2) + (lane & 1); }
TSD_HD int pair512_seq_of_reg(int reg) { return (reg >> 1) & 1; }
TSD_HD int pair512_freq_index(int lane, int reg)
{
  const int k1 = lane >> 2, j1lo = lane & 3, j1hi = reg >> 2, j2 = reg & 1;
  return k1 + 16 * (4 * j1hi + j1lo) + 256 * j2;
}
template <typename C> TSD_HD void stageC2(C (&v)[16])        // (a radix-2 butterfly is its own inverse)
{
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const C a = v[2 * q], b = v[2 * q + 1];
    v[2 * q] = cadd(a, b);
    v[2 * q + 1] = csub(a, b);
  }
}
template <int S = 1, typename C, typename TW1, typename TW2, typename SYNC>
TSD_HD void forward_pair512(C (&v)[16], C *lds, int lane, const TW1 &tw1, const TW2 &tw2, SYNC sync)
{
  stageA<false>(v, tw1);
  x1_write_rows<S>(v, lds, lane);
  sync();
  x1_read_cols<S>(v, lds, lane);
  stageB<false>(v, tw2);
  sync();
  x2_write_j1<S>(v, lds, lane);
  sync();
  x2_read_m2<S>(v, lds, lane);
  stageC2(v);
}
template <int S = 1, typename C, typename TW1, typename TW2, typename SYNC>
TSD_HD void inverse_pair512(C (&v)[16], C *lds, int lane, const TW1 &tw1, const TW2 &tw2, SYNC sync)
{
  stageC2(v);
  sync();
  x2_write_m2<S>(v, lds, lane);
  sync();
  x2_read_j1<S>(v, lds, lane);
  stageB<true>(v, tw2);
  sync();
  x1_write_cols<S>(v, lds, lane);
  sync();
  x1_read_rows<S>(v, lds, lane);
  stageA<true>(v, tw1);
}

}  // namespace w1024
}  // namespace tsdgpu
