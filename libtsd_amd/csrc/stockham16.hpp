// stockham16.hpp -- building blocks of the radix-16 Stockham FFT held in LDS (device only).
//
// A transform of n = R0 * 16^a points (R0 in {16, 8, 4, 2}) is owned by tpt = n/16 threads;
// thread j holds 16 points in registers in every pass:
//   pass 0 (radix R0, no twiddles): v[m] = x[j + m*tpt]; its 16/R0 butterflies jb = j + i*tpt
//            take v[i + q*(16/R0)], q < R0, and write y[jb*R0 + q];
//   pass Ns (radix 16, sub-transform length Ns = R0, 16 R0, ...): v[q] = x[j + q*tpt], times
//            W_{16 Ns}^(q k) with k = j mod Ns, 16-point DFT, write y[(j-k)*16 + k + q*Ns];
//   after the last pass (16 Ns = n) thread j holds X[j + q*tpt] -- natural order, and exactly
//   the register layout pass 0 expects, so a second transform can start without an exchange.
// LDS index i lives at i + i/16 (keeps the stride-16 writes conflict-free).
// Used by fft.hip (fft_s16_kernel, fft_cols16_kernel) and ols_long.hip.
#pragma once
#include "fft1024_wave.hpp"

namespace tsdgpu {
namespace s16 {

__device__ __forceinline__ cpx c_mk(float a, float b) { return make_float2(a, b); }
__device__ __forceinline__ cpx c_add(cpx a, cpx b) { return c_mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cpx c_sub(cpx a, cpx b) { return c_mk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cpx c_mul(cpx a, cpx b) { return c_mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__device__ __forceinline__ int pad(int i) { return i + (i >> 4); }
__device__ __forceinline__ int padded_size(int n) { return n + (n >> 4); }

// The butterflies below are written on the primitives of fft1024_wave.hpp and templated on the complex type: cpx (two scalar
// registers) or w1024::v2f (a 64-bit register pair on packed VOP3P arithmetic, half the instructions).
template <typename C> __device__ __forceinline__ void dft2(C &a, C &b)
{
  const C t = a;
  a = w1024::cadd(t, b);
  b = w1024::csub(t, b);
}
template <typename C> __device__ __forceinline__ void dft8(C (&e)[8])
{
  constexpr float R2 = 0.70710678118654752f;
  w1024::dft4<false>(e[0], e[2], e[4], e[6]);               // even samples -> E[0..3] in e[0],e[2],e[4],e[6]
  w1024::dft4<false>(e[1], e[3], e[5], e[7]);               // odd samples  -> O[0..3] in e[1],e[3],e[5],e[7]
  const C o0 = e[1], o1 = w1024::cmul(e[3], w1024::Make<C>::of(R2, -R2)), o2 = w1024::rot90<false>(e[5]),
          o3 = w1024::cmul(e[7], w1024::Make<C>::of(-R2, -R2));
  const C a0 = e[0], a1 = e[2], a2 = e[4], a3 = e[6];
  e[0] = w1024::cadd(a0, o0); e[4] = w1024::csub(a0, o0);
  e[1] = w1024::cadd(a1, o1); e[5] = w1024::csub(a1, o1);
  e[2] = w1024::cadd(a2, o2); e[6] = w1024::csub(a2, o2);
  e[3] = w1024::cadd(a3, o3); e[7] = w1024::csub(a3, o3);
}

// pass 0 in registers: afterwards v[i + q*(16/R0)] = output q of butterfly jb = j + i*tpt
template <int R0, typename C> __device__ __forceinline__ void pass0(C (&v)[16])
{
  if (R0 == 16) {
    w1024::dft16<false>(v);
  } else if (R0 == 8) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      C e[8];
#pragma unroll
      for (int q = 0; q < 8; q++) e[q] = v[i + 2 * q];
      dft8(e);
#pragma unroll
      for (int q = 0; q < 8; q++) v[i + 2 * q] = e[q];
    }
  } else if (R0 == 4) {
#pragma unroll
    for (int i = 0; i < 4; i++) w1024::dft4<false>(v[i], v[i + 4], v[i + 8], v[i + 12]);
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) dft2(v[i], v[i + 8]);
  }
}
// ... and its write to the (padded) LDS image s of the transform
template <int R0, typename C> __device__ __forceinline__ void pass0_store(C *s, const C (&v)[16], int j, int tpt)
{
#pragma unroll
  for (int i = 0; i < 16 / R0; i++)
#pragma unroll
    for (int q = 0; q < R0; q++) s[pad((j + i * tpt) * R0 + q)] = v[i + q * (16 / R0)];
}
// v[q] *= w1^q, q = 1..15, from ONE table value: products of depth <= 4
template <typename C> __device__ __forceinline__ void twiddle_powers(C (&v)[16], C w1)
{
  using w1024::cmul;
  const C w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
  const C w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3), w8 = cmul(w4, w4);
  v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3); v[4] = cmul(v[4], w4);
  v[5] = cmul(v[5], w5); v[6] = cmul(v[6], w6); v[7] = cmul(v[7], w7); v[8] = cmul(v[8], w8);
  v[9] = cmul(v[9], cmul(w8, w1)); v[10] = cmul(v[10], cmul(w8, w2)); v[11] = cmul(v[11], cmul(w8, w3));
  v[12] = cmul(v[12], cmul(w8, w4)); v[13] = cmul(v[13], cmul(w8, w5)); v[14] = cmul(v[14], cmul(w8, w6));
  v[15] = cmul(v[15], cmul(w8, w7));
}
// One radix-16 pass on registers already read from LDS: twiddle + DFT; returns the base index
// of the outputs: y[base + q*Ns].  TW[i] = W_n^i, i < n/16.
template <typename C> __device__ __forceinline__ int pass16(C (&v)[16], const C *__restrict__ TW, int j, int tpt, int Ns)
{
  const int k = j & (Ns - 1);
  twiddle_powers(v, TW[k * (tpt / Ns)]);
  w1024::dft16<false>(v);
  return (j - k) * 16 + k;
}

// A whole forward transform of the 16 register values of thread j (input v[m] = x[j + m*tpt]),
// result X[j + q*tpt] in v[q]; s = the transform's padded LDS image; SYNC = barrier over the
// threads of the transform.  Unnormalised.
template <int R0, typename C, typename SYNC>
__device__ __forceinline__ void transform(C (&v)[16], C *s, const C *__restrict__ TW, int n, int j, int tpt, SYNC sync)
{
  pass0<R0>(v);
  if (n == R0) return;
  pass0_store<R0>(s, v, j, tpt);
  for (int Ns = R0;; Ns <<= 4) {
    sync();
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = s[pad(j + q * tpt)];
    const int base = pass16(v, TW, j, tpt, Ns);
    if (Ns * 16 == n) return;                                // base = j: natural order in registers
    sync();
#pragma unroll
    for (int q = 0; q < 16; q++) s[pad(base + q * Ns)] = v[q];
  }
}

}  // namespace s16
}  // namespace tsdgpu
