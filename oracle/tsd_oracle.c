/* tsd_oracle.c -- CPU restatement of libtsd's FIR / IIR(SOS) / FFT / resample hot path.
 * TEST INFRASTRUCTURE ONLY (see tsd_oracle.h for the rules and the parity status).
 * Paths in comments are relative to /root/reference/core/.
 */
#include "tsd_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PI_D 3.14159265358979323846   /* tsd.hpp:66  static const double pi */
static const float PI_F = 3.14159265358979323846f; /* tsd.hpp:69 */

static inline orc_cf cf(float re, float im) { orc_cf r = {re, im}; return r; }
static inline orc_cf cadd(orc_cf a, orc_cf b) { return cf(a.re + b.re, a.im + b.im); }
static inline orc_cf csub(orc_cf a, orc_cf b) { return cf(a.re - b.re, a.im - b.im); }
/* complex product as gcc emits it under -fcx-limited-range (std-makefile-defs:176) */
static inline orc_cf cmul(orc_cf a, orc_cf b)
{ return cf(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline orc_cf cscale(orc_cf a, float s) { return cf(a.re * s, a.im * s); }
static inline orc_cf cconj(orc_cf a) { return cf(a.re, -a.im); }
/* limited-range complex division */
static inline orc_cf cdiv(orc_cf a, orc_cf b)
{
  float d = b.re * b.re + b.im * b.im;
  return cf((a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d);
}

/* ======================================================================================
 * FIR -- src/filtrage/filtre-rt.cc:67-108
 *   fen(index) = x; index = (index+1) % K;  then sum over the K window samples starting
 *   at the oldest one, against coefs[K-1], coefs[K-2], ... coefs[0] (cptr decrements).
 * ==================================================================================== */
void orc_fir_ff(const float *coefs, int K, float *fen, int *index,
                const float *x, float *y, int64_t n)
{
  int idx = *index;
  for (int64_t j = 0; j < n; j++) {
    const float *cptr = coefs + (K - 1);
    float somme = 0;
    fen[idx] = x[j];
    idx = (idx + 1) % K;
    const float *fptr = fen + idx;
    int K1 = K - idx, K2 = K - K1;
    for (int i = 0; i < K1; i++) somme += *fptr++ * *cptr--;
    fptr = fen;
    for (int i = 0; i < K2; i++) somme += *fptr++ * *cptr--;
    y[j] = somme;
  }
  *index = idx;
}

void orc_fir_cf(const float *coefs, int K, orc_cf *fen, int *index,
                const orc_cf *x, orc_cf *y, int64_t n)
{
  int idx = *index;
  for (int64_t j = 0; j < n; j++) {
    const float *cptr = coefs + (K - 1);
    float sr = 0, si = 0;
    fen[idx] = x[j];
    idx = (idx + 1) % K;
    const orc_cf *fptr = fen + idx;
    int K1 = K - idx, K2 = K - K1;
    /* complex<float> * float = component-wise scale */
    for (int i = 0; i < K1; i++) { float c = *cptr--; sr += fptr->re * c; si += fptr->im * c; fptr++; }
    fptr = fen;
    for (int i = 0; i < K2; i++) { float c = *cptr--; sr += fptr->re * c; si += fptr->im * c; fptr++; }
    y[j] = cf(sr, si);
  }
  *index = idx;
}

void orc_fir_cc(const orc_cf *coefs, int K, orc_cf *fen, int *index,
                const orc_cf *x, orc_cf *y, int64_t n)
{
  int idx = *index;
  for (int64_t j = 0; j < n; j++) {
    const orc_cf *cptr = coefs + (K - 1);
    orc_cf somme = cf(0, 0);
    fen[idx] = x[j];
    idx = (idx + 1) % K;
    const orc_cf *fptr = fen + idx;
    int K1 = K - idx, K2 = K - K1;
    for (int i = 0; i < K1; i++) somme = cadd(somme, cmul(*fptr++, *cptr--));
    fptr = fen;
    for (int i = 0; i < K2; i++) somme = cadd(somme, cmul(*fptr++, *cptr--));
    y[j] = somme;
  }
  *index = idx;
}

/* ======================================================================================
 * Direct-form-I IIR -- src/filtrage/filtre-rt.cc:210-280
 * ==================================================================================== */
void orc_rii_f(const float *numer, int Kx, const float *denom, int Ky,
               float *wndx, float *wndy, int *index, int *index_y,
               const float *x, float *y, int64_t n)
{
  int idx = *index, idy = *index_y;
  /* (1) non-recursive part, :224-248 */
  for (int64_t j = 0; j < n; j++) {
    const float *nptr = numer + (Kx - 1);
    float somme = 0;
    wndx[idx] = x[j];
    idx = (idx + 1) % Kx;
    const float *wptr = wndx + idx;
    int KK1 = Kx - idx, KK2 = Kx - KK1;
    for (int i = 0; i < KK1; i++) somme += *wptr++ * *nptr--;
    wptr = wndx;
    for (int i = 0; i < KK2; i++) somme += *wptr++ * *nptr--;
    y[j] = somme;
  }
  /* (2) recursive part, :251-279 */
  for (int64_t j = 0; j < n; j++) {
    const float *dptr = denom + 1;
    float somme = y[j];
    if (Ky > 0) {
      const float *wptr = wndy + idy;
      int KK1 = Ky - idy, KK2 = Ky - KK1;
      for (int i = 0; i < KK1; i++) somme -= *wptr++ * *dptr++;
      wptr = wndy;
      for (int i = 0; i < KK2; i++) somme -= *wptr++ * *dptr++;
    }
    y[j] = somme / denom[0];
    if (Ky > 0) {
      idy = (idy + Ky - 1) % Ky;
      wndy[idy] = y[j];
    }
  }
  *index = idx; *index_y = idy;
}

/* the same for T = Tc = complex<float> (filtre_rii<cfloat,cfloat>, instantiated at filtre-rt.cc:795):
 * interleaved (re, im); products and the final division in limited-range complex arithmetic, the
 * way -fcx-limited-range (core/std-makefile-defs:176) compiles std::complex<float> */
void orc_rii_c(const float *numer, int Kx, const float *denom, int Ky,
               float *wndx, float *wndy, int *index, int *index_y,
               const float *x, float *y, int64_t n)
{
  int idx = *index, idy = *index_y;
  for (int64_t j = 0; j < n; j++) {
    float sr = 0, si = 0;
    wndx[2 * idx] = x[2 * j]; wndx[2 * idx + 1] = x[2 * j + 1];
    idx = (idx + 1) % Kx;
    for (int i = 0; i < Kx; i++) {                       /* oldest sample x last coefficient first */
      const float *w = wndx + 2 * ((idx + i) % Kx), *c = numer + 2 * (Kx - 1 - i);
      sr += w[0] * c[0] - w[1] * c[1];
      si += w[0] * c[1] + w[1] * c[0];
    }
    y[2 * j] = sr; y[2 * j + 1] = si;
  }
  const float d0r = denom[0], d0i = denom[1], nd = d0r * d0r + d0i * d0i;
  for (int64_t j = 0; j < n; j++) {
    float sr = y[2 * j], si = y[2 * j + 1];
    for (int i = 0; i < Ky; i++) {
      const float *w = wndy + 2 * ((idy + i) % Ky), *c = denom + 2 * (1 + i);
      sr -= w[0] * c[0] - w[1] * c[1];
      si -= w[0] * c[1] + w[1] * c[0];
    }
    y[2 * j] = (sr * d0r + si * d0i) / nd;
    y[2 * j + 1] = (si * d0r - sr * d0i) / nd;
    if (Ky > 0) {
      idy = (idy + Ky - 1) % Ky;
      wndy[2 * idy] = y[2 * j]; wndy[2 * idy + 1] = y[2 * j + 1];
    }
  }
  *index = idx; *index_y = idy;
}

/* ======================================================================================
 * SOS chain -- src/filtrage/filtre-rt.cc:440-572
 * ==================================================================================== */
int orc_sos_from_zpk(orc_sos *s, const orc_cf *z, const orc_cf *p, int n,
                     orc_cf mlt_num, orc_cf mlt_den, int forme)
{
  /* pool = ordered set of the indices not yet consumed (:468-470) */
  char *in_pool = (char *) calloc((size_t) n + 1, 1);
  for (int i = 0; i < n; i++) in_pool[i] = 1;
  memset(s, 0, sizeof(*s));
  s->forme = forme;
  s->gain = 1.0f;
  int i;
  for (i = 0; i + 1 < n; i += 2) {
    int k = 0;
    while (!in_pool[k]) k++;              /* *(pool.begin()) */
    in_pool[k] = 0;
    float berr = 1e9f, sz = 0, pz = 0, sp = 0, pp = 0;
    int bj = 0;
    while (!in_pool[bj]) bj++;            /* bj = pool.begin() */
    for (int j = 0; j < n; j++) {
      if (!in_pool[j]) continue;
      orc_cf sz0 = cadd(z[j], z[k]); sz0 = cf(-sz0.re, -sz0.im);
      orc_cf pz0 = cmul(z[j], z[k]);
      orc_cf sp0 = cadd(p[j], p[k]); sp0 = cf(-sp0.re, -sp0.im);
      orc_cf pp0 = cmul(p[j], p[k]);
      float err = fabsf(sz0.im) + fabsf(pz0.im) + fabsf(sp0.im) + fabsf(pp0.im);
      if (err < berr) {
        bj = j; berr = err;
        sz = sz0.re; pz = pz0.re; sp = sp0.re; pp = pp0.re;
      }
    }
    in_pool[bj] = 0;
    /* coefs = {1, sz, pz, 1, sp, pp}, normalised by a0 = 1 (:317-328,522-525) */
    orc_biquad *b = &s->sec[s->nsec++];
    b->b0 = 1.0f / 1.0f; b->b1 = sz / 1.0f; b->b2 = pz / 1.0f;
    b->a1 = sp / 1.0f;   b->a2 = pp / 1.0f;
  }
  for (; i < n; i++) {
    int id = 0;
    while (!in_pool[id]) id++;
    orc_cf zer = z[id], pol = p[id];
    float a1 = -pol.re, b0 = mlt_num.re / mlt_den.re, b1 = -zer.re * b0;   /* :550-552 */
    s->r_b0 = b0; s->r_b1 = b1; s->r_a1 = a1;
    s->avec_rii1 = 1;
  }
  if (!s->avec_rii1) s->gain = mlt_num.re / mlt_den.re;                     /* :558-559 */
  free(in_pool);
  return s->nsec;
}

void orc_sos_state_init_f(orc_sos_state_f *st)
{ memset(st, 0, sizeof(*st)); for (int i = 0; i < 32; i++) st->sec[i].premier_appel = 1; }
void orc_sos_state_init_c(orc_sos_state_c *st)
{ memset(st, 0, sizeof(*st)); for (int i = 0; i < 32; i++) st->sec[i].premier_appel = 1; }

/* SOIS::step, :347-397.  In-place on y. */
static void biquad_step_f(const orc_biquad *c, orc_biquad_state_f *s, int forme, float *y, int64_t n)
{
  if (n == 0) return;
  if (s->premier_appel) {                     /* :361-365 */
    s->y0 = s->y1 = s->y2 = s->x2 = s->x1 = y[0];
    s->premier_appel = 0;
  }
  const float b0 = c->b0, b1 = c->b1, b2 = c->b2, a1 = c->a1, a2 = c->a2;
  if (forme == 2) {                           /* :369-380 */
    float y0 = s->y0, y1 = s->y1, y2 = s->y2;
    for (int64_t i = 0; i < n; i++) {
      float x0 = y[i];
      float d2 = x0 - a1 * y1 - a2 * y0;
      y2 = b0 * d2 + b1 * y1 + b2 * y0;
      y[i] = y2;
      y0 = y1;
      y1 = d2;
    }
    s->y0 = y0; s->y1 = y1; s->y2 = y2;
  } else {                                    /* :384-393 */
    float x1 = s->x1, x2 = s->x2, y0 = s->y0, y1 = s->y1, y2 = s->y2;
    for (int64_t i = 0; i < n; i++) {
      float x0 = y[i];
      y0 = b0 * x0 + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2;
      y2 = y1; y1 = y0; x2 = x1; x1 = x0;
      y[i] = y0;
    }
    s->x1 = x1; s->x2 = x2; s->y0 = y0; s->y1 = y1; s->y2 = y2;
  }
}

/* T = cfloat: ChaineSOIS<T,T,T> holds the (real-valued) coefficients as cfloat, so each
 * product is a complex*complex with a zero imaginary coefficient == component-wise scale. */
static void biquad_step_c(const orc_biquad *c, orc_biquad_state_c *s, int forme, orc_cf *y, int64_t n)
{
  if (n == 0) return;
  if (s->premier_appel) {
    s->y0 = s->y1 = s->y2 = s->x2 = s->x1 = y[0];
    s->premier_appel = 0;
  }
  const float b0 = c->b0, b1 = c->b1, b2 = c->b2, a1 = c->a1, a2 = c->a2;
  if (forme == 2) {
    orc_cf y0 = s->y0, y1 = s->y1, y2 = s->y2;
    for (int64_t i = 0; i < n; i++) {
      orc_cf x0 = y[i];
      orc_cf d2 = csub(csub(x0, cscale(y1, a1)), cscale(y0, a2));
      y2 = cadd(cadd(cscale(d2, b0), cscale(y1, b1)), cscale(y0, b2));
      y[i] = y2;
      y0 = y1;
      y1 = d2;
    }
    s->y0 = y0; s->y1 = y1; s->y2 = y2;
  } else {
    orc_cf x1 = s->x1, x2 = s->x2, y0 = s->y0, y1 = s->y1, y2 = s->y2;
    for (int64_t i = 0; i < n; i++) {
      orc_cf x0 = y[i];
      y0 = csub(csub(cadd(cadd(cscale(x0, b0), cscale(x1, b1)), cscale(x2, b2)),
                     cscale(y1, a1)), cscale(y2, a2));
      y2 = y1; y1 = y0; x2 = x1; x1 = x0;
      y[i] = y0;
    }
    s->x1 = x1; s->x2 = x2; s->y0 = y0; s->y1 = y1; s->y2 = y2;
  }
}

void orc_sos_step_f(const orc_sos *s, orc_sos_state_f *st, const float *x, float *y, int64_t n)
{
  if (y != x) memmove(y, x, (size_t) n * sizeof(float));      /* y = x.clone(), :564 */
  for (int k = 0; k < s->nsec; k++)                            /* :565-566 */
    biquad_step_f(&s->sec[k], &st->sec[k], s->forme, y, n);
  if (s->avec_rii1) {                                          /* RIIFoS::step :427-433 */
    float x1 = st->r_x1, y1 = st->r_y1;
    for (int64_t i = 0; i < n; i++) {
      float x0 = y[i];
      y1 = -s->r_a1 * y1 + s->r_b0 * x0 + s->r_b1 * x1;
      y[i] = y1;
      x1 = x0;
    }
    st->r_x1 = x1; st->r_y1 = y1;
  } else {
    for (int64_t i = 0; i < n; i++) y[i] *= s->gain;           /* :570 */
  }
}

void orc_sos_step_c(const orc_sos *s, orc_sos_state_c *st, const orc_cf *x, orc_cf *y, int64_t n)
{
  if (y != x) memmove(y, x, (size_t) n * sizeof(orc_cf));
  for (int k = 0; k < s->nsec; k++)
    biquad_step_c(&s->sec[k], &st->sec[k], s->forme, y, n);
  if (s->avec_rii1) {
    orc_cf x1 = st->r_x1, y1 = st->r_y1;
    for (int64_t i = 0; i < n; i++) {
      orc_cf x0 = y[i];
      y1 = cadd(cadd(cscale(y1, -s->r_a1), cscale(x0, s->r_b0)), cscale(x1, s->r_b1));
      y[i] = y1;
      x1 = x0;
    }
    st->r_x1 = x1; st->r_y1 = y1;
  } else {
    for (int64_t i = 0; i < n; i++) y[i] = cscale(y[i], s->gain);
  }
}

/* Same chain (DF2, first-sample seed, gain / first-order tail) evaluated in double: NOT a
 * reference path -- the conditioning yardstick the parity tests use to tell float32 rounding
 * noise of the recurrence itself from an implementation error. One-shot (no carried state). */
void orc_sos_run_f64(const orc_sos *s, const float *x, double *y, int64_t n)
{
  for (int64_t i = 0; i < n; i++) y[i] = x[i];
  for (int k = 0; k < s->nsec; k++) {
    const double b0 = s->sec[k].b0, b1 = s->sec[k].b1, b2 = s->sec[k].b2, a1 = s->sec[k].a1, a2 = s->sec[k].a2;
    double y0 = n ? y[0] : 0, y1 = y0;
    for (int64_t i = 0; i < n; i++) {
      double d2 = y[i] - a1 * y1 - a2 * y0;
      y[i] = b0 * d2 + b1 * y1 + b2 * y0;
      y0 = y1; y1 = d2;
    }
  }
  if (s->avec_rii1) {
    double x1 = 0, y1 = 0;
    for (int64_t i = 0; i < n; i++) {
      double x0 = y[i];
      y1 = -(double) s->r_a1 * y1 + (double) s->r_b0 * x0 + (double) s->r_b1 * x1;
      y[i] = y1; x1 = x0;
    }
  } else {
    for (int64_t i = 0; i < n; i++) y[i] *= (double) s->gain;
  }
}

/* ======================================================================================
 * FFT -- src/fourier/fourier.cc
 * ==================================================================================== */
int orc_next_pow2(int i)                      /* src/tsd.cc:287-291 */
{
  int lg2 = (int) ceilf(logf((float) i) / logf(2.0f));
  return (int) (1l << lg2);
}

void orc_fft_twiddles(orc_cf *rot, int n)     /* tfr_rotation_rapide, fourier.cc:32-46 */
{
  double rr = 1.0, ri = 0.0;
  double a = (-1 * 2 * PI_D) / n;
  double wr = cos(a), wi = sin(a);            /* std::polar<double>(1.0, a) */
  for (int i = 0; i < n; i++) {
    rot[i] = cf((float) rr, (float) ri);
    double tr = rr * wr - ri * wi, ti = rr * wi + ri * wr;
    rr = tr; ri = ti;
  }
}

/* tfr_radix2, fourier.cc:61-121.  X and scratch have N elements; x is read-only. */
static void radix2(orc_cf *X, const orc_cf *x, orc_cf *scratch, const orc_cf *rot, int N, int avant)
{
  int iteration_paire = ((N & 0x55555555) != 0);
  if (N == 1) { X[0] = x[0]; return; }
  const orc_cf *E = x;
  for (int n = 1; n < N; n *= 2) {
    orc_cf *Xstart = iteration_paire ? scratch : X;
    int pas = N / (2 * n);
    orc_cf *Xp = Xstart, *Xp2 = Xstart + N / 2;
    for (int k = 0; k < n; k++) {
      orc_cf r = rot[k * pas];
      if (!avant) r = cconj(r);
      const float tr = r.re, ti = r.im;
      for (int m = 0; m < pas; m++) {
        const orc_cf g = E[pas], e = *E;
        const orc_cf p = cf(tr * g.re - ti * g.im, tr * g.im + ti * g.re);
        *Xp++ = cadd(e, p);
        *Xp2++ = csub(e, p);
        E++;
      }
      E += pas;
    }
    E = Xstart;
    iteration_paire = !iteration_paire;
  }
  /* X /= sqrt((float) N): complex / real scalar, fourier.cc:120 */
  float s = sqrtf((float) N);
  for (int i = 0; i < N; i++) { X[i].re /= s; X[i].im /= s; }
}

/* tfr_czt_impl, fourier.cc:237-255, chirp from :391-400 */
static void czt(const orc_cf *x, orc_cf *y, int n, int n2, const orc_cf *rot)
{
  orc_cf *chirp = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) (2 * n - 1));
  float *t = (float *) malloc(sizeof(float) * (size_t) (2 * n - 1));
  orc_linspace((float) -(n - 1), (float) (n - 1), 2 * n - 1, t);
  for (int i = 0; i < 2 * n - 1; i++) {
    float v = (t[i] * t[i]) / 2;               /* square(linspace)/2 */
    v *= (float) (-2 * PI_D / n);              /* t *= -2*pi/n */
    chirp[i] = cf(cosf(v), sinf(v));           /* polar(t) */
  }
  orc_cf *xp = (orc_cf *) calloc((size_t) n2, sizeof(orc_cf));
  orc_cf *icp = (orc_cf *) calloc((size_t) n2, sizeof(orc_cf));
  orc_cf *Xp = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) n2);
  orc_cf *Xc = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) n2);
  orc_cf *y2 = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) n2);
  orc_cf *scr = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) n2);
  for (int i = 0; i < n; i++) xp[i] = cmul(x[i], chirp[n - 1 + i]);   /* x * chirp.tail(n) */
  for (int i = 0; i < 2 * n - 1; i++) icp[i] = cconj(chirp[i]);
  radix2(Xp, xp, scr, rot, n2, 1);
  radix2(Xc, icp, scr, rot, n2, 1);
  for (int i = 0; i < n2; i++) Xp[i] = cmul(Xp[i], Xc[i]);
  radix2(y2, Xp, scr, rot, n2, 0);
  float g = sqrtf((float) n2) / sqrtf((float) n);
  for (int i = 0; i < n; i++)
    y[i] = cscale(cmul(y2[n - 1 + i], chirp[n - 1 + i]), g);
  free(chirp); free(t); free(xp); free(icp); free(Xp); free(Xc); free(y2); free(scr);
}

/* tfr2itfr, fourier.cc:259-278: Y(k) = X((n-k) % n) */
static void tfr2itfr(const orc_cf *X, orc_cf *Y, int n)
{
  Y[0] = X[0];
  for (int k = 1; k < n; k++) Y[k] = X[n - k];
}

/* TFRPlanDefaut::step, fourier.cc:412-465 */
void orc_fft(const orc_cf *x, orc_cf *y, int n, int avant)
{
  if (n <= 0) return;
  if ((n & (n - 1)) == 0) {                                   /* power of two, :430-437 */
    orc_cf *rot = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) n);
    orc_cf *scr = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) n);
    orc_fft_twiddles(rot, n);
    radix2(y, x, scr, rot, n, avant);
    free(rot); free(scr);
  } else if ((n & 1) == 0) {                                  /* even: split, :438-463 */
    int h = n / 2;
    orc_cf *xe = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) h);
    orc_cf *xo = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) h);
    orc_cf *E = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) h);
    orc_cf *O = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) h);
    orc_cf *rot = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) n);
    for (int i = 0; i < h; i++) { xe[i] = x[2 * i]; xo[i] = x[2 * i + 1]; }
    orc_fft(xe, E, h, avant);
    orc_fft(xo, O, h, avant);
    orc_fft_twiddles(rot, n);
    const float isq2 = 1 / sqrtf(2.0f);
    for (int i = 0; i < n; i++) {
      orc_cf r = avant ? rot[i] : cconj(rot[i]);
      orc_cf v = cadd(E[i % h], cmul(r, O[i % h]));
      y[i] = cscale(v, isq2);
    }
    free(xe); free(xo); free(E); free(O); free(rot);
  } else {                                                    /* odd: Bluestein, :419-426 */
    int n2 = orc_next_pow2(2 * n - 1);
    orc_cf *rot = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) n2);
    orc_fft_twiddles(rot, n2);
    if (avant) czt(x, y, n, n2, rot);
    else {
      orc_cf *t = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) n);
      czt(x, t, n, n2, rot);
      tfr2itfr(t, y, n);
      free(t);
    }
    free(rot);
  }
}

void orc_csym_force(orc_cf *X, int n)          /* fourier.hpp:264-282 */
{
  X[0].im = 0;
  if ((n & 1) == 0) X[n / 2].im = 0;
  else X[n / 2 + 1] = cconj(X[n / 2]);
  /* X.tail(n/2-1) = X.segment(1,n/2-1).reverse().conjugate() */
  int m = n / 2 - 1;
  for (int i = 0; i < m; i++) X[n - m + i] = cconj(X[1 + (m - 1 - i)]);
}

void orc_rfft(const float *x, orc_cf *y, int n)   /* RTFRPlan::step, fourier.cc:311-354 */
{
  if ((n & 1) == 0) {
    int h = n / 2;
    orc_cf *x2 = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) (h > 0 ? h : 1));
    orc_cf *Xt = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) (h > 0 ? h : 1));
    orc_cf *rot = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) n);
    for (int i = 0; i < h; i++) x2[i] = cf(x[2 * i], x[2 * i + 1]);
    orc_fft(x2, Xt, h, 1);
    orc_fft_twiddles(rot, n);
    const float c = (float) (0.5 / sqrt(2.0));
    const orc_cf j2 = cf(0, c), r2 = cf(c, 0);
    for (int i = 0; i <= h; i++) {
      orc_cf X1 = (i == h) ? Xt[0] : Xt[i];
      orc_cf X2 = (i > 0) ? Xt[h - i] : Xt[0];
      orc_cf a = cmul(r2, cadd(X1, cconj(X2)));
      orc_cf b = cmul(cmul(j2, csub(X1, cconj(X2))), rot[i % n]);
      y[i] = csub(a, b);
    }
    orc_csym_force(y, n);
    free(x2); free(Xt); free(rot);
  } else {
    orc_cf *xc = (orc_cf *) malloc(sizeof(orc_cf) * (size_t) n);
    for (int i = 0; i < n; i++) xc[i] = cf(x[i], 0);
    orc_fft(xc, y, n, 1);
    free(xc);
  }
}

void orc_fftshift_c(const orc_cf *X, orc_cf *res, int n)   /* fourier.hpp:232-248 */
{
  if ((n & 1) == 0) {
    for (int i = 0; i < n / 2; i++) { res[n / 2 + i] = X[i]; res[i] = X[n / 2 + i]; }
  } else {
    for (int i = 0; i < 1 + n / 2; i++) res[n - (1 + n / 2) + i] = X[i];
    for (int i = 0; i < n / 2; i++) res[i] = X[n - n / 2 + i];
  }
}

/* ======================================================================================
 * Resampler
 * ==================================================================================== */
float orc_sinc2(float T, float f)              /* src/divers.cc:6-12 */
{
  float a = PI_F * T * f;
  if (fabsf(a) < 1e-7f) return T;
  return sinf(a) / (PI_F * f);
}

void orc_linspace(float a, float b, int n, float *x)   /* tsd.hpp:916-931 */
{
  if (n > 0) x[0] = a;
  if (n > 1) {
    double step = ((double) b - a) / (n - 1);
    for (int i = 1; i < n; i++) x[i] = (float) (a + step * i);
  }
}

/* InterpolateurSinc ctor + coefs_calcule, src/reechan/itrp.cc:24-54 (window "hn") */
void orc_itrp_sinc_lut(int K, int nphases, float fcut, float *lut)
{
  float *ls = (float *) malloc(sizeof(float) * (size_t) K);
  orc_linspace((float) (-K / 2), (float) ((K - 1) / 2), K, ls);
  for (int j = 0; j <= nphases; j++) {
    float tau = (float) ((1.0 * j) / nphases);
    for (int i = 0; i < K; i++) {
      float h = orc_sinc2(2 * fcut, (float) (i - K / 2) - tau);
      const float a = 0.5f, b = 0.25f;
      float t = (ls[i] - tau) * (float) (2 * PI_D / K);
      float r1 = cosf(t);
      float r2 = a + 2 * b * r1;
      lut[i + (size_t) j * K] = h * r2;
    }
  }
  free(ls);
}

void orc_ra_init(orc_ra *r, float ratio, int K, int nphases, const float *lut)
{                                              /* ra.cc:25-35 */
  memset(r, 0, sizeof(*r));
  r->ratio = ratio;
  r->increment = 1 / ratio;
  r->phase = 0;
  r->K = K; r->nphases = nphases; r->lut = lut;
}

void orc_ra_init_analytic(orc_ra *r, float ratio, int mode, int degree)
{
  orc_ra_init(r, ratio, mode == 1 ? 2 : degree + 1, 1, NULL);
  r->mode = mode;
}

/* coefs(tau): the table column (itrp.cc:16-22), InterpolateurLineaire::coefs (itrp.cc:83-86) or
 * InterpolateurLagrange::coefs (itrp.cc:112-132) */
static const float *ra_coefs(const orc_ra *r, float phase, float *buf)
{
  if (r->mode == 0) return r->lut + (size_t) ((int) (phase * r->nphases)) * r->K;
  if (r->mode == 1) {
    buf[0] = 1 - phase;
    buf[1] = phase;
    return buf;
  }
  const int d = r->K - 1;
  const float t = ((d - 1.0f) / 2) + phase;
  for (int j = 0; j <= d; j++) {
    float p = 1.0f;
    for (int k = 0; k <= d; k++)
      if (k != j) p *= (t - k) / (j - k);
    buf[j] = p;
  }
  return buf;
}

/* AdaptationRythmeSimple::step, ra.cc:39-77 with InterpolateurRIF::step
 * (filtrage.hpp:1873-1881) and InterpolateurSinc::coefs (itrp.cc:16-22) inlined. */
int64_t orc_ra_step_c(orc_ra *r, const orc_cf *x, int64_t n, orc_cf *y)
{
  int64_t j = 0;
  const int K = r->K;
  float phase = r->phase;
  const float inc = r->increment;
  for (int64_t i = 0; i < n; i++) {
    memmove(r->fen_c, r->fen_c + 1, sizeof(orc_cf) * (size_t) (K - 1));   /* :61 */
    r->fen_c[K - 1] = x[i];
    while (phase < 1) {
      float hbuf[256];
      const float *h = ra_coefs(r, phase, hbuf);
      orc_cf res = cf(0, 0);
      for (int t = 0; t < K; t++) {            /* res += h(i) * x((i+k)%K), k = 0 */
        res.re += h[t] * r->fen_c[t].re;
        res.im += h[t] * r->fen_c[t].im;
      }
      y[j++] = res;
      phase += inc;
    }
    phase--;
  }
  r->phase = phase;
  return j;
}

int64_t orc_ra_step_f(orc_ra *r, const float *x, int64_t n, float *y)
{
  int64_t j = 0;
  const int K = r->K;
  float phase = r->phase;
  const float inc = r->increment;
  for (int64_t i = 0; i < n; i++) {
    memmove(r->fen_f, r->fen_f + 1, sizeof(float) * (size_t) (K - 1));
    r->fen_f[K - 1] = x[i];
    while (phase < 1) {
      float hbuf[256];
      const float *h = ra_coefs(r, phase, hbuf);
      float res = 0;
      for (int t = 0; t < K; t++) res += h[t] * r->fen_f[t];
      y[j++] = res;
      phase += inc;
    }
    phase--;
  }
  r->phase = phase;
  return j;
}

int64_t orc_ra_schedule(orc_ra *r, int64_t n, int64_t *in_idx, int32_t *col, int64_t cap)
{
  int64_t j = 0;
  /* volatile keeps the additions in IEEE binary32 whatever the optimiser would like */
  volatile float phase = r->phase;
  const float inc = r->increment;
  for (int64_t i = 0; i < n; i++) {
    while (phase < 1) {
      if (j < cap) {
        if (in_idx) in_idx[j] = i;
        if (col) col[j] = (int32_t) (phase * r->nphases);
      }
      j++;
      phase = phase + inc;
    }
    phase = phase - 1;
  }
  r->phase = phase;
  return j;
}

void orc_reechan_config(float ratio, int *nb_decim, int *nb_ups, float *post, float *fcut)
{                                              /* ra.cc:104-149 */
  if ((ratio <= 0) || isinf(ratio) || (ratio >= 1e9f)) ratio = 1;
  float f = ratio;
  int nd = 0, nu = 0;
  while (f < 0.5) { nd++; f *= 2; }
  while (f >= 2) { nu++; f /= 2; }
  *nb_decim = nd; *nb_ups = nu; *post = f;
  float half = f / 2;
  *fcut = 0.4f < half ? 0.4f : half;
}

/* ======================================================================================
 * Polyphase stages used by filtre_reechan outside [0.5,2) -- src/reechan/polyphase.cc,
 * and Decimateur -- src/filtrage/filtre-rt.cc:127-169
 * ==================================================================================== */
/* Decimateur::step: y = x[cnt::R]; returns the number of outputs, updates *cnt. */
int64_t orc_decimateur_f(int R, int *cnt, const float *x, int64_t n, float *y)
{
  int64_t ny = (n + R - 1 - *cnt) / R, j = 0, i;
  for (i = *cnt; i < n; i += R) y[j++] = x[i];
  i -= R;
  *cnt = (int) (i - (n - R));
  return ny < j ? ny : j;
}

/* kind 0: FiltreRIFDecim (polyphase.cc:156-239)  -- taps forward against oldest->newest window
 * kind 1: FiltreRIFDemiBande (:54-149)           -- even taps only + 0.5 * centre sample, R = 2
 * state: fen[K] zero-initialised, *index, *cnt (the reference's cnt / odd).  float data.      */
int64_t orc_polydecim_f(int kind, const float *coefs, int K, int R, float *fen, int *index, int *cnt,
                        const float *x, int64_t n, float *y)
{
  int64_t no = 0;
  int idx = *index, c = *cnt;
  if (kind == 1) R = 2;
  for (int64_t j = 0; j < n; j++) {
    fen[idx] = x[j];
    idx = (idx + 1) % K;
    if (c < R - 1) { c++; continue; }
    c = 0;
    float somme = 0;
    if (kind == 0) {
      for (int i = 0; i < K; i++) somme += fen[(idx + i) % K] * coefs[i];
    } else {
      for (int i = 0; i < K; i += 2) somme += fen[(idx + i) % K] * coefs[i];
      somme += 0.5f * fen[(idx + K / 2) % K];
    }
    y[no++] = somme;
  }
  *index = idx; *cnt = c;
  return no;
}
int64_t orc_polydecim_c(int kind, const float *coefs, int K, int R, orc_cf *fen, int *index, int *cnt,
                        const orc_cf *x, int64_t n, orc_cf *y)
{
  int64_t no = 0;
  int idx = *index, c = *cnt;
  if (kind == 1) R = 2;
  for (int64_t j = 0; j < n; j++) {
    fen[idx] = x[j];
    idx = (idx + 1) % K;
    if (c < R - 1) { c++; continue; }
    c = 0;
    orc_cf somme = cf(0, 0);
    if (kind == 0) {
      for (int i = 0; i < K; i++) somme = cadd(somme, cscale(fen[(idx + i) % K], coefs[i]));
    } else {
      for (int i = 0; i < K; i += 2) somme = cadd(somme, cscale(fen[(idx + i) % K], coefs[i]));
      somme = cadd(somme, cscale(fen[(idx + K / 2) % K], 0.5f));
    }
    y[no++] = somme;
  }
  *index = idx; *cnt = c;
  return no;
}

/* FiltreRIFUps (polyphase.cc:246-341): coefs_in (Kin taps) are scaled by R and zero-padded to a
 * multiple of R by this function into coefs_pad (caller provides Kin + R floats); fen has
 * Kpad/R entries. returns n*R outputs. */
int orc_ups_prepare(const float *coefs_in, int Kin, int R, float *coefs_pad)
{
  int K = Kin;
  for (int i = 0; i < Kin; i++) coefs_pad[i] = coefs_in[i] * R;
  if (K % R) { int pad = R - (K % R); for (int i = 0; i < pad; i++) coefs_pad[K + i] = 0; K += pad; }
  return K;
}
int64_t orc_ups_f(const float *coefs, int K, int R, float *fen, int *index, const float *x, int64_t n, float *y)
{
  const int W = K / R;
  int idx = *index;
  int64_t no = 0;
  for (int64_t j = 0; j < n; j++) {
    fen[idx] = x[j];
    idx = (idx + 1) % W;
    for (int i = 0; i < R; i++) {
      float sum = 0;
      const float *cptr = coefs + (R - 1) - i;
      for (int t = 0; t < W; t++) { sum += fen[(idx + t) % W] * *cptr; cptr += R; }
      y[no++] = sum;
    }
  }
  *index = idx;
  return no;
}
int64_t orc_ups_c(const float *coefs, int K, int R, orc_cf *fen, int *index, const orc_cf *x, int64_t n, orc_cf *y)
{
  const int W = K / R;
  int idx = *index;
  int64_t no = 0;
  for (int64_t j = 0; j < n; j++) {
    fen[idx] = x[j];
    idx = (idx + 1) % W;
    for (int i = 0; i < R; i++) {
      orc_cf sum = cf(0, 0);
      const float *cptr = coefs + (R - 1) - i;
      for (int t = 0; t < W; t++) { sum = cadd(sum, cscale(fen[(idx + t) % W], *cptr)); cptr += R; }
      y[no++] = sum;
    }
  }
  *index = idx;
  return no;
}

/* ======================================================================================
 * Design helpers
 * ==================================================================================== */
void orc_design_rif_fen_hann(int n, int type, float fcut, float *h)
{
  /* window: fenêtre("hn", n, sym=oui) = Hamming_generalise(0.5, n, oui)
   * = a + (1-a) * cos(2*pi*fen_inter(n, oui)), fenetres.cc:16-60,127-130 */
  float *t = (float *) malloc(sizeof(float) * (size_t) n);
  float tmin = (float) (-n / 2), tmax = (float) (n / 2);   /* both parities when sym */
  orc_linspace(tmin / n, tmax / n, n, t);
  for (int i = 0; i < n; i++) {
    /* coefs_filtre_sinc, rif-fen.cc:31-41 */
    float k = (n & 1) ? (float) (i - n / 2) : (float) (i - (n - 1) / 2);
    float s = orc_sinc2(2 * fcut, k);
    if (type == 2) s = -s;                                   /* rif_fen_hp :44-50 */
    h[i] = s;
  }
  if (type == 2) h[(n - 1) / 2] += 1.0f;
  double somme = 0;
  for (int i = 0; i < n; i++) {
    float w = 0.5f + (1 - 0.5f) * cosf((float) (2 * PI_D) * t[i]);
    h[i] = h[i] * w;
    somme += h[i];                                           /* somme() accumulates in double */
  }
  if (type == 0) {
    float sf = (float) somme;
    for (int i = 0; i < n; i++) h[i] /= sf;                  /* only for the literal "lp", :96-98 */
  }
  free(t);
}

void orc_design_butter_lp(int n, float fcut, orc_cf *z, orc_cf *p, orc_cf *mlt_num, orc_cf *mlt_den)
{
  /* design_riia_laplace, rii.cc:405-444: wa = wd_vers_wa(2*pi*fcut, 1) (:20-23) */
  float wd = (float) (2 * PI_D * fcut);
  float wa = 2 * 1.0f * tanf(wd / (2 * 1.0f));
  /* butterworth_analogique, rii.cc:195-215 */
  for (int i = 0; i < n; i++) {
    float k = (float) (i + 1);
    float ang = ((float) PI_D * (2 * k + (float) (n - 1))) / (float) (2 * n);
    orc_cf pa = cf(cosf(ang), sinf(ang));
    /* pban_vers_pba, rii.cc:173-187: poles * wc */
    pa = cscale(pa, wa);
    /* trf_bilineaire with fe = 1, rii.cc:41-73: (p + 2) / (-p + 2) */
    p[i] = cdiv(cf(pa.re + 2, pa.im), cf(-pa.re + 2, -pa.im));
    z[i] = cf(-1, 0);                        /* numer *= from_roots(-ones(np-nz)) */
  }
  orc_cf num = cf((float) pow((double) wa, (double) n), 0);   /* mlt *= pow(wc, N - M) */
  orc_cf gain = cf(1, 0);
  for (int i = 0; i < n; i++) {
    float k = (float) (i + 1);
    float ang = ((float) PI_D * (2 * k + (float) (n - 1))) / (float) (2 * n);
    orc_cf pa = cscale(cf(cosf(ang), sinf(ang)), wa);
    gain = cdiv(gain, cf(2 - pa.re, -pa.im));                 /* gain /= (2*fe - pole) */
  }
  *mlt_num = cmul(num, gain);
  *mlt_den = cf(1, 0);
}
