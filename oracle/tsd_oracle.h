/* tsd_oracle.h -- CPU restatement ("oracle") of libtsd's streaming FIR / IIR(SOS) /
 * FFT / resample hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (libtsd_amd) never
 * links, imports or calls anything under oracle/.
 *
 * Parity status: the reference itself is UNBUILDABLE in this image (its array runtime
 * core/src/tableau.cc, core/src/filtrage/frat.cc and core/src/fenetres.cc need Eigen3,
 * which is absent), so this restatement is pinned by the reference's own known-answer
 * tests (core/tests/test-filtres.cc, test-fourier.cc, test-ra.cc, test-tsd.cc -- see
 * tests/test_oracle_pins.py).  The resampler's sample-exact schedule is not pinned by any
 * reference test (statistical checks only, core/tests/test-ra.cc:126-143): for that
 * function the oracle is "parity unpinned" beyond those statistical checks.
 *
 * Every function cites the reference file:line whose arithmetic (and operation order) it
 * follows.  Paths are relative to /root/reference/core/.
 */
#ifndef TSD_ORACLE_H
#define TSD_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } orc_cf;

/* ---- FIR, src/filtrage/filtre-rt.cc:53-109 (FiltreRIF<T,Tc>::step) -------------------
 * fen = circular window of K samples (zero at construction, :64), *index = write cursor.
 * ff: T=float,Tc=float   cf: T=cfloat,Tc=float   cc: T=cfloat,Tc=cfloat (:816-818).      */
void orc_fir_ff(const float *coefs, int K, float *fen, int *index,
                const float *x, float *y, int64_t n);
void orc_fir_cf(const float *coefs, int K, orc_cf *fen, int *index,
                const orc_cf *x, orc_cf *y, int64_t n);
void orc_fir_cc(const orc_cf *coefs, int K, orc_cf *fen, int *index,
                const orc_cf *x, orc_cf *y, int64_t n);

/* ---- generic direct-form-I IIR, src/filtrage/filtre-rt.cc:177-289 (FiltreRII) ---------
 * numer[Kx], denom[Ky+1] in powers of z^-1; wndx[Kx], wndy[Ky] zero-initialised.          */
void orc_rii_f(const float *numer, int Kx, const float *denom, int Ky,
               float *wndx, float *wndy, int *index, int *index_y,
               const float *x, float *y, int64_t n);
/* FiltreRII<cfloat,cfloat>: complex coefficients on complex data, interleaved (re, im) */
void orc_rii_c(const float *numer, int Kx, const float *denom, int Ky,
               float *wndx, float *wndy, int *index, int *index_y,
               const float *x, float *y, int64_t n);

/* ---- second-order sections, src/filtrage/filtre-rt.cc:303-400,407-437,440-572 --------- */
typedef struct {
  float b0, b1, b2, a1, a2;   /* normalised by a0 (:317-328) */
} orc_biquad;

typedef struct {
  int   nsec;                 /* number of biquads */
  orc_biquad sec[32];
  int   avec_rii1;            /* odd order: trailing 1st-order section (:530-556) */
  float r_b0, r_b1, r_a1;
  float gain;                 /* applied only when !avec_rii1 (:558-559,570) */
  int   forme;                /* 2 = FormeDirecte2 (default), 1 = FormeDirecte1 */
} orc_sos;

/* ChaineSOIS constructor: greedy conjugate pairing of zeros/poles (:448-560).
 * z,p = n roots each; mlt_num/mlt_den = leading multipliers of numerator/denominator.     */
int orc_sos_from_zpk(orc_sos *s, const orc_cf *z, const orc_cf *p, int n,
                     orc_cf mlt_num, orc_cf mlt_den, int forme);

/* per-chain state: for each section {x1,x2,y0,y1,y2,premier_appel}; then rii1 {x1,y1}.   */
typedef struct { float x1, x2, y0, y1, y2; int premier_appel; } orc_biquad_state_f;
typedef struct { orc_cf x1, x2, y0, y1, y2; int premier_appel; } orc_biquad_state_c;
typedef struct { orc_biquad_state_f sec[32]; float r_x1, r_y1; } orc_sos_state_f;
typedef struct { orc_biquad_state_c sec[32]; orc_cf r_x1, r_y1; } orc_sos_state_c;
void orc_sos_state_init_f(orc_sos_state_f *st);
void orc_sos_state_init_c(orc_sos_state_c *st);

/* ChaineSOIS::step (:562-571) over SOIS::step (:347-397) and RIIFoS::step (:422-436).     */
void orc_sos_step_f(const orc_sos *s, orc_sos_state_f *st, const float *x, float *y, int64_t n);
void orc_sos_step_c(const orc_sos *s, orc_sos_state_c *st, const orc_cf *x, orc_cf *y, int64_t n);
/* the same chain in double precision, one shot: conditioning yardstick only (not a reference path) */
void orc_sos_run_f64(const orc_sos *s, const float *x, double *y, int64_t n);

/* ---- FFT, src/fourier/fourier.cc:32-46,61-121,237-278,360-467 --------------------------
 * TFRPlanDefaut: pow2 -> radix-2 Stockham; even -> split recursion; odd -> Bluestein.
 * Unitary scaling (1/sqrt n) in both directions.  avant!=0: forward.                      */
void orc_fft(const orc_cf *x, orc_cf *y, int n, int avant);
void orc_fft_twiddles(orc_cf *rot, int n);                  /* fourier.cc:32-46 */
void orc_rfft(const float *x, orc_cf *y, int n);            /* RTFRPlan::step, fourier.cc:311-354 */
void orc_fftshift_c(const orc_cf *x, orc_cf *y, int n);     /* include/tsd/fourier.hpp:232-248 */
void orc_csym_force(orc_cf *x, int n);                      /* include/tsd/fourier.hpp:264-282 */
int  orc_next_pow2(int i);                                  /* src/tsd.cc:287-291 */

/* ---- resampler, src/reechan/ra.cc:13-79, include/tsd/filtrage.hpp:1873-1881,
 *      src/reechan/itrp.cc:10-55 ---------------------------------------------------------- */
/* LUT[K x (nphases+1)], column-major like Tabf (element (i,j) at lut[i + j*K]).            */
void orc_itrp_sinc_lut(int K, int nphases, float fcut, float *lut);
typedef struct {
  float phase, ratio, increment;      /* ra.cc:16-19 */
  int   K, nphases;
  const float *lut;
  float  fen_f[256];
  orc_cf fen_c[256];
  int   mode;                         /* 0: table (sinc / cspline); 1: InterpolateurLineaire; 2: InterpolateurLagrange */
} orc_ra;
void orc_ra_init(orc_ra *r, float ratio, int K, int nphases, const float *lut);
/* analytic interpolators, src/reechan/itrp.cc:80-133: mode 1 = linear (K = 2), mode 2 = Lagrange of degree d (K = d+1) */
void orc_ra_init_analytic(orc_ra *r, float ratio, int mode, int degree);
/* AdaptationRythmeSimple::step. y must hold ceil(ratio*n)+10 samples. returns n_out.      */
int64_t orc_ra_step_c(orc_ra *r, const orc_cf *x, int64_t n, orc_cf *y);
int64_t orc_ra_step_f(orc_ra *r, const float *x, int64_t n, float *y);
/* phase schedule only (no data): per output j -> input index and LUT column. Either
 * pointer may be NULL. Starts from the state in r and advances it. returns n_out.         */
int64_t orc_ra_schedule(orc_ra *r, int64_t n, int64_t *in_idx, int32_t *col, int64_t cap);
/* AdaptationRythmeArbitraire::configure_impl (ra.cc:104-156): folds ratio into [0.5,2).    */
void orc_reechan_config(float ratio, int *nb_decim, int *nb_ups, float *post, float *fcut);

/* ---- polyphase stages (src/reechan/polyphase.cc) and Decimateur (filtre-rt.cc:127-169) ---- */
int64_t orc_decimateur_f(int R, int *cnt, const float *x, int64_t n, float *y);
/* kind 0 = FiltreRIFDecim (:156-239), 1 = FiltreRIFDemiBande (:54-149, R forced to 2) */
int64_t orc_polydecim_f(int kind, const float *coefs, int K, int R, float *fen, int *index, int *cnt,
                        const float *x, int64_t n, float *y);
int64_t orc_polydecim_c(int kind, const float *coefs, int K, int R, orc_cf *fen, int *index, int *cnt,
                        const orc_cf *x, int64_t n, orc_cf *y);
/* FiltreRIFUps (:246-341): prepare scales by R and pads to a multiple of R, returns padded K */
int orc_ups_prepare(const float *coefs_in, int Kin, int R, float *coefs_pad);
int64_t orc_ups_f(const float *coefs, int K, int R, float *fen, int *index, const float *x, int64_t n, float *y);
int64_t orc_ups_c(const float *coefs, int K, int R, orc_cf *fen, int *index, const orc_cf *x, int64_t n, orc_cf *y);

/* ---- design helpers (host-side, run once) ---------------------------------------------- */
/* design_rif_fen(n,type,fcut,"hn") src/filtrage/rif-fen.cc:31-108 + fenetres.cc:16-60,127-130
 * type: 0 = "lp" (normalised), 1 = "pb" (not normalised), 2 = "hp".                        */
void orc_design_rif_fen_hann(int n, int type, float fcut, float *h);
/* design_riia(n,"lp","butt",fcut): src/filtrage/rii.cc:195-215,173-187,41-73,405-452.
 * Outputs n zeros, n poles (z-plane) and numerator / denominator multipliers.              */
void orc_design_butter_lp(int n, float fcut, orc_cf *z, orc_cf *p, orc_cf *mlt_num, orc_cf *mlt_den);
float orc_sinc2(float T, float f);                          /* src/divers.cc:6-12 */
void  orc_linspace(float a, float b, int n, float *x);      /* include/tsd/tsd.hpp:916-931 */

#ifdef __cplusplus
}
#endif
#endif
