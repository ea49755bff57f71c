/* tsdgpu.h -- C ABI of the MI355X-native streaming FIR / IIR(SOS) / FFT / resample path.
 *
 * This is the drop-in boundary: plain pointers and sizes, opaque handles, status codes.
 * libtsd (the reference) has no FFI; its plug points for this path are C++ virtual
 * interfaces and factory functions.  Each entry point below names the reference interface
 * it stands behind (paths relative to libtsd's core/ directory); the C++ adaptors in
 * libtsd_amd/host/ derive from those interfaces and forward here.
 *
 * Conventions
 *  - every function returns a tsdgpu_status; tsdgpu_last_error() gives the message of the
 *    last failure on the calling thread.
 *  - data pointers (x, y) may be DEVICE pointers (hipMalloc / torch) or plain HOST
 *    pointers; host buffers are staged through device memory by the call (H2D, kernel,
 *    D2H) and the call returns after the result is in the host buffer.  Coefficient
 *    pointers given to *_create are always HOST pointers and are copied.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls with
 *    device pointers are asynchronous on that stream.
 *  - handles are stateful exactly like the reference's filter objects (streaming: state
 *    is carried from one step to the next) and, like them, not thread-safe.
 *  - complex samples are interleaved (re, im) float32 pairs == std::complex<float>.
 */
#ifndef TSDGPU_H
#define TSDGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  TSDGPU_OK = 0,
  TSDGPU_ERR_INVALID = 1,      /* bad argument */
  TSDGPU_ERR_HIP = 2,          /* a HIP runtime call failed (no device, launch failure...) */
  TSDGPU_ERR_UNSUPPORTED = 3,  /* valid request this build cannot serve */
  TSDGPU_ERR_ALLOC = 4
} tsdgpu_status;

typedef enum { TSDGPU_F32 = 0, TSDGPU_C64 = 1 } tsdgpu_dtype;

const char *tsdgpu_last_error(void);
/* number of visible HIP devices (0 on a CPU-only host; never fails) */
int tsdgpu_device_count(void);
/* the calling thread's current HIP device (what allocations and handle creations land on); -1 without a device */
int tsdgpu_current_device(void);
/* "libtsd_amd x.y (gfx950)" */
const char *tsdgpu_version(void);

/* --------------------------------------------------------------------------------------
 * Device memory for RESIDENT vectors.  libtsd's array type can wrap foreign memory without owning
 * it (TabT::map, include/tsd/tableau.hpp:1067-1077; src/tableau.cc:723-739) and resize() to the
 * current size is a no-op (tableau.cc:702-705), so a Vecf/Veccf mapped on memory from
 * tsdgpu_malloc goes through every operator below without ever visiting the host.  tsdgpu_memcpy
 * copies in any direction (host<->device, device<->device); it returns once a host buffer
 * involved is safe to reuse.  tsdgpu_malloc_host gives page-locked host memory: host vectors
 * allocated from it are staged at the full PCIe rate and overlap their copies with the kernels
 * (the chunked pipeline of tsdgpu_fir_step / tsdgpu_sos_step on large host buffers).
 * ------------------------------------------------------------------------------------ */
int tsdgpu_malloc(void **out, size_t bytes);
int tsdgpu_free(void *p);
int tsdgpu_malloc_host(void **out, size_t bytes);
int tsdgpu_free_host(void *p);
int tsdgpu_memcpy(void *dst, const void *src, size_t bytes, void *stream);
int tsdgpu_memset(void *dev, int value, size_t bytes, void *stream);
int tsdgpu_synchronize(void *stream);
/* y[i] <- (Re y[i], 0) for n complex samples, device or host memory (what FiltreFFTRIF<cfloat> does to its
 * output, src/fourier/fourier.cc:976) */
int tsdgpu_zero_imag(void *y, int64_t n, void *stream);
/* Element-wise operations on RESIDENT vectors (device pointers only; n elements of data_type; dst may be a or b except
 * for REVERSE): what libtsd call sites put between two operators of the path -- Tab::reverse, *=, /=, + - *, abs, abs2,
 * real, imag, as_complex (include/tsd/tableau.hpp:883-901,1141-1257; src/tableau.cc:582-592,822-854,1243-1533).  Same IEEE
 * operations as the host loops (no contraction; complex quotients in double like libgcc).
 *   REVERSE dst[i] = a[n-1-i] | SCALE a*s | DIV_SCALAR a/s (s = s_re + i s_im, s_im ignored for F32) | ADD SUB MUL with b |
 *   NEG | ABS, ABS2 (dst float) | REAL, IMAG (C64 -> float) | TO_COMPLEX (F32 -> C64) | CONJ                                 */
typedef enum {
  TSDGPU_VEC_REVERSE = 0, TSDGPU_VEC_SCALE = 1, TSDGPU_VEC_DIV_SCALAR = 2, TSDGPU_VEC_ADD = 3, TSDGPU_VEC_SUB = 4, TSDGPU_VEC_MUL = 5,
  TSDGPU_VEC_NEG = 6, TSDGPU_VEC_ABS = 7, TSDGPU_VEC_ABS2 = 8, TSDGPU_VEC_REAL = 9, TSDGPU_VEC_IMAG = 10, TSDGPU_VEC_TO_COMPLEX = 11,
  TSDGPU_VEC_CONJ = 12
} tsdgpu_vec_opcode;
int tsdgpu_vec_op(int op, int data_type, void *dst, const void *a, const void *b, float s_re, float s_im, int64_t n, void *stream);
/* Reductions of a RESIDENT vector, returned to the host (one small copy): the sum (real and imaginary parts, accumulated in
 * double like Tab::somme, tableau.hpp:656-717) and, for F32 data, the largest / smallest value and the index of the first
 * largest one (valeur_max / valeur_min / index_max).  Any output pointer may be NULL.                                     */
int tsdgpu_vec_reduce(int data_type, const void *a, int64_t n, double *sum_re_im /* [2] */, float *max_min /* [2] */,
                      int64_t *arg_max, void *stream);
/* 1 when p is device or managed memory -- what every entry point treats as RESIDENT (no staging, asynchronous on the
 * caller's stream); page-locked / registered host memory counts as host memory (staged, but with asynchronous copies) */
int tsdgpu_is_device_pointer(const void *p);

/* --------------------------------------------------------------------------------------
 * FIR:  FiltreRIF<T,Tc>::step, factory filtre_rif<Tc,T>(coefs)
 *       (src/filtrage/filtre-rt.cc:53-109,171-175; include/tsd/filtrage.hpp:1367-1368)
 *       and the FFT-domain variant filtre_rif_fft<T> (src/fourier/fourier.cc:946-990).
 * y[n] = sum_k h[k] x[n-k], zero initial history, history carried across steps.
 * (data_type, tap_type) in {(F32,F32), (C64,F32), (C64,C64)} -- the reference's
 * instantiations (filtre-rt.cc:816-818).
 * ------------------------------------------------------------------------------------ */
typedef struct tsdgpu_fir tsdgpu_fir;
typedef enum {
  TSDGPU_FIR_AUTO = 0,          /* direct for short filters, overlap-save for long ones */
  TSDGPU_FIR_DIRECT = 1,        /* sliding dot product, same summation order as the reference */
  TSDGPU_FIR_OVERLAP_SAVE = 2   /* block FFT convolution; output aligned with DIRECT
                                   (no Nz-M delay, unlike the reference's OLA filter) */
} tsdgpu_fir_method;

int tsdgpu_fir_create(tsdgpu_fir **out, int data_type, int tap_type,
                      const void *taps_host, int ntaps, int method);
int tsdgpu_fir_step(tsdgpu_fir *f, const void *x, void *y, int64_t n, void *stream);
int tsdgpu_fir_reset(tsdgpu_fir *f);                     /* history <- zeros */
int tsdgpu_fir_reset_on(tsdgpu_fir *f, void *stream);    /* the same, ordered on `stream` (no host wait) */
/* History = the last ntaps-1 input samples (oldest first); this is the halo a multi-GPU
 * caller exchanges between neighbouring chunks.  dst/src: host or device pointers.      */
int tsdgpu_fir_get_history(tsdgpu_fir *f, void *dst, void *stream);
int tsdgpu_fir_set_history(tsdgpu_fir *f, const void *src, void *stream);
/* hipGraph capture of a streaming step.  By default a step alternates between two history buffers, so the
 * launch arguments change from one step to the next and a captured step would replay ONE step, not the
 * stream.  With capturable on, every step ends with the new history copied back to the first buffer (a
 * ~1 KB device copy): all launches of a step of fixed (x, y, n, stream) are then identical from step to
 * step, and the step can be captured once (hipStreamBeginCapture ... tsdgpu_fir_step ... EndCapture) and
 * replayed per block -- a few microseconds per small block instead of several launches' worth.  Run one
 * ordinary step of that size first (scratch buffers are allocated on first use, which capture forbids). */
/* The sharding hook without a copy: filters x[lead .. n) into y[lead .. n), taking the delay line from x[lead - L .. lead) itself
 * (L = tsdgpu_fir_lead(f) >= K - 1: the handle's history length, a whole number of 64-sample rows on the overlap-save plan; lead >= L).
 * Equivalent to tsdgpu_fir_set_history(f, x + lead - (K - 1)) followed by tsdgpu_fir_step(f, x + lead, y + lead, n - lead) -- what a
 * rank does with the halo-free interior of its chunk (libtsd: FiltreRIF::step on a chunk whose delay line holds the samples before
 * it, filtre-rt.cc:67-108) -- minus the history copy and its launch.  Device buffers, x != y, not on the partitioned plan
 * (more than 12289 taps): tsdgpu_fir_lead returns -1 there (and for a NULL handle), and the caller uses set_history + step. */
int tsdgpu_fir_step_after(tsdgpu_fir *f, const void *x, void *y, int64_t n, int64_t lead, void *stream);
int tsdgpu_fir_lead(const tsdgpu_fir *f);
int tsdgpu_fir_set_capturable(tsdgpu_fir *f, int on);
int tsdgpu_fir_method_used(const tsdgpu_fir *f);         /* DIRECT or OVERLAP_SAVE */
int tsdgpu_fir_destroy(tsdgpu_fir *f);

/* --------------------------------------------------------------------------------------
 * FFT:  FFTPlan::configure/step, TFRPlanDefaut, tfrplan_creation / fftplan_defaut hook
 *       (include/tsd/fourier.hpp:19-35; src/fourier/fourier.cc:61-121,360-486).
 * Unitary scaling 1/sqrt(n) in BOTH directions, natural-order output, any n >= 1
 * (power of two: Stockham passes; even: split recursion; odd: Bluestein).
 * `batch` transforms of length n laid out back to back (the reference has no batch API;
 * batch=1 is the FFTPlan::step equivalent).
 * hipGraph capture: a step on device buffers is stateless and may be captured (run one ordinary
 * step of the same batch first: scratch is allocated on first use); while the stream records, the
 * kernels with dynamic tile hand-out take their static partition (their counter base is a launch
 * argument and would be frozen).  The same holds for tsdgpu_resampler_step, whose captured launch
 * replays ONE step (stream position, window buffers and output count are host state of the handle).
 * ------------------------------------------------------------------------------------ */
typedef struct tsdgpu_fft tsdgpu_fft;
int tsdgpu_fft_create(tsdgpu_fft **out, int n, int batch_hint);
int tsdgpu_fft_step(tsdgpu_fft *p, const void *x, void *y, int batch, int forward, void *stream);
int tsdgpu_fft_size(const tsdgpu_fft *p);
int tsdgpu_fft_destroy(tsdgpu_fft *p);
/* Real FFT: RTFRPlan::step / rfft() / fft(Vecf) (src/fourier/fourier.cc:280-355,
 * include/tsd/fourier.hpp:99,116-122): n real samples -> the full n-bin complex spectrum.
 * Even n: n/2-point complex FFT of the packed pairs + untangling + forced conjugate symmetry,
 * all on the device; odd n: the complex FFT of x.as_complex(), like the reference.          */
typedef struct tsdgpu_rfft tsdgpu_rfft;
int tsdgpu_rfft_create(tsdgpu_rfft **out, int n);
int tsdgpu_rfft_step(tsdgpu_rfft *p, const void *x /* float[batch][n] */, void *y /* cfloat[batch][n] */, int batch,
                     void *stream);
int tsdgpu_rfft_destroy(tsdgpu_rfft *p);
/* --------------------------------------------------------------------------------------
 * OLA frequency-domain engine: OLA<cfloat>::step / step_interne behind filtre_fft()
 * (src/fourier/fourier.cc:737-940; include/tsd/fourier.hpp:305-370).  Blocks of Ne input
 * samples, N = nextpow2(Ne + min_zeros), frames = [Nz zeros | block]; per frame FFT ->
 * spectral processing -> inverse FFT -> overlap-add of the first Nz output samples onto the
 * previous tail.  window = NULL: simple OLA (one frame per block, Ne outputs per block);
 * window = Ne floats (the reference uses fenêtre("hn", Ne, false)): two half-overlapping
 * windowed frames per block, averaged; the very first block then yields no output
 * (cnt_ech < 0, fourier.cc:895-898).  Any call length: whole blocks are processed, the rest
 * waits in the handle (tampon_création semantics, fourier.cc:806-812).
 * The reference's user callback traitement_freq(X) has two device-side forms here:
 *  - built-in product X *= H (tsdgpu_ola_set_response) inside tsdgpu_ola_step -- nothing
 *    leaves the device (this is FiltreFFTRIF's own callback, fourier.cc:958);
 *  - tsdgpu_ola_analyse leaves the spectra of the new frames in a device buffer owned by the
 *    handle ([frames][N] complex); the caller edits them in place (own kernel, or copy out /
 *    copy back around a host callback) and tsdgpu_ola_synthese completes the step.
 * ------------------------------------------------------------------------------------ */
typedef struct tsdgpu_ola tsdgpu_ola;
int tsdgpu_ola_create(tsdgpu_ola **out, int block_len /* Ne; <= 0 -> 512 */, int min_zeros, const float *window);
int tsdgpu_ola_fft_size(const tsdgpu_ola *h);      /* N */
int tsdgpu_ola_block_len(const tsdgpu_ola *h);     /* Ne */
int tsdgpu_ola_set_response(tsdgpu_ola *h, const void *H /* N complex, host or device; NULL: none */);
int64_t tsdgpu_ola_max_out(const tsdgpu_ola *h, int64_t n);   /* bound on the outputs of a step of n inputs */
int tsdgpu_ola_step(tsdgpu_ola *h, const void *x, int64_t n, void *y, int64_t *n_out, void *stream);
int tsdgpu_ola_analyse(tsdgpu_ola *h, const void *x, int64_t n, void **spectra, int *frames, void *stream);
int tsdgpu_ola_synthese(tsdgpu_ola *h, void *y, int64_t *n_out, void *stream);
/* host-callback bridge: copy the pending spectra out of / back into the handle's device buffer
 * ([frames][N] complex, frames as returned by tsdgpu_ola_analyse); synchronous */
int tsdgpu_ola_apply_response(tsdgpu_ola *h, void *stream);   /* pending spectra *= H (no-op without a response) */
int tsdgpu_ola_read_spectra(tsdgpu_ola *h, void *host_dst, void *stream);
int tsdgpu_ola_write_spectra(tsdgpu_ola *h, const void *host_src, void *stream);
int tsdgpu_ola_destroy(tsdgpu_ola *h);

/* --------------------------------------------------------------------------------------
 * psd_welch (src/fourier/freqestim.cc:7-20): segments i = 0, N/2, N, ... while i + N < n of
 * x (cfloat), each multiplied by the window (N floats; the reference passes fenêtre(fen, N, non)),
 * S[k] = sum over the segments of |FFT_N(segment)|^2 (unitary FFT), fftshift-ed
 * (fourier.hpp:232-248).  S: N floats, linear power -- the caller applies pow2db.  Framing, ONE
 * batched FFT and the reduction over the segments run on the device; only x (if it is a host
 * buffer) and the N sums cross PCIe.  *n_segments (optional) receives the segment count.
 * ------------------------------------------------------------------------------------ */
int tsdgpu_welch(const void *x, int64_t n, int N, const float *window, float *S, int64_t *n_segments, void *stream);

/* fftshift (include/tsd/fourier.hpp:232-248): pure index permutation, bit-exact */
int tsdgpu_fftshift(const void *x, void *y, int n, int data_type, void *stream);

/* --------------------------------------------------------------------------------------
 * SOS IIR:  ChaineSOIS<T,T,T>::step over SOIS::step (DF2 / DF1) and RIIFoS::step,
 *           factory filtre_sois<T> (src/filtrage/filtre-rt.cc:303-602).
 * coefs: nsec rows of (b0,b1,b2,a1,a2) already normalised by a0; `gain` multiplies the
 * output when there is no first-order section; rii1 = (b0,b1,a1) or NULL.
 * First-call state seed: every section starts from y0=y1=(its first input sample)
 * (filtre-rt.cc:361-365).
 * ------------------------------------------------------------------------------------ */
typedef struct tsdgpu_sos tsdgpu_sos;
int tsdgpu_sos_create(tsdgpu_sos **out, int data_type, const float *coefs_host, int nsec,
                      float gain, const float *rii1_host, int forme);
int tsdgpu_sos_step(tsdgpu_sos *s, const void *x, void *y, int64_t n, void *stream);
int tsdgpu_sos_reset(tsdgpu_sos *s);
int tsdgpu_sos_reset_on(tsdgpu_sos *s, void *stream);
/* tsdgpu_sos_step whose first `skip` outputs are not stored (the sections run over all n samples; y[0 .. skip) is left untouched):
 * the halo-free interior of a sharded chunk warms up on the chunk's own first samples and writes from sample skip on, in ONE launch.
 * Distinct device buffers; skip * channels must be a multiple of 4 floats. */
int tsdgpu_sos_step_skip(tsdgpu_sos *s, const void *x, void *y, int64_t n, int64_t skip, void *stream);
int tsdgpu_sos_set_capturable(tsdgpu_sos *s, int on);   /* see tsdgpu_fir_set_capturable */   /* the same, ordered on `stream` (no host wait) */
/* number of warm-up samples a chunk needs before its first output for the carried state
 * to be exact to 2^-30 (multi-GPU halo size); -1 if the filter decays too slowly.       */
int64_t tsdgpu_sos_halo(const tsdgpu_sos *s);
/* The carried memories of the chain as a host vector of tsdgpu_sos_state_floats() floats: [0] = "the first sample has been
 * seen" (the seed of filtre-rt.cc:361-365 is spent), then per (section, channel) the reference's (d1, d2) -- FormeDirecte1:
 * (y1, y2, x1, x2) (:367-394); records are four floats wide, index 1 + (section * 2 + channel) * 4, and the slots a chain
 * does not use (the second channel of real data, the last two of a FormeDirecte2 section) are unspecified.  What a process-per-GPU sharding of a cascade with a LONG memory exchanges instead of a
 * warm-up halo (libtsd_amd/sharding.py, sos_step_exact): every rank but the first filters its chunk from zero memories
 * (state {1, 0, 0 ...}), the end states are all-gathered, rank r starts again from
 *     S_r = propagate(L_{r-1}, S_{r-1}, E_{r-1}),   S_1 = E_0
 * tsdgpu_sos_propagate_state: state_out = (the state `state_in` after n_samples of zero input) + end_state (NULL: none),
 * in double, host only; TSDGPU_ERR_UNSUPPORTED when the transition over n_samples leaves the float range.            */
int tsdgpu_sos_state_floats(void);
int tsdgpu_sos_get_state(tsdgpu_sos *s, float *state_host, void *stream);
int tsdgpu_sos_set_state(tsdgpu_sos *s, const float *state_host, void *stream);
int tsdgpu_sos_propagate_state(const tsdgpu_sos *s, int64_t n_samples, const float *state_in, const float *end_state,
                               float *state_out);
int tsdgpu_sos_destroy(tsdgpu_sos *s);

/* --------------------------------------------------------------------------------------
 * Resampler:  AdaptationRythmeSimple<T>::step (factory filtre_itrp) over
 *             InterpolateurRIF::step with the LUT-sinc interpolator itrp_sinc
 *             (src/reechan/ra.cc:13-79; include/tsd/filtrage.hpp:1873-1881;
 *             src/reechan/itrp.cc:10-55), as configured by filtre_reechan for a ratio in
 *             [0.5,2) (ra.cc:104-156).
 * lut_host: K x (nphases+1) float32, phase-major (lut[phase*K + i]).
 * The output count and the (input index, LUT column) of every output follow the
 * reference's float32 phase recurrence bit-exactly.
 * ------------------------------------------------------------------------------------ */
typedef struct tsdgpu_resampler tsdgpu_resampler;
int tsdgpu_resampler_create(tsdgpu_resampler **out, int data_type, float ratio,
                            const float *lut_host, int K, int nphases);
/* The interpolators whose taps are a function of the float phase itself (src/reechan/itrp.cc:80-133):
 * InterpolateurLineaire, coefs(tau) = {1 - tau, tau}, and InterpolateurLagrange of degree d
 * (d + 1 taps, evaluated at (d-1)/2 + tau).  Same recurrence, schedule and state as above. */
#define TSDGPU_ITRP_LINEAR 1
#define TSDGPU_ITRP_LAGRANGE 2
int tsdgpu_resampler_create_analytic(tsdgpu_resampler **out, int data_type, float ratio, int kind, int degree);
/* number of outputs the next step of n inputs will produce (advances nothing) */
int64_t tsdgpu_resampler_out_count(tsdgpu_resampler *r, int64_t n);
int tsdgpu_resampler_step(tsdgpu_resampler *r, const void *x, int64_t n,
                          void *y, int64_t y_capacity, int64_t *n_out, void *stream);
int tsdgpu_resampler_reset(tsdgpu_resampler *r);
/* jump the stream position: the next input is absolute sample `pos` of the stream
 * (phase and output offset follow the recurrence); window history <- `hist` (K-1 samples,
 * host or device, NULL = zeros).  This is the multi-GPU sharding hook.                   */
int tsdgpu_resampler_seek(tsdgpu_resampler *r, int64_t pos, const void *hist, void *stream);
int64_t tsdgpu_resampler_out_offset(const tsdgpu_resampler *r); /* outputs emitted before pos */
int tsdgpu_resampler_destroy(tsdgpu_resampler *r);

/* --------------------------------------------------------------------------------------
 * Integer-rate polyphase stages (what filtre_reechan chains for ratios outside [0.5,2)) and
 * the plain decimator:
 *   TSDGPU_POLY_DECIM     FiltreRIFDecim<T,float>     src/reechan/polyphase.cc:156-239
 *                         (taps applied in FORWARD order against the oldest->newest window)
 *   TSDGPU_POLY_HALFBAND  FiltreRIFDemiBande<T,float> polyphase.cc:54-149 (even taps + 0.5 centre, R = 2)
 *   TSDGPU_POLY_UPS       FiltreRIFUps<T,float>       polyphase.cc:246-341 (taps * R, R outputs per input)
 *   TSDGPU_POLY_PICK      Decimateur<T>               src/filtrage/filtre-rt.cc:127-169 (no taps)
 * ------------------------------------------------------------------------------------ */
typedef struct tsdgpu_polyfir tsdgpu_polyfir;
typedef enum { TSDGPU_POLY_DECIM = 0, TSDGPU_POLY_HALFBAND = 1, TSDGPU_POLY_UPS = 2, TSDGPU_POLY_PICK = 3 } tsdgpu_poly_kind;
int tsdgpu_polyfir_create(tsdgpu_polyfir **out, int kind, int data_type, const float *taps_host, int ntaps, int R);
int64_t tsdgpu_polyfir_out_count(tsdgpu_polyfir *p, int64_t n);
int tsdgpu_polyfir_step(tsdgpu_polyfir *p, const void *x, int64_t n, void *y, int64_t y_capacity,
                        int64_t *n_out, void *stream);
int tsdgpu_polyfir_reset(tsdgpu_polyfir *p);
int tsdgpu_polyfir_destroy(tsdgpu_polyfir *p);

/* --------------------------------------------------------------------------------------
 * Generic direct-form-I IIR: FiltreRII<T,float>::step, factory filtre_rii
 * (src/filtrage/filtre-rt.cc:177-289).  numer[Kx], denom[Kd] in powers of z^-1.
 * ------------------------------------------------------------------------------------ */
typedef struct tsdgpu_rii tsdgpu_rii;
int tsdgpu_rii_create(tsdgpu_rii **out, int data_type, const float *numer_host, int Kx,
                      const float *denom_host, int Kd);
/* coefficients of type coef_type (TSDGPU_F32, or TSDGPU_C64 = filtre_rii<cfloat,cfloat>, complex data only).
 * Real coefficients: the denominator is factored on the host and the recursion runs block-parallel as
 * zero-seeded sections on the SOS kernel whenever that cascade reproduces the direct form to 4e-6 on a
 * create-time check; complex coefficients: first-order complex sections (complex poles need no conjugate partner), same
 * check; otherwise -- poles outside the circle, memories beyond 2^20 samples, cascades that round too differently from the
 * direct form -- the literal sequential recursion runs.
 * tsdgpu_rii_path: 0 = sections only, 1 = FIR kernel + sections, 2 = FIR kernel + literal recursion,
 * 3 = FIR kernel + first-order complex sections. */
int tsdgpu_rii_create2(tsdgpu_rii **out, int data_type, int coef_type, const void *numer_host, int Kx,
                       const void *denom_host, int Kd);
int tsdgpu_rii_path(const tsdgpu_rii *r);
int tsdgpu_rii_step(tsdgpu_rii *r, const void *x, void *y, int64_t n, void *stream);
int tsdgpu_rii_destroy(tsdgpu_rii *r);

/* --------------------------------------------------------------------------------------
 * Pattern detector: Detecteur::step (src/fourier/detection.cc:100-400; include/tsd/fourier.hpp:545-660).
 * Per block of the stream: correlation with the unit-energy pattern (mode 0: the OLA engine with the
 * response conj(FFT(pattern)), blocks of exactly Ne samples; mode 1: a FIR with the reversed conjugated
 * pattern, any block length), |x|^2 through an M-tap moving average aligned with the correlator's delay,
 * the normalised score sqrt(N/M) |c| / sqrt(e), and the peak search -- all on the device.  A peak is a
 * score above the threshold that dominates the M-1 samples on either side; it is reported M samples
 * late (possibly in the following block: index < 0 then), so block borders need no special case.
 * peaks[k].index: position of the peak in the block's output vector; pattern start = index - delay.
 * scores: the block's n scores (host or device vector; NULL: not wanted).  One small D2H per block.
 * ------------------------------------------------------------------------------------ */
typedef struct tsdgpu_detector tsdgpu_detector;
typedef struct {
  int32_t index;                       /* relative to the start of the block being processed */
  float s_m1, s0, s_p1;                /* scores at index-1, index, index+1 */
  float c_m1[2], c0[2], c_p1[2];       /* complex correlation values there */
} tsdgpu_peak;
int tsdgpu_detector_create(tsdgpu_detector **out, const void *pattern_host /* M complex, unit energy */, int M, int Ne,
                           int mode, float threshold);
int tsdgpu_detector_delay(const tsdgpu_detector *d);      /* correlator delay: Ne (mode 0) or M-1 (mode 1) */
int tsdgpu_detector_fft_size(const tsdgpu_detector *d);   /* N of the OLA engine (1 in mode 1) */
int tsdgpu_detector_step(tsdgpu_detector *d, const void *x, int64_t n, float *scores, tsdgpu_peak *peaks, int max_peaks,
                         int *n_peaks, void *stream);
int tsdgpu_detector_destroy(tsdgpu_detector *d);

/* --------------------------------------------------------------------------------------
 * Cross-correlation and delay estimate on the device: xcorrb / xcorr (src/fourier/fourier.cc:489-597:
 * zero-padding to n + 2m, two forward FFTs in one batched call, correlation_freq = X0 conj(X1) sqrt(L)
 * with the lags reordered, one inverse FFT, lags -(m-1)..(m-1) scaled by 1/n and, unbiased, by
 * n / (n - |lag|)) and estimation_delais (src/fourier/estimation-delais.cc:9-14,100-118: |corr| over the
 * product of the two RMS values, arg max, quadratic interpolation).  x, y: n complex samples, host or
 * device (y NULL: autocorrelation); out: 2m-1 complex values.  tsdgpu_delay_estimate brings back two floats.
 * ------------------------------------------------------------------------------------ */
int tsdgpu_xcorr(const void *x, const void *y, int n, int m, int unbiased, void *out, void *stream);
int tsdgpu_delay_estimate(const void *x, const void *y, int n, float *delay, float *score, void *stream);

/* --------------------------------------------------------------------------------------
 * Real-time spectrum analyser: rt_spectrum / Spectrum::step (src/fourier/fourier.cc:1162-1342).
 * A block of BS = nsubs x Nf complex samples is cut in nsubs sub-blocks; each is multiplied by the window (Nf values,
 * already normalised to energy Nf by the caller, fourier.cc:1209-1212), transformed (unitary), |X|^2 taken in fftshift
 * order; the sub-blocks of nmeans consecutive blocks are summed -- all into one Nf-bin spectrum, or, in sweep mode, sub-block
 * i into bins [i step, i step + Nf) of a Ns = Nf + (nsubs - 1) step bin spectrum through a mask (Nf values; NULL: ones) and
 * divided by the number of contributions per bin -- scaled by 1 / (nmeans nsubs Nf) and returned as 10 log10(. + FLT_MIN).
 * tsdgpu_spectrum_step takes any number of whole blocks (x host or device); every nmeans-th block completes a spectrum, appended
 * to y (Ns floats each, host or device; n_spectra returns how many).  Sums of an incomplete group stay on the device between
 * calls.  Everything per-sample runs on the GPU: only x goes up, Ns floats come back per nmeans blocks.
 * ------------------------------------------------------------------------------------ */
typedef struct tsdgpu_spectrum tsdgpu_spectrum;
int tsdgpu_spectrum_create(tsdgpu_spectrum **out, int BS, int nsubs, int nmeans, const float *window_host, int sweep_active,
                           int sweep_step, const float *mask_host);
int tsdgpu_spectrum_bins(const tsdgpu_spectrum *h);                /* Ns */
int tsdgpu_spectrum_pending(const tsdgpu_spectrum *h);             /* blocks accumulated since the last spectrum */
int tsdgpu_spectrum_step(tsdgpu_spectrum *h, const void *x, int64_t nblocks, float *y, int64_t y_capacity, int64_t *n_spectra,
                         void *stream);
int tsdgpu_spectrum_reset(tsdgpu_spectrum *h, void *stream);
int tsdgpu_spectrum_destroy(tsdgpu_spectrum *h);

/* --------------------------------------------------------------------------------------
 * Several GPUs, ONE process: a long vector cut into contiguous chunks (shard g = samples
 * [n g / N, n (g+1) / N)), one operator handle per shard, and the one small left-neighbour halo each
 * operator needs -- FIR: K-1 input samples; SOS: the warm-up samples of tsdgpu_sos_halo; resampler:
 * its K-1-sample window and the absolute stream position.  The reference is single-threaded and has no
 * multi-device notion: this stands behind the SAME step() of FiltreRIF / ChaineSOIS /
 * AdaptationRythmeSimple (src/filtrage/filtre-rt.cc:53-109,440-572; src/reechan/ra.cc:13-79) for
 * vectors that are worth spreading over a node.  No collective, nothing exchanged but the halos.
 * (SOS cascades whose warm-up would exceed 2^16 samples, or that do not decay at all, are sharded EXACTLY instead: every
 * shard but the first runs from zero state, the end states -- a few floats per section -- meet on the host, the true start
 * states follow from the cascade's transition matrix over a shard, and the shards run again: tsdgpu_sharded_halo is 0.)
 * devices: nshards device ordinals (NULL: shard g on device g % device_count); several shards may name
 * the same device.  Calls are synchronous and keep the streaming contract (the tail of one call is the
 * halo of the next call's first shard).
 *   tsdgpu_sharded_step_host   x, y HOST vectors: each shard stages its chunk on its own thread and
 *                              stream; the halos are read from the host vector itself.
 *   tsdgpu_sharded_step_parts  x_parts[g] / y_parts[g] RESIDENT on device g (counts[g] samples): halos
 *                              move device to device (hipMemcpyPeerAsync over xGMI) on side streams while each
 *                              shard already filters the interior of its part; only the launch of a part's first
 *                              halo-length outputs waits for its halo.  For the resampler y_capacities[g] bounds
 *                              the outputs of shard g, out_counts[g] returns them.  The shards run on the handle's own
 *                              streams: this form first waits for ALL prior work on the shards' devices
 *                              (hipDeviceSynchronize) so that the parts have been produced;
 *   tsdgpu_sharded_step_parts_on  the same, ordered by events instead: producer_streams[g] is the stream whose work
 *                              produced x_parts[g] (NULL = the null stream); nothing else is waited for.
 * ------------------------------------------------------------------------------------ */
typedef struct tsdgpu_sharded tsdgpu_sharded;
int tsdgpu_fir_sharded_create(tsdgpu_sharded **out, int data_type, int tap_type, const void *taps_host, int ntaps, int method,
                              int nshards, const int *devices);
int tsdgpu_sos_sharded_create(tsdgpu_sharded **out, int data_type, const float *coefs_host, int nsec, float gain,
                              const float *rii1_host, int forme, int nshards, const int *devices);
int tsdgpu_resampler_sharded_create(tsdgpu_sharded **out, int data_type, float ratio, const float *lut_host, int K, int nphases,
                                    int nshards, const int *devices);
int tsdgpu_sharded_count(const tsdgpu_sharded *h);
int64_t tsdgpu_sharded_halo(const tsdgpu_sharded *h);                     /* halo length in samples */
int tsdgpu_sharded_device(const tsdgpu_sharded *h, int shard);
void tsdgpu_sharded_bounds(const tsdgpu_sharded *h, int64_t n, int shard, int64_t *lo, int64_t *hi);
int64_t tsdgpu_sharded_out_count(tsdgpu_sharded *h, int64_t n);           /* outputs of the next n inputs */
int tsdgpu_sharded_step_host(tsdgpu_sharded *h, const void *x, int64_t n, void *y, int64_t y_capacity, int64_t *n_out);
int tsdgpu_sharded_step_parts(tsdgpu_sharded *h, const void *const *x_parts, const int64_t *counts, void *const *y_parts,
                              const int64_t *y_capacities, int64_t *out_counts);
int tsdgpu_sharded_step_parts_on(tsdgpu_sharded *h, const void *const *x_parts, const int64_t *counts, void *const *y_parts,
                                 const int64_t *y_capacities, int64_t *out_counts, void *const *producer_streams);
int tsdgpu_sharded_reset(tsdgpu_sharded *h);
int tsdgpu_sharded_destroy(tsdgpu_sharded *h);

#ifdef __cplusplus
}
#endif
#endif
