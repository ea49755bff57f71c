// tsd/fourier.hpp -- host-side mirror of libtsd's FFT API for the hot path (namespace
// tsd::fourier), backed by the MI355X C ABI.  Same names / meaning as
//   core/include/tsd/fourier.hpp:19-35 (FFTPlan, fftplan_defaut), :69,:99 (tfrplan_création,
//   rtfrplan_création), :116-205 (rfft, fft, ifft), :232-282 (fftshift, csym_forçage).
// Unitary scaling 1/sqrt(n) in both directions, whatever `normalize` says -- that is what
// TFRPlanDefaut does (fourier.cc:362,372-376,120).
#pragma once
#include "tsd/tsd.hpp"
#include "tsd/filtrage.hpp"
#include <cstdint>
#include <tuple>

namespace tsd::fourier {

struct FFTPlan {
  virtual ~FFTPlan() {}
  virtual void configure(entier n, bouléen avant, bouléen normalize = true) = 0;
  virtual void step(const Veccf &x, Veccf &y, bouléen avant = true) = 0;
  Veccf step(const Veccf &x, bouléen avant = true)
  {
    Veccf y;
    step(x, y, avant);
    return y;
  }
};

// The run-time plug point: every fft()/ifft() asks this factory for its plan
// (fourier.cc:469-481).  Default: the MI355X plan.
extern fonction<sptr<FFTPlan>()> fftplan_defaut;
sptr<FFTPlan> tfrplan_création(entier n = -1, bouléen avant = true, bouléen normalize = true);
sptr<FiltreGen<float, cfloat>> rtfrplan_création(entier n = -1);

void csym_forçage_impl(Veccf &X);
template <typename T> void csym_forçage(Vecteur<T> &X)
{
  if constexpr (est_complexe<T>()) csym_forçage_impl(X);
}

template <typename T> Veccf fft(const Vecteur<T> &x)
{
  if constexpr (est_complexe<T>()) {
    return tfrplan_création()->step(x, true);
  } else {
    // real input goes through the real-FFT plan like the reference (fourier.hpp:163-170)
    return rtfrplan_création()->step(x);
  }
}
template <typename T> Veccf ifft(const Vecteur<T> &X)
{
  if constexpr (est_complexe<T>())
    return tfrplan_création()->step(X, false);
  else
    return tfrplan_création()->step(X.as_complex(), false);
}
inline Veccf rfft(const Vecf &x) { return rtfrplan_création()->step(x); }

// fftshift (fourier.hpp:232-248): pure index permutation
template <typename T> Vecteur<T> fftshift(const Vecteur<T> &X)
{
  const entier n = X.rows();
  Vecteur<T> res = Vecteur<T>::zeros(n);
  if ((n & 1) == 0) {
    res.tail(n / 2) = X.head(n / 2);
    res.head(n / 2) = X.tail(n / 2);
  } else {
    res.tail(1 + n / 2) = X.head(1 + n / 2);
    res.head(n / 2) = X.tail(n / 2);
  }
  return res;
}

// ---- "next" rows of the scope: thin compositions over the FFT plan ----
// correlations (fourier.cc:489-597): circular, biased and unbiased cross-correlation by FFT
std::tuple<Vecf, Veccf> ccorr(const Veccf &x0, const Veccf &x1 = Veccf());
std::tuple<Vecf, Veccf> xcorrb(const Veccf &x, const Veccf &y = Veccf(), entier m = -1);
std::tuple<Vecf, Veccf> xcorr(const Veccf &x, const Veccf &y = Veccf(), entier m = -1);
// rééchan_freq (fourier.cc:1391-1419): resampling by zero-padding / truncating the spectrum
Vecf rééchan_freq(const Vecf &x, float lom);
// czt (fourier.hpp:424, fourier.cc:1347-1389): the reference's chirp-z evaluation -- its chirp sequence, pre-multiplication,
// fft(hc) * fft(gc), ifft and post-division statement for statement, the transforms on the GPU plan.  As in the reference the
// two sequences it multiplies have m + n - 1 and 2m - 1 points, so only n == m is served (anything else fails there too).
Veccf czt(const Veccf &x, entier m, cfloat W, cfloat z0 = 1.0f);
// filtre_fft (fourier.cc:737-940; include/tsd/fourier.hpp:304-366): frequency-domain block
// processing by overlap-add with a user callback on every spectrum.  Returns the filter and
// the FFT size N.  Blocks of Ne = dim_blocs_temporel inputs (512 if <= 0), N = pp2(Ne +
// nb_zeros_min); with avec_fenetrage the blocks overlap by 1/2 under a Hann window (and the
// output is then Ne/2 samples late: the first half block is dropped).
// MI355X mapping: the FFTs of ALL the blocks a step() call completes run as one batched GPU
// transform each way; the callback (host code) sees the spectra one by one, in order.
struct FiltreFFTConfig {
  entier dim_blocs_temporel = 0;
  entier nb_zeros_min = 0;
  bouléen avec_fenetrage = false;
  fonction<void(Veccf &)> traitement_freq;
};   // (the device-side response X *= H is an extension: tsd_amd/extensions.hpp, filtre_fft_reponse)
std::tuple<sptr<Filtre<cfloat, cfloat, FiltreFFTConfig>>, entier> filtre_fft(const FiltreFFTConfig &config);
// cost model of the OLA engine (fourier.cc:700-735): flops per input sample, FFT size, zeros
void ola_complexité(entier M, entier Ne, float &C, entier &Nf, entier &Nz);
void ola_complexité_optimise(entier M, float &C, entier &Nf, entier &Nz, entier &Ne);
// spectral estimates (include/tsd/fourier.hpp:700-760, freqestim.cc:7-93): Hann-windowed
// periodogram in dB, and Welch's average over half-overlapping segments of N samples (all the
// segments of a call are transformed by ONE batched GPU FFT)
Vecf tfd_freqs(entier n, bouléen avec_shift = true);
Vecf psd_freqs(entier n, bouléen complexe = true);
template <typename T> std::tuple<Vecf, Vecf> psd(const Vecteur<T> &x);
std::tuple<Vecf, Vecf> psd_welch(const Veccf &x, entier N, cstring fen = "hn");
// délais (fourier.cc:607-698): integer delays shift (zero fill); fractional ones modulate the
// spectrum of the vector zero-padded to twice its length
template <typename T> Vecteur<T> délais(const Vecteur<T> &x, float τ);
// estimation_délais / aligne_entier (estimation-delais.cc:21-170): peak of the normalised biased
// cross-correlation with quadratic interpolation -> (delay, score); integer alignment of two vectors
std::tuple<float, float> estimation_délais(const Veccf &x, const Veccf &y);
template <typename T> std::tuple<Vecteur<T>, Vecteur<T>, entier, float> aligne_entier(const Vecteur<T> &x, const Vecteur<T> &y);

// ---- real-time spectrum (include/tsd/fourier.hpp:909-952; src/fourier/fourier.cc:1148-1342) ------
// Blocks of BS samples are cut in nsubs sub-blocks of Nf = BS / nsubs, windowed (window normalised
// to energy Nf), transformed -- ONE batched GPU FFT per block -- and |X|^2 (fftshift-ed) is averaged
// over nmeans blocks; the spectrum in dB comes out with every nmeans-th block (an empty vector
// otherwise).  sweep: the sub-blocks are successive tunings `step` bins apart, accumulated side by
// side into a Ns-bin spectrum with optional masks on the band edges / centre.
struct SpectrumConfig {
  entier BS = 1024, nmeans = 10, nsubs = 1;
  struct {
    bouléen active = false;
    entier step = 1024, masque_bf = 0, masque_hf = 0;
  } sweep;
  entier Nf() const { return BS / nsubs; }
  entier Ns() const { return sweep.active ? Nf() + (nsubs - 1) * sweep.step : Nf(); }
  tsd::filtrage::Fenetre fenetre = tsd::filtrage::Fenetre::HANN;
  sptr<FFTPlan> plan;              // accepted for compatibility; the batched GPU plan is always used
};
sptr<Filtre<cfloat, float, SpectrumConfig>> rt_spectrum(const SpectrumConfig &config);

// ---- pattern detector (include/tsd/fourier.hpp:545-660; src/fourier/detection.cc) ----------------
// Normalised correlation of the stream with a fixed pattern: the correlation runs on the GPU (the
// OLA engine above with X *= conj(FFT(pattern)) as spectral processing, or an FIR with the reversed
// conjugated pattern), the energy normalisation on the GPU moving average, the peak logic on the
// host like the reference.  position = first sample of the pattern relative to the start of the
// current block (negative: in an earlier block); position_prec adds the quadratic sub-sample
// interpolation; gain / θ = complex amplitude of the received pattern; SNR from the residual.
struct Detection {
  entier position = 0;
  float position_prec = 0, score = 0, gain = 0, θ = 0, SNR_dB = 0, σ_noise = 0;
};
struct DetecteurConfig {
  uint32_t Ne = 0;                 // block length (0: chosen by ola_complexité_optimise)
  Veccf motif;
  float seuil = 0.5f;
  bouléen debug_actif = false;
  enum Mode { MODE_OLA = 0, MODE_RIF = 1 } mode = MODE_OLA;
  fonction<void(const Detection &det)> gere_detection;
  bouléen calculer_signal_correlation = false;
};
struct Detecteur : Filtre<cfloat, float, DetecteurConfig> {};
sptr<Detecteur> détecteur_création(const DetecteurConfig &config);

}  // namespace tsd::fourier
