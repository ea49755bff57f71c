// dsp/fourier.hpp -- English names of the Fourier part of the hot path (libtsd
// core/include/dsp/fourier.hpp: FFTPlan / fftplan_defaut :22-24, fftplan_new :61-64, rfftplan_new
// :94-97, fft / ifft / rfft :113-194, fftshift / force_csym :223-246, resample_freq :140-143,
// filter_fft :252-318 and the correlation / spectrum / detector names further down).
#pragma once
#include "dsp/dsp.hpp"
#include "dsp/filter.hpp"
#include "tsd/fourier.hpp"

namespace dsp {
namespace fourier {
using tsd::fourier::FFTPlan;
using tsd::fourier::fftplan_defaut;
inline sptr<FFTPlan> fftplan_new(int n = -1, bool forward = true, bool normalize = true) { return tsd::fourier::tfrplan_création(n, forward, normalize); }
inline sptr<FilterGen<float, cfloat>> rfftplan_new(int n = -1) { return tsd::fourier::rtfrplan_création(n); }
template <typename T> Veccf fft(const Vector<T> &x) { return tsd::fourier::fft(x); }
template <typename T> Veccf ifft(const Vector<T> &X) { return tsd::fourier::ifft(X); }
inline Veccf rfft(const Vecf &x) { return tsd::fourier::rfft(x); }
template <typename T> Vector<T> fftshift(const Vector<T> &X) { return tsd::fourier::fftshift(X); }
template <typename T> void force_csym(Vector<T> &X) { tsd::fourier::csym_forçage(X); }
// "next" rows (dsp/fourier.hpp:140-143,251-355,397-457,488-505,623-672): same objects, English names
inline Vecf resample_freq(const Vecf &x, float ratio) { return tsd::fourier::rééchan_freq(x, ratio); }
inline Veccf czt(const Veccf &x, int m, cfloat W, cfloat z0 = 1.0f) { return tsd::fourier::czt(x, m, W, z0); }        // dsp/fourier.hpp: czt
struct FFTFilterConfig : tsd::fourier::FiltreFFTConfig {
  int &time_blocks_length = dim_blocs_temporel;
  int &minimum_zeros_count = nb_zeros_min;
  bool &enable_windowing = avec_fenetrage;
  std::function<void(Veccf &)> &freq_domain_processing = traitement_freq;
};
inline std::tuple<sptr<Filter<cfloat, cfloat, tsd::fourier::FiltreFFTConfig>>, int> filter_fft(const FFTFilterConfig &config)
{ return tsd::fourier::filtre_fft(config); }
inline void ola_complexity(int M, int Ne, float &C, int &Nf, int &Nz) { tsd::fourier::ola_complexité(M, Ne, C, Nf, Nz); }
inline void ola_complexity_optimize(int M, float &C, int &Nf, int &Nz, int &Ne) { tsd::fourier::ola_complexité_optimise(M, C, Nf, Nz, Ne); }
inline auto ccorr(const Veccf &x, const Veccf &y = Veccf()) { return tsd::fourier::ccorr(x, y); }
inline auto xcorr(const Veccf &x, const Veccf &y = Veccf(), int m = -1) { return tsd::fourier::xcorr(x, y, m); }
inline auto xcorrb(const Veccf &x, const Veccf &y = Veccf(), int m = -1) { return tsd::fourier::xcorrb(x, y, m); }
template <typename T> Vector<T> delay(const Vector<T> &x, float τ) { return tsd::fourier::délais(x, τ); }
inline std::tuple<float, float> delay_estimation(const Veccf &x, const Veccf &y) { return tsd::fourier::estimation_délais(x, y); }
template <typename T> std::tuple<Vector<T>, Vector<T>, int, float> align_int(const Vector<T> &x, const Vector<T> &y) { return tsd::fourier::aligne_entier(x, y); }
inline Vecf psd_freqs(int n, bool complexe = true) { return tsd::fourier::psd_freqs(n, complexe); }
template <typename T> std::tuple<Vecf, Vecf> psd(const Vector<T> &x) { return tsd::fourier::psd(x); }
inline std::tuple<Vecf, Vecf> psd_welch(const Veccf &x, int N, const std::string &fen = "hn") { return tsd::fourier::psd_welch(x, N, fen); }
// real-time spectrum (dsp/fourier.hpp:815-828)
using tsd::fourier::SpectrumConfig;
inline sptr<Filter<cfloat, float, SpectrumConfig>> rt_spectrum(const SpectrumConfig &config) { return tsd::fourier::rt_spectrum(config); }
// pattern detector (dsp/fourier.hpp:505-583)
using tsd::fourier::Detection;
using tsd::fourier::Detecteur;
struct DetectorConfig : tsd::fourier::DetecteurConfig {
  uint32_t &Ns = Ne;
  tsd::Veccf &pattern = motif;
  float &threshold = seuil;
  bool &debug_active = debug_actif;
  std::function<void(const Detection &det)> &on_detection = gere_detection;
  bool &compute_correlation_signal = calculer_signal_correlation;
};
inline sptr<Detecteur> detector_new(const DetectorConfig &config) { return tsd::fourier::détecteur_création(config); }
}  // namespace fourier
}  // namespace dsp
