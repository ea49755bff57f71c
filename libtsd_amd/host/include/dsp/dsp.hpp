// dsp/dsp.hpp -- English names of the hot-path API (libtsd core/include/dsp/*.hpp forwards
// every call to its French twin; SURVEY.md Appendix A lists the pairs).  Here the English
// API is the same objects under aliases, so dsp:: and tsd:: code can be mixed freely.
#pragma once
#include "tsd/tsd.hpp"
#include "tsd/filtrage.hpp"
#include "tsd/fourier.hpp"

namespace dsp {
using tsd::cfloat;
using tsd::sptr;
template <typename T> using Vector = tsd::Vecteur<T>;
using Vecf = tsd::Vecf;
using Veccf = tsd::Veccf;
template <typename Te, typename Ts = Te> using FilterGen = tsd::FiltreGen<Te, Ts>;   // dsp/dsp.hpp:462-472
template <typename Te, typename Ts = Te, typename Tc = tsd::Void> using Filter = tsd::Filtre<Te, Ts, Tc>;
using tsd::linspace;
// dsp::resample (dsp/dsp.hpp:499-503)
template <typename T> Vector<T> resample(const Vector<T> &x, float ratio) { return tsd::rééchan(x, ratio); }

namespace filter {
using tsd::filtrage::Design;
using tsd::filtrage::FRat;
using tsd::filtrage::RIIStructure;
using tsd::filtrage::FormeDirecte1;
using tsd::filtrage::FormeDirecte2;
inline Vecf window(const std::string &type, int n, bool symetrical = true) { return tsd::filtrage::fenêtre(type, n, symetrical); }
inline Vecf design_fir_wnd(int n, const std::string &type, float fc, const std::string &wnd = "hn", float fc2 = 0)
{ return tsd::filtrage::design_rif_fen(n, type, fc, wnd, fc2); }
inline FRat<cfloat> design_iira(int n, const std::string &type, const std::string &prototype, float fc, float δ_bp = 0.1f, float δ_bc = 60)
{ return tsd::filtrage::design_riia(n, type, prototype, fc, δ_bp, δ_bc); }
template <typename Tc, typename T = Tc> sptr<FilterGen<T>> filter_fir(const Vector<Tc> &h) { return tsd::filtrage::filtre_rif<Tc, T>(h); }
template <typename T> sptr<FilterGen<T>> filter_fir_fft(const Vecf &h) { return tsd::filtrage::filtre_rif_fft<T>(h); }
template <typename T> sptr<FilterGen<T>> filter_sois(const FRat<cfloat> &h, RIIStructure s = FormeDirecte2) { return tsd::filtrage::filtre_sois<T>(h, s); }
template <typename T> sptr<FilterGen<T>> filter_sois(const FRat<float> &h, RIIStructure s = FormeDirecte2) { return tsd::filtrage::filtre_sois<T>(h, s); }
template <typename T> sptr<Filter<T, T, float>> filter_resample(float ratio) { return tsd::filtrage::filtre_reechan<T>(ratio); }
// interpolators (dsp/filter.hpp:1762-1805): linear, Lagrange, cubic spline, windowed sinc
template <typename T> auto itrp_linear() { return tsd::filtrage::itrp_lineaire<T>(); }
template <typename T> auto itrp_lagrange(int degree) { return tsd::filtrage::itrp_lagrange<T>(degree); }
// dsp/filter.hpp:1128-1164,1288-1292,1354-1358,1407-1411,1578-1631,1827-1883,1910
using Frequency = tsd::filtrage::Fréquence;
inline float ema_coef(Frequency fc) { return tsd::filtrage::lexp_coef(fc); }
inline float ema_tc2coef(float tc) { return tsd::filtrage::lexp_tc_vers_coef(tc); }
inline float ema_coef2tc(float γ) { return tsd::filtrage::lexp_coef_vers_tc(γ); }
inline Frequency ema_fcut(float γ) { return tsd::filtrage::lexp_fcoupure(γ); }
template <typename T> sptr<FilterGen<T>> delay_line(unsigned int n) { return tsd::filtrage::ligne_a_retard<T>((int) n); }
template <typename T> sptr<FilterGen<T>> decimator(int R) { return tsd::filtrage::decimateur<T>(R); }
template <typename Tc, typename T = Tc> sptr<FilterGen<T>> filter_iir(const FRat<Tc> &h) { return tsd::filtrage::filtre_rii<Tc, T>(h); }
template <typename T> sptr<FilterGen<T>> filter_ema(float γ) { return tsd::filtrage::filtre_lexp<T>(γ); }
template <typename T> sptr<FilterGen<T>> filter_dc(float fc) { return tsd::filtrage::filtre_dc<T>(fc); }
template <typename T, typename Tacc> sptr<FilterGen<T>> filter_ma(unsigned int K) { return tsd::filtrage::filtre_mg<T, Tacc>((int) K); }
template <typename Tc, typename T = Tc> sptr<FilterGen<T>> filter_fir_decim(const Vector<Tc> &h, unsigned int R) { return tsd::filtrage::filtre_rif_decim<Tc, T>(h, (int) R); }
template <typename Tc, typename T = Tc> sptr<FilterGen<T>> filter_fir_half_band(const Vector<Tc> &c) { return tsd::filtrage::filtre_rif_demi_bande<Tc, T>(c); }
template <typename Tc, typename T = Tc> sptr<FilterGen<T>> filter_fir_ups(const Vector<Tc> &h, unsigned int R) { return tsd::filtrage::filtre_rif_ups<Tc, T>(h, (int) R); }
inline float filter_fir_ups_delay(int nc, int R) { return tsd::filtrage::filtre_rif_ups_délais(nc, R); }
template <typename T> Vector<T> filter(const Design &d, const Vector<T> &x) { return tsd::filtrage::filtrer<T>(d, x); }
template <typename T> Vector<T> filtfilt(const Design &d, const Vector<T> &x) { return tsd::filtrage::filtfilt<T>(d, x); }
template <typename T, typename Tc> Vector<T> convol(const Vector<Tc> &h, const Vector<T> &x) { return tsd::filtrage::convol<T, Tc>(h, x); }
}  // namespace filter

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
struct FFTFilterConfig : tsd::fourier::FiltreFFTConfig {
  int &time_blocks_length = dim_blocs_temporel;
  int &minimum_zeros_count = nb_zeros_min;
  bool &enable_windowing = avec_fenetrage;
  std::function<void(Veccf &)> &freq_domain_processing = traitement_freq;
  tsd::Veccf &frequency_response = réponse_freq;   // extension: device-side X *= H
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
