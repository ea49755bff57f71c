// dsp/filter.hpp -- English names of the filtering part of the hot path (libtsd
// core/include/dsp/filter.hpp: filter_fir :1333-1337, filter_fir_fft :1377-1381, filter_iir :1407-1411,
// filter_sois :1521,1548-1559, filter / filtfilt / convol :1662-1701, the rate-changing stages
// :1827-1913, design_fir_wnd :749-752, design_iira :601-605, window :79-104).  One-line forwarders to
// tsd::filtrage, like the reference's own English layer (SURVEY.md Appendix A lists the pairs).
#pragma once
#include "dsp/dsp.hpp"
#include "tsd/filtrage.hpp"

namespace dsp {
namespace filter {
using tsd::filtrage::Design;
using tsd::FRat;
using tsd::filtrage::RIIStructure;
using tsd::filtrage::FormeDirecte1;
using tsd::filtrage::FormeDirecte2;
inline Vecf window(const std::string &type, int n, bool symetrical = true) { return tsd::filtrage::fenêtre(type, n, symetrical); }
inline Vecf design_fir_wnd(int n, const std::string &type, float fc, const std::string &wnd = "hn", float fc2 = 0)
{ return tsd::filtrage::design_rif_fen(n, type, fc, wnd, fc2); }
inline FRat<cfloat> design_iira(int n, const std::string &type, const std::string &prototype, float fc, float δ_bp = 0.1f, float δ_bc = 60)
{ return tsd::filtrage::design_riia(n, type, prototype, fc, δ_bp, δ_bc); }
// dsp/filter.hpp:497-543
using BiquadSpec = tsd::filtrage::BiquadSpec;
inline FRat<float> design_biquad(const std::string &type, float f, float Q, float gain_dB = 0) { return tsd::filtrage::design_biquad(type, f, Q, gain_dB); }
inline FRat<float> design_biquad(const BiquadSpec &spec) { return tsd::filtrage::design_biquad(spec); }
template <typename Tc, typename T = Tc> sptr<FilterGen<T>> filter_fir(const Vector<Tc> &h) { return tsd::filtrage::filtre_rif<Tc, T>(h); }
template <typename T> sptr<FilterGen<T>> filter_fir_fft(const Vecf &h) { return tsd::filtrage::filtre_rif_fft<T>(h); }
template <typename T> sptr<FilterGen<T>> filter_sois(const FRat<cfloat> &h, RIIStructure s = FormeDirecte2) { return tsd::filtrage::filtre_sois<T>(h, s); }
template <typename T> sptr<FilterGen<T>> filter_sois(const FRat<float> &h, RIIStructure s = FormeDirecte2) { return tsd::filtrage::filtre_sois<T>(h, s); }
template <typename T> sptr<Filter<T, T, float>> filter_resample(float ratio) { return tsd::filtrage::filtre_reechan<T>(ratio); }
// interpolators (dsp/filter.hpp:1762-1805): linear, Lagrange, cubic spline, windowed sinc
template <typename T> using Interpolator = tsd::filtrage::Interpolateur<T>;
template <typename T> using InterpolatorFIR = tsd::filtrage::InterpolateurRIF<T>;
template <typename T> sptr<InterpolatorFIR<T>> itrp_cspline() { return tsd::filtrage::itrp_cspline<T>(); }                 // :1755-1758
template <typename T> auto itrp_linear() { return tsd::filtrage::itrp_lineaire<T>(); }
template <typename T> auto itrp_lagrange(int degree) { return tsd::filtrage::itrp_lagrange<T>(degree); }
// (:1801-1805.  The reference's forwarder passes these three positional arguments to tsd::filtrage::itrp_sinc, which takes an
// InterpolateurSincConfig -- it does not compile once instantiated.  This one builds the structure.)
template <typename T> sptr<InterpolatorFIR<T>> itrp_sinc(int ncoefs, float fcut = 0.5, const std::string &window_type = "hn")
{
  tsd::filtrage::InterpolateurSincConfig c;
  c.ncoefs = ncoefs;
  c.fcut = fcut;
  c.fenetre = window_type;
  return tsd::filtrage::itrp_sinc<T>(c);
}
// resampling at an arbitrary ratio through an interpolator (:1910): filtre_itrp
template <typename T> sptr<FilterGen<T>> filter_itrp(float ratio, sptr<Interpolator<T>> itrp = itrp_cspline<T>())
{ return tsd::filtrage::filtre_itrp<T>(ratio, itrp); }
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
}  // namespace dsp
