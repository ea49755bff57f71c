// tsd/filtrage.hpp -- host-side mirror of libtsd's filtering API for the streaming hot path
// (namespace tsd::filtrage), backed by the MI355X C ABI.  Same names / argument meaning as
//   core/include/tsd/filtrage.hpp:24-85 (Design), :839 (design_rif_fen), :700 (design_riia),
//   :1367 (filtre_rif), :1402 (filtre_rif_fft), :1585-1590 (filtre_sois), :1684-1780
//   (filtrer / filtfilt / convol), :1814-1943 (Interpolateur, itrp_sinc), :2029-2039
//   (filtre_reechan, filtre_itrp);  core/include/tsd/filtrage/frat.hpp (FRat / Poly subset).
#pragma once
#include "tsd/tsd.hpp"
#include "tsd/filtrage/frat.hpp"   // Poly / FRat live in namespace tsd, like in libtsd

namespace tsd::filtrage {

// ---- Design (filtrage.hpp:24-85) ---------------------------------------------------------------
struct Design {
  Design(const FRat<cfloat> &f) : est_complexe(true), est_rif(false), frat_c(f) {}
  Design(const FRat<float> &f) : est_complexe(false), est_rif(false), frat(f) {}
  Design(const Vecf &c) : est_complexe(false), est_rif(true), coefs(c) {}
  Design(const Vecf &coefs_numer, const Vecf &coefs_dénom) : est_complexe(false), est_rif(false), frat(FRat<float>::rii(coefs_numer, coefs_dénom)) {}
  Design() {}
  bool est_complexe = false, est_rif = false;
  FRat<float> frat;
  FRat<cfloat> frat_c;
  Vecf coefs;
};

typedef enum { FormeDirecte1, FormeDirecte2 } RIIStructure;

// ---- design helpers (run once on the host) ---------------------------------------------------
Vecf fenêtre(cstring type, entier n, bouléen symetrique = true);
// enum spelling (filtrage.hpp:119-134); BLACKMAN and CHEBYCHEV are design-time windows that are not built
enum class Fenetre { AUCUNE = 0, HANN, TRIANGLE, HAMMING, BLACKMAN, CHEBYCHEV };
Vecf fenêtre(Fenetre type, entier n, bouléen symetrique = true);                       // "hn","hm","re","tr"
Vecf design_rif_fen(entier n, cstring type, float fc, cstring fen = "hn", float fc2 = 0);
Vecf design_rif_prod(const Vecf &h1, const Vecf &h2);
FRat<cfloat> design_riia(entier n, cstring type, cstring prototype, float fc, float δ_bp = 0.1f, float δ_bc = 60);
float sinc(float T, float f);
// design_biquad (filtrage.hpp:565-652; rii.cc:489-667): second-order sections from the Audio-EQ-Cookbook
// prototypes; type strings "lp"/"pb", "hp"/"ph", "bp"/"passe-bande", "cb"/"notch"/"sb", "plateau-bf",
// "plateau-hf", "res".  The result is a coefficient-form FRat<float>: filtrer() factorises it (parity of that
// factorisation is unpinned -- Eigen's solver in the reference).
struct BiquadSpec {
  enum Type { PASSE_BAS = 0, PASSE_HAUT, PASSE_BANDE, COUPE_BANDE, RESONATEUR, PLATEAU_BF, PLATEAU_HF } type = PASSE_BAS;
  float f = 0.25f, Q = 0.707f, gain_dB = 1;
};
FRat<float> design_biquad(cstring type, float f, float Q, float gain_dB = 0);
FRat<float> design_biquad(const BiquadSpec &spec);

// ---- stateful operators (factories) ------------------------------------------------------------
template <typename Tc, typename T = Tc> sptr<FiltreGen<T>> filtre_rif(const Vecteur<Tc> &h);
template <typename T> sptr<FiltreGen<T>> filtre_id();                                        // filtrage.hpp:1376-1377
template <typename T> sptr<FiltreGen<T>> filtre_rif_fft(const Vecf &h);
template <typename T> sptr<FiltreGen<T>> filtre_sois(const FRat<cfloat> &h, RIIStructure structure = FormeDirecte2);
template <typename T> sptr<FiltreGen<T>> filtre_sois(const FRat<float> &h, RIIStructure structure = FormeDirecte2);

template <typename Tc, typename T = Tc> sptr<FiltreGen<T>> filtre_rii(const FRat<Tc> &h);   // filtrage.hpp:1428-1429

// ---- small first-order / running operators (filtrage.hpp:1127-1134,1192-1222,1324,1610-1652;
// src/filtrage/filtre-rt.cc:14-51,603-786; src/filtrage/filtrage.cc:121-139), all on the C ABI:
//   filtre_lexp(γ)  y_n = y_{n-1} + γ (x_n - y_{n-1}), y_{-1} = x_0   -> one seeded DF1 section (block-parallel SOS kernel)
//   filtre_dc(fc)   y_n = α ((x_n - x_{n-1}) + y_{n-1}), α = 1 - lexp_coef(fc)  -> tsdgpu_rii (zero-seeded DF1 section)
//   filtre_mg(K)    y_n = (1/K) sum_{k<K} x_{n-k}                       -> FIR with K equal taps (direct / overlap-save)
//   ligne_a_retard(n)  y_i = x_{i-n}                                    -> index shift (host: pure data movement)
struct Fréquence {
  float value;
  Fréquence(float v) : value(v) {}
  operator float() const { return value; }
};
float lexp_coef(Fréquence fc);
float lexp_tc_vers_coef(float τ);
float lexp_coef_vers_tc(float γ);
Fréquence lexp_fcoupure(float γ);
template <typename T> sptr<FiltreGen<T>> ligne_a_retard(entier n);
template <typename T> sptr<FiltreGen<T>> filtre_lexp(float γ);
template <typename T> sptr<FiltreGen<T>> filtre_dc(float fc);
template <typename T, typename Tacc> sptr<FiltreGen<T>> filtre_mg(entier K);

// ---- integer-rate stages (filtrage.hpp:1968-1998; src/reechan/polyphase.cc; filtre-rt.cc:127-169)
template <typename T> sptr<FiltreGen<T>> decimateur(entier R);
template <typename Tc, typename T = Tc> sptr<FiltreGen<T>> filtre_rif_decim(const Vecteur<Tc> &h, entier R);
template <typename Tc, typename T = Tc> sptr<FiltreGen<T>> filtre_rif_demi_bande(const Vecteur<Tc> &h);
template <typename Tc, typename T = Tc> sptr<FiltreGen<T>> filtre_rif_ups(const Vecteur<Tc> &h, entier R);
float filtre_rif_ups_délais(entier nc, entier R);
float rif_delais(entier nc);
// forme_polyphase (polyphase.cc:16-46): zero-pad to a multiple of M and view as M rows x n/M
// columns, column-major -- a pure index permutation (here: the identity on memory).
template <typename T> struct TabPoly {
  entier lignes = 0, colonnes = 0;
  Vecteur<T> données;                                   // column-major
  T &operator()(entier i, entier j) { return données(i + j * lignes); }
};
template <typename T> TabPoly<T> forme_polyphase(const Vecteur<T> &x, entier M)
{
  TabPoly<T> X;
  const entier n = x.rows();
  if (n == 0) return X;
  const entier r = n % M;
  X.données = r ? vconcat(x, Vecteur<T>::zeros(M - r)) : x.clone();
  X.lignes = M;
  X.colonnes = X.données.rows() / M;
  return X;
}
template <typename T> Vecteur<T> iforme_polyphase(const TabPoly<T> &X) { return X.données.clone(); }

// ---- interpolators / resampling (filtrage.hpp:1814-1943,2029-2039) -----------------------------
template <typename T> struct Interpolateur {
  entier K = 0;
  float delais = 0;
  std::string nom;
  virtual ~Interpolateur() {}
  virtual T step(const Vecteur<T> &x, entier k, float τ) = 0;
};
template <typename T> struct InterpolateurRIF : Interpolateur<T> {
  virtual Vecf coefs(float τ) = 0;
  T step(const Vecteur<T> &x, entier k, float τ) override
  {
    const Vecf h = coefs(τ);
    T res = T(0);
    for (entier i = 0; i < this->K; i++) res += h(i) * x((i + k) % this->K);
    return res;
  }
};
struct InterpolateurSincConfig {
  entier ncoefs = 31;
  entier nphases = 256;
  float fcut = 0.5;
  std::string fenetre = "hn";
};
// Interpolators whose coefficients come from a table indexed by (int)(τ * nphases) -- the sinc
// (itrp.cc:10-55) and the cubic spline (itrp.cc:56-79,293-320) -- expose it so that the GPU
// resampler takes it as is.  itrp_lineaire / itrp_lagrange evaluate their coefficients from the
// float phase itself (itrp.cc:80-133): the GPU resampler evaluates the same formulas per output
// (tsdgpu_resampler_create_analytic).
template <typename T> struct InterpolateurLut : InterpolateurRIF<T> {
  entier nphases = 0;
  std::vector<float> lut;   // phase-major [(nphases+1) x K]
  Vecf coefs(float τ) override;
};
template <typename T> struct InterpolateurSinc : InterpolateurLut<T> {
  InterpolateurSincConfig config;
  explicit InterpolateurSinc(const InterpolateurSincConfig &c);
};
template <typename T> struct InterpolateurCSpline : InterpolateurLut<T> {
  explicit InterpolateurCSpline(entier n = 256, float c = 0);
};
template <typename T> struct InterpolateurLineaire : InterpolateurRIF<T> {      // itrp.cc:80-94
  InterpolateurLineaire() { this->nom = "linéaire"; this->K = 2; this->delais = 0.5f; }
  Vecf coefs(float τ) override { return Vecf::valeurs({1 - τ, τ}); }
};
template <typename T> struct InterpolateurLagrange : InterpolateurRIF<T> {      // itrp.cc:96-133
  entier d;
  explicit InterpolateurLagrange(entier d_) : d(d_)
  {
    this->nom = "Lagrange degré " + std::to_string(d);
    this->K = d + 1;
    this->delais = 0.5f * d;
  }
  Vecf coefs(float τ) override
  {
    Vecf h(d + 1);
    const float t = ((d - 1.0f) / 2) + τ;
    for (entier j = 0; j <= d; j++) {
      float p = 1.0f;
      for (entier k = 0; k <= d; k++)
        if (k != j) p *= (t - k) / (j - k);
      h(j) = p;
    }
    return h;
  }
};
template <typename T> sptr<InterpolateurRIF<T>> itrp_lineaire();
template <typename T> sptr<InterpolateurRIF<T>> itrp_lagrange(entier degré);
template <typename T> sptr<InterpolateurRIF<T>> itrp_cspline();
template <typename T> sptr<InterpolateurRIF<T>> itrp_sinc(const InterpolateurSincConfig &config);
template <typename T> sptr<FiltreGen<T>> filtre_itrp(float ratio, sptr<Interpolateur<T>> itrp);
template <typename T> sptr<Filtre<T, T, float>> filtre_reechan(float ratio);

// ---- one-shot API (filtrage.hpp:1684-1780) -----------------------------------------------------
template <typename T> Vecteur<T> filtrer(const Design &d, const Vecteur<T> &x)
{
  if (d.est_rif) {
    auto f = filtre_rif<float, T>(d.coefs);
    return f->step(x);
  }
  if (d.est_complexe) {
    auto f = filtre_sois<T>(d.frat_c);
    return f->step(x);
  }
  if (d.frat.est_rif()) {
    // a FRat that is an FIR: numerator in z, reversed to get the taps (filtrage.hpp:1702-1704)
    auto f = filtre_rif<T, T>(Vecteur<T>(d.frat.numer.coefs.reverse()));
    return f->step(x);
  }
  auto f = filtre_sois<T>(d.frat);
  return f->step(x);
}
template <typename T> Vecteur<T> filtfilt(const Design &h, const Vecteur<T> &x)
{
  return filtrer(h, filtrer<T>(h, x).reverse()).reverse();
}
template <typename T, typename Tc> Vecteur<T> convol(const Vecteur<Tc> &h, const Vecteur<T> &x)
{
  auto f = filtre_rif<Tc, T>(h);
  return f->step(x);
}

}  // namespace tsd::filtrage
