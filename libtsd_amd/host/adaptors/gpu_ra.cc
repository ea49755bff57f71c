// gpu_ra.cc -- the factories of libtsd's core/src/reechan/ra.cc on the MI355X C ABI:
// filtre_itrp<T>(ratio, interpolateur) = AdaptationRythmeSimple (:13-79,185-188) and
// filtre_reechan<T>(ratio) = AdaptationRythmeArbitraire (:84-183); instantiated like ra.cc:191-194.
#include "gpu_commun.hpp"
#include <cmath>
#include <cstring>

namespace tsd::filtrage {

using tsd_amd::dtype_of;
using tsd_amd::gpu_fail;

// ---- what kind of interpolator is this? ------------------------------------------------------------
// libtsd's interpolator classes are private to itrp.cc; filtre_itrp only sees Interpolateur<T> /
// InterpolateurRIF<T> (filtrage.hpp:1814-1882).  The GPU resampler needs either the coefficient
// table (itrp_sinc, itrp_cspline: coefs(τ) = column (int)(τ nphases) of a table, itrp.cc:16-22,66-77)
// or the closed form (itrp_lineaire, itrp_lagrange: itrp.cc:80-133), so the interpolator is probed
// through its public coefs(τ): first breakpoint by bisection -> nphases, table read row by row, the
// hypothesis verified on a few thousand phases; otherwise compared with the two closed forms.
struct SondeItrp {
  enum Genre { TABLE, LINEAIRE, LAGRANGE, INCONNU } genre = INCONNU;
  entier K = 0, nphases = 0, degré = 0;
  std::vector<float> lut;   // phase-major [(nphases + 1) x K]
};

template <typename T> static SondeItrp sonde_interpolateur(const sptr<Interpolateur<T>> &itrp)
{
  SondeItrp s;
  auto rif = std::dynamic_pointer_cast<InterpolateurRIF<T>>(itrp);
  if (!rif || itrp->K < 1 || itrp->K > 256) return s;
  const entier K = s.K = itrp->K;
  auto coefs = [&](float τ) {
    const Vecteur<float> h = rif->coefs(τ);
    std::vector<float> v((size_t) K, 0.f);
    if (h.rows() == K) std::copy(h.data(), h.data() + K, v.begin());
    return v;
  };
  const std::vector<float> c0 = coefs(0.f);
  auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
  auto flt = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
  // smallest τ in (0, 1] whose coefficients differ from those of τ = 0 (positive floats order like
  // their bit patterns)
  uint32_t lo = 0, hi = bits(1.0f);
  if (coefs(1.0f) != c0) {
    while (hi - lo > 1) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (coefs(flt(mid)) != c0) hi = mid; else lo = mid;
    }
  }
  const double b = flt(hi);
  const double np_d = 1.0 / b;
  // deterministic probe phases in [0, 1)
  auto phase = [](int i) { return (float) ((i * 2654435761u) >> 8) * (1.0f / 16777216.0f); };
  if (np_d >= 0.5 && np_d < 8191.5) {
    const entier np = (entier) std::lround(np_d);
    s.lut.resize((size_t) (np + 1) * K);
    for (entier j = 0; j <= np; j++) {
      const std::vector<float> r = coefs(j < np ? ((float) j + 0.5f) / (float) np : 1.0f);
      std::copy(r.begin(), r.end(), s.lut.begin() + (size_t) j * K);
    }
    // the hypothesis coefs(τ) == row[(int) (τ np)], checked at both ends of every row's interval and at 256 scattered
    // phases (a probe per one-shot rééchan(): 4096 scattered phases cost 0.2 ms of allocations)
    bool ok = true;
    auto verifie = [&](float τ) {
      if (!(τ >= 0.f && τ < 1.f)) return;
      const entier idx = (entier) (τ * np);
      const Vecteur<float> r = rif->coefs(τ);
      ok = ok && r.rows() == K && std::equal(r.data(), r.data() + K, s.lut.begin() + (size_t) idx * K);
    };
    for (entier j = 0; j < np && ok; j++) {
      verifie(std::nextafter((float) j / (float) np, 2.0f));
      verifie(std::nextafter((float) (j + 1) / (float) np, -1.0f));
    }
    for (int i = 0; i < 256 && ok; i++) verifie(phase(i));
    if (ok) {
      s.genre = SondeItrp::TABLE;
      s.nphases = np;
      return s;
    }
    s.lut.clear();
  }
  // closed forms: linear {1 - τ, τ}; Lagrange of degree d = K - 1 evaluated at (d-1)/2 + τ
  bool lin = K == 2, lag = K >= 2;
  const entier d = K - 1;
  for (int i = 0; i < 256 && (lin || lag); i++) {
    const float τ = phase(i);
    const std::vector<float> r = coefs(τ);
    if (lin) lin = r[0] == 1 - τ && r[1] == τ;
    if (lag) {
      const float t = ((d - 1.0f) / 2) + τ;
      for (entier j = 0; j <= d && lag; j++) {
        float p = 1.0f;
        for (entier k = 0; k <= d; k++)
          if (k != j) p *= (t - k) / (j - k);
        lag = std::abs(r[j] - p) <= 1e-6f * (1.0f + std::abs(p));
      }
    }
  }
  if (lin) { s.genre = SondeItrp::LINEAIRE; s.degré = 1; }
  else if (lag) { s.genre = SondeItrp::LAGRANGE; s.degré = d; }
  return s;
}

// ---- AdaptationRythmeSimple (ra.cc:13-79) on the GPU resampler ---------------------------------------
template <typename T> struct AdaptationRythmeSimpleGpu : FiltreGen<T> {
  tsdgpu_resampler *h = nullptr;
  tsdgpu_sharded *hs = nullptr;      // several GPUs (see gpu_commun.hpp); table-driven interpolators only
  SondeItrp s;
  float ratio_v = 1;
  int mode = -1;
  AdaptationRythmeSimpleGpu(float ratio, sptr<Interpolateur<T>> itrp) : ratio_v(ratio)
  {
    if (!itrp) échec("filtre_itrp: null interpolator");
    s = sonde_interpolateur<T>(itrp);
    switch (s.genre) {
      case SondeItrp::TABLE:
        if (tsdgpu_resampler_create(&h, dtype_of<T>(), ratio, s.lut.data(), s.K, s.nphases)) gpu_fail("filtre_itrp");
        break;
      case SondeItrp::LINEAIRE:
        if (tsdgpu_resampler_create_analytic(&h, dtype_of<T>(), ratio, TSDGPU_ITRP_LINEAR, 1)) gpu_fail("filtre_itrp");
        break;
      case SondeItrp::LAGRANGE:
        if (tsdgpu_resampler_create_analytic(&h, dtype_of<T>(), ratio, TSDGPU_ITRP_LAGRANGE, s.degré)) gpu_fail("filtre_itrp");
        break;
      default:
        échec("filtre_itrp: interpolator '{}' is neither table-driven (itrp_sinc, itrp_cspline: coefs(τ) piecewise constant "
              "over at most 8191 phases) nor itrp_lineaire / itrp_lagrange: no GPU path for this coefs()", itrp->nom);
    }
  }
  ~AdaptationRythmeSimpleGpu()
  {
    tsdgpu_resampler_destroy(h);
    tsdgpu_sharded_destroy(hs);
  }
  void step(const Vecteur<T> &x, Vecteur<T> &y)
  {
    const entier n = x.rows();
    if (mode < 0 && n > 0) mode = (s.genre == SondeItrp::TABLE && tsd_amd::choisit_fragments(x, y)) ? 1 : 0;
    if (mode == 1) {
      tsd_amd::exige_hote_fragments(x, "filtre_itrp");
      if (!hs && tsdgpu_resampler_sharded_create(&hs, dtype_of<T>(), ratio_v, s.lut.data(), s.K, s.nphases, tsd_amd::nb_fragments(), nullptr))
        gpu_fail("filtre_itrp (multi-GPU)");
      const int64_t capm = tsdgpu_sharded_out_count(hs, n);
      tsd_amd::sortie_variable(x, y, capm, [&](T *out) {
        int64_t got = 0;
        if (n > 0 && tsdgpu_sharded_step_host(hs, x.data(), n, out, capm, &got)) gpu_fail("filtre_itrp::step (multi-GPU)");
      });
      return;
    }
    const int64_t cap = n > 0 ? tsdgpu_resampler_out_count(h, n) : 0;
    tsd_amd::sortie_variable(x, y, cap, [&](T *out) {
      int64_t got = 0;
      if (n > 0 && tsdgpu_resampler_step(h, x.data(), n, out, cap, &got, nullptr)) gpu_fail("filtre_itrp::step");
    });
  }
};
template <typename T> sptr<FiltreGen<T>> filtre_itrp(float ratio, sptr<Interpolateur<T>> itrp)
{
  return std::make_shared<AdaptationRythmeSimpleGpu<T>>(ratio, itrp);
}
template sptr<FiltreGen<float>> filtre_itrp<float>(float, sptr<Interpolateur<float>>);
template sptr<FiltreGen<cfloat>> filtre_itrp<cfloat>(float, sptr<Interpolateur<cfloat>>);

// ---- AdaptationRythmeArbitraire (ra.cc:84-183) ---------------------------------------------------------
// ratio folded into [0.5, 2) by half-band decimators / upsamplers (15-tap Hann, fc = 0.25: :122-144),
// then the 15-tap / 256-phase sinc interpolator with fcut = min(0.4, residual / 2) (:149-152); bypass
// when the ratio is 1 or the residual is 1 (:162-174).  All the stages are GPU operators.
template <typename T> struct AdaptationRythmeArbitraireGpu : Filtre<T, T, float> {
  sptr<FiltreGen<T>> interpolateur;
  std::vector<sptr<FiltreGen<T>>> décimateurs, suréchantillonneurs;
  float résiduel = 1, ratio = 1;
  explicit AdaptationRythmeArbitraireGpu(float r) { Configurable<float>::configure(r); }
  void configure_impl(const float &ratio_)
  {
    ratio = ratio_;
    if (ratio <= 0 || std::isinf(ratio) || ratio >= 1e9f) {
      msg("AdaptationRythmeArbitraire::configurer() : facteur de décimation invalide : {}.", ratio);
      ratio = 1;
    }
    résiduel = ratio;
    entier nb_sur = 0, nb_dec = 0;
    while (résiduel < 0.5) { nb_dec++; résiduel *= 2; }
    while (résiduel >= 2) { nb_sur++; résiduel /= 2; }
    const Vecteur<float> coefs = design_rif_fen(15, "lp", 0.25f, "hn");
    décimateurs.clear();
    suréchantillonneurs.clear();
    for (entier i = 0; i < nb_dec; i++) décimateurs.push_back(filtre_rif_demi_bande<float, T>(coefs));
    for (entier i = 0; i < nb_sur; i++) suréchantillonneurs.push_back(filtre_rif_ups<float, T>(coefs, 2));
    InterpolateurSincConfig ic;
    ic.ncoefs = 15;
    ic.nphases = 256;
    ic.fcut = std::min(0.4f, résiduel / 2);
    ic.fenetre = "hn";
    interpolateur = filtre_itrp<T>(résiduel, sptr<Interpolateur<T>>(itrp_sinc<T>(ic)));
  }
  void step(const Vecteur<T> &x, Vecteur<T> &y)
  {
    if (ratio == 1) {                                                         // (ra.cc:162-163)
      if (x.data() != y.data()) {
        tsd_amd::dimensionne(y, x.rows());
        tsd_amd::copie_octets(y.data(), x.data(), (size_t) x.rows() * sizeof(T));
      }
      return;
    }
    const bool avec_itrp = !(std::abs(résiduel - 1) < 1e-6f);                 // (ra.cc:172-174)
    const size_t netages = décimateurs.size() + suréchantillonneurs.size() + (avec_itrp ? 1 : 0);
    // the last stage writes the caller's vector; intermediate vectors are temporaries
    Vecteur<T> a, b;
    const Vecteur<T> *src = &x;
    size_t fait = 0;
    auto etage = [&](sptr<FiltreGen<T>> &f) {
      fait++;
      if (fait == netages) {
        f->step(*src, y);
      } else {
        Vecteur<T> &dst = (src == &a) ? b : a;
#ifdef TSD_AMD_MIRROR
        ResidenceGpu garde;        // the mirror's vectors can live on the device: the intermediate stays there (one
#endif                             // upload of x, one download of y for the whole chain); libtsd's Tab cannot
        f->step(*src, dst);
        src = &dst;
      }
    };
    for (auto &d : décimateurs) etage(d);
    for (auto &s : suréchantillonneurs) etage(s);
    if (avec_itrp) etage(interpolateur);
  }
};
template <typename T> sptr<Filtre<T, T, float>> filtre_reechan(float ratio)
{
  return std::make_shared<AdaptationRythmeArbitraireGpu<T>>(ratio);
}
template sptr<Filtre<float, float, float>> filtre_reechan<float>(float);
template sptr<Filtre<cfloat, cfloat, float>> filtre_reechan<cfloat>(float);

}  // namespace tsd::filtrage
