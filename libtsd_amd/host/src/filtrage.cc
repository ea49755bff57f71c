// filtrage.cc -- C++ adaptors: libtsd's filter factories on top of the MI355X C ABI, plus the
// host-side design helpers the hot path's configurations need.
#include "tsd/filtrage.hpp"
#include "tsd/fourier.hpp"
#include "../../../include/tsdgpu.h"
#include <set>

namespace tsd {

// ---- commons ---------------------------------------------------------------------------------
logger_t &get_logger()
{
  static logger_t l;
  return l;
}
void set_logger(logger_t l) { get_logger() = std::move(l); }

std::default_random_engine &generateur_aleatoire()
{
  static std::default_random_engine g;   // default-seeded, like core/src/tsd.cc:173
  return g;
}
Vecf randn(entier n)
{
  std::normal_distribution<float> d(0.f, 1.f);
  Vecf x(n);
  for (entier i = 0; i < n; i++) x(i) = d(generateur_aleatoire());
  return x;
}
Veccf randcn(entier n)
{
  std::normal_distribution<float> d(0.f, 1.f);
  Veccf x(n);
  for (entier i = 0; i < n; i++) {
    const float a = d(generateur_aleatoire()), b = d(generateur_aleatoire());
    x(i) = cfloat(a, b);
  }
  return x;
}
entier prochaine_puissance_de_2(entier i)
{
  const entier lg2 = (entier) std::ceil(std::log((float) i) / std::log(2.0f));
  return (entier) (1l << lg2);
}

namespace filtrage {

namespace {
template <typename T> constexpr int dtype_of() { return est_complexe<T>() ? TSDGPU_C64 : TSDGPU_F32; }
[[noreturn]] void gpu_fail(const char *what) { échec("{}: {}", what, tsdgpu_last_error()); }
}  // namespace

// ---- design: windowed sinc (rif-fen.cc:31-108, fenetres.cc:16-60,127-130, divers.cc:6-12) --
float sinc(float T, float f)
{
  const float a = π_f * T * f;
  if (std::abs(a) < 1e-7f) return T;
  return std::sin(a) / (π_f * f);
}

static Vecf fen_inter(entier n, bool sym)
{
  float tmin, tmax;
  if ((n & 1) == 0) {
    tmin = (float) (-n / 2);
    tmax = sym ? (float) (n / 2) : (float) ((n - 1) / 2);
  } else {
    tmin = (float) (-n / 2);
    tmax = sym ? (float) (n / 2) : (float) (n / 2) - ((float) n - 1) / n;
  }
  return linspace(tmin / n, tmax / n, n);
}

Vecf fenêtre(cstring type, entier n, bouléen symetrique)
{
  Vecf x = Vecf::zeros(n);
  if (type == "re" || type == "rect" || type == "none") {
    x.setConstant(1);
    return x;
  }
  const Vecf t = fen_inter(n, symetrique);
  if (type == "hn" || type == "hann" || type == "hm" || type == "hamming") {
    const float a = (type[1] == 'n' || type == "hann") ? 0.5f : 0.54f;
    for (entier i = 0; i < n; i++) x(i) = a + (1 - a) * std::cos((float) (2 * π) * t(i));
    return x;
  }
  if (type == "tr" || type == "triangle") {
    for (entier i = 0; i < n; i++) x(i) = t(i) < 0 ? 2 * (0.5f + t(i)) : 2 * (0.5f - t(i));
    return x;
  }
  échec("fenêtre: window type '{}' is not built in this hot-path mirror (have hn, hm, re, tr)", type);
}

Vecf fenêtre(Fenetre type, entier n, bouléen symetrique)
{
  switch (type) {
    case Fenetre::AUCUNE: return fenêtre("re", n, symetrique);
    case Fenetre::HANN: return fenêtre("hn", n, symetrique);
    case Fenetre::TRIANGLE: return fenêtre("tr", n, symetrique);
    case Fenetre::HAMMING: return fenêtre("hm", n, symetrique);
    default: échec("fenêtre: this window is not built in the hot-path mirror (have AUCUNE, HANN, TRIANGLE, HAMMING)");
  }
}

static Vecf coefs_filtre_sinc(entier n, float fc)
{
  if (n & 1) return Vecf::int_expr(n, [&](entier i) { return sinc(2 * fc, (float) (i - n / 2)); });
  return Vecf::int_expr(n, [&](entier i) { return sinc(2 * fc, (float) (i - (n - 1) / 2)); });
}

Vecf design_rif_fen(entier n, cstring type, float fc, cstring fen, float fc2)
{
  (void) fc2;
  const Vecf f = fenêtre(fen, n, true);
  Vecf h;
  if (type == "lp" || type == "pb") {
    h = coefs_filtre_sinc(n, fc);
  } else if (type == "hp" || type == "ph") {
    h = -coefs_filtre_sinc(n, fc);
    h((n - 1) / 2) += 1.0f;
  } else {
    échec("design_rif_fen: type '{}' is not built in this hot-path mirror (have lp/pb, hp/ph)", type);
  }
  Vecf h2 = h * f;
  if (type == "lp") h2 /= h2.somme();     // only the literal "lp" is normalised (rif-fen.cc:96-98)
  return h2;
}

Vecf design_rif_prod(const Vecf &h1, const Vecf &h2)
{
  // filtrage.cc:47-52: filtrer(h1, [h2, 0 ... 0])
  const Vecf h2p = vconcat(h2, Vecf::zeros(h1.rows() - 1));
  return filtrer<float>(Design(h1), h2p);
}

// ---- design: Butterworth low-pass through the bilinear transform (rii.cc:20-23,41-73,
//      173-187,195-215,405-452) --------------------------------------------------------------
FRat<cfloat> design_riia(entier n, cstring type, cstring prototype, float fc, float, float)
{
  if (!(type == "lp" || type == "pb") || prototype.substr(0, 1) != "b")
    échec("design_riia: only the Butterworth low-pass (\"lp\", \"butt\") is built in this hot-path mirror "
          "(got type '{}', prototype '{}'); other prototypes are design-time code outside the path",
          type, prototype);
  const float wd = (float) (2 * π * fc);
  const float wa = 2 * 1.0f * std::tan(wd / (2 * 1.0f));
  Veccf z(n), p(n);
  cfloat gain = 1.0f;
  for (entier i = 0; i < n; i++) {
    const float k = (float) (i + 1);
    const float ang = ((float) π * (2 * k + (float) (n - 1))) / (float) (2 * n);
    const cfloat pa = cfloat(std::cos(ang), std::sin(ang)) * wa;
    p(i) = (pa + 2.0f) / (-pa + 2.0f);
    z(i) = cfloat(-1.f, 0.f);
    gain /= (2.0f - pa);
  }
  FRat<cfloat> h;
  h.numer = Poly<cfloat>::from_roots(z);
  h.denom = Poly<cfloat>::from_roots(p);
  h.numer.mlt = cfloat((float) std::pow((double) wa, (double) n), 0.f) * gain;
  h.denom.mlt = cfloat(1.f, 0.f);
  return h;
}

// ---- polynomial roots -----------------------------------------------------------------------
template <typename T> Veccf Poly<T>::roots() const
{
  if (mode_racines) return Veccf(coefs.template as<cfloat>());
  // coefficient form, ascending powers: sum c_k x^k.  Durand-Kerner on the monic polynomial.
  entier deg = coefs.rows() - 1;
  while (deg > 0 && std::abs(cfloat(coefs(deg))) == 0.f) deg--;
  Veccf r(std::max(deg, 0));
  if (deg <= 0) return r;
  std::vector<cdouble> a((size_t) deg + 1), x((size_t) deg);
  for (entier k = 0; k <= deg; k++) a[k] = cdouble(cfloat(coefs(k))) / cdouble(cfloat(coefs(deg)));
  for (entier k = 0; k < deg; k++) x[k] = std::pow(cdouble(0.4, 0.9), k);
  for (int it = 0; it < 500; it++) {
    double delta = 0;
    for (entier i = 0; i < deg; i++) {
      cdouble num = 1, xp = 1;
      num = 0;
      for (entier k = 0; k <= deg; k++) { num += a[k] * xp; xp *= x[i]; }
      cdouble den = 1;
      for (entier j = 0; j < deg; j++) if (j != i) den *= (x[i] - x[j]);
      const cdouble d = num / den;
      x[i] -= d;
      delta = std::max(delta, std::abs(d));
    }
    if (delta < 1e-14) break;
  }
  for (entier i = 0; i < deg; i++) r(i) = cfloat(x[i]);
  return r;
}
template struct Poly<float>;
template struct Poly<cfloat>;

// ---- FIR: filtre_rif<Tc,T>  (filtre-rt.cc:53-109,171-175) ------------------------------------
template <typename T, typename Tc> struct FiltreRIFGpu : FiltreGen<T> {
  tsdgpu_fir *h = nullptr;
  FiltreRIFGpu(const Vecteur<Tc> &c, int method)
  {
    if (c.rows() <= 0) échec("filtre_rif: K > 0 required");          // assertion(K > 0), filtre-rt.cc:69
    if (tsdgpu_fir_create(&h, dtype_of<T>(), dtype_of<Tc>(), c.data(), c.rows(), method)) gpu_fail("filtre_rif");
  }
  ~FiltreRIFGpu() override { tsdgpu_fir_destroy(h); }
  void step(const Vecteur<T> &x, Vecteur<T> &y) override
  {
    const entier n = x.rows();
    if (x.data() != y.data()) y.resize(n);                            // in place allowed (filtre-rt.cc:76-80)
    if (tsdgpu_fir_step(h, x.data(), y.data(), n, nullptr)) gpu_fail("filtre_rif::step");
  }
};

template <typename Tc, typename T> sptr<FiltreGen<T>> filtre_rif(const Vecteur<Tc> &c)
{
  return std::make_shared<FiltreRIFGpu<T, Tc>>(c, TSDGPU_FIR_AUTO);
}
template sptr<FiltreGen<float>> filtre_rif<float, float>(const Vecf &);
template sptr<FiltreGen<cfloat>> filtre_rif<float, cfloat>(const Vecf &);
template sptr<FiltreGen<cfloat>> filtre_rif<cfloat, cfloat>(const Veccf &);

// ---- filtre_rif_fft<T> (fourier.cc:946-990): the reference's OLA FIR.  Its output is the
// direct FIR delayed by Nz - M samples (Ne = 512, N = pp2(Ne + M), Nz = N - Ne), and for
// T = cfloat only the real part survives (fourier.cc:976).  Both are reproduced: the block
// convolution runs on the GPU overlap-save kernel, the delay line lives here.
template <typename T> struct FiltreFFTRIFGpu : FiltreGen<T> {
  FiltreRIFGpu<T, float> rif;
  Vecteur<T> retard;      // the last d outputs not yet delivered
  FiltreFFTRIFGpu(const Vecf &c) : rif(c, TSDGPU_FIR_OVERLAP_SAVE)
  {
    const entier M = c.rows(), Ne = 512;
    const entier N = prochaine_puissance_de_2(Ne + M), Nz = N - Ne;
    if (Nz > Ne) échec("filtre_rif_fft: {} coefficients need Nz = {} > Ne = {} (the reference's OLA limit)", M, Nz, Ne);
    retard = Vecteur<T>::zeros(Nz - M);
  }
  void step(const Vecteur<T> &x, Vecteur<T> &y) override
  {
    Vecteur<T> z;
    rif.step(x, z);
    if constexpr (est_complexe<T>())
      for (entier i = 0; i < z.rows(); i++) z(i) = cfloat(z(i).real(), 0.f);   // real(...) re-widened
    const entier n = z.rows(), d = retard.rows();
    const Vecteur<T> all = vconcat(retard, z);
    y.resize(n);
    for (entier i = 0; i < n; i++) y(i) = all(i);
    for (entier i = 0; i < d; i++) retard(i) = all(n + i);
  }
};
template <typename T> sptr<FiltreGen<T>> filtre_rif_fft(const Vecf &c) { return std::make_shared<FiltreFFTRIFGpu<T>>(c); }
template sptr<FiltreGen<float>> filtre_rif_fft<float>(const Vecf &);
template sptr<FiltreGen<cfloat>> filtre_rif_fft<cfloat>(const Vecf &);

// ---- SOS chain: filtre_sois<T> (filtre-rt.cc:440-602) ---------------------------------------
// Greedy conjugate pairing exactly as ChaineSOIS' constructor does it: take the lowest unused
// index k, pair it with the unused j that minimises the imaginary residue of the two
// quadratics, zeros and poles paired with the SAME indices (:467-528).
template <typename T> struct ChaineSOISGpu : FiltreGen<T> {
  tsdgpu_sos *h = nullptr;
  ChaineSOISGpu(const FRat<cfloat> &f, RIIStructure structure)
  {
    const Veccf z = f.numer.roots(), p = f.denom.roots();
    const entier nz = z.rows(), np = p.rows();
    if (nz != np) échec("ChaineSOIS: numerator and denominator must have the same degree (nz={}, np={})", nz, np);
    std::set<entier> pool;
    for (entier i = 0; i < nz; i++) pool.insert(i);
    std::vector<float> coefs;
    entier i;
    for (i = 0; i + 1 < nz; i += 2) {
      const entier k = *pool.begin();
      pool.erase(pool.begin());
      float berr = 1e9f, sz = 0, pz = 0, sp = 0, pp = 0;
      auto bj = pool.begin();
      for (entier j = 0; j < nz; j++) {
        if (pool.count(j) == 0) continue;
        const cfloat sz0 = -(z(j) + z(k)), pz0 = z(j) * z(k), sp0 = -(p(j) + p(k)), pp0 = p(j) * p(k);
        const float err = std::abs(sz0.imag()) + std::abs(pz0.imag()) + std::abs(sp0.imag()) + std::abs(pp0.imag());
        if (err < berr) {
          bj = pool.find(j);
          berr = err;
          sz = sz0.real(); pz = pz0.real(); sp = sp0.real(); pp = pp0.real();
        }
      }
      pool.erase(bj);
      if (berr > 1e-5f) msg("Factorisation SOIS : erreur = {}", berr);
      // section {b0,b1,b2 ; a0,a1,a2} = {1, sz, pz ; 1, sp, pp}, normalised by a0 (:317-328)
      for (float v : {1.0f, sz, pz, sp, pp}) coefs.push_back(v);
    }
    float rii1[3];
    bool avec_rii1 = false;
    float gain = 1.0f;
    for (; i < nz; i++) {
      const entier id = *pool.begin();
      const cfloat zer = z(id), pol = p(id);
      const float b0 = f.numer.mlt.real() / f.denom.mlt.real();      // (:550-552)
      rii1[0] = b0;
      rii1[1] = -zer.real() * b0;
      rii1[2] = -pol.real();
      avec_rii1 = true;
    }
    if (!avec_rii1) gain = f.numer.mlt.real() / f.denom.mlt.real();  // (:558-559)
    if (tsdgpu_sos_create(&h, dtype_of<T>(), coefs.data(), (int) (coefs.size() / 5), gain, avec_rii1 ? rii1 : nullptr,
                          structure == FormeDirecte2 ? 2 : 1))
      gpu_fail("filtre_sois");
  }
  ~ChaineSOISGpu() override { tsdgpu_sos_destroy(h); }
  void step(const Vecteur<T> &x, Vecteur<T> &y) override
  {
    const entier n = x.rows();
    if (x.data() != y.data()) y.resize(n);
    if (tsdgpu_sos_step(h, x.data(), y.data(), n, nullptr)) gpu_fail("filtre_sois::step");
  }
};

template <typename T> sptr<FiltreGen<T>> filtre_sois(const FRat<cfloat> &h, RIIStructure structure)
{
  return std::make_shared<ChaineSOISGpu<T>>(h, structure);
}
template <typename T> sptr<FiltreGen<T>> filtre_sois(const FRat<float> &h, RIIStructure structure)
{
  // (:580-602) coefficient form is factorised first; H(z^-1) coefficient lists are in powers
  // of z^-1, i.e. descending powers of z once multiplied through.
  FRat<cfloat> h2;
  auto conv = [](const Poly<float> &p) {
    Poly<cfloat> q;
    if (p.mode_racines) {
      q = Poly<cfloat>::from_roots(p.coefs.as<cfloat>());
      q.mlt = p.mlt;
      return q;
    }
    // a0 + a1 z^-1 + ... + an z^-n  ==  z^-n (a0 z^n + ... + an): roots in z of the reversed list
    Poly<float> r;
    r.coefs = p.coefs.reverse();
    q = Poly<cfloat>::from_roots(r.roots());
    q.mlt = cfloat(p.coefs(0), 0.f);
    return q;
  };
  h2.numer = conv(h.numer);
  h2.denom = conv(h.denom);
  return std::make_shared<ChaineSOISGpu<T>>(h2, structure);
}
template sptr<FiltreGen<float>> filtre_sois<float>(const FRat<cfloat> &, RIIStructure);
template sptr<FiltreGen<cfloat>> filtre_sois<cfloat>(const FRat<cfloat> &, RIIStructure);
template sptr<FiltreGen<float>> filtre_sois<float>(const FRat<float> &, RIIStructure);
template sptr<FiltreGen<cfloat>> filtre_sois<cfloat>(const FRat<float> &, RIIStructure);

// ---- integer-rate stages and the generic IIR ---------------------------------------------
template <typename T> struct PolyFirGpu : FiltreGen<T> {
  tsdgpu_polyfir *h = nullptr;
  PolyFirGpu(int kind, const float *taps, int K, int R)
  {
    if (tsdgpu_polyfir_create(&h, kind, dtype_of<T>(), taps, K, R)) gpu_fail("filtre_rif_decim/_demi_bande/_ups/decimateur");
  }
  ~PolyFirGpu() override { tsdgpu_polyfir_destroy(h); }
  void step(const Vecteur<T> &x, Vecteur<T> &y) override
  {
    const entier n = x.rows();
    const int64_t cap = tsdgpu_polyfir_out_count(h, n);
    if (cap < 0 || cap > 0x7fffffff) échec("polyphase stage: output size {} not representable", (long long) cap);
    Vecteur<T> out((entier) cap);
    int64_t got = 0;
    if (n > 0 && tsdgpu_polyfir_step(h, x.data(), n, out.data(), cap, &got, nullptr)) gpu_fail("polyphase stage step");
    y = std::move(out);
  }
};
template <typename T> sptr<FiltreGen<T>> decimateur(entier R) { return std::make_shared<PolyFirGpu<T>>(TSDGPU_POLY_PICK, nullptr, 0, R); }
template <typename Tc, typename T> sptr<FiltreGen<T>> filtre_rif_decim(const Vecteur<Tc> &c, entier R)
{
  return std::make_shared<PolyFirGpu<T>>(TSDGPU_POLY_DECIM, c.data(), c.rows(), R);
}
template <typename Tc, typename T> sptr<FiltreGen<T>> filtre_rif_demi_bande(const Vecteur<Tc> &c)
{
  return std::make_shared<PolyFirGpu<T>>(TSDGPU_POLY_HALFBAND, c.data(), c.rows(), 2);
}
template <typename Tc, typename T> sptr<FiltreGen<T>> filtre_rif_ups(const Vecteur<Tc> &c, entier R)
{
  return std::make_shared<PolyFirGpu<T>>(TSDGPU_POLY_UPS, c.data(), c.rows(), R);
}
template sptr<FiltreGen<float>> decimateur<float>(entier);
template sptr<FiltreGen<cfloat>> decimateur<cfloat>(entier);
template sptr<FiltreGen<float>> filtre_rif_decim<float, float>(const Vecf &, entier);
template sptr<FiltreGen<cfloat>> filtre_rif_decim<float, cfloat>(const Vecf &, entier);
template sptr<FiltreGen<float>> filtre_rif_demi_bande<float, float>(const Vecf &);
template sptr<FiltreGen<cfloat>> filtre_rif_demi_bande<float, cfloat>(const Vecf &);
template sptr<FiltreGen<float>> filtre_rif_ups<float, float>(const Vecf &, entier);
template sptr<FiltreGen<cfloat>> filtre_rif_ups<float, cfloat>(const Vecf &, entier);
float filtre_rif_ups_délais(entier nc, entier R)     // polyphase.cc:363-369
{
  entier pad = 0;
  if ((nc % R) != 0) pad = R - (nc % R);
  return (float) ((nc - 1) / 2.0 + pad);
}
float rif_delais(entier nc) { return (nc - 1) / 2.0f; }

// FiltreRII (filtre-rt.cc:177-289) from a coefficient-form H(z^-1)
template <typename T> struct FiltreRIIGpu : FiltreGen<T> {
  tsdgpu_rii *h = nullptr;
  explicit FiltreRIIGpu(const FRat<float> &f)
  {
    if (f.numer.mode_racines || f.denom.mode_racines)
      échec("filtre_rii: give the transfer function in coefficient form (FRat::rii); pole/zero forms go through filtre_sois");
    if (tsdgpu_rii_create(&h, dtype_of<T>(), f.numer.coefs.data(), f.numer.coefs.rows(), f.denom.coefs.data(), f.denom.coefs.rows()))
      gpu_fail("filtre_rii");
  }
  ~FiltreRIIGpu() override { tsdgpu_rii_destroy(h); }
  void step(const Vecteur<T> &x, Vecteur<T> &y) override
  {
    const entier n = x.rows();
    if (x.data() != y.data()) y.resize(n);
    if (n > 0 && tsdgpu_rii_step(h, x.data(), y.data(), n, nullptr)) gpu_fail("filtre_rii::step");
  }
};
template <typename Tc, typename T> sptr<FiltreGen<T>> filtre_rii(const FRat<Tc> &h) { return std::make_shared<FiltreRIIGpu<T>>(h); }
template sptr<FiltreGen<float>> filtre_rii<float, float>(const FRat<float> &);
template sptr<FiltreGen<cfloat>> filtre_rii<float, cfloat>(const FRat<float> &);

// ---- filtre_lexp / filtre_dc / filtre_mg / ligne_a_retard (filtre-rt.cc:14-51,603-786) ----------
float lexp_coef(Fréquence fc) { return (float) (1.0 - std::exp(-fc.value * 2 * π)); }
float lexp_tc_vers_coef(float τ) { return lexp_coef((float) (1.0 / (2 * π * τ))); }
Fréquence lexp_fcoupure(float γ) { return (float) (-std::log(1.0 - γ) / (2 * π)); }
float lexp_coef_vers_tc(float γ) { return (float) (1.0 / (2 * π * lexp_fcoupure(γ).value)); }

// FiltreLExp: acc <- x(0) on the first sample, then acc += γ (x - acc).  As a transfer function
// that is the FormeDirecte1 section (γ, 0, 0 ; 1, -(1-γ), 0) whose memories all start at x(0) --
// exactly SOIS' first-call seed -- so it runs on the block-parallel SOS kernel.
template <typename T> struct FiltreLExpGpu : FiltreGen<T> {
  tsdgpu_sos *h = nullptr;
  explicit FiltreLExpGpu(float γ)
  {
    // pole a = fl(1 - γ), gain 1 - a (exact in float): the DC gain stays exactly 1, as it is for the
    // reference's incremental form -- with b0 = γ the float rounding of 1 - γ alone would move the
    // DC gain by 3e-8 / γ (1.5e-5 at γ = 0.002)
    const float a = 1.0f - γ;
    const float coefs[5] = {1.0f - a, 0.f, 0.f, -a, 0.f};
    if (tsdgpu_sos_create(&h, dtype_of<T>(), coefs, 1, 1.0f, nullptr, 1)) gpu_fail("filtre_lexp");
  }
  ~FiltreLExpGpu() override { tsdgpu_sos_destroy(h); }
  void step(const Vecteur<T> &x, Vecteur<T> &y) override
  {
    const entier n = x.rows();
    if (n == 0) return;
    if (x.data() != y.data()) y.resize(n);
    if (tsdgpu_sos_step(h, x.data(), y.data(), n, nullptr)) gpu_fail("filtre_lexp::step");
  }
};
template <typename T> sptr<FiltreGen<T>> filtre_lexp(float γ) { return std::make_shared<FiltreLExpGpu<T>>(γ); }
template sptr<FiltreGen<float>> filtre_lexp<float>(float);
template sptr<FiltreGen<cfloat>> filtre_lexp<cfloat>(float);

// FiltreDC: y = α ((x - xp) + yp) from zero memory = H(z^-1) = (α - α z^-1) / (1 - α z^-1)
template <typename T> sptr<FiltreGen<T>> filtre_dc(float fc)
{
  const float α = 1 - lexp_coef(Fréquence(fc));
  return std::make_shared<FiltreRIIGpu<T>>(FRat<float>::rii(Vecf::valeurs({α, -α}), Vecf::valeurs({1.0f, -α})));
}
template sptr<FiltreGen<float>> filtre_dc<float>(float);
template sptr<FiltreGen<cfloat>> filtre_dc<cfloat>(float);

// MoyenneGlissante: running sum in Tacc times (T)(1/K) = a K-tap FIR with equal taps (the GPU sums
// the K products in float: relative difference ~ 1e-7 sqrt(K) to the reference's double accumulator)
template <typename T, typename Tacc> sptr<FiltreGen<T>> filtre_mg(entier K)
{
  if (K <= 0) échec("filtre_mg: K = {}", K);
  Vecf h(K);
  h.setConstant((float) (1.0 / (double) K));
  return std::make_shared<FiltreRIFGpu<T, float>>(h, TSDGPU_FIR_AUTO);
}
template sptr<FiltreGen<float>> filtre_mg<float, double>(entier);
template sptr<FiltreGen<cfloat>> filtre_mg<cfloat, cdouble>(entier);
template sptr<FiltreGen<float>> filtre_mg<float, float>(entier);

// LigneARetard: y_i = x_{i-n} with n zeros first, streaming.  Pure index work on the caller's
// (host) vectors: no arithmetic, so no kernel.
template <typename T> struct LigneARetardHote : FiltreGen<T> {
  Vecteur<T> mem;          // the last n inputs, oldest first
  explicit LigneARetardHote(entier n) : mem(Vecteur<T>::zeros(n < 0 ? 0 : n)) {}
  void step(const Vecteur<T> &x, Vecteur<T> &y) override
  {
    const entier d = mem.rows(), n = x.rows();
    if (d == 0) {
      y = x;
      return;
    }
    const Vecteur<T> all = vconcat(mem, x);
    Vecteur<T> out(n);
    for (entier i = 0; i < n; i++) out(i) = all(i);
    for (entier i = 0; i < d; i++) mem(i) = all(n + i);
    y = out;
  }
};
template <typename T> sptr<FiltreGen<T>> ligne_a_retard(entier n) { return std::make_shared<LigneARetardHote<T>>(n); }
template sptr<FiltreGen<float>> ligne_a_retard<float>(entier);
template sptr<FiltreGen<cfloat>> ligne_a_retard<cfloat>(entier);

// ---- resampling: itrp_sinc / filtre_itrp / filtre_reechan -----------------------------------
template <typename T> InterpolateurSinc<T>::InterpolateurSinc(const InterpolateurSincConfig &c) : config(c)
{
  // itrp.cc:24-54: lut.col(j) = coefs_calcule(j / nphases), Hann window shifted by -tau
  const entier nc = c.ncoefs;
  this->K = nc;
  this->delais = 0.5f * nc;
  this->nom = detail::fmt("sinc - ncoefs={}, nphases={}, fcut={}, fen={}", nc, c.nphases, c.fcut, c.fenetre);
  this->nphases = c.nphases;
  auto &lut = this->lut;
  lut.resize((size_t) (c.nphases + 1) * nc);
  const Vecf ls = linspace((float) (-nc / 2), (float) ((nc - 1) / 2), nc);
  for (entier j = 0; j <= c.nphases; j++) {
    const float τ = (float) ((1.0 * j) / c.nphases);
    for (entier i = 0; i < nc; i++) {
      const float hv = sinc(2 * c.fcut, (float) (i - nc / 2) - τ);
      const float t = (ls(i) - τ) * (float) (2 * π / nc);
      // Hann window shifted by the fractional delay; any other window name = no window (itrp.cc:29-37)
      lut[(size_t) j * nc + i] = c.fenetre == "hn" ? hv * (0.5f + 2 * 0.25f * std::cos(t)) : hv;
    }
  }
}
template <typename T> Vecf InterpolateurLut<T>::coefs(float τ)
{
  if (!(τ >= 0 && τ <= 1)) échec("Interpolateur::coefs(τ={}) : délais invalide.", τ);
  const entier idx = (entier) (τ * nphases);
  return Vecf::int_expr(this->K, [&](entier i) { return lut[(size_t) idx * this->K + i]; });
}
template struct InterpolateurLut<float>;
template struct InterpolateurLut<cfloat>;
template struct InterpolateurSinc<float>;
template struct InterpolateurSinc<cfloat>;

// cubic (cardinal) spline, tension c: coefficients on (p-1, p0, p1, p2) from the Hermite basis
// (itrp.cc:293-320), tabulated at τ = i/n
template <typename T> InterpolateurCSpline<T>::InterpolateurCSpline(entier n, float c)
{
  this->nom = "cspline";
  this->K = 4;
  this->delais = 1.5f;
  this->nphases = n;
  this->lut.resize((size_t) (n + 1) * 4);
  for (entier i = 0; i <= n; i++) {
    const float t = ((float) i) / n;
    const float h0 = (1 + 2 * t) * (t - 1) * (t - 1), h1 = t * (t - 1) * (t - 1), h2 = t * t * (3 - 2 * t), h3 = t * t * (t - 1);
    float *o = &this->lut[(size_t) i * 4];
    o[0] = -(1 - c) * h1 / 2;
    o[1] = h0 - (1 - c) * h3 / 2;
    o[2] = h2 + (1 - c) * h1 / 2;
    o[3] = (1 - c) * h3 / 2;
  }
}
template struct InterpolateurCSpline<float>;
template struct InterpolateurCSpline<cfloat>;
template <typename T> sptr<Interpolateur<T>> itrp_cspline() { return std::make_shared<InterpolateurCSpline<T>>(); }
template sptr<Interpolateur<float>> itrp_cspline<float>();
template sptr<Interpolateur<cfloat>> itrp_cspline<cfloat>();

template <typename T> sptr<Interpolateur<T>> itrp_sinc(const InterpolateurSincConfig &config)
{
  return std::make_shared<InterpolateurSinc<T>>(config);
}
template sptr<Interpolateur<float>> itrp_sinc<float>(const InterpolateurSincConfig &);
template sptr<Interpolateur<cfloat>> itrp_sinc<cfloat>(const InterpolateurSincConfig &);

// AdaptationRythmeSimple (ra.cc:13-79) on the GPU resampler: table-driven or analytic interpolators
template <typename T> struct AdaptationRythmeSimpleGpu : FiltreGen<T> {
  tsdgpu_resampler *h = nullptr;
  AdaptationRythmeSimpleGpu(float ratio, sptr<Interpolateur<T>> itrp)
  {
    if (auto s = std::dynamic_pointer_cast<InterpolateurLut<T>>(itrp)) {
      if (tsdgpu_resampler_create(&h, dtype_of<T>(), ratio, s->lut.data(), s->K, s->nphases)) gpu_fail("filtre_itrp");
    } else if (std::dynamic_pointer_cast<InterpolateurLineaire<T>>(itrp)) {
      if (tsdgpu_resampler_create_analytic(&h, dtype_of<T>(), ratio, TSDGPU_ITRP_LINEAR, 1)) gpu_fail("filtre_itrp");
    } else if (auto l = std::dynamic_pointer_cast<InterpolateurLagrange<T>>(itrp)) {
      if (tsdgpu_resampler_create_analytic(&h, dtype_of<T>(), ratio, TSDGPU_ITRP_LAGRANGE, l->d)) gpu_fail("filtre_itrp");
    } else {
      échec("filtre_itrp: interpolator '{}' is neither table-driven (itrp_sinc, itrp_cspline) nor one of itrp_lineaire / "
            "itrp_lagrange: no GPU path for a user-defined coefs()", itrp ? itrp->nom : std::string("null"));
    }
  }
  ~AdaptationRythmeSimpleGpu() override { tsdgpu_resampler_destroy(h); }
  void step(const Vecteur<T> &x, Vecteur<T> &y) override
  {
    const entier n = x.rows();
    if (n == 0) {
      y.resize(0);
      return;
    }
    const int64_t cap = tsdgpu_resampler_out_count(h, n);
    if (cap < 0 || cap > 0x7fffffff) échec("filtre_itrp::step: output size {} not representable", (long long) cap);
    y.resize((entier) cap);
    int64_t got = 0;
    if (tsdgpu_resampler_step(h, x.data(), n, y.data(), cap, &got, nullptr)) gpu_fail("filtre_itrp::step");
  }
};
template <typename T> sptr<FiltreGen<T>> filtre_itrp(float ratio, sptr<Interpolateur<T>> itrp)
{
  return std::make_shared<AdaptationRythmeSimpleGpu<T>>(ratio, itrp);
}
template sptr<FiltreGen<float>> filtre_itrp<float>(float, sptr<Interpolateur<float>>);
template sptr<FiltreGen<cfloat>> filtre_itrp<cfloat>(float, sptr<Interpolateur<cfloat>>);

// AdaptationRythmeArbitraire (ra.cc:84-183)
template <typename T> struct AdaptationRythmeArbitraireGpu : Filtre<T, T, float> {
  sptr<FiltreGen<T>> interpolateur;
  std::vector<sptr<FiltreGen<T>>> décimateurs, suréchantilloneurs;
  float facteur_post_interpolation = 1, ratio = 1;
  explicit AdaptationRythmeArbitraireGpu(float r) { Configurable<float>::configure(r); }
  void configure_impl(const float &ratio_) override
  {
    ratio = ratio_;
    if (ratio <= 0 || std::isinf(ratio) || ratio >= 1e9f) {
      msg("AdaptationRythmeArbitraire::configurer() : facteur de décimation invalide : {}.", ratio);
      ratio = 1;
    }
    facteur_post_interpolation = ratio;
    entier nb_sur = 0, nb_dec = 0;
    while (facteur_post_interpolation < 0.5) { nb_dec++; facteur_post_interpolation *= 2; }
    while (facteur_post_interpolation >= 2) { nb_sur++; facteur_post_interpolation /= 2; }
    // half-band cascade around the interpolator (ra.cc:136-144): 15-tap Hann, fc = 0.25
    const Vecf coefs = design_rif_fen(15, "lp", 0.25f, "hn");
    décimateurs.clear();
    suréchantilloneurs.clear();
    for (entier i = 0; i < nb_dec; i++) décimateurs.push_back(filtre_rif_demi_bande<float, T>(coefs));
    for (entier i = 0; i < nb_sur; i++) suréchantilloneurs.push_back(filtre_rif_ups<float, T>(coefs, 2));
    const float fcut = std::min(0.4f, facteur_post_interpolation / 2);
    auto itrp = itrp_sinc<T>({15, 256, fcut, "hn"});
    interpolateur = filtre_itrp<T>(facteur_post_interpolation, itrp);
  }
  void step(const Vecteur<T> &x, Vecteur<T> &y) override
  {
    y = x;
    if (ratio == 1) return;                                                   // (ra.cc:162-163)
    for (auto &d : décimateurs) y = d->step(y);
    for (auto &s : suréchantilloneurs) y = s->step(y);
    if (std::abs(facteur_post_interpolation - 1) < 1e-6f) return;             // (ra.cc:172-174)
    y = interpolateur->step(y);
  }
};
template <typename T> sptr<Filtre<T, T, float>> filtre_reechan(float ratio)
{
  return std::make_shared<AdaptationRythmeArbitraireGpu<T>>(ratio);
}
template sptr<Filtre<float, float, float>> filtre_reechan<float>(float);
template sptr<Filtre<cfloat, cfloat, float>> filtre_reechan<cfloat>(float);

}  // namespace filtrage
}  // namespace tsd
