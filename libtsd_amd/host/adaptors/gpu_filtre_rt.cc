// gpu_filtre_rt.cc -- the factories of libtsd's core/src/filtrage/filtre-rt.cc on the MI355X C ABI.
// Compiled against libtsd's own headers this TU defines, with libtsd's mangled names, every factory
// filtre-rt.cc exports: filtre_rif (:171-175), filtre_rii (:285-289), filtre_sois x2 (:574-602),
// filtre_lexp / filtre_dc / filtre_mg (:764-781), ligne_a_retard (:14-51), decimateur (:127-169),
// filtre_id -- for T in {float, cfloat} like the explicit instantiations at :787-823.
#include "gpu_commun.hpp"
#include <set>

namespace tsd::filtrage {

using tsd_amd::dimensionne;
using tsd_amd::dtype_of;
using tsd_amd::gpu_fail;

// ---- FIR: FiltreRIF<T,Tc> (filtre-rt.cc:53-109) ----------------------------------------------------
template <typename T, typename Tc> struct FiltreRIFGpu : FiltreGen<T> {
  tsdgpu_fir *h = nullptr;
  tsdgpu_sharded *hs = nullptr;      // several GPUs (see gpu_commun.hpp)
  std::vector<Tc> coefs;
  int methode, mode = -1;            // mode: -1 undecided, 0 one GPU, 1 sharded over the node
  explicit FiltreRIFGpu(const Vecteur<Tc> &c, int method = TSDGPU_FIR_AUTO) : coefs(c.data(), c.data() + std::max(c.rows(), 0)), methode(method)
  {
    if (c.rows() <= 0) échec("filtre_rif: K > 0 required (K = {})", (int) c.rows());      // assertion(K > 0), :69
    if (tsdgpu_fir_create(&h, dtype_of<T>(), dtype_of<Tc>(), c.data(), c.rows(), method)) gpu_fail("filtre_rif");
  }
  ~FiltreRIFGpu()
  {
    tsdgpu_fir_destroy(h);
    tsdgpu_sharded_destroy(hs);
  }
  void step(const Vecteur<T> &x, Vecteur<T> &y)
  {
    const entier n = x.rows();
    if (x.data() != y.data()) dimensionne(y, n);                           // in place allowed (:76-80)
    if (mode < 0 && n > 0) mode = tsd_amd::choisit_fragments(x, y) ? 1 : 0;
    if (mode == 1) {
      tsd_amd::exige_hote_fragments(x, "filtre_rif");
      if (!hs && tsdgpu_fir_sharded_create(&hs, dtype_of<T>(), dtype_of<Tc>(), coefs.data(), (int) coefs.size(), methode,
                                           tsd_amd::nb_fragments(), nullptr))
        gpu_fail("filtre_rif (multi-GPU)");
      int64_t got = 0;
      if (tsdgpu_sharded_step_host(hs, x.data(), n, y.data(), n, &got)) gpu_fail("filtre_rif::step (multi-GPU)");
      return;
    }
    if (tsdgpu_fir_step(h, x.data(), y.data(), n, nullptr)) gpu_fail("filtre_rif::step");
  }
};
template <typename Tc, typename T> sptr<FiltreGen<T>> filtre_rif(const Vecteur<Tc> &c)
{
  return std::make_shared<FiltreRIFGpu<T, Tc>>(c);
}
template sptr<FiltreGen<float>> filtre_rif<float, float>(const Vecteur<float> &);
template sptr<FiltreGen<cfloat>> filtre_rif<float, cfloat>(const Vecteur<float> &);
template sptr<FiltreGen<cfloat>> filtre_rif<cfloat, cfloat>(const Vecteur<cfloat> &);

// ---- SOS chain: ChaineSOIS<T,T,T> (filtre-rt.cc:440-572) ----------------------------------------
// Greedy conjugate pairing as ChaineSOIS' constructor does it: the lowest unused index k is
// paired with the unused j that minimises the imaginary residue of the two quadratics; zeros and
// poles are paired with the SAME indices (:467-528).  An odd order leaves a first-order section
// that carries the gain (:530-556), otherwise y *= gain (:558-559,570).
template <typename T> struct ChaineSOISGpu : FiltreGen<T> {
  tsdgpu_sos *h = nullptr;
  tsdgpu_sharded *hs = nullptr;      // several GPUs (see gpu_commun.hpp)
  std::vector<float> coefs_sections, rii1_v;
  float gain_v = 1.0f;
  int forme = 2, mode = -1;
  ChaineSOISGpu(const FRat<cfloat> &f, RIIStructure structure)
  {
    const Vecteur<cfloat> z = f.numer.roots(), p = f.denom.roots();
    const entier nz = z.rows(), np = p.rows();
    if (nz != np) échec("ChaineSOIS: numerator and denominator must have the same degree (nz={}, np={})", (int) nz, (int) np);
    std::set<entier> libres;
    for (entier i = 0; i < nz; i++) libres.insert(i);
    std::vector<float> coefs;
    entier i;
    for (i = 0; i + 1 < nz; i += 2) {
      const entier k = *libres.begin();
      libres.erase(libres.begin());
      float meilleur = 1e9f, sz = 0, pz = 0, sp = 0, pp = 0;
      auto choix = libres.begin();
      for (auto it = libres.begin(); it != libres.end(); ++it) {
        const entier j = *it;
        const cfloat sz0 = -(z.data()[j] + z.data()[k]), pz0 = z.data()[j] * z.data()[k];
        const cfloat sp0 = -(p.data()[j] + p.data()[k]), pp0 = p.data()[j] * p.data()[k];
        const float err = std::abs(sz0.imag()) + std::abs(pz0.imag()) + std::abs(sp0.imag()) + std::abs(pp0.imag());
        if (err < meilleur) {
          choix = it;
          meilleur = err;
          sz = sz0.real(); pz = pz0.real(); sp = sp0.real(); pp = pp0.real();
        }
      }
      libres.erase(choix);
      if (meilleur > 1e-5f) msg("Factorisation SOIS : erreur = {}", meilleur);
      // section {b0,b1,b2 ; a0,a1,a2} = {1, sz, pz ; 1, sp, pp}, already normalised by a0 (:317-328)
      for (float v : {1.0f, sz, pz, sp, pp}) coefs.push_back(v);
    }
    float rii1[3] = {0, 0, 0};
    bool avec_rii1 = false;
    float gain = 1.0f;
    const float g = f.numer.mlt.real() / f.denom.mlt.real();
    if (i < nz) {
      const entier id = *libres.begin();
      rii1[0] = g;                                                          // (:550-552)
      rii1[1] = -z.data()[id].real() * g;
      rii1[2] = -p.data()[id].real();
      avec_rii1 = true;
    } else {
      gain = g;                                                             // (:558-559)
    }
    if (tsdgpu_sos_create(&h, dtype_of<T>(), coefs.data(), (int) (coefs.size() / 5), gain, avec_rii1 ? rii1 : nullptr,
                          structure == FormeDirecte2 ? 2 : 1))
      gpu_fail("filtre_sois");
    coefs_sections = coefs;
    if (avec_rii1) rii1_v.assign(rii1, rii1 + 3);
    gain_v = gain;
    forme = structure == FormeDirecte2 ? 2 : 1;
  }
  ~ChaineSOISGpu()
  {
    tsdgpu_sos_destroy(h);
    tsdgpu_sharded_destroy(hs);
  }
  void step(const Vecteur<T> &x, Vecteur<T> &y)
  {
    const entier n = x.rows();
    if (x.data() != y.data()) dimensionne(y, n);
    // (any cascade shards: by warm-up halos when it forgets its past within 2^16 samples, by an exact exchange of the
    // shards' end states otherwise -- tsdgpu_sos_sharded_create picks)
    if (mode < 0 && n > 0) mode = tsd_amd::choisit_fragments(x, y) ? 1 : 0;
    if (mode == 1) {
      tsd_amd::exige_hote_fragments(x, "filtre_sois");
      if (!hs && tsdgpu_sos_sharded_create(&hs, dtype_of<T>(), coefs_sections.data(), (int) (coefs_sections.size() / 5), gain_v,
                                           rii1_v.empty() ? nullptr : rii1_v.data(), forme, tsd_amd::nb_fragments(), nullptr))
        gpu_fail("filtre_sois (multi-GPU)");
      int64_t got = 0;
      if (tsdgpu_sharded_step_host(hs, x.data(), n, y.data(), n, &got)) gpu_fail("filtre_sois::step (multi-GPU)");
      return;
    }
    if (tsdgpu_sos_step(h, x.data(), y.data(), n, nullptr)) gpu_fail("filtre_sois::step");
  }
};
template <typename T> sptr<FiltreGen<T>> filtre_sois(const FRat<cfloat> &h, RIIStructure structure)
{
  return std::make_shared<ChaineSOISGpu<T>>(h, structure);
}
template <typename T> sptr<FiltreGen<T>> filtre_sois(const FRat<float> &h, RIIStructure structure)
{
  // (:580-602) a real fraction is brought to pole/zero form: roots kept as they are in
  // mode_racines, else the roots of the coefficient list times its leading coefficient
  auto vers_racines = [](const Poly<float> &p) {
    Poly<cfloat> q = Poly<cfloat>::from_roots(p.roots());
    q.vname = p.vname;
    q.mlt = p.mode_racines ? cfloat(p.mlt, 0.f) : cfloat(p.coefs.data()[p.coefs.rows() - 1], 0.f);
    return q;
  };
  FRat<cfloat> h2;
  h2.numer = vers_racines(h.numer);
  h2.denom = vers_racines(h.denom);
  return std::make_shared<ChaineSOISGpu<T>>(h2, structure);
}
template sptr<FiltreGen<float>> filtre_sois<float>(const FRat<cfloat> &, RIIStructure);
template sptr<FiltreGen<cfloat>> filtre_sois<cfloat>(const FRat<cfloat> &, RIIStructure);
template sptr<FiltreGen<float>> filtre_sois<float>(const FRat<float> &, RIIStructure);
template sptr<FiltreGen<cfloat>> filtre_sois<cfloat>(const FRat<float> &, RIIStructure);

// ---- FiltreRII<T,Tc> (filtre-rt.cc:177-289): direct form I from H(z), any order ------------------
template <typename T, typename Tc> struct FiltreRIIGpu : FiltreGen<T> {
  tsdgpu_rii *h = nullptr;
  explicit FiltreRIIGpu(const FRat<Tc> &f)
  {
    // powers of z^-1, like the reference's constructor (:185-193)
    const FRat<Tc> f2 = f.eval_inv_z();
    const Vecteur<Tc> num = f2.numer.vers_coefs().coefs, den = f2.denom.vers_coefs().coefs;
    init(num, den);
  }
  FiltreRIIGpu(const Vecteur<Tc> &num, const Vecteur<Tc> &den) { init(num, den); }
  void init(const Vecteur<Tc> &num, const Vecteur<Tc> &den)
  {
    if (tsdgpu_rii_create2(&h, dtype_of<T>(), dtype_of<Tc>(), num.data(), num.rows(), den.data(), den.rows())) gpu_fail("filtre_rii");
  }
  ~FiltreRIIGpu() { tsdgpu_rii_destroy(h); }
  void step(const Vecteur<T> &x, Vecteur<T> &y)
  {
    const entier n = x.rows();
    if (x.data() != y.data()) dimensionne(y, n);
    if (n > 0 && tsdgpu_rii_step(h, x.data(), y.data(), n, nullptr)) gpu_fail("filtre_rii::step");
  }
};
template <typename Tc, typename T> sptr<FiltreGen<T>> filtre_rii(const FRat<Tc> &h) { return std::make_shared<FiltreRIIGpu<T, Tc>>(h); }
template sptr<FiltreGen<float>> filtre_rii<float, float>(const FRat<float> &);
template sptr<FiltreGen<cfloat>> filtre_rii<cfloat, cfloat>(const FRat<cfloat> &);
template sptr<FiltreGen<cfloat>> filtre_rii<float, cfloat>(const FRat<float> &);    // not in libtsd: real H on complex data

// ---- FiltreLExp (filtre-rt.cc:725-763): acc <- x(0) on the first sample, then acc += γ (x - acc).
// As a transfer function: the FormeDirecte1 section (γ, 0, 0 ; 1, -(1-γ), 0) whose memories all
// start at x(0) -- exactly SOIS' first-call seed -- so it runs on the block-parallel SOS kernel.
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
  ~FiltreLExpGpu() { tsdgpu_sos_destroy(h); }
  void step(const Vecteur<T> &x, Vecteur<T> &y)
  {
    const entier n = x.rows();
    if (n == 0) return;
    if (x.data() != y.data()) dimensionne(y, n);
    if (tsdgpu_sos_step(h, x.data(), y.data(), n, nullptr)) gpu_fail("filtre_lexp::step");
  }
};
template <typename T> sptr<FiltreGen<T>> filtre_lexp(float γ) { return std::make_shared<FiltreLExpGpu<T>>(γ); }
template sptr<FiltreGen<float>> filtre_lexp<float>(float);
template sptr<FiltreGen<cfloat>> filtre_lexp<cfloat>(float);

// FiltreDC (filtre-rt.cc:604-629): y = α ((x - xp) + yp) from zero memory = (α - α z^-1) / (1 - α z^-1)
template <typename T> sptr<FiltreGen<T>> filtre_dc(float fc)
{
  const float α = 1 - lexp_coef(Fréquence(fc));
  Vecteur<float> num(2), den(2);
  num(0) = α; num(1) = -α;
  den(0) = 1.0f; den(1) = -α;
  return std::make_shared<FiltreRIIGpu<T, float>>(num, den);
}
template sptr<FiltreGen<float>> filtre_dc<float>(float);
template sptr<FiltreGen<cfloat>> filtre_dc<cfloat>(float);

// MoyenneGlissante (filtre-rt.cc:633-667): running sum in Tacc times 1/K = a K-tap FIR with equal
// taps (the GPU sums the K products in float: ~1e-7 sqrt(K) relative to the double accumulator)
template <typename T, typename Tacc> sptr<FiltreGen<T>> filtre_mg(entier K)
{
  if (K <= 0) échec("filtre_mg: K = {}", (int) K);
  Vecteur<float> h(K);
  for (entier i = 0; i < K; i++) h(i) = (float) (1.0 / (double) K);
  return std::make_shared<FiltreRIFGpu<T, float>>(h);
}
template sptr<FiltreGen<float>> filtre_mg<float, double>(entier);
template sptr<FiltreGen<cfloat>> filtre_mg<cfloat, cdouble>(entier);

// LigneARetard (filtre-rt.cc:14-51): y_i = x_{i-n}, n zeros first.  Pure index work: no kernel.
template <typename T> struct LigneARetardHote : FiltreGen<T> {
  std::vector<T> mem;          // the last n inputs, oldest first
  explicit LigneARetardHote(entier n) : mem((size_t) (n < 0 ? 0 : n), T(0)) {}
  void step(const Vecteur<T> &x, Vecteur<T> &y)
  {
    const size_t d = mem.size(), n = (size_t) x.rows(), sz = sizeof(T);
    // tout = mem ++ x (copies work from host or device vectors alike)
    std::vector<T> tout(d + n);
    std::copy(mem.begin(), mem.end(), tout.begin());
    tsd_amd::copie_octets(tout.data() + d, x.data(), n * sz);
    if (x.data() != y.data()) dimensionne(y, (entier) n);
    tsd_amd::copie_octets(y.data(), tout.data(), n * sz);
    std::copy(tout.begin() + n, tout.end(), mem.begin());
  }
};
template <typename T> sptr<FiltreGen<T>> ligne_a_retard(entier n) { return std::make_shared<LigneARetardHote<T>>(n); }
template sptr<FiltreGen<float>> ligne_a_retard<float>(entier);
template sptr<FiltreGen<cfloat>> ligne_a_retard<cfloat>(entier);

// filtre_id: y = x
template <typename T> struct FiltreIdHote : FiltreGen<T> {
  void step(const Vecteur<T> &x, Vecteur<T> &y)
  {
    if (x.data() == y.data()) return;
    dimensionne(y, x.rows());
    tsd_amd::copie_octets(y.data(), x.data(), (size_t) x.rows() * sizeof(T));
  }
};
template <typename T> sptr<FiltreGen<T>> filtre_id() { return std::make_shared<FiltreIdHote<T>>(); }
template sptr<FiltreGen<float>> filtre_id<float>();
template sptr<FiltreGen<cfloat>> filtre_id<cfloat>();

// Decimateur (filtre-rt.cc:127-169): one sample in R, phase carried across calls (bit-exact index pick)
template <typename T> struct DecimateurGpu : FiltreGen<T> {
  tsdgpu_polyfir *h = nullptr;
  explicit DecimateurGpu(entier R)
  {
    if (tsdgpu_polyfir_create(&h, TSDGPU_POLY_PICK, dtype_of<T>(), nullptr, 0, R)) gpu_fail("decimateur");
  }
  ~DecimateurGpu() { tsdgpu_polyfir_destroy(h); }
  void step(const Vecteur<T> &x, Vecteur<T> &y)
  {
    const entier n = x.rows();
    const int64_t cap = tsdgpu_polyfir_out_count(h, n);
    tsd_amd::sortie_variable(x, y, cap, [&](T *out) {
      int64_t got = 0;
      if (n > 0 && tsdgpu_polyfir_step(h, x.data(), n, out, cap, &got, nullptr)) gpu_fail("decimateur::step");
    });
  }
};
template <typename T> sptr<FiltreGen<T>> decimateur(entier R) { return std::make_shared<DecimateurGpu<T>>(R); }
template sptr<FiltreGen<float>> decimateur<float>(entier);
template sptr<FiltreGen<cfloat>> decimateur<cfloat>(entier);

}  // namespace tsd::filtrage
