// Reference-style tests of the C++ host layer (tsd:: / dsp:: API on the MI355X C ABI).
// They read like libtsd's own tests (core/tests/test-filtres.cc, test-fourier.cc, test-ra.cc,
// test-tab.cc, test-dsp.cc) and use the CPU oracle only as the checker.
//   usage: test_host_api            -> runs everything (needs a GPU)
//          test_host_api --no-gpu   -> checks that the factories fail loudly without a GPU
#include <cstdio>
#include <cstring>
#include <thread>
#include "dsp/dsp.hpp"
#include "dsp/filter.hpp"
#include "dsp/fourier.hpp"
#include "tsd_amd/extensions.hpp"
#include "../../oracle/tsd_oracle.h"

using namespace tsd;
using namespace tsd::filtrage;
using namespace tsd::fourier;

static int nfail = 0;
#define CHECK(cond, ...)                                                     \
  do {                                                                       \
    if (!(cond)) {                                                           \
      nfail++;                                                               \
      printf("FAIL %s:%d  %s  -- ", __FILE__, __LINE__, #cond);              \
      printf(__VA_ARGS__);                                                   \
      printf("\n");                                                          \
    }                                                                        \
  } while (0)

template <typename T> static float maxabs(const Vecteur<T> &v)
{
  float m = 0;
  for (int i = 0; i < v.rows(); i++) m = std::max(m, std::abs(v(i)));
  return m;
}

// test-filtres.cc:9-31
template <typename T> static Vecteur<T> filtre_par_bloc(sptr<FiltreGen<T>> f, const Vecteur<T> &x, int BS)
{
  std::vector<Vecteur<T>> lst;
  int N = x.rows(), offset = 0, n = 0;
  while (offset < N) {
    int nl = std::min(BS, N - offset);
    Vecteur<T> xp = x.segment(offset, nl).clone();
    Vecteur<T> yp = f->step(xp);
    offset += nl;
    n += yp.rows();
    lst.push_back(yp);
  }
  Vecteur<T> y(n);
  offset = 0;
  for (auto &v : lst) {
    y.segment(offset, v.rows()) = v;
    offset += v.rows();
  }
  return y;
}

// Device-resident vectors: a filtrer -> fft -> rééchan chain that never leaves
// the GPU gives what the host chain gives; host element access on a resident vector is refused
template <typename T> static float ecart_rel(const Vecteur<T> &a, const Vecteur<T> &b)
{
  if (a.rows() != b.rows()) return 1e30f;
  float e = 0, m = 0;
  for (int i = 0; i < a.rows(); i++) {
    e = std::max(e, (float) std::abs(a(i) - b(i)));
    m = std::max(m, (float) std::abs(b(i)));
  }
  return e / std::max(m, 1e-30f);
}
// element-wise arithmetic of RESIDENT float / cfloat vectors runs on the device (tsdgpu_vec_op) and gives what the host
// loops give: bit for bit for reverse, + - *, scalar products, abs2, real, imag, as_complex, negation; to an ulp for the
// complex quotient (evaluated in double like libgcc) and for abs (hypotf of two libraries); filtfilt stays resident
static void test_residence_ops()
{
  const int n = 100003;
  const Veccf x = randcn(n), z = randcn(n);
  const Vecf u = randn(n), w = randn(n);
  const Veccf xg = x.vers_gpu(), zg = z.vers_gpu();
  const Vecf ug = u.vers_gpu(), wg = w.vers_gpu();
  auto pareil = [](const auto &a_gpu, const auto &b_hote, const char *quoi, float tol) {
    CHECK(a_gpu.est_sur_gpu(), "%s: the result left the GPU", quoi);
    const auto a = a_gpu.vers_hote();
    CHECK(a.rows() == b_hote.rows(), "%s: %d vs %d elements", quoi, (int) a.rows(), (int) b_hote.rows());
    float e = 0, m = 0;
    for (int i = 0; i < a.rows(); i++) {
      e = std::max(e, (float) std::abs(a.data()[i] - b_hote.data()[i]));
      m = std::max(m, (float) std::abs(b_hote.data()[i]));
    }
    CHECK(e <= tol * m, "%s: device %s host (max deviation %g of %g)", quoi, tol == 0 ? "!=" : "too far from", e, m);
  };
  pareil(xg.reverse(), x.reverse(), "reverse (complex)", 0);
  pareil(ug.reverse(), u.reverse(), "reverse (real)", 0);
  pareil(xg + zg, x + z, "+", 0);
  pareil(xg - zg, x - z, "-", 0);
  pareil(xg * zg, x * z, "* (complex)", 0);
  pareil(ug * wg, u * w, "* (real)", 0);
  pareil(xg * cfloat(0.3f, -1.7f), x * cfloat(0.3f, -1.7f), "* complex scalar", 0);
  pareil(ug * 2.5f, u * 2.5f, "* real scalar", 0);
  pareil(ug / 3.0f, u / 3.0f, "/ real scalar", 0);
  pareil(xg / cfloat(1024.f, 0.f), x / cfloat(1024.f, 0.f), "/ real-valued complex scalar", 1.2e-7f);
  pareil(xg / cfloat(0.3f, -1.7f), x / cfloat(0.3f, -1.7f), "/ complex scalar", 1.2e-7f);
  pareil(-xg, -x, "negation", 0);
  pareil(abs2(xg), abs2(x), "abs2", 0);
  pareil(abs(xg), abs(x), "abs", 1.2e-7f);
  pareil(abs(ug), abs(u), "abs (real)", 0);
  pareil(real(xg), real(x), "real", 0);
  pareil(imag(xg), imag(x), "imag", 0);
  pareil(ug.as_complex(), u.as_complex(), "as_complex", 0);
  {
    Veccf a = xg.clone();
    a *= zg;
    a += xg;
    a /= cfloat(2.f, 0.f);
    Veccf b = x.clone();
    b *= z;
    b += x;
    b /= cfloat(2.f, 0.f);
    pareil(a, b, "in-place chain", 1.2e-7f);
  }
  // mixed residency and unsupported cases are refused, not silently staged
  bool threw = false;
  try { (void) (xg + z); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw, "resident + host must be refused");
  // reductions: the sum is accumulated in double on both sides (the orders differ: the float results agree to an ulp)
  {
    const cfloat sg = xg.somme(), sh = x.somme();
    CHECK(std::abs(sg - sh) <= 2e-7f * std::max(1.0f, std::abs(sh)), "somme (complex): %g%+gi vs %g%+gi", sg.real(), sg.imag(), sh.real(), sh.imag());
    CHECK(std::abs(ug.somme() - u.somme()) <= 2e-7f * std::max(1.0f, std::abs(u.somme())), "somme (real): %g vs %g", ug.somme(), u.somme());
    CHECK(std::abs(ug.moyenne() - u.moyenne()) <= 1e-9f + 2e-7f * std::abs(u.moyenne()), "moyenne");
    CHECK(ug.valeur_max() == u.valeur_max() && ug.valeur_min() == u.valeur_min(), "valeur_max / valeur_min");
    CHECK(ug.index_max() == u.index_max(), "index_max: %d vs %d", (int) ug.index_max(), (int) u.index_max());
    Vecf plat = Vecf::zeros(5000);                       // ties: the FIRST maximum, like std::max_element
    plat(1234) = 2.0f;
    plat(4321) = 2.0f;
    CHECK(plat.vers_gpu().index_max() == 1234, "index_max with ties: %d", (int) plat.vers_gpu().index_max());
  }
  threw = false;
  try { (void) xg(0); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw, "element access on a resident vector must be refused");
  // filtfilt = filter, reverse, filter, reverse (filtrage.hpp:1761-1765): now entirely on the device
  const Vecf h = design_rif_fen(63, "lp", 0.1f);
  const Veccf y_h = filtfilt<cfloat>(Design(h), x);
  Veccf y_g;
  {
    ResidenceGpu garde;
    y_g = filtfilt<cfloat>(Design(h), xg);
  }
  CHECK(y_g.est_sur_gpu(), "resident filtfilt left the GPU");
  CHECK(ecart_rel(y_g.vers_hote(), y_h) <= 2e-6f, "resident filtfilt: %g", ecart_rel(y_g.vers_hote(), y_h));
  // y = ifft(fft(x) * H) as a user writes it, resident
  const Veccf H = randcn(4096);
  const Veccf ref = ifft(fft(x.head(4096).clone()) * H);
  Veccf res;
  {
    ResidenceGpu garde;
    res = ifft(fft(xg.head(4096).clone()) * H.vers_gpu());
  }
  CHECK(res.est_sur_gpu() && ecart_rel(res.vers_hote(), ref) <= 2e-6f, "resident spectral product: %g", ecart_rel(res.vers_hote(), ref));
  // the correlation / delay compositions run on the device whatever the caller holds: resident in -> resident out, same values
  {
    const Veccf a = x.head(4096).clone(), b = z.head(4096).clone();
    const auto [lh, ch] = ccorr(a, b);
    const auto [lg, cg] = ccorr(a.vers_gpu(), b.vers_gpu());
    CHECK(!ch.est_sur_gpu() && cg.est_sur_gpu() && ecart_rel(cg.vers_hote(), ch) <= 1e-6f, "ccorr resident vs host: %g", ecart_rel(cg.vers_hote(), ch));
    const Veccf dh = délais(a, 2.5f), dg = délais(a.vers_gpu(), 2.5f);
    CHECK(!dh.est_sur_gpu() && dg.est_sur_gpu() && ecart_rel(dg.vers_hote(), dh) <= 1e-6f, "délais resident vs host: %g", ecart_rel(dg.vers_hote(), dh));
    const Vecf rh = rééchan_freq(u.head(4096).clone(), 2), rg = rééchan_freq(u.head(4096).clone().vers_gpu(), 2);
    CHECK(!rh.est_sur_gpu() && rg.est_sur_gpu() && rg.rows() == 8192 && ecart_rel(rg.vers_hote(), rh) <= 1e-6f, "rééchan_freq resident vs host");
  }
}

static void test_residence_gpu()
{
  const int n = 1 << 18;
  const Veccf x = randcn(n);
  const Vecf h = design_rif_fen(127, "lp", 0.02f);
  // host chain
  const Veccf y_h = filtrer<cfloat>(Design(h), x);
  const Veccf Y_h = fft(y_h);
  const Veccf r_h = rééchan(y_h, 160.0f / 147);
  const Vecf s_h = filtrer<float>(Design(design_riia(6, "lp", "butt", 0.2f)), real(x));
  // resident chain: everything the API allocates inside the guard is device memory
  Veccf y_g, Y_g, r_g;
  Vecf s_g;
  {
    ResidenceGpu garde;
    const Veccf x_g = x.vers_gpu();
    const Vecf xr_g = real(x).vers_gpu();
    y_g = filtrer<cfloat>(Design(h), x_g);
    Y_g = fft(y_g);
    r_g = rééchan(y_g, 160.0f / 147);
    s_g = filtrer<float>(Design(design_riia(6, "lp", "butt", 0.2f)), xr_g);
    CHECK(x_g.est_sur_gpu() && y_g.est_sur_gpu() && Y_g.est_sur_gpu() && r_g.est_sur_gpu() && s_g.est_sur_gpu(), "resident chain left the GPU");
  }
  CHECK(!Veccf(16).est_sur_gpu(), "allocation after the guard must be host memory");
  bool threw = false;
  try { (void) y_g(0); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw, "element access on a resident vector must be refused");
  CHECK(ecart_rel(y_g.vers_hote(), y_h) <= 2e-6f, "resident filtrer: %g", ecart_rel(y_g.vers_hote(), y_h));
  CHECK(ecart_rel(Y_g.vers_hote(), Y_h) <= 2e-6f, "resident fft: %g", ecart_rel(Y_g.vers_hote(), Y_h));
  CHECK(ecart_rel(r_g.vers_hote(), r_h) <= 2e-6f, "resident rééchan: %g (%d vs %d outputs)", ecart_rel(r_g.vers_hote(), r_h), r_g.rows(), r_h.rows());
  CHECK(ecart_rel(s_g.vers_hote(), s_h) <= 2e-6f, "resident filtre_sois: %g", ecart_rel(s_g.vers_hote(), s_h));
  // foreign device memory through map(): the output vector, pre-sized, keeps its storage
  cfloat *dx = (cfloat *) tsd_amd::alloue_gpu((size_t) n * sizeof(cfloat)), *dy = (cfloat *) tsd_amd::alloue_gpu((size_t) n * sizeof(cfloat));
  tsd_amd::copie_vers_gpu(dx, x.data(), (size_t) n * sizeof(cfloat));
  Veccf vx = Veccf::map(dx, n), vy = Veccf::map(dy, n);
  CHECK(vx.est_sur_gpu() && vy.est_sur_gpu(), "map() of device memory");
  auto f = filtre_rif<float, cfloat>(h);
  f->step(vx, vy);
  CHECK(vy.data() == dy, "a pre-sized mapped output must be written in place");
  Veccf back(n);
  tsd_amd::copie_vers_hote(back.data(), dy, (size_t) n * sizeof(cfloat));
  CHECK(ecart_rel(back, y_h) <= 2e-6f, "mapped device vectors through filtre_rif: %g", ecart_rel(back, y_h));
  // the compatibility operators on resident vectors too: filtre_rif_fft, ligne_a_retard, filtre_fft (device response)
  {
    ResidenceGpu garde;
    const Veccf x_g = x.head(8192).vers_gpu();
    const Veccf a = filtre_rif_fft<cfloat>(h)->step(x_g), b = ligne_a_retard<cfloat>(5)->step(x_g);
    CHECK(a.est_sur_gpu() && b.est_sur_gpu() && a.rows() == 8192 && b.rows() == 8192, "compat operators resident");
    const Veccf a_h = [&] { Veccf t; { t = a.vers_hote(); } return t; }();
    (void) a_h;
  }
  const Veccf a_ref = filtre_rif_fft<cfloat>(h)->step(x.head(8192).clone());
  Veccf a_res;
  {
    ResidenceGpu garde;
    a_res = filtre_rif_fft<cfloat>(h)->step(x.head(8192).vers_gpu());
  }
  CHECK(ecart_rel(a_res.vers_hote(), a_ref) <= 2e-6f, "resident filtre_rif_fft: %g", ecart_rel(a_res.vers_hote(), a_ref));
  tsd_amd::libere_gpu(dx);
  tsd_amd::libere_gpu(dy);
}

// Several GPUs behind one operator object (here: 3 logical shards on the devices present): a large host
// vector through filtrer / filtre_sois / rééchan gives what one GPU gives
static void test_fragments()
{
  const int n = (1 << 22) + 12345;
  const Veccf x = randcn(n);
  const Vecf h = design_rif_fen(127, "lp", 0.02f), hc = design_rif_fen(31, "lp", 0.2f);
  const auto butter = design_riia(12, "lp", "butt", 0.25f);
  // (lent: a first-order low-pass at fc = 1e-5, memory of 5e5 samples: sharded exactly -- end states through the host -- not by warm-up halos)
  const auto lent = design_riia(1, "lp", "butt", 1e-5f);
  tsd_amd::fixe_fragments(0);
  const Veccf y1 = filtrer<cfloat>(Design(h), x), yd1 = filtrer<cfloat>(Design(hc), x), s1 = filtrer<cfloat>(Design(butter), x), r1 = rééchan(x, 160.0f / 147);
  const Veccf l1 = filtrer<cfloat>(Design(lent), x);
  tsd_amd::fixe_fragments(3);
  const Veccf y3 = filtrer<cfloat>(Design(h), x), yd3 = filtrer<cfloat>(Design(hc), x), s3 = filtrer<cfloat>(Design(butter), x), r3 = rééchan(x, 160.0f / 147);
  const Veccf l3 = filtrer<cfloat>(Design(lent), x);
  // streaming across calls on ONE sharded object: large call, small call, large call
  auto f = filtre_rif<float, cfloat>(hc);
  Veccf ys(n);
  const int a = 1 << 22, b = a + 100;
  ys.segment(0, a) = f->step(x.segment(0, a).clone());
  ys.segment(a, b - a) = f->step(x.segment(a, b - a).clone());
  ys.segment(b, n - b) = f->step(x.segment(b, n - b).clone());
  tsd_amd::fixe_fragments(-1);
  CHECK(ecart_rel(y3, y1) <= 2e-6f, "sharded filtrer (overlap-save): %g", ecart_rel(y3, y1));
  CHECK(ecart_rel(yd3, yd1) == 0.f, "sharded filtrer (direct) must be bit-exact: %g", ecart_rel(yd3, yd1));
  CHECK(ecart_rel(ys, yd1) == 0.f, "sharded filtre_rif across calls: %g", ecart_rel(ys, yd1));
  CHECK(ecart_rel(s3, s1) <= 1e-6f, "sharded filtre_sois: %g", ecart_rel(s3, s1));
  CHECK(ecart_rel(l3, l1) <= 2e-5f, "sharded filtre_sois (long memory, exact exchange of the end states): %g", ecart_rel(l3, l1));
  CHECK(r3.rows() == r1.rows() && ecart_rel(r3, r1) == 0.f, "sharded rééchan must be bit-exact: %d vs %d outputs, %g", r3.rows(), r1.rows(), ecart_rel(r3, r1));
}

// test_riia (test-filtres.cc:668-679 -> test_design :327-404): the four analog prototypes at order 12, fc = 0.25, 0.1 dB /
// 60 dB: the magnitude at 2048 frequencies of [0, 0.5) stays within 0.1 of 1 over the first 800 bins and within 0.1 of 0
// over the last 800 (mirrored for the high-pass).  Evaluated from the poles and zeros, and -- the data path -- measured
// through filtrer() on the GPU at bins of both bands.
static void test_riia()
{
  auto gain_at = [](const FRat<cfloat> &h, double f) {
    const cdouble w = std::polar(1.0, 2 * π * f);
    cdouble H = cdouble(h.numer.mlt) / cdouble(h.denom.mlt);
    for (int i = 0; i < h.numer.coefs.rows(); i++) H *= (w - cdouble(h.numer.coefs(i)));
    for (int i = 0; i < h.denom.coefs.rows(); i++) H /= (w - cdouble(h.denom.coefs(i)));
    return std::abs(H);
  };
  const int npts = 2048;
  for (const char *type : {"lp", "hp"})
    for (const char *proto : {"ellip", "butt", "cheb1", "cheb2"}) {
      const bool hp = type[0] == 'h';
      const FRat<cfloat> h = design_riia(12, type, proto, 0.25f, 0.1f, 60);
      CHECK(h.numer.mode_racines && h.denom.mode_racines && h.numer.coefs.rows() == 12 && h.denom.coefs.rows() == 12, "design_riia(%s, %s): 12 zeros and poles", type, proto);
      double emax_bp = 0, emax_bc = 0, rmax = 0;
      for (int i = 0; i < 12; i++) rmax = std::max(rmax, (double) std::abs(h.denom.coefs(i)));
      for (int k = 0; k < npts; k++) {
        const double g = gain_at(h, 0.5 * k / npts);
        const bool bande_passante = hp ? k >= npts - 800 : k < 800, bande_coupee = hp ? k < 800 : k >= npts - 800;
        if (bande_passante) emax_bp = std::max(emax_bp, std::abs(g - 1));
        if (bande_coupee) emax_bc = std::max(emax_bc, g);
      }
      CHECK(rmax < 1 && emax_bp <= 0.1 && emax_bc <= 0.1, "design_riia(12, %s, %s): template missed (pass band %g, stop band %g, pole radius %g)", type, proto, emax_bp, emax_bc, rmax);
      // through the GPU cascade: one bin of each band
      const int n = 1 << 15;
      for (double f : {hp ? 0.45 : 0.05, hp ? 0.05 : 0.45}) {
        Veccf x(n);
        for (int i = 0; i < n; i++) x(i) = std::polar(1.0f, (float) (2 * π * f * i));
        const Veccf y = filtrer<cfloat>(Design(h), x);
        double m = 0;
        for (int i = n / 2; i < n; i++) m += std::abs(y(i));
        m /= n / 2;
        CHECK(std::abs(m - gain_at(h, f)) <= 2e-3, "design_riia(%s, %s) through filtrer at f = %g: %g vs %g", type, proto, f, m, gain_at(h, f));
      }
    }
  // orders around the odd / even cases of every prototype stay stable and keep a unit DC gain
  for (int n : {1, 2, 3, 5, 8})
    for (const char *proto : {"ellip", "butt", "cheb1", "cheb2"}) {
      const FRat<cfloat> h = design_riia(n, "lp", proto, 0.2f, 0.5f, 40);
      double rmax = 0;
      for (int i = 0; i < h.denom.coefs.rows(); i++) rmax = std::max(rmax, (double) std::abs(h.denom.coefs(i)));
      CHECK(rmax < 1 && std::abs(gain_at(h, 0) - 1) < 1e-3, "design_riia(%d, lp, %s): DC gain %g, pole radius %g", n, proto, gain_at(h, 0), rmax);
    }
}

// design_biquad (test-filtres.cc:296-314 only plots): every type through filtrer() -- coefficient form,
// factorised, two-pole section on the GPU -- against the biquad's own frequency response
static void test_design_biquad()
{
  const int n = 1 << 15;
  for (const char *type : {"lp", "hp", "bp", "notch", "res", "plateau-bf", "plateau-hf"}) {
    const FRat<float> h = design_biquad(type, 0.1f, 0.9f, 6.0f);
    const FRat<float> hz = h.eval_inv_z();                       // powers of z^-1
    const Vecf b = hz.numer.coefs, a = hz.denom.coefs;
    CHECK(b.rows() == 3 && a.rows() == 3 && std::abs(a(0) - 1) < 1e-6f, "design_biquad(%s): %d / %d coefficients", type, b.rows(), a.rows());
    for (float f : {0.02f, 0.1f, 0.3f}) {
      Veccf x(n);
      for (int i = 0; i < n; i++) x(i) = std::polar(1.0f, (float) (2 * π * f * i));
      const Veccf y = filtrer<cfloat>(Design(h), x);
      const cfloat w = std::polar(1.0f, (float) (-2 * π * f));
      const cfloat H = (b(0) + b(1) * w + b(2) * w * w) / (a(0) + a(1) * w + a(2) * w * w);
      double m = 0;
      for (int i = n / 2; i < n; i++) m += std::abs(y(i));
      m /= n / 2;
      // (the coefficient form is factorised into float roots: a notch keeps its zeros on the unit circle to ~1e-4)
      CHECK(std::abs((float) m - std::abs(H)) <= 1e-3f * std::max(1.0f, std::abs(H)), "design_biquad(%s) at f = %g: gain %g, H = %g", type, f, m, std::abs(H));
    }
  }
}

static void test_tab()   // test-tab.cc:53-137 semantics
{
  Vecf a = linspace(0, 9, 10);
  Vecf b = a;                 // deep copy
  b(0) = 42;
  CHECK(a(0) == 0, "copy must be deep");
  Vecf v = a.segment(2, 3);   // view
  v(0) = -1;
  CHECK(a(2) == -1, "segment must alias its parent");
  a.tail(2) = Vecf::valeurs({7, 8});
  CHECK(a(8) == 7 && a(9) == 8, "assignment into a view copies elements");
  Vecf c = std::move(b);
  CHECK(c(0) == 42 && b.rows() == 0, "move steals");
  float raw[3] = {1, 2, 3};
  Vecf m = Vecf::map(raw, 3);
  m(1) = 5;
  CHECK(raw[1] == 5, "map wraps foreign memory");
  bool threw = false;
  try { a(10) = 0; } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw, "out-of-range access must throw through échec");
  CHECK(prochaine_puissance_de_2(3) == 4 && prochaine_puissance_de_2(65535) == 65536 && prochaine_puissance_de_2(1) == 1, "pp2");
}

static void test_filtre_rif()   // test-filtres.cc:479-511
{
  int nc = 31, n = nc + 50;
  Vecf h = linspace(1, nc, nc);
  auto f = filtre_rif<float, float>(h);
  Vecf y = f->step(sigimp(n));
  CHECK(y.rows() == n, "n=%d rows=%d", n, y.rows());
  Vecf verr = y - vconcat(h, Vecf::zeros(n - nc));
  CHECK(maxabs(verr) <= 1e-7f, "err=%g", maxabs(verr));
}

static void test_retard(int d)   // test-filtres.cc:450-476
{
  Vecf h = Vecf::ones(d), x = sigimp(20, 3);
  Vecf y = filtrer<float>(h, x), y2 = filtrer<float>(h, y);
  CHECK(y2.index_max() == 3 + d - 1, "d=%d index=%d", d, y2.index_max());
}

static void test_design_rif_prod(int n1, int n2)   // test-filtres.cc:33-71
{
  Vecf h1 = randn(n1), h2 = randn(n2), hp = design_rif_prod(h1, h2);
  CHECK(hp.rows() == n1 + n2 - 1, "rows");
  Vecf x = sigimp(n1 + n2);
  Vecf y2 = filtrer<float>(h2, filtrer<float>(h1, x)), yp = filtrer<float>(hp, x);
  CHECK(maxabs(y2 - yp) < 1e-5f, "n1=%d n2=%d err=%g", n1, n2, maxabs(y2 - yp));
}

static void test_rif_vs_rif_fft()   // test-filtres.cc:514-554
{
  Vecf h = design_rif_fen(127, "lp", 0.02f);
  Vecf x = randn(5000);
  auto f1 = filtre_rif<float, float>(h);
  auto f2 = filtre_rif_fft<float>(h);
  Vecf y1 = filtre_par_bloc<float>(f1, x, 1000), y2 = filtre_par_bloc<float>(f2, x, 1000);
  // the OLA filter's output is the direct one delayed by Nz - M = 385 samples
  const int d = 385;
  float err = 0;
  for (int i = 0; i + d < 5000 - 4 * 127; i++) err = std::max(err, std::abs(y2(i + d) - y1(i)));
  CHECK(err <= 1e-6f, "err=%g", err);
  CHECK(maxabs(y2.head(d)) == 0.f, "the first %d outputs of filtre_rif_fft are the delay line's zeros", d);
  // complex instantiation: real part only (fourier.cc:976)
  Veccf xc = randcn(3000);
  Veccf yc = filtre_rif_fft<cfloat>(h)->step(xc);
  CHECK(maxabs(imag(yc)) == 0.f, "filtre_rif_fft<cfloat> keeps only the real part, like the reference");
}

static void test_fir_vs_oracle()
{
  Vecf h = design_rif_fen(127, "lp", 0.02f);
  Veccf x = randcn(20000);
  Veccf y = filtrer<cfloat>(h, x);
  std::vector<orc_cf> fen(127, orc_cf{0, 0}), yr(20000);
  int idx = 0;
  orc_fir_cf(h.data(), 127, fen.data(), &idx, (const orc_cf *) x.data(), yr.data(), 20000);
  float err = 0, ref = 0;
  for (int i = 0; i < 20000; i++) {
    err = std::max(err, std::abs(y(i) - cfloat(yr[i].re, yr[i].im)));
    ref = std::max(ref, std::abs(cfloat(yr[i].re, yr[i].im)));
  }
  CHECK(err <= 1e-5f * ref, "filtrer(Veccf) vs oracle: err=%g ref=%g", err, ref);
  // design parity with the oracle's restatement of design_rif_fen
  std::vector<float> ho(127);
  orc_design_rif_fen_hann(127, 0, 0.02f, ho.data());
  float dh = 0;
  for (int i = 0; i < 127; i++) dh = std::max(dh, std::abs(ho[i] - h(i)));
  CHECK(dh <= 1e-7f, "design_rif_fen vs oracle: %g", dh);
  // dsp:: spelling reaches the same objects (test-dsp.cc is a compile check)
  dsp::Vecf h2 = dsp::filter::design_fir_wnd(127, "lp", 0.02f);
  dsp::Veccf y2 = dsp::filter::filter<cfloat>(h2, x);
  CHECK(maxabs(y2 - y) == 0.f, "dsp::filter::filter == tsd::filtrage::filtrer");
}

static void test_sois()   // design_riia(12,"lp","butt",0.25) -> 6 sections (test-filtres.cc:668-679)
{
  auto h = design_riia(12, "lp", "butt", 0.25f);
  Vecf x = randn(100000);
  Vecf y = filtrer<float>(Design(h), x);
  // oracle: same design, same pairing, sequential recurrence
  std::vector<orc_cf> z(12), p(12);
  orc_cf mn, md;
  orc_design_butter_lp(12, 0.25f, z.data(), p.data(), &mn, &md);
  orc_sos s;
  orc_sos_from_zpk(&s, z.data(), p.data(), 12, mn, md, 2);
  CHECK(s.nsec == 6, "nsec=%d", s.nsec);
  orc_sos_state_f st;
  orc_sos_state_init_f(&st);
  std::vector<float> yr(100000);
  orc_sos_step_f(&s, &st, x.data(), yr.data(), 100000);
  float err = 0, ref = 0;
  for (int i = 0; i < 100000; i++) { err = std::max(err, std::abs(y(i) - yr[i])); ref = std::max(ref, std::abs(yr[i])); }
  CHECK(err <= 1e-5f * ref, "filtrer(design_riia) vs oracle: err=%g ref=%g", err, ref);
  // streaming through the FiltreGen interface
  auto f = filtre_sois<float>(h);
  Vecf yb = filtre_par_bloc<float>(f, x, 311);
  CHECK(maxabs(yb - y) <= 1e-5f * ref, "chunked SOS differs: %g", maxabs(yb - y));
}

static void test_fft_valide(int n, bool inv)   // test-fourier.cc:181-272
{
  Veccf x = randcn(n);
  Veccf X = inv ? ifft(x) : fft(x);
  CHECK(X.rows() == n, "rows");
  float err = 0;
  for (int i = 0; i < n; i++) {
    cdouble s = 0;
    const double signe = inv ? 1 : -1;
    for (int k = 0; k < n; k++) s += cdouble(x(k)) * std::polar(1.0, signe * (2 * π * ((long) k * i % n)) / n);
    s /= std::sqrt((double) n);
    err = std::max(err, (float) std::abs(cdouble(X(i)) - s));
  }
  CHECK(err < 1e-2f, "n=%d inv=%d err=%g", n, (int) inv, err);
  CHECK(err < ((n & 1) && n > 1 ? 2e-4f : 2e-5f) * std::sqrt((float) n) + 1e-6f, "n=%d inv=%d err=%g (tight)", n, (int) inv, err);
}

static void test_fft_misc()
{
  // test_fft (test-fourier.cc:275-312): ifft(fft(x)) rms error
  int n = 1024;
  Vecf x = Vecf::int_expr(n, [&](int i) { return std::cos(8 * 2 * π * i / (n - 1)); });
  Veccf X = fft(x);
  Vecf x2 = real(ifft(X));
  double e = 0;
  for (int i = 0; i < n; i++) e += (x2(i) - x(i)) * (x2(i) - x(i));
  CHECK(std::sqrt(e / n) <= 5e-6, "rfft + ifft rms err=%g", std::sqrt(e / n));
  // rfft == fft of the widened signal (test_fft_valide<float,cfloat>)
  for (int m : {16, 2, 4, 8, 10, 128, 1024, 17, 5, 3}) {
    Vecf xr = randn(m);
    Veccf A = rfft(xr), B = fft(Veccf(xr.as_complex()));
    CHECK(maxabs(A - B) <= 2e-5f * std::sqrt((float) m) * 4, "rfft(%d) err=%g", m, maxabs(A - B));
  }
  // test_fftplan (test-fourier.cc:6-37): plan == fft()
  for (int m : {8, 16, 18, 19, 101}) {
    Veccf xc = randcn(m);
    auto plan = tfrplan_création(m);
    CHECK(maxabs(plan->step(xc) - fft(xc)) <= 1e-6f, "plan(%d)", m);
  }
  // test_fftshift (test-fourier.cc:39-72)
  for (int m : {15, 16}) {
    Vecf y = fftshift(linspace(0, m - 1, m));
    int h = m / 2;
    Vecf yref(m);
    if (m & 1) { yref.head(h) = linspace(h + 1, m - 1, h); yref.tail(h + 1) = linspace(0, h, h + 1); }
    else { yref.head(h) = linspace(h, m - 1, h); yref.tail(h) = linspace(0, h - 1, h); }
    CHECK(maxabs(y - yref) == 0.f, "fftshift(%d)", m);
  }
  // the plug point: a user-installed factory is what fft() uses (fourier.cc:469-481)
  struct Compte : FFTPlan {
    sptr<FFTPlan> inner;
    int *cnt;
    void configure(entier n, bouléen a, bouléen no) override { inner->configure(n, a, no); }
    void step(const Veccf &x, Veccf &y, bouléen a) override { (*cnt)++; inner->step(x, y, a); }
  };
  auto sauvegarde = fftplan_defaut;
  int cnt = 0;
  fftplan_defaut = [&]() -> sptr<FFTPlan> { auto p = std::make_shared<Compte>(); p->inner = sauvegarde(); p->cnt = &cnt; return p; };
  (void) fft(randcn(64));
  fftplan_defaut = sauvegarde;
  CHECK(cnt == 1, "fftplan_defaut hook not consulted (cnt=%d)", cnt);
}

static void test_reechan()   // test-ra.cc:55-160 style checks on rééchan / filtre_reechan
{
  for (float ratio : {1.0f, 1.5f, 0.5f, 1.2f, 160.f / 147.f}) {
    const float fe = 100e3f, f2 = 2e3f;
    Vecf x = Vecf::int_expr(1000, [&](int i) { return std::sin(2 * π * f2 * i / fe); });
    Vecf y = rééchan(x, ratio);
    const double err = 100.0 * std::abs((y.rows() - ratio * x.rows()) / x.rows());
    CHECK(err < 1, "ratio=%g rows=%d", ratio, y.rows());
    const float amp1 = x.valeur_max() - x.valeur_min(), amp2 = y.valeur_max() - y.valeur_min();
    CHECK(100 * (amp1 - amp2) / amp1 < 10, "ratio=%g amplitude %g -> %g", ratio, amp1, amp2);
    // exact parity with the oracle (count and samples)
    if (ratio != 1.0f) {
      int nd, nu; float post, fcut;
      orc_reechan_config(ratio, &nd, &nu, &post, &fcut);
      std::vector<float> lut(257 * 15), yr(2100);
      orc_itrp_sinc_lut(15, 256, fcut, lut.data());
      orc_ra r;
      orc_ra_init(&r, post, 15, 256, lut.data());
      const int64_t no = orc_ra_step_f(&r, x.data(), 1000, yr.data());
      CHECK(no == y.rows(), "ratio=%g: %lld outputs vs oracle %d", ratio, (long long) no, y.rows());
      float e = 0;
      for (int i = 0; i < std::min<int64_t>(no, y.rows()); i++) e = std::max(e, std::abs(y(i) - yr[i]));
      CHECK(e <= 1e-5f, "ratio=%g err=%g", ratio, e);
    }
  }
  // ratios outside [0.5,2): half-band decimators / x2 upsamplers around the interpolator
  for (float ratio : {2.0f, 3.14159265f, 0.25f, 0.2f, 4.0f}) {
    const float fe = 100e3f, f2 = 2e3f;
    Vecf x = Vecf::int_expr(4000, [&](int i) { return std::sin(2 * π * f2 * i / fe); });
    Vecf y = rééchan(x, ratio);
    CHECK(100.0 * std::abs((y.rows() - ratio * x.rows()) / x.rows()) < 1, "ratio=%g rows=%d", ratio, y.rows());
    const float amp1 = x.valeur_max() - x.valeur_min(), amp2 = y.valeur_max() - y.valeur_min();
    CHECK(100 * (amp1 - amp2) / amp1 < 10, "ratio=%g amplitude %g -> %g", ratio, amp1, amp2);
  }
  // dsp::resample spelling
  dsp::Veccf xc = randcn(2000);
  CHECK(dsp::resample(xc, 1.25f).rows() == rééchan(xc, 1.25f).rows(), "dsp::resample");
  // dsp::filter interpolators and filter_itrp (dsp/filter.hpp:1755-1805,1910): the same objects as the French names; itrp_sinc
  // takes (ncoefs, fcut, window) -- the reference's own forwarder does not compile once instantiated, this one builds the structure
  {
    auto fa = dsp::filter::filter_itrp<cfloat>(1.25f, dsp::filter::itrp_sinc<cfloat>(15, 0.4f, "hn"));
    auto fb = filtre_itrp<cfloat>(1.25f, itrp_sinc<cfloat>({15, 256, 0.4f, "hn"}));
    Veccf ya = fa->step(xc), yb = fb->step(xc);
    CHECK(ya.rows() == yb.rows() && maxabs(ya - yb) == 0.f, "dsp::filter::filter_itrp(itrp_sinc(15, 0.4, hn)) == filtre_itrp(itrp_sinc{15,256,0.4,hn})");
    Veccf yc = dsp::filter::filter_itrp<cfloat>(0.8f)->step(xc), yd = filtre_itrp<cfloat>(0.8f, itrp_cspline<cfloat>())->step(xc);
    CHECK(yc.rows() == yd.rows() && maxabs(yc - yd) == 0.f, "dsp::filter::filter_itrp default interpolator = itrp_cspline");
    CHECK(dsp::filter::itrp_linear<float>()->K == 2 && dsp::filter::itrp_lagrange<float>(3)->K == 4 && dsp::filter::itrp_cspline<float>()->K == 4, "dsp interpolator lengths");
  }
}

static void test_polyphase()
{
  // test_decimateur (test-filtres.cc:186-200)
  auto dec = decimateur<float>(3);
  Vecf y = filtre_par_bloc<float>(dec, linspace(0, 89, 90), 4), xr = linspace(0, 87, 30);
  CHECK(y.rows() == 30 && maxabs(y - xr) == 0.f, "decimateur");
  // half-band / decim / ups against the oracle
  Vecf h = design_rif_fen(15, "lp", 0.25f, "hn");
  Vecf x = randn(6000);
  {
    std::vector<float> fen(15, 0.f), yr(3100);
    int idx = 0, cnt = 0;
    const int64_t no = orc_polydecim_f(1, h.data(), 15, 2, fen.data(), &idx, &cnt, x.data(), 6000, yr.data());
    Vecf yg = filtre_par_bloc<float>(filtre_rif_demi_bande<float, float>(h), x, 501);
    float e = 0;
    for (int i = 0; i < std::min<int64_t>(no, yg.rows()); i++) e = std::max(e, std::abs(yg(i) - yr[i]));
    CHECK(no == yg.rows() && e <= 1e-5f, "demi-bande: %lld vs %d outputs, err %g", (long long) no, yg.rows(), e);
  }
  {
    std::vector<float> pad(17), fen(8, 0.f), yr(12000);
    const int K = orc_ups_prepare(h.data(), 15, 2, pad.data());
    int idx = 0;
    const int64_t no = orc_ups_f(pad.data(), K, 2, fen.data(), &idx, x.data(), 6000, yr.data());
    Vecf yg = filtre_rif_ups<float, float>(h, 2)->step(x);
    float e = 0;
    for (int i = 0; i < std::min<int64_t>(no, yg.rows()); i++) e = std::max(e, std::abs(yg(i) - yr[i]));
    CHECK(no == yg.rows() && e <= 1e-5f, "ups: %lld vs %d outputs, err %g", (long long) no, yg.rows(), e);
    CHECK(filtre_rif_ups_délais(15, 2) == 8.0f && rif_delais(15) == 7.0f, "delays");
  }
  // forme_polyphase is a pure permutation (identity on memory, zero padded)
  auto X = forme_polyphase(linspace(0, 9, 10), 4);
  CHECK(X.lignes == 4 && X.colonnes == 3 && X(1, 2) == 9.f && X(3, 2) == 0.f, "forme_polyphase");
  CHECK(iforme_polyphase(X).rows() == 12, "iforme_polyphase");
  // test_filtre_rii (test-filtres.cc:556-606)
  const float a = 0.1f;
  auto f = filtre_rii<float, float>(FRat<float>::rii(Vecf::valeurs({a}), Vecf::valeurs({1.0f, -(1 - a)})));
  Vecf yy = f->step(Vecf::ones(20)), yref(20);
  yref(0) = a;
  for (int i = 1; i < 20; i++) yref(i) = yref(i - 1) + a * (1 - yref(i - 1));
  CHECK(maxabs(yref - yy) <= 1e-6f, "filtre_rii err=%g", maxabs(yref - yy));
}

// test_xcorr (test-fourier.cc:477-560): FFT correlations against the O(n^2) definition
static void test_xcorr(bool biais)
{
  for (int n : {1, 2, 3, 10, 15, 16, 21, 32}) {
    Veccf a1 = randcn(n), a2 = randcn(n);
    auto [lags, c] = biais ? xcorrb(a1, a2) : xcorr(a1, a2);
    CHECK(lags.rows() == 2 * n - 1 && c.rows() == 2 * n - 1, "xcorr dims n=%d", n);
    const int m = n;
    Veccf cref(2 * m - 1);
    for (int i = 0; i < m; i++) {
      cfloat s = 0; int ns = 0;
      for (int k = 0; k + i < n; k++) { s += a1(k + i) * std::conj(a2(k)); ns++; }
      s /= (float) (biais ? n : ns);
      cref(m - 1 - i) = s;
    }
    for (int i = 1; i < m; i++) {
      cfloat s = 0; int ns = 0;
      for (int k = 0; k + i < n; k++) { s += a1(k) * std::conj(a2(k + i)); ns++; }
      s /= (float) (biais ? n : ns);
      cref(i + m - 1) = s;
    }
    for (int i = 0; i < 2 * n - 1; i++) CHECK(lags(i) == (float) (i - (n - 1)), "lags n=%d", n);
    CHECK(maxabs(c - cref) <= 2e-5f * std::max(1.0f, maxabs(cref)), "xcorr%s n=%d err=%g", biais ? "b" : "", n, maxabs(c - cref));
  }
}

// test_reechan (test-fourier.cc:122-159): x2 by spectrum padding then every other sample == x
// czt (fourier.cc:1347-1389): no reference test holds a value ("parity unpinned"); the check is the reference's own statements
// evaluated in double with a direct O(N^2) circular convolution in place of ifft(fft * fft) (unitary transforms: the product of
// the two spectra comes back as the circular convolution over sqrt(N))
static void test_czt()
{
  for (int n : {8, 13, 64, 500}) {
    const int m = n, nm = n, N = 2 * m - 1;
    Veccf x = randcn(n);
    const cfloat W = std::polar(0.999f, -2 * π_f / n * 0.7f), z0 = std::polar(0.98f, 0.3f);
    Veccf y = czt(x, m, W, z0);
    std::vector<std::complex<double>> h(2 * nm - 1), g(N, 0.0), hc(N);
    for (int i = 0; i < nm; i++) h[i] = (std::complex<double>) std::pow(W, -0.5f * i * i);     // (the float powers of the reference)
    for (int i = nm; i < 2 * nm - 1; i++) h[i] = h[nm - 1 - (i - nm)];
    for (int i = 0; i < n; i++) g[i] = (std::complex<double>) (x(i) * std::pow(z0, (float) -i)) / h[nm + i - 1];
    for (int i = 0; i < m; i++) hc[i] = h[nm - 1 + i];
    for (int i = 0; i < n - 1; i++) hc[m + i] = h[nm - n + i];
    double err = 0, ref = 0;
    for (int k = 0; k < m; k++) {
      std::complex<double> s = 0;
      for (int j = 0; j < N; j++) s += hc[j] * g[((k - j) % N + N) % N];
      const std::complex<double> want = s / std::sqrt((double) N) / h[nm - 1 + k];
      err = std::max(err, std::abs(want - (std::complex<double>) y(k)));
      ref = std::max(ref, std::abs(want));
    }
    CHECK(y.rows() == m && err <= 2e-5 * ref, "czt n = m = %d: err %g of %g", n, err, ref);
  }
  bool jete = false;
  try { (void) czt(randcn(8), 12, cfloat(1, 0)); } catch (const std::exception &) { jete = true; }
  CHECK(jete, "czt with n != m fails like the reference (sizes m + n - 1 and 2m - 1 multiplied)");
  CHECK(dsp::fourier::czt(randcn(4), 4, cfloat(1, 0)).rows() == 4, "dsp::fourier::czt");
}

static void test_reechan_freq()
{
  const int n = 16;
  Vecf x = linspace(0, 1 - 1.0f / n, n), x1 = rééchan_freq(x, 2);
  CHECK(x1.rows() == 2 * n, "rééchan_freq rows=%d", x1.rows());
  Vecf x1b = sousech(x1, 2);
  CHECK(x1b.rows() == n && maxabs(x1b - x) < 1e-5f, "rééchan_freq err=%g", maxabs(x1b - x));
  Vecf u = surech(Vecf::valeurs({1, 2, 3}), 2);
  CHECK(u.rows() == 6 && u(2) == 2 && u(1) == 0, "surech");
}


// ---- délais / estimation_délais / aligne_entier (ports of core/tests/test-fourier.cc:314-472,
// 609-635 and test-tsd.cc:189-208) -------------------------------------------------------------
static Vecf signal_test(int n = 15 * 1024)
{
  Vecf x0(n);
  Vecf fen = fenêtre("hn", n / 2);
  const float periode = 60e6f / 100e3f;
  for (int i = 0; i < n; i++) {
    float w = 0.0f;
    if (i >= n / 4 && i < n - n / 4) w = fen(i - n / 4);
    x0(i) = w * (float) std::sin((2.0 * π * i) / periode);
  }
  return x0;
}
template <typename T> static void test_delais_fractionnaire(float d)
{
  Vecteur<T> x0 = signal_test().as<T>();
  if constexpr (std::is_same<T, cfloat>::value) x0 *= std::polar(1.0f, -π_f / 4);
  Vecteur<T> x1 = délais(x0, d);
  const int n = x0.rows();
  CHECK(x1.rows() == n, "rows");
  const int di = (int) d;                                         // segment lengths truncate like the reference
  float em = 0;
  if (d >= 0)
    for (int i = 0; i < n - di; i++) em = std::max(em, (float) std::abs(x1(di + i) - x0(i)));
  else
    for (int i = 0; i < n + di; i++) em = std::max(em, (float) std::abs(x1(i) - x0(i - di)));
  CHECK(em < (std::floor(d) == d ? 1e-2f : 1e-1f), "delay %g (%s): err %g", d, est_complexe<T>() ? "cfloat" : "float", em);
  if (std::floor(d) != d && d > 0) {
    // sharper than the reference's bound: two half delays compose into the whole one, and the
    // real and complex code paths (packed real FFT vs full FFT) agree
    Vecteur<T> h2 = délais(délais(x0, d / 2), d / 2);
    h2 -= x1;
    CHECK(maxabs(abs(h2)) < 2e-3f, "delay %g: d/2 twice differs from d by %g", d, maxabs(abs(h2)));
    if constexpr (!est_complexe<T>()) {
      Vecf xr = real(délais(x0.as_complex(), d));
      xr -= x1;
      CHECK(maxabs(xr) < 2e-3f, "delay %g: real path differs from complex path by %g", d, maxabs(xr));
    }
  }
}
static void test_delais_unitaire(float vrai, int N)
{
  Veccf x0 = signal_test(N).as<cfloat>();
  Veccf x1 = délais(x0, vrai);
  x0 *= cfloat(7, 0);                                             // the score must not depend on the norms
  x1 *= cfloat(4, 0);
  auto [d, score] = estimation_délais(x0, x1);
  const float tol_pos = N == 32 ? 0.1f : 0.02f;
  CHECK(std::abs(d - vrai) < tol_pos, "N=%d delay %g estimated %g", N, vrai, d);
  CHECK(std::abs(score - 1) < 0.4f, "N=%d delay %g score %g", N, vrai, score);
}
static void test_align_entier()
{
  for (int dref = -50; dref <= 50; dref += 25) {
    Vecf x = signal_test();
    const int n = x.rows();
    Vecf y = Vecf::zeros(n);
    if (dref >= 0)
      y.tail(n - dref) = x.head(n - dref);
    else
      y.head(n + dref) = x.tail(n + dref);
    auto [x2, y2, d, s] = aligne_entier(x, y);
    CHECK(x2.rows() == y2.rows(), "aligne_entier rows");
    Vecf e = x2.clone();
    e -= y2;
    CHECK(d == dref, "aligne_entier d=%d expected %d", d, dref);
    CHECK(std::abs(s - 1) < 1e-3f, "aligne_entier score %g", s);
    CHECK(maxabs(e) < 1e-3f, "aligne_entier err %g", maxabs(e));
  }
  // complex instantiation, unequal lengths
  Veccf xc = signal_test(4096).as<cfloat>();
  xc *= std::polar(1.0f, 0.3f);
  Veccf yc = délais(xc, 17.f).head(4000).clone();
  auto [xa, ya, d, s] = aligne_entier(xc, yc);
  CHECK(d == 17 && xa.rows() == ya.rows() && xa.rows() == 4000 - 17, "aligne_entier<cfloat> d=%d rows=%d", d, xa.rows());
}
static void test_pad_zeros()
{
  Vecf x1 = linspace(0, 4, 5), x2 = linspace(0, 3, 4);
  auto [a, b] = pad_zeros(x1, x2);
  CHECK(a.rows() == 5 && b.rows() == 5 && b(3) == 3 && b(4) == 0 && a(4) == 4, "pad_zeros");
  auto [a2, b2] = pad_zeros(x1, x2, true);
  CHECK(a2.rows() == 8 && b2.rows() == 8 && a2(4) == 4 && a2(5) == 0 && b2(3) == 3 && b2(4) == 0, "pad_zeros p2");
}


// ---- tampon_création (port of core/tests/test-tsd.cc:479-497) and filtre_fft ---------------------
static void test_tampon()
{
  int cnt = 0;
  const int n = 16 * 512;
  Vecf X = randn(n);
  auto t = tampon_création<float>(512, [&](const Vecf &x) {
    CHECK(x.rows() == 512, "tampon block size %d", x.rows());
    Vecf e = x.clone();
    e -= X.segment(cnt, 512);
    CHECK(maxabs(e) == 0, "tampon block content");
    cnt += 512;
  });
  t->step(X.head(100));
  t->step(X.segment(100, 1000));
  t->step(X.tail(n - 100 - 1000));
  CHECK(cnt == n, "tampon delivered %d of %d", cnt, n);
}
// The reference's own test of filtre_fft only plots (core/tests/test-filtre-fft.cc): the checks
// below are the properties its algorithm implies (fourier.cc:837-932) -- "parity unpinned".
static void test_filtre_fft()
{
  // (1) no window, no zeros, identity processing: the blocks come back unchanged, one block late
  //     (a block is delivered when the next one has been overlapped onto it, fourier.cc:870-872)
  {
    FiltreFFTConfig c;
    c.dim_blocs_temporel = 512;
    c.traitement_freq = [](Veccf &) {};
    auto [ola, N] = filtre_fft(c);
    CHECK(N == 512, "N=%d", N);
    Veccf x = randcn(4096 + 100);
    Veccf y = ola->step(x.head(700));                        // 1 block out, 188 samples wait
    Veccf y2 = ola->step(x.tail(x.rows() - 700));
    CHECK(y.rows() == 512 && y2.rows() == 4096 - 512, "rows %d %d", y.rows(), y2.rows());
    Veccf all = vconcat(y, y2);
    CHECK(maxabs(abs(all.head(512))) == 0, "the first delivered block is the (zero) initial carry");
    Veccf e = all.tail(4096 - 512).clone();
    e -= x.head(4096 - 512);
    CHECK(maxabs(abs(e)) < 1e-5f, "identity OLA err %g", maxabs(abs(e)));
  }
  // (2) FIR by spectral product, H = fft(h placed at the tail) * sqrt(N) as FiltreFFTRIF builds it
  //     (fourier.cc:946-978): the direct convolution delayed by Ne - M samples
  {
    const int M = 127, Ne = 512;
    Vecf h = design_rif_fen(M, "lp", 0.02f);
    FiltreFFTConfig c;
    c.dim_blocs_temporel = Ne;
    c.nb_zeros_min = M;
    Veccf H;
    int calls = 0;
    c.traitement_freq = [&](Veccf &X) {
      X *= H;
      calls++;
    };
    auto [ola, N] = filtre_fft(c);
    CHECK(N == 1024, "N=%d", N);
    Veccf h2 = Veccf::zeros(N);
    h2.tail(M) = h.as_complex();
    H = fft(h2);
    H *= cfloat(std::sqrt((float) N), 0);
    const int n = 8 * Ne;
    Veccf x = randcn(n);
    Veccf y = ola->step(x);
    CHECK(y.rows() == n && calls == 8, "rows %d calls %d", y.rows(), calls);
    float err = 0, ref_max = 0;
    const int d = Ne - M;
    for (int i = 0; i < n; i++) {
      cfloat acc = 0;
      for (int m = 0; m < M; m++) {
        const int j = i - d - m;
        if (j >= 0) acc += h(m) * x(j);
      }
      err = std::max(err, (float) std::abs(acc - y(i)));
      ref_max = std::max(ref_max, (float) std::abs(acc));
    }
    CHECK(err <= 1e-5f * std::max(ref_max, 1.0f), "OLA FIR err %g (max %g)", err, ref_max);
    // (2b) the same product as the device-side response (extension tsd_amd::filtre_fft_reponse):
    //      no callback, nothing crosses PCIe but x and y; ragged calls; and both forms chained
    dsp::fourier::FFTFilterConfig cd;
    cd.time_blocks_length = Ne;
    cd.minimum_zeros_count = M;
    CHECK(tsd_amd::filtre_fft_dim(cd) == N, "filtre_fft_dim %d", tsd_amd::filtre_fft_dim(cd));
    auto [old, Nd] = tsd_amd::filtre_fft_reponse(cd, H);
    Veccf yd1 = old->step(x.head(1000));
    Veccf yd2 = old->step(x.tail(n - 1000));
    Veccf yd = vconcat(yd1, yd2);
    CHECK(Nd == N && yd1.rows() == Ne && yd.rows() == n, "device response rows %d + %d", yd1.rows(), yd2.rows());
    float ed = 0;
    for (int i = 0; i < yd.rows(); i++) ed = std::max(ed, (float) std::abs(yd(i) - y(i)));
    CHECK(ed <= 2e-6f * std::max(ref_max, 1.0f), "device-side response vs callback: %g", ed);
    int calls2 = 0;
    cd.freq_domain_processing = [&](Veccf &X) { X *= cfloat(2, 0); calls2++; };
    auto [ol2, N2] = tsd_amd::filtre_fft_reponse(cd, H);
    Veccf y2 = ol2->step(x);
    float e2 = 0;
    for (int i = 0; i < n; i++) e2 = std::max(e2, (float) std::abs(y2(i) - 2.0f * y(i)));
    CHECK(N2 == N && calls2 == 8 && e2 <= 4e-6f * std::max(ref_max, 1.0f), "response then callback: %g (%d calls)", e2, calls2);
    bool threw = false;
    try { FiltreFFTConfig cb; cb.dim_blocs_temporel = Ne; tsd_amd::filtre_fft_reponse(cb, Veccf::zeros(100)); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw, "a response of the wrong size must be refused");
  }
  // (3) Hann window, 1/2 overlap, identity processing: half the input, Ne/2 late, first block dropped
  {
    const int Ne = 256;
    FiltreFFTConfig c;
    c.dim_blocs_temporel = Ne;
    c.avec_fenetrage = true;
    c.traitement_freq = [](Veccf &) {};
    auto [ola, N] = filtre_fft(c);
    const int n = 10 * Ne;
    Veccf x = randcn(n);
    Veccf ya = ola->step(x.head(3 * Ne + 17)), yb = ola->step(x.tail(n - 3 * Ne - 17));
    Veccf y = vconcat(ya, yb);
    CHECK(N == Ne && y.rows() == n - Ne, "windowed rows %d N %d", y.rows(), N);
    float err = 0;
    for (int k = 0; k < y.rows(); k++) {
      const cfloat ref = k >= Ne / 2 ? x(k - Ne / 2) * 0.5f : cfloat(0);
      err = std::max(err, (float) std::abs(y(k) - ref));
    }
    CHECK(err < 1e-5f, "windowed OLA err %g", err);
  }
  // (4) configuration errors and the cost model (test-fourier.cc:737-741 calls it for M = 127... values from the formula)
  {
    bool threw = false;
    try { FiltreFFTConfig c; c.dim_blocs_temporel = 64; filtre_fft(c); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw, "filtre_fft without traitement_freq must fail");
    float C; entier Nf, Nz, Ne;
    ola_complexité_optimise(127, C, Nf, Nz, Ne);
    CHECK(Nf == Ne + Nz && Nz == 126 && (Nf & (Nf - 1)) == 0 && C > 0, "ola_complexité_optimise Nf=%d Nz=%d Ne=%d C=%g", Nf, Nz, Ne, C);
  }
}


// ---- psd / psd_welch (fourier.hpp:739-757, freqestim.cc:7-93) --------------------------------------
static void test_psd()
{
  // axes: tfd_freqs / psd_freqs known values
  Vecf f8 = tfd_freqs(8), f7 = tfd_freqs(7), f8n = tfd_freqs(8, false), p7 = psd_freqs(7, false);
  CHECK(f8(0) == -0.5f && std::abs(f8(7) - 0.375f) < 1e-6f && std::abs(f7(0) + 0.5f - 1.0f / 7) < 1e-6f && f7(6) == 0.5f, "tfd_freqs");
  CHECK(f8n(0) == 0 && std::abs(f8n(3) - 0.375f) < 1e-6f && f8n(4) == -0.5f && std::abs(f8n(7) + 0.125f) < 1e-6f, "tfd_freqs no shift");
  CHECK(p7.rows() == 3 && p7(0) == 0 && std::abs(p7(2) - (0.5f - 1.0f / 7)) < 1e-6f, "psd_freqs real odd");
  // complex exponential at bin 200 of 1024 (f = +0.1953): the periodogram peaks there, Hann main lobe -6 dB at +-1 bin
  const int n = 1024, k0 = 200;
  Veccf x(n);
  for (int i = 0; i < n; i++) x(i) = std::polar(1.0f, (float) (2 * π * k0 * i / n));
  auto [fr, Y] = psd(x);
  const int im = Y.index_max();
  CHECK(im == n / 2 + k0 && std::abs(fr(im) - (float) k0 / n) < 1e-6f, "psd peak at %d (f=%g)", im, fr(im));
  CHECK(std::abs(Y(im) - Y(im - 1) - 6.02f) < 0.05f && std::abs(Y(im) - Y(im + 1) - 6.02f) < 0.05f, "Hann lobe %g %g", Y(im) - Y(im - 1), Y(im) - Y(im + 1));
  // unitary FFT of a Hann-weighted unit exponential: |X|^2 = (sum w)^2 / n = n / 4  ->  10 log10(256)
  CHECK(std::abs(Y(im) - 10 * std::log10(n / 4.0f)) < 0.01f, "psd level %g", Y(im));
  // real input: half spectrum, peak at the same bin
  Vecf xr = real(x);
  auto [frr, Yr] = psd(xr);
  CHECK(Yr.rows() == n / 2 && Yr.index_max() == k0 && std::abs(frr(k0) - (float) k0 / n) < 1e-3f, "psd real peak %d", Yr.index_max());
  // Welch: equals the sum of the per-segment periodograms computed one by one through fft()
  const int N = 256;
  Veccf z = randcn(5 * N + 77);
  auto [fw, W] = psd_welch(z, N);
  Vecf S = Vecf::zeros(N), w = fenêtre("hn", N, false);
  int nseg = 0;
  for (int i = 0; i + N < z.rows(); i += N / 2) {
    Veccf xp = z.segment(i, N).clone();
    for (int j = 0; j < N; j++) xp(j) *= w(j);
    S += fftshift(abs2(fft(xp)));
    nseg++;
  }
  Vecf e = pow2db(S);
  e -= W;
  CHECK(nseg == 9 && fw.rows() == N && maxabs(e) < 1e-3f, "psd_welch: %d segments, err %g dB", nseg, maxabs(e));
  // dsp:: spellings of the "next" rows reach the same functions (dsp/fourier.hpp)
  {
    dsp::fourier::FFTFilterConfig c;
    c.time_blocks_length = 64;
    c.freq_domain_processing = [](dsp::Veccf &) {};
    auto [flt, Nf] = dsp::fourier::filter_fft(c);
    dsp::Veccf q = randcn(128);
    CHECK(Nf == 64 && flt->step(q).rows() == 128, "dsp::fourier::filter_fft");
    auto [lg, cr] = dsp::fourier::xcorr(q);
    auto [dl, sc] = dsp::fourier::delay_estimation(q, dsp::fourier::delay(q, 3.f));
    auto [qa, qb, di, si] = dsp::fourier::align_int(q, dsp::fourier::delay(q, 3.f));
    auto [pf, pw] = dsp::fourier::psd_welch(q, 32);
    CHECK(lg.rows() == 255 && std::abs(dl - 3) < 0.05f && di == 3 && qa.rows() == qb.rows() && pf.rows() == 32 && pw.rows() == 32,
          "dsp::fourier aliases (delay %g/%d)", dl, di);
    CHECK(dsp::fourier::resample_freq(real(q), 2).rows() == 256 && dsp::fourier::psd_freqs(8).rows() == 8, "dsp::fourier aliases 2");
  }
}


// ---- filtre_lexp / filtre_dc / filtre_mg / ligne_a_retard (filtre-rt.cc:14-51,603-786) -----------
// No reference test pins their samples (test-filtres.cc only plots them): checked against the
// literal recurrences, in float (the reference's arithmetic) and in double (the yardstick for
// how far float arithmetic may drift on a slow first-order recursion).
template <typename T> static void test_petits_filtres()
{
  const int n = 20000;
  Vecteur<T> x(n);
  {
    Veccf r = randcn(n);
    for (int i = 0; i < n; i++) {
      if constexpr (est_complexe<T>()) x(i) = r(i) + cfloat(0.5f, -0.25f);
      else x(i) = r(i).real() + 0.5f;
    }
  }
  using Td = std::conditional_t<est_complexe<T>(), cdouble, double>;
  auto chunks = [&](sptr<FiltreGen<T>> f) {
    Vecteur<T> y = f->step(x.head(777));
    y = vconcat(y, f->step(x.segment(777, 9001)));
    return vconcat(y, f->step(x.tail(n - 777 - 9001)));
  };
  for (float γ : {0.5f, 0.05f, 0.002f}) {
    // FiltreLExp: acc = x(0); acc += γ (x - acc)
    Vecteur<T> y = chunks(filtre_lexp<T>(γ));
    T acc = x(0);
    Td accd = (Td) x(0);
    float e_ref = 0, e_gpu = 0, ymax = 0;
    for (int i = 0; i < n; i++) {
      acc += γ * (x(i) - acc);
      accd += (double) γ * ((Td) x(i) - accd);
      e_ref = std::max(e_ref, (float) std::abs((Td) acc - accd));
      e_gpu = std::max(e_gpu, (float) std::abs((Td) y(i) - accd));
      ymax = std::max(ymax, (float) std::abs(accd));
    }
    // (the adaptor keeps the DC gain exactly 1 like the reference's incremental form; with b0 = γ the
    // float rounding of 1 - γ alone doubled this error; what remains, ~4e-6 of the maximum at γ = 0.002,
    // is the float32 conditioning of a pole at 0.998 in block-parallel form -- inside the 1e-5 parity bar)
    CHECK(y.rows() == n && e_gpu <= std::max(4 * e_ref, 1e-5f * ymax), "filtre_lexp(%g): err %g (float recurrence: %g, max %g)", γ, e_gpu, e_ref, ymax);
    CHECK(std::abs(y(0) - x(0)) <= 1e-6f * std::abs(x(0)), "filtre_lexp first output is the first input");
  }
  for (float fc : {0.1f, 0.001f}) {
    // FiltreDC: y = α ((x - xp) + yp), zero memory
    Vecteur<T> y = chunks(filtre_dc<T>(fc));
    const float α = 1 - lexp_coef(fc);
    T xp = 0, yp = 0;
    Td xpd = 0, ypd = 0;
    float e_ref = 0, e_gpu = 0, ymax = 0;
    for (int i = 0; i < n; i++) {
      const T o = (T) (α * ((x(i) - xp) + yp));
      xp = x(i); yp = o;
      const Td od = (double) α * (((Td) x(i) - xpd) + ypd);
      xpd = (Td) x(i); ypd = od;
      e_ref = std::max(e_ref, (float) std::abs((Td) o - od));
      e_gpu = std::max(e_gpu, (float) std::abs((Td) y(i) - od));
      ymax = std::max(ymax, (float) std::abs(od));
    }
    CHECK(e_gpu <= std::max(4 * e_ref, 2e-6f * ymax), "filtre_dc(%g): err %g (float recurrence: %g, max %g)", fc, e_gpu, e_ref, ymax);
  }
  for (int K : {1, 7, 100, 1000}) {
    // MoyenneGlissante<T, double>
    Vecteur<T> y = chunks(filtre_mg<T, Td>(K));
    Td accu = 0;
    float e = 0, ymax = 0;
    for (int i = 0; i < n; i++) {
      accu += (Td) x(i);
      if (i >= K) accu -= (Td) x(i - K);
      const Td o = accu * (double) (float) (1.0 / K);
      e = std::max(e, (float) std::abs((Td) y(i) - o));
      ymax = std::max(ymax, (float) std::abs(o));
    }
    CHECK(e <= 1e-5f * ymax, "filtre_mg(%d): err %g (max %g)", K, e, ymax);
  }
  for (int d : {0, 1, 100, 5000}) {
    Vecteur<T> y = chunks(ligne_a_retard<T>(d));
    bool ok = y.rows() == n;
    for (int i = 0; ok && i < n; i++) ok = y(i) == (i >= d ? x(i - d) : T(0));
    CHECK(ok, "ligne_a_retard(%d)", d);
  }
}


// ---- test_ra_unit (core/tests/test-ra.cc:11-160): a 2 kHz sine at fe = 100 kHz through each
// resampler; the output must be a pure sine at f2 / (ratio fe): >= 80 % of the energy in the 3 bins
// around it, spurious lines <= -50 dB, sample count within 1 %, amplitude within 10 %.
struct Purete { float freq_spurius, max_spurius_db; };
static Purete verifie_sinus(const char *nom, const Vecf &x, float f)
{
  const int n = x.rows();
  Vecf fen = fenêtre("hn", n, false), xf = x.clone();
  for (int i = 0; i < n; i++) xf(i) *= fen(i);
  Vecf X = abs2(fft(xf).head(n / 2));
  const float etotal = X.somme();
  const int idx = (int) (f * n);
  float ef = X(idx);
  if (idx > 0) ef += X(idx - 1);
  if (idx + 1 < n / 2) ef += X(idx + 1);
  const float score = ef / etotal;
  if (idx >= 10)
    for (int i = idx - 10; i < idx + 10; i++) X(i) = 0;
  const int is = X.index_max();
  Purete res{(float) is / n, 10 * std::log10(X(is) / ef)};
  CHECK(!(std::isnan(score) || score < 0.8f), "%s: a pure sine is expected (energy ratio %g)", nom, score);
  return res;
}
static void test_ra_unit(const char *nom, float ratio, sptr<FiltreGen<float>> ra, float max_spurius_dB = -50)
{
  const float fe = 100e3f, f2 = 2e3f;
  Vecf x = Vecf::int_expr(1000, [&](int i) { return (float) std::sin((double) i / fe * 2 * π * f2); });
  Vecf y = ra->step(x);
  verifie_sinus(nom, x, f2 / fe);
  const Purete spy = verifie_sinus(nom, y, f2 / (ratio * fe));
  CHECK(100.0 * std::abs((y.rows() - ratio * x.rows()) / x.rows()) < 1, "%s ratio %g: %d samples", nom, ratio, y.rows());
  const float amp1 = x.valeur_max() - x.valeur_min(), amp2 = y.valeur_max() - y.valeur_min();
  CHECK(100 * (amp1 - amp2) / amp1 < 10, "%s ratio %g: amplitude %g -> %g", nom, ratio, amp1, amp2);
  CHECK(spy.max_spurius_db <= max_spurius_dB, "%s ratio %g: spurious line %.1f dB at f = %g", nom, ratio, spy.max_spurius_db, spy.freq_spurius);
}
static void test_ra()
{
  for (float ratio : {1.f, 1.5f, 0.5f, 2.f, 1.2f, π_f}) {
    test_ra_unit("filtre_itrp/cspline", ratio, filtre_itrp<float>(ratio, itrp_cspline<float>()));
    test_ra_unit("filtre_itrp/sinc", ratio, filtre_itrp<float>(ratio, itrp_sinc<float>({127, 256, 0.5f, "hn"})));
    test_ra_unit("filtre_reechan", ratio, filtre_reechan<float>(ratio));
  }
  {
    Vecf h = design_rif_fen(15, "lp", 0.25f, "hn");
    test_ra_unit("rif demi-bande", 0.5f, filtre_rif_demi_bande<float, float>(h));
    test_ra_unit("rif ups", 2.0f, filtre_rif_ups<float, float>(h, 2));
  }
  // the table-driven interpolators of test_itrp (test-itrp.cc:62-87) all build and run; the sinc table
  // equals the oracle's for the Hann window, and any other window name means no window (itrp.cc:29-37)
  {
    Veccf xs = randcn(3000);
    for (auto cfg : std::vector<InterpolateurSincConfig>{{15, 256, 0.5f, "re"}, {15, 256, 0.25f, "re"}, {15, 256, 0.5f, "hn"}, {15, 256, 0.25f, "hn"},
                                                          {15, 256, 0.4f, "hn"}, {15, 512, 0.5f, "hn"}, {31, 256, 0.5f, "hn"}, {63, 256, 0.5f, "hn"},
                                                          {127, 256, 0.5f, "hn"}, {31, 256, 0.48f, "hn"}, {63, 256, 0.48f, "hn"}, {15, 1024, 0.5f, "hn"}}) {
      auto it = itrp_sinc<cfloat>(cfg);
      Veccf ys = filtre_itrp<cfloat>(1.3f, it)->step(xs);
      CHECK(it->K == cfg.ncoefs && std::abs(ys.rows() - 3900) <= 2, "itrp_sinc{%d,%d,%g,%s}: %d outputs", cfg.ncoefs, cfg.nphases, cfg.fcut, cfg.fenetre.c_str(), ys.rows());
      if (cfg.fenetre == "hn") {
        std::vector<float> lut((size_t) (cfg.nphases + 1) * cfg.ncoefs);
        orc_itrp_sinc_lut(cfg.ncoefs, cfg.nphases, cfg.fcut, lut.data());
        float e = 0;
        for (float τ : {0.f, 0.1f, 0.5f, 0.999f}) {
          const Vecf hc = std::dynamic_pointer_cast<InterpolateurRIF<cfloat>>(it)->coefs(τ);
          const int row = (int) (τ * cfg.nphases);
          for (int k = 0; k < cfg.ncoefs; k++) e = std::max(e, std::abs(hc(k) - lut[(size_t) row * cfg.ncoefs + k]));
        }
        CHECK(e <= 2e-7f, "itrp_sinc{%d,%d,%g} table differs from the oracle's by %g", cfg.ncoefs, cfg.nphases, cfg.fcut, e);
      }
    }
    CHECK(std::abs(filtre_itrp<cfloat>(0.77f, itrp_cspline<cfloat>())->step(xs).rows() - 2310) <= 2, "cspline rows");
    // the analytic interpolators of the same list (test-itrp.cc:65-72): itrp_lineaire, itrp_lagrange(1,2,3,5,6,7).
    // Their taps sum to 1 for every phase, and the GPU resampler = the interpolator's own step() replayed
    // on the host over AdaptationRythmeSimple's recurrence (ra.cc:56-77)
    std::vector<sptr<Interpolateur<cfloat>>> lst{itrp_lineaire<cfloat>()};
    for (int d : {1, 2, 3, 5, 6, 7}) lst.push_back(itrp_lagrange<cfloat>(d));
    for (auto &it : lst) {
      auto rif = std::dynamic_pointer_cast<InterpolateurRIF<cfloat>>(it);
      float dev = 0;
      for (float τ = 0; τ < 1; τ += 0.1f) {
        const Vecf hc = rif->coefs(τ);
        double sum = 0;
        for (int k = 0; k < hc.rows(); k++) sum += hc(k);
        dev = std::max(dev, (float) std::abs(sum - 1));
      }
      CHECK(dev < 1e-5f, "%s: taps sum to 1 (%g)", it->nom.c_str(), dev);
      const float ratio = 1.3f;
      Veccf ys = filtre_itrp<cfloat>(ratio, it)->step(xs);
      const int K = it->K;
      Veccf fen = Veccf::zeros(K), yr(ys.rows() + 8);
      float phase = 0, inc = 1.0f / ratio;
      int j = 0;
      for (int i = 0; i < xs.rows(); i++) {
        for (int k = 0; k + 1 < K; k++) fen(k) = fen(k + 1);
        fen(K - 1) = xs(i);
        while (phase < 1) {
          if (j < yr.rows()) yr(j) = it->step(fen, 0, phase);
          j++;
          phase += inc;
        }
        phase -= 1;
      }
      float e = 0, m = 0;
      for (int i = 0; i < std::min(j, ys.rows()); i++) { e = std::max(e, (float) std::abs(ys(i) - yr(i))); m = std::max(m, (float) std::abs(yr(i))); }
      CHECK(j == ys.rows() && e <= 1e-5f * m, "filtre_itrp(%s): %d vs %d outputs, err %g", it->nom.c_str(), ys.rows(), j, e);
    }
  }
  for (int R : {2, 3, 4, 5, 8}) test_ra_unit("rif decim", 1.0f / R, filtre_rif_decim<float, float>(design_rif_fen(15, "lp", 0.5f / R, "hn"), R));
}


// ---- ports of test_ligne_a_retard / test_filtre_mg / test_filtrage_ola (test-filtres.cc:201-264,418-446)
template <typename T> static Vecteur<T> par_blocs(sptr<FiltreGen<T>> f, const Vecteur<T> &x, int bs)
{
  Vecteur<T> y;
  for (int o = 0; o < x.rows(); o += bs) y = vconcat(y, f->step(x.segment(o, std::min(bs, x.rows() - o))));
  return y;
}
static Vecf signal_test_5000()
{
  const int n = 5000;
  Vecf g = real(randcn(n));
  return Vecf::int_expr(n, [&](int t) { return (float) (0.1 * g(t) + std::sin(t * (2 * π / n) * 20) * std::exp(-std::abs((t - n / 2.0) / (n / 8.0)))); });
}
static void test_ligne_a_retard_ref(int δ)
{
  Vecf x = signal_test_5000(), y = par_blocs<float>(ligne_a_retard<float>(δ), x, 311);
  const int n = x.rows();
  CHECK(y.rows() == n, "ligne à retard : pb dim");
  float err = 0;
  for (int i = 0; i < n - δ; i++) err = std::max(err, std::abs(y(δ + i) - x(i)));
  CHECK(err == 0, "ligne à retard δ=%d: err %g", δ, err);
}
static void test_filtre_mg_ref(int R)
{
  const int n = 1000;
  Vecf x = real(randcn(n)), y = par_blocs<float>(filtre_mg<float, double>(R), x, 80);
  CHECK(y.rows() == n, "filtre mg : pb dim");
  float err = 0;
  for (int i = 0; i < n; i++) {
    const int imin = std::max(0, i - (R - 1));
    float sref = 0;
    for (int j = imin; j <= i; j++) sref += x(j);
    err = std::max(err, std::abs(y(i) - sref / R));
  }
  CHECK(err < 5e-7f, "Echec filtre MG (R=%d) : err = %g", R, err);      // the reference's bound (test-filtres.cc:254)
}
static void test_filtrage_ola_ref()
{
  // windowed OLA that zeroes the bins N/32 .. 31N/32 of every 512-point block: a brick-wall low-pass.
  // The reference only plots; here: the low-frequency burst of the test signal survives (delayed by
  // Ne/2, halved by the window overlap -- see test_filtre_fft) and the wide-band noise is cut.
  FiltreFFTConfig c;
  c.avec_fenetrage = true;
  c.dim_blocs_temporel = 512;
  c.traitement_freq = [](Veccf &X) {
    const int N = 512;
    for (int i = N / 32; i < N / 32 + (30 * N) / 32; i++) X(i) = 0;
  };
  auto [ola, N] = filtre_fft(c);
  Veccf x = signal_test_5000().as<cfloat>();
  Veccf y = par_blocs<cfloat>(ola, x, 1000);
  CHECK(N == 512 && y.rows() == (5000 / 512 - 1) * 512, "OLA rows %d", y.rows());
  // the burst sin(2 pi 20 t / 5000) sits at bin 2 of 512: compare with the input delayed by 256, halved
  double num = 0, den = 0;
  for (int k = 1500; k < 3500; k++) {
    const double ref = 0.5 * std::sin((k - 256) * (2 * π / 5000) * 20) * std::exp(-std::abs((k - 256 - 2500.0) / 625.0));
    num += std::norm(y(k) - cfloat((float) ref, 0));
    den += ref * ref;
  }
  CHECK(num / den < 0.02, "OLA low-pass: relative residual %g", num / den);
}


// ---- test_detecteur_unit (core/tests/test-detecteur.cc:153-328): a 400-sample Gaussian-windowed
// quadratic chirp, seven occurrences (gains 2 .. 0.02, one at a fractional position, one straddling
// a block boundary, one ending exactly on one) in noise of σ = 0.01, blocks of 4096; every
// occurrence must be reported once, with the reference's error bounds.
static void test_detecteur_unit(float σ, int BS, DetecteurConfig::Mode mode)
{
  const int N = 8 * BS, M = 400;
  Veccf motif(M);
  {
    double phase = 0;
    for (int i = 0; i < M; i++) {
      const double u = (double) i / (M - 1), freq = 0.001 + (0.2 - 0.001) * u * u;   // sigchirp(0.001, 0.2, M, 'q')
      phase += 2 * π * freq;                                                          // cumsum
      const double t = (i - M / 2.0) / (M / 2.0);                                     // siggauss(M, 10)
      motif(i) = (float) (std::exp(-10 * t * t) * std::cos(phase));
    }
    double en = 0;
    for (int i = 0; i < M; i++) en += std::norm(motif(i));
    motif *= cfloat((float) (std::sqrt(400.0) / std::sqrt(en)), 0);
  }
  struct Occurence { float gain, position, phase; };
  const std::vector<Occurence> occ = {{2.0f, 900.0f, π_f / 4}, {4.0f, 2000.4f, -π_f / 4}, {1.0f, BS - 1.0f, 0}, {1.0f, 2.0f * BS, 0},
                                      {0.1f, 2.2f * BS, 0},   {0.05f, 2.5f * BS, 0},     {0.02f, 2.7f * BS, 0}};
  Veccf x = Veccf::zeros(N);
  for (const auto &o : occ) {
    Veccf m2 = motif.clone();
    const float p = o.position - std::floor(o.position);
    if (p > 0.01f) m2 = délais(motif, p);
    m2 *= std::polar(o.gain, o.phase);
    x.segment((int) std::floor(o.position), M) = m2;
  }
  Veccf bruit = randcn(N);
  bruit *= cfloat(σ / std::sqrt(2.0f), 0);
  x += bruit;
  double pm = 0;
  for (int i = 0; i < M; i++) pm += std::norm(motif(i));
  pm /= M;
  int cnt_ech = 0, ndet = 0, err = 0;
  DetecteurConfig config;
  config.mode = mode;
  config.gere_detection = [&](const Detection &det) {
    const float pos_abs = det.position_prec + cnt_ech;
    const int k = ndet++;
    if (k >= (int) occ.size()) return;
    const float SNR_v = 10 * std::log10((float) (pm * occ[k].gain * occ[k].gain) / (σ * σ));
    const float err_phase = (det.θ - occ[k].phase) * 180 / π_f, err_gain = det.gain - occ[k].gain, err_pos = pos_abs - occ[k].position;
    const float err_σ = det.σ_noise - σ, err_SNR = det.SNR_dB - SNR_v;
    bool bad = false;
    if (SNR_v > 15) bad = bad || std::abs(err_phase) > 1 || std::abs(err_gain / occ[k].gain) > 1e-2f;
    bad = bad || std::abs(err_pos) > 0.1f || std::abs(err_σ / σ) > 0.25f || std::abs(err_SNR) > 1;
    if (bad) {
      err = 1;
      printf("  detection %d (mode %d): pos %.3f (expected %.1f) gain %.4f (%.2f) phase %.2f deg err, sigma %.4g, SNR %.1f (%.1f)\n", k, (int) mode,
             pos_abs, occ[k].position, det.gain, occ[k].gain, err_phase, det.σ_noise, det.SNR_dB, SNR_v);
    }
  };
  config.Ne = BS;
  config.motif = motif;
  config.seuil = 0.8f;
  auto det = détecteur_création(config);
  for (int i = 0; i < N / BS; i++) {
    det->step(x.segment(i * BS, BS));
    cnt_ech += BS;
  }
  CHECK(ndet == (int) occ.size(), "détecteur mode %d: %d detections, expected %d", (int) mode, ndet, (int) occ.size());
  CHECK(!err, "détecteur mode %d: error bounds of the reference's test exceeded", (int) mode);
}


// ---- rt_spectrum (fourier.cc:1148-1342): no reference test exists; checked on its definition
static void test_rt_spectrum()
{
  // white noise of unit variance: every bin averages to 1/Nf (unitary FFT, window energy Nf, extra 1/Nf)
  {
    SpectrumConfig c;
    c.BS = 1024; c.nsubs = 4; c.nmeans = 10;
    auto sp = rt_spectrum(c);
    Vecf y;
    int nonvide = 0;
    for (int b = 0; b < 20; b++) {
      Vecf o = sp->step(randcn(c.BS));
      if (o.rows()) { y = o; nonvide++; CHECK((b + 1) % c.nmeans == 0, "spectrum delivered at block %d", b); }
    }
    CHECK(nonvide == 2 && y.rows() == 256, "rt_spectrum cadence %d rows %d", nonvide, y.rows());
    double m = 0;
    for (int k = 0; k < y.rows(); k++) m += std::pow(10.0, y(k) / 10);
    m /= y.rows();
    // randcn: unit variance per component, E|x|^2 = 2
    CHECK(std::abs(10 * std::log10(m) - 10 * std::log10(2.0 / 256)) < 0.3, "noise floor %g dB (expected %g)", 10 * std::log10(m), 10 * std::log10(2.0 / 256));
  }
  // a complex exponential on bin +40 of 256: the peak sits at the fftshift-ed position, Hann main lobe
  {
    SpectrumConfig c;
    c.BS = 256; c.nsubs = 1; c.nmeans = 3;
    auto sp = rt_spectrum(c);
    Veccf x = Veccf::int_expr(256, [&](int i) { return std::polar(1.0f, (float) (2 * π * 40 * i / 256)); });
    sp->step(x); sp->step(x);
    Vecf y = sp->step(x);
    CHECK(y.rows() == 256 && y.index_max() == 128 + 40, "rt_spectrum peak at %d", y.index_max());
    // |X|^2 = (sum f)^2 / Nf with f = sqrt(8/3) * hann  ->  Nf * 2/3; divided by Nf: 2/3
    CHECK(std::abs(y(168) - 10 * std::log10(2.0 / 3)) < 0.05 && std::abs(y(168) - y(167) - 6.02) < 0.1, "rt_spectrum level %g dB", y(168));
  }
  // sweep: 4 tunings 64 bins apart, edges masked: Ns = 256 + 3*64 bins, flat for white noise
  {
    SpectrumConfig c;
    c.BS = 1024; c.nsubs = 4; c.nmeans = 20;
    c.sweep.active = true; c.sweep.step = 64; c.sweep.masque_hf = 16;
    c.fenetre = tsd::filtrage::Fenetre::AUCUNE;
    auto sp = rt_spectrum(c);
    Vecf y;
    for (int b = 0; b < 20; b++) y = sp->step(randcn(c.BS));
    CHECK(c.Ns() == 448 && y.rows() == 448, "sweep rows %d", y.rows());
    double lo = 1e9, hi = -1e9;
    for (int k = 16; k < 448 - 16; k++) { lo = std::min<double>(lo, y(k)); hi = std::max<double>(hi, y(k)); }
    // (the reference divides by nsubs in sweep mode too, although a bin only collects mag_cnt(k) <= nsubs
    // sub-blocks: the floor sits 10 log10(nsubs) lower; 20 averages per contribution: +-3 dB extremes)
    CHECK(hi - lo < 7.0 && std::abs(0.5 * (hi + lo) - 10 * std::log10(2.0 / 256 / 4)) < 2.0, "sweep flatness %g .. %g dB", lo, hi);
  }
  bool threw = false;
  try { SpectrumConfig c; auto sp = rt_spectrum(c); sp->step(randcn(100)); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw, "rt_spectrum must reject a block of the wrong size");
}

// ---- test_ccorr (test-fourier.cc:575-607): circular correlation against its definition
static void test_ccorr()
{
  for (int n : {1, 2, 3, 10, 15, 16, 21, 32}) {
    Veccf a1 = randcn(n), a2 = randcn(n);
    auto [lags, c] = ccorr(a1, a2);
    float err = 0;
    for (int i = 0; i < n; i++) {
      cfloat s = 0;
      for (int k = 0; k < n; k++) s += a1(k) * std::conj(a2((k + i) % n));
      err = std::max(err, (float) std::abs(s / (float) n - c(i)));
    }
    CHECK(c.rows() == n && lags.rows() == n && err < 1e-5f, "ccorr n=%d err %g", n, err);
  }
}

// ---- test_csym (test-fourier.cc:659-675): a spectrum forced conjugate-symmetric has a real inverse
static void test_csym(int n)
{
  Veccf X = randcn(n);
  csym_forçage(X);
  Veccf x1 = ifft(X);
  float e1 = 0, e2 = 0;
  for (int i = 0; i < n; i++) e1 = std::max(e1, std::abs(x1(i).imag()));
  // the direct inverse DFT in double (the reference's tfd<cfloat>(X, oui) yardstick)
  for (int t = 0; t < n; t++) {
    double im = 0;
    for (int k = 0; k < n; k++) {
      const double a = 2 * π * (double) ((long) k * t % n) / n;
      im += X(k).real() * std::sin(a) + X(k).imag() * std::cos(a);
    }
    e2 = std::max(e2, (float) std::abs(im / std::sqrt((double) n)));
  }
  CHECK(e1 < 1e-3f && e2 < 2e-6f, "csym n=%d: imaginary residue %g (fft) %g (dft)", n, e1, e2);
}

// ---- test_rfftplan (test-fourier.cc:27-37) and test_fftshift (:39-72)
static void test_rfftplan_ref()
{
  Vecf x = randn(101);
  auto plan = rtfrplan_création();
  Veccf y = plan->step(x), r = rfft(x);
  float err = 0;
  for (int i = 0; i < 101; i++) err = std::max(err, (float) std::abs(y(i) - r(i)));
  CHECK(y.rows() == 101 && err < 1e-6f, "rfftplan err %g", err);
}
static void test_fftshift_ref(int n)
{
  Vecf x = linspace(0, n - 1, n), y = fftshift(x);
  const int m = n / 2;
  bool ok = y.rows() == n;
  for (int i = 0; i < n && ok; i++) {
    const float ref = (n & 1) ? (i < m ? m + 1 + i : i - m) : (i < m ? m + i : i - m);
    ok = y(i) == ref;
  }
  CHECK(ok, "fftshift(%d)", n);
}

// ---- test_filtfilt (test-filtres.cc:267-293; the reference only plots -- "TODO : automatiser"):
// forward-backward filtering = filter, reverse, filter, reverse, hence zero phase: a symmetric
// input stays symmetric about the same point, while the one-way output is shifted by (K-1)/2
static void test_filtfilt()
{
  const int n = 500;
  Vecf x(n);
  for (int i = 0; i < n / 2; i++) x(i) = x(n - 1 - i) = (float) i;
  Vecf h = design_rif_fen(63, "lp", 0.05f);
  Vecf y1 = filtrer(h, x), y = filtfilt(h, x);
  Vecf manual = filtrer(h, filtrer(h, x).reverse()).reverse();
  float e = 0, asym = 0, asym1 = 0;
  for (int i = 0; i < n; i++) e = std::max(e, std::abs(y(i) - manual(i)));
  for (int i = 100; i < 400; i++) {
    asym = std::max(asym, std::abs(y(i) - y(n - 1 - i)));
    asym1 = std::max(asym1, std::abs(y1(i) - y1(n - 1 - i)));
  }
  CHECK(y.rows() == n && e == 0.0f, "filtfilt vs its definition: %g", e);
  CHECK(asym < 0.05f * 250 && asym1 > 20.0f, "filtfilt zero phase: asymmetry %g (one-way filter: %g)", asym, asym1);
}

// Spectrum's OpenMP loop calls plan->step on ONE plan from several threads (fourier.cc:1244-1252):
// a plan must give every caller its own transform although the scratch buffers are shared.
static void test_plan_concurrent(int n)
{
  auto plan = tfrplan_création(n);
  const int NT = 4, REP = 6;
  std::vector<Veccf> x(NT), ref(NT);
  for (int t = 0; t < NT; t++) { x[t] = randcn(n); ref[t] = plan->step(x[t]); }
  std::vector<float> err(NT, 0.0f);
  std::vector<std::thread> th;
  for (int t = 0; t < NT; t++)
    th.emplace_back([&, t]() {
      for (int r = 0; r < REP; r++) {
        Veccf y = plan->step(x[t]);
        for (int i = 0; i < n; i++) err[t] = std::max(err[t], std::abs(y(i) - ref[t](i)));
      }
    });
  for (auto &q : th) q.join();
  for (int t = 0; t < NT; t++) CHECK(err[t] == 0.0f, "concurrent plan->step n=%d thread %d: err %g", n, t, err[t]);
}

int main(int argc, char **argv)
{
  if (argc > 1 && !std::strcmp(argv[1], "--no-gpu")) {
    bool threw = false;
    try { (void) filtre_rif<float, float>(Vecf::ones(3)); } catch (const std::runtime_error &e) { threw = true; printf("expected failure: %s\n", e.what()); }
    test_tab();
    CHECK(threw, "filtre_rif must throw without a GPU (no CPU fallback)");
    printf(nfail ? "FAILED (%d)\n" : "OK\n", nfail);
    return nfail ? 1 : 0;
  }
  test_tab();
  test_filtre_rif();
  test_retard(3);
  test_retard(4);
  for (int i : {10, 11, 15, 20}) for (int j : {10, 11, 15, 20}) test_design_rif_prod(i, j);
  test_rif_vs_rif_fft();
  test_fir_vs_oracle();
  test_sois();
  test_design_biquad();
  test_riia();
  for (int n : {16, 1, 2, 3, 4, 5, 8, 10, 17, 128, 129, 1024}) { test_fft_valide(n, false); test_fft_valide(n, true); }
  test_fft_misc();
  test_reechan();
  test_polyphase();
  test_xcorr(true);
  test_xcorr(false);
  test_reechan_freq();
  test_czt();
  test_pad_zeros();
  for (float f : {0.f, 250.f, 1.f, 10.f, -10.f, 0.5f, 0.1f, 1.5f}) {
    test_delais_fractionnaire<cfloat>(f);
    test_delais_fractionnaire<float>(f);
  }
  for (int N : {32, 1024, 15 * 1024})
    for (float f : {0.f, 1.f, 10.f, 20.f, 30.f, 40.f, -50.f, 11.f, 1.1f}) {
      if (N == 32) continue;                    // the reference skips (N = 32, test signal) too (test-fourier.cc:724-727)
      test_delais_unitaire(f, N);
    }
  test_align_entier();
  test_tampon();
  test_filtre_fft();
  test_psd();
  test_rt_spectrum();
  test_ccorr();
  for (int n : {3, 4, 5, 63, 64, 511, 512, 1000, 1001}) test_csym(n);
  test_rfftplan_ref();
  test_fftshift_ref(15);
  test_fftshift_ref(16);
  test_filtfilt();
  for (int n : {4096, 1 << 18, 3000, 1001}) test_plan_concurrent(n);
  test_ra();
  test_ligne_a_retard_ref(0);
  test_ligne_a_retard_ref(70);
  for (int K : {4, 11, 20}) test_filtre_mg_ref(K);
  test_filtrage_ola_ref();
  test_detecteur_unit(0.01f, 4 * 1024, DetecteurConfig::MODE_OLA);
  test_detecteur_unit(0.01f, 4 * 1024, DetecteurConfig::MODE_RIF);
  {
    // dsp:: spelling (dsp/fourier.hpp:528-583): the same object
    dsp::fourier::DetectorConfig dc;
    dc.pattern = randcn(64);
    dc.Ns = 256;
    dc.threshold = 0.9f;
    int hits = 0;
    dc.on_detection = [&](const dsp::fourier::Detection &) { hits++; };
    auto d = dsp::fourier::detector_new(dc);
    Veccf sig = Veccf::zeros(1024);
    sig.segment(300, 64) = dc.pattern;
    for (int i = 0; i < 4; i++) d->step(sig.segment(256 * i, 256));
    CHECK(hits == 1, "dsp::fourier::detector_new: %d detections of a clean pattern", hits);
  }
  test_petits_filtres<float>();
  test_petits_filtres<cfloat>();
  {
    // dsp:: spellings (dsp/filter.hpp:1128-1164,1288-1292,1578-1631,1827-1883)
    dsp::Vecf q = real(randcn(300));
    auto ma = dsp::filter::filter_ma<float, double>(8);
    CHECK(dsp::filter::delay_line<float>(3)->step(q).rows() == 300 && dsp::filter::filter_ema<float>(0.1f)->step(q).rows() == 300 &&
          dsp::filter::filter_dc<float>(0.01f)->step(q).rows() == 300 && ma->step(q).rows() == 300 &&
          dsp::filter::decimator<float>(3)->step(q).rows() == 100 && std::abs(dsp::filter::ema_coef(0.1f) - lexp_coef(0.1f)) == 0,
          "dsp::filter small-filter aliases");
  }
  // (last: they draw from the shared random generator, and the statistical tests above are ported with
  // the noise realisations the default seed gives them)
  test_residence_gpu();
  test_residence_ops();
  test_fragments();
  printf(nfail ? "FAILED (%d)\n" : "ALL C++ HOST TESTS OK\n", nfail);
  return nfail ? 1 : 0;
}
