// test_design_cpu.cc -- the parts of the mirror that need no GPU (array type, windows, FIR / IIR designs, polynomial
// roots, interpolator tables, signal helpers), built with -fsanitize=address,undefined and run in the CPU test tier:
// a memory-safety and undefined-behaviour sweep over the design-time code.  Values are checked where a closed form exists.
#include <cmath>
#include <cstdio>
#include "tsd/tsd-all.hpp"

using namespace tsd;
using namespace tsd::filtrage;

static int nfail = 0;
#define CHECK(cond, ...)                                                                 \
  do {                                                                                   \
    if (!(cond)) {                                                                       \
      nfail++;                                                                           \
      printf("FAIL %s:%d: ", __FILE__, __LINE__);                                        \
      printf(__VA_ARGS__);                                                               \
      printf("\n");                                                                      \
    }                                                                                    \
  } while (0)

static double gain_at(const FRat<cfloat> &h, double f)
{
  const cdouble w = std::polar(1.0, 2 * π * f);
  cdouble H = cdouble(h.numer.mlt) / cdouble(h.denom.mlt);
  for (int i = 0; i < h.numer.coefs.rows(); i++) H *= (w - cdouble(h.numer.coefs(i)));
  for (int i = 0; i < h.denom.coefs.rows(); i++) H /= (w - cdouble(h.denom.coefs(i)));
  return std::abs(H);
}

int main()
{
  // ---- array type: views, copies, element-wise operators, bounds
  for (int n : {0, 1, 2, 7, 64, 1000}) {
    Vecf a = linspace(0, 1, n), b = Vecf::ones(n);
    Vecf c = a + b;
    c *= 2.0f;
    Vecf d = c.reverse();
    CHECK(d.rows() == n, "reverse size");
    if (n >= 7) {
      Vecf h = c.head(3), t = c.tail(3), s = c.segment(2, 4);
      CHECK(h.rows() == 3 && t.rows() == 3 && s.rows() == 4, "views");
      CHECK(h(0) == c(0) && t(2) == c(n - 1) && s(0) == c(2), "view contents");
      Vecf e = s.clone();
      e(0) = -1;
      CHECK(s(0) == c(2), "clone is a deep copy");
    }
    bool threw = false;
    try { (void) a(n); } catch (const std::exception &) { threw = true; }
    CHECK(threw, "operator() out of range must fail loudly (n = %d)", n);
    Veccf z = sigexp(0.1f, n);
    for (int i = 0; i < n; i++) CHECK(std::abs(std::abs(z(i)) - 1.0f) < 1e-5f, "sigexp modulus");
    Vecf r = randn(n);
    CHECK(r.rows() == n, "randn size");
  }
  // ---- windows and windowed FIR designs: every size 1..130, every type
  for (const char *fen : {"hn", "hm", "re", "tr"})
    for (int n = 1; n <= 130; n++) {
      Vecf w = fenêtre(fen, n, true), wp = fenêtre(fen, n, false);
      CHECK(w.rows() == n && wp.rows() == n, "fenêtre(%s, %d)", fen, n);
      for (int i = 0; i < n; i++) CHECK(std::isfinite(w(i)) && w(i) >= -1e-6f && w(i) <= 1.0f + 1e-6f, "fenêtre(%s, %d)(%d) = %g", fen, n, i, w(i));
      for (int i = 0; i < n / 2; i++) CHECK(std::abs(w(i) - w(n - 1 - i)) < 1e-6f, "symmetric window");
    }
  auto gain_rif = [](const Vecf &h, double f) {
    cdouble H = 0;
    for (int i = 0; i < h.rows(); i++) H += (double) h(i) * std::polar(1.0, -2 * π * f * i);
    return std::abs(H);
  };
  for (const char *type : {"lp", "hp", "bp", "sb"})
    for (int n : {3, 4, 15, 31, 64, 127, 128}) {
      if (n % 2 == 0 && (type[0] == 'b' || type[0] == 's')) {
        bool threw = false;
        try { (void) design_rif_fen(n, type, 0.2f, "hn", 0.35f); } catch (const std::exception &) { threw = true; }
        CHECK(threw, "design_rif_fen(%d, %s): even band designs are refused like the reference's", n, type);
        continue;
      }
      Vecf h = design_rif_fen(n, type, 0.2f, "hn", 0.35f);
      CHECK(h.rows() == n, "design_rif_fen(%d, %s)", n, type);
      double s = 0;
      for (int i = 0; i < n; i++) { CHECK(std::isfinite(h(i)), "finite taps"); s += h(i); }
      if (type[0] == 'l') CHECK(std::abs(s - 1) < 1e-3, "design_rif_fen(%d, lp): DC gain %g", n, s);
      if (n >= 127) {
        // pass / stop bands of the four types (band edges 0.2 .. 0.35)
        const double g0 = gain_rif(h, 0.02), gm = gain_rif(h, 0.275), g1 = gain_rif(h, 0.48);
        const bool ok = type[0] == 'l' ? (g0 > 0.99 && gm < 0.01 && g1 < 0.01)
                        : type[0] == 'h' ? (g0 < 0.01 && gm > 0.99 && g1 > 0.99)
                        : type[0] == 'b' ? (g0 < 0.01 && gm > 0.99 && g1 < 0.01)
                                         : (g0 > 0.99 && gm < 0.01 && g1 > 0.99);
        CHECK(ok, "design_rif_fen(%d, %s): gains %g %g %g", n, type, g0, gm, g1);
      }
    }
  // ---- analog prototypes -> bilinear: orders 1..14, both types, a few cut-offs and ripples
  for (const char *proto : {"butt", "cheb1", "cheb2", "ellip"})
    for (const char *type : {"lp", "hp"})
      for (int n = 1; n <= 14; n++)
        for (float fc : {0.02f, 0.1f, 0.25f, 0.45f})
          for (float rp : {0.05f, 1.0f}) {
            const FRat<cfloat> h = design_riia(n, type, proto, fc, rp, 50);
            CHECK(h.denom.coefs.rows() == n && h.numer.coefs.rows() == n, "design_riia(%d, %s, %s, %g): %d poles, %d zeros", n, type, proto, fc,
                  (int) h.denom.coefs.rows(), (int) h.numer.coefs.rows());
            double rmax = 0;
            for (int i = 0; i < n; i++) rmax = std::max(rmax, (double) std::abs(h.denom.coefs(i)));
            CHECK(rmax < 1.0, "design_riia(%d, %s, %s, %g): pole radius %g", n, type, proto, fc, rmax);
            const double gp = gain_at(h, type[0] == 'l' ? 0.0 : 0.5), gs = gain_at(h, type[0] == 'l' ? 0.5 : 0.0);
            CHECK(std::isfinite(gp) && gp > 0.5 && gp < 1.3 && gs < 0.72, "design_riia(%d, %s, %s, %g, %g): pass %g stop %g", n, type, proto, fc, rp, gp, gs);
          }
  // ---- cookbook biquads
  for (const char *type : {"lp", "hp", "bp", "notch", "res", "plateau-bf", "plateau-hf"})
    for (float f : {0.01f, 0.1f, 0.4f})
      for (float Q : {0.3f, 0.707f, 5.0f}) {
        const FRat<float> h = design_biquad(type, f, Q, 6.0f).eval_inv_z();
        CHECK(h.numer.coefs.rows() == 3 && h.denom.coefs.rows() == 3, "design_biquad(%s)", type);
        for (int i = 0; i < 3; i++) CHECK(std::isfinite(h.numer.coefs(i)) && std::isfinite(h.denom.coefs(i)), "design_biquad(%s): finite", type);
      }
  // ---- polynomial roots: random real polynomials built from known roots
  {
    Vecf r = randn(1);
    (void) r;
    for (int deg = 1; deg <= 12; deg++) {
      std::vector<double> c(1, 1.0);
      std::vector<cdouble> racines;
      for (int i = 0; i < deg; i++) {
        const double x0 = -0.9 + 1.8 * (i + 0.5) / deg;
        racines.push_back(x0);
        std::vector<double> nc(c.size() + 1, 0.0);
        for (size_t k = 0; k < c.size(); k++) { nc[k] += c[k]; nc[k + 1] -= x0 * c[k]; }
        c = nc;
      }
      Vecf co(deg + 1);
      for (int k = 0; k <= deg; k++) co(k) = (float) c[deg - k];       // ascending powers
      Poly<float> p(co);
      const Veccf z = p.roots();
      CHECK(z.rows() == deg, "roots: degree %d gave %d roots", deg, (int) z.rows());
      for (int i = 0; i < z.rows(); i++) {
        double best = 1e9;
        for (auto &q : racines) best = std::min(best, (double) std::abs(cdouble(z(i)) - q));
        CHECK(best < (deg <= 8 ? 2e-3 : 5e-2), "roots: degree %d root %d off by %g", deg, i, best);
      }
    }
  }
  // ---- interpolators: coefficient vectors over the whole phase range
  {
    InterpolateurSincConfig cfg;
    for (int K : {2, 7, 15, 31, 127})
      for (int nph : {16, 256, 1000}) {
        cfg.ncoefs = K;
        cfg.nphases = nph;
        cfg.fcut = 0.4f;
        auto it = itrp_sinc<cfloat>(cfg);
        for (float τ : {0.0f, 0.001f, 0.5f, 0.999f, 0.99999f}) {
          const Vecf h = it->coefs(τ);
          CHECK(h.rows() == K, "itrp_sinc coefs size");
          for (int i = 0; i < K; i++) CHECK(std::isfinite(h(i)), "itrp_sinc coefs finite");
        }
      }
    for (int d : {1, 2, 3, 5, 7}) {
      auto it = itrp_lagrange<float>(d);
      for (float τ : {0.0f, 0.25f, 0.75f, 0.999f}) {
        const Vecf h = it->coefs(τ);
        double s = 0;
        for (int i = 0; i < h.rows(); i++) s += h(i);
        CHECK(h.rows() == d + 1 && std::abs(s - 1) < 1e-4, "itrp_lagrange(%d) at %g: %d coefs, sum %g", d, τ, (int) h.rows(), s);
      }
    }
    auto lin = itrp_lineaire<float>();
    const Vecf h = lin->coefs(0.3f);
    CHECK(h.rows() == 2 && std::abs(h(0) - 0.7f) < 1e-6f && std::abs(h(1) - 0.3f) < 1e-6f, "itrp_lineaire");
    auto cs = itrp_cspline<float>();
    const Vecf hc = cs->coefs(0.5f);
    double s = 0;
    for (int i = 0; i < hc.rows(); i++) s += hc(i);
    CHECK(std::abs(s - 1) < 1e-4, "itrp_cspline: sum %g", s);
  }
  printf(nfail ? "FAILED (%d)\n" : "DESIGN LAYER OK\n", nfail);
  return nfail ? 1 : 0;
}
