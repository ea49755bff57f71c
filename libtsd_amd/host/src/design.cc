// design.cc -- mirror runtime: the host-side design helpers the hot path's configurations need
// (run once per filter, on the CPU, like in libtsd).  Mirror only: against libtsd itself these come
// from libtsd's own rif-fen.cc / fenetres.cc / rii.cc / frat.cc / filtrage.cc.
#include "tsd/filtrage.hpp"

namespace tsd {
namespace filtrage {

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
  const Vecf f = fenêtre(fen, n, true);
  Vecf h;
  const entier no2 = (n - 1) / 2;
  if (type == "lp" || type == "pb") {
    h = coefs_filtre_sinc(n, fc);
  } else if (type == "hp" || type == "ph") {
    h = -coefs_filtre_sinc(n, fc);
    h(no2) += 1.0f;
  } else if (type == "bp" || type == "pm" || type == "sb") {
    // rif-fen.cc:61-78: the low-pass of half the band width, moved to the band centre by 2 cos(ωc k), k = -no2 .. no2
    // (n samples only when n is odd: the reference's element-wise product refuses an even n); stop-band = delta - band-pass
    if (n % 2 == 0) échec("design_rif_fen: type '{}' needs an odd number of coefficients (n = {})", type, n);
    const float ωc = (float) (π * (double) (fc2 + fc)), δf = (fc2 - fc) / 2;
    h = coefs_filtre_sinc(n, δf);
    for (entier i = 0; i < n; i++) h(i) *= 2.0f * std::cos(ωc * (float) (i - no2));
    if (type == "sb") {
      h = -h;
      h(no2) += 1.0f;
    }
  } else {
    échec("design_rif_fen: invalid type '{}' (lp/pb, hp/ph, bp/pm, sb)", type);
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
// Analog prototypes with cut-off 1 rad/s as zeros / poles / gain, H(s) = k prod(s - z) / prod(s - p), DC gain 1 like the
// reference's (rii.cc:195-404 forces it through numer.mlt).  Textbook forms: Butterworth and Chebyshev poles on the circle /
// ellipse; inverse Chebyshev = the reciprocal pole set with zeros at j / cos θ; elliptic through Landen sequences
// (S. Orfanidis, "Lecture notes on elliptic filter design": degree equation, cd / sn / asn by descending / ascending
// Landen transformations).  All in double; rounded to float once, at the end of design_riia.
namespace {
typedef std::complex<double> cd_t;
struct Proto {
  std::vector<cd_t> z, p;
  double k = 1;
};
void gain_dc_unitaire(Proto &h)
{
  cd_t g = 1;
  for (const cd_t &p : h.p) g *= -p;
  for (const cd_t &z : h.z) g /= -z;
  h.k = g.real();
}
Proto proto_butterworth(int n)
{
  Proto h;
  for (int k = 1; k <= n; k++) h.p.push_back(std::polar(1.0, π * (2.0 * k + n - 1) / (2.0 * n)));
  gain_dc_unitaire(h);
  return h;
}
Proto proto_tchebychev(int n, double ondulation_dB, bool inverse)
{
  Proto h;
  const double e2 = std::pow(10.0, ondulation_dB / 10) - 1;
  const double eps = inverse ? 1 / std::sqrt(e2) : std::sqrt(e2), a = std::asinh(1 / eps) / n;
  for (int k = 1; k <= n; k++) {
    const double th = (2.0 * k - 1) * π / (2.0 * n);
    const cd_t q(-std::sinh(a) * std::sin(th), std::cosh(a) * std::cos(th));
    h.p.push_back(inverse ? 1.0 / q : q);
    if (inverse && std::fabs(std::cos(th)) > 1e-12) h.z.push_back(cd_t(0, 1 / std::cos(th)));   // (odd n: the middle zero is at infinity)
  }
  gain_dc_unitaire(h);
  return h;
}
// Landen sequence of a modulus: v_{i+1} = (v_i / (1 + sqrt(1 - v_i^2)))^2
std::vector<double> landen(double k)
{
  std::vector<double> v;
  for (int i = 0; i < 12 && k > 1e-18; i++) {
    const double kp = std::sqrt((1 - k) * (1 + k));
    k = (k / (1 + kp)) * (k / (1 + kp));
    v.push_back(k);
  }
  return v;
}
cd_t jacobi_cd(cd_t u, double k)     // cd(u K, k), u in units of the quarter period
{
  const std::vector<double> v = landen(k);
  cd_t w = std::cos(u * (π / 2));
  for (int i = (int) v.size() - 1; i >= 0; i--) w = (1 + v[i]) * w / (1.0 + v[i] * w * w);
  return w;
}
cd_t jacobi_sn(cd_t u, double k)
{
  const std::vector<double> v = landen(k);
  cd_t w = std::sin(u * (π / 2));
  for (int i = (int) v.size() - 1; i >= 0; i--) w = (1 + v[i]) * w / (1.0 + v[i] * w * w);
  return w;
}
cd_t jacobi_asn(cd_t w, double k)    // inverse of sn, in units of the quarter period
{
  std::vector<double> v = landen(k);
  double avant = k;
  for (size_t i = 0; i < v.size(); i++) {
    w = 2.0 * w / ((1 + v[i]) * (1.0 + std::sqrt(1.0 - avant * avant * w * w)));
    avant = v[i];
  }
  return (2 / π) * std::asin(w);
}
Proto proto_elliptique(int n, double rp_dB, double rs_dB)
{
  Proto h;
  const double ep = std::sqrt(std::pow(10.0, rp_dB / 10) - 1), es = std::sqrt(std::pow(10.0, rs_dB / 10) - 1);
  const double k1 = ep / es, k1p = std::sqrt((1 - k1) * (1 + k1));
  const int L = n / 2;
  // degree equation: k' = k1'^n prod sn(u_i K1', k1')^4, k = sqrt(1 - k'^2)
  double kp = std::pow(k1p, n);
  for (int i = 1; i <= L; i++) kp *= std::pow(std::abs(jacobi_sn((2.0 * i - 1) / n, k1p)), 4);
  const double k = std::sqrt((1 - kp) * (1 + kp));
  const cd_t v0 = cd_t(0, -1) * jacobi_asn(cd_t(0, 1) / ep, k1) / (double) n;
  for (int i = 1; i <= L; i++) {
    const double u = (2.0 * i - 1) / n;
    const cd_t zeta = jacobi_cd(u, k);
    const cd_t z = cd_t(0, 1) / (k * zeta), pl = cd_t(0, 1) * jacobi_cd(cd_t(u, 0) - cd_t(0, 1) * v0, k);
    h.z.push_back(z);
    h.z.push_back(std::conj(z));
    h.p.push_back(pl);
    h.p.push_back(std::conj(pl));
  }
  if (n & 1) h.p.push_back(cd_t(0, 1) * jacobi_sn(cd_t(0, 1) * v0, k));
  gain_dc_unitaire(h);
  return h;
}
}  // namespace

// design_riia (filtrage.hpp:666-701; rii.cc:405-487): analog prototype -> low-pass / high-pass at the pre-warped
// frequency -> bilinear transform (fe = 1), the result in pole / zero form.  "butt" + "lp" keeps the reference's own
// float operation order (configs[3]'s design, compared with the oracle's restatement); the other prototypes and the
// high-pass go through the double-precision pipeline above.
static FRat<cfloat> design_riia_generique(entier n, bool passe_haut, const Proto &proto, float fc)
{
  const double wa = 2 * std::tan(π * (double) fc);                   // pre-warping, fe = 1
  std::vector<cd_t> z, p;
  double k = proto.k;
  const int np = (int) proto.p.size(), nz = (int) proto.z.size();
  if (!passe_haut) {                                                  // s -> s / wa
    for (const cd_t &r : proto.z) z.push_back(r * wa);
    for (const cd_t &r : proto.p) p.push_back(r * wa);
    k *= std::pow(wa, np - nz);
  } else {                                                            // s -> wa / s
    cd_t g = k;
    for (const cd_t &r : proto.z) { z.push_back(wa / r); g *= -r; }
    for (const cd_t &r : proto.p) { p.push_back(wa / r); g /= -r; }
    for (int i = nz; i < np; i++) z.push_back(0.0);
    k = g.real();
  }
  // bilinear: s = 2 (z - 1) / (z + 1); zeros at infinity land on z = -1
  Veccf zd((entier) p.size()), pd((entier) p.size());
  cd_t g = k;
  for (size_t i = 0; i < p.size(); i++) {
    pd((entier) i) = cfloat((2.0 + p[i]) / (2.0 - p[i]));
    g /= (2.0 - p[i]);
    if (i < z.size()) {
      zd((entier) i) = cfloat((2.0 + z[i]) / (2.0 - z[i]));
      g *= (2.0 - z[i]);
    } else {
      zd((entier) i) = cfloat(-1.f, 0.f);
    }
  }
  (void) n;
  FRat<cfloat> h;
  h.numer = Poly<cfloat>::from_roots(zd);
  h.denom = Poly<cfloat>::from_roots(pd);
  h.numer.mlt = cfloat((float) g.real(), 0.f);
  h.denom.mlt = cfloat(1.f, 0.f);
  return h;
}

FRat<cfloat> design_riia(entier n, cstring type, cstring prototype, float fc, float δ_bp, float δ_bc)
{
  if (n < 1) échec("design_riia: order {}", n);
  const bool lp = type == "lp" || type == "pb", hp = type == "hp" || type == "ph";
  if (!lp && !hp) échec("design_riia: type '{}' is not built (have lp / pb, hp / ph; the reference has no band types either, rii.cc:438-441)", type);
  const bool butt = prototype.substr(0, 1) == "b";
  if (!butt || hp) {
    Proto proto;
    if (butt) proto = proto_butterworth(n);
    else if (prototype == "cheb1") proto = proto_tchebychev(n, δ_bp, false);
    else if (prototype == "cheb2") proto = proto_tchebychev(n, δ_bc, true);
    else if (prototype.substr(0, 5) == "ellip") proto = proto_elliptique(n, δ_bp, δ_bc);
    else échec("design_riia: unknown analog prototype '{}' (butt, cheb1, cheb2, ellip)", prototype);
    return design_riia_generique(n, hp, proto, fc);
  }
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

// ---- design: biquads of the Audio-EQ-Cookbook (the source rii.cc:576 names; rii.cc:489-507,577-667) ----
// H(z^-1) = (b0 + b1 z^-1 + b2 z^-2) / (a0 + a1 z^-1 + a2 z^-2), normalised by a0.  With A = 10^(gain/40),
// w = 2 pi f, alpha = sin w / (2 Q), beta = sqrt(2 A) (the shelf slope the reference fixes):
FRat<float> design_biquad(const BiquadSpec &sp)
{
  const float A = std::sqrt(std::pow(10.0f, sp.gain_dB / 20)), w = (float) (2 * π * sp.f);
  const float c = std::cos(w), sn = std::sin(w), al = sn / (2 * sp.Q), be = std::sqrt(2 * A);
  struct { float b[3], a[3]; } k{};
  auto pose = [&](float b0, float b1, float b2, float a0, float a1, float a2) { k = {{b0, b1, b2}, {a0, a1, a2}}; };
  const float ap = A + 1, am = A - 1;
  switch (sp.type) {
    case BiquadSpec::PASSE_BAS: pose((1 - c) / 2, 1 - c, (1 - c) / 2, 1 + al, -2 * c, 1 - al); break;
    case BiquadSpec::PASSE_HAUT: pose((1 + c) / 2, -(1 + c), (1 + c) / 2, 1 + al, -2 * c, 1 - al); break;
    case BiquadSpec::PASSE_BANDE: pose(al, 0, -al, 1 + al, -2 * c, 1 - al); break;
    case BiquadSpec::COUPE_BANDE: pose(1, -2 * c, 1, 1 + al, -2 * c, 1 - al); break;
    case BiquadSpec::RESONATEUR: pose(1 + al * A, -2 * c, 1 - al * A, 1 + al / A, -2 * c, 1 - al / A); break;
    case BiquadSpec::PLATEAU_BF:
      pose(A * (ap - am * c + be * sn), 2 * A * (am - ap * c), A * (ap - am * c - be * sn), ap + am * c + be * sn, -2 * (am + ap * c), ap + am * c - be * sn);
      break;
    case BiquadSpec::PLATEAU_HF:
      pose(A * (ap + am * c + be * sn), -2 * A * (am + ap * c), A * (ap + am * c - be * sn), ap - am * c + be * sn, 2 * (am - ap * c), ap - am * c - be * sn);
      break;
    default: échec("Type de biquad invalide ({}).", (int) sp.type);
  }
  const float a0 = k.a[0];
  return FRat<float>::rii(Vecf::valeurs({k.b[0] / a0, k.b[1] / a0, k.b[2] / a0}), Vecf::valeurs({1.0f, k.a[1] / a0, k.a[2] / a0}));
}
FRat<float> design_biquad(cstring type, float f, float Q, float gain_dB)
{
  static const struct { const char *nom; BiquadSpec::Type t; } noms[] = {
      {"lp", BiquadSpec::PASSE_BAS}, {"pb", BiquadSpec::PASSE_BAS}, {"hp", BiquadSpec::PASSE_HAUT}, {"ph", BiquadSpec::PASSE_HAUT},
      {"bp", BiquadSpec::PASSE_BANDE}, {"passe-bande", BiquadSpec::PASSE_BANDE}, {"cb", BiquadSpec::COUPE_BANDE},
      {"notch", BiquadSpec::COUPE_BANDE}, {"sb", BiquadSpec::COUPE_BANDE}, {"plateau-bf", BiquadSpec::PLATEAU_BF},
      {"plateau-hf", BiquadSpec::PLATEAU_HF}, {"res", BiquadSpec::RESONATEUR}};
  BiquadSpec sp;
  sp.f = f;
  sp.Q = Q;
  sp.gain_dB = gain_dB;
  for (const auto &e : noms)
    if (type == e.nom) sp.type = e.t;          // (an unknown name leaves the low-pass default, like the reference)
  return design_biquad(sp);
}

// ---- polynomial roots -----------------------------------------------------------------------
}  // namespace filtrage

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

namespace filtrage {


// ---- first-order smoother helpers (src/filtrage/filtrage.cc:121-139) ----------------------------------
float lexp_coef(Fréquence fc) { return (float) (1.0 - std::exp(-fc.value * 2 * π)); }
float lexp_tc_vers_coef(float τ) { return lexp_coef((float) (1.0 / (2 * π * τ))); }
Fréquence lexp_fcoupure(float γ) { return (float) (-std::log(1.0 - γ) / (2 * π)); }
float lexp_coef_vers_tc(float γ) { return (float) (1.0 / (2 * π * lexp_fcoupure(γ).value)); }


}  // namespace filtrage
}  // namespace tsd
