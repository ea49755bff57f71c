// itrp.cc -- mirror runtime: the interpolators of libtsd's core/src/reechan/itrp.cc:10-160 (windowed
// sinc table, cardinal cubic spline table, linear, Lagrange).  Mirror only: against libtsd itself the
// interpolators are libtsd's own objects and filtre_itrp probes them through coefs() (adaptors/gpu_ra.cc).
#include "tsd/filtrage.hpp"

#include <cstring>
#include <mutex>
#include <string>

namespace tsd {
namespace filtrage {

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
  // (a one-shot rééchan() designs this table at every call -- 33 us for the 256 x 15 table of filtre_reechan: the last few
  // designs are kept, keyed by their parameters)
  struct Memo {
    std::mutex m;
    std::vector<std::pair<std::string, std::vector<float>>> tables;
  };
  static Memo *memo = new Memo();
  uint32_t fb;
  std::memcpy(&fb, &c.fcut, sizeof fb);
  const std::string clef = detail::fmt("{}/{}/{}/{}", nc, c.nphases, fb, c.fenetre);
  {
    std::lock_guard<std::mutex> g(memo->m);
    for (const auto &e : memo->tables)
      if (e.first == clef) {
        lut = e.second;
        return;
      }
  }
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
  std::lock_guard<std::mutex> g(memo->m);
  if (memo->tables.size() >= 8) memo->tables.erase(memo->tables.begin());
  memo->tables.emplace_back(clef, lut);
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
template <typename T> sptr<InterpolateurRIF<T>> itrp_cspline() { return std::make_shared<InterpolateurCSpline<T>>(); }
template sptr<InterpolateurRIF<float>> itrp_cspline<float>();
template sptr<InterpolateurRIF<cfloat>> itrp_cspline<cfloat>();

template <typename T> sptr<InterpolateurRIF<T>> itrp_sinc(const InterpolateurSincConfig &config)
{
  return std::make_shared<InterpolateurSinc<T>>(config);
}
template sptr<InterpolateurRIF<float>> itrp_sinc<float>(const InterpolateurSincConfig &);
template sptr<InterpolateurRIF<cfloat>> itrp_sinc<cfloat>(const InterpolateurSincConfig &);

template <typename T> sptr<InterpolateurRIF<T>> itrp_lineaire() { return std::make_shared<InterpolateurLineaire<T>>(); }
template <typename T> sptr<InterpolateurRIF<T>> itrp_lagrange(entier degré) { return std::make_shared<InterpolateurLagrange<T>>(degré); }
template sptr<InterpolateurRIF<float>> itrp_lineaire<float>();
template sptr<InterpolateurRIF<cfloat>> itrp_lineaire<cfloat>();
template sptr<InterpolateurRIF<float>> itrp_lagrange<float>(entier);
template sptr<InterpolateurRIF<cfloat>> itrp_lagrange<cfloat>(entier);

}  // namespace filtrage
}  // namespace tsd
