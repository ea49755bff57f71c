// commun.cc -- mirror runtime: logger, default-seeded random generators, prochaine_puissance_de_2,
// tampon_création (libtsd core/src/tsd.cc:45-126,173,287-291,307-381,410-483).  Mirror only: against
// libtsd itself these come from libtsd's own tsd.cc.
#include "tsd/tsd.hpp"

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

}  // namespace tsd

// ---- tampon_création (core/src/tsd.cc:307-381) ----------------------------------------------------
namespace tsd {
namespace {
template <typename T> struct TamponBlocs : Sink<T, entier> {
  entier N = 0, rempli = 0;
  fonction<void(const Vecteur<T> &)> callback;
  Vecteur<T> bloc;
  TamponBlocs(entier N_, fonction<void(const Vecteur<T> &)> cb) : callback(std::move(cb)) { Configurable<entier>::configure(N_); }
  void configure_impl(const entier &N_) override
  {
    N = N_;
    rempli = 0;
  }
  void step(const Vecteur<T> &x) override
  {
    if (rempli == 0 && x.rows() == N) {                     // a ready-made block goes straight through
      if (callback) callback(x);
      return;
    }
    if (bloc.rows() == 0 && N > 0) bloc.resize(N);
    entier i = 0;
    const entier n = x.rows();
    while (i < n) {
      const entier k = std::min(N - rempli, n - i);
      bloc.segment(rempli, k) = x.segment(i, k);
      i += k;
      rempli += k;
      if (rempli == N) {
        if (callback) callback(bloc);
        rempli = 0;
      }
    }
  }
};
}  // namespace
template <typename T> sptr<Sink<T, entier>> tampon_création(entier N, fonction<void(const Vecteur<T> &)> callback)
{
  return std::make_shared<TamponBlocs<T>>(N, std::move(callback));
}
template sptr<Sink<float, entier>> tampon_création<float>(entier, fonction<void(const Vecf &)>);
template sptr<Sink<cfloat, entier>> tampon_création<cfloat>(entier, fonction<void(const Veccf &)>);
}  // namespace tsd
