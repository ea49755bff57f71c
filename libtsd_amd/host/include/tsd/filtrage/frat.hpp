// tsd/filtrage/frat.hpp -- mirror of the part of libtsd's polynomial / rational-function types
// (core/include/tsd/filtrage/frat.hpp:16-500 Poly, :501-900 FRat) that the filter factories of the
// hot path read: same member names, same representation, so the adaptor TUs compile against either.
//
// Representation kept from libtsd: a Poly is a coefficient list in ASCENDING powers of its
// variable, or (mode_racines) a list of roots with a leading multiplier, mlt * prod (z - r_i);
// a FRat is a quotient of two polynomials IN z.  FRat::rii(a, b) takes the difference-equation
// coefficients (powers of z^-1) and stores H as a fraction in z (frat.hpp:693-707);
// eval_inv_z() goes back to powers of z^-1 (frat.hpp:652-671) -- that is what filtre_rii reads.
#pragma once
#include "tsd/tsd.hpp"

namespace tsd {

template <typename T> struct Poly {
  Vecteur<T> coefs;
  std::string vname = "z";
  bouléen mode_racines = false;
  T mlt = T(1.0f);

  Poly() {}
  Poly(const Vecteur<T> &c) : coefs(c) {}
  static Poly from_roots(const Vecteur<T> &r)
  {
    Poly p;
    p.coefs = r;
    p.mode_racines = true;
    p.mlt = T(1.0f);
    return p;
  }
  // product of two coefficient-form polynomials (convolution of the lists)
  friend Poly operator*(const Poly &a, const Poly &b)
  {
    const Poly ca = a.vers_coefs(), cb = b.vers_coefs();
    const entier na = ca.coefs.rows(), nb = cb.coefs.rows();
    Poly r;
    r.vname = a.vname;
    if (na == 0 || nb == 0) return r;
    r.coefs = Vecteur<T>::zeros(na + nb - 1);
    for (entier i = 0; i < na; i++)
      for (entier j = 0; j < nb; j++) r.coefs.data()[i + j] += ca.coefs.data()[i] * cb.coefs.data()[j];
    return r;
  }
  friend Poly operator*(const Poly &a, const T &s)
  {
    Poly r = a;
    if (r.mode_racines)
      r.mlt = r.mlt * s;
    else
      for (entier i = 0; i < r.coefs.rows(); i++) r.coefs.data()[i] *= s;
    return r;
  }
  // z^k as a polynomial
  static Poly monome(entier k)
  {
    Poly p;
    p.coefs = Vecteur<T>::zeros(k + 1);
    p.coefs.data()[k] = T(1.0f);
    return p;
  }
  // expanded form: mlt * prod (z - r_i) multiplied out (frat.hpp:91-116)
  Poly vers_coefs() const
  {
    if (!mode_racines) return *this;
    Poly r;
    r.vname = vname;
    r.coefs = Vecteur<T>::ones(1);
    for (entier i = 0; i < coefs.rows(); i++) {
      Poly m;
      m.coefs = Vecteur<T>::hote(2);
      m.coefs.data()[0] = -coefs.data()[i];
      m.coefs.data()[1] = T(1.0f);
      r = r * m;
    }
    return r * mlt;
  }
  // roots: the list itself in mode_racines (frat.cc:43-46), else a Durand-Kerner iteration in
  // double precision (the reference calls Eigen's PolynomialSolver there: "parity unpinned")
  Vecteur<cfloat> roots() const;
};

template <typename T> struct FRat {
  Poly<T> numer, denom;
  FRat() {}
  FRat(const Poly<T> &n, const Poly<T> &d) : numer(n), denom(d) {}

  // H(z) -> the same function written in powers of z^-1 (and back: the map is an involution up to
  // the common power of z): coefficient lists reversed, the shorter side padded
  FRat eval_inv_z() const
  {
    const Vecteur<T> a = numer.vers_coefs().coefs, b = denom.vers_coefs().coefs;
    const entier nn = a.rows(), nd = b.rows();
    FRat r;
    r.numer = Poly<T>(a.reverse());
    r.denom = Poly<T>(b.reverse());
    if (nn > nd)
      r.denom = r.denom * Poly<T>::monome(nn - nd);
    else if (nd > nn)
      r.numer = r.numer * Poly<T>::monome(nd - nn);
    return r;
  }
  // b0 y_n + b1 y_{n-1} + ... = a0 x_n + a1 x_{n-1} + ...
  static FRat rii(const Vecteur<T> &a, const Vecteur<T> &b) { return FRat(Poly<T>(a), Poly<T>(b)).eval_inv_z(); }
  static FRat rif(const Vecteur<T> &a)
  {
    const entier K = a.rows();
    if (K == 0) return {};
    Vecteur<T> c = Vecteur<T>::zeros(K);
    c.data()[K - 1] = T(1.0f);
    return FRat(Poly<T>(a.reverse()), Poly<T>(c));
  }
  // denominator = z^(K-1) exactly
  bouléen est_rif() const
  {
    if (denom.mode_racines) return false;
    const entier Nd = denom.coefs.rows();
    if (Nd == 0) return false;
    for (entier i = 0; i + 1 < Nd; i++)
      if (denom.coefs.data()[i] != T(0.0f)) return false;
    return denom.coefs.data()[Nd - 1] == T(1.0f);
  }
  Vecteur<T> coefs_rif() const
  {
    if (!est_rif()) échec("FRat::coefs_rif(): ce filtre n'est pas un filtre RIF.");
    return numer.coefs.reverse();
  }
};

}  // namespace tsd
