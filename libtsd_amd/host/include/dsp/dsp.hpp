// dsp/dsp.hpp -- English names of the core of the hot-path API (libtsd core/include/dsp/dsp.hpp:
// FilterGen / Filter aliases :462-472, resample :499-503).  libtsd's English layer forwards every
// call to its French twin; here the English API is the same objects under aliases, so dsp:: and
// tsd:: code can be mixed freely.  The filter and Fourier names live in dsp/filter.hpp and
// dsp/fourier.hpp, like in libtsd.
#pragma once
#include "tsd/tsd.hpp"

namespace dsp {
using tsd::cfloat;
using tsd::sptr;
template <typename T> using Vector = tsd::Vecteur<T>;
using Vecf = tsd::Vecf;
using Veccf = tsd::Veccf;
template <typename Te, typename Ts = Te> using FilterGen = tsd::FiltreGen<Te, Ts>;   // dsp/dsp.hpp:462-472
template <typename Te, typename Ts = Te, typename Tc = tsd::Void> using Filter = tsd::Filtre<Te, Ts, Tc>;
using tsd::linspace;
using tsd::randn;
using tsd::randcn;
using tsd::sigcos;
using tsd::sigsin;
using tsd::sigexp;
// dsp::resample (dsp/dsp.hpp:499-503)
template <typename T> Vector<T> resample(const Vector<T> &x, float ratio) { return tsd::rééchan(x, ratio); }
}  // namespace dsp
