// gpu_commun.hpp -- shared by the adaptor translation units (libtsd_amd/host/adaptors/*.cc).
//
// The adaptor TUs are the drop-in boundary on the C++ side: they define libtsd's factories
// (filtre_rif, filtre_sois, filtre_reechan, ... ) and FFTPlan / FiltreGen subclasses on top of the
// C ABI of include/tsdgpu.h and use ONLY the part of libtsd's API that exists in libtsd's own
// headers.  They therefore compile against EITHER header set, unchanged:
//   * this repository's mirror   (-Ilibtsd_amd/host/include)            -> libtsd_host.so
//   * libtsd's own headers       (-I<libtsd>/core/include, fmt header-only) -> objects whose
//     mangled symbols are the ones libtsd's call sites reference (tests/test_boundary_ref_headers.py
//     compiles them that way and compares the symbol tables with libtsd's own objects).
// Rules that keep it so: Vecteur<T> is touched only through data() / rows() / resize() /
// operator() / the (n) constructor; errors go through échec("literal {}", ...); nothing here
// depends on a member the mirror adds.
#pragma once
#include "tsd/tsd.hpp"
#include "tsd/filtrage.hpp"
#include "tsd/fourier.hpp"
#include "tsdgpu.h"
#include <complex>
#include <string>
#include <vector>

namespace tsd_amd {
using namespace tsd;   // échec / msg are functions of namespace tsd in the mirror, macros over tsd:: functions in libtsd

template <typename T> constexpr int dtype_of()
{
  return (std::is_same_v<T, std::complex<float>> || std::is_same_v<T, std::complex<double>>) ? TSDGPU_C64 : TSDGPU_F32;
}
// status code of the C ABI -> libtsd's error path (échec: logger level 4, then an exception)
[[noreturn]] inline void gpu_fail(const char *what)
{
  échec("{}: {}", what, std::string(tsdgpu_last_error()));
}
// output vector of a step(): left alone when it already has the right size (so a caller may hand in
// a vector mapped on device memory, TabT::map(ptr, n), and the data never leaves the GPU)
template <typename V> inline void dimensionne(V &y, int n)
{
  if (y.rows() != n) y.resize(n);
}

// step() of a stage whose output length differs from its input length: `lance(ptr)` fills cap
// elements.  x and y may be the same object (y = f->step(y) style chains): then a temporary is filled.
template <typename VX, typename VY, typename F> inline void sortie_variable(const VX &x, VY &y, long long cap, F lance)
{
  if (cap < 0 || cap > 0x7fffffffLL) échec("output size {} not representable in a Vecteur", cap);
  if ((const void *) x.data() == (const void *) y.data() && y.rows() != (int) cap) {
    VY tmp((int) cap);
    lance(tmp.data());
    y = std::move(tmp);
    return;
  }
  dimensionne(y, (int) cap);
  lance(y.data());
}

}  // namespace tsd_amd
