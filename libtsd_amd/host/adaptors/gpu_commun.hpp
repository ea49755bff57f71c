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
#include <cstdlib>
#include <cstring>
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

// bytes from anywhere to anywhere: plain memcpy between host buffers, the C ABI's copy as soon as one
// side is device memory (resident vectors)
inline void copie_octets(void *dst, const void *src, size_t octets)
{
  if (octets == 0) return;
  if (tsdgpu_is_device_pointer(dst) || tsdgpu_is_device_pointer(src)) {
    if (tsdgpu_memcpy(dst, src, octets, nullptr)) gpu_fail("copy");
  } else {
    std::memcpy(dst, src, octets);
  }
}
// grow-only device scratch owned by an adaptor object
struct TamponGpu {
  void *p = nullptr;
  size_t cap = 0;
  TamponGpu() {}
  TamponGpu(const TamponGpu &) = delete;
  TamponGpu &operator=(const TamponGpu &) = delete;
  ~TamponGpu() { tsdgpu_free(p); }
  void *reserve(size_t octets)
  {
    if (octets > cap) {
      tsdgpu_free(p);
      p = nullptr;
      cap = 0;
      if (tsdgpu_malloc(&p, octets + octets / 8 + 256)) gpu_fail("device scratch");
      cap = octets + octets / 8 + 256;
    }
    return p;
  }
};

// ---- several GPUs behind one operator object --------------------------------------------------------
// A large HOST vector handed to a FIR / SOS / resampler object is cut into contiguous chunks, one per
// GPU of the node (tsdgpu_sharded_step_host: chunk + the operator's small halo per device, all devices
// at once).  The choice is made at the object's first step() and kept (the stream state then lives in
// the sharded handle): shards = the device count when there are several, or tsd_amd::fixe_fragments(n)
// / TSD_AMD_SHARDS=n (n logical shards spread over the devices present; 0 or 1: off).
int &fragments_forces();                 // -1: automatic
inline int nb_fragments()
{
  if (fragments_forces() >= 0) return fragments_forces();
  static const int n = [] {
    if (const char *e = std::getenv("TSD_AMD_SHARDS")) return std::atoi(e);
    const int d = tsdgpu_device_count();
    return d > 1 ? d : 0;
  }();
  return n;
}
constexpr int SEUIL_FRAGMENTS = 1 << 22;     // samples: below this one GPU is as good
template <typename VX, typename VY> inline bool choisit_fragments(const VX &x, const VY &y)
{
  return nb_fragments() >= 2 && x.rows() >= SEUIL_FRAGMENTS && !tsdgpu_is_device_pointer(x.data()) &&
         !(y.rows() > 0 && tsdgpu_is_device_pointer(y.data()));
}
template <typename VX> inline void exige_hote_fragments(const VX &x, const char *qui)
{
  if (x.rows() > 0 && tsdgpu_is_device_pointer(x.data()))
    échec("{}: this object runs in multi-GPU host mode (its first step() took a large host vector); resident vectors need their own object", qui);
}

// step() of a stage whose output length differs from its input length: `lance(ptr)` fills cap
// elements.  x and y may be the same object (y = f->step(y) style chains): then a temporary is filled.
template <typename VX, typename VY, typename F> inline void sortie_variable(const VX &x, VY &y, long long cap, F lance)
{
  if (cap < 0 || cap > 0x7fffffffLL) échec("output size {} not representable in a Vecteur", cap);
  if ((const void *) x.data() == (const void *) y.data() && y.rows() != (int) cap) {
    VY tmp;
    tmp.resize((int) cap);
    lance(tmp.data());
    y = std::move(tmp);
    return;
  }
  dimensionne(y, (int) cap);
  lance(y.data());
}

}  // namespace tsd_amd
