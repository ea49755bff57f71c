// commun.cc -- mirror runtime: logger, default-seeded random generators, prochaine_puissance_de_2,
// tampon_création (libtsd core/src/tsd.cc:45-126,173,287-291,307-381,410-483).  Mirror only: against
// libtsd itself these come from libtsd's own tsd.cc.
#include "tsd/tsd.hpp"
#include "tsdgpu.h"
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>

namespace tsd {

// ---- device residency plumbing (tsd.hpp: detail::*) over the C ABI ----------------------------------
namespace detail {
// Large vectors come and go with every one-shot call (filtrer() allocates its output, libtsd-style):
// page-locking 512 MiB costs ~85 ms and a device allocation + free synchronises the GPU, so freed
// blocks are kept and handed out again (best fit within 25 %, at most `limite` bytes kept per kind).
namespace {
struct CacheBlocs {
  // a block belongs to the device that was current when it was allocated (-1: page-locked host memory)
  typedef std::pair<int, size_t> Cle;
  std::mutex m;
  std::multimap<Cle, void *> libres;
  std::unordered_map<void *, Cle> tailles;        // (device, capacity) of every live block of this kind
  size_t garde = 0, limite;
  explicit CacheBlocs(size_t lim) : limite(lim) {}
  void *prend(size_t octets, int dev)
  {
    std::lock_guard<std::mutex> l(m);
    auto it = libres.lower_bound(Cle(dev, octets));
    if (it == libres.end() || it->first.first != dev || it->first.second > octets + octets / 4) return nullptr;
    void *p = it->second;
    garde -= it->first.second;
    libres.erase(it);
    return p;
  }
  void note(void *p, size_t octets, int dev)
  {
    std::lock_guard<std::mutex> l(m);
    tailles[p] = Cle(dev, octets);
  }
  // true: kept for reuse; false: the caller frees it
  bool rend(void *p)
  {
    std::lock_guard<std::mutex> l(m);
    auto it = tailles.find(p);
    if (it == tailles.end()) return false;
    if (garde + it->second.second > limite) {
      tailles.erase(it);
      return false;
    }
    garde += it->second.second;
    libres.emplace(it->second, p);
    return true;
  }
};
size_t limite_cache(const char *var, size_t defaut_mio)
{
  const char *e = std::getenv(var);
  return ((size_t) (e ? std::atoll(e) : (long long) defaut_mio)) << 20;
}
CacheBlocs &cache_gpu()
{
  static CacheBlocs *c = new CacheBlocs(limite_cache("TSD_AMD_CACHE_GPU_MIB", 8192));     // never destroyed: outlives the HIP runtime teardown
  return *c;
}
CacheBlocs &cache_hote()
{
  static CacheBlocs *c = new CacheBlocs(limite_cache("TSD_AMD_CACHE_HOTE_MIB", 4096));
  return *c;
}
}  // namespace

void *gpu_alloc(size_t octets)
{
  const int dev = tsdgpu_current_device();
  if (void *q = cache_gpu().prend(octets, dev)) return q;
  void *p = nullptr;
  if (tsdgpu_malloc(&p, octets)) échec("Vecteur (device): {}", tsdgpu_last_error());
  cache_gpu().note(p, octets, dev);
  return p;
}
void gpu_free(void *p)
{
  if (p && !cache_gpu().rend(p)) tsdgpu_free(p);
}
void *hote_alloc(size_t octets, bool *verrouillée)
{
  void *p = nullptr;
  static const bool gpu = tsdgpu_device_count() > 0 && std::getenv("TSD_AMD_NO_PINNED") == nullptr;
  if (gpu) {
    *verrouillée = true;
    if (void *q = cache_hote().prend(octets, -1)) return q;
    if (tsdgpu_malloc_host(&p, octets) == 0 && p) {
      cache_hote().note(p, octets, -1);
      return p;
    }
  }
  *verrouillée = false;
  p = std::malloc(octets);
  if (!p) échec("Vecteur: out of memory ({} bytes)", octets);
  return p;
}
void hote_free(void *p, bool verrouillée)
{
  if (!p) return;
  if (!verrouillée) std::free(p);
  else if (!cache_hote().rend(p)) tsdgpu_free_host(p);
}
void gpu_copie(void *dst, const void *src, size_t octets)
{
  if (tsdgpu_memcpy(dst, src, octets, nullptr)) échec("Vecteur (device copy): {}", tsdgpu_last_error());
}
void gpu_zero(void *p, size_t octets)
{
  if (tsdgpu_memset(p, 0, octets, nullptr)) échec("Vecteur (device setZero): {}", tsdgpu_last_error());
}
bool est_ptr_gpu(const void *p) { return tsdgpu_is_device_pointer(p) != 0; }
void gpu_op_vec(int op, bool complexe, void *dst, const void *a, const void *b, float s_re, float s_im, size_t n)
{
  if (tsdgpu_vec_op(op, complexe ? TSDGPU_C64 : TSDGPU_F32, dst, a, b, s_re, s_im, (int64_t) n, nullptr))
    échec("Vecteur (device arithmetic): {}", tsdgpu_last_error());
}
void gpu_reduction(bool complexe, const void *a, size_t n, double *somme2, float *maxmin2, long long *imax)
{
  int64_t im = -1;
  if (tsdgpu_vec_reduce(complexe ? TSDGPU_C64 : TSDGPU_F32, a, (int64_t) n, somme2, maxmin2, &im, nullptr))
    échec("Vecteur (device reduction): {}", tsdgpu_last_error());
  if (imax) *imax = (long long) im;
}
bool &residence_active()
{
  static thread_local bool actif = false;
  return actif;
}
}  // namespace detail

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
Veccf sigexp(float f, entier n)
{
  Veccf x = Veccf::hote(n);
  const cdouble pas = std::polar(1.0, (double) f * 2 * π);
  cdouble r = 1;
  for (entier i = 0; i < n; i++) {
    if (i > 0 && i % 1000 == 0) r /= std::abs(r);          // the modulus is brought back to 1 every 1000 samples
    x(i) = cfloat(r);
    r *= pas;
  }
  return x;
}
Vecf sigcos(float f, entier n) { return real(sigexp(f, n)); }
Vecf sigsin(float f, entier n) { return imag(sigexp(f, n)); }

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
