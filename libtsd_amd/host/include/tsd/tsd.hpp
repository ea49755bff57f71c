// tsd/tsd.hpp -- host-side mirror of the part of libtsd's root API (namespace tsd) that the
// streaming FIR / IIR / FFT / resample path touches, backed by the MI355X C ABI
// (include/tsdgpu.h).  Written from scratch; it keeps libtsd's NAMES, argument meaning and
// error behaviour for this path so existing call sites compile unchanged:
//   Vecteur<T> / Vecf / Veccf          core/include/tsd/tableau.hpp:290-1445 (subset)
//   FiltreGen / Filtre / Configurable  core/include/tsd/tsd.hpp:544-579,626-668
//   rééchan                            core/include/tsd/tsd.hpp:700-705
//   linspace, sigimp, randn ...        core/include/tsd/tsd.hpp:916-931, core/src/tsd.cc:179-483
// Not a reimplementation of libtsd's array runtime: only what the hot path and its tests use.
#pragma once
#define TSD_AMD_MIRROR 1   // tells the adaptor TUs which header set they are compiled against
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <memory>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

namespace tsd {

using entier = int;
using bouléen = bool;
using cfloat = std::complex<float>;
using cdouble = std::complex<double>;
template <typename T> using sptr = std::shared_ptr<T>;
template <typename T> using fonction = std::function<T>;
using cstring = const std::string &;
static const double π = 3.14159265358979323846;
static const float π_f = 3.14159265358979323846f;
struct Void {};

// ---- errors and logging (core/include/tsd/commun.hpp:41-54,132-178) -----------------------
// logger(niveau, message): niveau 0..5; échec() logs at level 4 then throws std::runtime_error
// (what libtsd's default logger does, core/src/tsd.cc:114-118).
using logger_t = std::function<void(int niveau, const std::string &msg)>;
logger_t &get_logger();
void set_logger(logger_t l);

namespace detail {
inline void fmt_rec(std::ostringstream &os, const char *f)
{
  os << f;
}
template <typename A, typename... R> void fmt_rec(std::ostringstream &os, const char *f, const A &a, const R &...r)
{
  for (; *f; f++) {
    if (f[0] == '{') {
      const char *e = std::strchr(f, '}');
      if (e) {
        os << a;
        fmt_rec(os, e + 1, r...);
        return;
      }
    }
    os << *f;
  }
}
template <typename... A> std::string fmt(const char *f, const A &...a)
{
  std::ostringstream os;
  fmt_rec(os, f, a...);
  return os.str();
}
}  // namespace detail

template <typename... A> void msg(const char *f, const A &...a)
{
  if (get_logger()) get_logger()(1, detail::fmt(f, a...));
}
template <typename... A> [[noreturn]] void échec(const char *f, const A &...a)
{
  const std::string m = detail::fmt(f, a...);
  if (get_logger()) get_logger()(4, m);
  throw std::runtime_error(m);
}
#define tsd_assertion(cond)                                                                  \
  do {                                                                                       \
    if (!(cond)) ::tsd::échec("assertion failed: {} ({}:{})", #cond, __FILE__, __LINE__);    \
  } while (0)

template <typename T> constexpr bool est_complexe()
{
  return std::is_same_v<T, cfloat> || std::is_same_v<T, cdouble>;
}

// ---- device residency ("device-buffer side-channel") -------------------------
// A Vecteur may live in GPU memory.  Every operator of the path takes such vectors as they are (the C
// ABI accepts device pointers), so a chain filtrer -> fft -> rééchan never crosses PCIe; only
// element-wise host code (operator(), arithmetic) refuses them.  Three ways in:
//   Vecteur<T>::sur_gpu(n) / x.vers_gpu()       explicit device vectors
//   Vecteur<T>::map(device_ptr, n)              foreign device memory (torch, hipMalloc)
//   { ResidenceGpu r; y = filtrer(h, x); }      while the guard lives, every vector an operator SIZES
//                                               on this thread (resize(): step() outputs, temporaries of
//                                               operator chains) is device memory; vectors built by
//                                               constructors / zeros() / valeurs() -- what host code fills
//                                               element by element -- stay on the host
// Host vectors of 1 MiB and more are page-locked, so the chunked staging pipeline of the C ABI runs
// their H2D / D2H copies asynchronously and in both directions at once.
namespace detail {
void *gpu_alloc(size_t octets);
void gpu_free(void *p);
void *hote_alloc(size_t octets, bool *verrouillée);      // page-locked when a GPU is present
void hote_free(void *p, bool verrouillée);
void gpu_copie(void *dst, const void *src, size_t octets);   // host<->device, device<->device
void gpu_zero(void *p, size_t octets);
bool est_ptr_gpu(const void *p);
bool &residence_active();
// element-wise operation on resident vectors (tsdgpu_vec_op; `op` is a tsdgpu_vec_opcode, complexe: cfloat elements)
enum OpVec { OP_REVERSE = 0, OP_SCALE = 1, OP_DIV_SCALAIRE = 2, OP_ADD = 3, OP_SUB = 4, OP_MUL = 5, OP_NEG = 6, OP_ABS = 7, OP_ABS2 = 8,
             OP_REAL = 9, OP_IMAG = 10, OP_VERS_COMPLEXE = 11, OP_CONJ = 12 };
void gpu_op_vec(int op, bool complexe, void *dst, const void *a, const void *b, float s_re, float s_im, size_t n);
// sum (re, im; accumulated in double), max / min and index of the first max of a resident float / cfloat vector (tsdgpu_vec_reduce)
void gpu_reduction(bool complexe, const void *a, size_t n, double *somme2, float *maxmin2, long long *imax);
}  // namespace detail
struct ResidenceGpu {
  bool avant;
  ResidenceGpu() : avant(detail::residence_active()) { detail::residence_active() = true; }
  ~ResidenceGpu() { detail::residence_active() = avant; }
  ResidenceGpu(const ResidenceGpu &) = delete;
  ResidenceGpu &operator=(const ResidenceGpu &) = delete;
};

// ---- Vecteur<T>: contiguous column vector, int-indexed -------------------------------------
// Semantics kept from TabT<T,1> (tableau.hpp:530-592, pinned by core/tests/test-tab.cc):
// copy construction = deep copy, move = steal, head/tail/segment = views aliasing the parent,
// assignment INTO a view = element copy into the parent, map() wraps foreign memory.
template <typename T> class Vecteur {
  std::shared_ptr<T[]> buf_;
  T *p_ = nullptr;
  entier n_ = 0;
  bool vue_ = false;
  bool gpu_ = false;     // p_ is device memory

 public:
  using Scalar = T;
  Vecteur() = default;
  explicit Vecteur(entier n) { alloc(n); }
  Vecteur(const Vecteur &o)
  {
    alloc(o.n_, o.gpu_);          // a copy lives where its source lives
    copie_depuis(o);
  }
  Vecteur(Vecteur &&o) noexcept : buf_(std::move(o.buf_)), p_(o.p_), n_(o.n_), vue_(o.vue_), gpu_(o.gpu_)
  {
    o.p_ = nullptr;
    o.n_ = 0;
    o.vue_ = false;
    o.gpu_ = false;
  }
  // widening: real -> complex (tableau.hpp:515-521)
  template <typename U, typename = std::enable_if_t<!std::is_same_v<U, T> && std::is_convertible_v<U, T>>>
  Vecteur(const Vecteur<U> &o)
  {
    o.exige_hote("conversion");
    alloc(o.rows(), false);
    for (entier i = 0; i < n_; i++) p_[i] = (T) o.data()[i];
  }
  Vecteur &operator=(const Vecteur &o)
  {
    if (this == &o) return *this;
    if (vue_) {
      if (o.n_ != n_) échec("Vecteur: assignment into a view of {} elements from {} elements", n_, o.n_);
    } else if (o.n_ != n_ || o.gpu_ != gpu_) {
      alloc(o.n_, o.gpu_);
    }
    copie_depuis(o);
    return *this;
  }
  Vecteur &operator=(Vecteur &&o)
  {
    if (this == &o) return *this;
    if (vue_) return *this = static_cast<const Vecteur &>(o);
    buf_ = std::move(o.buf_);
    p_ = o.p_;
    n_ = o.n_;
    vue_ = o.vue_;
    gpu_ = o.gpu_;
    o.p_ = nullptr;
    o.n_ = 0;
    o.vue_ = false;
    o.gpu_ = false;
    return *this;
  }

  static Vecteur map(T *ptr, entier n)
  {
    Vecteur v;
    v.p_ = ptr;
    v.n_ = n;
    v.vue_ = true;
    v.gpu_ = n > 0 && detail::est_ptr_gpu(ptr);
    return v;
  }
  // ---- device residency ----
  static Vecteur sur_gpu(entier n)
  {
    Vecteur v;
    v.alloc(n, true);
    return v;
  }
  bool est_sur_gpu() const { return gpu_; }
  Vecteur vers_gpu() const
  {
    Vecteur v;
    v.alloc(n_, true);
    v.copie_depuis(*this);
    return v;
  }
  Vecteur vers_hote() const
  {
    Vecteur v;
    v.alloc(n_, false);
    v.copie_depuis(*this);
    return v;
  }
  void exige_hote(const char *quoi) const
  {
    if (gpu_) échec("Vecteur: {} on a device-resident vector (bring it back with vers_hote())", quoi);
  }
  static Vecteur zeros(entier n)
  {
    Vecteur v(n);
    v.setZero();
    return v;
  }
  static Vecteur ones(entier n)
  {
    Vecteur v(n, false);
    v.setConstant((T) 1);
    return v;
  }
  static Vecteur valeurs(std::initializer_list<T> l)
  {
    Vecteur v((entier) l.size(), false);
    std::copy(l.begin(), l.end(), v.p_);
    return v;
  }
  template <typename F> static Vecteur int_expr(entier n, F f)
  {
    Vecteur v(n, false);
    for (entier i = 0; i < n; i++) v.p_[i] = (T) f(i);
    return v;
  }

  T *data() { return p_; }
  const T *data() const { return p_; }
  entier rows() const { return n_; }
  entier dim() const { return n_; }
  bool est_vide() const { return n_ == 0; }
  void resize(entier n)
  {
    if (n == n_) return;
    if (vue_) échec("Vecteur::resize on a view");
    alloc(n, gpu_ || detail::residence_active());
  }
  void setZero()
  {
    if (gpu_) detail::gpu_zero(p_, (size_t) n_ * sizeof(T));
    else std::fill(p_, p_ + n_, T());
  }
  void setZero(entier n)
  {
    resize(n);
    setZero();
  }
  void setConstant(T v) { exige_hote("setConstant"); std::fill(p_, p_ + n_, v); }
  Vecteur clone() const { return Vecteur(*this); }

  T &operator()(entier i)
  {
    exige_hote("element access");
    if (i < 0 || i >= n_) échec("Vecteur: index {} out of range (dim = {})", i, n_);
    return p_[i];
  }
  const T &operator()(entier i) const
  {
    exige_hote("element access");
    if (i < 0 || i >= n_) échec("Vecteur: index {} out of range (dim = {})", i, n_);
    return p_[i];
  }

  Vecteur segment(entier i, entier n) const
  {
    if (i < 0 || n < 0 || i + n > n_) échec("Vecteur::segment({}, {}) out of range (dim = {})", i, n, n_);
    Vecteur v;
    v.buf_ = buf_;
    v.p_ = p_ + i;
    v.n_ = n;
    v.vue_ = true;
    v.gpu_ = gpu_;
    return v;
  }
  Vecteur head(entier n) const { return segment(0, n); }
  Vecteur tail(entier n) const { return segment(n_ - n, n); }
  // Resident float / cfloat vectors run their element-wise arithmetic on the device (tsdgpu_vec_op): what call sites put
  // between two operators of the path -- filtfilt's reverse, y = fft(x) * H, x / s -- then stays in HBM.  Other element
  // types, reductions and element access remain host-only and refuse a resident vector loudly.
  static constexpr bool op_gpu_possible() { return std::is_same_v<T, float> || std::is_same_v<T, cfloat>; }
  static float re_de(T s) { if constexpr (est_complexe<T>()) return s.real(); else return (float) s; }
  static float im_de(T s) { if constexpr (est_complexe<T>()) return s.imag(); else { (void) s; return 0.f; } }
  void op_gpu(int op, Vecteur &dst, const Vecteur *b, T s, const char *quoi) const
  {
    if constexpr (op_gpu_possible()) {
      if (b && (!b->gpu_ || b->n_ != n_)) échec("Vecteur: {} between a resident vector and a host vector or one of another size", quoi);
      detail::gpu_op_vec(op, est_complexe<T>(), dst.p_, p_, b ? b->p_ : nullptr, re_de(s), im_de(s), (size_t) n_);
    } else {
      exige_hote(quoi);
    }
  }
  Vecteur reverse() const
  {
    Vecteur v;
    if (gpu_ && op_gpu_possible()) {
      v.alloc(n_, true);
      op_gpu(detail::OP_REVERSE, v, nullptr, T(), "reverse");
      return v;
    }
    exige_hote("reverse");
    v.alloc(n_, false);
    for (entier i = 0; i < n_; i++) v.p_[i] = p_[n_ - 1 - i];
    return v;
  }
  Vecteur<cfloat> as_complex() const
  {
    if constexpr (std::is_same_v<T, float>) {
      if (gpu_) {
        Vecteur<cfloat> v = Vecteur<cfloat>::sur_gpu(n_);
        detail::gpu_op_vec(detail::OP_VERS_COMPLEXE, false, v.data(), p_, nullptr, 0.f, 0.f, (size_t) n_);
        return v;
      }
    }
    exige_hote("as_complex");
    Vecteur<cfloat> v = Vecteur<cfloat>::hote(n_);
    for (entier i = 0; i < n_; i++) v.data()[i] = cfloat(p_[i]);
    return v;
  }
  Vecteur conjugate() const
  {
    if constexpr (std::is_same_v<T, cfloat>) {
      Vecteur v;
      v.alloc(n_, gpu_);
      if (gpu_) detail::gpu_op_vec(detail::OP_CONJ, true, v.p_, p_, nullptr, 0.f, 0.f, (size_t) n_);
      else for (entier i = 0; i < n_; i++) v.p_[i] = std::conj(p_[i]);
      return v;
    } else {
      return clone();
    }
  }
  // a vector of n zeros on the side (host / device) this one lives on
  Vecteur zeros_du_meme_cote(entier n) const
  {
    Vecteur v;
    v.alloc(n, gpu_);
    v.setZero();
    return v;
  }
  template <typename U> Vecteur<U> as() const
  {
    exige_hote("as");
    Vecteur<U> v = Vecteur<U>::hote(n_);
    for (entier i = 0; i < n_; i++) v.data()[i] = (U) p_[i];
    return v;
  }

  // element-wise arithmetic used around the hot path
  Vecteur &operator*=(T s)
  {
    if (gpu_ && op_gpu_possible()) { op_gpu(detail::OP_SCALE, *this, nullptr, s, "arithmetic"); return *this; }
    exige_hote("arithmetic"); for (entier i = 0; i < n_; i++) p_[i] *= s; return *this;
  }
  Vecteur &operator/=(T s)
  {
    if (gpu_ && op_gpu_possible()) { op_gpu(detail::OP_DIV_SCALAIRE, *this, nullptr, s, "arithmetic"); return *this; }
    exige_hote("arithmetic"); for (entier i = 0; i < n_; i++) p_[i] /= s; return *this;
  }
  Vecteur &operator+=(const Vecteur &o)
  {
    if (gpu_ && op_gpu_possible()) { op_gpu(detail::OP_ADD, *this, &o, T(), "arithmetic"); return *this; }
    chk(o); for (entier i = 0; i < n_; i++) p_[i] += o.p_[i]; return *this;
  }
  Vecteur &operator-=(const Vecteur &o)
  {
    if (gpu_ && op_gpu_possible()) { op_gpu(detail::OP_SUB, *this, &o, T(), "arithmetic"); return *this; }
    chk(o); for (entier i = 0; i < n_; i++) p_[i] -= o.p_[i]; return *this;
  }
  Vecteur &operator*=(const Vecteur &o)
  {
    if (gpu_ && op_gpu_possible()) { op_gpu(detail::OP_MUL, *this, &o, T(), "arithmetic"); return *this; }
    chk(o); for (entier i = 0; i < n_; i++) p_[i] *= o.p_[i]; return *this;
  }
  Vecteur operator-() const
  {
    if (gpu_ && op_gpu_possible()) { Vecteur v; v.alloc(n_, true); op_gpu(detail::OP_NEG, v, nullptr, T(), "arithmetic"); return v; }
    exige_hote("arithmetic"); Vecteur v(*this); for (entier i = 0; i < n_; i++) v.p_[i] = -v.p_[i]; return v;
  }
  friend Vecteur operator+(const Vecteur &a, const Vecteur &b) { Vecteur v(a); v += b; return v; }
  friend Vecteur operator-(const Vecteur &a, const Vecteur &b) { Vecteur v(a); v -= b; return v; }
  friend Vecteur operator*(const Vecteur &a, const Vecteur &b) { Vecteur v(a); v *= b; return v; }
  friend Vecteur operator*(const Vecteur &a, T s) { Vecteur v(a); v *= s; return v; }
  friend Vecteur operator*(T s, const Vecteur &a) { Vecteur v(a); v *= s; return v; }
  friend Vecteur operator/(const Vecteur &a, T s) { Vecteur v(a); v /= s; return v; }

  // reductions accumulate in double (tableau.hpp:656-717)
  T somme() const
  {
    if constexpr (op_gpu_possible())
      if (gpu_) {
        double s2[2];
        detail::gpu_reduction(est_complexe<T>(), p_, (size_t) n_, s2, nullptr, nullptr);
        if constexpr (est_complexe<T>()) return T((float) s2[0], (float) s2[1]);
        else return (T) s2[0];
      }
    exige_hote("somme");
    if constexpr (est_complexe<T>()) {
      cdouble s = 0;
      for (entier i = 0; i < n_; i++) s += cdouble(p_[i]);
      return T(s);
    } else {
      double s = 0;
      for (entier i = 0; i < n_; i++) s += p_[i];
      return (T) s;
    }
  }
  T moyenne() const { return n_ ? somme() / (T) n_ : T(); }
  T valeur_max() const
  {
    if constexpr (std::is_same_v<T, float>)
      if (gpu_) { tsd_assertion(n_ > 0); float mm[2]; detail::gpu_reduction(false, p_, (size_t) n_, nullptr, mm, nullptr); return mm[0]; }
    exige_hote("valeur_max"); tsd_assertion(n_ > 0); return *std::max_element(p_, p_ + n_);
  }
  T valeur_min() const
  {
    if constexpr (std::is_same_v<T, float>)
      if (gpu_) { tsd_assertion(n_ > 0); float mm[2]; detail::gpu_reduction(false, p_, (size_t) n_, nullptr, mm, nullptr); return mm[1]; }
    exige_hote("valeur_min"); tsd_assertion(n_ > 0); return *std::min_element(p_, p_ + n_);
  }
  entier index_max() const
  {
    if constexpr (std::is_same_v<T, float>)
      if (gpu_) { long long i = -1; detail::gpu_reduction(false, p_, (size_t) n_, nullptr, nullptr, &i); return (entier) i; }
    exige_hote("index_max"); return n_ ? (entier) (std::max_element(p_, p_ + n_) - p_) : -1;
  }

 private:
  explicit Vecteur(entier n, bool sur_gpu_) { alloc(n, sur_gpu_); }
 public:
  // a host vector whatever the residency guard says (results that host code reads element by element)
  static Vecteur hote(entier n) { return Vecteur(n, false); }
 private:
  void alloc(entier n) { alloc(n, false); }
  void alloc(entier n, bool sur_gpu_)
  {
    if (n < 0) échec("Vecteur: negative size {}", n);
    const size_t octets = (size_t) n * sizeof(T);
    if (n == 0) {
      buf_ = nullptr;
    } else if (sur_gpu_) {
      buf_ = std::shared_ptr<T[]>(static_cast<T *>(detail::gpu_alloc(octets)), [](T *p) { detail::gpu_free(p); });
    } else if (octets >= ((size_t) 1 << 20)) {
      bool verr = false;
      T *p = static_cast<T *>(detail::hote_alloc(octets, &verr));
      buf_ = std::shared_ptr<T[]>(p, [verr](T *q) { detail::hote_free(q, verr); });
    } else {
      buf_ = std::shared_ptr<T[]>(new T[(size_t) n]);
    }
    p_ = buf_.get();
    n_ = n;
    vue_ = false;
    gpu_ = n > 0 && sur_gpu_;
  }
  void copie_depuis(const Vecteur &o)
  {
    if (n_ == 0) return;
    if (gpu_ || o.gpu_) detail::gpu_copie(p_, o.p_, (size_t) n_ * sizeof(T));
    else std::copy(o.p_, o.p_ + n_, p_);
  }
  void chk(const Vecteur &o) const
  {
    exige_hote("arithmetic");
    o.exige_hote("arithmetic");
    if (o.n_ != n_) échec("Vecteur: size mismatch ({} vs {})", n_, o.n_);
  }
};

using Vecf = Vecteur<float>;
using Vecd = Vecteur<double>;
using Veccf = Vecteur<cfloat>;
using Veci = Vecteur<int32_t>;

template <typename T> Vecteur<float> abs(const Vecteur<T> &x)
{
  if constexpr (std::is_same_v<T, float> || std::is_same_v<T, cfloat>)
    if (x.est_sur_gpu()) {
      Vecteur<float> y = Vecteur<float>::sur_gpu(x.rows());
      detail::gpu_op_vec(detail::OP_ABS, est_complexe<T>(), y.data(), x.data(), nullptr, 0.f, 0.f, (size_t) x.rows());
      return y;
    }
  x.exige_hote("abs");
  Vecteur<float> y = Vecteur<float>::hote(x.rows());
  for (entier i = 0; i < x.rows(); i++) y.data()[i] = std::abs(x.data()[i]);
  return y;
}
// |x|^2 element-wise and power -> dB (core/include/tsd/tsd.hpp:414-421, tableau.hpp abs2)
template <typename T> Vecteur<float> abs2(const Vecteur<T> &x)
{
  if constexpr (std::is_same_v<T, float> || std::is_same_v<T, cfloat>)
    if (x.est_sur_gpu()) {
      Vecteur<float> y = Vecteur<float>::sur_gpu(x.rows());
      detail::gpu_op_vec(detail::OP_ABS2, est_complexe<T>(), y.data(), x.data(), nullptr, 0.f, 0.f, (size_t) x.rows());
      return y;
    }
  x.exige_hote("abs2");
  Vecteur<float> y = Vecteur<float>::hote(x.rows());
  for (entier i = 0; i < x.rows(); i++) y(i) = std::norm(x.data()[i]);
  return y;
}
inline Vecf pow2db(const Vecf &x)
{
  Vecf y = Vecf::hote(x.rows());
  for (entier i = 0; i < x.rows(); i++) y(i) = 10 * std::log10(x(i));
  return y;
}
inline Vecf real(const Veccf &x)
{
  if (x.est_sur_gpu()) {
    Vecf y = Vecf::sur_gpu(x.rows());
    detail::gpu_op_vec(detail::OP_REAL, true, y.data(), x.data(), nullptr, 0.f, 0.f, (size_t) x.rows());
    return y;
  }
  return Vecf::int_expr(x.rows(), [&](entier i) { return x.data()[i].real(); });
}
inline Vecf imag(const Veccf &x)
{
  if (x.est_sur_gpu()) {
    Vecf y = Vecf::sur_gpu(x.rows());
    detail::gpu_op_vec(detail::OP_IMAG, true, y.data(), x.data(), nullptr, 0.f, 0.f, (size_t) x.rows());
    return y;
  }
  return Vecf::int_expr(x.rows(), [&](entier i) { return x.data()[i].imag(); });
}
template <typename T> Vecteur<T> vconcat(const Vecteur<T> &a, const Vecteur<T> &b)
{
  Vecteur<T> v(a.rows() + b.rows());      // follows the residency guard; the two halves are views
  if (a.rows() > 0) v.head(a.rows()) = a;
  if (b.rows() > 0) v.tail(b.rows()) = b;
  return v;
}

// linspace (tsd.hpp:916-931): step in double, samples rounded to float
inline Vecf linspace(float a, float b, entier n)
{
  Vecf x = Vecf::hote(n);
  if (n > 0) x(0) = a;
  if (n > 1) {
    const double step = ((double) b - a) / (n - 1);
    for (entier i = 1; i < n; i++) x(i) = (float) (a + step * i);
  }
  return x;
}
// sigimp (core/src/tsd.cc): unit impulse at position p
inline Vecf sigimp(entier n, entier p = 0)
{
  Vecf x = Vecf::zeros(n);
  if (n > 0) x(p) = 1;
  return x;
}
// sousech / surech (tsd.hpp:367-401): pure index permutations
template <typename T> Vecteur<T> sousech(const Vecteur<T> &x, entier R)
{
  const entier n = x.dim();
  Vecteur<T> y = Vecteur<T>::hote(n / R);
  for (entier i = 0; i < n / R; i++) y(i) = x(i * R);
  return y;
}
template <typename T> Vecteur<T> surech(const Vecteur<T> &x, entier R)
{
  const entier n = x.dim();
  Vecteur<T> y = Vecteur<T>::hote(n * R);
  y.setZero();
  for (entier i = 0; i < n; i++) y(i * R) = x(i);
  return y;
}
// sigexp / sigcos / sigsin (core/src/tsd.cc:217-245): the phasor advanced by repeated multiplication in double, its
// modulus renormalised every 1000 samples; cos / sin = its real / imaginary part
Vecteur<cfloat> sigexp(float f, entier n);
Vecteur<float> sigcos(float f, entier n);
Vecteur<float> sigsin(float f, entier n);
// randn / randcn on a default-seeded engine (core/src/tsd.cc:173,410-483)
std::default_random_engine &generateur_aleatoire();
Vecf randn(entier n);
Veccf randcn(entier n);
// prochaine_puissance_de_2 (core/src/tsd.cc:287-291), float-log based like the reference
entier prochaine_puissance_de_2(entier i);

// pad_zeros (core/include/tsd/tsd.hpp:514-533): both vectors zero-padded at the end to the
// larger of the two dimensions (rounded up to a power of two when p2)
template <typename T> std::tuple<Vecteur<T>, Vecteur<T>> pad_zeros(const Vecteur<T> &x, const Vecteur<T> &y, bouléen p2 = false)
{
  const entier n1 = x.dim(), n2 = y.dim();
  entier n3 = std::max(n1, n2);
  if (p2) n3 = prochaine_puissance_de_2(n3);
  Vecteur<T> a = Vecteur<T>::zeros(n3), b = Vecteur<T>::zeros(n3);
  a.head(n1) = x;
  b.head(n2) = y;
  return {a, b};
}

// ---- operator interfaces (tsd.hpp:544-579,626-668) -------------------------------------------
template <typename C> struct Configurable {
  virtual ~Configurable() {}
  void configure(const C &c)
  {
    if (callback_modif) callback_modif(c);
    configure_impl(c);
    config = c;
  }
  virtual void configure_impl(const C &c) = 0;
  const C &lis_config() const { return config; }
  fonction<void(const C &)> callback_modif;

 protected:
  C config;
};

template <typename Te, typename Ts = Te> struct FiltreGen {
  virtual ~FiltreGen() {}
  virtual void step(const Vecteur<Te> &x, Vecteur<Ts> &y) = 0;
  Vecteur<Ts> step(const Vecteur<Te> &x)
  {
    Vecteur<Ts> y;
    step(x, y);
    return y;
  }
  Ts step(Te x)
  {
    Vecteur<Te> vx(1);
    Vecteur<Ts> vy(1);
    vx(0) = x;
    step(vx, vy);
    return vy(0);
  }
};

template <typename Te, typename Ts = Te, typename Tc = Void> struct Filtre : Configurable<Tc>, FiltreGen<Te, Ts> {
  virtual ~Filtre() {}
};

// data sinks (core/include/tsd/tsd.hpp:584-599) and the re-blocking buffer tampon_création
// (core/src/tsd.cc:307-381): calls `callback` with consecutive blocks of exactly N samples,
// whatever the sizes of the vectors handed to step()
template <typename Te> struct SinkGen {
  virtual ~SinkGen() {}
  virtual void step(const Vecteur<Te> &x) = 0;
};
template <typename Te, typename Tc = Void> struct Sink : Configurable<Tc>, SinkGen<Te> {
  virtual ~Sink() {}
};
template <typename T> sptr<Sink<T, entier>> tampon_création(entier N, fonction<void(const Vecteur<T> &)> callback);


namespace filtrage {
template <typename T> sptr<Filtre<T, T, float>> filtre_reechan(float ratio);
}

// rééchan (tsd.hpp:700-705): one-shot resampling through filtre_reechan
template <typename T> Vecteur<T> rééchan(const Vecteur<T> &x, float ratio)
{
  auto f = filtrage::filtre_reechan<T>(ratio);
  return f->FiltreGen<T, T>::step(x);
}

}  // namespace tsd
