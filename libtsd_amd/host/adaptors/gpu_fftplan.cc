// gpu_fftplan.cc -- libtsd's FFT plug point on the MI355X C ABI.
// libtsd asks the global factory tsd::fourier::fftplan_defaut for a plan whenever it needs one
// (core/include/tsd/fourier.hpp:19-35; core/src/fourier/fourier.cc:469-481: tfrplan_création, hence
// every fft() / ifft(), TFRCorrelateurBloc :508 and Spectrum :1225).  This TU defines a FFTPlan on
// tsdgpu_fft and the function that installs it; it defines no libtsd symbol, so it links beside an
// unmodified fourier.cc.
#include "gpu_commun.hpp"
#include <vector>
#include <mutex>
#include <map>
#include "tsd_amd/extensions.hpp"
#include "reserve_plans.hpp"

namespace tsd_amd {

using namespace tsd;
using tsd::fourier::FFTPlan;

// TFRPlanDefaut's contract (fourier.cc:360-467): (re)configures itself when the input size changes;
// unitary scaling 1/sqrt(n) in BOTH directions whatever `normalize` says (the reference never stores
// it, fourier.cc:362,372-376); natural order.
// libtsd's fft() / ifft() make a plan per CALL (fourier.hpp:163-205).  A device plan costs 30 us (n = 4096) to 1.1 ms (2^20) to
// build -- its twiddle tables -- so finished plans go to a small per-size reserve instead of being destroyed, and the next
// plan of that size takes one from there (a handle serves one caller at a time: it leaves the reserve while in use).
// The reserve is keyed by (device, n): a plan's tables live on the device it was created on (reserve_plans.hpp).
static void detruit_fft(tsdgpu_fft *h) { tsdgpu_fft_destroy(h); }
static ReserveParCle<tsdgpu_fft> &reserve_plans()
{
  static ReserveParCle<tsdgpu_fft> *r = new ReserveParCle<tsdgpu_fft>(detruit_fft);      // never destroyed: plans may be returned during static destruction
  return *r;
}

struct FFTPlanGpu : FFTPlan {
  tsdgpu_fft *h = nullptr;
  entier n = -1;
  int dev = -1;                      // the device h lives on
  bouléen avant_defaut = true;
  virtual ~FFTPlanGpu() { reserve_plans().rend(dev, n, h); }     // FFTPlan has no virtual destructor: make_shared's deleter knows the type
  void configure(entier n_, bouléen avant, bouléen)
  {
    avant_defaut = avant;
    const int ici = tsdgpu_current_device();
    if (n_ == n && (h == nullptr || ici == dev)) return;
    reserve_plans().rend(dev, n, h);
    h = nullptr;
    n = n_;
    dev = ici;
    if (n < 1) return;
    h = reserve_plans().prend(dev, n);
    if (!h && tsdgpu_fft_create(&h, n, 1)) gpu_fail("FFTPlan::configure");
  }
  void step(const Veccf &x, Veccf &y, bouléen avant)
  {
    if (x.rows() <= 0) échec("FFTPlan::step: empty input");          // assertion(x.rows() > 0), fourier.cc:414
    if (x.rows() != n || tsdgpu_current_device() != dev) configure(x.rows(), avant_defaut, true);
    if (x.data() != y.data()) dimensionne(y, n);
    if (tsdgpu_fft_step(h, x.data(), y.data(), 1, avant ? 1 : 0, nullptr)) gpu_fail("FFTPlan::step");
  }
};

sptr<FFTPlan> fftplan_gpu() { return std::make_shared<FFTPlanGpu>(); }
void installe_fftplan_gpu() { tsd::fourier::fftplan_defaut = fftplan_gpu; }

// RTFRPlan (fourier.cc:280-355) on tsdgpu_rfft: packed n/2-point complex FFT, untangling with the
// 0.5/sqrt(2) factors and the forced conjugate symmetry all run on the device.
// (the same reserve for the real-input plans: rfft() / fft(Vecf) make one per call too)
static void detruit_rfft(tsdgpu_rfft *h) { tsdgpu_rfft_destroy(h); }
static ReserveParCle<tsdgpu_rfft> &reserve_plans_reels()
{
  static ReserveParCle<tsdgpu_rfft> *r = new ReserveParCle<tsdgpu_rfft>(detruit_rfft);
  return *r;
}

struct RTFRPlanGpu : FiltreGen<float, cfloat> {
  entier n = -1;
  int dev = -1;
  tsdgpu_rfft *h = nullptr;
  explicit RTFRPlanGpu(entier n_) { configure(n_); }
  ~RTFRPlanGpu() { reserve_plans_reels().rend(dev, n, h); }
  void configure(entier n_)
  {
    const int ici = tsdgpu_current_device();
    if (n_ == n && (h == nullptr || ici == dev)) return;
    reserve_plans_reels().rend(dev, n, h);
    h = nullptr;
    n = n_;
    dev = ici;
    if (n <= 0) return;
    h = reserve_plans_reels().prend(dev, n);
    if (!h && tsdgpu_rfft_create(&h, n)) gpu_fail("rtfrplan_création");
  }
  void step(const Vecf &x, Veccf &y)
  {
    if (x.rows() != n || tsdgpu_current_device() != dev) configure(x.rows());
    if (n <= 0) {
      dimensionne(y, 0);
      return;
    }
    dimensionne(y, n);
    if (tsdgpu_rfft_step(h, x.data(), y.data(), 1, nullptr)) gpu_fail("rfft");
  }
};
sptr<FiltreGen<float, cfloat>> rtfrplan_gpu(entier n) { return std::make_shared<RTFRPlanGpu>(n); }

int &fragments_forces()
{
  static int n = -1;
  return n;
}
void fixe_fragments(int n) { fragments_forces() = n; }

// ---- device memory helpers for resident vectors ------------------------------------------------------
void *alloue_gpu(size_t octets)
{
  void *p = nullptr;
  if (tsdgpu_malloc(&p, octets)) gpu_fail("alloue_gpu");
  return p;
}
void libere_gpu(void *p) { tsdgpu_free(p); }
void copie_vers_gpu(void *dst, const void *src, size_t octets)
{
  if (tsdgpu_memcpy(dst, src, octets, nullptr)) gpu_fail("copie_vers_gpu");
}
void copie_vers_hote(void *dst, const void *src, size_t octets)
{
  if (tsdgpu_memcpy(dst, src, octets, nullptr)) gpu_fail("copie_vers_hote");
}
void synchronise_gpu()
{
  if (tsdgpu_synchronize(nullptr)) gpu_fail("synchronise_gpu");
}

}  // namespace tsd_amd
