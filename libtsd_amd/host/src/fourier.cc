// fourier.cc -- FFTPlan on the MI355X C ABI, the fftplan_defaut hook and the real-FFT plan.
#include "tsd/fourier.hpp"
#include "../../../include/tsdgpu.h"

namespace tsd::fourier {

namespace {

// TFRPlanDefaut's contract (fourier.cc:360-467): (re)configures itself when the input size
// changes, unitary scaling in both directions, `normalize` accepted and ignored.
struct FFTPlanGpu : FFTPlan {
  tsdgpu_fft *h = nullptr;
  entier n = -1;
  bouléen avant_defaut = true;
  ~FFTPlanGpu() override { tsdgpu_fft_destroy(h); }
  void configure(entier n_, bouléen avant, bouléen) override
  {
    avant_defaut = avant;
    if (n_ == n) return;
    tsdgpu_fft_destroy(h);
    h = nullptr;
    n = n_;
    if (n < 1) return;
    if (tsdgpu_fft_create(&h, n, 1)) échec("FFTPlan::configure({}): {}", n, tsdgpu_last_error());
  }
  void step(const Veccf &x, Veccf &y, bouléen avant) override
  {
    if (x.rows() <= 0) échec("FFTPlan::step: empty input");          // assertion(x.rows() > 0), fourier.cc:414
    if (x.rows() != n) configure(x.rows(), avant_defaut, true);
    if (x.data() != y.data()) y.resize(n);
    if (tsdgpu_fft_step(h, x.data(), y.data(), 1, avant ? 1 : 0, nullptr)) échec("FFTPlan::step: {}", tsdgpu_last_error());
  }
};

// RTFRPlan (fourier.cc:280-355): even n -> n/2-point complex FFT of the packed pairs, then the
// untangling pass with the 0.5/sqrt(2) factors, then forced conjugate symmetry.
struct RTFRPlanGpu : FiltreGen<float, cfloat> {
  entier n = -1;
  sptr<FFTPlan> cplan;
  Veccf rotations;
  explicit RTFRPlanGpu(entier n_) { configure(n_); }
  void configure(entier n_)
  {
    n = n_;
    if (n <= 0) return;
    if ((n & 1) == 0) {
      cplan = tfrplan_création(n / 2);
      // tfr_rotation_rapide (fourier.cc:32-46): double recurrence rounded to float
      rotations.resize(n);
      cdouble r = 1, w0 = std::polar<double>(1.0, (-2 * π) / n);
      for (entier i = 0; i < n; i++) {
        rotations(i) = cfloat(r);
        r *= w0;
      }
    } else {
      cplan = tfrplan_création(n);
    }
  }
  void step(const Vecf &x, Veccf &y) override
  {
    if (x.rows() != n) configure(x.rows());
    if (n <= 0) {
      y.resize(0);
      return;
    }
    if ((n & 1) == 0) {
      y.resize(n);
      Veccf x2(n / 2);
      for (entier i = 0; i < n / 2; i++) x2(i) = cfloat(x(2 * i), x(2 * i + 1));
      const Veccf Xt = n / 2 > 0 ? cplan->step(x2) : Veccf();
      const cfloat j2(0, (float) (0.5 / std::sqrt(2.0))), r2((float) (0.5 / std::sqrt(2.0)), 0);
      for (entier i = 0; i <= n / 2; i++) {
        const cfloat X1 = (i == n / 2) ? Xt(0) : Xt(i);
        const cfloat X2 = (i > 0) ? Xt(n / 2 - i) : Xt(0);
        y(i) = r2 * (X1 + std::conj(X2)) - j2 * (X1 - std::conj(X2)) * rotations(i);
      }
      csym_forçage(y);
    } else {
      const Veccf y1 = x.as_complex();
      cplan->step(y1, y);
    }
  }
};

}  // namespace

fonction<sptr<FFTPlan>()> fftplan_defaut = []() -> sptr<FFTPlan> { return std::make_shared<FFTPlanGpu>(); };

sptr<FFTPlan> tfrplan_création(entier n, bouléen avant, bouléen normalize)
{
  auto res = fftplan_defaut();
  if (n >= 0) res->configure(n, avant, normalize);
  return res;
}

sptr<FiltreGen<float, cfloat>> rtfrplan_création(entier n) { return std::make_shared<RTFRPlanGpu>(n); }

void csym_forçage_impl(Veccf &X)
{
  const entier n = X.rows();
  if (n == 0) return;
  X(0).imag(0);
  if ((n & 1) == 0)
    X(n / 2).imag(0);
  else if (n > 1)
    X(n / 2 + 1) = std::conj(X(n / 2));
  const entier m = n / 2 - 1;
  for (entier i = 0; i < m; i++) X(n - m + i) = std::conj(X(1 + (m - 1 - i)));
}

}  // namespace tsd::fourier
