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

// ---- correlations (fourier.cc:489-597) --------------------------------------------------------
static Veccf correlation_freq(const Veccf &X0, const Veccf &X1)
{
  const entier n = X0.rows();
  if (n != X1.rows()) échec("correlation_freq: dimensions {} != {}", n, X1.rows());
  Veccf Y(n);
  Y(0) = X0(0) * std::conj(X1(0));
  // Y.tail(n-1) = X0.tail(n-1).reverse() * X1.tail(n-1).reverse().conjugate()
  for (entier i = 1; i < n; i++) Y(i) = X0(n - i) * std::conj(X1(n - i));
  Y *= cfloat(std::sqrt((float) n), 0.f);
  return Y;
}
static Veccf correlateur_bloc(const Veccf &x0, const Veccf &x1)
{
  const Veccf &x1p = x1.rows() == 0 ? x0 : x1;
  if (x0.rows() != x1p.rows()) échec("the two input vectors should have the dimension {} != {}.", x0.rows(), x1p.rows());
  auto plan = fftplan_defaut();
  const Veccf X0 = plan->step(x0, true), X1 = plan->step(x1p, true);
  return plan->step(correlation_freq(X0, X1), false);
}
std::tuple<Vecf, Veccf> ccorr(const Veccf &x0, const Veccf &x1)
{
  const entier m = x0.rows();
  Veccf r = correlateur_bloc(x0, x1);
  r /= cfloat((float) m, 0.f);
  return {linspace(0, (float) (m - 1), m), r};
}
std::tuple<Vecf, Veccf> xcorrb(const Veccf &x, const Veccf &y, entier m)
{
  const entier n = x.rows();
  if (m < 0) m = n;
  const Veccf &yp = y.rows() == 0 ? x : y;
  Veccf x2 = Veccf::zeros(m + n + m), y2 = Veccf::zeros(m + n + m);
  x2.segment(m, n) = x;
  y2.segment(m, n) = yp;
  const Veccf r = correlateur_bloc(x2, y2);
  Veccf res(2 * m - 1);
  for (entier i = 0; i < m; i++) res(m - 1 + i) = r(i) / (float) n;              // res.tail(m) = r.head(m) / n
  for (entier i = 0; i < m - 1; i++) res(i) = r(r.rows() - (m - 1) + i) / (float) n;   // res.head(m-1) = r.tail(m-1) / n
  return {linspace((float) -(m - 1), (float) (m - 1), 2 * m - 1), res};
}
std::tuple<Vecf, Veccf> xcorr(const Veccf &x, const Veccf &y, entier m)
{
  const entier n = x.rows();
  if (m < 0) m = n;
  auto [lags, zb] = xcorrb(x, y, m);
  if (m > 1) {
    const Vecf a = linspace((float) (n - (m - 1)), (float) (n - 1), m - 1), b = linspace((float) (n - 1), (float) (n - (m - 1)), m - 1);
    for (entier i = 0; i < m - 1; i++) {
      zb(i) /= cfloat(a(i) / n, 0.f);
      zb(zb.rows() - (m - 1) + i) /= cfloat(b(i) / n, 0.f);
    }
  }
  return {lags, zb};
}

// ---- rééchan_freq (fourier.cc:1391-1419) -------------------------------------------------------
Vecf rééchan_freq(const Vecf &x, float lom)
{
  if (lom == 1) return x;
  const entier n = x.rows(), n2 = (entier) std::round(n * lom);
  const Veccf X = fft(x);
  Veccf X2 = Veccf::zeros(n2);
  const entier h = (lom > 1 ? n : n2) / 2;
  X2.head(h) = X.head(h);
  X2.tail(h) = X.tail(h);
  Veccf xi = ifft(X2);
  xi *= cfloat(std::sqrt(lom), 0.f);
  return real(xi);
}

}  // namespace tsd::fourier
