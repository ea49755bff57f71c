// fourier.cc -- mirror runtime: the free functions of libtsd's fourier.cc around the plan hook
// (tfrplan_création, rtfrplan_création, csym, correlations, délais, psd, rt_spectrum).  Mirror only:
// against libtsd itself these are libtsd's own and reach the GPU through fftplan_defaut.
#include "tsd/fourier.hpp"
#include "tsd/filtrage.hpp"
#include <limits>
#include <vector>
#include "tsd_amd/extensions.hpp"
#include "tsdgpu.h"

namespace tsd::fourier {

// the plug point (fourier.hpp:35): the mirror starts with the MI355X plan installed
fonction<sptr<FFTPlan>()> fftplan_defaut = tsd_amd::fftplan_gpu;

sptr<FFTPlan> tfrplan_création(entier n, bouléen avant, bouléen normalize)
{
  auto res = fftplan_defaut();
  if (n >= 0) res->configure(n, avant, normalize);
  return res;
}

sptr<FiltreGen<float, cfloat>> rtfrplan_création(entier n) { return tsd_amd::rtfrplan_gpu(n); }

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
// The element-wise steps are written with the array operators, which run where the vectors live; the entry points bring
// host vectors to the device once and the result back once, so nothing but the transforms' inputs / outputs crosses PCIe
// and no step runs on the host (a resident caller gets a resident result).
static Veccf correlation_freq(const Veccf &X0, const Veccf &X1)
{
  const entier n = X0.rows();
  if (n != X1.rows()) échec("correlation_freq: dimensions {} != {}", n, X1.rows());
  // Y(0) = X0(0) conj(X1(0));  Y.tail(n-1) = X0.tail(n-1).reverse() * X1.tail(n-1).reverse().conjugate()   (fourier.cc:489-505)
  const Veccf P = X0 * X1.conjugate();
  Veccf Y = X0.zeros_du_meme_cote(n);
  if (n > 0) Y.head(1) = P.head(1);
  if (n > 1) Y.tail(n - 1) = P.tail(n - 1).reverse();
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
  const bool hote = !x0.est_sur_gpu();
  if (m == 0) return {Vecf(), Veccf()};
  Veccf r;
  {
    ResidenceGpu garde;        // the transforms' outputs stay on the device
    r = hote ? correlateur_bloc(x0.vers_gpu(), x1.rows() ? x1.vers_gpu() : Veccf()) : correlateur_bloc(x0, x1);
    r /= cfloat((float) m, 0.f);
  }
  return {linspace(0, (float) (m - 1), m), hote ? r.vers_hote() : r};
}
// xcorrb / xcorr: zero-padding, the two forward transforms (one batched call), correlation_freq, the inverse
// transform and the extraction / scaling of the lags all run on the device (tsdgpu_xcorr): the vectors go
// up once, the 2m - 1 lags come back once
static std::tuple<Vecf, Veccf> xcorr_gpu(const Veccf &x, const Veccf &y, entier m, bool non_biaisé)
{
  const entier n = x.rows();
  if (m < 0) m = n;
  if (y.rows() != 0 && y.rows() != n) échec("the two input vectors should have the dimension {} != {}.", n, y.rows());
  Veccf res(2 * m - 1);
  if (tsdgpu_xcorr(x.data(), y.rows() == 0 ? nullptr : y.data(), n, m, non_biaisé ? 1 : 0, res.data(), nullptr))
    échec("xcorr: {}", tsdgpu_last_error());
  return {linspace((float) -(m - 1), (float) (m - 1), 2 * m - 1), res};
}
std::tuple<Vecf, Veccf> xcorrb(const Veccf &x, const Veccf &y, entier m) { return xcorr_gpu(x, y, m, false); }
std::tuple<Vecf, Veccf> xcorr(const Veccf &x, const Veccf &y, entier m) { return xcorr_gpu(x, y, m, true); }

// ---- rééchan_freq (fourier.cc:1391-1419) -------------------------------------------------------
Vecf rééchan_freq(const Vecf &x, float lom)
{
  if (lom == 1) return x;
  const entier n = x.rows(), n2 = (entier) std::round(n * lom);
  const bool hote = !x.est_sur_gpu();
  Vecf y;
  {
    ResidenceGpu garde;        // the transforms' outputs stay on the device
    const Veccf X = fft(hote ? x.vers_gpu() : x);
    Veccf X2 = X.zeros_du_meme_cote(n2);
    const entier h = (lom > 1 ? n : n2) / 2;
    X2.head(h) = X.head(h);
    X2.tail(h) = X.tail(h);
    Veccf xi = ifft(X2);
    xi *= cfloat(std::sqrt(lom), 0.f);
    y = real(xi);
  }
  return hote ? y.vers_hote() : y;
}

// ---- czt (fourier.cc:1347-1389) ----------------------------------------------------------------------
// The sequences are built on the host exactly as the reference builds them (float pow of complex numbers, its index
// arithmetic included -- a transliteration of a 1-based script that the reference left as it is), the three transforms run
// on the device plan through fft() / ifft(), i.e. 2m - 1 points: an odd size, the one-kernel Bluestein.
Veccf czt(const Veccf &x, entier m, cfloat W, cfloat z0)
{
  const entier n = x.rows(), nm = std::max(n, m);
  if (n < 1 || m < 1) échec("czt: n = {}, m = {}", n, m);
  if (m + n - 1 != 2 * m - 1) échec("czt: {} samples, {} points: the reference multiplies transforms of {} and {} points -- only n == m is served", n, m, m + n - 1, 2 * m - 1);
  const Veccf xh = x.est_sur_gpu() ? x.vers_hote() : x;
  Veccf h(2 * nm - 1);
  for (entier i = 0; i < nm; i++) h(i) = std::pow(W, -0.5f * i * i);                                 // :1356-1357
  for (entier i = nm; i < 2 * nm - 1; i++) h(i) = h(nm - 1 - (i - nm));                              // :1358-1359
  Veccf g(n);
  for (entier i = 0; i < n; i++) g(i) = xh(i) * std::pow(z0, (float) -i) / h(nm + i - 1);           // :1364-1368
  Veccf hc(m + n - 1);                                                                               // :1374-1376
  hc.head(m) = h.segment(nm - 1, m);
  if (n > 1) hc.tail(n - 1) = h.segment(nm - n, n - 1);
  Veccf gc = Veccf::zeros(m + m - 1);                                                                // :1378-1379
  gc.head(n) = g;
  const Veccf hcg = ifft(fft(hc) * fft(gc));                                                         // :1384
  Veccf y(m);                                                                                        // :1389 (element-wise division)
  for (entier i = 0; i < m; i++) y(i) = hcg(i) / h(nm - 1 + i);
  return y;
}

// ---- délais (fourier.cc:607-698) ----------------------------------------------------------------
// Fractional delay = a linear phase on the spectrum of the vector zero-padded to twice its length (a quarter
// of the padded length on either side): bin k of the padded transform, at signed frequency f_k = k/n for
// k < n/2 and (k - n)/n above, is rotated by exp(-2 pi j f_k τ).  One forward and one inverse GPU transform.
// A real vector goes through the same complex path and keeps the real part (the reference runs a packed
// half-size real transform there and rounds its Nyquist bin differently: below 1e-6 of the peak on the
// band-limited signals a fractional delay is meant for).
static Veccf délais_frac(const Veccf &x, float τ)
{
  const entier m = x.rows(), n = 2 * m, marge = m / 2;
  const bool hote = !x.est_sur_gpu();
  // the linear phase, tabulated on the host (n phasors) and multiplied in where the spectrum lives
  Veccf phase = Veccf::hote(n);
  for (entier k = 0; k < n; k++) {
    const entier ks = k < n / 2 ? k : k - n;                       // signed bin
    phase.data()[k] = std::polar(1.0f, (float) (-2.0 * π * (double) ks * (double) τ / (double) n));
  }
  Veccf y;
  {
    ResidenceGpu garde;        // the transforms' outputs stay on the device
    const Veccf xg = hote ? x.vers_gpu() : x;
    Veccf étendu = xg.zeros_du_meme_cote(n);
    étendu.segment(marge, m) = xg;
    Veccf X = fft(étendu);
    X *= phase.vers_gpu();
    y = ifft(X).segment(marge, m).clone();
  }
  return hote ? y.vers_hote() : y;
}
static Vecf délais_frac(const Vecf &x, float τ) { return real(délais_frac(x.as_complex(), τ)); }
// whole-sample delay: y(i) = x(i - d), zeros shifted in
template <typename T> static Vecteur<T> délais_entier(const Vecteur<T> &x, entier d)
{
  const entier n = x.rows();
  Vecteur<T> y = x.zeros_du_meme_cote(n);          // (two views and a copy: runs where x lives)
  if (d >= 0 && d < n) y.segment(d, n - d) = x.segment(0, n - d);
  else if (d < 0 && -d < n) y.segment(0, n + d) = x.segment(-d, n + d);
  return y;
}
template <typename T> Vecteur<T> délais(const Vecteur<T> &x, float τ)
{
  if (std::floor(τ) != τ) return délais_frac(x, τ);
  return délais_entier<T>(x, (entier) τ);
}
template Vecf délais<float>(const Vecf &, float);
template Veccf délais<cfloat>(const Veccf &, float);

// ---- estimation_délais, aligne_entier (estimation-delais.cc:9-170) ----------------------------
// estimation_délais: the cross-correlation, the two energies, the arg max of the normalised magnitude and
// its neighbours are computed on the device (tsdgpu_delay_estimate); two floats come back
std::tuple<float, float> estimation_délais(const Veccf &x, const Veccf &y)
{
  auto [xp, yp] = pad_zeros(x, y);
  float retard = 0, score = 0;
  if (tsdgpu_delay_estimate(xp.data(), yp.data(), xp.rows(), &retard, &score, nullptr)) échec("estimation_délais: {}", tsdgpu_last_error());
  return {retard, score};
}
// aligne_entier (estimation-delais.cc:120-170): the two vectors cut to their common support once the
// integer part of the estimated delay is removed -- y late by d > 0 loses its first d samples, x late
// loses its first |d|; the longer remainder is then cut at its end
template <typename T> std::tuple<Vecteur<T>, Vecteur<T>, entier, float> aligne_entier(const Vecteur<T> &x, const Vecteur<T> &y)
{
  auto [xp, yp] = pad_zeros(x.as_complex(), y.as_complex(), true);
  auto [retard, score] = estimation_délais(xp, yp);
  const entier d = (entier) std::round(retard);
  const entier saut_x = d < 0 ? -d : 0, saut_y = d > 0 ? d : 0;
  const entier commun = std::max(0, std::min(x.rows() - saut_x, y.rows() - saut_y));
  return {x.segment(saut_x, commun).clone(), y.segment(saut_y, commun).clone(), d, score};
}
template std::tuple<Vecf, Vecf, entier, float> aligne_entier<float>(const Vecf &, const Vecf &);
template std::tuple<Veccf, Veccf, entier, float> aligne_entier<cfloat>(const Veccf &, const Veccf &);

// ---- psd, psd_welch and their frequency axes (fourier.hpp:739-757; freqestim.cc:7-93) ----------
Vecf tfd_freqs(entier n, bouléen avec_shift)
{
  const bool pair = (n & 1) == 0;
  if (avec_shift) return pair ? linspace(-0.5f, 0.5f - 1.0f / n, n) : linspace(-0.5f + 1.0f / n, 0.5f, n);
  Vecf f(n);
  if (pair) {
    f.head(n / 2) = linspace(0, 0.5f - 1.0f / n, n / 2);
    f.tail(n / 2) = linspace(-0.5f, -1.0f / n, n / 2);
  } else {
    f.head(n / 2 + 1) = linspace(0, 0.5f - 0.5f / n, n / 2 + 1);
    f.tail(n / 2) = linspace(-0.5f, -1.0f / n, n / 2);
  }
  return f;
}
Vecf psd_freqs(entier n, bouléen complexe)
{
  const bool pair = (n & 1) == 0;
  if (complexe) {
    double t0 = -0.5, t1 = 0.5;
    if (pair)
      t1 -= 1.0 / n;
    else
      t0 += 1.0 / n;
    return linspace((float) t0, (float) t1, n);
  }
  double t1 = 0.5;
  if (!pair) t1 -= 1.0 / n;
  return linspace(0.0f, (float) t1, n / 2);
}
template <typename T> std::tuple<Vecf, Vecf> psd(const Vecteur<T> &x)
{
  const Vecf fen = tsd::filtrage::fenêtre("hn", x.rows(), false);
  Vecteur<T> xf = x.clone();
  for (entier i = 0; i < x.rows(); i++) xf(i) *= fen(i);
  if constexpr (est_complexe<T>()) {
    return {psd_freqs(x.rows(), true), fftshift(pow2db(abs2(fft(xf))))};
  } else {
    const Vecf Y = pow2db(abs2(rfft(xf)));
    return {psd_freqs(x.rows(), false), Y.head(Y.rows() / 2).clone()};
  }
}
template std::tuple<Vecf, Vecf> psd<float>(const Vecf &);
template std::tuple<Vecf, Vecf> psd<cfloat>(const Veccf &);

std::tuple<Vecf, Vecf> psd_welch(const Veccf &x, entier N, cstring fen)
{
  if (N < 1) échec("psd_welch: N = {}", N);
  const Vecf f = tsd::filtrage::fenêtre(fen, N, false);
  // segments i = 0, N/2, N, ... while i + N < x.rows() (freqestim.cc:13): framing, one batched FFT
  // and the sum of the periodograms run on the device (tsdgpu_welch); N floats come back
  Vecf S(N);
  if (tsdgpu_welch(x.data(), x.rows(), N, f.data(), S.data(), nullptr, nullptr)) échec("psd_welch: {}", tsdgpu_last_error());
  return {psd_freqs(N), pow2db(S)};
}

// ---- rt_spectrum (fourier.cc:1148-1342) ---------------------------------------------------------------
// Spectrum::step on the device (tsdgpu_spectrum_*): window x sub-block, transform, |X|^2 in fftshift order, the sums over
// nsubs x nmeans sub-blocks (sweep: side by side through the masks, divided by the contributions per bin) and the dB all
// run on the GPU -- x goes up (or is resident already), Ns floats come back with every nmeans-th block.
namespace {
struct SpectreGpu : Filtre<cfloat, float, SpectrumConfig> {
  entier Nf = 0, Ns = 0;
  tsdgpu_spectrum *h = nullptr;
  ~SpectreGpu() override { tsdgpu_spectrum_destroy(h); }
  void configure_impl(const SpectrumConfig &c) override
  {
    if (c.nsubs < 1 || c.BS < c.nsubs || c.nmeans < 1) échec("rt_spectrum: BS = {}, nsubs = {}, nmeans = {}", c.BS, c.nsubs, c.nmeans);
    Nf = c.BS / c.nsubs;
    Ns = c.Ns();
    // masks of the sweep (fourier.cc:1187-1194)
    Vecf masque = Vecf::ones(Nf);
    if (c.sweep.masque_hf > 0) {
      masque.head(c.sweep.masque_hf).setZero();
      masque.tail(c.sweep.masque_hf).setZero();
    }
    if (c.sweep.masque_bf > 0) masque.segment(Nf / 2 - c.sweep.masque_bf, 2 * c.sweep.masque_bf).setZero();
    Vecf f = tsd::filtrage::fenêtre(c.fenetre, Nf, false);
    // window energy normalised to Nf: the total energy of an uncorrelated signal is preserved (:1209-1212)
    double e = 0;
    for (entier k = 0; k < Nf; k++) e += (double) f(k) * f(k);
    f *= (float) std::sqrt(Nf / e);
    tsdgpu_spectrum_destroy(h);
    h = nullptr;
    if (tsdgpu_spectrum_create(&h, c.BS, c.nsubs, c.nmeans, f.data(), c.sweep.active ? 1 : 0, c.sweep.step, masque.data()))
      échec("rt_spectrum: {}", tsdgpu_last_error());
  }
  void step(const Veccf &x, Vecf &y) override
  {
    const SpectrumConfig &c = Configurable<SpectrumConfig>::config;
    if (x.rows() != c.BS) échec("Spectrum : dimension invalide ({} au lieu de {}).", x.rows(), c.BS);
    const bool complet = tsdgpu_spectrum_pending(h) + 1 == c.nmeans;
    y.resize(complet ? Ns : 0);                                        // (an empty vector until the nmeans-th block, :1334)
    int64_t n = 0;
    if (tsdgpu_spectrum_step(h, x.data(), 1, complet ? y.data() : nullptr, complet ? 1 : 0, &n, nullptr)) échec("rt_spectrum: {}", tsdgpu_last_error());
  }
};
}  // namespace
sptr<Filtre<cfloat, float, SpectrumConfig>> rt_spectrum(const SpectrumConfig &config)
{
  auto res = std::make_shared<SpectreGpu>();
  res->configure(config);
  return res;
}

// ---- filtre_fft: the OLA engine with a spectral callback (fourier.cc:700-940) ------------------
void ola_complexité(entier M, entier Ne, float &C, entier &Nf, entier &Nz)
{
  Nf = prochaine_puissance_de_2(Ne + M - 1);
  Nz = Nf - Ne;
  C = (1.0f / Ne) * 2 * 5 * Nf * std::log(1.0f * Nf) / std::log(2.0f);
}
void ola_complexité_optimise(entier M, float &C_, entier &Nf_, entier &Nz_, entier &Ne_)
{
  const entier kmin = (entier) std::ceil(std::log(M) / std::log(2));       // Nf must be at least M
  for (entier k = kmin; k < kmin + 20 && k < 31; k++) {
    entier Nf, Nz, Ne = (1 << k) - (M - 1);
    float C;
    ola_complexité(M, Ne, C, Nf, Nz);
    if (k == kmin || C < C_) {
      Nf_ = Nf;
      Nz_ = Nf - Ne;
      Ne_ = Ne;
      C_ = C;
    }
  }
}

}  // namespace tsd::fourier

