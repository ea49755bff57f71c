// gpu_filtre_fft.cc -- libtsd's frequency-domain filters on the MI355X C ABI:
//   tsd::fourier::filtre_fft(FiltreFFTConfig)   = OLA<cfloat>     core/src/fourier/fourier.cc:737-940
//   tsd::filtrage::filtre_rif_fft<T>(coefs)     = FiltreFFTRIF<T> core/src/fourier/fourier.cc:946-990,1443-1447
// Compiled against libtsd's headers the two definitions carry libtsd's mangled names: they take the
// place of the ones in fourier.cc (see INTEGRATION.md for the two ways of doing that).
#include "gpu_commun.hpp"
#include "tsd_amd/extensions.hpp"
#include <algorithm>

namespace tsd_amd {

using namespace tsd;
using tsd::fourier::FiltreFFTConfig;

// The OLA engine (tsdgpu_ola_*, ola.hip): re-blocking, framing ([Nz zeros | block], or two
// Hann-windowed half-overlapping frames per block), ONE batched FFT each way for all the blocks of a
// call and the overlap-add run on the GPU.  config.traitement_freq (host code) sees the spectra one
// by one, in order, through one copy out / copy back per call; a device-side response (extension)
// needs no copy at all.
struct OLAGpu : Filtre<cfloat, cfloat, FiltreFFTConfig> {
  entier N = 0, Ne = 0;
  tsdgpu_ola *h = nullptr;
  std::vector<cfloat> réponse;      // extension: X *= réponse on the device
  ~OLAGpu() { tsdgpu_ola_destroy(h); }

  void configure_impl(const FiltreFFTConfig &c)
  {
    if (!c.traitement_freq && réponse.empty()) échec("configuration OLA : traitement fréquentiel non précisé.");
    tsdgpu_ola_destroy(h);
    h = nullptr;
    const entier ne = c.dim_blocs_temporel > 0 ? c.dim_blocs_temporel : 512;
    Vecf fen;
    if (c.avec_fenetrage) fen = tsd::filtrage::fenêtre("hn", ne, false);          // fourier.cc:795
    if (tsdgpu_ola_create(&h, c.dim_blocs_temporel, c.nb_zeros_min, c.avec_fenetrage ? fen.data() : nullptr)) gpu_fail("filtre_fft");
    N = tsdgpu_ola_fft_size(h);
    Ne = tsdgpu_ola_block_len(h);
    if (!réponse.empty()) {
      if ((entier) réponse.size() != N) échec("filtre_fft: the response has {} values, the FFT size is {}", (int) réponse.size(), (int) N);
      if (tsdgpu_ola_set_response(h, réponse.data())) gpu_fail("filtre_fft");
    }
  }

  TamponGpu sortie_gpu;
  void step(const Veccf &x, Veccf &y)
  {
    const FiltreFFTConfig &c = Configurable<FiltreFFTConfig>::config;
    if (!h) échec("filtre_fft: not configured");
    // the engine tells the output count only when it is done: it writes a scratch of the bounded size --
    // device memory when x is resident (nothing crosses PCIe), host memory otherwise -- and the
    // outputs go from there to y
    const size_t borne = (size_t) std::max<int64_t>(1, tsdgpu_ola_max_out(h, x.rows()));
    const bool résident = x.rows() > 0 && tsdgpu_is_device_pointer(x.data());
    std::vector<cfloat> sortie_hote;
    cfloat *out;
    if (résident) out = static_cast<cfloat *>(sortie_gpu.reserve(borne * sizeof(cfloat)));
    else { sortie_hote.resize(borne); out = sortie_hote.data(); }
    int64_t nout = 0;
    if (!c.traitement_freq) {
      // device-side processing only: framing, FFTs, product and overlap-add in one call
      if (tsdgpu_ola_step(h, x.data(), x.rows(), out, &nout, nullptr)) gpu_fail("filtre_fft");
    } else {
      void *sp = nullptr;
      int nf = 0;
      if (tsdgpu_ola_analyse(h, x.data(), x.rows(), &sp, &nf, nullptr)) gpu_fail("filtre_fft");
      if (nf > 0) {
        if (tsdgpu_ola_apply_response(h, nullptr)) gpu_fail("filtre_fft");          // the device-side response first, if any
        std::vector<cfloat> S((size_t) nf * N);
        if (tsdgpu_ola_read_spectra(h, S.data(), nullptr)) gpu_fail("filtre_fft");
        Veccf X(N);
        for (entier f = 0; f < nf; f++) {
          copie_octets(X.data(), S.data() + (size_t) f * N, (size_t) N * sizeof(cfloat));
          c.traitement_freq(X);
          if (X.rows() != N) échec("filtre_fft: traitement_freq must keep the dimension of the spectrum ({})", (int) N);
          copie_octets(S.data() + (size_t) f * N, X.data(), (size_t) N * sizeof(cfloat));
        }
        if (tsdgpu_ola_write_spectra(h, S.data(), nullptr)) gpu_fail("filtre_fft");
      }
      if (tsdgpu_ola_synthese(h, out, &nout, nullptr)) gpu_fail("filtre_fft");
    }
    dimensionne(y, (entier) nout);
    copie_octets(y.data(), out, (size_t) nout * sizeof(cfloat));
  }
};

std::tuple<sptr<Filtre<cfloat, cfloat, FiltreFFTConfig>>, entier> filtre_fft_reponse(const FiltreFFTConfig &config, const Veccf &H)
{
  auto res = std::make_shared<OLAGpu>();
  res->réponse.assign(H.data(), H.data() + H.rows());
  res->Configurable<FiltreFFTConfig>::configure(config);
  return {res, res->N};
}
entier filtre_fft_dim(const FiltreFFTConfig &c)
{
  const entier ne = c.dim_blocs_temporel > 0 ? c.dim_blocs_temporel : 512;
  return prochaine_puissance_de_2(ne + c.nb_zeros_min);                             // fourier.cc:764-770
}

// FiltreFFTRIF (fourier.cc:946-990): the reference's OLA FIR.  Its output is the direct FIR delayed
// by Nz - M samples (Ne = 512, N = pp2(Ne + M), Nz = N - Ne), and for T = cfloat only the real part
// survives (fourier.cc:976).  Both are reproduced: the block convolution runs on the GPU overlap-save
// kernel (one HBM pass, 60 % of roofline), the delay line lives here.
template <typename T> struct FiltreFFTRIFGpu : FiltreGen<T> {
  tsdgpu_fir *h = nullptr;
  size_t d = 0;               // Nz - M
  void *retard = nullptr;     // device: the last d outputs not yet delivered
  TamponGpu z;                // device: the filter output of the current call
  explicit FiltreFFTRIFGpu(const Vecf &c)
  {
    const entier M = c.rows(), Ne = 512;
    if (M <= 0) échec("filtre_rif_fft: no coefficient");
    const entier N = prochaine_puissance_de_2(Ne + M), Nz = N - Ne;
    if (Nz > Ne) échec("filtre_rif_fft: {} coefficients need Nz = {} > Ne = {} (the reference's OLA limit)", (int) M, (int) Nz, (int) Ne);
    d = (size_t) (Nz - M);
    if (d > 0) {
      if (tsdgpu_malloc(&retard, d * sizeof(T)) || tsdgpu_memset(retard, 0, d * sizeof(T), nullptr)) gpu_fail("filtre_rif_fft");
    }
    if (tsdgpu_fir_create(&h, dtype_of<T>(), TSDGPU_F32, c.data(), M, TSDGPU_FIR_OVERLAP_SAVE)) gpu_fail("filtre_rif_fft");
  }
  ~FiltreFFTRIFGpu()
  {
    tsdgpu_fir_destroy(h);
    tsdgpu_free(retard);
  }
  void step(const Vecteur<T> &x, Vecteur<T> &y)
  {
    const size_t n = (size_t) x.rows(), sz = sizeof(T);
    if (n == 0) {
      dimensionne(y, 0);
      return;
    }
    // z = FIR(x) on the device (the input may be a host or a resident vector); only its real part for
    // complex data (fourier.cc:976); then y = (retard ++ z)[0, n), retard <- the rest
    char *zz = static_cast<char *>(z.reserve((n + d) * sz));
    if (tsdgpu_fir_step(h, x.data(), zz + d * sz, (int64_t) n, nullptr)) gpu_fail("filtre_rif_fft::step");
    if constexpr (std::is_same_v<T, cfloat>)
      if (tsdgpu_zero_imag(zz + d * sz, (int64_t) n, nullptr)) gpu_fail("filtre_rif_fft::step");
    if (d > 0) copie_octets(zz, retard, d * sz);
    if (x.data() != y.data()) dimensionne(y, (entier) n);
    copie_octets(y.data(), zz, n * sz);
    if (d > 0) copie_octets(retard, zz + n * sz, d * sz);
  }
};

}  // namespace tsd_amd

namespace tsd::fourier {
std::tuple<sptr<Filtre<cfloat, cfloat, FiltreFFTConfig>>, entier> filtre_fft(const FiltreFFTConfig &config)
{
  auto res = std::make_shared<tsd_amd::OLAGpu>();
  res->Configurable<FiltreFFTConfig>::configure(config);
  return {res, res->N};
}
}  // namespace tsd::fourier

namespace tsd::filtrage {
template <typename T> sptr<FiltreGen<T>> filtre_rif_fft(const Vecf &c) { return std::make_shared<tsd_amd::FiltreFFTRIFGpu<T>>(c); }
template sptr<FiltreGen<float>> filtre_rif_fft<float>(const Vecf &);
template sptr<FiltreGen<cfloat>> filtre_rif_fft<cfloat>(const Vecf &);
}  // namespace tsd::filtrage
