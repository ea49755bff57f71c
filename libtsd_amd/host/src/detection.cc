// detection.cc -- mirror runtime: détecteur_création (libtsd core/src/fourier/detection.cc,
// core/include/tsd/fourier.hpp:545-660) on the device-side detector of the C ABI (tsdgpu_detector_*,
// csrc/detect.hip).  The GPU produces the score stream and, per block, the list of peaks with the three
// scores and the three complex correlation values around each; what is left here is per DETECTION, not
// per sample: sub-sample position, complex amplitude, and the noise estimate against the received
// samples.  A peak is decided M samples after it occurred, so it may be reported with the block that
// FOLLOWS the one it lies in; its position is always relative to the block being processed (negative
// then), which is the convention of Detection::position.
#include "tsd/fourier.hpp"
#include "tsd/filtrage.hpp"
#include "tsdgpu.h"
#include <algorithm>
#include <vector>

namespace tsd::fourier {

namespace {

struct DetecteurGpu : Detecteur {
  tsdgpu_detector *h = nullptr;
  entier M = 0, N = 1, retard = 0;
  float norme = 1;                       // sqrt of the pattern's energy
  std::vector<cfloat> passé;             // the stream's most recent samples before the current block

  explicit DetecteurGpu(const DetecteurConfig &c) { configure(c); }
  ~DetecteurGpu() override { tsdgpu_detector_destroy(h); }

  void configure_impl(const DetecteurConfig &c) override
  {
    tsdgpu_detector_destroy(h);
    h = nullptr;
    M = c.motif.rows();
    if (M < 3) échec("détecteur: pattern of {} samples (need at least 3)", M);
    double énergie = 0;
    for (entier i = 0; i < M; i++) énergie += std::norm(c.motif(i));
    norme = (float) std::sqrt(énergie);
    std::vector<cfloat> unitaire((size_t) M);
    for (entier i = 0; i < M; i++) unitaire[(size_t) i] = c.motif(i) / norme;
    entier Ne = (entier) c.Ne;
    if (Ne == 0) {
      float coût;
      entier Nf, Nz;
      ola_complexité_optimise(M, coût, Nf, Nz, Ne);
    }
    if (tsdgpu_detector_create(&h, unitaire.data(), M, Ne, c.mode == DetecteurConfig::MODE_OLA ? 0 : 1, c.seuil))
      échec("détecteur: {}", tsdgpu_last_error());
    N = tsdgpu_detector_fft_size(h);
    retard = tsdgpu_detector_delay(h);
    passé.assign((size_t) (retard + 2 * M), cfloat(0));
  }

  void step(const Veccf &x, Vecf &y) override
  {
    const DetecteurConfig &c = Configurable<DetecteurConfig>::config;
    const entier n = x.rows();
    if (n < 2) échec("détecteur: blocks of at least 2 samples expected (got {})", n);
    y.resize(n);
    tsdgpu_peak pics[256];
    int npics = 0;
    if (tsdgpu_detector_step(h, x.data(), n, y.data(), pics, 256, &npics, nullptr)) échec("détecteur: {}", tsdgpu_last_error());
    for (int k = 0; k < npics; k++) {
      const tsdgpu_peak &p = pics[k];
      Detection det;
      det.score = p.s0;
      det.position = p.index - retard;
      // vertex of the parabola through the three scores; the complex amplitude read at the vertex
      float δ = (p.s_p1 - p.s_m1) / (2 * (2 * p.s0 - p.s_p1 - p.s_m1));
      δ = std::clamp(δ, -0.5f, 0.5f);
      det.position_prec = (float) det.position + δ;
      const cfloat am1(p.c_m1[0], p.c_m1[1]), a0(p.c0[0], p.c0[1]), ap1(p.c_p1[0], p.c_p1[1]);
      const cfloat amplitude = (a0 - (am1 - ap1) * (0.25f * δ)) * (std::sqrt((float) N) / norme);
      det.gain = std::abs(amplitude);
      det.θ = std::arg(amplitude);
      // noise: the M received samples minus the pattern as estimated (gain, phase, fractional position)
      Veccf modèle = c.motif.clone();
      modèle *= amplitude;
      modèle = délais(modèle, δ);
      const int64_t début = (int64_t) passé.size() + det.position;        // in (passé ++ x)
      double écart = 0;
      entier compte = 0;
      for (entier i = 1; i <= M - 2; i++) {
        const int64_t q = début + i;
        if (q < 0 || q >= (int64_t) passé.size() + n) continue;
        const cfloat reçu = q < (int64_t) passé.size() ? passé[(size_t) q] : x.data()[q - (int64_t) passé.size()];
        écart += std::norm(reçu - modèle(i));
        compte++;
      }
      const float var_bruit = compte > 0 ? (float) (écart / compte) : 0.f;
      const float var_signal = det.gain * det.gain * norme * norme / M;
      det.σ_noise = std::sqrt(var_bruit);
      det.SNR_dB = 10 * std::log10(var_signal / var_bruit);
      if (c.gere_detection) c.gere_detection(det);
    }
    // keep the tail of the stream for the detections the next block will report
    const size_t garde = passé.size();
    std::vector<cfloat> tout(passé);
    tout.insert(tout.end(), x.data(), x.data() + n);
    passé.assign(tout.end() - (std::ptrdiff_t) garde, tout.end());
  }
};

}  // namespace

sptr<Detecteur> détecteur_création(const DetecteurConfig &config) { return std::make_shared<DetecteurGpu>(config); }

}  // namespace tsd::fourier
