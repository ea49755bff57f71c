// detection.cc -- pattern detector on the GPU correlators (libtsd core/src/fourier/detection.cc,
// core/include/tsd/fourier.hpp:545-660).  Host side = the reference's peak logic; the two
// per-sample streams it works on (correlation with the pattern, sliding energy) come from the
// MI355X operators: filtre_fft (OLA engine, batched FFTs) or filtre_rif, and filtre_mg.
#include "tsd/fourier.hpp"
#include "tsd_amd/extensions.hpp"
#include "tsd/filtrage.hpp"
#include <algorithm>
#include <vector>

namespace tsd::fourier {

namespace {

// the last K input samples, whatever the block sizes (detection.cc:24-64)
struct MemoireEntree {
  entier K = 0;
  Veccf mem;
  void configure(entier K_)
  {
    K = K_;
    mem = Veccf::zeros(K);
  }
  void step(const Veccf &x)
  {
    const entier n = x.rows();
    if (n >= K) {
      mem = x.tail(K).clone();
    } else {
      const Veccf vieux = mem.tail(K - n).clone();
      mem.head(K - n) = vieux;
      mem.tail(n) = x;
    }
  }
  Veccf derniers(entier n) const { return mem.tail(n).clone(); }
};

// quadratic interpolation of a peak from three samples: position and value (detection.cc:9-21)
float pic_position(float ym1, float y0, float yp1) { return (yp1 - ym1) / (2 * (2 * y0 - yp1 - ym1)); }
cfloat pic_valeur(cfloat ym1, cfloat y0, cfloat yp1, float δ) { return y0 - (ym1 - yp1) * δ * 0.25f; }

struct DetecteurGpu : Detecteur {
  MemoireEntree entree;                         // input memory for the noise estimate
  sptr<FiltreGen<float>> retard_energie;        // aligns the energy with the OLA correlator's delay
  sptr<FiltreGen<cfloat>> correlateur;
  sptr<FiltreGen<float>> filtre_energie;
  entier itr = 0, dernier_n = 0, Ne = 0, N = 1, M = 0, delais_corr = 0;
  Veccf T_motif, motif;                         // pattern normalised to unit energy, and its spectrum
  float norme_motif = 1;
  bouléen pic_final_a_traiter = false;
  Detection pic_final;
  cfloat lc = 0, lc0 = 0;                       // last two correlation samples of the previous block
  float alc = 0, alc0 = 0;                      // ... and their normalised magnitudes

  explicit DetecteurGpu(const DetecteurConfig &c) { configure(c); }

  void configure_impl(const DetecteurConfig &c) override
  {
    pic_final_a_traiter = false;
    itr = 0;
    double e = 0;
    for (entier i = 0; i < c.motif.rows(); i++) e += std::norm(c.motif(i));
    norme_motif = (float) std::sqrt(e);
    M = c.motif.rows();
    if (M < 3) échec("détecteur: pattern of {} samples (need at least 3)", M);
    motif = c.motif.clone();
    motif /= cfloat(norme_motif, 0);
    filtre_energie = tsd::filtrage::filtre_mg<float, double>(M);
    Ne = (entier) c.Ne;
    if (Ne == 0) {
      float C;
      entier Nf, Nz;
      ola_complexité_optimise(M, C, Nf, Nz, Ne);
    }
    if (c.mode == DetecteurConfig::MODE_OLA) {
      FiltreFFTConfig oc;
      oc.nb_zeros_min = M - 1;
      oc.dim_blocs_temporel = Ne;
      // the reference's callback X *= conj(T_motif) (detection.cc:166-169) as the engine's
      // device-side response: the correlation never leaves the GPU between the two FFTs
      N = prochaine_puissance_de_2(Ne + M - 1);
      if (2 * M > N) échec("détecteur: pattern of {} samples does not fit the {}-point OLA blocks", M, N);
      Veccf tmp = Veccf::zeros(N);
      tmp.head(M) = motif;
      T_motif = fft(tmp);
      Veccf réponse(N);
      for (entier i = 0; i < N; i++) réponse(i) = std::conj(T_motif(i));
      auto [f, n_fft] = tsd_amd::filtre_fft_reponse(oc, réponse);
      correlateur = f;
      if (n_fft != N) échec("détecteur: OLA engine configured with N = {} instead of {}", n_fft, N);
      delais_corr = Ne;
      retard_energie = tsd::filtrage::ligne_a_retard<float>(delais_corr - M + 1);
    } else {
      N = 1;                                    // (the reference leaves N at its initial value in this mode)
      double im = 0, tot = 0;
      for (entier i = 0; i < M; i++) {
        im += std::abs(motif(i).imag());
        tot += std::abs(motif(i));
      }
      if (im / tot < 1e-7) {
        correlateur = tsd::filtrage::filtre_rif<float, cfloat>(real(motif.reverse()));
      } else {
        Veccf h = motif.reverse();
        for (entier i = 0; i < M; i++) h(i) = std::conj(h(i));
        correlateur = tsd::filtrage::filtre_rif<cfloat, cfloat>(h);
      }
      retard_energie = nullptr;
      delais_corr = M - 1;
    }
    entree.configure(delais_corr + 1);
  }

  void step(const Veccf &x, Vecf &y) override
  {
    const DetecteurConfig &c = Configurable<DetecteurConfig>::config;
    const entier n = x.rows();
    if (n < 2) échec("détecteur: blocks of at least 2 samples expected (got {})", n);
    Vecf en = filtre_energie->step(abs2(x));
    if (retard_energie) en = retard_energie->step(en);
    Veccf corr = correlateur->step(x);
    if (corr.rows() != n) échec("Sortie OLA (corr) devrait faire {} échantillons, mais {}.", n, corr.rows());
    const float ratio = std::sqrt(1.0f * N) / std::sqrt(1.0f * M);
    y.resize(n);
    for (entier i = 0; i < n; i++) {
      if (std::abs(corr(i)) <= std::sqrt(1e-12f)) corr(i) = 0;       // drop numerically empty values
      y(i) = ratio * std::sqrt(std::norm(corr(i)) / (en(i) + 1e-20f));
    }
    // candidates: the largest value of every M-sample segment, above the threshold, not dominated
    // by a larger candidate closer than M samples
    std::vector<entier> cand, pics;
    for (entier i = 0; i < n; i += M) {
      const entier len = std::min(M, n - i);
      entier im = i;
      for (entier k = i; k < i + len; k++)
        if (y(k) > y(im)) im = k;
      if (y(im) > c.seuil) cand.push_back(im);
    }
    if (pic_final_a_traiter) {
      pic_final_a_traiter = false;
      pics.push_back(-1);
    }
    for (entier idx : cand) {
      bool ok = true;
      for (entier idx2 : cand)
        if (y(idx2) > y(idx) && std::abs(idx - idx2) < M) {
          ok = false;
          break;
        }
      if (ok) pics.push_back(idx);
    }
    for (entier idx : pics) {
      Detection det;
      cfloat c0, c1, c2;
      float ac0, ac1, ac2;
      if (idx == -1) {
        // the last sample of the previous block was a candidate: it needed this block's first sample
        det = pic_final;
        det.position -= dernier_n;
        det.position_prec -= dernier_n;
        ac0 = alc0; c0 = lc0;
        ac1 = alc; c1 = lc;
        ac2 = y(0); c2 = corr(0);
        if (ac1 < ac2) break;                    // it was not the peak after all
      } else {
        det.score = y(idx);
        det.position = idx - delais_corr;
        det.θ = std::arg(corr(idx));
        det.gain = std::abs(corr(idx)) / (norme_motif / std::sqrt((float) N));
        det.position_prec = (float) det.position;
        if (idx == 0) {
          ac0 = alc; c0 = lc;
          ac1 = y(0); c1 = corr(0);
          ac2 = y(1); c2 = corr(1);
          if (ac1 < ac0) break;                  // the previous block's last sample was larger
        } else if (idx == n - 1) {
          pic_final_a_traiter = true;            // handled at the start of the next block
          pic_final = det;
          break;
        } else {
          ac0 = y(idx - 1); c0 = corr(idx - 1);
          ac1 = y(idx); c1 = corr(idx);
          ac2 = y(idx + 1); c2 = corr(idx + 1);
        }
      }
      float δ = pic_position(ac0, ac1, ac2);
      δ = std::clamp(δ, -0.5f, 0.5f);
      det.position_prec += δ;
      {
        const cfloat g2 = pic_valeur(c0, c1, c2, δ) * (std::sqrt((float) N) / norme_motif);
        det.gain = std::abs(g2);
        det.θ = std::arg(g2);
      }
      // noise = what was received minus the pattern as estimated (gain, phase, fractional position)
      Veccf recu_theo = c.motif.clone();
      recu_theo *= std::polar(det.gain, det.θ);
      recu_theo = délais(recu_theo, δ);
      Veccf recu(M);
      const entier id = idx - delais_corr;
      entier dans_x = id >= 0 ? M : (id > -M ? M + id : 0);
      const entier avant = M - dans_x;
      if (dans_x > 0) recu.tail(dans_x) = x.segment(std::max(id, 0), dans_x);
      if (avant > 0) {
        if (avant < M)
          recu.head(avant) = entree.derniers(avant);
        else
          recu.head(avant) = entree.mem.segment(std::clamp(id + entree.K, 0, entree.K - avant), avant);
      }
      double vb = 0;
      for (entier i = 1; i <= M - 2; i++) vb += std::norm(recu(i) - recu_theo(i));
      const float var_bruit = (float) (vb / (M - 2));
      const float var_signal = std::pow(det.gain * norme_motif, 2.0f) / M;
      det.σ_noise = std::sqrt(var_bruit);
      det.SNR_dB = 10 * std::log10(var_signal / var_bruit);
      if (c.gere_detection) c.gere_detection(det);
    }
    entree.step(x);
    alc0 = y(n - 2);
    lc0 = corr(n - 2);
    alc = y(n - 1);
    lc = corr(n - 1);
    itr++;
    dernier_n = n;
  }
};

}  // namespace

sptr<Detecteur> détecteur_création(const DetecteurConfig &config) { return std::make_shared<DetecteurGpu>(config); }

}  // namespace tsd::fourier
