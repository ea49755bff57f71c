// gpu_polyphase.cc -- the factories of libtsd's core/src/reechan/polyphase.cc on the MI355X C ABI:
// filtre_rif_decim (:156-239), filtre_rif_demi_bande (:54-149), filtre_rif_ups (:246-341),
// filtre_rif_ups_délais / rif_delais (:363-376), forme_polyphase / iforme_polyphase (:16-46);
// instantiated like polyphase.cc:379-384.
#include "gpu_commun.hpp"

namespace tsd::filtrage {

using tsd_amd::dtype_of;
using tsd_amd::gpu_fail;

// One fused kernel per stage (polyphase.hip): only the kept outputs are computed.
//   DECIM: taps applied in FORWARD order against the oldest->newest window, one output in R (:223-229)
//   HALFBAND: even taps + the 0.5 centre tap, R = 2 (:120-139)
//   UPS: taps * R, zero-padded to a multiple of R, phase i = taps (R-1-i) + jR (:259-270,313-338)
template <typename T> struct EtagePolyphaseGpu : FiltreGen<T> {
  tsdgpu_polyfir *h = nullptr;
  EtagePolyphaseGpu(int kind, const float *taps, int K, int R)
  {
    if (tsdgpu_polyfir_create(&h, kind, dtype_of<T>(), taps, K, R)) gpu_fail("filtre_rif_decim/_demi_bande/_ups");
  }
  ~EtagePolyphaseGpu() { tsdgpu_polyfir_destroy(h); }
  void step(const Vecteur<T> &x, Vecteur<T> &y)
  {
    const entier n = x.rows();
    const int64_t cap = tsdgpu_polyfir_out_count(h, n);
    tsd_amd::sortie_variable(x, y, cap, [&](T *out) {
      int64_t got = 0;
      if (n > 0 && tsdgpu_polyfir_step(h, x.data(), n, out, cap, &got, nullptr)) gpu_fail("polyphase stage step");
    });
  }
};

template <typename Tc, typename T> sptr<FiltreGen<T>> filtre_rif_decim(const Vecteur<Tc> &c, entier R)
{
  return std::make_shared<EtagePolyphaseGpu<T>>(TSDGPU_POLY_DECIM, c.data(), c.rows(), R);
}
template <typename Tc, typename T> sptr<FiltreGen<T>> filtre_rif_demi_bande(const Vecteur<Tc> &c)
{
  return std::make_shared<EtagePolyphaseGpu<T>>(TSDGPU_POLY_HALFBAND, c.data(), c.rows(), 2);
}
template <typename Tc, typename T> sptr<FiltreGen<T>> filtre_rif_ups(const Vecteur<Tc> &c, entier R)
{
  return std::make_shared<EtagePolyphaseGpu<T>>(TSDGPU_POLY_UPS, c.data(), c.rows(), R);
}
template sptr<FiltreGen<float>> filtre_rif_decim<float, float>(const Vecteur<float> &, entier);
template sptr<FiltreGen<cfloat>> filtre_rif_decim<float, cfloat>(const Vecteur<float> &, entier);
template sptr<FiltreGen<float>> filtre_rif_demi_bande<float, float>(const Vecteur<float> &);
template sptr<FiltreGen<cfloat>> filtre_rif_demi_bande<float, cfloat>(const Vecteur<float> &);
template sptr<FiltreGen<float>> filtre_rif_ups<float, float>(const Vecteur<float> &, entier);
template sptr<FiltreGen<cfloat>> filtre_rif_ups<float, cfloat>(const Vecteur<float> &, entier);

float filtre_rif_ups_délais(entier nc, entier R)     // polyphase.cc:363-371
{
  entier pad = 0;
  if ((nc % R) != 0) pad = R - (nc % R);
  return (float) ((nc - 1) / 2.0 + pad);
}
float rif_delais(entier nc) { return (nc - 1) / 2.0f; }

#ifndef TSD_AMD_MIRROR
// forme_polyphase (polyphase.cc:16-46) on libtsd's own 2-D array type: zero-pad to a multiple of M,
// then M rows x n/M columns, column-major -- a pure index permutation.  (The mirror has no 2-D array;
// its header-inline twin returns the same memory image, include/tsd/filtrage.hpp.)
template <typename T> TabT<T, 2> forme_polyphase(const Vecteur<T> &x, entier M)
{
  const entier n = x.rows(), nb = (n + M - 1) / M;
  TabT<T, 2> X(M, nb);
  for (entier j = 0; j < nb; j++)
    for (entier i = 0; i < M; i++) X(i, j) = (i + j * M < n) ? x(i + j * M) : T(0);
  return X;
}
template <typename T> Vecteur<T> iforme_polyphase(const TabT<T, 2> &X)
{
  const entier M = X.rows(), nb = X.cols();
  Vecteur<T> x(M * nb);
  for (entier j = 0; j < nb; j++)
    for (entier i = 0; i < M; i++) x(i + j * M) = X(i, j);
  return x;
}
template TabT<float, 2> forme_polyphase<float>(const Vecteur<float> &, entier);
template TabT<cfloat, 2> forme_polyphase<cfloat>(const Vecteur<cfloat> &, entier);
template Vecteur<float> iforme_polyphase<float>(const TabT<float, 2> &);
template Vecteur<cfloat> iforme_polyphase<cfloat>(const TabT<cfloat, 2> &);
#endif

}  // namespace tsd::filtrage
