// tsd_amd/extensions.hpp -- what the MI355X adaptors offer BEYOND libtsd's own API.  Nothing in
// libtsd's headers is changed to make room for these: they are free functions in namespace
// tsd_amd, declared against whichever "tsd/*.hpp" is on the include path (libtsd's or the mirror).
#pragma once
#include "tsd/tsd.hpp"
#include "tsd/filtrage.hpp"
#include "tsd/fourier.hpp"
#include <tuple>

namespace tsd_amd {
using namespace tsd;   // sptr / entier / cfloat: members of tsd in the mirror, global aliases in libtsd (fr.hpp, commun.hpp)

// ---- FFT plan hook (core/include/tsd/fourier.hpp:35, core/src/fourier/fourier.cc:469-481) ----------
// A FFTPlan on tsdgpu_fft.  installe_fftplan_gpu() assigns the factory to tsd::fourier::fftplan_defaut,
// after which every fft() / ifft() / tfrplan_création() / TFRCorrelateurBloc / Spectrum of libtsd
// runs its transforms on the GPU.  (The mirror's fftplan_defaut starts out as this factory.)
sptr<tsd::fourier::FFTPlan> fftplan_gpu();
void installe_fftplan_gpu();
// RTFRPlan on tsdgpu_rfft (packed half-size transform, untangling and forced symmetry fused on the device)
sptr<FiltreGen<float, cfloat>> rtfrplan_gpu(entier n = -1);

// ---- filtre_fft with a DEVICE-side spectral response ---------------------------------------------
// libtsd's OLA engine calls config.traitement_freq(X) on the host for every block.  When the
// processing is a product by a fixed response H (N values, N = the FFT size filtre_fft returns) this
// form keeps everything on the GPU: X *= H.  config.traitement_freq, when also set, still runs (after
// the product) through the host bridge.
std::tuple<sptr<Filtre<cfloat, cfloat, tsd::fourier::FiltreFFTConfig>>, entier>
filtre_fft_reponse(const tsd::fourier::FiltreFFTConfig &config, const Veccf &H);
// FFT size N the OLA engine picks for a configuration (what H must be sized to)
entier filtre_fft_dim(const tsd::fourier::FiltreFFTConfig &config);

// ---- several GPUs behind one operator object -----------------------------------------------------------
// filtre_rif / filtre_sois / filtre_itrp (hence filtrer, rééchan ...) cut a large host vector (>= 2^22
// samples) into one contiguous chunk per GPU of the node, each with its small halo, and run all devices
// at once (include/tsdgpu.h: tsdgpu_sharded_step_host).  Automatic when the process sees several GPUs;
// fixe_fragments(n) forces n logical shards (spread round-robin over the devices present; 0 or 1: off,
// -1: automatic again), like the environment variable TSD_AMD_SHARDS.
void fixe_fragments(int n);

// ---- device memory for resident vectors ------------------------------------------------------------
// A vector mapped on device memory, TabT<T,1>::map(ptr, n) (tableau.hpp:1067-1077), is accepted by
// every adaptor as input, and as output when it already has the size the step produces (resize() to
// the same size is a no-op in libtsd, tableau.cc:702-705): a chain filtre -> fft -> rééchan then
// never leaves the GPU.  These two wrap hipMalloc / hipFree so that C++ callers need no HIP headers.
void *alloue_gpu(size_t octets);
void libere_gpu(void *p);
void copie_vers_gpu(void *dst_gpu, const void *src_hote, size_t octets);
void copie_vers_hote(void *dst_hote, const void *src_gpu, size_t octets);
void synchronise_gpu();

}  // namespace tsd_amd
