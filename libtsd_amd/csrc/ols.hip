// ols.hip -- overlap-save FFT convolution for long FIR filters (placeholder until the
// wave-level FFT lands; AUTO never selects it and an explicit request is refused).
#include "fir_internal.hpp"
namespace tsdgpu {
bool ols_preferred(const tsdgpu_fir *) { return false; }
int ols_plan_create(tsdgpu_fir *) { return set_err(TSDGPU_ERR_UNSUPPORTED, "overlap-save FIR not built yet"); }
void ols_plan_destroy(tsdgpu_fir *f) { if (f->d_H) (void) hipFree(f->d_H); f->d_H = nullptr; }
int ols_step(tsdgpu_fir *, const void *, void *, int64_t, hipStream_t) { return set_err(TSDGPU_ERR_UNSUPPORTED, "overlap-save FIR not built yet"); }
}
