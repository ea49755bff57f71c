// fir_internal.hpp -- FIR handle shared by the direct (fir.hip) and overlap-save (ols.hip) paths.
#pragma once
#include "common.hpp"
#include <vector>

struct tsdgpu_fir {
  int data_type = 0, tap_type = 0;
  int K = 0;            // taps
  int KP = 0;           // taps padded to a multiple of 2R (direct kernel)
  int R = 8;            // outputs per lane of the direct kernel
  int HL = 0;           // history length kept in hist[]: >= KP and >= the overlap-save overlap
  int method = TSDGPU_FIR_DIRECT;
  void *d_hrev = nullptr;   // reversed zero-padded taps, KP entries of tap_type
  void *hist[2] = {nullptr, nullptr};   // last HL input samples (double-buffered), newest last
  const void *hist_ext = nullptr;       // tsdgpu_fir_step_after: this call reads its HL history samples here instead (the caller's own buffer)
  int cur = 0;
  bool capturable = false;  // tsdgpu_fir_set_capturable: every step reads hist[0] and leaves the new history there
  std::vector<char> taps_host;
  tsdgpu::DevBuf in_stage, out_stage;
  // overlap-save plan (ols.hip)
  void *d_H = nullptr;      // frequency response in the kernel's register order
  int ols_N = 0;            // FFT block size
  int ols_L = 0;            // valid outputs per block = N - (K-1)
  int ols_grid = 0;         // persistent grid size (waves)
  unsigned *d_ctr = nullptr;   // work counters of the dynamic block hand-out (ols.hip: OlsDyn), behind d_H
  int ctr_nc = 0;              // how many of them the launches so far have used
  unsigned ctr_base = 0;       // their common value before the next launch (every launch advances all of them alike)
  bool ols_long = false;    // long-filter plan (ols_long.hip): N = 4096..16384, H in natural order + W_N table
  // partitioned plan for more than 12289 taps: the taps cut in segments of part_S, one child filter per
  // segment (each on the long-filter overlap-save plan), y = sum_p child_p(x delayed by p * part_S)
  std::vector<tsdgpu_fir *> parts;
  int part_S = 0;
  tsdgpu::DevBuf part_x, part_t;   // [history ++ x] and the partial sum of one segment
};

namespace tsdgpu {
inline const void *fir_hist_read(const tsdgpu_fir *f) { return f->hist_ext ? f->hist_ext : f->hist[f->cur]; }
int fir_direct_step(tsdgpu_fir *f, const void *x, void *y, int64_t n, hipStream_t st);
int fir_update_history(tsdgpu_fir *f, const void *x, int64_t n, hipStream_t st);
int fir_settle_history(tsdgpu_fir *f, hipStream_t st);
// overlap-save
bool ols_preferred(const tsdgpu_fir *f);
int ols_plan_create(tsdgpu_fir *f);
void ols_plan_destroy(tsdgpu_fir *f);
int ols_step(tsdgpu_fir *f, const void *x, void *y, int64_t n, hipStream_t st);
// overlap-ADD fast path (Ne = 512, N = 1024, no window, device response) on the in-wave 1024-point transform: ols.hip
void ola1024_tables(const float2 *H_host, float2 *out3);
// Welch sums for N = 1024 on the in-wave transform (ols.hip): part[grid][1024], grid = ceil(nseg / per)
int welch1024_launch(const float2 *x, const float *w, const float2 *tw2x1024, float *part, int64_t nseg, int per, hipStream_t st);
void welch1024_tables(float2 *out2);
int ola1024_launch(const float2 *x, float2 *y, const float2 *tables3, const float2 *svg_in, float2 *svg_out, int64_t B, hipStream_t st);
// the windowed engine at Ne = N = 512 on the in-wave pair transform (ols.hip: olaw512_kernel); tables: 512 + 2 x 1024 complex values
void olaw512_tables(const float2 *H_host, float2 *out);
int olaw512_launch(const float2 *blk0, int nrest, const float2 *x, float2 *y, const float2 *tables, const float *fen, const float2 *svg_in,
                   const float2 *last_in, const float2 *prev_half_in, float2 *svg_out, float2 *last_out, int64_t B, int skip_first, hipStream_t st);
// overlap-save for long filters (514..12289 taps), radix-16 Stockham blocks
bool ols_long_supported(const tsdgpu_fir *f);
int ols_long_plan_create(tsdgpu_fir *f);
int ols_long_step(tsdgpu_fir *f, const void *x, void *y, int64_t n, hipStream_t st);
}  // namespace tsdgpu
