// perf_host_api.cc -- what the C++ drop-in API costs on the headline configuration (127-tap FIR,
// 2^26 cfloat), by where the vectors live:
//   host, pageable   a Veccf wrapped around malloc'd memory (what libtsd's own Tab allocates)
//   host, pinned     the mirror's Veccf (page-locked from 1 MiB): chunked H2D / kernel / D2H pipeline
//   resident         vectors in device memory (ResidenceGpu / vers_gpu / map): the benchmarked kernel
// usage: perf_host_api [log2n]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "dsp/dsp.hpp"
#include "dsp/filter.hpp"
#include "tsd_amd/extensions.hpp"

using namespace tsd;
using namespace tsd::filtrage;

static double ms_since(std::chrono::steady_clock::time_point t0)
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char **argv)
{
  const int log2n = argc > 1 ? atoi(argv[1]) : 26;
  const int n = 1 << log2n;
  const Vecf h = design_rif_fen(127, "lp", 0.02f);
  auto f = filtre_rif<float, cfloat>(h);
  const double gs = 1e-6 * n;   // Msamples
  // pinned host vectors (the mirror's default for large vectors)
  Veccf x(n), y(n);
  for (int i = 0; i < n; i++) x.data()[i] = cfloat((float) (i & 1023) / 1024.f, (float) ((i * 7) & 511) / 512.f);
  f->step(x, y);
  double best = 1e30;
  for (int it = 0; it < 5; it++) {
    auto t0 = std::chrono::steady_clock::now();
    f->step(x, y);
    best = std::min(best, ms_since(t0));
  }
  printf("{\"case\": \"filtre_rif step, host pinned Veccf\", \"log2n\": %d, \"ms\": %.3f, \"Msamples_s\": %.0f}\n", log2n, best, gs / best * 1e3);
  // one-shot filtrer(): creates the filter and the output vector per call, like libtsd
  best = 1e30;
  for (int it = 0; it < 3; it++) {
    auto t0 = std::chrono::steady_clock::now();
    Veccf yy = filtrer<cfloat>(Design(h), x);
    best = std::min(best, ms_since(t0));
  }
  printf("{\"case\": \"filtrer() one-shot, host pinned Veccf (allocates its output)\", \"log2n\": %d, \"ms\": %.3f, \"Msamples_s\": %.0f}\n", log2n, best, gs / best * 1e3);
  // rééchan-style step on host vectors: variable output length through the same chunked pipeline
  {
    auto r = filtre_reechan<cfloat>(160.f / 147);
    Veccf yr;
    r->step(x, yr);
    best = 1e30;
    for (int it = 0; it < 3; it++) {
      auto t0 = std::chrono::steady_clock::now();
      r->step(x, yr);
      best = std::min(best, ms_since(t0));
    }
    printf("{\"case\": \"filtre_reechan(160/147) step, host pinned Veccf\", \"log2n\": %d, \"ms\": %.3f, \"Msamples_s\": %.0f, \"outputs\": %d}\n", log2n, best,
           gs / best * 1e3, yr.rows());
  }
  // pageable host memory (malloc): what a libtsd Tab holds
  {
    cfloat *px = (cfloat *) malloc((size_t) n * sizeof(cfloat)), *py = (cfloat *) malloc((size_t) n * sizeof(cfloat));
    memcpy(px, x.data(), (size_t) n * sizeof(cfloat));
    memset(py, 0, (size_t) n * sizeof(cfloat));
    Veccf vx = Veccf::map(px, n), vy = Veccf::map(py, n);
    f->step(vx, vy);
    best = 1e30;
    for (int it = 0; it < 3; it++) {
      auto t0 = std::chrono::steady_clock::now();
      f->step(vx, vy);
      best = std::min(best, ms_since(t0));
    }
    printf("{\"case\": \"filtre_rif step, host pageable (malloc) memory\", \"log2n\": %d, \"ms\": %.3f, \"Msamples_s\": %.0f}\n", log2n, best, gs / best * 1e3);
    free(px);
    free(py);
  }
  // resident vectors
  {
    Veccf xg = x.vers_gpu(), yg = Veccf::sur_gpu(n);
    for (int it = 0; it < 20; it++) f->step(xg, yg);
    tsd_amd::synchronise_gpu();
    const int reps = 50;
    auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < reps; it++) f->step(xg, yg);
    tsd_amd::synchronise_gpu();
    const double ms = ms_since(t0) / reps;
    printf("{\"case\": \"filtre_rif step, resident Veccf (FiltreGen::step through the adaptor)\", \"log2n\": %d, \"ms\": %.4f, \"Msamples_s\": %.0f}\n", log2n, ms, gs / ms * 1e3);
    ResidenceGpu garde;
    for (int it = 0; it < 3; it++) { Veccf w = filtrer<cfloat>(Design(h), xg); }
    tsd_amd::synchronise_gpu();
    auto t1 = std::chrono::steady_clock::now();
    for (int it = 0; it < 10; it++) { Veccf w = filtrer<cfloat>(Design(h), xg); }
    tsd_amd::synchronise_gpu();
    const double ms2 = ms_since(t1) / 10;
    printf("{\"case\": \"filtrer() one-shot, resident (creates filter + output per call)\", \"log2n\": %d, \"ms\": %.4f, \"Msamples_s\": %.0f}\n", log2n, ms2, gs / ms2 * 1e3);
  }
  return 0;
}
