// perf_oneshot.cc -- what libtsd's one-shot free functions cost on SMALL host vectors through the mirror (object creation
// included, like libtsd): filtrer (FIR and IIR), rééchan, fft, xcorr.  usage: perf_oneshot [n]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "tsd/tsd-all.hpp"

using namespace tsd;
using namespace tsd::filtrage;
using namespace tsd::fourier;

template <typename F> static double med_us(F f, int reps = 30)
{
  f();
  std::vector<double> t;
  for (int i = 0; i < reps; i++) {
    auto a = std::chrono::steady_clock::now();
    f();
    t.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count());
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main(int argc, char **argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 4096;
  const Vecf x = randn(n);
  const Veccf xc = randcn(n);
  const Vecf h = design_rif_fen(31, "lp", 0.2f);
  const auto hi = design_riia(6, "lp", "butt", 0.2f);
  printf("{\"n\": %d", n);
  printf(", \"filtrer_rif_us\": %.1f", med_us([&] { Vecf y = filtrer<float>(Design(h), x); }));
  printf(", \"filtrer_riia6_us\": %.1f", med_us([&] { Vecf y = filtrer<float>(Design(hi), x); }));
  printf(", \"reechan_160_147_us\": %.1f", med_us([&] { Veccf y = rééchan(xc, 160.0f / 147); }));
  printf(", \"reechan_x4_us\": %.1f", med_us([&] { Veccf y = rééchan(xc, 4.0f); }));
  printf(", \"filtre_reechan_create_us\": %.1f", med_us([&] { auto f = filtre_reechan<cfloat>(160.0f / 147); }));
  printf(", \"itrp_sinc_create_us\": %.1f", med_us([&] { InterpolateurSincConfig c; c.ncoefs = 15; c.nphases = 256; c.fcut = 0.4f; auto it = itrp_sinc<cfloat>(c); }));
  {
    InterpolateurSincConfig c; c.ncoefs = 15; c.nphases = 256; c.fcut = 0.4f;
    auto it = itrp_sinc<cfloat>(c);
    printf(", \"filtre_itrp_create_us\": %.1f", med_us([&] { auto f = filtre_itrp<cfloat>(160.0f / 147, it); }));
    auto f = filtre_reechan<cfloat>(160.0f / 147);
    Veccf y;
    printf(", \"reechan_step_only_us\": %.1f", med_us([&] { f->step(xc, y); }));
  }
  printf(", \"fft_us\": %.1f", med_us([&] { Veccf y = fft(xc); }));
  printf(", \"xcorr_us\": %.1f", med_us([&] { auto r = xcorr(xc, xc, 64); }));
  printf("}\n");
  return 0;
}
