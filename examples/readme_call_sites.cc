// examples/readme_call_sites.cc -- the filtering parts of libtsd's own README examples (README.md:28-44 English API,
// :57-73 French API), statement for statement (the plotting lines left out: the view layer is outside the path),
// compiled against the mirror headers and run on the MI355X: "existing call sites compile unchanged".
//   g++ -std=c++20 -Ilibtsd_amd/host/include examples/readme_call_sites.cc -Llibtsd_amd/lib -ltsd_host -ltsdgpu
#include <cstdio>
#include "dsp/dsp-all.hpp"
#include "tsd/tsd-all.hpp"

static int exemple_anglais()
{
  // Example 2: filter design
  let h = design_fir_wnd(31, "lp", 0.25);

  let n = 500;
  // (the README writes `y = filter(h, x)`: with the umbrella header's `using namespace dsp; using namespace dsp::filter;`
  //  that unqualified name designates both the namespace dsp::filter and the function dsp::filter::filter -- ambiguous for
  //  g++ against libtsd's own headers as well; qualified here)
  let x = sigcos(0.01, n) + 0.1 * randn(n),
      y = dsp::filter::filter(h, x);
  // what the figure of the README shows: the tone comes through, the noise is cut
  double bruit_x = 0, bruit_y = 0;
  for (int i = 100; i < n; i++) {
    const double ref = std::cos(2 * π * 0.01 * (i - 15));      // the 31-tap filter delays by 15 samples
    bruit_x += (x(i) - std::cos(2 * π * 0.01 * i)) * (x(i) - std::cos(2 * π * 0.01 * i));
    bruit_y += (y(i) - ref) * (y(i) - ref);
  }
  std::printf("english API: %d taps, %d samples, residual noise power %.4f -> %.4f\n", h.rows(), y.rows(), bruit_x / (n - 100), bruit_y / (n - 100));
  // resampling through an interpolator, English names (dsp/filter.hpp:1755-1805,1910)
  let z = dsp::filter::filter_itrp<float>(1.5f, dsp::filter::itrp_sinc<float>(15, 0.4f, "hn"))->step(y);
  std::printf("english API: filter_itrp(1.5, itrp_sinc(15, 0.4, \"hn\")): %d -> %d samples\n", y.rows(), z.rows());
  if (std::abs(z.rows() - 750) > 2) return 1;
  return (h.rows() == 31 && y.rows() == n && bruit_y < 0.7 * bruit_x) ? 0 : 1;
}

static int exemple_français()
{
  // Exemple 2 : conception de filtre
  soit h = design_rif_fen(31, "pb", 0.25);

  soit n = 500;
  soit x = sigcos(0.01, n) + 0.1 * randn(n),
       y = filtrer(h, x);
  std::printf("API française : %d coefficients, %d échantillons\n", h.rows(), y.rows());
  retourne (h.rows() == 31 et y.rows() == n) ? 0 : 1;
}

int main()
{
  const int rc = exemple_anglais() + exemple_français();
  std::printf(rc ? "README CALL SITES FAILED\n" : "README CALL SITES OK\n");
  return rc;
}
