// CPU structural test of libtsd_amd/csrc/fft1024_wave.hpp: emulates the 64 lanes of a wave
// phase by phase (each phase for all lanes before the next = what the LDS sync guarantees)
// and checks forward() against a double-precision DFT and inverse(forward(x)) == N*x.
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../libtsd_amd/csrc/fft1024_wave.hpp"
using namespace tsdgpu;
using namespace tsdgpu::w1024;

int main()
{
  const int N = 1024;
  std::vector<cpx> tw1(1024), tw2(1024), lds(LDS_ELEMS);
  fill_twiddles(tw1.data(), tw2.data());
  std::vector<std::complex<double>> x(N), X(N);
  srand(1);
  for (auto &v : x) v = {rand() / (double) RAND_MAX - 0.5, rand() / (double) RAND_MAX - 0.5};
  const double PI = 3.14159265358979323846;
  for (int k = 0; k < N; k++) {
    std::complex<double> s = 0;
    for (int n = 0; n < N; n++) s += x[n] * std::polar(1.0, -2 * PI * (double) ((long) k * n % N) / N);
    X[k] = s;
  }
  cpx v[64][16], t1[64][16], t2[64][16];
  for (int l = 0; l < 64; l++)
    for (int r = 0; r < 16; r++) {
      auto s = x[time_index(l, r)];
      v[l][r] = mk((float) s.real(), (float) s.imag());
      t1[l][r] = tw1[r * 64 + l];
      t2[l][r] = tw2[r * 64 + l];
    }
#define ALL(stmt) for (int l = 0; l < 64; l++) { stmt; }
  // forward (same sequence as w1024::forward)
  ALL(stageA<false>(v[l], t1[l]));  ALL(x1_write_rows(v[l], lds.data(), l));
  ALL(x1_read_cols(v[l], lds.data(), l)); ALL(stageB<false>(v[l], t2[l]));
  ALL(x2_write_j1(v[l], lds.data(), l)); ALL(x2_read_m2(v[l], lds.data(), l));
  ALL(stageC<false>(v[l]));
  double emax = 0, ref = 0;
  for (int l = 0; l < 64; l++)
    for (int r = 0; r < 16; r++) {
      auto e = X[freq_index(l, r)];
      emax = std::max(emax, std::abs(std::complex<double>(v[l][r].x, v[l][r].y) - e));
      ref = std::max(ref, std::abs(e));
    }
  printf("forward: max err %.3e (max |X| %.3e)\n", emax, ref);
  if (emax > 2e-6 * ref * 10) { printf("FAIL forward\n"); return 1; }
  // every frequency index appears exactly once
  std::vector<int> seen(N, 0);
  for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) seen[freq_index(l, r)]++;
  for (int k = 0; k < N; k++) if (seen[k] != 1) { printf("FAIL freq_index not a bijection\n"); return 1; }
  // inverse (same sequence as w1024::inverse)
  ALL(stageC<true>(v[l])); ALL(x2_write_m2(v[l], lds.data(), l)); ALL(x2_read_j1(v[l], lds.data(), l));
  ALL(stageB<true>(v[l], t2[l])); ALL(x1_write_cols(v[l], lds.data(), l)); ALL(x1_read_rows(v[l], lds.data(), l));
  ALL(stageA<true>(v[l], t1[l]));
  double e2 = 0;
  for (int l = 0; l < 64; l++)
    for (int r = 0; r < 16; r++) {
      auto s = x[time_index(l, r)] * (double) N;
      e2 = std::max(e2, std::abs(std::complex<double>(v[l][r].x, v[l][r].y) - s));
    }
  printf("round trip: max err %.3e (scale %d)\n", e2, N);
  if (e2 > 1e-5 * N) { printf("FAIL inverse\n"); return 1; }
  // the same with GENERATED twiddles (PowGen: the table's entries 1, 2, 4, 8 kept, the others their products) -- what the overlap-save FIR
  // runs at three waves per SIMD: accuracy against the double-precision DFT and the round trip
  {
    cpx g[64][16];
    for (int l = 0; l < 64; l++)
      for (int r = 0; r < 16; r++) {
        auto s = x[time_index(l, r)];
        g[l][r] = mk((float) s.real(), (float) s.imag());
      }
#define GEN1(l) PowGen<cpx>{tw1[64 + l], tw1[128 + l], tw1[256 + l], tw1[512 + l]}
#define GEN2(l) PowGen<cpx>{tw2[64 + l], tw2[128 + l], tw2[256 + l], tw2[512 + l]}
    ALL(stageA<false>(g[l], GEN1(l)));  ALL(x1_write_rows(g[l], lds.data(), l));
    ALL(x1_read_cols(g[l], lds.data(), l)); ALL(stageB<false>(g[l], GEN2(l)));
    ALL(x2_write_j1(g[l], lds.data(), l)); ALL(x2_read_m2(g[l], lds.data(), l));
    ALL(stageC<false>(g[l]));
    double eg = 0;
    for (int l = 0; l < 64; l++)
      for (int r = 0; r < 16; r++) eg = std::max(eg, std::abs(std::complex<double>(g[l][r].x, g[l][r].y) - X[freq_index(l, r)]));
    printf("forward, generated twiddles: max err %.3e (table: %.3e)\n", eg, emax);
    if (eg > 3 * emax + 1e-7 * ref) { printf("FAIL generated twiddles lose accuracy\n"); return 1; }
    ALL(stageC<true>(g[l])); ALL(x2_write_m2(g[l], lds.data(), l)); ALL(x2_read_j1(g[l], lds.data(), l));
    ALL(stageB<true>(g[l], GEN2(l))); ALL(x1_write_cols(g[l], lds.data(), l)); ALL(x1_read_rows(g[l], lds.data(), l));
    ALL(stageA<true>(g[l], GEN1(l)));
    double eg2 = 0;
    for (int l = 0; l < 64; l++)
      for (int r = 0; r < 16; r++) eg2 = std::max(eg2, std::abs(std::complex<double>(g[l][r].x, g[l][r].y) - x[time_index(l, r)] * (double) N));
    printf("round trip, generated twiddles: max err %.3e (table: %.3e)\n", eg2, e2);
    if (eg2 > 3 * e2 + 1e-7 * N) { printf("FAIL generated twiddles lose accuracy (round trip)\n"); return 1; }
  }
  // bank-conflict audit of both exchanges (8-byte slots): reads in 32-lane halves over 32
  // slots, writes in 16-lane groups over 16 slots
  auto audit = [&](auto addr, int group, int slots, const char *name) {
    for (int r = 0; r < 16; r++)
      for (int g = 0; g < 64; g += group) {
        std::vector<int> cnt(slots, 0);
        for (int l = g; l < g + group; l++) if (++cnt[addr(l, r) % slots] > 1) { printf("FAIL bank conflict in %s\n", name); exit(1); }
      }
  };
  audit([](int l, int r) { return LDS_ROW * r + l; }, 16, 16, "x1 rows write");
  audit([](int l, int r) { return LDS_ROW * r + l; }, 32, 32, "x1 rows read");
  audit([](int l, int r) { return LDS_ROW * (l >> 2) + (l & 3) + 4 * r; }, 16, 16, "x1 cols write");
  audit([](int l, int r) { return LDS_ROW * (l >> 2) + (l & 3) + 4 * r; }, 32, 32, "x1 cols read");
  audit([](int l, int r) { return LDS_ROW * (l >> 2) + 17 * (l & 3) + r; }, 16, 16, "x2 j1 write");
  audit([](int l, int r) { return LDS_ROW * (l >> 2) + 17 * (l & 3) + r; }, 32, 32, "x2 j1 read");
  audit([](int l, int r) { return LDS_ROW * (l >> 2) + (l & 3) + 17 * (r & 3) + 4 * (r >> 2); }, 16, 16, "x2 m2 write");
  audit([](int l, int r) { return LDS_ROW * (l >> 2) + (l & 3) + 17 * (r & 3) + 4 * (r >> 2); }, 32, 32, "x2 m2 read");
  // two 512-point transforms in one wave (forward_pair512 / inverse_pair512: the windowed overlap-add engine): both sequences against
  // the double-precision DFT, the index maps bijections per sequence, and the round trip
  {
    std::vector<cpx> p1(1024), p2(1024);
    fill_twiddles_pair512(p1.data(), p2.data());
    std::vector<std::complex<double>> xs[2] = {std::vector<std::complex<double>>(512), std::vector<std::complex<double>>(512)}, Xs[2];
    for (int q = 0; q < 2; q++) {
      for (auto &u : xs[q]) u = {rand() / (double) RAND_MAX - 0.5, rand() / (double) RAND_MAX - 0.5};
      Xs[q].resize(512);
      for (int k = 0; k < 512; k++) {
        std::complex<double> a = 0;
        for (int n = 0; n < 512; n++) a += xs[q][n] * std::polar(1.0, -2 * PI * (double) ((long) k * n % 512) / 512);
        Xs[q][k] = a;
      }
    }
    cpx w[64][16], u1[64][16], u2[64][16];
    std::vector<int> seen_t(1024, 0), seen_f(1024, 0);
    for (int l = 0; l < 64; l++)
      for (int r = 0; r < 16; r++) {
        auto a = xs[pair512_seq_of_lane(l)][pair512_time_index(l, r)];
        seen_t[512 * pair512_seq_of_lane(l) + pair512_time_index(l, r)]++;
        seen_f[512 * pair512_seq_of_reg(r) + pair512_freq_index(l, r)]++;
        w[l][r] = mk((float) a.real(), (float) a.imag());
        u1[l][r] = p1[r * 64 + l];
        u2[l][r] = p2[r * 64 + l];
      }
    for (int i = 0; i < 1024; i++)
      if (seen_t[i] != 1 || seen_f[i] != 1) { printf("FAIL pair512 index maps are not bijections\n"); return 1; }
    ALL(stageA<false>(w[l], u1[l]));  ALL(x1_write_rows(w[l], lds.data(), l));
    ALL(x1_read_cols(w[l], lds.data(), l)); ALL(stageB<false>(w[l], u2[l]));
    ALL(x2_write_j1(w[l], lds.data(), l)); ALL(x2_read_m2(w[l], lds.data(), l));
    ALL(stageC2(w[l]));
    double ep = 0, rp = 0;
    for (int l = 0; l < 64; l++)
      for (int r = 0; r < 16; r++) {
        auto e = Xs[pair512_seq_of_reg(r)][pair512_freq_index(l, r)];
        ep = std::max(ep, std::abs(std::complex<double>(w[l][r].x, w[l][r].y) - e));
        rp = std::max(rp, std::abs(e));
      }
    printf("pair512 forward: max err %.3e (max |X| %.3e)\n", ep, rp);
    if (ep > 2e-5 * rp) { printf("FAIL pair512 forward\n"); return 1; }
    ALL(stageC2(w[l])); ALL(x2_write_m2(w[l], lds.data(), l)); ALL(x2_read_j1(w[l], lds.data(), l));
    ALL(stageB<true>(w[l], u2[l])); ALL(x1_write_cols(w[l], lds.data(), l)); ALL(x1_read_rows(w[l], lds.data(), l));
    ALL(stageA<true>(w[l], u1[l]));
    double er = 0;
    for (int l = 0; l < 64; l++)
      for (int r = 0; r < 16; r++) {
        auto a = xs[pair512_seq_of_lane(l)][pair512_time_index(l, r)] * 512.0;
        er = std::max(er, std::abs(std::complex<double>(w[l][r].x, w[l][r].y) - a));
      }
    printf("pair512 round trip: max err %.3e (scale 512)\n", er);
    if (er > 1e-5 * 512) { printf("FAIL pair512 inverse\n"); return 1; }
  }
  printf("OK\n");
  return 0;
}
