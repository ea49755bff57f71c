// test_graph_capture.hip -- a streaming step of fixed size captured ONCE into a hipGraph and replayed per
// block gives the stream a sequence of ordinary steps gives, and costs a few
// microseconds per block instead of several launches' worth.  FIR (direct and overlap-save) and SOS.
//   usage: test_graph_capture            (needs a GPU; prints GRAPH CAPTURE OK)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "tsdgpu.h"

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(r_), __FILE__, __LINE__); return 2; } } while (0)
#define TS(e) do { if (e) { printf("tsdgpu error: %s (%s:%d)\n", tsdgpu_last_error(), __FILE__, __LINE__); return 2; } } while (0)

typedef std::complex<float> cf;

template <typename Step> static int run_case(const char *name, size_t esz, int n, int nblk, Step make)
{
  // the stream: nblk blocks of n samples
  std::vector<float> hx((size_t) n * nblk * esz / 4);
  uint32_t s = 1u;
  for (auto &v : hx) { s = s * 1664525u + 1013904223u; v = (float) ((int32_t) s) * (1.0f / 2147483648.0f); }
  char *dx = nullptr, *dy1 = nullptr, *dy2 = nullptr, *bx = nullptr, *by = nullptr;
  const size_t bytes = (size_t) n * nblk * esz, bb = (size_t) n * esz;
  CK(hipMalloc(&dx, bytes)); CK(hipMalloc(&dy1, bytes)); CK(hipMalloc(&dy2, bytes)); CK(hipMalloc(&bx, bb)); CK(hipMalloc(&by, bb));
  CK(hipMemcpy(dx, hx.data(), bytes, hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  // (1) ordinary steps, block after block
  void *h1 = make(0);
  if (!h1) return 2;
  auto stepf = (int (*)(void *, const void *, void *, int64_t, void *)) make(-1);
  for (int b = 0; b < nblk; b++) TS(stepf(h1, dx + b * bb, dy1 + b * bb, n, st));
  CK(hipStreamSynchronize(st));
  auto t0 = std::chrono::steady_clock::now();
  const int reps = 2000;
  for (int r = 0; r < reps; r++) TS(stepf(h1, bx, by, n, st));
  CK(hipStreamSynchronize(st));
  const double us_plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  // (2) one step captured, replayed per block
  void *h2 = make(1);                          // capturable handle, already warmed on a scratch block and reset
  if (!h2) return 2;
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  TS(stepf(h2, bx, by, n, st));
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int b = 0; b < nblk; b++) {
    CK(hipMemcpyAsync(bx, dx + b * bb, bb, hipMemcpyDeviceToDevice, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipMemcpyAsync(dy2 + b * bb, by, bb, hipMemcpyDeviceToDevice, st));
  }
  CK(hipStreamSynchronize(st));
  std::vector<float> y1(hx.size()), y2(hx.size());
  CK(hipMemcpy(y1.data(), dy1, bytes, hipMemcpyDeviceToHost));
  CK(hipMemcpy(y2.data(), dy2, bytes, hipMemcpyDeviceToHost));
  size_t diff = 0;
  double energy = 0;
  for (size_t i = 0; i < y1.size(); i++) { diff += y1[i] != y2[i]; energy += (double) y1[i] * y1[i]; }
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  const double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  printf("{\"case\": \"%s\", \"n\": %d, \"blocks\": %d, \"us_per_step_plain\": %.2f, \"us_per_step_graph\": %.2f, \"mismatches\": %zu}\n", name, n, nblk,
         us_plain, us_graph, diff);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  CK(hipFree(dx)); CK(hipFree(dy1)); CK(hipFree(dy2)); CK(hipFree(bx)); CK(hipFree(by));
  CK(hipStreamDestroy(st));
  if (diff != 0 || !(energy > 0)) { printf("FAIL %s: %zu samples differ between the graph replay and the plain steps\n", name, diff); return 1; }
  return 0;
}

static std::vector<float> taps(int K)
{
  std::vector<float> h((size_t) K);
  for (int i = 0; i < K; i++) h[i] = (float) ((0.5 - 0.5 * std::cos(2 * M_PI * (i + 0.5) / K)) / (0.5 * K));
  return h;
}

int main()
{
  if (tsdgpu_device_count() < 1) { printf("no GPU\n"); return 2; }
  int rc = 0;
  const int n = 4096, nblk = 16;
  // FIR: 31 real taps on real data (direct kernel) and 127 taps on complex data (overlap-save)
  for (int which = 0; which < 2; which++) {
    const int K = which == 0 ? 31 : 127, dt = which == 0 ? TSDGPU_F32 : TSDGPU_C64;
    const std::vector<float> h = taps(K);
    auto make = [&](int mode) -> void * {
      if (mode < 0) return (void *) +[](void *f, const void *x, void *y, int64_t m, void *s) { return tsdgpu_fir_step((tsdgpu_fir *) f, x, y, m, s); };
      tsdgpu_fir *f = nullptr;
      if (tsdgpu_fir_create(&f, dt, TSDGPU_F32, h.data(), K, TSDGPU_FIR_AUTO)) return nullptr;
      if (mode == 1) {
        if (tsdgpu_fir_set_capturable(f, 1)) return nullptr;
        void *scratch = nullptr;
        if (hipMalloc(&scratch, (size_t) n * 8) != hipSuccess || hipMemset(scratch, 0, (size_t) n * 8) != hipSuccess) return nullptr;
        if (tsdgpu_fir_step(f, scratch, scratch, n, nullptr) || tsdgpu_fir_reset(f)) return nullptr;   // in place too: every scratch buffer exists now
        (void) hipDeviceSynchronize();
        (void) hipFree(scratch);
      }
      return f;
    };
    rc |= run_case(which == 0 ? "fir 31 taps, real, 4096-sample steps" : "fir 127 taps, complex, 4096-sample steps", which == 0 ? 4 : 8, n, nblk, make);
  }
  // SOS: 3 sections on real data
  {
    const float co[15] = {1, 2, 1, -0.5f, 0.3f, 1, 2, 1, -0.2f, 0.5f, 1, -1, 0.2f, 0.1f, 0.05f};
    auto make = [&](int mode) -> void * {
      if (mode < 0) return (void *) +[](void *f, const void *x, void *y, int64_t m, void *s) { return tsdgpu_sos_step((tsdgpu_sos *) f, x, y, m, s); };
      tsdgpu_sos *f = nullptr;
      if (tsdgpu_sos_create(&f, TSDGPU_F32, co, 3, 0.1f, nullptr, 2)) return nullptr;
      if (mode == 1) {
        if (tsdgpu_sos_set_capturable(f, 1)) return nullptr;
        void *scratch = nullptr;
        if (hipMalloc(&scratch, (size_t) n * 4) != hipSuccess || hipMemset(scratch, 0, (size_t) n * 4) != hipSuccess) return nullptr;
        if (tsdgpu_sos_step(f, scratch, scratch, n, nullptr) || tsdgpu_sos_reset(f)) return nullptr;
        (void) hipDeviceSynchronize();
        (void) hipFree(scratch);
      }
      return f;
    };
    rc |= run_case("sos 3 sections, real, 4096-sample steps", 4, n, nblk, make);
  }
  // FFT 2^20 x 32: the step is stateless and graph-replayable; its kernels hand their tiles out through a
  // never-reset counter whose base is a launch argument, so while the stream records they must take the static partition --
  // a replay with a frozen base would find the counter spent and write nothing.  Two replays on different inputs.
  // (round 4: the same for the three-pass plan of 2^24 -- two dynamic column passes around a plane pass -- and for the persistent
  // wave-level Bluestein kernels of n = 1000 (fused 8 x 125) and n = 1001, whose grid comes from an occupancy query)
  const int fft_cases[4][2] = {{1 << 20, 32}, {1 << 24, 2}, {1000, 30000}, {1001, 20000}};
  for (int fc = 0; fc < 4; fc++) {
    const int nfft = fft_cases[fc][0], batch = fft_cases[fc][1];
    const size_t tot = (size_t) nfft * batch;
    std::vector<float> hx(2 * tot);
    uint32_t s = 7u;
    for (auto &v : hx) { s = s * 1664525u + 1013904223u; v = (float) ((int32_t) s) * (1.0f / 2147483648.0f); }
    char *dx = nullptr, *dy1 = nullptr, *dy2 = nullptr;
    CK(hipMalloc(&dx, tot * 8)); CK(hipMalloc(&dy1, tot * 8)); CK(hipMalloc(&dy2, tot * 8));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    tsdgpu_fft *p = nullptr;
    TS(tsdgpu_fft_create(&p, nfft, batch));
    CK(hipMemcpy(dx, hx.data(), tot * 8, hipMemcpyHostToDevice));
    TS(tsdgpu_fft_step(p, dx, dy2, batch, 1, st));              // (scratch allocated, counters advanced)
    CK(hipStreamSynchronize(st));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    TS(tsdgpu_fft_step(p, dx, dy2, batch, 1, st));
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    size_t diff = 0;
    for (int rep = 0; rep < 2; rep++) {
      for (auto &v : hx) { s = s * 1664525u + 1013904223u; v = (float) ((int32_t) s) * (1.0f / 2147483648.0f); }
      CK(hipMemcpy(dx, hx.data(), tot * 8, hipMemcpyHostToDevice));
      TS(tsdgpu_fft_step(p, dx, dy1, batch, 1, st));            // ordinary step (dynamic hand-out)
      CK(hipMemsetAsync(dy2, 0xff, tot * 8, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      std::vector<float> y1(2 * tot), y2(2 * tot);
      CK(hipMemcpy(y1.data(), dy1, tot * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(y2.data(), dy2, tot * 8, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < y1.size(); i++) diff += std::memcmp(&y1[i], &y2[i], 4) != 0;
    }
    char label[64];
    snprintf(label, sizeof label, "fft n = %d x %d", nfft, batch);
    printf("%-44s replayed twice: %zu differing floats\n", label, diff);
    rc |= diff != 0;
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    tsdgpu_fft_destroy(p);
    CK(hipFree(dx)); CK(hipFree(dy1)); CK(hipFree(dy2));
  }
  printf(rc ? "GRAPH CAPTURE FAILED\n" : "GRAPH CAPTURE OK\n");
  return rc;
}
