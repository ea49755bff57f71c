/* The drop-in boundary used from plain C99 (no C++, no Python): a 31-tap FIR on a host buffer,
 * a batched FFT, a resampler -- each checked against a few lines of C.
 *   gcc -std=c99 -Iinclude examples/fir_from_c.c -Llibtsd_amd/lib -ltsdgpu -lm -Wl,-rpath,$PWD/libtsd_amd/lib
 * Exit status 0 = every check passed; 2 = no GPU (the library refuses to run: no CPU fallback). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "tsdgpu.h"

#define N 4096
#define K 31

static int fail(const char *what)
{
  fprintf(stderr, "%s: %s\n", what, tsdgpu_last_error());
  return 1;
}

int main(void)
{
  if (tsdgpu_device_count() < 1) {
    fprintf(stderr, "no GPU: %s\n", tsdgpu_last_error());
    return 2;
  }
  static float x[N], y[N], h[K];
  for (int i = 0; i < N; i++) x[i] = sinf(0.01f * (float) i) + 0.1f * (float) ((i * 7919) % 13 - 6);
  for (int k = 0; k < K; k++) h[k] = 1.0f / K;

  /* filtre_rif<float,float>(h)->step(x) */
  tsdgpu_fir *f = NULL;
  if (tsdgpu_fir_create(&f, TSDGPU_F32, TSDGPU_F32, h, K, TSDGPU_FIR_AUTO)) return fail("fir_create");
  if (tsdgpu_fir_step(f, x, y, N / 2, NULL)) return fail("fir_step");            /* two calls: the history carries over */
  if (tsdgpu_fir_step(f, x + N / 2, y + N / 2, N / 2, NULL)) return fail("fir_step");
  tsdgpu_fir_destroy(f);
  double emax = 0;
  for (int n = 0; n < N; n++) {
    float acc = 0;
    for (int k = K - 1; k >= 0; k--)                                             /* oldest sample first, filtre-rt.cc:98-104 */
      if (n - k >= 0) acc += h[k] * x[n - k];
    if (fabs(acc - y[n]) > emax) emax = fabs(acc - y[n]);
  }
  printf("fir: max error %.3g\n", emax);
  if (emax > 1e-5) return 1;

  /* fft(x): 8 transforms of 512 points, unitary scaling; Parseval per transform */
  static float z[N * 2], Z[N * 2];
  for (int i = 0; i < N; i++) { z[2 * i] = x[i]; z[2 * i + 1] = -x[N - 1 - i]; }
  tsdgpu_fft *p = NULL;
  if (tsdgpu_fft_create(&p, 512, 8)) return fail("fft_create");
  if (tsdgpu_fft_step(p, z, Z, 8, 1, NULL)) return fail("fft_step");
  tsdgpu_fft_destroy(p);
  for (int b = 0; b < 8; b++) {
    double e1 = 0, e2 = 0;
    for (int i = 0; i < 1024; i++) { e1 += (double) z[1024 * b + i] * z[1024 * b + i]; e2 += (double) Z[1024 * b + i] * Z[1024 * b + i]; }
    if (fabs(e2 / e1 - 1) > 1e-5) { fprintf(stderr, "fft: Parseval off by %.3g in transform %d\n", e2 / e1 - 1, b); return 1; }
  }
  printf("fft: Parseval holds on 8 x 512 points\n");

  /* filtre_itrp(1.25, itrp_lineaire): a ramp stays a ramp */
  tsdgpu_resampler *r = NULL;
  if (tsdgpu_resampler_create_analytic(&r, TSDGPU_F32, 1.25f, TSDGPU_ITRP_LINEAR, 1)) return fail("resampler_create");
  static float ramp[N], out[2 * N];
  for (int i = 0; i < N; i++) ramp[i] = (float) i;
  int64_t got = 0;
  if (tsdgpu_resampler_step(r, ramp, N, out, 2 * N, &got, NULL)) return fail("resampler_step");
  tsdgpu_resampler_destroy(r);
  double dmax = 0;
  for (int64_t j = 2; j < got; j++) {
    const double d = fabs(out[j] - ((double) j / 1.25 - 1.0));
    if (d > dmax) dmax = d;
  }
  printf("resampler: %lld outputs, ramp error %.3g\n", (long long) got, dmax);
  if (got < (int64_t) (N * 1.25) - 2 || got > (int64_t) (N * 1.25) + 2 || dmax > 2e-3) return 1;
  printf("C ABI example OK\n");
  return 0;
}
