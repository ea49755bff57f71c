// common.hpp -- shared host-side helpers of the C-ABI shim (error reporting, host/device
// pointer staging).  Product code: never includes or links anything under oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <functional>
#include <mutex>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/tsdgpu.h"

namespace tsdgpu {

std::string &last_error_ref();
int set_err(int code, const char *fmt, ...);

#define TSD_HIP(expr)                                                                   \
  do {                                                                                  \
    hipError_t e__ = (expr);                                                            \
    if (e__ != hipSuccess)                                                              \
      return ::tsdgpu::set_err(TSDGPU_ERR_HIP, "%s failed: %s (%s:%d)", #expr,          \
                               hipGetErrorString(e__), __FILE__, __LINE__);             \
  } while (0)

#define TSD_CHECK(cond, ...)                                                            \
  do {                                                                                  \
    if (!(cond)) return ::tsdgpu::set_err(TSDGPU_ERR_INVALID, __VA_ARGS__);             \
  } while (0)

inline size_t dtype_size(int dt) { return dt == TSDGPU_C64 ? 8 : 4; }

// true while `st` records a graph.  A captured launch replays with FROZEN arguments, so the kernels whose work counters are
// never reset (a host-tracked base goes in as a launch argument: fft1m_cols, OlsDyn, RsDyn) take their static partition there.
inline bool stream_is_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}

// true when p is device or managed memory (treated as resident); page-locked / registered host
// memory is host memory here: it is staged, its copies being asynchronous
bool is_device_ptr(const void *p);

// Grow-only device scratch buffer owned by a handle.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes);
  void release();
  template <typename T> T *as() const { return static_cast<T *>(p); }
};

// Presents a (host or device) input as a device pointer; host data is copied to `buf`.
int stage_in(const void *src, size_t bytes, DevBuf &buf, hipStream_t st, const void **dev);
// Gives a device pointer to write the output to: `dst` itself when it is device memory,
// else `buf`.  finish_out() copies back to the host buffer and synchronises when staged.
int stage_out(void *dst, size_t bytes, DevBuf &buf, void **dev, bool *staged);
int finish_out(void *dst, size_t bytes, const void *dev, bool staged, hipStream_t st);

// Large HOST buffers through a same-length streaming operator (FIR, SOS, RII): the vector is cut in
// chunks that are copied in, processed and copied out on three streams, so that H2D of chunk i+1, the
// kernels of chunk i and D2H of chunk i-1 overlap (the operator's state lives on the device and is
// carried from chunk to chunk exactly as between two step() calls).  With page-locked host memory
// (tsdgpu_malloc_host, hipHostRegister) the copies are asynchronous and PCIe runs in both directions
// at once; with pageable memory the runtime stages each copy itself and only the kernels overlap.
// step(dx, dy, count, stream) must enqueue the operator on device pointers.  Returns after y is complete.
constexpr size_t PIPE_MIN_BYTES = (size_t) 8 << 20;      // below this one H2D / kernel / D2H round is as good
constexpr size_t PIPE_CHUNK_BYTES = (size_t) 16 << 20;
int pipelined_host_step(const void *x, void *y, int64_t n, size_t elem_bytes, hipStream_t user_stream,
                        const std::function<int(const void *, void *, int64_t, hipStream_t)> &step);
// variable output length (resampler, integer-rate stages): `step(in, out, cnt, cap, &got, stream)` reports the chunk's
// output count on the host; chunk lengths are multiples of chunk_align; out_cap(cnt) bounds the outputs of cnt inputs
int pipelined_host_step_var(const void *x, int64_t n, size_t esz_in, void *y, size_t esz_out, int64_t *n_out, int64_t chunk_align,
                            hipStream_t user_stream, const std::function<int64_t(int64_t)> &out_cap,
                            const std::function<int(const void *, void *, int64_t, int64_t, int64_t *, hipStream_t)> &step);
bool host_ranges_overlap(const void *a, size_t na, const void *b, size_t nb);
// small device-to-device copy as a kernel launch (see common.hip)
int device_copy_small(void *dst, const void *src, size_t bytes, hipStream_t st);
bool host_pipe_enabled();     // false with TSDGPU_NO_PIPE=1 (A/B switch: whole-vector staging instead)
// Developer / test switches: plan overrides and hand-out thresholds that let the tests put an ALTERNATIVE PRODUCT PATH (the static
// partition of a dynamic kernel, the plan another size would take, the literal recursion ...) under the parity tests at small
// sizes (scripts/check_switches.sh).  One door for all of them: dev_switch("FFT_NO_SMOOTH") is the environment variable
// TSDGPU_FFT_NO_SMOOTH, read at every call (the tests flip them between handles).  Not a tuning surface: measured-and-rejected
// variants are not in the tree (profiles/EXPERIMENTS.md has their numbers); the switches a USER may want are listed in DESIGN.md 6.
const char *dev_switch(const char *name);                 // nullptr when unset
int dev_switch_int(const char *name, int dflt);

// Scratch of a STATELESS entry point (xcorr, welch, delay estimate ...): buffers and plans are borrowed for the call from a
// small free list and go back to it, instead of a hipMalloc / plan construction / hipFree per call (a hipFree also
// synchronises the device).  Ctx needs `int dev`, `size_t octets()` and `void libere()`; prefere(ctx) picks a context that already fits.
template <typename Ctx> struct CtxReserve {
  std::mutex m;
  std::vector<Ctx *> libres;
  size_t garde;
  explicit CtxReserve(size_t g = 8) : garde(g) {}
  template <typename Pref> Ctx *prend(Pref prefere)
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) (void) hipGetLastError();
    {
      std::lock_guard<std::mutex> l(m);
      for (int pass = 0; pass < 2; pass++)
        for (size_t i = 0; i < libres.size(); i++)
          if (libres[i]->dev == dev && (pass == 1 || prefere(*libres[i]))) {
            Ctx *c = libres[i];
            libres.erase(libres.begin() + (long) i);
            return c;
          }
    }
    Ctx *c = new Ctx();
    c->dev = dev;
    return c;
  }
  void rend(Ctx *c)
  {
    // (a context grown by one very large call is not kept: the reserve is for the many ordinary ones)
    if (c->octets() <= ((size_t) 512 << 20)) {
      std::lock_guard<std::mutex> l(m);
      if (libres.size() < garde) {
        libres.push_back(c);
        return;
      }
    }
    c->libere();
    delete c;
  }
};

// Serialises use of a handle's scratch buffers: host threads through the mutex, streams through
// an event (a step on another stream waits for the previous step's work).  libtsd's Spectrum calls
// plan->step from OpenMP threads on one plan (fourier.cc:1244-1252), so plans must tolerate it.
struct StepOrder {
  std::mutex mu;
  hipEvent_t ev = nullptr;
  hipStream_t last = nullptr;
  bool used = false;
  int enter(hipStream_t st);   // call with mu held, before enqueueing
  int leave(hipStream_t st);   // call with mu held, after enqueueing
  void release();
};

}  // namespace tsdgpu
struct tsdgpu_fft;
namespace tsdgpu {
// twiddles W_n^i, i < n/16, of a plan served by the radix-16 Stockham kernel (n = 16 .. 16384), else NULL:
// lets other translation units run s16::transform on the plan's table (ola.hip)
const float2 *fft_s16_twiddles(const tsdgpu_fft *p);
int fft_blu_framed_launch(const tsdgpu_fft *p, const float2 *x, int64_t pas, const float *win, int64_t nseg, float *pw, int64_t cap_rows,
                          int64_t *rows, void *stream);

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace tsdgpu
