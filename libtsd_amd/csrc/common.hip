// common.hip -- error reporting and host/device staging for the C-ABI shim.
#include "common.hpp"
#include "per_device.hpp"
#include <map>
#include <vector>
#include <string>
#include <mutex>
#include <thread>
#include <deque>
#include <condition_variable>
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace tsdgpu {

std::string &last_error_ref()
{
  static thread_local std::string e;
  return e;
}

int set_err(int code, const char *fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return code;
}

bool is_device_ptr(const void *p)
{
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void) hipGetLastError();  // plain host memory: clear the sticky error
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged ||
         attr.type == hipMemoryTypeUnified;
}

int DevBuf::reserve(size_t bytes)
{
  if (bytes <= cap) return TSDGPU_OK;
  if (p) {
    TSD_HIP(hipFree(p));
    p = nullptr;
    cap = 0;
  }
  size_t want = bytes + bytes / 8 + 256;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    p = nullptr;
    return set_err(TSDGPU_ERR_ALLOC, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  }
  cap = want;
  return TSDGPU_OK;
}

void DevBuf::release()
{
  if (p) (void) hipFree(p);
  p = nullptr;
  cap = 0;
}

int StepOrder::enter(hipStream_t st)
{
  if (used && st != last) TSD_HIP(hipStreamWaitEvent(st, ev, 0));
  return TSDGPU_OK;
}

int StepOrder::leave(hipStream_t st)
{
  if (!ev) TSD_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  TSD_HIP(hipEventRecord(ev, st));
  last = st;
  used = true;
  return TSDGPU_OK;
}

void StepOrder::release()
{
  if (ev) (void) hipEventDestroy(ev);
  ev = nullptr;
  used = false;
}

// Small HOST buffers go through page-locked bounce buffers of the calling thread: a copy from / to pageable user memory
// makes the runtime stage it and block the caller (measured, scripts/ubench/small_copy.hip: H2D + kernel + D2H + sync on
// 16 KiB 30.6 us pageable against 20.5 us with a memcpy into / out of page-locked memory around the same DMA copies).
namespace {
constexpr size_t BOUNCE_MAX = (size_t) 20 << 10;       // measured cross-over (31-tap FIR step: 16 KiB 31 against 39 us, 64 KiB 50 against 46 us)
constexpr int BOUNCE_SLOTS = 4;                        // inputs of one call (x and y of xcorr, ...) do not wait for each other
struct Bounce {
  char *in[BOUNCE_SLOTS] = {nullptr, nullptr, nullptr, nullptr}, *out = nullptr;
  hipEvent_t ev[BOUNCE_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
  hipStream_t st[BOUNCE_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
  bool pending[BOUNCE_SLOTS] = {false, false, false, false};
  int next = 0;
  int dev = 0;        // the device its events belong to (per_device.hpp: never used under another current device)
};
// Blocks are never given back to the runtime: a thread borrows one PER DEVICE it works on for its lifetime and returns
// them to the per-device lists when it ends (no HIP call at thread or process exit; the next borrower waits on whatever
// the previous one left pending), so the page-locked memory held is 100 KiB x the largest number of (thread, device)
// pairs that ever staged small buffers at the same time.
Bounce *bounce_make(int)
{
  static const bool off = getenv("TSDGPU_NO_BOUNCE") != nullptr;
  if (off) return nullptr;
  char *blk = nullptr;
  if (hipHostMalloc((void **) &blk, BOUNCE_MAX * (BOUNCE_SLOTS + 1), hipHostMallocDefault) != hipSuccess) {
    (void) hipGetLastError();
    return nullptr;
  }
  Bounce *b = new Bounce();
  for (int i = 0; i < BOUNCE_SLOTS; i++) {
    b->in[i] = blk + (size_t) i * BOUNCE_MAX;
    if (hipEventCreateWithFlags(&b->ev[i], hipEventDisableTiming) != hipSuccess) {      // (created on the CURRENT device = the key)
      (void) hipGetLastError();
      return nullptr;       // (a box that cannot create an event has bigger problems: the block is abandoned)
    }
  }
  b->out = blk + (size_t) BOUNCE_SLOTS * BOUNCE_MAX;
  return b;
}
PerDevicePool<Bounce> &bounce_pool()
{
  static PerDevicePool<Bounce> *p = new PerDevicePool<Bounce>();      // never destroyed: threads may end after main() has returned
  return *p;
}
Bounce *bounce()
{
  static thread_local PerDeviceHeld<Bounce> held;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void) hipGetLastError();
    return nullptr;
  }
  return held.get(bounce_pool(), dev, bounce_make);
}
}  // namespace

int stage_in(const void *src, size_t bytes, DevBuf &buf, hipStream_t st, const void **dev)
{
  if (bytes == 0 || is_device_ptr(src)) {
    *dev = src;
    return TSDGPU_OK;
  }
  int rc = buf.reserve(bytes);
  if (rc) return rc;
  Bounce *Bp = bytes <= BOUNCE_MAX ? bounce() : nullptr;
  if (Bp) {
    Bounce &B = *Bp;
    const int k = B.next;
    B.next = (k + 1) % BOUNCE_SLOTS;
    if (B.pending[k]) TSD_HIP(hipEventSynchronize(B.ev[k]));     // the DMA engine has read the slot (immediate after a synchronised call)
    std::memcpy(B.in[k], src, bytes);
    TSD_HIP(hipMemcpyAsync(buf.p, B.in[k], bytes, hipMemcpyHostToDevice, st));
    TSD_HIP(hipEventRecord(B.ev[k], st));
    B.pending[k] = true;
    B.st[k] = st;
    *dev = buf.p;
    return TSDGPU_OK;
  }
  TSD_HIP(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, st));
  *dev = buf.p;
  return TSDGPU_OK;
}

int stage_out(void *dst, size_t bytes, DevBuf &buf, void **dev, bool *staged)
{
  if (bytes == 0 || is_device_ptr(dst)) {
    *dev = dst;
    *staged = false;
    return TSDGPU_OK;
  }
  int rc = buf.reserve(bytes);
  if (rc) return rc;
  *dev = buf.p;
  *staged = true;
  return TSDGPU_OK;
}

int finish_out(void *dst, size_t bytes, const void *dev, bool staged, hipStream_t st)
{
  if (!staged || bytes == 0) return TSDGPU_OK;
  Bounce *Bp = bytes <= BOUNCE_MAX ? bounce() : nullptr;
  if (Bp) {
    Bounce &B = *Bp;
    TSD_HIP(hipMemcpyAsync(B.out, dev, bytes, hipMemcpyDeviceToHost, st));
    TSD_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < BOUNCE_SLOTS; k++)
      if (B.st[k] == st) B.pending[k] = false;                    // everything queued on st before has run
    std::memcpy(dst, B.out, bytes);
    return TSDGPU_OK;
  }
  TSD_HIP(hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, st));
  TSD_HIP(hipStreamSynchronize(st));
  return TSDGPU_OK;
}

__global__ void copy_words_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// A few hundred bytes from one device buffer to another (histories, carried blocks, stream states): as a kernel.  A
// device-to-device hipMemcpyAsync of that size spends 15-40 us in the runtime (measured on the OLA engine's buffering
// calls: 52 -> 12 us per call).  Larger or odd-sized copies go to the runtime.
int device_copy_small(void *dst, const void *src, size_t bytes, hipStream_t st)
{
  if (bytes == 0) return TSDGPU_OK;
  if (bytes <= (256u << 10) && bytes % 4 == 0 && ((uintptr_t) dst & 3) == 0 && ((uintptr_t) src & 3) == 0) {
    const int n = (int) (bytes / 4);
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, (const uint32_t *) src, (uint32_t *) dst, n);
    TSD_HIP(hipGetLastError());
    return TSDGPU_OK;
  }
  TSD_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
  return TSDGPU_OK;
}

__global__ void zero_imag_kernel(float2 *__restrict__ y, int64_t n)
{
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i].y = 0.f;
}

namespace {
struct HostPipe {
  hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;
  hipEvent_t e_in[2] = {nullptr, nullptr}, e_k[2] = {nullptr, nullptr}, e_out[2] = {nullptr, nullptr}, e_user = nullptr;
  DevBuf bin[2], bout[2];
  bool ready = false;
  int init()
  {
    if (ready) return TSDGPU_OK;
    TSD_HIP(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
    TSD_HIP(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
    TSD_HIP(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
      TSD_HIP(hipEventCreateWithFlags(&e_in[i], hipEventDisableTiming));
      TSD_HIP(hipEventCreateWithFlags(&e_k[i], hipEventDisableTiming));
      TSD_HIP(hipEventCreateWithFlags(&e_out[i], hipEventDisableTiming));
    }
    TSD_HIP(hipEventCreateWithFlags(&e_user, hipEventDisableTiming));
    ready = true;
    return TSDGPU_OK;
  }
};
// Pipes are borrowed for the duration of ONE call (a pipelined step returns only when everything has run) and go back to
// a per-device free list: concurrent calls get a pipe each, a thread that ends leaves nothing behind.  Never freed.
struct HostPipePool {
  std::mutex m;
  std::map<int, std::vector<HostPipe *>> libres;
  HostPipe *prend(int dev)
  {
    {
      std::lock_guard<std::mutex> l(m);
      auto &v = libres[dev];
      if (!v.empty()) {
        HostPipe *p = v.back();
        v.pop_back();
        return p;
      }
    }
    return new HostPipe();
  }
  void rend(int dev, HostPipe *p)
  {
    std::lock_guard<std::mutex> l(m);
    libres[dev].push_back(p);
  }
};
HostPipePool &host_pipe_pool()
{
  static HostPipePool *pool = new HostPipePool();
  return *pool;
}
struct HostPipeRef {
  int dev = 0;
  HostPipe *p = nullptr;
  HostPipeRef()
  {
    if (hipGetDevice(&dev) != hipSuccess) (void) hipGetLastError();
    p = host_pipe_pool().prend(dev);      // (streams belong to the device that is current when they are created)
  }
  ~HostPipeRef() { host_pipe_pool().rend(dev, p); }
  HostPipeRef(const HostPipeRef &) = delete;
  HostPipeRef &operator=(const HostPipeRef &) = delete;
};
}  // namespace

static int hip_rc(hipError_t e, const char *what)
{
  if (e == hipSuccess) return TSDGPU_OK;
  (void) hipGetLastError();
  return set_err(TSDGPU_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

// true for plain (malloc'd, not registered) host memory: copies from / to it block the calling thread
static bool is_pageable_host(const void *p)
{
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void) hipGetLastError();
    return true;
  }
  return attr.type == hipMemoryTypeUnregistered;
}

int pipelined_host_step(const void *x, void *y, int64_t n, size_t esz, hipStream_t user,
                        const std::function<int(const void *, void *, int64_t, hipStream_t)> &step)
{
  // same-length operator: every chunk produces what it consumed (y may be x: a chunk is written after it was read)
  return pipelined_host_step_var(
      x, n, esz, y, esz, nullptr, 1, user, [](int64_t c) { return c; },
      [&step](const void *cx, void *cy, int64_t cnt, int64_t, int64_t *got, hipStream_t q) {
        *got = cnt;
        return step(cx, cy, cnt, q);
      });
}

// One pipeline for both kinds of operator.  `step(in, out, cnt, cap, &got, stream)` enqueues the operator on device
// pointers and reports the chunk's output count on the host; the chunks' outputs are laid one behind the other.
//  * page-locked host memory: the copies are asynchronous, one thread enqueues everything on three streams;
//  * pageable host memory (what a libtsd Tab holds): a copy blocks its caller until the runtime has staged it, so the
//    copies OUT run on a second host thread -- PCIe then carries both directions at once here too.
int pipelined_host_step_var(const void *x, int64_t n, size_t esz_in, void *y, size_t esz_out, int64_t *n_out, int64_t chunk_align,
                            hipStream_t user, const std::function<int64_t(int64_t)> &out_cap,
                            const std::function<int(const void *, void *, int64_t, int64_t, int64_t *, hipStream_t)> &step)
{
  HostPipeRef ref;
  HostPipe &P = *ref.p;
  int rc = P.init();
  if (rc) return rc;
  static const size_t CHUNK = getenv("TSDGPU_PIPE_CHUNK_MB") ? (size_t) atoi(getenv("TSDGPU_PIPE_CHUNK_MB")) << 20 : PIPE_CHUNK_BYTES;
  static const bool one_thread = getenv("TSDGPU_PIPE_ONE_THREAD") != nullptr;
  const int64_t al = std::max<int64_t>(1, chunk_align);
  const int64_t per = std::max<int64_t>(al, std::max<int64_t>(1024, (int64_t) (CHUNK / esz_in)) / al * al);
  const int64_t nchunks = (n + per - 1) / per;
  const int64_t cap = std::max<int64_t>(out_cap(std::min(per, n)), 1);
  for (int b = 0; b < 2; b++) {
    rc = P.bin[b].reserve((size_t) std::min(per, n) * esz_in);
    if (!rc) rc = P.bout[b].reserve((size_t) cap * esz_out);
    if (rc) return rc;
  }
  // work already queued on the caller's stream (e.g. a history upload) comes first
  TSD_HIP(hipEventRecord(P.e_user, user));
  TSD_HIP(hipStreamWaitEvent(P.s_in, P.e_user, 0));
  TSD_HIP(hipStreamWaitEvent(P.s_k, P.e_user, 0));
  const char *xs = (const char *) x;
  char *ys = (char *) y;
  const bool two_threads = !one_thread && nchunks >= 2 && (is_pageable_host(x) || is_pageable_host(y));

  // ---- the copies out, on their own thread when they block -------------------------------------------------------
  struct Sortie { int b; int64_t off, got; };
  std::mutex m;
  std::condition_variable cv;
  std::deque<Sortie> file;
  int64_t sorties_faites = 0;          // chunks whose outputs are on the host
  bool fini = false;
  int rc_out = TSDGPU_OK;
  std::string msg_out;
  int dev = 0;
  (void) hipGetDevice(&dev);
  auto copie_sortie = [&](const Sortie &o) -> int {
    TSD_HIP(hipStreamWaitEvent(P.s_out, P.e_k[o.b], 0));
    if (o.got > 0) TSD_HIP(hipMemcpyAsync(ys + (size_t) o.off * esz_out, P.bout[o.b].p, (size_t) o.got * esz_out, hipMemcpyDeviceToHost, P.s_out));
    TSD_HIP(hipEventRecord(P.e_out[o.b], P.s_out));
    return TSDGPU_OK;
  };
  std::thread fil;
  if (two_threads)
    fil = std::thread([&] {
      (void) hipSetDevice(dev);
      for (;;) {
        Sortie o;
        {
          std::unique_lock<std::mutex> l(m);
          cv.wait(l, [&] { return fini || !file.empty(); });
          if (file.empty()) return;
          o = file.front();
          file.pop_front();
        }
        int r = copie_sortie(o);
        if (!r && hipStreamSynchronize(P.s_out) != hipSuccess) r = set_err(TSDGPU_ERR_HIP, "host pipeline: copy out failed");
        std::lock_guard<std::mutex> l(m);
        if (r && !rc_out) {
          rc_out = r;
          msg_out = last_error_ref();
        }
        sorties_faites++;
        cv.notify_all();
      }
    });
  auto arrete_fil = [&] {
    if (!two_threads) return;
    {
      std::lock_guard<std::mutex> l(m);
      fini = true;
    }
    cv.notify_all();
    fil.join();
  };

  int64_t produced = 0;
  for (int64_t c = 0; c < nchunks && !rc; c++) {
    const int b = (int) (c & 1);
    const int64_t off = c * per, cnt = std::min(per, n - off);
    if (two_threads) {
      if (c >= 2) {
        // chunk c-2 is on the host: its kernels have read bin[b] and its copy has read bout[b]
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return sorties_faites >= c - 1 || rc_out; });
        if (rc_out) break;
      }
    } else if (c >= 2) {
      rc = hip_rc(hipStreamWaitEvent(P.s_in, P.e_k[b], 0), "host pipeline");          // chunk c-2's kernels have read bin[b]
      if (!rc) rc = hip_rc(hipStreamWaitEvent(P.s_k, P.e_out[b], 0), "host pipeline");  // chunk c-2's D2H has read bout[b]
      if (rc) break;
    }
    rc = hip_rc(hipMemcpyAsync(P.bin[b].p, xs + (size_t) off * esz_in, (size_t) cnt * esz_in, hipMemcpyHostToDevice, P.s_in), "host pipeline: copy in");
    if (!rc) rc = hip_rc(hipEventRecord(P.e_in[b], P.s_in), "host pipeline");
    if (!rc) rc = hip_rc(hipStreamWaitEvent(P.s_k, P.e_in[b], 0), "host pipeline");
    if (rc) break;
    int64_t got = 0;
    rc = step(P.bin[b].p, P.bout[b].p, cnt, cap, &got, P.s_k);
    if (rc) break;
    rc = hip_rc(hipEventRecord(P.e_k[b], P.s_k), "host pipeline");
    if (rc) break;
    const Sortie o{b, produced, got};
    if (two_threads) {
      {
        std::lock_guard<std::mutex> l(m);
        file.push_back(o);
      }
      cv.notify_all();
    } else {
      rc = copie_sortie(o);
    }
    produced += got;
  }
  arrete_fil();        // (drains its queue first)
  const std::string msg = rc ? last_error_ref() : std::string();
  (void) hipStreamSynchronize(P.s_out);
  (void) hipStreamSynchronize(P.s_k);
  if (rc) return set_err(rc, "%s", msg.c_str());
  if (rc_out) return set_err(rc_out, "%s", msg_out.c_str());
  // later work on the caller's stream sees the operator's new state
  TSD_HIP(hipEventRecord(P.e_user, P.s_k));
  TSD_HIP(hipStreamWaitEvent(user, P.e_user, 0));
  if (n_out) *n_out = produced;
  return TSDGPU_OK;
}

const char *dev_switch(const char *name)
{
  char key[96];
  snprintf(key, sizeof key, "TSDGPU_%s", name);
  return getenv(key);
}
int dev_switch_int(const char *name, int dflt)
{
  const char *v = dev_switch(name);
  return v ? atoi(v) : dflt;
}

bool host_pipe_enabled()
{
  static const bool on = getenv("TSDGPU_NO_PIPE") == nullptr;
  return on;
}

bool host_ranges_overlap(const void *a, size_t na, const void *b, size_t nb)
{
  const char *pa = (const char *) a, *pb = (const char *) b;
  return pa < pb + nb && pb < pa + na;
}

}  // namespace tsdgpu

extern "C" {

const char *tsdgpu_last_error(void) { return tsdgpu::last_error_ref().c_str(); }

int tsdgpu_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void) hipGetLastError();
    return 0;
  }
  return n;
}

int tsdgpu_current_device(void)
{
  int d = -1;
  if (hipGetDevice(&d) != hipSuccess) {
    (void) hipGetLastError();
    return -1;
  }
  return d;
}

const char *tsdgpu_version(void) { return "libtsd_amd 0.2 (gfx950)"; }

int tsdgpu_malloc(void **out, size_t bytes)
{
  TSD_CHECK(out != nullptr, "tsdgpu_malloc: out is NULL");
  *out = nullptr;
  if (bytes == 0) return TSDGPU_OK;
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess) {
    *out = nullptr;
    return tsdgpu::set_err(TSDGPU_ERR_ALLOC, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  }
  return TSDGPU_OK;
}
int tsdgpu_free(void *p)
{
  if (p) TSD_HIP(hipFree(p));
  return TSDGPU_OK;
}
int tsdgpu_malloc_host(void **out, size_t bytes)
{
  TSD_CHECK(out != nullptr, "tsdgpu_malloc_host: out is NULL");
  *out = nullptr;
  if (bytes == 0) return TSDGPU_OK;
  hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    *out = nullptr;
    return tsdgpu::set_err(TSDGPU_ERR_ALLOC, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  }
  return TSDGPU_OK;
}
int tsdgpu_free_host(void *p)
{
  if (p) TSD_HIP(hipHostFree(p));
  return TSDGPU_OK;
}
int tsdgpu_memcpy(void *dst, const void *src, size_t bytes, void *stream)
{
  if (bytes == 0) return TSDGPU_OK;
  TSD_CHECK(dst != nullptr && src != nullptr, "tsdgpu_memcpy: NULL pointer");
  hipStream_t st = (hipStream_t) stream;
  TSD_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st));
  if (!(tsdgpu::is_device_ptr(dst) && tsdgpu::is_device_ptr(src))) TSD_HIP(hipStreamSynchronize(st));
  return TSDGPU_OK;
}
int tsdgpu_memset(void *dev, int value, size_t bytes, void *stream)
{
  if (bytes == 0) return TSDGPU_OK;
  TSD_CHECK(dev != nullptr, "tsdgpu_memset: NULL pointer");
  TSD_HIP(hipMemsetAsync(dev, value, bytes, (hipStream_t) stream));
  return TSDGPU_OK;
}
int tsdgpu_zero_imag(void *y, int64_t n, void *stream)
{
  if (n <= 0) return TSDGPU_OK;
  TSD_CHECK(y != nullptr, "tsdgpu_zero_imag: NULL pointer");
  if (!tsdgpu::is_device_ptr(y)) {
    float *f = (float *) y;
    for (int64_t i = 0; i < n; i++) f[2 * i + 1] = 0.f;
    return TSDGPU_OK;
  }
  hipLaunchKernelGGL(tsdgpu::zero_imag_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, (hipStream_t) stream, (float2 *) y, n);
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}
int tsdgpu_synchronize(void *stream)
{
  TSD_HIP(hipStreamSynchronize((hipStream_t) stream));
  return TSDGPU_OK;
}
int tsdgpu_is_device_pointer(const void *p) { return tsdgpu::is_device_ptr(p) ? 1 : 0; }

}  // extern "C"
