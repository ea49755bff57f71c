// common.hip -- error reporting and host/device staging for the C-ABI shim.
#include "common.hpp"

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

int stage_in(const void *src, size_t bytes, DevBuf &buf, hipStream_t st, const void **dev)
{
  if (bytes == 0 || is_device_ptr(src)) {
    *dev = src;
    return TSDGPU_OK;
  }
  int rc = buf.reserve(bytes);
  if (rc) return rc;
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
  TSD_HIP(hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, st));
  TSD_HIP(hipStreamSynchronize(st));
  return TSDGPU_OK;
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
int tsdgpu_synchronize(void *stream)
{
  TSD_HIP(hipStreamSynchronize((hipStream_t) stream));
  return TSDGPU_OK;
}
int tsdgpu_is_device_pointer(const void *p) { return tsdgpu::is_device_ptr(p) ? 1 : 0; }

}  // extern "C"
