// vecops.hip -- element-wise operations on RESIDENT vectors: the slice of libtsd's array arithmetic that user code puts
// between two operators of the path (Tab::reverse / operator*= / operator/= / + - * / abs / abs2 / real / imag / as_complex:
// core/include/tsd/tableau.hpp:883-901,1141-1257; core/src/tableau.cc:582-592,822-854,1243-1533,1704-1714), so that
// filtfilt (filter, reverse, filter, reverse) or y = fft(x) * H never leave the GPU.  One thread per element, 16-B-free plain
// accesses: these are copies with an operation attached (HBM-bound, 8-24 B per element).  Same IEEE operations as the host
// loops of the mirror: no contraction (this file is built with -ffp-contract=off), complex products as ac - bd / ad + bc,
// complex quotients evaluated in double like libgcc's __divsc3.
#include "common.hpp"
#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>

namespace tsdgpu {
namespace {

__device__ __forceinline__ float2 cmulv(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cdivv(float2 a, float2 b)
{
  const double aa = a.x, bb = a.y, cc = b.x, dd = b.y;
  const double den = cc * cc + dd * dd;
  return make_float2((float) ((aa * cc + bb * dd) / den), (float) ((bb * cc - aa * dd) / den));
}

template <typename T> __device__ __forceinline__ T op_mul(T a, T b);
template <> __device__ __forceinline__ float op_mul<float>(float a, float b) { return a * b; }
template <> __device__ __forceinline__ float2 op_mul<float2>(float2 a, float2 b) { return cmulv(a, b); }
template <typename T> __device__ __forceinline__ T op_div(T a, T b);
template <> __device__ __forceinline__ float op_div<float>(float a, float b) { return a / b; }
template <> __device__ __forceinline__ float2 op_div<float2>(float2 a, float2 b) { return cdivv(a, b); }
__device__ __forceinline__ float op_add(float a, float b) { return a + b; }
__device__ __forceinline__ float2 op_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float op_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float2 op_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float op_neg(float a) { return -a; }
__device__ __forceinline__ float2 op_neg(float2 a) { return make_float2(-a.x, -a.y); }

// same-type operations: dst, a, b of type T
template <typename T>
__global__ __launch_bounds__(256) void vec_same_kernel(int op, T *__restrict__ dst, const T *a, const T *b, T s, int64_t n)
{
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  switch (op) {
    case TSDGPU_VEC_REVERSE: dst[i] = a[n - 1 - i]; break;
    case TSDGPU_VEC_SCALE: dst[i] = op_mul<T>(a[i], s); break;
    case TSDGPU_VEC_DIV_SCALAR: dst[i] = op_div<T>(a[i], s); break;
    case TSDGPU_VEC_ADD: dst[i] = op_add(a[i], b[i]); break;
    case TSDGPU_VEC_SUB: dst[i] = op_sub(a[i], b[i]); break;
    case TSDGPU_VEC_MUL: dst[i] = op_mul<T>(a[i], b[i]); break;
    case TSDGPU_VEC_NEG: dst[i] = op_neg(a[i]); break;
    default: break;
  }
}

// complex -> real, real -> complex, conjugate
__global__ __launch_bounds__(256) void vec_c2r_kernel(int op, float *__restrict__ dst, const float2 *__restrict__ a, int64_t n)
{
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float2 v = a[i];
  switch (op) {
    case TSDGPU_VEC_ABS: dst[i] = hypotf(v.x, v.y); break;
    case TSDGPU_VEC_ABS2: dst[i] = v.x * v.x + v.y * v.y; break;
    case TSDGPU_VEC_REAL: dst[i] = v.x; break;
    case TSDGPU_VEC_IMAG: dst[i] = v.y; break;
    default: break;
  }
}
__global__ __launch_bounds__(256) void vec_r2r_kernel(int op, float *__restrict__ dst, const float *a, int64_t n)
{
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = a[i];
  dst[i] = op == TSDGPU_VEC_ABS ? fabsf(v) : (op == TSDGPU_VEC_ABS2 ? v * v : v);
}
__global__ __launch_bounds__(256) void vec_r2c_kernel(float2 *__restrict__ dst, const float *__restrict__ a, int64_t n)
{
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = make_float2(a[i], 0.f);
}
__global__ __launch_bounds__(256) void vec_conj_kernel(float2 *__restrict__ dst, const float2 *a, int64_t n)
{
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = make_float2(a[i].x, -a[i].y);
}

// ---- reductions: sum in double (the host loops accumulate in double too: tableau.hpp:656-717), max / min / first arg max
struct RedPart { double s_re, s_im; float vmax, vmin; long long imax; };

template <bool CPLX>
__global__ __launch_bounds__(256) void vec_reduce_kernel(const float *__restrict__ a, int64_t n, RedPart *__restrict__ part)
{
  __shared__ RedPart sh[256];
  RedPart r{0.0, 0.0, -INFINITY, INFINITY, -1};
  for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256) {
    if (CPLX) {
      r.s_re += (double) a[2 * i];
      r.s_im += (double) a[2 * i + 1];
    } else {
      const float v = a[i];
      r.s_re += (double) v;
      if (v > r.vmax) { r.vmax = v; r.imax = i; }      // strictly greater: the FIRST maximum of the thread's ascending indices
      if (v < r.vmin) r.vmin = v;
    }
  }
  sh[threadIdx.x] = r;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int) threadIdx.x < d) {
      RedPart &x = sh[threadIdx.x];
      const RedPart y = sh[threadIdx.x + d];
      x.s_re += y.s_re;
      x.s_im += y.s_im;
      if (y.vmax > x.vmax || (y.vmax == x.vmax && y.imax >= 0 && (x.imax < 0 || y.imax < x.imax))) { x.vmax = y.vmax; x.imax = y.imax; }
      if (y.vmin < x.vmin) x.vmin = y.vmin;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

}  // namespace
}  // namespace tsdgpu

using namespace tsdgpu;

extern "C" int tsdgpu_vec_reduce(int data_type, const void *a, int64_t n, double *sum_re_im, float *max_min, int64_t *arg_max,
                                 void *stream)
{
  TSD_CHECK(n >= 0, "vec_reduce: negative length");
  TSD_CHECK(data_type == TSDGPU_F32 || data_type == TSDGPU_C64, "vec_reduce: bad data_type %d", data_type);
  if (sum_re_im) sum_re_im[0] = sum_re_im[1] = 0.0;
  if (arg_max) *arg_max = -1;
  if (n == 0) return TSDGPU_OK;
  TSD_CHECK(a != nullptr && is_device_ptr(a), "vec_reduce: a resident vector is expected");
  hipStream_t st = (hipStream_t) stream;
  const int nb = (int) std::min<int64_t>(cdiv(n, 256), 1024);
  // partial results: a 48-KiB device block borrowed for the call from a per-device free list (never freed; as many
  // blocks as calls ever ran at once)
  struct Bloc { int dev; DevBuf buf; };
  static std::mutex m;
  static std::vector<Bloc *> *libres = new std::vector<Bloc *>();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) (void) hipGetLastError();
  Bloc *bl = nullptr;
  {
    std::lock_guard<std::mutex> l(m);
    for (size_t i = 0; i < libres->size(); i++)
      if ((*libres)[i]->dev == dev) {
        bl = (*libres)[i];
        libres->erase(libres->begin() + (long) i);
        break;
      }
  }
  if (!bl) {
    bl = new Bloc();
    bl->dev = dev;
  }
  struct Rend {
    Bloc *b;
    ~Rend() { std::lock_guard<std::mutex> l(m); libres->push_back(b); }
  } rend{bl};
  DevBuf &parts = bl->buf;
  int rc = parts.reserve((size_t) 1024 * sizeof(RedPart));
  if (rc) return rc;
  if (data_type == TSDGPU_C64) hipLaunchKernelGGL(vec_reduce_kernel<true>, dim3(nb), dim3(256), 0, st, (const float *) a, n, (RedPart *) parts.p);
  else hipLaunchKernelGGL(vec_reduce_kernel<false>, dim3(nb), dim3(256), 0, st, (const float *) a, n, (RedPart *) parts.p);
  TSD_HIP(hipGetLastError());
  std::vector<RedPart> h((size_t) nb);
  TSD_HIP(hipMemcpyAsync(h.data(), parts.p, (size_t) nb * sizeof(RedPart), hipMemcpyDeviceToHost, st));
  TSD_HIP(hipStreamSynchronize(st));
  // the block partials are folded on the host in block order: deterministic, and a few KB
  RedPart t{0.0, 0.0, -INFINITY, INFINITY, -1};
  for (const RedPart &p : h) {
    t.s_re += p.s_re;
    t.s_im += p.s_im;
    if (p.vmax > t.vmax || (p.vmax == t.vmax && p.imax >= 0 && (t.imax < 0 || p.imax < t.imax))) { t.vmax = p.vmax; t.imax = p.imax; }
    if (p.vmin < t.vmin) t.vmin = p.vmin;
  }
  if (sum_re_im) { sum_re_im[0] = t.s_re; sum_re_im[1] = t.s_im; }
  if (max_min) { max_min[0] = t.vmax; max_min[1] = t.vmin; }
  if (arg_max) *arg_max = t.imax;
  return TSDGPU_OK;
}


extern "C" int tsdgpu_vec_op(int op, int data_type, void *dst, const void *a, const void *b, float s_re, float s_im, int64_t n,
                             void *stream)
{
  TSD_CHECK(n >= 0, "vec_op: negative length");
  if (n == 0) return TSDGPU_OK;
  TSD_CHECK(data_type == TSDGPU_F32 || data_type == TSDGPU_C64, "vec_op: bad data_type %d", data_type);
  TSD_CHECK(dst != nullptr && a != nullptr, "vec_op: NULL vector");
  TSD_CHECK(is_device_ptr(dst) && is_device_ptr(a) && (b == nullptr || is_device_ptr(b)),
            "vec_op: resident vectors only (host vectors are host arithmetic)");
  hipStream_t st = (hipStream_t) stream;
  const dim3 grid((unsigned) cdiv(n, 256)), blk(256);
  const bool cplx = data_type == TSDGPU_C64;
  switch (op) {
    case TSDGPU_VEC_REVERSE:
      TSD_CHECK(dst != a, "vec_op: reverse cannot run in place");
      [[fallthrough]];
    case TSDGPU_VEC_SCALE:
    case TSDGPU_VEC_DIV_SCALAR:
    case TSDGPU_VEC_NEG:
    case TSDGPU_VEC_ADD:
    case TSDGPU_VEC_SUB:
    case TSDGPU_VEC_MUL:
      if (op == TSDGPU_VEC_ADD || op == TSDGPU_VEC_SUB || op == TSDGPU_VEC_MUL) TSD_CHECK(b != nullptr, "vec_op: second operand missing");
      if (cplx)
        hipLaunchKernelGGL(vec_same_kernel<float2>, grid, blk, 0, st, op, (float2 *) dst, (const float2 *) a, (const float2 *) b,
                           make_float2(s_re, s_im), n);
      else
        hipLaunchKernelGGL(vec_same_kernel<float>, grid, blk, 0, st, op, (float *) dst, (const float *) a, (const float *) b, s_re, n);
      break;
    case TSDGPU_VEC_ABS:
    case TSDGPU_VEC_ABS2:
      if (cplx) hipLaunchKernelGGL(vec_c2r_kernel, grid, blk, 0, st, op, (float *) dst, (const float2 *) a, n);
      else hipLaunchKernelGGL(vec_r2r_kernel, grid, blk, 0, st, op, (float *) dst, (const float *) a, n);
      break;
    case TSDGPU_VEC_REAL:
    case TSDGPU_VEC_IMAG:
      TSD_CHECK(cplx, "vec_op: real / imag take a complex vector");
      hipLaunchKernelGGL(vec_c2r_kernel, grid, blk, 0, st, op, (float *) dst, (const float2 *) a, n);
      break;
    case TSDGPU_VEC_TO_COMPLEX:
      TSD_CHECK(!cplx, "vec_op: as_complex takes a real vector");
      hipLaunchKernelGGL(vec_r2c_kernel, grid, blk, 0, st, (float2 *) dst, (const float *) a, n);
      break;
    case TSDGPU_VEC_CONJ:
      TSD_CHECK(cplx, "vec_op: conj takes a complex vector");
      hipLaunchKernelGGL(vec_conj_kernel, grid, blk, 0, st, (float2 *) dst, (const float2 *) a, n);
      break;
    default:
      return set_err(TSDGPU_ERR_INVALID, "vec_op: unknown operation %d", op);
  }
  TSD_HIP(hipGetLastError());
  return TSDGPU_OK;
}
