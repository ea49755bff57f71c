// sharded.hip -- one process, several GPUs: a long vector cut into contiguous chunks, one stateful
// operator handle per chunk, and the ONE small left-neighbour halo each operator needs
// (north_star: "long vectors shard by contiguous chunk with halo exchange"):
//   FIR        the K-1 input samples before the chunk           -> tsdgpu_fir_set_history
//   SOS        W warm-up samples (state transition < 1e-9)      -> reset + step on the halo
//   resampler  the K-1-sample window + the absolute position    -> tsdgpu_resampler_seek
// No collective and no data-path exchange beyond those few samples: shards run concurrently, one
// host thread and one stream per shard.  Two forms:
//   *_step_host   one HOST vector in, one out: every shard stages its chunk (and reads its halo straight
//                 from the host vector); this is what the C++ adaptors call for large host vectors
//   *_step_parts  chunk g already RESIDENT on device g: the halo moves device-to-device
//                 (hipMemcpyPeerAsync over xGMI), nothing touches the host
// Shards are logical: several may name the same device (how the single-GPU tests run N shards and
// compare with one handle, bit for bit where the operator is chunk-invariant).
// The streaming contract is kept across calls: the tail of call c is the halo of shard 0 in call c+1.
#include "common.hpp"
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <thread>

using namespace tsdgpu;

namespace {
enum Kind { K_FIR = 0, K_SOS = 1, K_RES = 2 };

// reusable barrier for the shard threads (C++17: no std::barrier)
struct Rendezvous {
  std::mutex m;
  std::condition_variable cv;
  int n = 0, arrived = 0, phase = 0;
  void wait()
  {
    std::unique_lock<std::mutex> l(m);
    const int ph = phase;
    if (++arrived == n) {
      arrived = 0;
      phase++;
      cv.notify_all();
    } else {
      cv.wait(l, [&] { return phase != ph; });
    }
  }
};
}  // namespace

namespace tsdgpu {
// sos.hip: the stream state of a cascade as a host vector, and its zero-input propagation over L samples (+ an end state)
int sos_state_floats();
int sos_state_get(tsdgpu_sos *s, float *host, hipStream_t st);
int sos_state_set(tsdgpu_sos *s, const float *host, hipStream_t st);
bool sos_state_propagate(const tsdgpu_sos *s, int64_t L, const float *in, const float *add, float *out);
}

namespace {
// Restores the caller's current device on EVERY exit path of an entry point that switches devices.
struct DeviceGuard {
  int prev = 0;
  bool ok = false;
  DeviceGuard() { ok = hipGetDevice(&prev) == hipSuccess; if (!ok) (void) hipGetLastError(); }
  ~DeviceGuard() { if (ok) (void) hipSetDevice(prev); }
};

// One persistent host thread per shard beyond the first (tsdgpu_sharded_step_host): created at the handle's first host call,
// parked on a condition variable between calls, joined when the handle is destroyed.  (A std::thread per shard per call
// cost a thread creation + its first hipSetDevice on every step.)
struct ShardWorkers {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv, cv_done;
  const std::function<void(int)> *job = nullptr;
  uint64_t gen = 0;
  int pending = 0;
  bool stop = false;
  void start(int nshards)
  {
    if (!th.empty() || nshards < 2) return;
    for (int g = 1; g < nshards; g++)
      th.emplace_back([this, g] {
        uint64_t seen = 0;
        for (;;) {
          const std::function<void(int)> *f = nullptr;
          {
            std::unique_lock<std::mutex> l(m);
            cv.wait(l, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            f = job;
          }
          (*f)(g);
          {
            std::lock_guard<std::mutex> l(m);
            if (--pending == 0) cv_done.notify_all();
          }
        }
      });
  }
  // f(g) for g = 1 .. nshards-1 on the workers, f(0) on the caller; returns when all have finished
  void run(int nshards, const std::function<void(int)> &f)
  {
    start(nshards);
    {
      std::lock_guard<std::mutex> l(m);
      job = &f;
      pending = (int) th.size();
      gen++;
    }
    cv.notify_all();
    f(0);
    std::unique_lock<std::mutex> l(m);
    cv_done.wait(l, [&] { return pending == 0; });
    job = nullptr;
  }
  ~ShardWorkers()
  {
    {
      std::lock_guard<std::mutex> l(m);
      stop = true;
    }
    cv.notify_all();
    for (auto &t : th) t.join();
  }
};
}  // namespace

struct tsdgpu_sharded {
  int kind = 0, data_type = 0, nshards = 0;
  int64_t H = 0;                         // halo length in samples
  int K = 0;                             // resampler / FIR taps
  std::vector<int> dev;
  std::vector<void *> handle;            // tsdgpu_fir* / tsdgpu_sos* / tsdgpu_resampler*
  std::vector<void *> edge;              // FIR / resampler: a second handle of the same operator for the first H outputs of a
                                         // resident part (the interior is launched on `handle` before the halo has arrived)
  std::vector<hipStream_t> stream;
  std::vector<hipStream_t> hstream;      // resident form: the halo copies of shard g run here, beside the interior on stream[g]
  std::vector<hipEvent_t> ev_ready, ev_halo, ev_tail;   // part g produced / halo g landed / the call's tail read out of part g
  char *pin_carry = nullptr;             // page-locked staging of the new carry (resident form: an asynchronous D2H)
  ShardWorkers workers;
  std::vector<DevBuf> halo, in, out, scratch;
  std::vector<char> carry;               // host: the last H inputs of the stream (zeros before its start)
  int64_t seen = 0;                      // inputs consumed so far (all calls)
  int64_t out_total = 0;                 // resampler: outputs produced so far
  // SOS cascades whose warm-up would be longer than a shard is worth (or that do not decay): EXACT sharding instead of
  // halos.  Every shard but the first runs from zero state and hands its end state E_g to the host; the true start states
  // follow from S_{g+1} = Phi^(L_g) S_g + E_g (a 2-sections-wide product per shard, in double); the shards run again from
  // them.  The exchange is a few hundred bytes per shard through the host -- no collective; `sos_state` is the stream's
  // state between calls.
  bool exact = false;
  std::vector<float> sos_state;
  size_t esz() const { return dtype_size(data_type); }
};

namespace {

int with_device(int d, const std::function<int()> &fn)
{
  int prev = 0;
  TSD_HIP(hipGetDevice(&prev));
  if (prev != d) TSD_HIP(hipSetDevice(d));
  const int rc = fn();
  if (prev != d) (void) hipSetDevice(prev);
  return rc;
}

int sharded_alloc(tsdgpu_sharded **out, int kind, int data_type, int nshards, const int *devices)
{
  TSD_CHECK(out != nullptr, "sharded_create: out is NULL");
  *out = nullptr;
  TSD_CHECK(nshards >= 1 && nshards <= 64, "sharded_create: %d shards (1..64)", nshards);
  const int ndev = tsdgpu_device_count();
  TSD_CHECK(ndev > 0, "sharded_create: no HIP device");
  tsdgpu_sharded *h = new tsdgpu_sharded();
  h->kind = kind;
  h->data_type = data_type;
  h->nshards = nshards;
  for (int g = 0; g < nshards; g++) {
    const int d = devices ? devices[g] : g % ndev;
    if (d < 0 || d >= ndev) {
      delete h;
      return set_err(TSDGPU_ERR_INVALID, "sharded_create: device %d of shard %d does not exist (%d devices)", d, g, ndev);
    }
    h->dev.push_back(d);
  }
  h->handle.assign((size_t) nshards, nullptr);
  h->edge.assign((size_t) nshards, nullptr);
  h->stream.assign((size_t) nshards, nullptr);
  h->hstream.assign((size_t) nshards, nullptr);
  h->ev_ready.assign((size_t) nshards, nullptr);
  h->ev_halo.assign((size_t) nshards, nullptr);
  h->ev_tail.assign((size_t) nshards, nullptr);
  h->halo.resize((size_t) nshards);
  h->in.resize((size_t) nshards);
  h->out.resize((size_t) nshards);
  h->scratch.resize((size_t) nshards);
  *out = h;
  return TSDGPU_OK;
}

int sharded_finish_create(tsdgpu_sharded *h)
{
  h->carry.assign((size_t) std::max<int64_t>(h->H, 1) * h->esz(), 0);
  for (int g = 0; g < h->nshards; g++) {
    const int rc = with_device(h->dev[g], [&]() -> int {
      TSD_HIP(hipStreamCreateWithFlags(&h->stream[g], hipStreamNonBlocking));
      TSD_HIP(hipStreamCreateWithFlags(&h->hstream[g], hipStreamNonBlocking));
      TSD_HIP(hipEventCreateWithFlags(&h->ev_ready[g], hipEventDisableTiming));
      TSD_HIP(hipEventCreateWithFlags(&h->ev_halo[g], hipEventDisableTiming));
      TSD_HIP(hipEventCreateWithFlags(&h->ev_tail[g], hipEventDisableTiming));
      return h->halo[g].reserve((size_t) std::max<int64_t>(h->H, 1) * h->esz());
    });
    if (rc) return rc;
  }
  if (h->H > 0 && hipHostMalloc((void **) &h->pin_carry, (size_t) h->H * h->esz(), hipHostMallocDefault) != hipSuccess) {
    (void) hipGetLastError();
    h->pin_carry = nullptr;              // (the resident form then reads the tail with a blocking copy)
  }
  // neighbours exchange halos device to device in the resident form
  for (int g = 1; g < h->nshards; g++)
    if (h->dev[g] != h->dev[g - 1])
      (void) with_device(h->dev[g], [&]() -> int {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, h->dev[g], h->dev[g - 1]) == hipSuccess && can) {
          if (hipDeviceEnablePeerAccess(h->dev[g - 1], 0) != hipSuccess) (void) hipGetLastError();   // already enabled is fine
        }
        return TSDGPU_OK;
      });
  return TSDGPU_OK;
}

// the operator on shard g: halo (H samples, device pointer on this shard's device, `avail` of them real
// stream samples at its END), then the chunk dx -> dy of cnt samples.  For the resampler dy receives
// out_cnt outputs.  Must be called with the shard's device current.
int run_shard(tsdgpu_sharded *h, int g, const void *halo, int64_t avail, int64_t pos, const void *dx, void *dy, int64_t cnt,
              int64_t ycap, int64_t *out_cnt)
{
  hipStream_t st = h->stream[g];
  const size_t esz = h->esz();
  if (h->kind == K_FIR) {
    tsdgpu_fir *f = (tsdgpu_fir *) h->handle[g];
    if (h->H > 0) {
      const int rc = tsdgpu_fir_set_history(f, halo, st);
      if (rc) return rc;
    }
    return cnt > 0 ? tsdgpu_fir_step(f, dx, dy, cnt, st) : TSDGPU_OK;
  }
  if (h->kind == K_SOS) {
    tsdgpu_sos *s = (tsdgpu_sos *) h->handle[g];
    if (pos > 0) {
      // not the start of the stream: warm the sections up on the samples before the chunk.  When these
      // reach back to the very first sample the warm-up IS the stream (first-sample seed included)
      int rc = tsdgpu_sos_reset_on(s, st);
      if (rc) return rc;
      if (avail > 0) {
        rc = h->scratch[g].reserve((size_t) avail * esz);
        if (rc) return rc;
        rc = tsdgpu_sos_step(s, (const char *) halo + (size_t) (h->H - avail) * esz, h->scratch[g].p, avail, st);
        if (rc) return rc;
      }
    }
    return cnt > 0 ? tsdgpu_sos_step(s, dx, dy, cnt, st) : TSDGPU_OK;
  }
  tsdgpu_resampler *r = (tsdgpu_resampler *) h->handle[g];
  int rc = tsdgpu_resampler_seek(r, pos, h->H > 0 ? halo : nullptr, st);
  if (rc) return rc;
  *out_cnt = 0;
  if (cnt <= 0) return TSDGPU_OK;
  return tsdgpu_resampler_step(r, dx, cnt, dy, ycap, out_cnt, st);
}

// new carry = last H samples of (old carry ++ x[0, n)), x on the host
void update_carry_host(tsdgpu_sharded *h, const char *x, int64_t n)
{
  const size_t esz = h->esz();
  const int64_t H = h->H;
  if (H <= 0) return;
  if (n >= H) {
    std::memcpy(h->carry.data(), x + (size_t) (n - H) * esz, (size_t) H * esz);
  } else if (n > 0) {
    std::memmove(h->carry.data(), h->carry.data() + (size_t) n * esz, (size_t) (H - n) * esz);
    std::memcpy(h->carry.data() + (size_t) (H - n) * esz, x, (size_t) n * esz);
  }
}

}  // namespace

extern "C" {

int tsdgpu_fir_sharded_create(tsdgpu_sharded **out, int data_type, int tap_type, const void *taps_host, int ntaps, int method,
                              int nshards, const int *devices)
{
  int rc = sharded_alloc(out, K_FIR, data_type, nshards, devices);
  if (rc) return rc;
  tsdgpu_sharded *h = *out;
  h->K = ntaps;
  h->H = std::max(ntaps - 1, 0);
  for (int g = 0; g < nshards && !rc; g++)
    rc = with_device(h->dev[g], [&] {
      const int rc2 = tsdgpu_fir_create((tsdgpu_fir **) &h->handle[g], data_type, tap_type, taps_host, ntaps, method);
      if (rc2 || ntaps < 2) return rc2;
      // the first K-1 outputs of a resident part: one small launch of the direct kernel
      return tsdgpu_fir_create((tsdgpu_fir **) &h->edge[g], data_type, tap_type, taps_host, ntaps, TSDGPU_FIR_DIRECT);
    });
  if (!rc) rc = sharded_finish_create(h);
  if (rc) {
    tsdgpu_sharded_destroy(h);
    *out = nullptr;
  }
  return rc;
}

int tsdgpu_sos_sharded_create(tsdgpu_sharded **out, int data_type, const float *coefs_host, int nsec, float gain, const float *rii1_host,
                              int forme, int nshards, const int *devices)
{
  int rc = sharded_alloc(out, K_SOS, data_type, nshards, devices);
  if (rc) return rc;
  tsdgpu_sharded *h = *out;
  for (int g = 0; g < nshards && !rc; g++)
    rc = with_device(h->dev[g], [&] { return tsdgpu_sos_create((tsdgpu_sos **) &h->handle[g], data_type, coefs_host, nsec, gain, rii1_host, forme); });
  if (!rc) {
    h->H = tsdgpu_sos_halo((tsdgpu_sos *) h->handle[0]);
    static const bool toujours_halo = dev_switch("SHARD_SOS_HALO") != nullptr;
    if (h->H < 0 || (h->H > 65536 && !toujours_halo) || h->H > ((int64_t) 1 << 24)) {
      // a warm-up of more than 2^16 samples per shard (or no decay at all): the exact scheme
      h->exact = true;
      h->H = 0;
      h->sos_state.assign((size_t) sos_state_floats(), 0.f);
    }
  }
  if (!rc) rc = sharded_finish_create(h);
  if (rc) {
    tsdgpu_sharded_destroy(h);
    *out = nullptr;
  }
  return rc;
}

int tsdgpu_resampler_sharded_create(tsdgpu_sharded **out, int data_type, float ratio, const float *lut_host, int K, int nphases,
                                    int nshards, const int *devices)
{
  int rc = sharded_alloc(out, K_RES, data_type, nshards, devices);
  if (rc) return rc;
  tsdgpu_sharded *h = *out;
  h->K = K;
  h->H = std::max(K - 1, 0);
  for (int g = 0; g < nshards && !rc; g++)
    rc = with_device(h->dev[g], [&] {
      const int rc2 = tsdgpu_resampler_create((tsdgpu_resampler **) &h->handle[g], data_type, ratio, lut_host, K, nphases);
      if (rc2 || K < 2) return rc2;
      return tsdgpu_resampler_create((tsdgpu_resampler **) &h->edge[g], data_type, ratio, lut_host, K, nphases);
    });
  if (!rc) rc = sharded_finish_create(h);
  if (rc) {
    tsdgpu_sharded_destroy(h);
    *out = nullptr;
  }
  return rc;
}

int tsdgpu_sharded_count(const tsdgpu_sharded *h) { return h ? h->nshards : 0; }
int64_t tsdgpu_sharded_halo(const tsdgpu_sharded *h) { return h ? h->H : -1; }
int tsdgpu_sharded_device(const tsdgpu_sharded *h, int shard) { return (h && shard >= 0 && shard < h->nshards) ? h->dev[shard] : -1; }

void tsdgpu_sharded_bounds(const tsdgpu_sharded *h, int64_t n, int shard, int64_t *lo, int64_t *hi)
{
  const int N = h ? h->nshards : 1;
  *lo = (int64_t) (((__int128) n * shard) / N);
  *hi = (int64_t) (((__int128) n * (shard + 1)) / N);
}

int64_t tsdgpu_sharded_out_count(tsdgpu_sharded *h, int64_t n)
{
  if (!h || n < 0) return -1;
  if (h->kind != K_RES) return n;
  // outputs of the next n inputs: position-based, read from shard 0's handle (shared schedule)
  tsdgpu_resampler *r = (tsdgpu_resampler *) h->handle[0];
  int64_t a = 0, b = 0;
  if (with_device(h->dev[0], [&]() -> int {
        int rc = tsdgpu_resampler_seek(r, h->seen, nullptr, h->stream[0]);
        if (rc) return rc;
        a = tsdgpu_resampler_out_offset(r);
        rc = tsdgpu_resampler_seek(r, h->seen + n, nullptr, h->stream[0]);
        if (rc) return rc;
        b = tsdgpu_resampler_out_offset(r);
        return TSDGPU_OK;
      }))
    return -1;
  return b - a;
}

int tsdgpu_sharded_step_host(tsdgpu_sharded *h, const void *x, int64_t n, void *y, int64_t y_capacity, int64_t *n_out)
{
  TSD_CHECK(h != nullptr, "sharded_step_host: NULL handle");
  TSD_CHECK(n >= 0, "sharded_step_host: negative length");
  if (n_out) *n_out = 0;
  if (n == 0) return TSDGPU_OK;
  TSD_CHECK(x != nullptr, "sharded_step_host: NULL input");
  TSD_CHECK(!is_device_ptr(x) && !is_device_ptr(y), "sharded_step_host: x and y must be host vectors (resident chunks go through tsdgpu_sharded_step_parts)");
  const size_t esz = h->esz();
  const int N = h->nshards;
  const int64_t H = h->H;
  const char *xs = (const char *) x;
  char *ys = (char *) y;
  // output placement (resampler): offsets from the schedule
  std::vector<int64_t> off((size_t) N + 1, 0);
  if (h->kind == K_RES) {
    tsdgpu_resampler *r = (tsdgpu_resampler *) h->handle[0];
    const int rc = with_device(h->dev[0], [&]() -> int {
      for (int g = 0; g <= N; g++) {
        int64_t lo, hi;
        tsdgpu_sharded_bounds(h, n, std::min(g, N - 1), &lo, &hi);
        const int64_t p = h->seen + (g < N ? lo : n);
        const int rc2 = tsdgpu_resampler_seek(r, p, nullptr, h->stream[0]);
        if (rc2) return rc2;
        off[(size_t) g] = tsdgpu_resampler_out_offset(r);
      }
      return TSDGPU_OK;
    });
    if (rc) return rc;
    TSD_CHECK(off[(size_t) N] - off[0] <= y_capacity, "sharded_step_host: output capacity %lld < %lld outputs", (long long) y_capacity,
              (long long) (off[(size_t) N] - off[0]));
    // (a few inputs of a decimating ratio may produce nothing: no output vector is needed then)
    TSD_CHECK(y != nullptr || off[(size_t) N] == off[0], "sharded_step_host: NULL output");
  } else {
    TSD_CHECK(y != nullptr, "sharded_step_host: NULL output");
    TSD_CHECK(y_capacity >= n, "sharded_step_host: output capacity %lld < %lld", (long long) y_capacity, (long long) n);
  }
  // the halo of shard 0 comes from the previous calls; save the tail of this one first (y may be x)
  const std::vector<char> carry_avant = h->carry;
  const int64_t seen_avant = h->seen;
  update_carry_host(h, xs, n);

  std::vector<int> rcs((size_t) N, TSDGPU_OK);
  std::vector<std::string> msgs((size_t) N);
  std::vector<int64_t> outs((size_t) N, 0);
  Rendezvous rdv, rdv2;
  rdv.n = rdv2.n = N;
  // exact SOS sharding: the shards' end states, the first and the last shard with samples, the stream state after the call
  std::vector<float> etats, etat_final;
  int first = 0, dernier = 0;
  if (h->exact) {
    etats.assign((size_t) N * sos_state_floats(), 0.f);
    etat_final = h->sos_state;
    first = -1;
    for (int q = 0; q < N; q++) {
      int64_t ql, qh;
      tsdgpu_sharded_bounds(h, n, q, &ql, &qh);
      if (qh > ql) {
        if (first < 0) first = q;
        dernier = q;
      }
    }
    if (first < 0) first = 0;
  }
  auto work = [&](int g) {
    int rc = TSDGPU_OK;
    int64_t lo, hi;
    tsdgpu_sharded_bounds(h, n, g, &lo, &hi);
    const int64_t cnt = hi - lo;
    do {
      if (hipSetDevice(h->dev[g]) != hipSuccess) { rc = set_err(TSDGPU_ERR_HIP, "hipSetDevice(%d) failed", h->dev[g]); break; }
      hipStream_t st = h->stream[g];
      // (1) stage the halo and the chunk: everything read from the host vector BEFORE anyone writes y
      std::vector<char> hal((size_t) std::max<int64_t>(H, 1) * esz, 0);
      const int64_t du_x = std::min(H, lo);            // halo samples found in x, the rest in the carry
      if (du_x > 0) std::memcpy(hal.data() + (size_t) (H - du_x) * esz, xs + (size_t) (lo - du_x) * esz, (size_t) du_x * esz);
      if (du_x < H) std::memcpy(hal.data(), carry_avant.data() + (size_t) du_x * esz, (size_t) (H - du_x) * esz);
      if (H > 0 && hipMemcpyAsync(h->halo[g].p, hal.data(), (size_t) H * esz, hipMemcpyHostToDevice, st) != hipSuccess) {
        rc = set_err(TSDGPU_ERR_HIP, "halo upload failed: %s", hipGetErrorString(hipGetLastError()));
      }
      if (!rc && cnt > 0) rc = h->in[g].reserve((size_t) cnt * esz);
      const int64_t ocap = h->kind == K_RES ? off[(size_t) g + 1] - off[(size_t) g] : cnt;
      if (!rc && ocap > 0) rc = h->out[g].reserve((size_t) ocap * esz);
      if (!rc && cnt > 0 && hipMemcpyAsync(h->in[g].p, xs + (size_t) lo * esz, (size_t) cnt * esz, hipMemcpyHostToDevice, st) != hipSuccess)
        rc = set_err(TSDGPU_ERR_HIP, "chunk upload failed: %s", hipGetErrorString(hipGetLastError()));
      // unconditional: `hal` must outlive its copy whatever failed above
      if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = set_err(TSDGPU_ERR_HIP, "upload sync failed");
    } while (0);
    rdv.wait();                                         // every shard holds its inputs: y may now be written
    if (h->exact) {
      // exact SOS sharding: zero-state pass (the first shard with samples: the true pass), end states to the host, the
      // true start states by propagation, second pass.  Every thread passes both rendezvous whatever failed.
      tsdgpu_sos *sg = (tsdgpu_sos *) h->handle[g];
      const int SF = sos_state_floats();
      hipStream_t st = h->stream[g];
      if (!rc && cnt > 0) {
        std::vector<float> zero((size_t) SF, 0.f);
        zero[0] = 1.f;                                  // (zero memories, no first-sample seed)
        rc = sos_state_set(sg, g == first ? h->sos_state.data() : zero.data(), st);
        if (!rc) rc = tsdgpu_sos_step(sg, h->in[g].p, h->out[g].p, cnt, st);
        if (!rc) rc = sos_state_get(sg, &etats[(size_t) g * SF], st);
      }
      rcs[(size_t) g] = rc;
      if (rc) msgs[(size_t) g] = tsdgpu_last_error();
      rdv2.wait();
      bool ok = true;
      for (int q = 0; q < N; q++) ok = ok && rcs[(size_t) q] == TSDGPU_OK;
      if (ok && cnt > 0) {
        if (g != first) {
          std::vector<float> cur(etats.begin() + (size_t) first * SF, etats.begin() + (size_t) (first + 1) * SF), nxt((size_t) SF);
          for (int q = first + 1; q < g && !rc; q++) {
            int64_t ql, qh;
            tsdgpu_sharded_bounds(h, n, q, &ql, &qh);
            if (!sos_state_propagate(sg, qh - ql, cur.data(), &etats[(size_t) q * SF], nxt.data()))
              rc = set_err(TSDGPU_ERR_UNSUPPORTED, "sharded sos: the cascade's state transition over a shard leaves the float range");
            cur.swap(nxt);
          }
          if (!rc) rc = sos_state_set(sg, cur.data(), st);
          if (!rc) rc = tsdgpu_sos_step(sg, h->in[g].p, h->out[g].p, cnt, st);
        }
        if (!rc && g == dernier) rc = sos_state_get(sg, etat_final.data(), st);
        if (!rc && hipMemcpyAsync(ys + (size_t) lo * esz, h->out[g].p, (size_t) cnt * esz, hipMemcpyDeviceToHost, st) != hipSuccess)
          rc = set_err(TSDGPU_ERR_HIP, "download failed: %s", hipGetErrorString(hipGetLastError()));
        if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = set_err(TSDGPU_ERR_HIP, "shard %d: stream sync failed", g);
      }
      outs[(size_t) g] = cnt;
      rcs[(size_t) g] = rc;
      if (rc) msgs[(size_t) g] = tsdgpu_last_error();
      return;
    }
    if (!rc) {
      const int64_t ocap = h->kind == K_RES ? off[(size_t) g + 1] - off[(size_t) g] : cnt;
      int64_t got = 0;
      rc = run_shard(h, g, h->halo[g].p, std::min(H, seen_avant + lo), seen_avant + lo, h->in[g].p, h->out[g].p, cnt, ocap, &got);
      if (h->kind != K_RES) got = cnt;
      if (!rc && h->kind == K_RES && got != ocap)
        rc = set_err(TSDGPU_ERR_INVALID, "shard %d produced %lld outputs, the schedule says %lld", g, (long long) got, (long long) ocap);
      const int64_t o0 = h->kind == K_RES ? off[(size_t) g] - off[0] : lo;
      if (!rc && got > 0 &&
          hipMemcpyAsync(ys + (size_t) o0 * esz, h->out[g].p, (size_t) got * esz, hipMemcpyDeviceToHost, h->stream[g]) != hipSuccess)
        rc = set_err(TSDGPU_ERR_HIP, "download failed: %s", hipGetErrorString(hipGetLastError()));
      if (hipStreamSynchronize(h->stream[g]) != hipSuccess && !rc) rc = set_err(TSDGPU_ERR_HIP, "shard %d: stream sync failed", g);
      outs[(size_t) g] = got;
    }
    rcs[(size_t) g] = rc;
    if (rc) msgs[(size_t) g] = tsdgpu_last_error();
  };
  {
    DeviceGuard guard;                 // (work(0) runs on the calling thread and switches its device)
    h->workers.run(N, work);
  }
  int64_t total = 0;
  for (int g = 0; g < N; g++) {
    if (rcs[(size_t) g]) {
      h->carry = carry_avant;       // a failed call consumes nothing: the stream position and its tail stay as they were
      return set_err(rcs[(size_t) g], "shard %d: %s", g, msgs[(size_t) g].c_str());
    }
    total += outs[(size_t) g];
  }
  h->seen += n;
  h->out_total += total;
  if (h->exact) h->sos_state = etat_final;
  if (n_out) *n_out = total;
  return TSDGPU_OK;
}

// The resident form.  `producers` (may be NULL): producers[g] = the stream whose work produced x_parts[g] (and last touched
// y_parts[g]); the shard streams wait for an event recorded there.  Without it the call first waits for ALL prior work on the
// shards' devices (hipDeviceSynchronize), which is what a caller gets from tsdgpu_sharded_step_parts.
//
// Schedule (FIR, resampler): the halo copies run on a side stream per shard while the shard's stream already filters the
// INTERIOR of its part -- everything behind the first H samples, primed with the part's own first H samples, on the main
// handle -- and only the EDGE launch (the first H outputs, on the shard's second handle) waits for the halo.  A part filtered
// in place first lets the copies that read its tail (the next shards' halos, the call's carry) finish.
static int sharded_step_parts_impl(tsdgpu_sharded *h, const void *const *x_parts, const int64_t *counts, void *const *y_parts,
                                   const int64_t *y_capacities, int64_t *out_counts, void *const *producers)
{
  TSD_CHECK(h != nullptr && x_parts != nullptr && counts != nullptr && y_parts != nullptr, "sharded_step_parts: NULL argument");
  const size_t esz = h->esz();
  const int N = h->nshards;
  const int64_t H = h->H;
  int64_t n = 0;
  std::vector<int64_t> lo((size_t) N + 1, 0);
  for (int g = 0; g < N; g++) {
    TSD_CHECK(counts[g] >= 0, "sharded_step_parts: negative count");
    TSD_CHECK(counts[g] == 0 || (x_parts[g] && y_parts[g]), "sharded_step_parts: NULL part %d", g);
    lo[(size_t) g + 1] = lo[(size_t) g] + counts[g];
  }
  n = lo[(size_t) N];
  if (n == 0) return TSDGPU_OK;
  DeviceGuard guard;                                   // the caller's device comes back on every exit path
  TSD_CHECK(guard.ok, "sharded_step_parts: hipGetDevice failed");
  // (0) the parts must have been produced: an event per part on its producer's stream, or a device-wide wait
  if (producers) {
    for (int g = 0; g < N; g++)
      if (counts[g] > 0) {
        TSD_HIP(hipSetDevice(h->dev[g]));
        TSD_HIP(hipEventRecord(h->ev_ready[g], (hipStream_t) producers[g]));
      }
  } else {
    std::vector<int> vus;
    for (int g = 0; g < N; g++)
      if (counts[g] > 0 && std::find(vus.begin(), vus.end(), h->dev[g]) == vus.end()) {
        vus.push_back(h->dev[g]);
        TSD_HIP(hipSetDevice(h->dev[g]));
        TSD_HIP(hipDeviceSynchronize());
      }
  }
  auto wait_ready = [&](hipStream_t st, int part) -> int {
    if (producers && counts[part] > 0) TSD_HIP(hipStreamWaitEvent(st, h->ev_ready[part], 0));
    return TSDGPU_OK;
  };
  if (h->exact) {
    // exact SOS sharding (see tsdgpu_sharded): pass 1 of the later shards goes to a scratch buffer (a part may be filtered
    // in place), the end states come to the host, pass 2 starts from the propagated states
    const int SF = sos_state_floats();
    std::vector<float> etats((size_t) N * SF, 0.f), zero((size_t) SF, 0.f), fin = h->sos_state;
    zero[0] = 1.f;
    int first = -1, dernier = 0, rc = TSDGPU_OK;
    for (int g = 0; g < N; g++)
      if (counts[g] > 0) {
        if (first < 0) first = g;
        dernier = g;
      }
    for (int g = 0; g < N && !rc; g++) {
      if (counts[g] == 0) continue;
      TSD_HIP(hipSetDevice(h->dev[g]));
      tsdgpu_sos *sg = (tsdgpu_sos *) h->handle[g];
      TSD_CHECK(!y_capacities || y_capacities[g] >= counts[g], "sharded_step_parts: output capacity of part %d", g);
      rc = wait_ready(h->stream[g], g);
      if (!rc) rc = sos_state_set(sg, g == first ? h->sos_state.data() : zero.data(), h->stream[g]);
      void *dst = y_parts[g];
      if (!rc && g != first) {
        rc = h->scratch[g].reserve((size_t) counts[g] * esz);
        dst = h->scratch[g].p;
      }
      if (!rc) rc = tsdgpu_sos_step(sg, x_parts[g], dst, counts[g], h->stream[g]);
    }
    for (int g = 0; g < N && !rc; g++) {
      if (counts[g] == 0) continue;
      TSD_HIP(hipSetDevice(h->dev[g]));
      rc = sos_state_get((tsdgpu_sos *) h->handle[g], &etats[(size_t) g * SF], h->stream[g]);
    }
    if (!rc && first >= 0) {
      std::vector<float> cur(etats.begin() + (size_t) first * SF, etats.begin() + (size_t) (first + 1) * SF), nxt((size_t) SF);
      for (int g = first + 1; g < N && !rc; g++) {
        if (counts[g] == 0) continue;
        TSD_HIP(hipSetDevice(h->dev[g]));
        tsdgpu_sos *sg = (tsdgpu_sos *) h->handle[g];
        rc = sos_state_set(sg, cur.data(), h->stream[g]);
        if (!rc) rc = tsdgpu_sos_step(sg, x_parts[g], y_parts[g], counts[g], h->stream[g]);
        if (!rc && !sos_state_propagate(sg, counts[g], cur.data(), &etats[(size_t) g * SF], nxt.data()))
          rc = set_err(TSDGPU_ERR_UNSUPPORTED, "sharded sos: the cascade's state transition over a shard leaves the float range");
        cur.swap(nxt);
      }
      if (!rc) {
        TSD_HIP(hipSetDevice(h->dev[dernier]));
        rc = sos_state_get((tsdgpu_sos *) h->handle[dernier], fin.data(), h->stream[dernier]);
      }
    }
    for (int g = 0; g < N; g++) {
      (void) hipSetDevice(h->dev[g]);
      if (hipStreamSynchronize(h->stream[g]) != hipSuccess && !rc) rc = set_err(TSDGPU_ERR_HIP, "shard %d: stream sync failed", g);
    }
    if (rc) return rc;
    for (int g = 0; g < N; g++)
      if (out_counts) out_counts[g] = counts[g];
    h->sos_state = fin;
    h->seen += n;
    return TSDGPU_OK;
  }
  auto in_place = [&](int g) {
    if (counts[g] == 0 || h->kind == K_RES) return false;
    const char *xa = (const char *) x_parts[g], *ya = (const char *) y_parts[g];
    const size_t len = (size_t) counts[g] * esz;
    return xa < ya + len && ya < xa + len;
  };
  // readers[p]: events a part filtered in place must wait for before anything is written into it
  std::vector<std::vector<hipEvent_t>> readers((size_t) N);
  int rc = TSDGPU_OK;
  // (1) halos, device to device, on the shards' side streams: the last H samples before each shard, walking back over the
  //     parts and ending in the carry of the previous calls
  for (int g = 0; g < N && H > 0; g++) {
    TSD_HIP(hipSetDevice(h->dev[g]));
    hipStream_t hs = h->hstream[g];
    int64_t need = H;
    char *dst = (char *) h->halo[g].p;
    for (int p = g - 1; p >= 0 && need > 0; p--) {
      const int64_t take = std::min(need, counts[p]);
      if (take > 0) {
        rc = wait_ready(hs, p);
        if (rc) return rc;
        const char *src = (const char *) x_parts[p] + (size_t) (counts[p] - take) * esz;
        if (h->dev[p] == h->dev[g]) TSD_HIP(hipMemcpyAsync(dst + (size_t) (need - take) * esz, src, (size_t) take * esz, hipMemcpyDeviceToDevice, hs));
        else TSD_HIP(hipMemcpyPeerAsync(dst + (size_t) (need - take) * esz, h->dev[g], src, h->dev[p], (size_t) take * esz, hs));
        readers[(size_t) p].push_back(h->ev_halo[g]);
        need -= take;
      }
    }
    if (need > 0) TSD_HIP(hipMemcpyAsync(dst, h->carry.data() + (size_t) (H - need) * esz, (size_t) need * esz, hipMemcpyHostToDevice, hs));
    TSD_HIP(hipEventRecord(h->ev_halo[g], hs));
  }
  // the new carry: the tail of this call, read out of the last parts before any of them is overwritten in place --
  // asynchronously into page-locked memory (gathered into the handle only when the call has succeeded)
  std::vector<char> nc = h->carry;
  struct Piece { int64_t at, len; };
  std::vector<Piece> pieces;
  {
    int64_t need = std::min(H, n), kept = H - need;
    if (kept > 0 && need > 0) std::memmove(nc.data(), nc.data() + (size_t) need * esz, (size_t) kept * esz);
    int64_t fill = H;
    for (int p = N - 1; p >= 0 && need > 0; p--) {
      const int64_t take = std::min(need, counts[p]);
      if (take > 0) {
        TSD_HIP(hipSetDevice(h->dev[p]));
        const char *src = (const char *) x_parts[p] + (size_t) (counts[p] - take) * esz;
        rc = wait_ready(h->hstream[p], p);
        if (rc) return rc;
        if (h->pin_carry) {
          TSD_HIP(hipMemcpyAsync(h->pin_carry + (size_t) (fill - take) * esz, src, (size_t) take * esz, hipMemcpyDeviceToHost, h->hstream[p]));
          TSD_HIP(hipEventRecord(h->ev_tail[p], h->hstream[p]));
          readers[(size_t) p].push_back(h->ev_tail[p]);
          pieces.push_back(Piece{fill - take, take});
        } else {
          TSD_HIP(hipMemcpyAsync(nc.data() + (size_t) (fill - take) * esz, src, (size_t) take * esz, hipMemcpyDeviceToHost, h->hstream[p]));
          TSD_HIP(hipStreamSynchronize(h->hstream[p]));
        }
        fill -= take;
        need -= take;
      }
    }
  }
  // (2) the shards, concurrently (one stream each; the enqueue itself is cheap)
  std::vector<int64_t> got((size_t) N, 0);
  static const bool no_split = dev_switch("SHARD_NO_OVERLAP") != nullptr;     // A/B switch: wait for the halo, one launch
  for (int g = 0; g < N && !rc; g++) {
    TSD_HIP(hipSetDevice(h->dev[g]));
    hipStream_t st = h->stream[g];
    const int64_t pos = h->seen + lo[(size_t) g];
    const int64_t cnt = counts[g];
    const int64_t ycap = y_capacities ? y_capacities[g] : cnt;
    rc = wait_ready(st, g);
    if (rc) break;
    if (in_place(g))
      for (hipEvent_t e : readers[(size_t) g]) TSD_HIP(hipStreamWaitEvent(st, e, 0));
    const bool split = !no_split && H > 0 && cnt > H && h->edge[g] != nullptr && (h->kind == K_FIR || h->kind == K_RES);
    if (!split) {
      if (H > 0) TSD_HIP(hipStreamWaitEvent(st, h->ev_halo[g], 0));
      rc = run_shard(h, g, h->halo[g].p, std::min(H, pos), pos, x_parts[g], y_parts[g], cnt, ycap, &got[(size_t) g]);
      if (h->kind != K_RES) got[(size_t) g] = cnt;
      continue;
    }
    const char *xg = (const char *) x_parts[g];
    char *yg = (char *) y_parts[g];
    if (h->kind == K_FIR) {
      tsdgpu_fir *fm = (tsdgpu_fir *) h->handle[g], *fe = (tsdgpu_fir *) h->edge[g];
      // interior: the part's own head is its delay line -- read in place when the part is not filtered in place and is longer
      // than the handle's history (tsdgpu_fir_step_after: no copy, one launch), copied otherwise; the edge owes the outputs before it
      const int64_t lead = tsdgpu_fir_lead(fm);
      int64_t He = H;
      if (!in_place(g) && lead >= H && cnt > lead) {            // (lead = -1: partitioned plan, the copying form serves it)
        rc = tsdgpu_fir_step_after(fm, xg, yg, cnt, lead, st);
        He = lead;
      } else {
        rc = tsdgpu_fir_set_history(fm, xg, st);
        if (!rc) rc = tsdgpu_fir_step(fm, xg + (size_t) H * esz, yg + (size_t) H * esz, cnt - H, st);
      }
      if (rc) break;
      TSD_HIP(hipStreamWaitEvent(st, h->ev_halo[g], 0));                         // edge: the first outputs need the halo
      rc = tsdgpu_fir_set_history(fe, h->halo[g].p, st);
      if (!rc) rc = tsdgpu_fir_step(fe, xg, yg, He, st);
      got[(size_t) g] = cnt;
    } else {
      tsdgpu_resampler *rm = (tsdgpu_resampler *) h->handle[g], *re = (tsdgpu_resampler *) h->edge[g];
      // outputs of the first H inputs / of the whole part, from the schedule
      rc = tsdgpu_resampler_seek(re, pos, nullptr, st);
      if (rc) break;
      const int64_t o0 = tsdgpu_resampler_out_offset(re);
      rc = tsdgpu_resampler_seek(re, pos + H, nullptr, st);
      if (rc) break;
      const int64_t c_edge = tsdgpu_resampler_out_offset(re) - o0;
      int64_t g1 = 0, g0 = 0;
      if (c_edge > ycap) { rc = set_err(TSDGPU_ERR_INVALID, "sharded_step_parts: output capacity of part %d", g); break; }
      rc = tsdgpu_resampler_seek(rm, pos + H, xg, st);                           // interior: window = the part's own head
      if (!rc) rc = tsdgpu_resampler_step(rm, xg + (size_t) H * esz, cnt - H, yg + (size_t) c_edge * esz, ycap - c_edge, &g1, st);
      if (rc) break;
      TSD_HIP(hipStreamWaitEvent(st, h->ev_halo[g], 0));
      rc = tsdgpu_resampler_seek(re, pos, h->halo[g].p, st);
      if (!rc && c_edge > 0) rc = tsdgpu_resampler_step(re, xg, H, yg, c_edge, &g0, st);
      if (!rc && g0 != c_edge) rc = set_err(TSDGPU_ERR_INVALID, "shard %d: edge produced %lld outputs, the schedule says %lld", g, (long long) g0, (long long) c_edge);
      got[(size_t) g] = g0 + g1;
    }
  }
  for (int g = 0; g < N; g++) {
    (void) hipSetDevice(h->dev[g]);
    if (hipStreamSynchronize(h->stream[g]) != hipSuccess && !rc) rc = set_err(TSDGPU_ERR_HIP, "shard %d: stream sync failed", g);
    if (hipStreamSynchronize(h->hstream[g]) != hipSuccess && !rc) rc = set_err(TSDGPU_ERR_HIP, "shard %d: halo stream sync failed", g);
  }
  if (rc) return rc;
  for (const Piece &pc : pieces) std::memcpy(nc.data() + (size_t) pc.at * esz, h->pin_carry + (size_t) pc.at * esz, (size_t) pc.len * esz);
  int64_t total = 0;
  for (int g = 0; g < N; g++) {
    if (out_counts) out_counts[g] = got[(size_t) g];
    total += got[(size_t) g];
  }
  h->carry.swap(nc);                // committed with the position: a failed call consumes nothing
  h->seen += n;
  h->out_total += total;
  return TSDGPU_OK;
}

int tsdgpu_sharded_step_parts(tsdgpu_sharded *h, const void *const *x_parts, const int64_t *counts, void *const *y_parts,
                              const int64_t *y_capacities, int64_t *out_counts)
{
  return sharded_step_parts_impl(h, x_parts, counts, y_parts, y_capacities, out_counts, nullptr);
}

int tsdgpu_sharded_step_parts_on(tsdgpu_sharded *h, const void *const *x_parts, const int64_t *counts, void *const *y_parts,
                                 const int64_t *y_capacities, int64_t *out_counts, void *const *producer_streams)
{
  TSD_CHECK(producer_streams != nullptr, "sharded_step_parts_on: NULL stream list (tsdgpu_sharded_step_parts waits for the devices instead)");
  return sharded_step_parts_impl(h, x_parts, counts, y_parts, y_capacities, out_counts, producer_streams);
}

int tsdgpu_sharded_reset(tsdgpu_sharded *h)
{
  TSD_CHECK(h != nullptr, "sharded_reset: NULL handle");
  std::fill(h->carry.begin(), h->carry.end(), 0);
  std::fill(h->sos_state.begin(), h->sos_state.end(), 0.f);
  h->seen = 0;
  h->out_total = 0;
  for (int g = 0; g < h->nshards; g++) {
    const int rc = with_device(h->dev[g], [&]() -> int {
      if (h->kind == K_FIR) return tsdgpu_fir_reset((tsdgpu_fir *) h->handle[g]);
      if (h->kind == K_SOS) return tsdgpu_sos_reset((tsdgpu_sos *) h->handle[g]);
      return tsdgpu_resampler_reset((tsdgpu_resampler *) h->handle[g]);
    });
    if (rc) return rc;
  }
  return TSDGPU_OK;
}

int tsdgpu_sharded_destroy(tsdgpu_sharded *h)
{
  if (!h) return TSDGPU_OK;
  int prev = 0;
  (void) hipGetDevice(&prev);
  for (int g = 0; g < h->nshards; g++) {
    (void) hipSetDevice(h->dev[g]);
    if (h->handle[g]) {
      if (h->kind == K_FIR) tsdgpu_fir_destroy((tsdgpu_fir *) h->handle[g]);
      else if (h->kind == K_SOS) tsdgpu_sos_destroy((tsdgpu_sos *) h->handle[g]);
      else tsdgpu_resampler_destroy((tsdgpu_resampler *) h->handle[g]);
    }
    if (h->edge[g]) {
      if (h->kind == K_FIR) tsdgpu_fir_destroy((tsdgpu_fir *) h->edge[g]);
      else if (h->kind == K_RES) tsdgpu_resampler_destroy((tsdgpu_resampler *) h->edge[g]);
    }
    if (h->stream[g]) (void) hipStreamDestroy(h->stream[g]);
    if (h->hstream[g]) (void) hipStreamDestroy(h->hstream[g]);
    if (h->ev_ready[g]) (void) hipEventDestroy(h->ev_ready[g]);
    if (h->ev_halo[g]) (void) hipEventDestroy(h->ev_halo[g]);
    if (h->ev_tail[g]) (void) hipEventDestroy(h->ev_tail[g]);
    h->halo[g].release();
    h->in[g].release();
    h->out[g].release();
    h->scratch[g].release();
  }
  if (h->pin_carry) (void) hipHostFree(h->pin_carry);
  (void) hipSetDevice(prev);
  delete h;                              // (joins the shard workers)
  return TSDGPU_OK;
}

}  // extern "C"
