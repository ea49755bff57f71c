// CPU unit test of the per-device keying: libtsd_amd/csrc/per_device.hpp (bounce blocks of the small-host-call
// path) and libtsd_amd/host/adaptors/reserve_plans.hpp (FFT plan reserves).  A resource made under device d must only
// ever be handed to a caller whose current device is d.
#include <cstdio>
#include <thread>
#include "../../libtsd_amd/csrc/per_device.hpp"
#include "../../libtsd_amd/host/adaptors/reserve_plans.hpp"

struct Bloc { int dev = -1; int serial = 0; };
static int made = 0;
static Bloc *fabrique(int) { Bloc *b = new Bloc(); b->serial = ++made; return b; }

struct Plan { int dev; long n; };
static int detruits = 0;
static void detruit(Plan *p) { detruits++; delete p; }

#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main()
{
  tsdgpu::PerDevicePool<Bloc> pool;
  Bloc *first0 = nullptr, *first1 = nullptr;
  {
    tsdgpu::PerDeviceHeld<Bloc> held;
    Bloc *a = held.get(pool, 0, fabrique);
    CHECK(a && a->dev == 0);
    CHECK(held.get(pool, 0, fabrique) == a);            // same thread, same device: the same block
    Bloc *b = held.get(pool, 1, fabrique);              // the thread moves to device 1: a block of device 1, not `a`
    CHECK(b && b != a && b->dev == 1);
    CHECK(held.get(pool, 0, fabrique) == a);            // back on device 0
    first0 = a; first1 = b;
  }                                                       // thread "ends": both blocks go back to their device's list
  CHECK(made == 2);
  {
    tsdgpu::PerDeviceHeld<Bloc> held;
    CHECK(held.get(pool, 1, fabrique) == first1);       // recycled per device
    CHECK(held.get(pool, 0, fabrique) == first0);
    CHECK(made == 2);
    Bloc *c = nullptr;
    std::thread t([&] { tsdgpu::PerDeviceHeld<Bloc> h2; c = h2.get(pool, 0, fabrique); });   // another thread, same device: its own block
    t.join();
    CHECK(c && c != first0 && c->dev == 0 && made == 3);
  }
  // a device whose make() fails is remembered (no retry per call) and served as "no block"
  {
    tsdgpu::PerDeviceHeld<Bloc> held;
    int tries = 0;
    auto echoue = [&](int) -> Bloc * { tries++; return nullptr; };
    CHECK(held.get(pool, 7, echoue) == nullptr && held.get(pool, 7, echoue) == nullptr && tries == 1);
  }

  tsd_amd::ReserveParCle<Plan> res(detruit, 2, 3);
  res.rend(0, 1024, new Plan{0, 1024});
  CHECK(res.prend(1, 1024) == nullptr);                 // a plan of device 0 is never handed to device 1
  CHECK(res.prend(0, 512) == nullptr);
  Plan *p = res.prend(0, 1024);
  CHECK(p && p->dev == 0 && p->n == 1024 && res.total == 0);
  res.rend(0, 1024, p);
  res.rend(1, 1024, new Plan{1, 1024});
  Plan *q = res.prend(1, 1024);
  CHECK(q && q->dev == 1);
  res.rend(1, 1024, q);
  res.rend(0, 1024, new Plan{0, 1024});                 // 2 per key, 3 in all: this one is kept ...
  CHECK(res.total == 3 && detruits == 0);
  res.rend(0, 1024, new Plan{0, 1024});                 // ... this one destroyed (key full)
  res.rend(2, 64, new Plan{2, 64});                     // ... and this one too (reserve full)
  CHECK(res.total == 3 && detruits == 2);
  res.rend(0, 1, nullptr);                              // a null handle is ignored
  std::printf("OK\n");
  return 0;
}
