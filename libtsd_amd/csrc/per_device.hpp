// per_device.hpp -- resources that belong to ONE device (events, page-locked staging blocks with their events, plan
// tables) must never be handed to a caller whose current device is another one: an event of device 0 recorded on a
// stream of device 1 is an invalid resource handle.  Plain C++ (no HIP call): the keying is unit-tested on the CPU
// (tests/cpu/test_per_device.cc).
#pragma once
#include <map>
#include <mutex>
#include <vector>

namespace tsdgpu {

// Free list per device.  T needs a public `int dev`.  make(dev) builds a new T for that device (may return nullptr).
template <typename T> struct PerDevicePool {
  std::mutex m;
  std::map<int, std::vector<T *>> libres;
  template <typename Make> T *prend(int dev, Make make)
  {
    {
      std::lock_guard<std::mutex> l(m);
      auto &v = libres[dev];
      if (!v.empty()) {
        T *p = v.back();
        v.pop_back();
        return p;
      }
    }
    T *p = make(dev);
    if (p) p->dev = dev;
    return p;
  }
  void rend(T *p)
  {
    if (!p) return;
    std::lock_guard<std::mutex> l(m);
    libres[p->dev].push_back(p);
  }
};

// What ONE thread has borrowed from a PerDevicePool, one item per device it has worked on; everything goes back to the
// pool when the thread ends.  `tried` remembers a device whose make() failed so that it is not retried at every call.
template <typename T> struct PerDeviceHeld {
  PerDevicePool<T> *pool = nullptr;
  std::map<int, T *> held;
  ~PerDeviceHeld()
  {
    if (pool)
      for (auto &kv : held) pool->rend(kv.second);
  }
  template <typename Make> T *get(PerDevicePool<T> &from, int dev, Make make)
  {
    pool = &from;
    auto it = held.find(dev);
    if (it != held.end()) return it->second;
    T *p = from.prend(dev, make);
    held[dev] = p;
    return p;
  }
};

}  // namespace tsdgpu
