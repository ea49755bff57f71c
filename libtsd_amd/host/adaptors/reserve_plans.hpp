// reserve_plans.hpp -- small reserve of finished device plans, keyed by (device, size): a plan's tables live on the device
// it was created on, so it may only be handed to a caller whose current device is that one.  Plain C++ (the destroy
// function is a parameter): unit-tested on the CPU (tests/cpu/test_per_device.cc).
#pragma once
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace tsd_amd {

template <typename H> struct ReserveParCle {
  typedef void (*Detruit)(H *);
  std::mutex m;
  std::map<std::pair<int, long>, std::vector<H *>> libres;
  size_t total = 0;
  Detruit detruit;
  size_t par_cle, en_tout;
  explicit ReserveParCle(Detruit d, size_t par_cle_ = 4, size_t en_tout_ = 64) : detruit(d), par_cle(par_cle_), en_tout(en_tout_) {}
  H *prend(int dev, long n)
  {
    std::lock_guard<std::mutex> l(m);
    auto it = libres.find(std::make_pair(dev, n));
    if (it == libres.end() || it->second.empty()) return nullptr;
    H *h = it->second.back();
    it->second.pop_back();
    total--;
    return h;
  }
  void rend(int dev, long n, H *h)
  {
    if (!h) return;
    {
      std::lock_guard<std::mutex> l(m);
      auto &v = libres[std::make_pair(dev, n)];
      if (v.size() < par_cle && total < en_tout) {        // a few plans per (device, size), a few dozen in all
        v.push_back(h);
        total++;
        return;
      }
    }
    detruit(h);
  }
};

}  // namespace tsd_amd
