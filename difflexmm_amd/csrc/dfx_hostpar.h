// dfx_hostpar.h -- the one piece of host-side parallelism libdfx has: independent items (ensemble members, designs) on a handful of threads,
// one contiguous chunk each.  No OpenMP runtime in the library; small jobs stay on the calling thread.
#pragma once
#include <stddef.h>

#include <algorithm>
#include <thread>
#include <vector>

namespace dfx_hostpar {

template <class F>
inline void for_each(int n_items, size_t work_per_item, F&& body) {
  const unsigned hw = std::thread::hardware_concurrency();
  int nt = (int)std::min<unsigned>(hw ? hw : 1u, 16u);
  nt = std::max(1, std::min(nt, n_items));
  if (nt == 1 || work_per_item * (size_t)n_items < 200000) { for (int m = 0; m < n_items; ++m) body(m); return; }
  std::vector<std::thread> th;
  th.reserve(nt);
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t]() { for (int m = (int)((long long)n_items * t / nt); m < (int)((long long)n_items * (t + 1) / nt); ++m) body(m); });
  for (auto& x : th) x.join();
}

}  // namespace dfx_hostpar
