// Host-only check of OrbxWorkPool (orbx_internal.hpp): built and run by tests/test_workpool.py.
#include <cstdio>
#include <new>
#include <vector>

#include "../orb-slam3-rust_amd/csrc/orbx_internal.hpp"

int main() {
  int bad = 0;
  for (int workers : {0, 1, 3, 7}) {
    OrbxWorkPool pool(workers);
    if (pool.workers() != workers) ++bad;
    for (int round = 0; round < 200; ++round) {
      const int items = (round * 37) % 101, want = round % (workers + 2);     // want may exceed the pool: clamped
      std::vector<std::atomic<int>> hits(items);
      for (auto& x : hits) x.store(0);
      const bool ok = pool.run(items, want, [&](int i) { hits[i].fetch_add(1); });
      if (!ok) ++bad;
      for (int i = 0; i < items; ++i) if (hits[i].load() != 1) ++bad;
    }
    // an exception inside the job is reported, not thrown, and the pool keeps working
    std::atomic<int> done{0};
    const bool ok = pool.run(64, workers, [&](int i) { if (i == 17) throw std::bad_alloc(); done.fetch_add(1); });
    if (ok) ++bad;
    std::atomic<int> again{0};
    if (!pool.run(50, workers, [&](int) { again.fetch_add(1); }) || again.load() != 50) ++bad;
  }
  std::printf("workpool %s\n", bad ? "FAILED" : "ok");
  return bad ? 1 : 0;
}
