// CPU test of the host driver's collective (altair-raytracing_amd/host/isx_comm.*): N "ranks" = N threads joined by an
// in-process Transport double.  Checks (a) the semantics of reduce_collective (sums, MAX of the times, worst status) through ONE sum, (b) the ADVICE r01 case: ONE rank
// fails before the collective -> every rank still enters it, nobody blocks, every rank returns the same error.
// Built and run by tests/test_host_driver.py::test_collective_status_with_a_failing_rank (no GPU needed).
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../altair-raytracing_amd/host/isx_comm.hpp"

using namespace isxhost;

struct Hub {
  int world;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0, generation = 0, exchanges = 0;
  std::vector<unsigned long long> acc;
};

// ONE sum per collective: the double counts the exchanges, so the test can assert "a single all-reduce per call"
struct Loopback : Transport {
  Hub& h;
  explicit Loopback(Hub& hub) : h(hub) {}
  bool exchange_sum(unsigned long long* buf, size_t n) override {
    std::unique_lock<std::mutex> lk(h.m);
    if (h.arrived == 0) { h.acc.assign(buf, buf + n); h.exchanges++; }
    else {
      if (n != h.acc.size()) return false;   // ranks disagree on the layout
      for (size_t k = 0; k < n; ++k) h.acc[k] += buf[k];
    }
    const int gen = h.generation;
    if (++h.arrived == h.world) { h.arrived = 0; h.generation++; h.cv.notify_all(); }
    else h.cv.wait(lk, [&] { return h.generation != gen; });
    std::memcpy(buf, h.acc.data(), n * sizeof(unsigned long long));
    return true;
  }
};

static int run_case(int world, int failing_rank, int fail_code) {
  Hub hub; hub.world = world;
  std::vector<int> rc((size_t)world, 12345);
  std::vector<std::vector<uint64_t>> hits((size_t)world, std::vector<uint64_t>(5));
  std::vector<isx_stats> st((size_t)world);
  std::vector<std::thread> th;
  for (int r = 0; r < world; ++r) {
    for (int k = 0; k < 5; ++k) hits[(size_t)r][(size_t)k] = (uint64_t)(100 * (r + 1) + k);
    std::memset(&st[(size_t)r], 0, sizeof(isx_stats));
    st[(size_t)r].launched = 10 + (uint64_t)r; st[(size_t)r].wall_hits = 1000; st[(size_t)r].t_kernel_ms = 1.5 * (r + 1);
    th.emplace_back([&, r] {
      Loopback t(hub);
      rc[(size_t)r] = reduce_collective(t, r, world, r == failing_rank ? fail_code : ISX_OK, hits[(size_t)r].data(), 5, &st[(size_t)r], 1);
    });
  }
  for (auto& t : th) t.join();   // a rank left waiting in the collective would hang here (the test runs under a time-out)
  int bad = 0;
  if (hub.exchanges != 1) { std::printf("%d exchanges for one collective (SURVEY.md 8e: ONE all-reduce)\n", hub.exchanges); bad++; }
  for (int r = 0; r < world; ++r) {
    if (failing_rank >= 0) {
      if (rc[(size_t)r] != fail_code) { std::printf("rank %d: rc %d, expected %d\n", r, rc[(size_t)r], fail_code); bad++; }
      // outputs untouched on failure
      if (hits[(size_t)r][0] != (uint64_t)(100 * (r + 1))) { std::printf("rank %d: hits modified on failure\n", r); bad++; }
    } else {
      if (rc[(size_t)r] != ISX_OK) { std::printf("rank %d: rc %d\n", r, rc[(size_t)r]); bad++; }
      uint64_t want0 = 0, wantL = 0;
      for (int q = 0; q < world; ++q) { want0 += (uint64_t)(100 * (q + 1)); wantL += 10 + (uint64_t)q; }
      if (hits[(size_t)r][0] != want0 || st[(size_t)r].launched != wantL || st[(size_t)r].wall_hits != 1000ull * (uint64_t)world) {
        std::printf("rank %d: wrong sums\n", r); bad++;
      }
      if (st[(size_t)r].t_kernel_ms < 1.5 * world - 1e-3 || st[(size_t)r].t_kernel_ms > 1.5 * world + 1e-3) {
        std::printf("rank %d: t_kernel_ms %f is not the max\n", r, st[(size_t)r].t_kernel_ms); bad++;
      }
    }
  }
  return bad;
}

int main() {
  int bad = 0;
  bad += run_case(2, -1, 0);
  bad += run_case(8, -1, 0);
  bad += run_case(2, 1, ISX_ERR_BAD_CONFIG);
  bad += run_case(2, 0, ISX_ERR_HIP);
  bad += run_case(8, 5, ISX_ERR_TOO_LARGE);
  bad += run_case(3, 2, ISX_ERR_NO_DEVICE);
  std::printf(bad ? "FAILED %d\n" : "OK\n", bad);
  return bad ? 1 : 0;
}
