// ThreadSanitizer run of the queue that gathers single calls into batches (zerokit_amd/csrc/gather.h; the FFI's
// prove_one / finish_one).  T threads call in a loop; `run` stands for the device: it takes a while and fills in every
// request of its batch.  Checked: every call gets ITS result, no call is lost or served twice, batches never exceed the
// cap, only one batch runs at a time, a run that throws leaves every request of its batch with the failure mark, the
// leader's wait for recent callers makes T looping threads go out as batches of T; TSan must stay silent.
// Build: g++ -O1 -g -std=c++17 -fsanitize=thread -I zerokit_amd/csrc tests/host/gather_tsan.cpp -o gather_tsan -lpthread
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <thread>
#include <vector>

#include "gather.h"

struct Req {
  long in = 0, out = -1;
  bool done = false, failed = false;
  void gather_failed() { failed = true; }
};

static int fails = 0;
#define CHECK(c, msg)                                   \
  do {                                                  \
    if (!(c)) {                                         \
      fails++;                                          \
      fprintf(stderr, "FAIL: %s (line %d)\n", msg, __LINE__); \
    }                                                   \
  } while (0)

int main() {
  for (long window : {0l, 200l}) {
    for (int T : {1, 2, 3, 8, 24}) {
      rlnamd::GatherQueue<Req> G;
      G.most = 16;
      G.window_us = window;
      std::atomic<int> running{0}, overlap{0}, served{0}, too_big{0};
      std::atomic<long> wrong{0};
      auto run = [&](const std::vector<Req*>& batch) {
        if (running.fetch_add(1) != 0) overlap++;
        if (batch.size() > G.most) too_big++;
        std::this_thread::sleep_for(std::chrono::microseconds(300));   // "the device"
        for (Req* r : batch) {
          if (r->in % 97 == 13) throw std::runtime_error("a run that fails as a whole");
          r->out = 2 * r->in + 1;
          served++;
        }
        running.fetch_sub(1);
      };
      const int calls = 200;
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
          for (int j = 0; j < calls; j++) {
            Req me;
            me.in = 1000L * t + j;
            try {
              G.pass(me, [&](const std::vector<Req*>& b) {
                try {
                  run(b);
                } catch (...) {
                  running.fetch_sub(1);
                  throw;
                }
              });
            } catch (...) {
              wrong++;
            }
            if (!me.done) wrong++;
            if (me.failed) {
              // its batch threw: every request of that batch carries the mark; a result may or may not have been written
            } else if (me.out != 2 * me.in + 1) {
              wrong++;
            }
          }
        });
      for (auto& x : th) x.join();
      CHECK(wrong == 0, "every call got its own result or its batch's failure mark");
      CHECK(overlap == 0, "one batch at a time");
      CHECK(too_big == 0, "no batch above the cap");
      CHECK(G.calls == (uint64_t)T * calls, "every call went out in exactly one batch");
      CHECK(G.q.empty() && !G.leader, "nothing left behind");
      if (T == 1) CHECK(G.largest == 1 && G.waited == 0, "a lone caller is a batch of one and never waits");
      if (T == 8 && window) CHECK(G.largest >= 6, "looping threads go out together when the leader waits for them");
      printf("window %ld us, %2d threads: %llu calls in %llu batches, largest %llu, waited %llu\n", window, T,
             (unsigned long long)G.calls, (unsigned long long)G.batches, (unsigned long long)G.largest, (unsigned long long)G.waited);
    }
  }
  printf("%d failures\n", fails);
  return fails ? 1 : 0;
}
