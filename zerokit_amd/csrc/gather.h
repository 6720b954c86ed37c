// gather.h -- single calls from several threads gathered into batches (no HIP, no FFI types: tests/host/gather_tsan.cpp
// runs it under ThreadSanitizer).
#pragma once
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

namespace rlnamd {

// generate_rln_proof takes &self and may be called from several threads (public.rs:624); the prover has one set of
// workspaces per batch in flight, so single-proof calls used to take turns: T threads got the rate of one (1.2 k
// proofs/s at 0.8 ms per call).  Now the calls that arrive while a proof is on the device are gathered: the first
// caller that finds no batch running leads -- it takes everything queued (its own request included), proves it as ONE
// batch, hands every request its proof or its own error text, and steps down; a caller whose request went out with
// somebody else's batch just wakes up with its result.  A lone caller leads a batch of one: the path it always took.
// Req: what a caller queues; it needs a member `bool done` (set under the queue's lock when its batch has been run)
// and `void gather_failed()` (called for every request of a batch whose run threw: leave an error unless there is a result).
template <class Req>
struct GatherQueue {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Req*> q;
  bool leader = false;
  size_t most = 0;                        // 0: off
  uint64_t batches = 0, calls = 0, largest = 0;
  // Threads that call in a loop arrive just behind their results, after the next leader has taken its batch.  The
  // leader therefore knows who else called within the last 20 ms and gives those callers window_us to arrive
  // (spinning, off the lock) before it takes the batch.  A caller that is alone never waits.
  std::vector<std::pair<std::thread::id, std::chrono::steady_clock::time_point>> recent;
  long window_us = 500;                   // RLNAMD_GATHER_WINDOW_US / "gather_window_us"; 0: take what is there
  uint64_t misses = 0, no_wait_until = 0; // the wait's back-off (see pass)
  uint64_t waited = 0;                    // batches whose leader waited for a recent caller
  uint64_t busy_ns = 0;                   // time the leaders spent proving their batches
  // queue `me`, lead or follow until it is done; run(batch) proves a batch and never throws
  template <class Run>
  void pass(Req& me, Run&& run) {
    {
      std::unique_lock<std::mutex> lk(mu);
      q.push_back(&me);
      {   // this thread among the recent callers
        const auto now = std::chrono::steady_clock::now();
        bool seen = false;
        for (auto& e : recent)
          if (e.first == std::this_thread::get_id()) {
            e.second = now;
            seen = true;
          }
        if (!seen && recent.size() < 256) recent.emplace_back(std::this_thread::get_id(), now);
      }
      while (!me.done) {
        if (leader) {
          cv.wait(lk);
          continue;
        }
        leader = true;   // nobody is proving: lead, with everything that is queued now
        if (window_us > 0 && batches >= no_wait_until) {
          // Threads that call in a loop come back just behind their results.  Without a wait they split into two halves
          // that take turns (one half on the device while the other gathers: T threads, batches of T / 2, two batch times
          // per call); with it the leader gives everybody it saw within the last 20 ms window_us to arrive and the T calls
          // go out together -- one (longer) batch time per call.  Callers that stopped cost a few leaders the window until
          // they age out; callers that are slower than the window (an interpreter between the calls) make the leader
          // give up waiting for the next 64 batches after three misses in a row.
          const auto t0 = std::chrono::steady_clock::now();
          size_t expect = 0;
          for (size_t i = 0; i < recent.size();) {
            if (t0 - recent[i].second > std::chrono::milliseconds(20)) {
              recent[i] = recent.back();
              recent.pop_back();
            } else {
              expect++;
              i++;
            }
          }
          expect = std::min(expect, most);
          if (q.size() < expect) {
            waited++;
            const auto until = t0 + std::chrono::microseconds(window_us);
            while (q.size() < expect && std::chrono::steady_clock::now() < until) {
              lk.unlock();
              std::this_thread::yield();
              lk.lock();
            }
            if (q.size() >= expect) {
              misses = 0;
            } else if (++misses >= 3) {
              misses = 0;
              no_wait_until = batches + 64;
            }
          }
        }
        std::vector<Req*> batch;
        try {
          batch.reserve(std::min(q.size(), most));
        } catch (...) {      // (out of memory before anything was taken: step down, this call fails, the others go on)
          leader = false;
          for (auto it = q.begin(); it != q.end(); ++it)
            if (*it == &me) {
              q.erase(it);
              break;
            }
          cv.notify_all();
          throw;
        }
        while (!q.empty() && batch.size() < most) {
          batch.push_back(q.front());   // (reserved: cannot throw)
          q.pop_front();
        }
        lk.unlock();
        const auto t_run = std::chrono::steady_clock::now();
        try {
          run(batch);
        } catch (...) {      // (run catches what proving throws; this is for its own allocations)
          for (Req* r : batch) r->gather_failed();
        }
        lk.lock();
        for (Req* r : batch) r->done = true;   // (not touched again: its owner may return now)
        busy_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_run).count();
        batches++;
        calls += batch.size();
        largest = std::max<uint64_t>(largest, batch.size());
        leader = false;
        cv.notify_all();
      }
    }
  }
};

}  // namespace rlnamd
