// Native multi-GPU side of include/rln_amd.h (BASELINE north_star: "batches of independent proofs, and
// optionally a single large MSM, shard across the 8 GPUs of one node"):
//
//   rlnamd_pool  -- one Prover replica + one host thread per device (hipSetDevice is per-thread state).  A job of n
//                   proofs is cut into contiguous index shards, one per replica (BASELINE config 4: 8 x 8 192), every
//                   replica streams its shard through Prover::prove_stream (fresh inputs H2D, proofs D2H, chunks
//                   overlapping on the device) and writes its results at their index.  No data-path collective: the
//                   proofs are independent (the reference's guidance is one worker per proof, rln/README.md:324-332).
//   rlnamd_comm  -- an RCCL communicator (multi-process: unique id + rank; single process: ncclCommInitAll) for the
//                   one path that has an exchange step, the config-5 MSM: rlnamd_msm_run_sharded = local Pippenger to
//                   16 window sums, ONE ncclAllGather of 2 KiB per rank over xGMI, local add + fold.
#include <rccl/rccl.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "capi_util.h"
#include "common.h"

using namespace rlnamd;

#define RLN_NCCL(expr)                                                                                       \
  do {                                                                                                       \
    ncclResult_t r_ = (expr);                                                                                \
    if (r_ != ncclSuccess) throw Error(std::string("RCCL error: ") + ncclGetErrorString(r_) + " (" #expr ")"); \
  } while (0)

struct rlnamd_comm {
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0, device = 0;
};
void* rlnamd_comm_handle(rlnamd_comm* c) { return (void*)c->comm; }
rlnamd::Prover* rlnamd_pool_replica_prover(rlnamd_pool* p, size_t replica);
int rlnamd_comm_size(rlnamd_comm* c) { return c->nranks; }

namespace {

struct Job {
  size_t n = 0;
  const uint8_t* inputs = nullptr;
  const uint8_t* rs = nullptr;
  uint8_t* proofs = nullptr;
  uint8_t* values = nullptr;
  uint32_t* errors = nullptr;
  int mode = 0;   // 0: the contiguous shard [lo, hi); 1: the shared cursor over [0, n); 2: the shared cursor over the pool's redo list
};

using Chunk = std::pair<size_t, size_t>;   // first proof, proofs

struct Replica {
  int device = 0;
  std::unique_ptr<Prover> prover;
  std::thread th;
  // the worker's mailbox
  std::mutex mu;
  std::condition_variable cv;
  bool has_job = false, stop = false, ready = false;
  Job job;
  size_t lo = 0, hi = 0;
  std::string error;   // of construction or of the last job
  float last_ms = 0;   // wall time of the last shard on this replica
  size_t last_proofs = 0, chunks_taken = 0;   // proofs of the last job this replica made; chunks it was handed (all jobs)
  // failover (rlnamd_pool_set_failover): what the last dispatch handed this replica, and what it never reached of its
  // static shard -- if the dispatch failed, all of it is proved again elsewhere (the writes are idempotent)
  std::vector<Chunk> taken;
  size_t rest_lo = 0, rest_hi = 0;
  size_t failures = 0;        // dispatches that ended in an error (pool lifetime)
  bool quarantined = false;   // failed under failover: handed nothing until rlnamd_pool_revive (or its probation ends)
  size_t sat_out = 0;         // jobs it has sat out since it was quarantined
};

}  // namespace

struct rlnamd_pool {
  std::vector<std::unique_ptr<Replica>> rep;
  std::mutex job_mu;          // one job at a time
  std::mutex done_mu;
  std::condition_variable done_cv;
  size_t pending = 0;
  size_t inputs_size = 0;
  // Assignment of a job's proofs to the replicas.  static (default): contiguous index shards, one per replica (SURVEY
  // 8e; equal devices finish together and a shard streams through every workspace slot).  dynamic: a cursor shared by
  // the replicas over chunks of capacity() proofs -- a replica takes the next chunk whenever one of its (at most three)
  // in-flight places is free, so a device that runs slower (a box-to-box clock spread of +- 3 % was measured, a
  // throttled or shared device is worse) simply takes fewer.  The result is index-identical either way.
  bool dynamic = false;
  std::atomic<size_t> cursor{0};
  // test hook (rlnamd_pool_inject_fault): replica `fault_replica` throws when it is handed its `fault_after`-th chunk
  // (counted over the pool's lifetime); one shot
  std::atomic<long> fault_replica{-1};
  size_t fault_after = 0;
  // Failover (off by default: a failing replica fails the job, and the caller sees which device and why).  With
  // failover_rounds > 0 the chunks of a failed replica -- everything it was handed in the dispatch, finished or not, and
  // what it had not reached of its static shard -- go onto a redo list that the surviving replicas draw from, up to that
  // many times per job; the failed replica is quarantined (rlnamd_pool_health, rlnamd_pool_revive).  The job fails only
  // when no replica is left or the rounds are spent.
  int failover_rounds = 0;
  // rlnamd_pool_set_probation: a quarantined replica sits out this many jobs and is then handed work again without a
  // call to rlnamd_pool_revive (a transient error must not remove a device for the pool's lifetime: callers behind the
  // FFI object have no handle on the pool); 0 = quarantine until revived.  A replica that fails again is quarantined
  // again, and its chunks go to the others as before.
  size_t probation_jobs = 0;
  std::vector<Chunk> redo;
  std::atomic<size_t> redo_cursor{0};

  void worker(Replica* R, const uint8_t* zkey, size_t zkey_len, const uint8_t* graph, size_t graph_len, ProverConfig cfg) {
    try {
      RLN_HIP(hipSetDevice(R->device));
      R->prover.reset(new Prover(zkey, zkey_len, graph, graph_len, cfg));
    } catch (const std::exception& e) {
      R->error = e.what();
    } catch (...) {
      R->error = "unknown exception while building the replica";
    }
    {
      std::lock_guard<std::mutex> lk(done_mu);
      R->ready = true;
    }
    done_cv.notify_all();
    for (;;) {
      std::unique_lock<std::mutex> lk(R->mu);
      R->cv.wait(lk, [&] { return R->has_job || R->stop; });
      if (R->stop) return;
      Job j = R->job;
      size_t lo = R->lo, hi = R->hi;
      R->has_job = false;
      lk.unlock();
      R->error.clear();
      R->taken.clear();
      R->rest_lo = R->rest_hi = 0;
      size_t made = 0;
      try {
        const int mode = j.mode;
        if (mode != 0 || hi > lo) {
          auto t0 = std::chrono::steady_clock::now();
          const size_t cap = R->prover->capacity();
          size_t at = lo;
          R->rest_lo = lo;
          R->rest_hi = hi;
          auto next = [&](size_t* off, size_t* cnt) {
            size_t o;
            if (mode == 1) {
              o = cursor.fetch_add(cap);
              if (o >= j.n) return false;
              *cnt = std::min(cap, j.n - o);
            } else if (mode == 2) {
              const size_t k = redo_cursor.fetch_add(1);
              if (k >= redo.size()) return false;
              o = redo[k].first;
              *cnt = redo[k].second;
            } else {
              if (at >= hi) return false;
              o = at;
              *cnt = std::min(cap, hi - at);
              at += *cnt;
              R->rest_lo = at;
            }
            *off = o;
            R->taken.emplace_back(o, *cnt);   // (before the fault hook: a chunk that was drawn from a shared cursor is this replica's to account for)
            long fr = fault_replica.load();   // ONE load: another worker may clear it between two
            if (fr >= 0 && (size_t)fr < rep.size() && rep[(size_t)fr].get() == R && R->chunks_taken >= fault_after &&
                fault_replica.compare_exchange_strong(fr, -1))   // fires exactly once
              throw Error("injected fault (rlnamd_pool_inject_fault)");
            R->chunks_taken++;
            made += *cnt;
            return true;
          };
          R->prover->prove_stream_from(next, j.inputs, j.rs, j.proofs, j.values, j.errors, mode == 0 ? 0 : 3);
          R->last_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        R->last_proofs += made;
      } catch (const std::exception& e) {
        R->error = e.what();
      } catch (...) {
        R->error = "unknown exception in the replica's worker";
      }
      if (!R->error.empty()) R->failures++;
      {
        std::lock_guard<std::mutex> dl(done_mu);
        pending--;
      }
      done_cv.notify_all();
    }
  }

  ~rlnamd_pool() {
    for (auto& R : rep) {
      {
        std::lock_guard<std::mutex> lk(R->mu);
        R->stop = true;
      }
      R->cv.notify_all();
    }
    for (auto& R : rep)
      if (R->th.joinable()) R->th.join();
    // the Prover of a replica frees device memory: do it with its device current, and leave the CALLER's device as it
    // was (hipSetDevice is per-thread state; the thread that frees an RLN object goes on using its own device)
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (auto& R : rep) {
      (void)hipSetDevice(R->device);
      R->prover.reset();
    }
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

rlnamd::Prover* rlnamd_pool_replica_prover(rlnamd_pool* p, size_t replica) {
  return replica < p->rep.size() ? p->rep[replica]->prover.get() : nullptr;
}

extern "C" {

int rlnamd_pool_new(const uint8_t* zkey, size_t zkey_len, const uint8_t* graph, size_t graph_len, size_t max_batch,
                    int window_bits, const int* devices, size_t n_devices, rlnamd_pool** out) {
  RLN_TRY
  require_gpu();
  int have = 0;
  RLN_HIP(hipGetDeviceCount(&have));
  std::vector<int> devs;
  if (!devices || n_devices == 0) {
    for (int d = 0; d < have; d++) devs.push_back(d);   // every device of the node
  } else {
    devs.assign(devices, devices + n_devices);
  }
  for (int d : devs)
    if (d < 0 || d >= have) throw Error("rlnamd_pool_new: device " + std::to_string(d) + " does not exist (" +
                                        std::to_string(have) + " visible)");
  int prev = 0;
  RLN_HIP(hipGetDevice(&prev));
  ProverConfig cfg;
  cfg.max_batch = max_batch ? max_batch : 1024;
  cfg.window_bits = window_bits;
  std::unique_ptr<rlnamd_pool> P(new rlnamd_pool);
  for (int d : devs) {
    P->rep.emplace_back(new Replica);
    P->rep.back()->device = d;
  }
  // the replicas build their tables in parallel (one thread per device; ~10 s for the 228 GiB schedule)
  for (auto& R : P->rep) {
    Replica* r = R.get();
    rlnamd_pool* pp = P.get();
    r->th = std::thread([pp, r, zkey, zkey_len, graph, graph_len, cfg] { pp->worker(r, zkey, zkey_len, graph, graph_len, cfg); });
  }
  {
    std::unique_lock<std::mutex> lk(P->done_mu);
    P->done_cv.wait(lk, [&] {
      for (auto& R : P->rep)
        if (!R->ready) return false;
      return true;
    });
  }
  (void)hipSetDevice(prev);
  for (auto& R : P->rep)
    if (!R->error.empty()) throw Error("rlnamd_pool_new: device " + std::to_string(R->device) + ": " + R->error);
  P->inputs_size = P->rep[0]->prover->inputs_per_proof();
  *out = P.release();
  RLN_CATCH
}

void rlnamd_pool_free(rlnamd_pool* p) { delete p; }
size_t rlnamd_pool_size(rlnamd_pool* p) { return p->rep.size(); }
int rlnamd_pool_device(rlnamd_pool* p, size_t replica) { return replica < p->rep.size() ? p->rep[replica]->device : -1; }

int rlnamd_pool_get_info(rlnamd_pool* p, rlnamd_prover_info* info) {
  RLN_TRY
  fill_prover_info(*p->rep[0]->prover, info);
  RLN_CATCH
}

int rlnamd_pool_prove(rlnamd_pool* p, size_t n, const uint8_t* inputs_le, const uint8_t* rs_le, uint8_t* proofs,
                      uint8_t* values, uint32_t* errors) {
  RLN_TRY
  if (n == 0) return RLNAMD_OK;
  if (!inputs_le || !rs_le) throw Error("rlnamd_pool_prove: inputs and rs are required");
  std::lock_guard<std::mutex> job_lk(p->job_mu);
  std::vector<Replica*> team;   // who is handed work in this round
  for (auto& R : p->rep) {
    R->last_proofs = 0;
    if (R->quarantined && p->probation_jobs > 0 && R->sat_out >= p->probation_jobs) R->quarantined = false;
    if (!R->quarantined) team.push_back(R.get());
    else R->sat_out++;
  }
  if (team.empty() && p->probation_jobs > 0)   // nobody left and probation is on: everybody gets another chance now
    for (auto& R : p->rep) {
      R->quarantined = false;
      team.push_back(R.get());
    }
  if (team.empty()) throw Error("rlnamd_pool_prove: every replica is quarantined (rlnamd_pool_revive)");
  p->cursor.store(0);
  std::string first_error;
  for (int round = 0;; round++) {
    const size_t N = team.size();
    {
      std::lock_guard<std::mutex> dl(p->done_mu);
      p->pending = N;
    }
    // round 0, static: contiguous shards by index, in units of whole waves (64 proofs) so that no replica pads more than
    // the last one; later rounds: the redo list
    const size_t waves = (n + 63) / 64;
    for (size_t i = 0; i < N; i++) {
      Replica& R = *team[i];
      size_t lo = 0, hi = 0;
      if (round == 0 && !p->dynamic) {
        lo = std::min(n, (waves * i / N) * 64);
        hi = std::min(n, (waves * (i + 1) / N) * 64);
        if (i + 1 == N) hi = n;
      }
      {
        std::lock_guard<std::mutex> lk(R.mu);
        R.job = Job{n, inputs_le, rs_le, proofs, values, errors, round > 0 ? 2 : p->dynamic ? 1 : 0};
        R.lo = lo;
        R.hi = hi;
        R.has_job = true;
      }
      R.cv.notify_all();
    }
    {
      std::unique_lock<std::mutex> lk(p->done_mu);
      p->done_cv.wait(lk, [&] { return p->pending == 0; });
    }
    std::vector<Chunk> again;
    std::vector<Replica*> alive;
    for (Replica* R : team) {
      if (R->error.empty()) {
        alive.push_back(R);
        continue;
      }
      if (first_error.empty()) first_error = "device " + std::to_string(R->device) + ": " + R->error;
      again.insert(again.end(), R->taken.begin(), R->taken.end());
      const size_t cap = R->prover->capacity();
      for (size_t at = R->rest_lo; at < R->rest_hi; at += cap) again.emplace_back(at, std::min(cap, R->rest_hi - at));
      if (p->failover_rounds > 0) {
        R->quarantined = true;
        R->sat_out = 0;
      }
    }
    if (again.empty()) break;   // every chunk of the job has been proved by a replica that finished its dispatch
    if (round >= p->failover_rounds) throw Error("rlnamd_pool_prove: " + first_error);
    if (alive.empty()) throw Error("rlnamd_pool_prove: no replica left to take over (" + first_error + ")");
    p->redo = std::move(again);
    p->redo_cursor.store(0);
    team = std::move(alive);
  }
  RLN_CATCH
}

int rlnamd_pool_set_dynamic(rlnamd_pool* p, int on) {
  std::lock_guard<std::mutex> job_lk(p->job_mu);
  p->dynamic = on != 0;
  return RLNAMD_OK;
}
int rlnamd_pool_inject_fault(rlnamd_pool* p, size_t replica, size_t after_chunks) {
  RLN_TRY
  if (replica >= p->rep.size()) throw Error("rlnamd_pool_inject_fault: no such replica");
  std::lock_guard<std::mutex> job_lk(p->job_mu);
  p->fault_after = p->rep[replica]->chunks_taken + after_chunks;
  p->fault_replica.store((long)replica);
  RLN_CATCH
}
int rlnamd_pool_set_failover(rlnamd_pool* p, int rounds) {
  std::lock_guard<std::mutex> job_lk(p->job_mu);
  p->failover_rounds = rounds > 0 ? rounds : 0;
  return RLNAMD_OK;
}
int rlnamd_pool_set_probation(rlnamd_pool* p, size_t jobs) {
  std::lock_guard<std::mutex> job_lk(p->job_mu);
  p->probation_jobs = jobs;
  return RLNAMD_OK;
}
int rlnamd_pool_health(rlnamd_pool* p, int* quarantined_per_replica, size_t* failures_per_replica) {
  std::lock_guard<std::mutex> job_lk(p->job_mu);
  for (size_t i = 0; i < p->rep.size(); i++) {
    if (quarantined_per_replica) quarantined_per_replica[i] = p->rep[i]->quarantined ? 1 : 0;
    if (failures_per_replica) failures_per_replica[i] = p->rep[i]->failures;
  }
  return RLNAMD_OK;
}
int rlnamd_pool_revive(rlnamd_pool* p, size_t replica) {
  RLN_TRY
  if (replica >= p->rep.size()) throw Error("rlnamd_pool_revive: no such replica");
  std::lock_guard<std::mutex> job_lk(p->job_mu);
  p->rep[replica]->quarantined = false;
  RLN_CATCH
}
int rlnamd_pool_last_proofs(rlnamd_pool* p, size_t* proofs_per_replica) {
  for (size_t i = 0; i < p->rep.size(); i++) proofs_per_replica[i] = p->rep[i]->last_proofs;
  return RLNAMD_OK;
}
int rlnamd_pool_last_ms(rlnamd_pool* p, float* ms_per_replica) {
  for (size_t i = 0; i < p->rep.size(); i++) ms_per_replica[i] = p->rep[i]->last_ms;
  return RLNAMD_OK;
}

/* verification stays on the host; any replica's zkey will do */
int rlnamd_pool_verify_many(rlnamd_pool* p, size_t n, const uint8_t* proofs, const uint8_t* values_le, size_t n_values,
                            int threads, uint8_t* ok) {
  RLN_TRY
  verify_many_common(p->rep[0]->prover->zkey(), n, proofs, values_le, n_values, threads, ok);
  RLN_CATCH
}

// ------------------------------------------------------------------------------------------------ RCCL communicator
int rlnamd_comm_unique_id(uint8_t id[RLNAMD_COMM_ID_BYTES]) {
  RLN_TRY
  require_gpu();
  static_assert(sizeof(ncclUniqueId) == RLNAMD_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId u;
  RLN_NCCL(ncclGetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  RLN_CATCH
}

int rlnamd_comm_init_rank(const uint8_t id[RLNAMD_COMM_ID_BYTES], int nranks, int rank, rlnamd_comm** out) {
  RLN_TRY
  require_gpu();
  if (nranks < 1 || rank < 0 || rank >= nranks) throw Error("rlnamd_comm_init_rank: bad rank / size");
  std::unique_ptr<rlnamd_comm> c(new rlnamd_comm);
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  RLN_HIP(hipGetDevice(&c->device));
  RLN_NCCL(ncclCommInitRank(&c->comm, nranks, u, rank));
  c->nranks = nranks;
  c->rank = rank;
  *out = c.release();
  RLN_CATCH
}

int rlnamd_comm_init_all(const int* devices, size_t n_devices, rlnamd_comm** out_array) {
  RLN_TRY
  require_gpu();
  if (!devices || n_devices == 0) throw Error("rlnamd_comm_init_all: no devices");
  std::vector<ncclComm_t> comms(n_devices);
  RLN_NCCL(ncclCommInitAll(comms.data(), (int)n_devices, devices));
  for (size_t i = 0; i < n_devices; i++) {
    rlnamd_comm* c = new rlnamd_comm;
    c->comm = comms[i];
    c->nranks = (int)n_devices;
    c->rank = (int)i;
    c->device = devices[i];
    out_array[i] = c;
  }
  RLN_CATCH
}

void rlnamd_comm_free(rlnamd_comm* c) {
  if (!c) return;
  if (c->comm) (void)ncclCommDestroy(c->comm);
  delete c;
}
int rlnamd_comm_rank(rlnamd_comm* c) { return c->rank; }
int rlnamd_comm_ranks(rlnamd_comm* c) { return c->nranks; }

// One MSM over n_total generated points sharded over the devices of THIS process: a thread per device, each with its
// own MsmG1 on its slice and its rank of one communicator.  ms[0..3]: max over the devices of sort / buckets /
// all-gather / combine; ms[4] = wall time of the slowest device including the host wait.
int rlnamd_msm_generated_multi(const int* devices, size_t n_devices, uint64_t seed, size_t n_total, uint32_t mode,
                               int repeats, uint8_t out_xy_le[64], float ms[5]) {
  RLN_TRY
  require_gpu();
  if (!devices || n_devices == 0) throw Error("rlnamd_msm_generated_multi: no devices");
  const size_t N = n_devices;
  std::vector<rlnamd_comm*> comms(N, nullptr);
  if (rlnamd_comm_init_all(devices, N, comms.data()) != RLNAMD_OK) throw Error(rlnamd_last_error());
  std::vector<std::string> errs(N);
  std::vector<std::array<float, 5>> t(N);
  std::vector<std::array<uint8_t, 64>> res(N);
  std::vector<std::thread> th;
  // Two phases.  Setup (device, workspace, generated points) can fail on one rank only -- out of memory on one GPU --
  // and a rank that never reaches its ncclAllGather leaves the others blocked in theirs for ever.  So every thread
  // reports its setup, all meet at a host barrier, and the collective loop is entered only when every rank is ready.
  std::mutex bmu;
  std::condition_variable bcv;
  size_t arrived = 0;
  bool all_ok = true;
  auto abort_all = [&] {
    std::lock_guard<std::mutex> lk(bmu);
    for (auto* c : comms)
      if (c->comm) {
        (void)ncclCommAbort(c->comm);
        c->comm = nullptr;
      }
  };
  for (size_t i = 0; i < N; i++)
    th.emplace_back([&, i] {
      std::unique_ptr<MsmG1> m;
      std::string err;
      try {
        RLN_HIP(hipSetDevice(devices[i]));
        size_t lo = n_total * i / N, hi = n_total * (i + 1) / N;
        m.reset(new MsmG1(hi - lo ? hi - lo : 1));
        m->generate(seed, lo, hi - lo, mode);
      } catch (const std::exception& e) {
        err = e.what();
      } catch (...) {
        err = "unknown exception during setup";
      }
      {
        std::unique_lock<std::mutex> lk(bmu);
        if (!err.empty()) all_ok = false;
        if (++arrived == N) bcv.notify_all();
        else bcv.wait(lk, [&] { return arrived == N; });
        if (!all_ok) {
          errs[i] = err;
          return;
        }
      }
      try {
        for (int r = 0; r < std::max(1, repeats); r++) {
          auto t0 = std::chrono::steady_clock::now();
          m->run_sharded(comms[i]->comm, (int)N, res[i].data(), t[i].data());
          t[i][4] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
      } catch (const std::exception& e) {
        errs[i] = e.what();
        abort_all();   // release the ranks blocked in the collective
      } catch (...) {
        errs[i] = "unknown exception in the collective phase";
        abort_all();
      }
    });
  for (auto& x : th) x.join();
  for (auto* c : comms) rlnamd_comm_free(c);
  for (size_t i = 0; i < N; i++)
    if (!errs[i].empty()) throw Error("device " + std::to_string(devices[i]) + ": " + errs[i]);
  if (!all_ok) throw Error("rlnamd_msm_generated_multi: setup failed");
  for (size_t i = 1; i < N; i++)
    if (memcmp(res[i].data(), res[0].data(), 64) != 0) throw Error("ranks disagree on the MSM result");
  memcpy(out_xy_le, res[0].data(), 64);
  if (ms)
    for (int k = 0; k < 5; k++) {
      ms[k] = 0;
      for (size_t i = 0; i < N; i++) ms[k] = std::max(ms[k], t[i][k]);
    }
  RLN_CATCH
}

}  // extern "C"
