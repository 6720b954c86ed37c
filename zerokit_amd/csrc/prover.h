// Batched RLN Groth16 prover resident on one MI355X.
//
// Device-side replacement for the path below generate_zk_proof_with_rs
// (/root/reference/rln/src/protocol/proof.rs:753-777):
//   calc_witness            (circuit/iden3calc.rs:20-60, iden3calc/graph.rs:246-272)  -> k_witness
//   CircomReduction         (circuit/qap.rs:30-98)                                     -> k_matvec, k_ntt_pass, k_hquot
//   create_proof_with_...   (ark-groth16 0.5.0; restated in partial_proof.rs:182-274)  -> k_recode, k_msm_*, k_finalize
//   proof_values_from_witness (protocol/witness.rs:759-828)                            -> k_proof_values
//
// Data layout (everything "batch-lane"): every per-proof quantity is stored [index][proof] with 32-byte
// elements, so the 64 lanes of a wavefront are 64 different proofs executing the same straight-line work
// on the same index: all loads/stores are 2 KiB-contiguous per wave, table rows / twiddles / matrix
// coefficients are wave-uniform, and no kernel needs LDS transposes or cross-lane traffic until the final
// per-proof reduction.
//
// MSM: the bases are fixed by the zkey, so Pippenger's bucket phase is replaced by a fixed-base comb table
// held in HBM: for every base P_k and every c-bit window j the table stores d * 2^(c j) * P_k for
// d = 1..2^(c-1) in affine form (signed digits).  A proof's MSM is then  sum_k sum_j +-T[k][j][|d_kj|]:
// mixed additions only, no bucket reduction, no data-dependent writes.  288 GB of HBM is what makes the
// table affordable (c = 8: 10 GB; c = 12: 90 GB).
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "curve.h"
#include "zkey.h"

namespace rlnamd {

constexpr int PROVER_STAGES = 8;
// Walks over the fixed-base tables (partial_proof.rs:108-274): the whole proof, the part fixed by a partial
// witness (secret, limit, Merkle path), or the remainder given the partial points.
enum ProveMode { PROVE_FULL = 0, PROVE_PARTIAL = 1, PROVE_FINISH = 2 };
extern const char* const kProverStageNames[PROVER_STAGES];

struct ProverConfig {
  int window_bits = 0;      // 0 = take RLNAMD_WINDOW_BITS or the default (120010: G1 c = 10, G2 c = 12); g1 + 10000 * g2, spec = c + 100 * wide
  size_t max_batch = 1024;  // workspace capacity in proofs (rounded up to a multiple of 64)
  long partial_cache = -1;  // entries of the partial-proof cache (collect_partial_cached); -1 = RLNAMD_PARTIAL_CACHE or 64
};

// Every switch of the prover, read from the environment ONCE when a Prover is built (ProverTuning::from_env; printed by
// rlnamd_prover_describe).  Sizes choose an operating point; the shape switches force one of two production shapes --
// both are what batches of some size / pipeline state take anyway -- so that the parity tests can pin each of them
// (tests/test_gpu_parity.py: test_small_batch_shape_variants_give_the_golden_bytes).  Tuning constants whose alternative
// lost every A/B (NTT through Fr29, the 8 x 32 walk, copy-engine staging, one interpreter stream, ...) are gone.
struct ProverTuning {
  // ---- sizes
  int window_bits = 120010;            // RLNAMD_WINDOW_BITS: comb schedule g1 + 10000 * g2, each c + 100 * wide;
                                       // default G1 c = 10 (13 windows), G2 c = 12 (11 windows): 20 GiB, one proof 0.1 ms sooner than c = 8
  int slots = 5;                       // RLNAMD_SLOTS: workspace slots = batches in flight (2 .. 6)
  uint32_t lanechunk_max = 128;        // RLNAMD_LANECHUNK: largest batch that takes the small-batch (latency) shapes
  uint32_t lanechunk_walk_max = 48;    // RLNAMD_LANECHUNK_WALK: largest lone batch walked with lanes = chunks
  uint32_t witlanes_max = 1024;        // RLNAMD_WITLANES_MAX: largest lone batch interpreted with lanes = nodes (a wave and a CU's
                                       // LDS per proof: ceil(n / 256) x 1.5 ms against 11 ms for lanes = proofs; lone 512 / 1 024-proof
                                       // batches 46.9 -> 38.0 / 74.0 -> 68.5 ms, the first batch of a stream 5 ms earlier)
  uint32_t tiny_max = 5;               // RLNAMD_TINY: largest lone batch walked with ONE (row, half) per lane (0: never); 5 proofs 3.15 -> 3.0 ms, 6 even, 8 slower
  uint32_t partial_cache = 64;         // RLNAMD_PARTIAL_CACHE: entries of the partial-proof cache (the known stored values of a
                                       // partial run, ~0.26 MB each on the depth-20 circuit); 0: finish always re-walks the whole graph
  uint32_t ntt_lg_max = 96;            // RLNAMD_NTT_LG_MAX: largest small batch whose NTTs run as the three LDS kernels (above, the
                                       // single-wave passes finish earlier beside the walks: 128 proofs 13.2 -> 12.6 ms)
  // ---- shapes (1 = default)
  bool glv = true;                     // RLNAMD_GLV: walk the 127-bit GLV halves (0: the plain 255-bit walk)
  bool wit29 = true;                   // RLNAMD_WIT29: interpreter in the 9 x 29 form (0: the 8 x 32 fallback k_witness)
  int lone = -1;                       // RLNAMD_LONE: -1 detect whether a batch is alone on the device, 0 / 1 force
  uint32_t lone_small_max = 48;        // RLNAMD_LONE_SMALL: batches of at most this many proofs take the lone (latency) shapes even behind a batch in flight
  bool early_walk = true;              // RLNAMD_EARLY_WALK: small batches walk the h-independent rows beside the NTTs
  bool early_fin = true;               // RLNAMD_EARLY_FIN: small batches finish A, B1 before the h rows are walked
  bool fused_smul = true;              // RLNAMD_FUSED_SMUL: a lone small proof takes s A, r B1 as rows of the C segment
  bool values_from_witness = true;     // RLNAMD_VALUES_WITNESS: small batches read the proof values off the witness
  uint32_t hint_max = 24;              // RLNAMD_HINTS: largest lone batch interpreted as independent segments behind host-computed hints (cold chains on 8 host threads: 12 / 16 / 24 proofs 3.2 / 3.7 / 4.3 -> 2.5 / 3.0 / 3.9 ms, even at 32)
                                       // (the values between the circuit's chained hashes; 0: never).  A proof's hints are ~0.3 ms of
                                       // hashing on a host core (the proofs of a batch on a thread each) against ~1.3 ms of interpreter
  uint32_t hint_chains = 64;           // RLNAMD_HINT_CHAINS: members whose public chain of hints (rate commitment, the hash after every level) is remembered on the host; 0: none
  uint32_t hint_max_warm = 64;         // RLNAMD_HINTS_WARM: ... and up to this many when at most 2.5 chains per host thread have to be hashed (the others are remembered: hint_chains)
  uint32_t hint_threads = 8;           // RLNAMD_HINT_THREADS: host threads (the caller's included) that hash the hint chains of a batch's proofs; at most half of the host's hardware threads unless set
  int hint_fault = 0;                  // RLNAMD_HINT_FAULT (test hook): j > 0 corrupts hint j - 1 of the first proof of every hinted batch
  bool d2h_kernel = true;              // RLNAMD_D2H_KERNEL: big batches copy their results home by a single-wave kernel (0: hipMemcpyAsync)
  // ---- diagnostics
  bool marks_small = false;            // RLNAMD_MARKS_SMALL: record stage timing marks for small batches too
  static ProverTuning from_env();
  std::string describe() const;
};
// (the lanes = nodes interpreter reads RLNAMD_WITLANES / RLNAMD_WITROWS / RLNAMD_WL_REASSOC / RLNAMD_WL_HOIST when its program is built:
// witness_lanes.hip, witness_sched.cpp)

struct ProofOut {            // one proof, host side
  uint8_t compressed[128];   // ark-serialize compressed Proof{a,b,c}
  uint8_t coords[256];       // affine A.x A.y | B.x.c0 B.x.c1 B.y.c0 B.y.c1 | C.x C.y  (canonical LE)
  uint8_t values[5][32];     // y, root, nullifier, x, external_nullifier (verifier order, proof.rs:863-869)
  uint32_t error;            // 0 ok; otherwise witness-graph evaluation failed for this proof
};

class Prover {
 public:
  Prover(const uint8_t* zkey, size_t zkey_len, const uint8_t* graph, size_t graph_len, ProverConfig cfg);
  ~Prover();

  const ProverTuning& tuning() const;
  // who else proves on this device: bit 0 another prover of this process, bit 1 a prover of ANOTHER process (a record lock
  // on /dev/shm/rlnamd_<PCI bus id>.lock, probed at most every 50 ms).  Either one keeps the wide latency shapes (a wave
  // and a CU's LDS per proof for up to 1 024 proofs) off: they assume the chip is this prover's.
  int device_shared() const;
  const Zkey& zkey() const { return zk_; }
  const Graph& graph() const { return graph_; }
  size_t capacity() const { return B_; }
  // constructor wall time: [0] parsing + verifier precomputation, [1] hipMalloc of the comb tables, [2] building them,
  // [3] the rest (plans, constants, workspaces, pinned staging)
  const float* init_ms() const { return init_ms_; }
  int window_bits() const { return c_; }      // G1 comb: narrow window width
  int windows() const { return W_; }          // table additions per G1 point and proof (windows x GLV halves)
  int window_bits_g2() const { return c2_; }
  int windows_g2() const { return W2_; }
  bool glv() const { return glv_; }
  size_t g1_rows() const;                     // table rows (finite points) of the G1 / G2 walk
  size_t g2_rows() const;
  size_t inputs_per_proof() const { return graph_.inputs_size; }
  size_t table_bytes() const;

  // Host inputs: `inputs` = n x inputs_size canonical 32-byte LE values (the witness-graph inputs buffer,
  // slot 0 = 1, iden3calc.rs:122-181); rs = n x 2 x 32 bytes (r, s).  Copies to the device.
  void upload(size_t n, const uint8_t* inputs, const uint8_t* rs);
  // Runs the whole pipeline on the resident inputs and waits for it.
  // Externally calculated witnesses (n x num_signals canonical LE) for the NEXT run(n, PROVE_FULL): they replace
  // the graph interpreter's output (generate_zk_proof_with_witness, protocol/proof.rs:705-732).
  void upload_witness(size_t n, const uint8_t* w_le);
  size_t num_signals() const { return graph_.signals.size(); }
  void run(size_t n, int mode = PROVE_FULL);
  // Same, but only enqueues: consecutive calls overlap (batch k+1's witness/NTT front end and batch k-1's
  // finalize back end run beside batch k's MSM on separate HIP streams; two workspace slots).  Every
  // batch ends with its proofs + values copied to pinned host memory.  sync() drains the pipeline.
  void run_async(size_t n, int mode = PROVE_FULL);
  void sync();
  void sync_measure(bool last_only);
  // Streamed batches: submit() stages n fresh inputs (+ rs, + the partial points in finish mode) in the slot's pinned
  // buffer, copies them to the slot's own device buffers on the batch's front-end stream and enqueues the batch; it
  // returns a ticket (> 0) at once unless all slots() workspace slots are in flight (then it waits for the oldest).
  // collect() waits for THAT batch only and copies its results out of pinned host memory (any pointer may be null;
  // partial320 is the result of a PROVE_PARTIAL batch).  A ticket expires when its slot is reused, i.e. slots()
  // submits later.  This is the path of SURVEY 8(d)'s timed region: H2D of witness inputs -> D2H of proofs.
  uint64_t submit(size_t n, const uint8_t* inputs, const uint8_t* rs, int mode = PROVE_FULL,
                  const uint8_t* partial320 = nullptr);
  // The hints of a lone small batch (the values between the circuit's chained hashes: prover.hip, Impl::rln_hints) may be
  // computed ahead of the call, by any thread, one proof at a time: hint_words() 32-bit words per proof (0: this circuit
  // has no such form), hints_for() fills them from one proof's packed inputs.  submit_hinted() is submit() for a full
  // proof batch of at most 64 proofs whose hints are at hand (n x hint_words() words): nothing is hashed inside the call,
  // and the batch takes the segments whatever its members' chains would have cost.  The device checks every hint as
  // always; hints that do not belong to the inputs cost a run over the whole graph, never a wrong proof.
  uint32_t hint_words() const;
  void hints_for(const uint8_t* inputs, uint32_t* hints) const;
  uint64_t submit_hinted(size_t n, const uint8_t* inputs, const uint8_t* rs, const uint32_t* hints);
  // wipe_after (default): the batch's inputs -- pinned staging and device copies -- its (r, s) and its witness values are
  // overwritten behind the copy-out (the reference zeroises the identity secret and the witness calculator's inputs,
  // rln/src/utils.rs:440-527, circuit/iden3calc.rs:45-56).  Pass false to read more of the batch (collect_public), then
  // call wipe(ticket).
  void collect(uint64_t ticket, size_t n, uint8_t* proofs, uint8_t* values, uint32_t* errors, uint8_t* coords = nullptr,
               uint8_t* partial320 = nullptr, bool wipe_after = true);
  // ticket 0: the resident-input path (upload / run / download keeps its data for the fetch_* taps until this is called)
  void wipe(uint64_t ticket = 0);
  void collect_public(uint64_t ticket, size_t n, std::vector<uint8_t>* out_le);
  int slots() const;
  // n proofs (any n) through submit / collect in chunks of <= capacity(), results in index order
  // the same stream with the chunks handed out by `next` (false: no more; chunks of at most capacity() proofs at any
  // offsets into the caller's arrays): a replica of a pool takes its contiguous shard this way, or whatever a cursor
  // shared with the other replicas gives it (rlnamd_pool, dynamic assignment).  max_in_flight 0 = every workspace slot.
  typedef std::function<bool(size_t* off, size_t* cnt)> ChunkSource;
  void prove_stream_from(const ChunkSource& next, const uint8_t* inputs, const uint8_t* rs, uint8_t* proofs, uint8_t* values,
                         uint32_t* errors, int max_in_flight = 0);
  void prove_stream(size_t n, const uint8_t* inputs, const uint8_t* rs, uint8_t* proofs, uint8_t* values,
                    uint32_t* errors);
  // Partial proofs.  PROVE_PARTIAL: inputs carry only the partial witness (unknown slots zero); the result is
  // four points per proof, canonical affine [pi_a x,y | rho x,y | pi_b x.c0,x.c1,y.c0,y.c1 | pi_c x,y] = 320 B
  // (create_partial_proof_from_assignment, partial_proof.rs:108-179).  PROVE_FINISH: full inputs + (r, s) +
  // the partial points uploaded with upload_partial (finish_partial_proof_with_assignment, :182-274).
  // Finish without re-walking the known cone (round 6).  finish_zk_proof_with_rs calculates the whole witness again
  // (protocol/proof.rs:822-849), although everything the partial witness fixes -- the identity commitment, the 20-level
  // Merkle chain: 21 488 of the 23 414 nodes and 11 / 12 of the multiplication depth -- came out of the partial run.
  // collect_partial_cached = collect of a PROVE_PARTIAL batch that also keeps, per proof, the stored values of the KNOWN
  // nodes in a device-resident cache entry and hands back an opaque handle (0: the cache is full or off -- such a proof
  // finishes through the full interpreter).  submit_finish = submit(PROVE_FINISH) with those handles: when every proof of
  // a small batch has a live handle the front end restores the known rows from the cache and interprets only the cone
  // evaluate_partial leaves unknown (witness_sched.h: wl_cone); bytes identical either way.  The PartialProof wire form
  // (partial_proof.rs:31-43) is untouched: the handle travels beside it.  A cache entry holds witness values (the identity
  // secret among them): release_partial overwrites it, ~Prover overwrites what is left.
  void collect_partial_cached(uint64_t ticket, size_t n, uint8_t* partial320, uint64_t* handles, uint32_t* errors);
  uint64_t submit_finish(size_t n, const uint8_t* inputs, const uint8_t* rs, const uint8_t* partial320, const uint64_t* handles);
  void release_partial(const uint64_t* handles, size_t n);
  // [0] capacity in entries, [1] entries in use, [2] bytes per entry, [3] 16-byte words that are not zero in the entries
  // NOT in use (all of them zero: a released entry was wiped), [4] batches that took the cone so far, [5] nodes of the cone
  // program, [6] steps of the cone program, [7] steps of the full program
  // the graph as segments behind hints (witness_sched.h: wl_segments): [0] segments, [1] hints per proof, [2] steps of the
  // longest segment, [3] steps of the whole graph's program, [4] batches interpreted that way, [5] of those, batches whose
  // hints did not check and were run again over the whole graph (0 unless the RLNAMD_HINT_FAULT test hook is set), [6] proofs
  // whose chain of hints was found among the remembered ones (the same member at the same root: two host hashes instead of 22)
  static constexpr int HINT_STATS_FIELDS = 7;
  void hint_stats(uint64_t out[HINT_STATS_FIELDS]) const;
  static constexpr int PARTIAL_CACHE_FIELDS = 8;
  void partial_cache_info(uint64_t out[PARTIAL_CACHE_FIELDS]);
  void upload_partial(size_t n, const uint8_t* coords320);
  void download_partial(size_t n, uint8_t* coords320);
  // per witness signal (length = number of signals): 1 when fixed by the partial witness (PartialProof::mask
  // is this vector without its first entry)
  const std::vector<uint8_t>& known_mask() const;
  void download(size_t n, ProofOut* out);
  // convenience
  void prove(size_t n, const uint8_t* inputs, const uint8_t* rs, ProofOut* out) {
    upload(n, inputs, rs);
    run(n);
    download(n, out);
  }
  // stage spans in ms (HIP events on the stage's own stream), averaged over the batches still held in the workspace
  // slots (the last <= 5 launches of the same kind); after run() of a single batch: that batch alone
  void stage_ms(float out[PROVER_STAGES]) const;
  // mean shader clock (MHz) under the G1 / G2 table walks since the previous call (sampled workgroups: shader-clock
  // cycles over 100 MHz wall ticks); drains the pipeline
  void walk_clock_mhz(double out[2]);
  // debug / parity taps (host copies, canonical LE): witness signals and h for proof p of the last run
  // public signals w[1..num_instance) of the first n proofs of the last run, n x (num_instance-1) x 32 bytes
  void fetch_public(size_t n, std::vector<uint8_t>* out_le);
  size_t num_public() const { return (size_t)zk_.num_instance_variables - 1; }
  void fetch_witness(size_t p, std::vector<uint8_t>* w_le);
  void fetch_h(size_t p, std::vector<uint8_t>* h_le);
  // tap of the wipes: 16-byte words of the last batch's slot that are not zero, whole buffers --
  // [G1 digits, G2 digits, a|b|c (h), G1 partial sums, G2 partial sums, staged inputs + (r, s)]
  static constexpr int RESIDUE_FIELDS = 6;
  void residue(uint64_t out[RESIDUE_FIELDS]);

 private:
  uint64_t settle_hints(uint64_t ticket);
  uint64_t enqueue(size_t n, int mode, const uint8_t* h_inputs, const uint8_t* h_rs, const uint8_t* h_pp320,
                   const uint64_t* cone_handles = nullptr, const uint32_t* pre_hints = nullptr);
  void fetch_public_slot(void* slot, size_t n, std::vector<uint8_t>* out_le);
  struct Impl;
  std::unique_ptr<Impl> d_;
  Zkey zk_;
  Graph graph_;
  size_t B_ = 0;
  float init_ms_[4] = {0, 0, 0, 0};
  int c_ = 8, W_ = 32, c2_ = 8, W2_ = 32;
  bool glv_ = true;
};

}  // namespace rlnamd
