/* rln_amd.h -- batch / device-resident extension entry points of the MI355X RLN backend.
 *
 * The zerokit-compatible drop-in surface is include/rln.h (same symbols and struct layouts as the header
 * safer-ffi generates from /root/reference/rln/src/ffi/).  The reference has NO batch, no deterministic
 * (r, s) and no device-resident API (SURVEY.md §8b "Threading" / "Determinism gap"); the functions below
 * add exactly those, with plain pointers and sizes only.  All field elements cross this boundary as
 * 32-byte little-endian canonical integers (the form `fr_to_bytes_le` produces,
 * /root/reference/rln/src/utils.rs:75-120).
 *
 * Every function returns 0 on success and a non-zero code on failure; the message is available from
 * rlnamd_last_error() (thread-local).  There is no CPU fallback: without a HIP device every compute
 * entry fails with RLNAMD_ERR_NO_DEVICE.
 */
#ifndef RLN_AMD_H
#define RLN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLNAMD_OK 0
#define RLNAMD_ERR 1
#define RLNAMD_ERR_NO_DEVICE 2

#define RLNAMD_PROVER_STAGES 8

const char* rlnamd_last_error(void);
int rlnamd_device_count(void);
int rlnamd_set_device(int ordinal);
int rlnamd_get_device(int* ordinal);   /* the calling thread's current device */
int rlnamd_device_name(char* buf, size_t cap);

/* ---- hashing ----------------------------------------------------------------------------------------
 * Batched Poseidon: poseidon_hash / poseidon_hash_pair (rln/src/hashers.rs:32-54) over n independent
 * inputs.  inputs: n * arity * 32 bytes, out: n * 32 bytes (host memory). arity 1..8 (t = 2..9). */
int rlnamd_poseidon_hash(const uint8_t* inputs_le, size_t n, size_t arity, uint8_t* out_le);
/* hash_to_field_le / _be (rln/src/hashers.rs:73-93): Keccak-256 then reduction mod r.  Host only. */
int rlnamd_hash_to_field_le(const uint8_t* data, size_t len, uint8_t out_le[32]);
int rlnamd_hash_to_field_be(const uint8_t* data, size_t len, uint8_t out_le[32]);

/* ---- HBM-resident Poseidon Merkle tree ---------------------------------------------------------------
 * FullMerkleTree semantics (utils/src/merkle_tree/full_merkle_tree.rs).  Calls on one handle are serialised by the
 * handle's own mutex (one stream and shared staging buffers per tree: readers touch object state too), so a handle may
 * be shared between threads; different handles do not wait for each other. */
typedef struct rlnamd_tree rlnamd_tree;
int rlnamd_tree_new(size_t depth, rlnamd_tree** out);                         /* ::default(depth) :74-80 */
void rlnamd_tree_free(rlnamd_tree* t);
int rlnamd_tree_set_range(rlnamd_tree* t, size_t start, const uint8_t* leaves_le, size_t n); /* :197-223 */
/* k single-leaf writes (any order; a later entry for the same index wins) followed by ONE bottom-up pass over the union
 * of their paths -- what k set() calls (:141-147, depth hashes each) leave behind, in one pass */
int rlnamd_tree_set_leaves(rlnamd_tree* t, const uint64_t* indices, const uint8_t* leaves_le, size_t k);
int rlnamd_tree_root(rlnamd_tree* t, uint8_t out_le[32]);                                    /* :137-139 */
int rlnamd_tree_get_leaf(rlnamd_tree* t, size_t index, uint8_t out_le[32]);                  /* :149-154 */
/* one proof: elems = depth*32 bytes bottom-up, bits = depth bytes (1 = node is a right child) :288-304 */
int rlnamd_tree_proof(rlnamd_tree* t, size_t index, uint8_t* elems_le, uint8_t* bits);
/* `count` proofs for leaves [first, first+count) copied back to host buffers */
int rlnamd_tree_proofs(rlnamd_tree* t, size_t first, size_t count, uint8_t* elems_le, uint8_t* bits);
/* Device-resident workload of BASELINE config 3: leaves first_value + i generated in HBM, full rebuild,
 * all `count` proofs emitted into an internal HBM buffer and (if verify != 0) every proof recomputed to
 * the root on the device.  ms[0] = build, ms[1] = proof emission (HIP events); bad = failed proofs. */
int rlnamd_tree_fill_sequential(rlnamd_tree* t, size_t start, size_t n, uint64_t first_value);
int rlnamd_tree_bench(rlnamd_tree* t, size_t n_leaves, uint64_t first_value, int verify, float ms[2], size_t* bad);

/* ---- batched Groth16 prover --------------------------------------------------------------------------
 * Replaces generate_zk_proof_with_rs (rln/src/protocol/proof.rs:753-777) + proof_values_from_witness
 * (protocol/witness.rs:759-804) for n independent proofs at once. */
typedef struct rlnamd_prover rlnamd_prover;
/* zkey/graph: the bytes of rln_final.arkzkey and graph.bin (circuit/mod.rs:140-203).
 * max_batch: workspace capacity; window_bits: 0 = default / RLNAMD_WINDOW_BITS, else g1 + 10000 * g2 with each
 * spec = c + 100 * wide (c-bit windows, the first `wide` one bit wider; g2 = 0: same as g1). */
int rlnamd_prover_new(const uint8_t* zkey, size_t zkey_len, const uint8_t* graph, size_t graph_len,
                      size_t max_batch, int window_bits, rlnamd_prover** out);
void rlnamd_prover_free(rlnamd_prover* p);
typedef struct {
  uint64_t inputs_size;      /* slots of the witness-graph inputs buffer (slot 0 = 1) */
  uint64_t num_signals;      /* witness length */
  uint64_t domain_size;      /* NTT domain */
  uint64_t tree_depth, max_out;
  uint64_t capacity;         /* max batch */
  uint64_t table_bytes;      /* HBM held by the fixed-base tables */
  int32_t window_bits, windows;       /* G1 comb: narrow window width; table additions per G1 point and proof */
  int32_t window_bits_g2, windows_g2; /* the same for the G2 comb */
  int32_t glv;                        /* 1: scalars are split k1 + lambda k2 and the combs cover 127 bits */
  int32_t reserved;
  uint64_t g1_rows, g2_rows;          /* finite points of the G1 / G2 walk (table rows) */
} rlnamd_prover_info;
int rlnamd_prover_get_info(rlnamd_prover* p, rlnamd_prover_info* info);
/* the same for the prover behind an object of include/rln.h (FFI_RLN_t* / FFI_RLNV3_t*, passed as void*): shows how
 * the "window_bits" / "max_batch" keys of the config_path JSON (or RLNAMD_WINDOW_BITS / RLNAMD_MAX_BATCH) sized it */
int rlnamd_ffi_prover_info(const void* ffi_rln, rlnamd_prover_info* info);
/* The member memo of an object whose config_path JSON carries "auto_partial": N (0 = off, the default).  With it,
 * ffi_generate_rln_proof remembers the partial proof of up to N members (identity secret, limit, Merkle path): the first
 * proof of a member at a root is made from scratch and its partial proof follows on the device behind the call; later
 * proofs of that member at that root are finishes of it (rlnamd_prover_submit_finish: 0.8 instead of 2.1 ms), the same
 * proof for the same (r, s).  Opt-in because of what it keeps: the member's witness values on the device and the key
 * (identity secret included) in host memory until the entry is evicted, or the object freed.
 * out: [0] members remembered, [1] proofs that were finishes, [2] proofs from scratch, [3] 1 while a partial proof is pending */
int rlnamd_ffi_memo_stats(const void* ffi_rln, uint64_t out[4]);
/* Concurrent callers.  generate_rln_proof takes &self in the reference (rln/src/public.rs:624) and an RLN object may be
 * shared by threads; the prover proves one batch at a time, so the single-proof calls (ffi_generate_rln_proof,
 * ffi_rln_v3_generate_proof and their _with_rs twins) that arrive while a proof is on the device are gathered and go out
 * together, as one batch, when it returns -- each call gets its own proof or its own error text, a lone caller is a
 * batch of one as before.  "gather_calls": N in the config_path JSON (or RLNAMD_GATHER_CALLS) caps a batch (default:
 * the workspace's capacity; 0 or 1: off); on an object with "auto_partial" the gathered calls of remembered members go
 * out as one batch of finishes, the others one by one.  Threads that call in a
 * loop arrive just behind their results: the leader gives the callers it saw within the last 20 ms "gather_window_us"
 * (500; RLNAMD_GATHER_WINDOW_US; 0: none) to arrive before it takes the batch, so that T threads go out as batches of T
 * instead of two halves taking turns; a lone caller never waits, callers slower than the window make it stop waiting.
 * out: [0] batches led, [1] calls that went out in them, [2] the largest batch, [3] the cap (0: off), [4] batches whose
 * leader waited for a recent caller, [5] nanoseconds the leaders spent proving their batches; ffi_finish_rln_proof and
 * its twins (single message-id) are gathered the same way in a queue of their own: [6] its batches, [7] its calls */
int rlnamd_ffi_gather_stats(const void* ffi_rln, uint64_t out[8]);
/* offset/len of a named input signal in the inputs buffer (iden3calc.rs:122-146); returns RLNAMD_ERR if absent */
int rlnamd_prover_input_slot(rlnamd_prover* p, const char* name, uint32_t* offset, uint32_t* len);
/* inputs: n * inputs_size * 32 bytes (slot 0 must hold 1); rs: n * 64 bytes (r then s). */
int rlnamd_prover_upload(rlnamd_prover* p, size_t n, const uint8_t* inputs_le, const uint8_t* rs_le);
int rlnamd_prover_run(rlnamd_prover* p, size_t n);      /* inputs already resident; blocks until done */
/* Enqueue only: consecutive batches overlap on the device (front end of batch k+1 and back end of batch
 * k-1 beside the MSM of batch k); each batch ends with its proofs copied to pinned host memory.
 * rlnamd_prover_sync drains the pipeline; download refers to the last enqueued batch, stage_ms to the mean over the
 * batches still held in the workspace slots (the last <= 5 launches of the same kind). */
int rlnamd_prover_run_async(rlnamd_prover* p, size_t n);
int rlnamd_prover_sync(rlnamd_prover* p);
/* proofs: n*128 (ark-serialize compressed Proof), coords: n*256 (affine A|B|C) or NULL,
 * values: n*160 (y, root, nullifier, x, external_nullifier) or NULL, errors: n*4 or NULL */
int rlnamd_prover_download(rlnamd_prover* p, size_t n, uint8_t* proofs, uint8_t* coords, uint8_t* values,
                           uint32_t* errors);
/* ---- streamed batches: the path of SURVEY 8(d)'s timed region, "H2D of witness inputs -> D2H of proofs".  The reference
 * takes a fresh witness per call (rln/src/protocol/proof.rs:753-777, public.rs:624-631); rlnamd_prover_submit is n such
 * calls at once: it stages the inputs (n * inputs_size * 32 bytes), rs (n * 64) and, in RLNAMD_MODE_FINISH, the partial
 * points (n * 320, else NULL) in pinned memory owned by a workspace slot, copies them to that slot's own device buffers
 * on the batch's front-end stream and enqueues the batch.  It returns at once with a ticket (> 0) unless all
 * rlnamd_prover_slots() slots are in flight; consecutive submits of DIFFERENT batches overlap on the device exactly as
 * run_async's do.  rlnamd_prover_collect waits for that batch only (any output pointer may be NULL; partial320 is the
 * result of a RLNAMD_MODE_PARTIAL batch).  A ticket expires when its slot is reused, i.e. slots() submits later.
 * rlnamd_prover_prove_stream = submit / collect over any n in chunks of `capacity`, results in index order. */
int rlnamd_prover_slots(rlnamd_prover* p);
int rlnamd_prover_submit(rlnamd_prover* p, size_t n, const uint8_t* inputs_le, const uint8_t* rs_le, int mode,
                         const uint8_t* partial320, uint64_t* ticket);
int rlnamd_prover_collect(rlnamd_prover* p, uint64_t ticket, size_t n, uint8_t* proofs, uint8_t* coords, uint8_t* values,
                          uint32_t* errors, uint8_t* partial320);
/* public signals w[1..num_instance) of a finished batch (n * num_public * 32 bytes), circuit-generic.  Call it BEFORE
 * rlnamd_prover_collect: collect ends the batch -- see the next comment. */
int rlnamd_prover_collect_public(rlnamd_prover* p, uint64_t ticket, size_t n, uint8_t* out_le);
/* Secrets do not outlive the batch.  The reference zeroises the identity secret wherever it holds it (IdSecret,
 * rln/src/utils.rs:440-527) and the witness calculator's inputs (rln/src/circuit/iden3calc.rs:45-56).  Here
 * rlnamd_prover_collect (and prove_stream, the pool, every ffi_* proving call) overwrites the batch's staged inputs --
 * pinned host buffer and device copy -- its (r, s), its witness values and what was derived from them (the signed window
 * digits of both walks, a | b | c / h, the walks' partial sums: rlnamd_prover_residue) behind the copy-out; a later
 * rlnamd_prover_collect_public of that ticket is an error.  The resident-input calls (upload / run / download, kept for
 * the fetch_* parity taps) hold their data until rlnamd_prover_wipe or rlnamd_prover_free. */
int rlnamd_prover_wipe(rlnamd_prover* p);
/* the switches this prover was built with, "name=value ..." (ProverTuning, zerokit_amd/csrc/prover.h: every RLNAMD_*
 * variable the prover reads is read once, at construction, and listed there) */
int rlnamd_prover_describe(rlnamd_prover* p, char* buf, size_t cap);
/* where the constructor's wall time went, ms: [0] parsing the arkzkey / graph + the verifier's precomputation, [1] hipMalloc
 * of the comb tables, [2] building them on the device, [3] the rest (walk plans, constants, workspaces, pinned staging) */
int rlnamd_prover_init_ms(rlnamd_prover* p, float ms[4]);
int rlnamd_prover_prove_stream(rlnamd_prover* p, size_t n, const uint8_t* inputs_le, const uint8_t* rs_le,
                               uint8_t* proofs, uint8_t* values, uint32_t* errors);
int rlnamd_prover_stage_ms(rlnamd_prover* p, float ms[RLNAMD_PROVER_STAGES]);
const char* rlnamd_prover_stage_name(int i);
/* mean shader clock (MHz) under the G1 / G2 table walks since the previous call (the walks are VALU-issue bound: their
 * rate is SIMDs x clock / instructions); drains the pipeline */
int rlnamd_prover_walk_clock_mhz(rlnamd_prover* p, double mhz[2]);
/* parity taps of the last run: full witness (num_signals*32) / h (domain_size*32) of proof `index` */
int rlnamd_prover_fetch_witness(rlnamd_prover* p, size_t index, uint8_t* out_le);
int rlnamd_prover_fetch_h(rlnamd_prover* p, size_t index, uint8_t* out_le);
/* tap of the wipes: the number of 16-byte words that are not zero in the buffers of the slot the last batch used, whole
 * buffers: [0] G1 window digits, [1] G2 window digits, [2] a | b | c (the quotient's operands, then h), [3] / [4] partial
 * sums of the G1 / G2 walks and everything their reduction leaves behind (block sums, the unblinded A / B / C sums, the
 * affine points, the ladder's products and tables), [5] staged inputs + (r, s).  All zero behind a collect that wipes. */
int rlnamd_prover_residue(rlnamd_prover* p, uint64_t out[6]);
/* ---- partial proofs (generate_partial_zk_proof / finish_zk_proof_with_rs, protocol/proof.rs:783-849;
 * Groth16Partial, partial_proof.rs:108-274).  mode: 0 full proof, 1 partial (inputs hold only identitySecret,
 * userMessageLimit, pathElements, identityPathIndex; other slots zero), 2 finish (full inputs + r, s + the
 * partial points given with rlnamd_prover_upload_partial).  A partial proof is four points per proof as
 * canonical LE affine coordinates [pi_a x,y | rho x,y | pi_b x.c0,x.c1,y.c0,y.c1 | pi_c x,y] = 320 bytes;
 * rlnamd_prover_known_mask returns, per witness signal, whether the partial witness fixes it
 * (PartialProof::mask is this vector without entry 0). */
#define RLNAMD_MODE_FULL 0
#define RLNAMD_MODE_PARTIAL 1
#define RLNAMD_MODE_FINISH 2
int rlnamd_prover_run_mode(rlnamd_prover* p, size_t n, int mode);
int rlnamd_prover_run_async_mode(rlnamd_prover* p, size_t n, int mode);
/* A few proofs per call: the witness graph as independent segments behind hints.  The depth-20 circuit is 22 Poseidon
 * hashes in a row -- nine tenths of its interpreter steps are that one dependency chain, and a dependent 256-bit product
 * costs a lone GPU wave 0.31 us against 0.02 us on a host core.  For a lone batch of at most RLNAMD_HINTS (24) proofs the
 * calling thread (and a helper thread per further proof) computes the values BETWEEN the hashes (identity commitment, rate commitment, the running hash after
 * every Merkle level, a1: depth + 2 hashes with the library's host Poseidon, ~0.3 ms), the device interprets the 24
 * segments those values separate all at once (the longest 268 steps instead of 4 813) and then compares every cut node's
 * own value with the hint its consumers were given.  Every witness value is still computed on the device; a hint that does
 * not check (not expected: test hook RLNAMD_HINT_FAULT) makes collect run the batch again over the whole graph.
 * The chain part of a member's hints (rate commitment, the running hash after every level) depends on public values only:
 * the last 64 (RLNAMD_HINT_CHAINS) are remembered under a fingerprint of (identity commitment, limit, path), so a member that proves
 * again at the same root costs the host two hashes (identity commitment, a1) instead of depth + 2.
 * out: [0] segments, [1] hints per proof, [2] steps of the longest segment, [3] steps of the whole graph, [4] batches
 * interpreted as segments, [5] of those, batches run again, [6] proofs whose chain was remembered. */
int rlnamd_prover_hint_stats(rlnamd_prover* p, uint64_t out[7]);
/* The hints may be computed ahead of the call, by any thread, one proof at a time -- a server whose request threads each
 * hash their own proof's chain while the device is busy: rlnamd_prover_hint_words() 32-bit words per proof (0: this
 * circuit has no such form), rlnamd_prover_hints_for() fills them from one proof's packed inputs (host only, no device
 * call, safe beside a running batch), rlnamd_prover_submit_hinted() is rlnamd_prover_submit (full proofs) for a batch of
 * at most 64 proofs whose hints (n x hint_words) are at hand: nothing is hashed inside the call and the batch takes the
 * segments whatever its members' chains would have cost.  The device checks every hint as always: hints that do not
 * belong to the inputs cost a run over the whole graph, never a wrong proof. */
uint32_t rlnamd_prover_hint_words(rlnamd_prover* p);
int rlnamd_prover_hints_for(rlnamd_prover* p, const uint8_t* inputs_le, uint32_t* hints);
int rlnamd_prover_submit_hinted(rlnamd_prover* p, size_t n, const uint8_t* inputs_le, const uint8_t* rs_le,
                                const uint32_t* hints, uint64_t* ticket);
/* Who else proves on this prover's device: bit 0 = another prover of this process, bit 1 = a prover of ANOTHER process
 * (every process with a prover on a device holds a read record lock on /dev/shm/rlnamd_<PCI bus id>.lock; probed at most
 * every 50 ms).  A shared device keeps the wide latency shapes off (they assume the chip is this prover's). */
int rlnamd_prover_device_shared(rlnamd_prover* p, int* who);
/* Finish without re-walking what the partial witness fixed (round 6).  finish_zk_proof_with_rs recomputes the whole
 * witness (protocol/proof.rs:822-849) although the identity commitment and the 20-level Merkle chain -- 21 488 of the
 * depth-20 circuit's 23 414 graph nodes, eleven twelfths of its multiplication depth -- came out of the partial run.
 * rlnamd_prover_collect_partial_cached = rlnamd_prover_collect of a RLNAMD_MODE_PARTIAL batch that also keeps, per
 * proof, the stored values of the KNOWN nodes in a device-resident cache entry (0.26 MB on the depth-20 circuit;
 * RLNAMD_PARTIAL_CACHE entries, 64 unless set) and returns an opaque handle per proof: 0 when the cache is full or off
 * -- such a proof finishes through the full interpreter.  rlnamd_prover_submit_finish = a RLNAMD_MODE_FINISH submit
 * with those handles: when every proof of a small batch (one the wave-per-proof interpreter takes) has a live handle,
 * the known rows are restored from the cache and only the cone evaluate_partial (iden3calc/graph.rs:274-312) leaves
 * unknown is interpreted.  Bytes are identical either way; a stale or foreign handle is treated as 0.  The PartialProof
 * wire form (partial_proof.rs:31-43) is untouched -- the handle travels beside it and means nothing to another prover.
 * An entry holds witness values, the identity secret among them: rlnamd_prover_release_partial overwrites and frees it,
 * rlnamd_prover_free overwrites what is left.  rlnamd_prover_partial_cache_info: [0] capacity, [1] entries in use,
 * [2] bytes per entry, [3] non-zero 16-byte words in the entries NOT in use (0: released entries were wiped),
 * [4] batches that took the cone, [5] nodes of the cone, [6] steps of the cone program, [7] steps of the full program. */
int rlnamd_prover_collect_partial_cached(rlnamd_prover* p, uint64_t ticket, size_t n, uint8_t* partial320, uint64_t* handles,
                                         uint32_t* errors);
int rlnamd_prover_submit_finish(rlnamd_prover* p, size_t n, const uint8_t* inputs_le, const uint8_t* rs_le,
                                const uint8_t* partial320, const uint64_t* handles, uint64_t* ticket);
int rlnamd_prover_release_partial(rlnamd_prover* p, const uint64_t* handles, size_t n);
int rlnamd_prover_partial_cache_info(rlnamd_prover* p, uint64_t out[8]);
int rlnamd_prover_upload_partial(rlnamd_prover* p, size_t n, const uint8_t* coords320);
int rlnamd_prover_download_partial(rlnamd_prover* p, size_t n, uint8_t* coords320);
int rlnamd_prover_known_mask(rlnamd_prover* p, uint8_t* out_num_signals);
/* Externally calculated witnesses (n x num_signals x 32 canonical LE) for the NEXT full run of n proofs; they
 * replace the witness-graph interpreter's output (generate_zk_proof_with_witness, protocol/proof.rs:705-732). */
int rlnamd_prover_upload_witness(rlnamd_prover* p, size_t n, const uint8_t* witness_le);
/* Public signals (the circuit outputs/inputs w[1..num_instance)) of the first n proofs of the last run, read
 * from the witness: n * num_public * 32 bytes.  Circuit-generic (multi message-id: ys, root, nullifiers, x,
 * external_nullifier, selector_used -- the verifier order of protocol/proof.rs:870-885). */
size_t rlnamd_prover_num_public(rlnamd_prover* p);
int rlnamd_prover_download_public(rlnamd_prover* p, size_t n, uint8_t* out_le);
/* generic-arity verification: n_values public inputs */
int rlnamd_verify_public(rlnamd_prover* p, const uint8_t proof[128], const uint8_t* values_le, size_t n_values, int* ok);
/* verify_zk_proof (protocol/proof.rs:856-894) on the host CPU, as in the reference.
 * values: y, root, nullifier, x, external_nullifier.  *ok = 1 valid, 0 invalid. */
int rlnamd_verify(rlnamd_prover* p, const uint8_t proof[128], const uint8_t values_le[160], int* ok);
/* n independent verifications on `threads` host threads (0 = one per hardware thread): proofs n * 128 bytes, values
 * n * n_values * 32 bytes, ok[i] = 1 valid, 0 invalid or malformed.  EXT: the reference verifies one proof per call
 * (protocol/proof.rs:856-894); a relay node verifies every message it forwards. */
int rlnamd_verify_many(rlnamd_prover* p, size_t n, const uint8_t* proofs, const uint8_t* values_le, size_t n_values,
                       int threads, uint8_t* ok);
int rlnamd_verify_many_with_zkey(const uint8_t* zkey, size_t zkey_len, size_t n, const uint8_t* proofs,
                                 const uint8_t* values_le, size_t n_values, int threads, uint8_t* ok);
/* same check straight from arkzkey bytes; needs no GPU (host parser + host pairing only) */
int rlnamd_verify_with_zkey(const uint8_t* zkey, size_t zkey_len, const uint8_t proof[128],
                            const uint8_t values_le[160], int* ok);
/* host-only parse of the two circuit resources (zkey_from_raw / graph_from_raw, circuit/mod.rs:140-203);
 * counts[0..9] = instance vars, witness vars, constraints, a_nnz, b_nnz, |a_query|, |h_query|, |l_query|,
 * graph nodes, witness signals; counts[10] = tree depth, counts[11] = max_out, counts[12] = inputs size. */
int rlnamd_parse_resources(const uint8_t* zkey, size_t zkey_len, const uint8_t* graph, size_t graph_len,
                           uint64_t counts[13]);
/* ark-serialize point compression helpers (host): coords = A.x A.y B.x.c0 B.x.c1 B.y.c0 B.y.c1 C.x C.y */
int rlnamd_proof_compress(const uint8_t coords_le[256], uint8_t proof[128]);
int rlnamd_proof_decompress(const uint8_t proof[128], uint8_t coords_le[256]);

/* ---- variable-base MSM on G1 / G2 (BASELINE config 5; north_star: "windowed Pippenger MSM on G1/G2") ---------
 * Replaces VariableBaseMSM::msm_bigint (ark-ec 0.5.0; call sites rln/src/partial_proof.rs:98-104,255-256: generic
 * over the group) for large n with bases that are not fixed.  Multi-GPU: every rank runs rlnamd_msm_run on its slice of
 * the points, the window sums (rlnamd_msm_window_sums_bytes_of(m) bytes per rank) are all-gathered, and
 * rlnamd_msm_combine adds them and folds the windows.
 * rlnamd_msm_new = G1: points are 64 bytes (x | y), results 64 bytes.  rlnamd_msm_new_g2 = G2: points and results are
 * 128 bytes (x.c0 | x.c1 | y.c0 | y.c1); every call below takes either handle (the *_xy_le buffers are
 * rlnamd_msm_point_bytes(m) bytes per point); the synthetic workload of a G2 handle is P_i = k_i G2 (the generator of
 * the twist's order-r subgroup) under the same SplitMix64 stream. */
typedef struct rlnamd_msm rlnamd_msm;
int rlnamd_msm_new(size_t capacity, rlnamd_msm** out);
int rlnamd_msm_new_g2(size_t capacity, rlnamd_msm** out);
void rlnamd_msm_free(rlnamd_msm* m);
size_t rlnamd_msm_point_bytes(rlnamd_msm* m);
size_t rlnamd_msm_window_sums_bytes_of(rlnamd_msm* m);
/* Device self-test: the 9 x 29-bit-limb group law the MSM walks use (csrc/fq29.h) against the 8 x 32-bit one on
 * `threads` pseudo-random walks of `iters` signed additions each (doublings, cancellations, restarts from infinity
 * included).  group 1 = G1, 2 = G2, 3 = G2 computed by lane pairs, 4 = G1 general additions by lane pairs (g2_gen_xy_le = generator x.c0 | x.c1 | y.c0 | y.c1,
 * canonical LE; NULL for G1).
 * *mismatches = number of walks whose affine results differ (0 expected). */
/* Parameter self-check, host only (no device, not a hashing path): derives the Poseidon parameters for `arity`
 * inputs (Grain LFSR, utils/src/poseidon/poseidon_constants.rs:207-261) and evaluates ONE hash twice on the host --
 * with the reference's dense rounds (poseidon_hash.rs:97-135) and with the equivalent sparse partial rounds the device
 * kernels use -- so the CPU test suite can compare both with the oracle. */
int rlnamd_poseidon_params_check(const uint8_t* inputs_le, size_t arity, uint8_t out_dense_le[32], uint8_t out_sparse_le[32]);
int rlnamd_selftest_fq29(int group, uint32_t threads, uint32_t iters, const uint8_t* g2_gen_xy_le, uint32_t* mismatches);
/* points: n x (x || y) canonical LE affine, all-zero = infinity; scalars: n x 32 bytes canonical LE */
int rlnamd_msm_set(rlnamd_msm* m, const uint8_t* points_xy_le, const uint8_t* scalars_le, size_t n);
/* synthetic config-5 workload generated in HBM: P_i = k_i G, scalars s_i, SplitMix64(seed) at index first+i */
int rlnamd_msm_generate(rlnamd_msm* m, uint64_t seed, uint64_t first_index, size_t n);
/* distribution variants of the same workload: mode bit 0 = every scalar equals s_0 (all n points in ONE bucket per
 * window), bit 1 = k_i = k_(i mod 4) (four distinct bases).  The library carries no expected value for its own
 * workload: the closed form (sum k_i s_i mod r) G lives in the oracle (oracle/c: oracle_msm_expected). */
#define RLNAMD_MSM_EQUAL_SCALARS 1u
#define RLNAMD_MSM_FOUR_POINTS 2u
int rlnamd_msm_generate_mode(rlnamd_msm* m, uint64_t seed, uint64_t first_index, size_t n, uint32_t mode);
/* reads points [first, first + count) of the loaded / generated workload back (affine x || y, scalars; canonical LE) */
int rlnamd_msm_fetch(rlnamd_msm* m, size_t first, size_t count, uint8_t* points_xy_le, uint8_t* scalars_le);
size_t rlnamd_msm_window_sums_bytes(void);
/* ms[0] digits + counting sort, ms[1] bucket accumulation, ms[2] bucket reduction */
int rlnamd_msm_run(rlnamd_msm* m, uint8_t* window_sums, float ms[3]);
int rlnamd_msm_combine(rlnamd_msm* m, const uint8_t* window_sums, size_t contributors, uint8_t out_xy_le[64]);


/* ---- multi-GPU (BASELINE configs 4 and 5) -----------------------------------------------------
 * rlnamd_pool: one prover replica and one host thread per device of THIS process.  rlnamd_pool_prove cuts n proofs
 * into contiguous index shards, one per replica (config 4: 65 536 proofs = 8 x 8 192), each replica streams its shard
 * (rlnamd_prover_prove_stream) and writes its results at their index: proofs n * 128, values n * 160, errors n * 4
 * (any may be NULL).  No data-path collective -- the proofs are independent (the reference's guidance is one worker per
 * proof, rln/README.md:324-332).  devices = NULL / n_devices = 0: every visible device.  An ordinal may repeat (two
 * replicas sharing one device, for tests on a 1-GPU box). */
typedef struct rlnamd_pool rlnamd_pool;
int rlnamd_pool_new(const uint8_t* zkey, size_t zkey_len, const uint8_t* graph, size_t graph_len, size_t max_batch,
                    int window_bits, const int* devices, size_t n_devices, rlnamd_pool** out);
void rlnamd_pool_free(rlnamd_pool* p);
size_t rlnamd_pool_size(rlnamd_pool* p);
int rlnamd_pool_device(rlnamd_pool* p, size_t replica);
int rlnamd_pool_get_info(rlnamd_pool* p, rlnamd_prover_info* info);
int rlnamd_pool_prove(rlnamd_pool* p, size_t n, const uint8_t* inputs_le, const uint8_t* rs_le, uint8_t* proofs,
                      uint8_t* values, uint32_t* errors);
int rlnamd_pool_last_ms(rlnamd_pool* p, float* ms_per_replica);   /* host wall time of each replica's last shard */
/* Assignment of a job's proofs to the replicas: 0 (default) = contiguous index shards, one per replica; 1 = dynamic, a
 * cursor shared by the replicas over chunks of max_batch proofs (a slower device takes fewer chunks; at most three chunks
 * in flight per replica).  The result is index-identical either way.  rlnamd_pool_last_proofs: how many proofs of the
 * last job each replica made. */
int rlnamd_pool_set_dynamic(rlnamd_pool* p, int on);
int rlnamd_pool_last_proofs(rlnamd_pool* p, size_t* proofs_per_replica);
/* test hook: replica `replica` fails (throws inside its worker) when it is handed its after_chunks-th next chunk; one
 * shot.  The job that meets the fault returns an error naming the device, the other replicas finish their chunks, the
 * witnesses of the failed replica's in-flight chunks are wiped, and the pool stays usable. */
int rlnamd_pool_inject_fault(rlnamd_pool* p, size_t replica, size_t after_chunks);
/* Failover.  rounds = 0 (default): a failing replica fails the job as described above.  rounds > 0: everything the
 * failed replica was handed in the job (finished or not) and what it had not reached of its shard is proved again by the
 * replicas that finished, up to `rounds` times per job; the failed replica is quarantined -- later jobs are cut among the
 * others -- until rlnamd_pool_revive.  The job fails only when no replica is left or the rounds are spent; its result
 * is index-identical to a job without faults.  rlnamd_pool_health: per replica, quarantined (0 / 1) and the number of
 * dispatches that ended in an error since the pool was built (either array may be NULL). */
int rlnamd_pool_set_failover(rlnamd_pool* p, int rounds);
/* Probation.  jobs = 0 (default): a quarantined replica stays out until rlnamd_pool_revive.  jobs > 0: it sits out that
 * many jobs and is then handed work again by itself (a replica that fails again is quarantined again, its chunks go to
 * the others as before); when every replica is quarantined all of them are tried again at once instead of failing the
 * call.  Behind an FFI object (config key "failover") this is on, 8 jobs unless "revive_after" says otherwise: a caller
 * of include/rln.h has no handle on the pool.  Replica 0 of an FFI object's pool is also its single-proof prover and
 * its tree's device: calls with n <= max_batch are NOT failed over. */
int rlnamd_pool_set_probation(rlnamd_pool* p, size_t jobs);
int rlnamd_pool_health(rlnamd_pool* p, int* quarantined_per_replica, size_t* failures_per_replica);
int rlnamd_pool_revive(rlnamd_pool* p, size_t replica);
int rlnamd_pool_verify_many(rlnamd_pool* p, size_t n, const uint8_t* proofs, const uint8_t* values_le, size_t n_values,
                            int threads, uint8_t* ok);
/* RCCL communicator for the one path with an exchange step, the config-5 MSM.  Multi-process (one process per GPU,
 * e.g. under torchrun): rank 0 calls rlnamd_comm_unique_id, the 128 bytes reach the other ranks by any means, every
 * rank calls rlnamd_comm_init_rank with its device current.  Single process: rlnamd_comm_init_all fills one
 * communicator per listed device (ncclCommInitAll; ordinals must differ). */
#define RLNAMD_COMM_ID_BYTES 128
typedef struct rlnamd_comm rlnamd_comm;
int rlnamd_comm_unique_id(uint8_t id[RLNAMD_COMM_ID_BYTES]);
int rlnamd_comm_init_rank(const uint8_t id[RLNAMD_COMM_ID_BYTES], int nranks, int rank, rlnamd_comm** out);
int rlnamd_comm_init_all(const int* devices, size_t n_devices, rlnamd_comm** out_array);
void rlnamd_comm_free(rlnamd_comm* c);
int rlnamd_comm_rank(rlnamd_comm* c);
int rlnamd_comm_ranks(rlnamd_comm* c);
/* The whole of config 5 on one rank: local Pippenger on this rank's points down to the 16 window sums, ONE
 * ncclAllGather of 2 KiB per rank over xGMI on the MSM's own stream (RCCL has no elliptic-curve reduce op: the
 * "all-reduce of bucket partials" is gather + local add), then the add + Horner fold on every rank.  Collective: every
 * rank of the communicator calls it.  ms[0] digits + sort, ms[1] buckets, ms[2] all-gather, ms[3] combine (HIP events). */
int rlnamd_msm_run_sharded(rlnamd_msm* m, rlnamd_comm* c, uint8_t out_xy_le[64], float ms[4]);
/* the same for a caller that owns several devices in one process: a thread per device, generated points
 * (rlnamd_msm_generate_mode at the rank's index range), `repeats` timed runs; ms[0..3] = max over devices of the stage times
 * of the last run, ms[4] = its wall time */
int rlnamd_msm_generated_multi(const int* devices, size_t n_devices, uint64_t seed, size_t n_total, uint32_t mode,
                               int repeats, uint8_t out_xy_le[64], float ms[5]);

#ifdef __cplusplus
}
#endif
#endif /* RLN_AMD_H */
