#!/usr/bin/env python3
"""bench.py -- RLN Groth16 proofs/s on MI355X (BASELINE.json metric), one JSON line on rank 0.

Timed region: H2D of fresh witness inputs -> witness graph -> QAP/NTT -> table MSMs -> finalize ->
D2H of proofs + public values, through rlnamd_prover_submit / rlnamd_prover_collect (include/rln_amd.h).  The comb
tables are resident (built before the timed region); the witness inputs are NOT: every step stages its inputs in pinned
host memory and copies them to the device inside the timer, and every step's results are copied out and kept.

  N = 1   a step is one batch of BATCH = 1024 proofs (BASELINE config 2).  The K timed steps prove K DISTINCT batches
          (witness indices 1024 k .. 1024 (k + 1) of the SplitMix64(0xC0FFEE) stream; the stream wraps after 64 batches),
          all workspace slots in flight.
  N > 1   BASELINE config 4: 65 536-index stream cut into contiguous shards, 8 192 per GPU; a step is one pass over the
          rank's shard (8 chunks of 1024, uploaded again every step).  Two launch modes:
            * under torchrun (WORLD_SIZE set; how the driver runs it): one process per GPU, barrier + device sync on
              both sides of the timed region, max over ranks; torch.distributed is plumbing only.
            * plain `python bench.py --gpus N`: ONE process drives N devices through rlnamd_pool (one host thread and
              one prover replica per device).  Exits non-zero when fewer than N devices are visible -- it never runs a
              smaller job silently.
          No data-path collective for proofs (independent units).  The one collective of this code base, the
          ncclAllGather of the config-5 MSM's window sums, is exercised after the timed region (config5 object;
          `rccl_ranks` = size of the communicator created through the C ABI).

Extra objects on the line (every fraction can be recomputed from the line itself plus profiles/):
  roofline      top level = the contract's object for k_msm29<G1>, the kernel with most of the step's work (61 % of a
                batch's VALU instructions): `frac` = `hbm_frac` = SURVEY 8(d)'s definition, algorithmic bytes per launch
                (`algorithmic_bytes_per_proof` x batch) / `launch_ms` / 8 TB/s -- the number BASELINE.json's north_star
                asks for.  Neither walk is HBM-bound: what binds is VALU ISSUE CYCLES at a power-limited clock, so
                `kernels` prices BOTH walks (k_msm29<G2> is the longest launch of a step) on both axes, and `whole_step`
                prices every kernel of one batch against the step time:
                  issue cycles = sum over instructions of the cycles one wave-instruction occupies its SIMD --
                  4 for the quarter-rate class (v_mad_u64_u32, v_mul_lo_u32, 64-bit adds / shifts, carries: measured
                  450 - 590 G wave-instr/s), 2 for plain 32-bit ops (measured ~1 020 G/s); the mix of one addition from the
                  ISA (profiles/r5_walk_isa_mix.json), the instruction counts from PMC (profiles/r5_pmc_walks.json), both
                  gated by a hash of the walk's sources.  `issue_frac` = issue cycles / (1024 SIMDs x 2.4 GHz x time):
                  cannot exceed 1 (the clock never exceeds 2.4 GHz); `issue_frac_at_held_clock` uses the clock the walk
                  kernels measured for themselves.  `peaks` names the two issue rates (the 4-cycle one is what
                  v_mad_u64_u32 can reach, 614.4 G/s; the 2-cycle one, 1 228.8 G/s, only plain 32-bit ops reach).
  stage_ms      {"overlapped": spans inside the pipelined timed region (each includes whatever shared the chip with it),
                 "alone": the same stages of ONE batch with nothing else in flight}
  sustained     the same stream continued for --sustained-seconds (default 20 s) behind the timed region: proofs/s, the
                clock the walks held, the ratio to `value` (which, at the driver's K = 20, is 0.9 s inside the boost window)
  finish        finish_zk_proof_with_rs from cached partial proofs (the reference's one published claim,
                rln/README.md:370-375): proofs/s at 1024 per batch and one finish per call, beside the full-proof
                figures and beside oracle/c's full / partial / finish on the host (`finish.cpu_baseline`); every finished
                proof byte-identical to the full proof of the same (witness, r, s); the partial points judged by oracle/c
  cpu_baseline  oracle/c (kind "port") on the host's usable cores, N = 1 only: config 2 (`value`, proofs/s) plus
                `config3` (2^20-leaf tree build on all cores, one single-leaf update + root, 1 000 scattered updates + root)
                and `config5` (msm_bigint Pippenger on all cores over a 2^20-point sample of the workload, x 16 stated)
  config3 / config5   the other single-GPU BASELINE configs, measured after the timed region with the prover released;
                their `correct` flags are judged by the ORACLE (tree roots / the closed form (sum k_i s_i) G), never by the
                library's own arithmetic.  config3 also carries the tree-mutation calls through the drop-in boundary
                (ffi_set_leaf / ffi_get_root: one update + root, 1 000 scattered updates + root) beside oracle/c.
"""
import argparse
import collections
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before torch / HIP initialise: see zerokit_amd/csrc/common.cpp

# per-proof algorithmic bytes
BYTES_PER_PROOF = 7468404
MSM_G1_BYTES_PER_PROOF = (5844 + 5844 + 8192 + 5838) * 96   # A, B1, H, L operands (point 64 B + scalar 32 B)
MSM_G2_BYTES_PER_PROOF = 5844 * (128 + 32)                  # B2 operands (G2 point 128 B + scalar 32 B) = 935 040
CONFIG3_BYTES = 792723424
HBM_PEAK_GBPS = 8000.0
SIMDS, CLOCK_NOMINAL_HZ = 1024, 2.4e9
# issue rates in wave-instructions per second over the chip at the nominal clock.  Measured (tools/microbench_dfma.hip,
# profiles/r5_microbench_dfma.txt, at the ~2.0 GHz the box held): v_mad_u64_u32 495 - 505 G/s, v_mul_lo_u32 / carries /
# 64-bit adds and shifts 530 - 590 G/s (4 cycles per wave-instruction); v_add_u32 / v_and_b32 / v_mov_b32 ~1 020 G/s (2)
QUARTER_RATE_PEAK_GINST = SIMDS * CLOCK_NOMINAL_HZ / 4 / 1e9   # 614.4: the mad slot (was mislabelled "VALU peak")
PLAIN_RATE_PEAK_GINST = SIMDS * CLOCK_NOMINAL_HZ / 2 / 1e9     # 1228.8: the guide's VALU issue rate, plain 32-bit ops only
ONE_TIME_KERNELS = ("k_table_build", "k_table_to29", "k_consts_to29", "k_fr_to29")   # constructor, not part of a batch
STREAM_WRAP = 64                                            # distinct batches kept in host memory
SHARD = 8192                                                # config 4: proofs per GPU


def walk_source_hash():
    """identifies the code the table walks are compiled from; the PMC passes under profiles/ record it, and the VALU
    view is only emitted when it matches (an instruction count belongs to one build).  Comments and white space do not
    count: they do not reach the compiler."""
    import re
    h = hashlib.sha256()
    for f in ("walk29.h", "walk29_impl.h", "fq29.h", "fq29_constants.h", "curve.h", "glv.h"):
        src = open(os.path.join(ROOT, "zerokit_amd", "csrc", f), "r").read()
        src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
        src = re.sub(r"//[^\n]*", " ", src)
        h.update(" ".join(src.split()).encode())
    return h.hexdigest()[:16]


def cpu_baseline(ws, rs, target_seconds=10.0, want=("config3", "config5")):
    """Times oracle/c (the C restatement of the arkworks CPU path) on bounded samples; returns the dict for the JSON
    line or None when the oracle library has not been built."""
    try:
        from oracle.c import binding as ob
    except Exception:
        return None
    try:
        out = ob.time_baseline(ws, rs, target_seconds)
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}
    cores = out["cores"]
    if "config3" not in want and "config5" not in want:
        return out
    try:   # config 3: FullMerkleTree::set_range over 2^20 leaves on all cores + the mutation calls (full_merkle_tree.rs)
        t = ob.tree_bench(20, 1 << 20, first_value=1, threads=cores, singles=TREE_SINGLES, scattered=TREE_SCATTERED)
        out["config3"] = {
            "build_s": round(t["build_s"], 3), "hashes_per_s": round(((1 << 20) - 1) / t["build_s"], 1),
            "paths_s": round(t["paths_s"], 4),
            "single_update_plus_root_ms": round(t["single_update_root_s"] * 1e3, 4),
            "scattered_updates_plus_root_ms": round(t["scattered_updates_root_s"] * 1e3, 3),
            "scattered_updates": TREE_SCATTERED, "cores": cores, "kind": "port",
            "sample": "full size: 2^20 leaves i -> i + 1, parents of a level spread over %d threads; one set() = 20 dependent "
                      "hashes on one core (mean of %d); %d set() calls at pseudo-random indices + one root read"
                      % (cores, TREE_SINGLES, TREE_SCATTERED),
            "root": hex(t["root"]), "root_after_updates": hex(t["root_after_scattered"])}
    except Exception as e:  # noqa: BLE001
        out["config3"] = {"error": str(e)}
    try:   # config 5: msm_bigint (windowed Pippenger, windows spread over the cores like ark-ec's `parallel`)
        log2s = 20
        pt, secs, gen = ob.msm_pippenger(0xC0FFEE, 0, 1 << log2s, threads=cores)
        ok = pt == ob.msm_expected(0xC0FFEE, 0, 1 << log2s)
        out["config5"] = {
            "sample_points": 1 << log2s, "sample_s": round(secs, 3),
            "estimated_s_at_2^24": round(secs * (1 << (24 - log2s)), 2), "cores": cores, "kind": "port",
            "points_per_s": round((1 << log2s) / secs, 1), "sample_correct": bool(ok),
            "sample": "2^%d points of the config-5 workload (x %d for 2^24: Pippenger is linear in n at fixed window "
                      "width; the 2^24 run itself would use a wider window, ~15 %% fewer additions per point), windows "
                      "spread over %d threads; generation of the points (%.1f s) untimed"
                      % (log2s, 1 << (24 - log2s), cores, gen)}
    except Exception as e:  # noqa: BLE001
        out["config5"] = {"error": str(e)}
    return out


TREE_SINGLES = 32        # single-leaf updates, each followed by a root read
TREE_SCATTERED = 1000    # scattered updates followed by ONE root read


def load_pmc(info, B):
    """the committed PMC passes of the walks, if they were taken on this schedule and this build of the walk"""
    for name in ("r5_pmc_walks.json", "r4_pmc_walks.json", "r3_pmc_walks.json", "r2_pmc_walks.json"):
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:  # noqa: BLE001
            continue
        sched = pm.get("schedule")
        if not sched or pm.get("walk_source_hash") != walk_source_hash():
            continue
        if (sched.get("window_bits"), sched.get("windows"), sched.get("window_bits_g2"), sched.get("windows_g2"),
                sched.get("glv"), sched.get("batch")) != (int(info.window_bits), int(info.windows),
                                                          int(info.window_bits_g2), int(info.windows_g2),
                                                          int(info.glv), B):
            continue
        pm["_file"] = "profiles/" + name
        return pm
    return None


def load_mix():
    """the ISA mix of one addition of the two walks (tools/isa_mix.py), if it belongs to this build of the walk"""
    try:
        mx = json.load(open(os.path.join(ROOT, "profiles", "r5_walk_isa_mix.json")))
    except Exception:  # noqa: BLE001
        return None
    return mx if mx.get("walk_source_hash") == walk_source_hash() else None


def issue_view(pm, mix, info, B, stage_ms, stage_alone, clock_timed, clock_alone, ms_per_step):
    """Both walks and the whole step against the bound that binds: VALU issue cycles (module docstring).  Returns
    (kernels dict, whole_step dict) or (None, None) when the committed PMC / ISA files do not match this build."""
    if pm is None or mix is None:
        return None, None
    spec = {"k_msm29<G1>": ("msm_g1", "g1_walk", int(info.g1_rows) * int(info.windows), MSM_G1_BYTES_PER_PROOF),
            "k_msm29<G2>": ("msm_g2", "g2_walk", int(info.g2_rows) * int(info.windows_g2), MSM_G2_BYTES_PER_PROOF)}
    kernels, walk_cycles, walk_mads = {}, {}, 0.0
    for tag, (stage, clk, adds_per_proof, alg) in spec.items():
        k, m = pm["kernels"].get(tag), mix["kernels"].get(tag)
        if not k or not m:
            return None, None
        wave_adds = adds_per_proof * B / 64
        insts = k["SQ_INSTS_VALU"] / (k["lane_additions_per_launch"] / 64) * wave_adds
        cycles = m["issue_cycles_per_wave_addition"] * wave_adds
        walk_cycles[tag] = cycles
        walk_mads += m["v_mad_u64_u32"] * wave_adds
        o = {"algorithmic_bytes_per_proof": alg, "algorithmic_bytes_per_launch": alg * B,
             "lane_additions_per_launch": adds_per_proof * B,
             "valu_insts_per_launch_G": round(insts / 1e9, 3), "valu_insts_per_wave_addition": round(insts / wave_adds, 1),
             "mad_u64_u32_share": m["mad_u64_u32_share"], "quarter_rate_share": m["quarter_rate_share"],
             "issue_cycles_per_wave_addition": m["issue_cycles_per_wave_addition"],
             "issue_cycles_per_launch_G": round(cycles / 1e9, 3),
             "vgprs": m["registers"].get("NumVgprs"), "scratch_bytes_per_lane": m["registers"].get("scratch_bytes_per_lane"),
             "waves_per_simd": m["registers"].get("Occupancy")}
        if "traffic_bytes_per_launch" in k:
            o["traffic_GB_per_launch"] = round(k["traffic_bytes_per_launch"] / 1e9, 3)
        for label, t_ms, mhz in (("", stage_ms.get(stage, 0.0), clock_timed.get(clk, 0.0)),
                                 ("_alone", stage_alone.get(stage, 0.0), clock_alone.get(clk, 0.0))):
            if t_ms <= 0:
                continue
            t = t_ms * 1e-3
            o["launch_ms" + label] = round(t_ms, 3)
            o["hbm_GBps" + label] = round(alg * B / t / 1e9, 3)
            o["hbm_frac" + label] = round(alg * B / t / 1e9 / HBM_PEAK_GBPS, 6)
            o["valu_Ginst_per_s" + label] = round(insts / t / 1e9, 1)
            o["issue_frac" + label] = round(cycles / (SIMDS * CLOCK_NOMINAL_HZ * t), 4)
            if mhz > 0:
                o["clock_mhz" + label] = round(mhz, 1)
                o["issue_frac_at_held_clock" + label] = round(cycles / (SIMDS * mhz * 1e6 * t), 4)
        kernels[tag] = o
    # every kernel of one batch: instruction counts from PMC x launches per batch; the walks at their ISA mix, the rest
    # between "every instruction the counters do not class as 64-bit integer is a 2-cycle one" and "all are 4-cycle ones"
    n_all, lo, hi, per_kernel = 0.0, 0.0, 0.0, {}
    for tag, k in pm["kernels"].items():
        lpb = k.get("launches_per_batch")
        if not lpb or tag in ONE_TIME_KERNELS or "SQ_INSTS_VALU" not in k:
            continue
        n = k["SQ_INSTS_VALU"] * lpb
        n_all += n
        if tag in walk_cycles:
            c_lo = c_hi = walk_cycles[tag]
        else:
            q = min(k.get("SQ_INSTS_VALU_INT64", 0.0) * lpb, n)
            c_lo, c_hi = 4 * q + 2 * (n - q), 4 * n
        lo += c_lo
        hi += c_hi
        per_kernel[tag] = round(n / 1e9, 4)
    t = ms_per_step * 1e-3
    held = [v for v in clock_timed.values() if v > 0]
    mhz = sum(held) / len(held) if held else 0.0
    denom = SIMDS * CLOCK_NOMINAL_HZ * t
    whole = {"what": "every kernel of one 1024-proof batch (PMC instruction counts x launches per batch) against ms_per_step",
             "valu_insts_per_batch_G": round(n_all / 1e9, 3), "valu_insts_by_kernel_G": dict(sorted(per_kernel.items(), key=lambda kv: -kv[1])),
             "walks_share_of_insts": round(sum(per_kernel.get(w, 0) for w in walk_cycles) * 1e9 / n_all, 4),
             "valu_Ginst_per_s": round(n_all / t / 1e9, 1),
             "issue_cycles_per_batch_G": [round(lo / 1e9, 3), round(hi / 1e9, 3)],
             "issue_frac": [round(lo / denom, 4), round(hi / denom, 4)],
             "mad_u64_u32_slot_frac": round(4 * walk_mads / denom, 4),
             "note": "issue_frac = [lower, upper] bound of busy VALU issue cycles / (1024 SIMDs x 2.4 GHz x ms_per_step): the "
                     "walks (85 % of the instructions) at their measured ISA mix, the other kernels between 2 and 4 cycles per "
                     "instruction; mad_u64_u32_slot_frac = the share of all issue cycles that v_mad_u64_u32 of the two walks "
                     "alone occupies"}
    if mhz > 0:
        whole["clock_mhz_under_the_walks"] = round(mhz, 1)
        whole["issue_frac_at_held_clock"] = [round(lo / (SIMDS * mhz * 1e6 * t), 4), round(hi / (SIMDS * mhz * 1e6 * t), 4)]
    return kernels, whole


def tree_update_stream(n):
    """the mutation stream oracle_tree_bench applies after its build: TREE_SINGLES single updates (a root read behind
    each), then TREE_SCATTERED updates and one root read (the product's own generator; oracle/c has its own)"""
    from zerokit_amd import workload
    return (workload.tree_update_stream(n, TREE_SINGLES, 0x7EE, 0x5157000000000000),
            workload.tree_update_stream(n, TREE_SCATTERED, 0x5CA7, 0x5CA7000000000000))


def measure_config3(steps=3):
    """BASELINE config 3 on this GPU: 2^20-leaf build + 2^20 membership paths, every path recomputed on device; then
    the tree-mutation calls: on the same full tree through the extension API (roots comparable with the oracle's run of
    the same stream) and through the drop-in boundary (ffi_set_leaf / ffi_get_root)"""
    from zerokit_amd.batch import PoseidonTree
    depth, n = 20, 1 << 20
    t = PoseidonTree(depth)
    try:
        t.bench(n, 1, verify=False)
        rs = [t.bench(n, 1, verify=False) for _ in range(steps)]
        bad = t.bench(n, 1, verify=True)["bad"]
        root = t.root()
        singles, scattered = tree_update_stream(n)
        ts = []
        for i, v in singles:
            t0 = time.perf_counter()
            t.set_leaves([(i, v)])
            t.root()
            ts.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        t.set_leaves(scattered)
        root_after = t.root()
        scat_ms = (time.perf_counter() - t0) * 1e3
    finally:
        t.close()
    b = sum(r["build_ms"] for r in rs) / len(rs)
    p = sum(r["proofs_ms"] for r in rs) / len(rs)
    path_bytes = n * (depth * 32 + depth)
    out = {"workload": "config 3: 2^20-leaf Poseidon tree build + 2^20 membership paths", "build_ms": round(b, 3),
           "paths_ms": round(p, 3), "hashes_per_s": round((n - 1) / (b * 1e-3), 1),
           "achieved_GBps": round(CONFIG3_BYTES / ((b + p) * 1e-3) / 1e9, 2),
           "paths_failed_device_verification": bad,
           "root": hex(root), "root_after_updates": hex(root_after),
           "updates_ext_api": {"single_update_plus_root_ms_median": round(sorted(ts)[len(ts) // 2], 3),
                               "scattered_updates_plus_root_ms": round(scat_ms, 3), "scattered_updates": len(scattered),
                               "what": "rlnamd_tree_set_leaves + rlnamd_tree_root on the full 2^20-leaf tree: ONE bottom-up pass "
                                       "over the union of the dirty paths"},
           "roofline": {"bound": "hbm", "kernel": "k_proofs_lds (path emission)",
                        "algorithmic_bytes_per_launch": path_bytes,
                        "achieved": round(path_bytes / (p * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(path_bytes / (p * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}}
    out["updates_ffi"] = measure_tree_updates_ffi()
    # the judge of `correct` is the oracle (cpu_baseline.config3 carries the same two roots when it ran; here the check is
    # made directly so that the flag does not depend on that leg)
    try:
        from oracle.c import binding as ob
        o = ob.Tree(depth)
        o.set_range(0, list(range(1, n + 1)), threads=ob.usable_cores())
        ok_root = o.root() == root
        for i, v in singles + scattered:
            o.set(i, v)
        out["correct"] = bool(bad == 0 and ok_root and o.root() == root_after and out["updates_ffi"].get("correct", False))
        out["checked_by"] = "oracle/c FullMerkleTree restatement: build root, root after %d updates; every path on device" \
                            % (len(singles) + len(scattered))
        o.close()
    except Exception as e:  # noqa: BLE001
        out["correct"] = False
        out["checked_by"] = "oracle unavailable: %s" % e
    return out


def measure_tree_updates_ffi():
    """ffi_set_leaf / ffi_get_root through include/rln.h on a depth-20 object: the most frequent tree calls of a caller
    (rln/src/ffi/ffi_tree.rs).  Writes are recorded and hashed by the first reader in one pass (deferred, coalesced)."""
    from zerokit_amd.public import RLN
    n = 1 << 20
    singles, scattered = tree_update_stream(n)
    r = RLN(20)
    try:
        r.get_root()
        ts = []
        for i, v in singles:
            t0 = time.perf_counter()
            r.set_leaf(i, v)
            r.get_root()
            ts.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        for i, v in scattered:
            r.set_leaf(i, v)
        t1 = time.perf_counter()
        root = r.get_root()
        t2 = time.perf_counter()
        proof = r.get_merkle_proof(scattered[0][0])
    finally:
        del r
    out = {"single_update_plus_root_ms_median": round(sorted(ts)[len(ts) // 2], 3),
           "single_update_plus_root_ms_min": round(min(ts), 3),
           "scattered_updates_plus_root_ms": round((t2 - t0) * 1e3, 3), "scattered_updates": len(scattered),
           "of_which_set_leaf_calls_ms": round((t1 - t0) * 1e3, 3), "root_read_ms": round((t2 - t1) * 1e3, 3),
           "what": "ffi_set_leaf x k then ffi_get_root on an otherwise empty depth-20 tree"}
    try:
        from oracle.c import binding as ob
        o = ob.Tree(20)
        for i, v in singles + scattered:
            o.set(i, v)
        pe, pb = o.proof(scattered[0][0])
        out["correct"] = bool(o.root() == root and list(proof[0]) == pe and list(proof[1]) == pb)
        o.close()
    except Exception as e:  # noqa: BLE001
        out["correct"] = False
        out["error"] = str(e)
    return out


def measure_config5(comm, rank, world, steps=3, log2n=24):
    """BASELINE config 5: one 2^24-point G1 MSM split by point index over the ranks of `comm`; local Pippenger to the
    window sums, ncclAllGather through the C ABI, local fold; compared with the scalar-side closed form"""
    from zerokit_amd.batch import MsmG1
    n_total = 1 << log2n
    lo, hi = n_total * rank // world, n_total * (rank + 1) // world     # contiguous slices by point index
    m = MsmG1(hi - lo)
    try:
        m.generate(0xC0FFEE, lo, hi - lo)
        m.run_sharded(comm)
        times, st = [], {}
        for _ in range(steps):
            t0 = time.perf_counter()
            res, st = m.run_sharded(comm)
            times.append(time.perf_counter() - t0)
    finally:
        m.close()
    ms = sum(times) / len(times) * 1e3
    gbps = n_total * 96 / (ms * 1e-3) / 1e9
    # what each of eight GPUs would run: ONE 2^(log2n - 3)-point shard through the same call, on this device,
    # with this one-rank communicator -- the 8-way time is that plus the all-gather of 8 x 2 KiB
    shard, shard_res = None, None
    if world == 1 and log2n >= 8:
        ns = n_total // 8
        m = MsmG1(ns)
        try:
            m.generate(0xC0FFEE, 0, ns)
            m.run_sharded(comm)
            ts, sst = [], {}
            for _ in range(max(steps, 3)):
                t0 = time.perf_counter()
                shard_res, sst = m.run_sharded(comm)
                ts.append(time.perf_counter() - t0)
        finally:
            m.close()
        sms = sum(ts) / len(ts) * 1e3
        shard = {"points": ns, "ms": round(sms, 3), "stage_ms": {k: round(v, 3) for k, v in sst.items()},
                 "projected_8_way_ms": round(sms, 3),
                 "projected_speedup_on_8_gpus": round(ms / sms, 2),
                 "note": "one of the eight contiguous slices by point index, timed alone on this device through "
                         "rlnamd_msm_run_sharded (sort, buckets, a one-rank ncclAllGather, the fold); on eight devices the "
                         "all-gather moves 8 x 2 KiB over xGMI instead (tens of microseconds): NOT measured on hardware"}
    # the same Pippenger on G2 (north_star: "windowed Pippenger MSM on G1/G2"): 2^(log2n - 2) points of the twist
    g2, g2_res = None, None
    if world == 1 and log2n >= 8:
        from zerokit_amd.batch import MsmG2
        n2 = n_total // 4
        m = MsmG2(n2)
        try:
            m.generate(0xC0FFEE, 0, n2)
            m.run_sharded(comm)
            ts, gst = [], {}
            for _ in range(max(steps, 3)):
                t0 = time.perf_counter()
                g2_res, gst = m.run_sharded(comm)
                ts.append(time.perf_counter() - t0)
        finally:
            m.close()
        g2 = {"points": n2, "ms": round(sum(ts) / len(ts) * 1e3, 3), "stage_ms": {k: round(v, 3) for k, v in gst.items()},
              "G2_additions_per_s_G": round(n2 * 16 / (gst.get("buckets_ms", 0.0) * 1e-3) / 1e9, 2) if gst.get("buckets_ms") else None}
    ok, judge = True, None
    if rank == 0:   # the closed form (sum k_i s_i) G comes from the ORACLE; the library states no expected value of its own
        try:
            from oracle.c import binding as ob
            ok, judge = res == ob.msm_expected(0xC0FFEE, 0, n_total), "oracle/c closed form (sum k_i s_i mod r) G"
            if shard is not None:
                shard["correct"] = bool(shard_res == ob.msm_expected(0xC0FFEE, 0, n_total // 8))
                ok = ok and shard["correct"]
            if g2 is not None:
                g2["correct"] = bool(g2_res == ob.msm_expected_g2(0xC0FFEE, 0, n_total // 4))
                ok = ok and g2["correct"]
        except Exception as e:  # noqa: BLE001
            ok, judge = False, "oracle unavailable: %s" % e
    return {"workload": "config 5: single 2^%d-point BN254 G1 MSM, %d-way split, ncclAllGather of window sums"
                        % (log2n, world),
            "ms": round(ms, 3), "rccl_ranks": comm.ranks(), "correct": bool(ok), "checked_by": judge,
            "stage_ms_rank0": {k: round(v, 3) for k, v in st.items()},
            "fold": os.environ.get("RLNAMD_MSM_FOLD", "host") + " (the 240 dependent doublings of the last step: 0.05 ms on a "
                    "host core, 1.5 - 1.9 ms on a lone GPU lane)",
            "shard_2^%d" % (log2n - 3): shard,
            "g2_2^%d" % (log2n - 2): g2,
            "roofline": {"bound": "hbm", "limiter": "valu issue (bucket additions)",
                         "algorithmic_bytes_per_launch": n_total * 96, "achieved": round(gbps, 2),
                         "peak": HBM_PEAK_GBPS * world, "unit": "GB/s", "frac": round(gbps / (HBM_PEAK_GBPS * world), 5)}}


def _stream_proofs_per_s(prover, batches, n_batches, B, collect_public=None):
    """n_batches submits over the distinct `batches` with every workspace slot in flight -> (proofs/s, {k: raw result})"""
    import collections as _c
    nslots, inflight, results = prover.n_slots(), _c.deque(), {}

    def take():
        t, kk = inflight.popleft()
        if collect_public is not None:
            collect_public(t, kk)
        results[kk] = prover.collect_raw(t, B)
    prover.sync()
    t0 = time.perf_counter()
    for j in range(n_batches):
        k = j % len(batches)
        if len(inflight) == nslots:
            take()
        inflight.append((prover.submit(batches[k][0], batches[k][1])[0], k))
    while inflight:
        take()
    prover.sync()
    return n_batches * B / (time.perf_counter() - t0), results


def _partial_inputs(prover, packed_inputs, n):
    """the partial witness of witness.rs:887-937 as an inputs buffer: the message-dependent slots (messageId, x,
    externalNullifier) zeroed -- evaluate_partial never reads them"""
    import numpy as np
    a = np.frombuffer(packed_inputs, dtype=np.uint8).reshape(n, prover.inputs_size, 32).copy()
    for name in ("messageId", "x", "externalNullifier"):
        off, ln = prover.slots[name]
        a[:, off:off + ln, :] = 0
    return a.tobytes()


def measure_finish(prover, batches, full_results, B, full_rate, full_latency_ms, K=16):
    """SURVEY 8(f) rank 2, the reference's one published performance claim (rln/README.md:370-375: finishing a cached
    partial proof is "roughly 2.5-3x faster" than a full proof; criterion targets rln/benches/partial_proof.rs:57-75):
    partial proofs of the members computed once (RLNAMD_MODE_PARTIAL, timed as its own rate), then ONLY
    finish_zk_proof_with_rs per message (RLNAMD_MODE_FINISH) -- K batches of B streamed exactly like the headline, and
    one finish per call.  Judges: every finished proof of a batch must be byte-identical to the FULL proof the timed
    region made from the same (witness, r, s); a sample is verified on the host."""
    nb = min(len(batches), 4)
    zero_rs = bytes(64 * B)
    prover.sync()
    t0 = time.perf_counter()
    tk = [prover.submit(_partial_inputs(prover, batches[k][0], B), zero_rs, 1)[0] for k in range(nb)]
    parts = [prover.collect_partial(t, B) for t in tk]
    prover.sync()
    partial_s = time.perf_counter() - t0
    nslots, inflight, res = prover.n_slots(), collections.deque(), {}

    def take():
        t, kk = inflight.popleft()
        res[kk] = prover.collect_raw(t, B)

    def stream(n_batches):
        prover.sync()
        t1 = time.perf_counter()
        for j in range(n_batches):
            k = j % nb
            if len(inflight) == nslots:
                take()
            inflight.append((prover.submit(batches[k][0], batches[k][1], 2, parts[k])[0], k))
        while inflight:
            take()
        prover.sync()
        return n_batches * B / (time.perf_counter() - t1)
    stream(2)
    prover.walk_clock_mhz()
    rate = stream(K)
    clock = prover.walk_clock_mhz()
    stage = prover.stage_ms()
    same = all(k in full_results and res[k][0] == full_results[k][0] and not any(res[k][2]) for k in res) and len(res) == nb
    vp, vv = [], []
    for k, (proofs, values, errs) in sorted(res.items()):
        for i in (0, B - 1):
            vp.append(proofs[128 * i:128 * i + 128])
            vv.append([int.from_bytes(values[160 * i + 32 * q:160 * i + 32 * q + 32], "little") for q in range(5)])
    verified = bool(all(prover.verify_many(vp, vv)))
    # one finish per call: submit + collect with nothing else in flight.  (a) with the partial run's cache handle
    # (rlnamd_prover_collect_partial_cached / _submit_finish: the interpreter walks only the cone of the witness graph that
    # depends on the message), (b) without one: the whole graph again, as the reference's finish_zk_proof_with_rs does
    n1 = prover.inputs_size * 32
    one_in, one_rs = batches[0][0][:n1], batches[0][1][:64]
    one_pin, tp = _partial_inputs(prover, one_in, 1), []
    for i in range(7):   # the third criterion target of rln/benches/partial_proof.rs:57-75: one partial proof per call
        t1 = time.perf_counter()
        t, _ = prover.submit(one_pin, bytes(64), 1)
        one_pp, one_h, _ = prover.collect_partial_cached(t, 1)
        tp.append((time.perf_counter() - t1) * 1e3)
        if i < 6:
            prover.release_partial(one_h)
    cone0 = prover.partial_cache_info()
    ts, ts_full = [], []
    same1 = bool(one_pp[0] == parts[0][0])
    for i in range(22):
        cached = i % 2 == 0
        t1 = time.perf_counter()
        t, _ = prover.submit_finish(one_in, one_rs, one_pp, one_h if cached else [0])
        pr, _, er = prover.collect_raw(t, 1)
        if i >= 4:
            (ts if cached else ts_full).append((time.perf_counter() - t1) * 1e3)
        same1 = bool(same1 and 0 in full_results and pr[:128] == full_results[0][0][:128] and not any(er))
    cone1 = prover.partial_cache_info()
    prover.release_partial(one_h)
    lat, lat_full = sorted(ts)[len(ts) // 2], sorted(ts_full)[len(ts_full) // 2]
    took_cone = cone1["cone_batches"] - cone0["cone_batches"] == 11 and one_h[0] != 0
    out = {"what": "finish_zk_proof_with_rs from cached partial proofs (protocol/proof.rs:821-849): %d batches of %d streamed, "
                   "H2D of inputs + partial points and D2H of proofs inside; the same witnesses, r, s as the headline's first "
                   "%d batches" % (K, B, nb),
           "proofs_per_s": round(rate, 1), "ms_per_batch": round(B / rate * 1e3, 3),
           "full_proofs_per_s": round(full_rate, 1), "speedup_over_full": round(rate / full_rate, 3),
           "single_call_ms_median": round(lat, 3), "single_call_ms_min": round(min(ts), 3),
           "single_call_whole_graph_ms_median": round(lat_full, 3),
           "single_call_took_the_cone": bool(took_cone),
           "witness_program_steps": {"cone": cone1["cone_steps"], "full": cone1["full_steps"], "cone_nodes": cone1["cone_nodes"],
                                     "cache_entry_bytes": cone1["entry_bytes"], "cache_entries": cone1["capacity"]},
           "full_single_call_ms_median": full_latency_ms,
           "single_call_speedup_over_full": round(full_latency_ms / lat, 3) if full_latency_ms else None,
           "partial_generation_proofs_per_s": round(nb * B / partial_s, 1),
           "partial_single_call_ms_median": round(sorted(tp[2:])[len(tp[2:]) // 2], 3),
           "reference_claim": "finish roughly 2.5-3x faster than a full proof (rln/README.md:370-375), one proof per call on a CPU",
           "byte_identical_to_the_full_proofs": bool(same), "single_call_byte_identical": same1,
           "verified": verified, "verified_proofs": len(vp),
           "stage_ms_overlapped": {k: round(v, 3) for k, v in stage.items()},
           "shader_clock_mhz": {k: round(v, 1) for k, v in clock.items()}}
    out["correct"] = bool(same and same1 and verified)
    return out, parts


def cpu_finish_baseline(ws, rs, cores):
    """oracle/c on the host: one full proof, one partial proof and one finish on ONE thread (the criterion targets of
    rln/benches/partial_proof.rs:57-75), then `cores` finishes, one per thread"""
    from oracle.c import binding as ob
    c = ob.Circuit(20)
    packed = [c.pack(w) for w in ws]
    rsb = [r.to_bytes(32, "little") + s_.to_bytes(32, "little") for r, s_ in rs]
    t0 = time.perf_counter()
    full = c.prove_packed(packed[0], rs[0][0], rs[0][1])["proof"]
    t1 = time.perf_counter()
    part = c.prove_partial_packed(packed[0])
    t2 = time.perf_counter()
    fin = c.finish_packed(packed[0], rs[0][0], rs[0][1], part)
    t3 = time.perf_counter()
    n = min(len(ws), max(cores, 2))
    parts = [part] + [c.prove_partial_packed(p) for p in packed[1:n]]
    secs, proofs = c.finish_many_packed(b"".join(packed[:n]), b"".join(rsb[:n]), b"".join(parts), threads=cores)
    return {"kind": "port", "cores": cores, "finish_proofs_per_s": round(n / secs, 3),
            "single_thread_ms": {"full": round((t1 - t0) * 1e3, 2), "partial": round((t2 - t1) * 1e3, 2),
                                 "finish": round((t3 - t2) * 1e3, 2)},
            "finish_speedup_over_full": round((t1 - t0) / (t3 - t2), 3),
            "finish_equals_full": bool(fin == full),
            "sample": "%d finishes of the config-2 witnesses, one per thread on %d threads (oracle/c: the four MSMs over the "
                      "rows evaluate_partial leaves unknown + all of h, full witness recomputed as the reference does)" % (n, cores)}, parts[0], fin


def operating_points_main(args):
    """VERDICT r4 item 5 -- the operating points as measured data (outside any timed region of the headline): for the comb
    schedules 8 / 120010 (the default object) / 12 / 114 / 7150114 (the bench) the table size, the constructor's time and
    where it goes, proofs/s over a stream of distinct 1024-proof batches, and one proof per call; then TWO circuits
    resident on one device (the single-message circuit at 114 and the multi-message-id circuit at 10) proving alternating
    1024-proof batches.  Every row's sample of proofs is verified on the host."""
    from zerokit_amd import lib, workload
    from zerokit_amd._native import check
    from zerokit_amd.batch import BatchProver
    B, nb, K = args.batch, 6, max(args.steps, 12)
    rows = []
    packed = None
    for wb in (8, 120010, 12, 114, 7150114):
        t0 = time.time()
        try:
            p = BatchProver(max_batch=B, window_bits=wb)
        except Exception as e:  # noqa: BLE001
            rows.append({"window_bits": wb, "error": str(e)})
            continue
        init_s = time.time() - t0
        try:
            if packed is None:
                packed = [workload.config2_packed(p.slots, p.inputs_size, B * k, B) for k in range(nb)]
            _stream_proofs_per_s(p, packed, 3, B)
            rate, res = _stream_proofs_per_s(p, packed, K, B)
            vp, vv = [], []
            for k, (proofs, values, errs) in sorted(res.items()):
                for i in (0, B - 1):
                    vp.append(proofs[128 * i:128 * i + 128])
                    vv.append([int.from_bytes(values[160 * i + 32 * q:160 * i + 32 * q + 32], "little") for q in range(5)])
            ok = all(not any(r[2]) for r in res.values()) and all(p.verify_many(vp, vv))
            one = workload.config2_packed(p.slots, p.inputs_size, 0, 1)
            ts = []
            for i in range(7):
                t1 = time.perf_counter()
                t, _ = p.submit(*one)
                p.collect_raw(t, 1)
                ts.append((time.perf_counter() - t1) * 1e3)
            rows.append({"window_bits": wb, "windows_g1": int(p.info.windows), "windows_g2": int(p.info.windows_g2),
                         "table_gib": round(p.info.table_bytes / 2**30, 2), "max_batch": B,
                         "init_s": round(init_s, 2), "init_ms": p.init_ms(),
                         "proofs_per_s": round(rate, 1), "ms_per_1024": round(B / rate * 1e3, 2),
                         "single_proof_ms_median": round(sorted(ts[2:])[len(ts[2:]) // 2], 3), "verified": bool(ok),
                         "verified_proofs": len(vp)})
        finally:
            p.close()
    # two circuits on one device
    co = None
    try:
        t0 = time.time()
        a = BatchProver(max_batch=B, window_bits=114)
        m = BatchProver(max_batch=B, window_bits=10, depth=20, multi=True)
        init_s = time.time() - t0
        try:
            named, rs = workload.circuit_range(0, 2 * B, 20, True)
            mp = [(m.pack_named_inputs(named[k * B:(k + 1) * B]),
                   b"".join(r.to_bytes(32, "little") + s_.to_bytes(32, "little") for r, s_ in rs[k * B:(k + 1) * B])) for k in range(2)]
            pubs = {}

            def grab(t, kk):
                buf = C.create_string_buffer(32 * m.num_public * B)
                check(lib().rlnamd_prover_collect_public(m._h, t, B, buf))
                pubs[kk] = buf.raw
            _stream_proofs_per_s(a, packed, 2, B)
            _stream_proofs_per_s(m, mp, 2, B)
            alone_a, _ = _stream_proofs_per_s(a, packed, K, B)
            alone_m, _ = _stream_proofs_per_s(m, mp, K, B, grab)
            # alternating: one batch of each circuit at a time, both provers' slots in flight
            import collections as _c
            qa, qm, ra, rm = _c.deque(), _c.deque(), {}, {}
            a.sync()
            m.sync()
            t1 = time.perf_counter()
            for j in range(K):
                if len(qa) == a.n_slots():
                    t, kk = qa.popleft()
                    ra[kk] = a.collect_raw(t, B)
                qa.append((a.submit(*packed[j % nb])[0], j % nb))
                if len(qm) == m.n_slots():
                    t, kk = qm.popleft()
                    grab(t, kk)
                    rm[kk] = m.collect_raw(t, B)
                qm.append((m.submit(*mp[j % 2])[0], j % 2))
            while qa:
                t, kk = qa.popleft()
                ra[kk] = a.collect_raw(t, B)
            while qm:
                t, kk = qm.popleft()
                grab(t, kk)
                rm[kk] = m.collect_raw(t, B)
            a.sync()
            m.sync()
            both = 2 * K * B / (time.perf_counter() - t1)
            q = m.num_public
            ok = all(not any(r[2]) for r in list(ra.values()) + list(rm.values()))
            for kk, r in rm.items():
                for i in (0, B - 1):
                    pub = [int.from_bytes(pubs[kk][32 * (i * q + j):32 * (i * q + j + 1)], "little") for j in range(q)]
                    ok = ok and m.verify_public(r[0][128 * i:128 * i + 128], pub)
            for kk, r in ra.items():
                v = [int.from_bytes(r[1][32 * j:32 * j + 32], "little") for j in range(5)]
                ok = ok and a.verify(r[0][:128], v)
            co = {"what": "the depth-20 single-message circuit (schedule 114) and the depth-20 multi-message-id circuit (max_out 4, "
                          "schedule 10) resident together on one device, 1024-proof workspaces each",
                  "table_gib": [round(a.info.table_bytes / 2**30, 2), round(m.info.table_bytes / 2**30, 2)],
                  "init_s_both": round(init_s, 2),
                  "proofs_per_s_single_circuit_alone": round(alone_a, 1), "proofs_per_s_multi_circuit_alone": round(alone_m, 1),
                  "proofs_per_s_alternating_batches_total": round(both, 1),
                  # equal numbers of proofs of both circuits: with the chip to itself a pair (one proof of each) costs
                  # 1 / single + 1 / multi seconds -- the harmonic mean is what a perfect time-sharing of the device gives
                  "time_weighted_mix_proofs_per_s": round(2.0 / (1.0 / alone_a + 1.0 / alone_m), 1),
                  "alternating_over_time_weighted_mix": round(both / (2.0 / (1.0 / alone_a + 1.0 / alone_m)), 4),
                  "device_shared_flags": [a.device_shared(), m.device_shared()],
                  "verified": bool(ok)}
        finally:
            a.close()
            m.close()
    except Exception as e:  # noqa: BLE001
        co = {"error": str(e)}
    # the other shipped circuits at a throughput point of their own (f3 had no throughput figure before round 5)
    others = []
    for label, depth, multi, wb in (("depth-20 multi-message-id (max_out 4)", 20, True, 114), ("depth-10 single", 10, False, 7150114)):
        try:
            t0 = time.time()
            q = BatchProver(max_batch=B, window_bits=wb, depth=depth, multi=multi)
            init_s = time.time() - t0
            try:
                named, rs = workload.circuit_range(0, 2 * B, depth, multi)
                bt = [(q.pack_named_inputs(named[k * B:(k + 1) * B]),
                       b"".join(r.to_bytes(32, "little") + s_.to_bytes(32, "little") for r, s_ in rs[k * B:(k + 1) * B])) for k in range(2)]
                pubs = {}

                def grab2(t, kk):
                    buf = C.create_string_buffer(32 * q.num_public * B)
                    check(lib().rlnamd_prover_collect_public(q._h, t, B, buf))
                    pubs[kk] = buf.raw
                _stream_proofs_per_s(q, bt, 2, B)
                rate, res = _stream_proofs_per_s(q, bt, K, B, grab2)
                npub = q.num_public
                ok = all(not any(r[2]) for r in res.values())
                for kk, r in res.items():
                    for i in (0, B - 1):
                        pub = [int.from_bytes(pubs[kk][32 * (i * npub + j):32 * (i * npub + j + 1)], "little") for j in range(npub)]
                        ok = ok and q.verify_public(r[0][128 * i:128 * i + 128], pub)
                # ... and judged by the ORACLE at this batch size and on these tables (at 1 024 per batch the
                # other circuits had only the host verifier): proofs 0, 511 and 1 023 of both batches against oracle/c
                judged, same = 0, True
                try:
                    from oracle.c import binding as ob
                    oc = ob.Circuit(depth, multi=multi)
                    for kk, r in res.items():
                        for i in (0, B // 2 - 1, B - 1):
                            w = oc.pack_named(named[kk * B + i])
                            rr, ss = rs[kk * B + i]
                            o = oc.prove_packed(w, rr, ss)
                            pub = [int.from_bytes(pubs[kk][32 * (i * npub + j):32 * (i * npub + j + 1)], "little") for j in range(npub)]
                            same = same and o["proof"] == r[0][128 * i:128 * i + 128] and o["public_inputs"] == pub
                            judged += 1
                except Exception as e:  # noqa: BLE001
                    same, judged = False, str(e)
                others.append({"circuit": label, "window_bits": wb, "table_gib": round(q.info.table_bytes / 2**30, 2),
                               "init_s": round(init_s, 2), "proofs_per_s": round(rate, 1), "verified": bool(ok),
                               "equal_to_oracle_c": bool(same), "proofs_judged_by_oracle_c": judged})
            finally:
                q.close()
        except Exception as e:  # noqa: BLE001
            others.append({"circuit": label, "window_bits": wb, "error": str(e)})
    # the CPU port on the same two circuits (oracle/c, one proof per host thread): the multi-message-id circuit had no CPU
    # figure before round 5
    try:
        from oracle.c import binding as ob
        cores = ob.usable_cores()
        k = max(cores, 16)
        o1, om = ob.Circuit(20), ob.Circuit(20, multi=True)
        n1, r1 = workload.circuit_range(0, k, 20, False)
        nm, rm_ = workload.circuit_range(0, k, 20, True)
        rsb = lambda rs_: b"".join(r.to_bytes(32, "little") + s_.to_bytes(32, "little") for r, s_ in rs_)  # noqa: E731
        s1, _, _ = o1.prove_many_packed(b"".join(o1.pack_named(w) for w in n1), rsb(r1), threads=cores)
        sm, _, _ = om.prove_many_packed(b"".join(om.pack_named(w) for w in nm), rsb(rm_), threads=cores)
        if co is not None:
            co["cpu_port"] = {"kind": "port", "cores": cores, "sample": "%d proofs of each circuit, one per host thread" % k,
                              "single_circuit_proofs_per_s": round(k / s1, 2), "multi_circuit_proofs_per_s": round(k / sm, 2)}
    except Exception as e:  # noqa: BLE001
        if co is not None:
            co["cpu_port"] = {"error": str(e)}
    name = C.create_string_buffer(128)
    lib().rlnamd_device_name(name, 128)
    OUT.emit({"metric": "RLN Groth16 proofs/sec (BN254, h=20) -- operating points (side measurement)", "unit": "proofs/s",
              "device": name.value.decode(), "batch": B, "batches_per_row": K, "operating_points": rows,
              "two_circuits_on_one_device": co, "other_circuits": others})


def merkle_main(args):
    OUT.emit(measure_config3(max(args.steps, 1)))


def make_comm(rank, world, dist):
    """an RCCL communicator created through the C ABI; the 128-byte id travels over torch.distributed (plumbing)"""
    from zerokit_amd.batch import Comm
    uid = [Comm.unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(uid, src=0)
    return Comm.init_rank(uid[0], world, rank)


def msm_main(args):
    """config 5 by itself (side measurement, one JSON line on rank 0)"""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:   # torch first: see main()
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    from zerokit_amd import lib
    from zerokit_amd._native import check
    check(lib().rlnamd_set_device(local_rank))
    comm = make_comm(rank, world, dist)
    out = measure_config5(comm, rank, world, max(args.steps, 1), int(os.environ.get("RLNAMD_MSM_LOG2", "24")))
    comm.close()
    if rank == 0:
        OUT.emit(out)
    if world > 1:
        dist.destroy_process_group()


def open_prover(B, factory):
    """GLV comb schedule g1 + 10000 * g2 over the 127-bit scalar halves: 114 = 15 + 8 x 14 bits (9 windows, 18 additions
    per G1 point), 715 = 7 x 16 + 15 bits (8 windows, 16 additions per G2 point): 228 GiB of fixed-base tables, sized for
    288 GB of HBM.  Smaller tables are tried if that does not fit."""
    wbits = int(os.environ.get("RLNAMD_WINDOW_BITS", "7150114"))
    for wb in dict.fromkeys([wbits, 114, 13, 12, 10]):
        try:
            return factory(wb)
        except Exception as e:  # noqa: BLE001
            print("bench: window schedule %d not available (%s)" % (wb, e), file=sys.stderr)
    raise SystemExit("bench: no table size fits this device")


def pool_main(args):
    """plain `python bench.py --gpus N` (no torchrun): one process, N devices, rlnamd_pool"""
    from zerokit_amd import lib, workload
    from zerokit_amd.batch import BatchProver, ProverPool
    N = args.gpus
    have = lib().rlnamd_device_count()
    devices = list(range(N))
    if "RLNAMD_BENCH_POOL_DEVICES" in os.environ:   # test hook for 1-GPU boxes: replicas sharing a device ("0,0")
        devices = [int(x) for x in os.environ["RLNAMD_BENCH_POOL_DEVICES"].split(",")]
        if len(devices) != N:
            raise SystemExit("bench: RLNAMD_BENCH_POOL_DEVICES must list --gpus devices")
    elif have < N:
        raise SystemExit("bench: --gpus %d but %d device(s) visible; refusing to run a smaller job" % (N, have))
    B = args.batch
    # input slots come from the graph; any single prover knows them -- read them from a tiny one on device 0 BEFORE the
    # pool's 228 GiB replicas exist
    probe = BatchProver(max_batch=64, window_bits=8)
    slots, ni = dict(probe.slots), probe.inputs_size
    probe.close()
    t0 = time.time()
    pool = open_prover(B, lambda wb: ProverPool(devices=devices, max_batch=B, window_bits=wb))
    init_s = time.time() - t0
    n = SHARD * N
    inputs, rsb = workload.config2_packed(slots, ni, 0, n)
    for _ in range(args.warmup):
        pool.prove_raw(inputs, rsb)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proofs, values, errs = pool.prove_raw(inputs, rsb)
    elapsed = time.perf_counter() - t0
    idx = list(range(0, n, 509))
    pub = [[int.from_bytes(values[160 * i + 32 * k:160 * i + 32 * k + 32], "little") for k in range(5)] for i in idx]
    ok = not any(errs) and all(pool.verify_many([proofs[128 * i:128 * i + 128] for i in idx], pub))
    value = n * args.steps / elapsed
    name = C.create_string_buffer(128)
    lib().rlnamd_device_name(name, 128)
    OUT.emit({
        "metric": "RLN Groth16 proofs/sec (BN254, h=20)", "value": round(value, 2), "unit": "proofs/s", "n_gpus": N,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32 limbs (256-bit Montgomery integers, BN254 Fr/Fq; 8x32 and 9x29 forms)",
        "data": "synthetic (SplitMix64 0xC0FFEE witnesses, shipped depth-20 arkzkey + graph)",
        "config": {"workload": "config 4: %d proofs as %d contiguous shards of %d, one process driving %d devices "
                               "through rlnamd_pool; H2D of inputs and D2H of proofs inside the timed region" % (n, N, SHARD, N),
                   "batch_per_gpu": SHARD, "chunk": B, "parallelism": "proof-sharded x%d (rlnamd_pool), no collective" % N,
                   "window_bits": int(pool.info.window_bits), "windows": int(pool.info.windows),
                   "table_gib": round(pool.info.table_bytes / 2**30, 2), "device": name.value.decode(),
                   "init_s": round(init_s, 2), "verified": bool(ok), "verified_sample": len(idx)},
        "achieved_GBps_whole_proof": round(value * BYTES_PER_PROOF / 1e9, 3),
        "replica_ms_last_step": [round(x, 2) for x in pool.last_ms()]})
    pool.close()
    if not ok:
        sys.exit(3)


class OneLineStdout:
    """The contract is ONE JSON line on stdout.  Native libraries print there too (RCCL's version banner at communicator
    creation), so file descriptor 1 points at stderr for the whole run and the line goes to the saved descriptor."""

    def __init__(self):
        sys.stdout.flush()
        self.fd = os.dup(1)
        os.dup2(2, 1)

    def emit(self, obj):
        sys.stdout.flush()
        os.write(self.fd, (json.dumps(obj) + "\n").encode())


OUT = None


def main():
    global OUT
    OUT = OneLineStdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("RLNAMD_BENCH_BATCH", "1024")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true", help="= --side none")
    ap.add_argument("--side", default="all", help="side objects of the default line, comma list of latency, finish, config3, "
                                                  "config5 (all | none)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="bound of the cpu_baseline's config-2 sample")
    ap.add_argument("--sustained-seconds", type=float, default=float(os.environ.get("RLNAMD_BENCH_SUSTAINED_S", "20")),
                    help="after the timed region: this many seconds of back-to-back batches (0: skip)")
    ap.add_argument("--workload", default="proofs", choices=["proofs", "merkle", "msm", "finish", "operating-points"],
                    help="proofs = BASELINE metric (default); merkle / msm / finish / operating-points = side measurements")
    args = ap.parse_args()
    if args.workload == "merkle":
        return merkle_main(args)
    if args.workload == "msm":
        return msm_main(args)
    if args.workload == "operating-points":
        return operating_points_main(args)

    side_all = ("latency", "finish", "config3", "config5")
    want = set() if (args.no_side_configs or args.side == "none") else \
        set(side_all) if args.side == "all" else {x.strip() for x in args.side.split(",") if x.strip()}
    if want - set(side_all):
        raise SystemExit("bench: --side takes %s" % ", ".join(side_all))
    under_torchrun = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if under_torchrun and world != args.gpus:
        raise SystemExit("bench: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not under_torchrun and args.gpus > 1:
        return pool_main(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks for boxes with fewer devices than ranks (tests/test_gpu_stream_pool.py::test_bench_under_torchrun_with_eight_ranks_on_one_device): every rank on one device, gloo
    # instead of RCCL (which refuses two ranks on one device); the driver never sets them
    backend = os.environ.get("RLNAMD_BENCH_BACKEND", "nccl")
    if "RLNAMD_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["RLNAMD_BENCH_DEVICE"])

    # A single-GPU run imports no torch at all: librln.so brings in the HIP runtime it was built for (/opt/rocm).  Under
    # torchrun torch is needed for the barrier and the max over ranks, and it goes FIRST: with librln.so (ROCm 7.2's
    # libamdhip64 / librccl) loaded before a torch wheel built for ROCm 7.0 the process aborted at exit ("double free or
    # corruption", tests/test_gpu_stream_pool.py::test_bench_under_torchrun_with_eight_ranks_on_one_device).  The streamed path no longer depends on which runtime it gets: the inputs
    # are staged by a kernel, not by the copy path that was slow under the wheel's runtime (profiles/r3 section 1).
    use_dist = under_torchrun                              # under torchrun the process group is always created
    torch = dist = None
    if use_dist:
        import torch  # plumbing only: barrier, max-reduce
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from zerokit_amd import lib, workload
    from zerokit_amd._native import check
    from zerokit_amd.batch import BatchProver
    check(lib().rlnamd_set_device(local_rank))
    name = C.create_string_buffer(128)
    lib().rlnamd_device_name(name, 128)

    B = args.batch
    finish = args.workload == "finish"
    t0 = time.time()
    prover = open_prover(B, lambda wb: BatchProver(max_batch=B, window_bits=wb))
    init_s = time.time() - t0
    init_ms = prover.init_ms()        # where the constructor's time went: parse / hipMalloc of the tables / table build / rest
    nslots = prover.n_slots()

    # ---- the witness stream of this rank, packed on the host before the timed region
    if world == 1:
        per_step = 1                                                   # batches per step
        nbatches = min(max(args.steps, 1), STREAM_WRAP)
        first = 0
    else:
        per_step = SHARD // B                                          # config 4: the rank's 8 192-proof shard
        nbatches = per_step
        first = SHARD * rank
    batches = [workload.config2_packed(prover.slots, prover.inputs_size, first + B * k, B) for k in range(nbatches)]
    partials = None
    if finish:
        # side measurement: partial proofs of the members computed once, then only
        # finish_zk_proof_with_rs per message (protocol/proof.rs:783-849)
        partials = []
        for k in range(nbatches):
            ws, _ = workload.config2_range(first + B * k, B)
            partials.append(prover.prove_partial([{q: w[q] for q in ("identity_secret", "user_message_limit",
                                                                       "path_elements", "identity_path_index")} for w in ws]))

    def sync():
        prover.sync()                     # every stream of the prover drained (hipStreamSynchronize on each)
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()

    results = {}                      # batch index -> (proofs, values, errors) of its LAST pass
    inflight = collections.deque()

    def pump(total_batches):
        for j in range(total_batches):
            k = j % nbatches
            if len(inflight) == nslots:
                t, kk = inflight.popleft()
                results[kk] = prover.collect_raw(t, B)
            t, _ = prover.submit(batches[k][0], batches[k][1], 2 if finish else 0, partials[k] if finish else None)
            inflight.append((t, k))
        while inflight:
            t, kk = inflight.popleft()
            results[kk] = prover.collect_raw(t, B)

    pump(args.warmup * per_step)
    prover.walk_clock_mhz()           # reset the clock tap: what follows is the timed region's clock
    results.clear()
    sync()
    t0 = time.perf_counter()
    c0, m0 = time.process_time(), time.thread_time()
    pump(args.steps * per_step)       # H2D of every batch's inputs ... D2H of its proofs, all inside
    sync()
    elapsed = time.perf_counter() - t0
    cpu_timed, cpu_main = time.process_time() - c0, time.thread_time() - m0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness outside the timed region: no error flags; first / last / middle proof of EVERY distinct batch
    #      of the timed region verifies on the host (rlnamd_verify_many)
    vp, vv = [], []
    ok = len(results) == min(nbatches, args.steps * per_step) or args.steps == 0
    for k, (proofs, values, errs) in sorted(results.items()):
        ok = ok and not any(errs)
        for i in (0, B // 2, B - 1):
            vp.append(proofs[128 * i:128 * i + 128])
            vv.append([int.from_bytes(values[160 * i + 32 * q:160 * i + 32 * q + 32], "little") for q in range(5)])
    ok = bool(ok and (all(prover.verify_many(vp, vv)) if vp else True))
    # distinct inputs give distinct proofs: the stream really was K different batches
    distinct = len({r[0][:128] for r in results.values()}) == len(results)

    # ---- what the HOST pays to feed one GPU (eight of them hang off one host): process CPU seconds over the
    #      timed region, and the wall time of the two calls by themselves -- submit with a free slot (copy into pinned
    #      memory + ~40 kernel launches) and collect of a finished batch (copy-out + the wipe's launches)
    host_feed = None
    if not finish and world == 1 and args.steps > 0:
        t_sub, t_col = [], []
        sync()
        tk = []
        for k in range(nslots):
            t0 = time.perf_counter()
            tk.append(prover.submit(batches[k % nbatches][0], batches[k % nbatches][1])[0])
            t_sub.append(time.perf_counter() - t0)
        prover.sync()
        for t in tk:
            t0 = time.perf_counter()
            prover.collect_raw(t, B)
            t_col.append(time.perf_counter() - t0)
        sub_ms, col_ms = sorted(t_sub)[len(t_sub) // 2] * 1e3, sorted(t_col)[len(t_col) // 2] * 1e3
        host_feed = {"cpu_s_over_timed_region": round(cpu_timed, 3), "wall_s_timed_region": round(elapsed, 3),
                     "host_cores_busy": round(cpu_timed / elapsed, 3),
                     "calling_thread_cores_busy": round(cpu_main / elapsed, 3),
                     "submit_ms_per_batch": round(sub_ms, 3), "collect_ms_per_batch": round(col_ms, 3),
                     "host_us_per_proof": round((sub_ms + col_ms) * 1e3 / B, 3),
                     "host_cores_for_8_gpus_at_this_rate": round(8 * (sub_ms + col_ms) / (elapsed / max(args.steps, 1) * 1e3), 3),
                     "note": "submit = copy of the batch's inputs into the slot's pinned buffer + the launches of the whole "
                             "pipeline; collect (batch already finished) = copy-out of proofs / values + the wipe's launches; "
                             "host_cores_busy = process CPU time / wall over the timed region, all threads (the HIP / ROCr "
                             "runtime's own threads included); calling_thread_cores_busy = the thread that calls submit / "
                             "collect (it polls and sleeps while a big batch runs: Prover::collect)"}
    if finish and 0 in results:   # the finished proofs of batch 0 against a FULL proof of the same (witness, r, s)
        t, _ = prover.submit(batches[0][0], batches[0][1])
        full0 = prover.collect_raw(t, B)
        ok = bool(ok and full0[0] == results[0][0] and not any(full0[2]))
    clock_mhz = prover.walk_clock_mhz()   # mean shader clock under the two walks over the timed region
    stage_ms = prover.stage_ms()          # HIP-event spans, mean over the last five launches (overlapped with their neighbours)
    # the dominant kernel by itself: single batches with nothing else in flight (outside the timed region)
    alone = []
    stage_ms_alone = {}
    ws0, rs0 = workload.config2_range(first, B)
    if not finish:
        prover.upload(batches[0][0], rs0)
        for _ in range(3):
            prover.run(B)
            alone.append(prover.stage_ms())
        alone.sort(key=lambda d: d.get("msm_g1", 0.0))
        stage_ms_alone = alone[1]
    g1_alone_ms = stage_ms_alone.get("msm_g1", 0.0)
    clock_alone_mhz = prover.walk_clock_mhz()
    info = prover.info
    g1_adds = int(info.g1_rows) * int(info.windows) * B
    # the other end of the same prover: ONE proof per call (what a caller of the reference does, rln/README.md:324-332),
    # submit -> collect with nothing else in flight; outside the timed region, reported beside the headline
    latency = None
    if not finish and "latency" in want and world == 1:
        # a different member on every call: the chain part of the hints is hashed (22 host hashes); then the bench's first
        # member seven times: its chain is remembered after the first of them (two host hashes, Impl::rln_hints)
        ts = []
        for i in range(9):
            one = workload.config2_packed(prover.slots, prover.inputs_size, first + 1 + i, 1)
            t0 = time.perf_counter()
            t, _ = prover.submit(*one)
            prover.collect_raw(t, 1)
            if i >= 2:
                ts.append((time.perf_counter() - t0) * 1e3)
        one = workload.config2_packed(prover.slots, prover.inputs_size, first, 1)
        ta = []
        for i in range(7):
            t0 = time.perf_counter()
            t, _ = prover.submit(*one)
            pr, va, er = prover.collect_raw(t, 1)
            if i >= 2:
                ta.append((time.perf_counter() - t0) * 1e3)
        same = bool(pr[:128] == results[0][0][:128] and not any(er)) if 0 in results else None
        few = {}
        for k in (2, 4, 8):   # a few proofs per call (the segments-behind-hints form, RLNAMD_HINTS = 24 proofs at most)
            if k > B:
                continue
            tk = []
            for i in range(7):   # other members on every call but the last, which is compared with the batch's bytes
                kin, krs = workload.config2_packed(prover.slots, prover.inputs_size, first + (16 + 8 * i if i < 6 else 0), k)
                t0 = time.perf_counter()
                t, _ = prover.submit(kin, krs)
                kp, _, ke = prover.collect_raw(t, k)
                if i >= 2:
                    tk.append((time.perf_counter() - t0) * 1e3)
            few[str(k)] = round(sorted(tk)[len(tk) // 2], 3)
            same = bool(same and kp[:128 * k] == results[0][0][:128 * k] and not any(ke)) if 0 in results else same
        latency = {"ms_min": round(min(ts), 3), "ms_median": round(sorted(ts)[len(ts) // 2], 3), "calls": len(ts),
                   "ms_median_same_member_again": round(sorted(ta)[len(ta) // 2], 3),
                   "ms_median_by_proofs_per_call": few, "witness_graph_as_segments": prover.hint_stats(),
                   "what": "one proof per call on the bench's prover (228 GiB tables): submit + collect, H2D and D2H included",
                   "same_bytes_as_in_the_batch": same}
    # ---- the steady state: `value` above times K steps -- under a second at the driver's K = 20, i.e. inside
    #      the boost window of the power management.  Here the same stream runs for --sustained-seconds with every slot
    #      in flight; proofs/s, the clock the walks held and the ratio to `value` go on the line.
    sustained = None
    if not finish and world == 1 and args.sustained_seconds > 0 and args.steps > 0:
        timed_first = {k: r[0][:256] for k, r in results.items()}
        prover.walk_clock_mhz()
        sync()
        t0 = time.perf_counter()
        j, errs_seen = 0, 0
        windows, w_t, w_j = [], t0, 0      # the rate over consecutive quarters of the leg: does it still fall at the end?
        while True:
            if len(inflight) == nslots:
                t, kk = inflight.popleft()
                results[kk] = prover.collect_raw(t, B)
                errs_seen += int(any(results[kk][2]))
            now = time.perf_counter()
            if now - w_t >= args.sustained_seconds / 4 and j > w_j:
                windows.append(round((j - w_j) * B / (now - w_t), 1))
                w_t, w_j = now, j
            if now - t0 >= args.sustained_seconds:
                break
            inflight.append((prover.submit(batches[j % nbatches][0], batches[j % nbatches][1])[0], j % nbatches))
            j += 1
        while inflight:
            t, kk = inflight.popleft()
            results[kk] = prover.collect_raw(t, B)
            errs_seen += int(any(results[kk][2]))
        sync()
        s_elapsed = time.perf_counter() - t0
        s_clock = prover.walk_clock_mhz()
        s_rate = j * B / s_elapsed
        sustained = {"proofs_per_s": round(s_rate, 1), "seconds": round(s_elapsed, 2), "batches": j,
                     "ms_per_batch": round(s_elapsed / max(j, 1) * 1e3, 3),
                     "ratio_to_value": round(s_rate / (world * B * per_step * args.steps / elapsed), 4),
                     "proofs_per_s_by_quarter": windows,
                     "shader_clock_mhz": {k: round(v, 1) for k, v in s_clock.items()},
                     "shader_clock_mhz_timed_region": {k: round(v, 1) for k, v in clock_mhz.items()},
                     "same_bytes_as_the_timed_region": bool(errs_seen == 0 and all(results[k][0][:256] == v for k, v in timed_first.items())),
                     "what": "the timed region's stream of distinct batches continued for --sustained-seconds with every "
                             "workspace slot in flight (H2D and D2H inside): the rate at the clock the power management "
                             "holds once its boost window is over.  `value` is the contract's K-step figure"}
        ok = bool(ok and sustained["same_bytes_as_the_timed_region"])
    # ---- SURVEY 8(f) rank 2 on the line: finish-from-partial, rate and latency, beside the full-proof figures
    finish_obj, finish_part0 = None, None
    if not finish and "finish" in want and world == 1 and args.steps > 0:
        try:
            finish_obj, parts = measure_finish(prover, batches, results, B, world * B * per_step * args.steps / elapsed,
                                               latency["ms_median"] if latency else None)
            finish_part0 = parts[0][0]
            ok = bool(ok and finish_obj["correct"])
        except Exception as e:  # noqa: BLE001   (a side leg that could not RUN does not fail the headline; a wrong byte does)
            finish_obj = {"error": str(e)}
    prover.close()

    # ---- the other single-GPU BASELINE configs, with the prover's HBM released
    side = {}
    hung = False
    if want & {"config3", "config5"} and not finish:
        # N > 1: the config-5 MSM is the one place with a collective (ncclAllGather through the C ABI).  It runs on a
        # helper thread with a deadline: a rank that fails before its collective would leave the others blocked in theirs,
        # and a blocked RCCL call cannot be cancelled -- the headline line must come out regardless.
        import threading

        def side_work():
            comm = None
            try:
                check(lib().rlnamd_set_device(local_rank))   # the current device is per-thread state (HIP and torch)
                if use_dist and backend == "nccl":
                    torch.cuda.set_device(local_rank)
                if world == 1 and "config3" in want:
                    side["config3"] = measure_config3()
                if "config5" in want:
                    comm = make_comm(rank, world, dist)
                    side["config5"] = measure_config5(comm, rank, world)
            except Exception as e:  # noqa: BLE001
                side["side_config_error"] = str(e)
            finally:
                if comm is not None:
                    comm.close()
        th = threading.Thread(target=side_work, daemon=True)
        th.start()
        th.join(timeout=300 if world == 1 else 180)
        if th.is_alive():
            hung = True
            side = dict(side, side_config_error="side configs did not finish within the deadline (abandoned)")

    if rank == 0:
        steps = max(args.steps, 1)
        proofs_total = world * B * per_step * args.steps
        value = proofs_total / elapsed
        msm_ms = stage_ms.get("msm_g1", 0.0)
        hbm_achieved = (MSM_G1_BYTES_PER_PROOF * B) / (msm_ms * 1e-3) / 1e9 if msm_ms > 0 else 0.0
        pm = load_pmc(info, B)
        mix = load_mix()
        ms_per_step = elapsed / steps * 1e3 / per_step       # per 1024-proof batch
        kernels, whole = issue_view(pm, mix, info, B, stage_ms, stage_ms_alone, clock_mhz, clock_alone_mhz, ms_per_step)
        traffic = round(pm["kernels"]["k_msm29<G1>"]["traffic_bytes_per_launch"] / 1e9, 3) \
            if pm and "traffic_bytes_per_launch" in pm["kernels"]["k_msm29<G1>"] else None
        # The contract's object (top level) is for k_msm29<G1>: SURVEY 8(d) / BASELINE.json define the fraction on the HBM
        # side -- ALGORITHMIC bytes per launch (A, B1, H, L operands: a 64-byte point + a 32-byte scalar each) / launch
        # duration / 8 TB/s.  Neither walk is HBM-bound; `kernels` and `whole_step` price them against VALU issue cycles.
        alg_bytes = MSM_G1_BYTES_PER_PROOF * B
        roof = {"bound": "hbm",
                "limiter": "VALU issue cycles at a power-limited clock (see kernels / whole_step: issue_frac)",
                "kernel": "k_msm29<G1> (G1 fixed-base table MSM, 9 x 29-bit limbs): 61 % of a batch's VALU instructions; "
                          "k_msm29<G2>, the longest launch of a step, is priced beside it under `kernels`",
                "algorithmic_bytes_per_proof": MSM_G1_BYTES_PER_PROOF, "algorithmic_bytes_per_launch": alg_bytes,
                "launch_ms": round(msm_ms, 3), "launch_ms_alone": round(g1_alone_ms, 3),
                "achieved": round(hbm_achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(hbm_achieved / HBM_PEAK_GBPS, 6), "hbm_frac": round(hbm_achieved / HBM_PEAK_GBPS, 6),
                "traffic": traffic,
                "traffic_unit": "GB per launch (2*FETCH_SIZE + WRITE_SIZE, PMC pass of the same build: %s)"
                                % (pm["_file"] if pm else "none"),
                "peaks": {"hbm_GBps": HBM_PEAK_GBPS,
                          "mad_slot_peak_Ginst_per_s": round(QUARTER_RATE_PEAK_GINST, 1),
                          "valu_issue_peak_Ginst_per_s": round(PLAIN_RATE_PEAK_GINST, 1),
                          "issue_cycles_per_s": SIMDS * CLOCK_NOMINAL_HZ,
                          "note": "1024 SIMDs x 2.4 GHz; a wave-instruction of the quarter-rate class (v_mad_u64_u32, "
                                  "v_mul_lo_u32, 64-bit adds / shifts, carries, FP64) occupies its SIMD for 4 cycles = the mad "
                                  "slot, 614.4 G/s; a plain 32-bit op for 2 = the guide's VALU issue rate, 1 228.8 G/s "
                                  "(profiles/r5_microbench_dfma.txt)"}}
        if kernels:
            roof["kernels"] = kernels
            roof["whole_step"] = whole
            roof["issue_frac"] = kernels["k_msm29<G1>"].get("issue_frac")
            roof["issue_frac_alone"] = kernels["k_msm29<G1>"].get("issue_frac_alone")
            roof["sources"] = {"pmc": pm["_file"], "isa_mix": "profiles/r5_walk_isa_mix.json",
                               "walk_source_hash": walk_source_hash()}
        else:
            roof["kernels"] = roof["whole_step"] = None
            roof["issue_view_omitted"] = ("no PMC pass / ISA mix under profiles/ matches this schedule and this build of the "
                                          "walk (walk_source_hash %s)" % walk_source_hash())
        roof.update({
            # what the table-walk algorithm itself reads: one 64-byte entry per mixed addition (HBM capacity and traffic
            # traded for Pippenger's bucket reduction): not the algorithmic bytes of 8(d), reported beside them
            "table_walk_gb_per_launch": round(g1_adds * 64 / 1e9, 3),
            "table_walk_GBps_alone": round(g1_adds * 64 / (g1_alone_ms * 1e-3) / 1e9, 1) if g1_alone_ms > 0 else None,
            "madd_per_s": round(g1_adds / (msm_ms * 1e-3) / 1e9, 2) if msm_ms > 0 else None,
            "madd_per_s_alone": round(g1_adds / (g1_alone_ms * 1e-3) / 1e9, 2) if g1_alone_ms > 0 else None,
            "note": "launch_ms = mean HIP-event span of the last five launches of the timed region on the kernel's own "
                    "stream (it shares the SIMDs with the G2 walk and the front / back ends of the neighbouring batches; "
                    "launch_ms_alone: one batch with nothing else in flight, its two walks still side by side).  frac = "
                    "algorithmic_bytes_per_launch / launch_ms / peak.  See DESIGN.md 4 and 6"})
        line = {
            "metric": "RLN Groth16 proofs/sec (BN254, h=20)" + (" -- finish_rln_proof from cached partial proofs "
                                                                 "(side measurement)" if finish else ""),
            "value": round(value, 2),
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (256-bit Montgomery integers, BN254 Fr/Fq; 8x32 and 9x29 forms)",
            "data": "synthetic (SplitMix64 0xC0FFEE witnesses, shipped depth-20 arkzkey + graph)",
            "config": {"workload": ("config 2: %d DISTINCT batches of %d independent RLN proofs, tree_height=20; H2D of the "
                                    "witness inputs and D2H of proofs + values inside the timed region, %d workspace slots "
                                    "in flight" % (nbatches, B, nslots)) if world == 1 else
                                   ("config 4: %d proofs as %d contiguous shards of %d (one process per GPU), %d chunks of "
                                    "%d per step uploaded again every step; H2D and D2H inside the timed region"
                                    % (SHARD * world, world, SHARD, per_step, B)),
                       "batch_per_gpu": B * per_step, "chunk": B,
                       "parallelism": "proof-sharded x%d, no data-path collective" % world,
                       "glv": bool(info.glv),
                       "window_bits": int(info.window_bits), "windows": int(info.windows),
                       "window_bits_g2": int(info.window_bits_g2), "windows_g2": int(info.windows_g2),
                       "table_gib": round(info.table_bytes / 2**30, 2),
                       "device": name.value.decode(), "init_s": round(init_s, 2), "init_ms": init_ms,
                       "verified": ok, "verified_proofs": len(vp), "distinct_batches_gave_distinct_proofs": bool(distinct)},
            "achieved_GBps_whole_proof": round(value * BYTES_PER_PROOF / 1e9, 3),
            "stage_ms": {"overlapped": {k: round(v, 3) for k, v in stage_ms.items()},
                         "alone": {k: round(v, 3) for k, v in stage_ms_alone.items()},
                         "note": "overlapped = HIP-event spans inside the pipelined timed region: every span includes "
                                 "whatever shared the chip with it (five batches in flight), so they do not add up to "
                                 "ms_per_step; alone = the same stages of one batch with nothing else in flight"},
            # the walks are VALU-issue bound: additions/s = SIMDs x 64 x clock / (4 x instructions per addition), so
            # the clock the power management holds is part of the result (2.4 GHz nominal)
            "shader_clock_mhz": {"timed_region": {k: round(v, 1) for k, v in clock_mhz.items()},
                                 "walks_alone": {k: round(v, 1) for k, v in clock_alone_mhz.items()}},
            "roofline": roof,
        }
        line.update(side)
        if latency is not None:
            line["single_proof_latency"] = latency
        if sustained is not None:
            line["sustained"] = sustained
        if finish_obj is not None:
            line["finish"] = finish_obj
        if host_feed is not None:
            line["host_feed"] = host_feed
        # how many ranks RCCL really had: the communicator created through the C ABI for config 5 when it ran, else the
        # torch process group the barrier and the max-reduce went through.  A line that says n_gpus = N is only printed
        # when that number is N; the gloo test hook for boxes with fewer devices than ranks says so.
        if "config5" in side:
            line["rccl_ranks"] = side["config5"]["rccl_ranks"]
        elif use_dist and backend == "nccl":
            line["rccl_ranks"] = dist.get_world_size()
            line["rccl_ranks_source"] = "torch.distributed process group (backend nccl = RCCL)"
        elif use_dist:
            line["rccl_ranks"] = None
            line["rccl_ranks_source"] = "test hook RLNAMD_BENCH_BACKEND=%s: no RCCL in this run" % backend
        if use_dist and backend == "nccl" and line.get("rccl_ranks") != world:
            print("bench: --gpus %d but RCCL had %s rank(s): refusing to print a line" % (world, line.get("rccl_ranks")),
                  file=sys.stderr)
            sys.exit(4)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ws0, rs0, args.cpu_seconds, want)
            if finish_obj is not None and "correct" in finish_obj and line["cpu_baseline"] and "cores" in line["cpu_baseline"]:
                try:   # the reference's own comparison (full vs finish, one call on a CPU) restated beside the GPU's
                    fb, part0, fin0 = cpu_finish_baseline(ws0, rs0, line["cpu_baseline"]["cores"])
                    # the oracle judges the product's partial points and the product's finished proof of witness 0
                    fb["gpu_partial_points_equal_oracle"] = bool(finish_part0 == part0)
                    fb["gpu_finish_equals_oracle_finish"] = bool(0 in results and results[0][0][:128] == fin0)
                    finish_obj["cpu_baseline"] = fb
                    finish_obj["correct"] = bool(finish_obj["correct"] and fb["gpu_partial_points_equal_oracle"]
                                                 and fb["gpu_finish_equals_oracle_finish"] and fb["finish_equals_full"])
                    ok = bool(ok and finish_obj["correct"])
                    line["config"]["verified"] = ok
                except Exception as e:  # noqa: BLE001
                    finish_obj["cpu_baseline"] = {"error": str(e)}
        OUT.emit(line)
    if hung:            # a thread is stuck inside a collective: leave without the runtime's teardown
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0 if ok else 3)
    if use_dist:
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
