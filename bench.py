#!/usr/bin/env python3
"""bench.py -- RLN Groth16 proofs/s on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the proving hot path over one batch of synthetic witnesses: per GPU, BATCH
(default 1024 = BASELINE config 2) independent depth-20 RLN proofs from the seeded config-2 generator
(SplitMix64 0xC0FFEE; oracle/pyref/workload.py restates the generator, this file re-implements it so the
product path never imports the oracle).  Inputs are uploaded to HBM before the timed region; the timed
region is K calls of rlnamd_prover_run (witness -> QAP/NTT -> MSM -> finalize -> proof bytes in HBM).
N > 1: one process per GPU (torchrun), every rank proves its own shard, no data-path collective (weak
scaling); timing = barrier + sync on both sides, max over ranks.

Extra objects on the same line:
  roofline     -- dominant kernel (G1 table MSM): algorithmic MSM operand bytes per launch / its mean launch
                  time (HIP events on the kernel's own stream, last five launches of the timed region: the walks
                  of neighbouring batches overlap on two streams, so this span includes sharing the SIMDs;
                  launch_ms_alone = the same launch with nothing else in flight, measured after the timed
                  region), against the 8 TB/s HBM peak.  Defaults: K = 50 steps, W = 2 warm-up steps, comb
                  tables on the 19-window schedule (207 GiB; smaller tables are tried if that does not fit).
  cpu_baseline -- the oracle's C restatement of the arkworks CPU path (oracle/c, kind "port") timed on a
                  bounded sample of the same witnesses on the host cores (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before torch / HIP initialise: see zerokit_amd/csrc/common.cpp

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MASK = (1 << 64) - 1
# per-proof algorithmic bytes
BYTES_PER_PROOF = 7468404
MSM_G1_BYTES_PER_PROOF = (5844 + 5844 + 8192 + 5838) * 96   # A, B1, H, L operands (point 64 B + scalar 32 B)
HBM_PEAK_GBPS = 8000.0


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def fr(self):
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v % R


def config2_witnesses(n, seed=0xC0FFEE, depth=20):
    g = SplitMix64(seed)
    ws, rs = [], []
    for i in range(n):
        ws.append(dict(identity_secret=g.fr(), user_message_limit=100, message_id=i % 100,
                       path_elements=[g.fr() for _ in range(depth)],
                       identity_path_index=[g.next() & 1 for _ in range(depth)], x=g.fr(),
                       external_nullifier=g.fr()))
        rs.append((g.fr(), g.fr()))
    return ws, rs


def cpu_baseline(ws, rs, target_seconds=12.0):
    """Times oracle/c (the C restatement of the arkworks CPU path) on a bounded sample; returns the dict for
    the JSON line or None when the oracle library has not been built."""
    try:
        from oracle.c import binding as ob
    except Exception:
        return None
    try:
        return ob.time_baseline(ws, rs, target_seconds)
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def valu_view(prover, B, alone_ms, clock_mhz):
    """VALU side of the roofline for the G1 walk: instructions per launch from the committed PMC pass (they scale with
    the number of additions), rate = instructions / duration of a launch with nothing else in flight"""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r2_pmc_walks.json")))["kernels"]["k_msm29<G1>"]
        per_wave_add = pm["SQ_INSTS_VALU"] / (pm["lane_additions_per_launch"] / 64)
    except Exception:  # noqa: BLE001
        return None
    adds = int(prover.info.g1_rows) * int(prover.info.windows) * B
    insts = per_wave_add * adds / 64
    if alone_ms <= 0:
        return None
    rate = insts / (alone_ms * 1e-3) / 1e9
    peak_nominal = 1024 * 2.4e9 / 4 / 1e9
    out = {"valu_insts_per_wave_addition": round(per_wave_add, 1), "valu_insts_per_launch_G": round(insts / 1e9, 3),
           "achieved_Ginst_per_s": round(rate, 1), "peak_Ginst_per_s_at_2400MHz": round(peak_nominal, 1),
           "frac_of_nominal_clock_peak": round(rate / peak_nominal, 4)}
    if clock_mhz > 0:
        out["clock_mhz"] = round(clock_mhz, 1)
        out["frac_of_peak_at_measured_clock"] = round(rate / (1024 * clock_mhz * 1e6 / 4 / 1e9), 4)
    return out


def prover_alone_ms(prover):
    """msm_g1 span of the batch that just ran alone (run() = one batch, pipeline drained)"""
    return prover.stage_ms().get("msm_g1", 0.0)


def merkle_main(args):
    """BASELINE config 3: build a 2^20-leaf Poseidon tree (leaves i+1 generated in HBM) and emit all 2^20
    membership paths into HBM; algorithmic bytes 792 723 424.  Side measurement, one JSON line."""
    from zerokit_amd.batch import PoseidonTree
    depth, n = 20, 1 << 20
    t = PoseidonTree(depth)
    t.bench(n, 1, verify=False)   # warm-up
    build, paths = [], []
    for _ in range(max(args.steps, 1)):
        r = t.bench(n, 1, verify=False)
        build.append(r["build_ms"])
        paths.append(r["proofs_ms"])
    bad = t.bench(n, 1, verify=True)["bad"]
    b, p = sum(build) / len(build), sum(paths) / len(paths)
    path_bytes = n * (depth * 32 + depth)
    print(json.dumps({
        "metric": "Poseidon Merkle: 2^20-leaf build + 2^20 membership paths (config 3)", "unit": "ms",
        "build_ms": round(b, 3), "paths_ms": round(p, 3), "hashes_per_s": round((n - 1) / (b * 1e-3), 1),
        "paths_failed_device_verification": bad,
        "achieved_GBps_config3": round(792723424 / ((b + p) * 1e-3) / 1e9, 2),
        "roofline_paths": {"bound": "hbm", "achieved": round(path_bytes / (p * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS,
                           "unit": "GB/s", "frac": round(path_bytes / (p * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}}))


def msm_main(args):
    """BASELINE config 5: one 2^24-point G1 MSM split by point index over the ranks (1 rank = whole MSM on one
    GPU), one RCCL all_gather of the per-window sums, local fold.  Side measurement, one JSON line on rank 0."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    from zerokit_amd import lib
    from zerokit_amd._native import check
    from zerokit_amd.batch import MsmG1
    from zerokit_amd.distributed import all_gather_bytes, shard_bounds
    check(lib().rlnamd_set_device(local_rank))
    n_total = 1 << int(os.environ.get("RLNAMD_MSM_LOG2", "24"))
    lo, hi = shard_bounds(n_total, world)[rank]
    m = MsmG1(hi - lo)
    m.generate(0xC0FFEE, lo, hi - lo)          # bases + scalars resident in HBM, not timed
    times, stage = [], {}
    for it in range(args.warmup + max(args.steps, 1)):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        blob, stage = m.run_windows()
        blobs = all_gather_bytes(blob, device="cuda")
        res = m.combine(blobs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if it >= args.warmup:
            times.append(dt)
    if rank == 0:
        ok = res == MsmG1.expected(0xC0FFEE, 0, n_total)
        ms = sum(times) / len(times) * 1e3
        gbps = n_total * 96 / (ms * 1e-3) / 1e9
        print(json.dumps({"metric": "single 2^%d-point BN254 G1 MSM (config 5)" % (n_total.bit_length() - 1),
                          "ms": round(ms, 3), "n_gpus": world, "correct": bool(ok), "stage_ms_rank0": stage,
                          "roofline": {"bound": "hbm", "achieved": round(gbps, 2), "peak": HBM_PEAK_GBPS * world,
                                       "unit": "GB/s", "frac": round(gbps / (HBM_PEAK_GBPS * world), 5)}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("RLNAMD_BENCH_BATCH", "1024")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="proofs", choices=["proofs", "merkle", "msm", "finish"],
                    help="proofs = BASELINE metric (default); merkle = config 3 side measurement (not the bench line)")
    args = ap.parse_args()
    if args.workload == "merkle":
        return merkle_main(args)
    if args.workload == "msm":
        return msm_main(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch  # plumbing only: device selection, barrier, max-reduce
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ          # under torchrun the RCCL group is always created
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from zerokit_amd import lib
    from zerokit_amd._native import check
    from zerokit_amd.batch import BatchProver
    check(lib().rlnamd_set_device(local_rank))
    name = C.create_string_buffer(128)
    lib().rlnamd_device_name(name, 128)

    B = args.batch
    t0 = time.time()
    # GLV comb schedule g1 + 10000 * g2 over the 127-bit scalar halves: 114 = 15 + 8 x 14 bits (9 windows, 18 additions
    # per G1 point), 715 = 7 x 16 + 15 bits (8 windows, 16 additions per G2 point): 228 GiB of fixed-base tables, sized for
    # 288 GB of HBM.  Smaller tables are tried if that does not fit (RLNAMD_GLV=0 + 813: the round-1 19-window walk).
    wbits = int(os.environ.get("RLNAMD_WINDOW_BITS", "7150114"))
    prover = None
    for wb in dict.fromkeys([wbits, 114, 13, 12, 10]):   # a box with less free HBM still runs: smaller tables
        try:
            prover = BatchProver(max_batch=B, window_bits=wb)
            break
        except Exception as e:  # noqa: BLE001
            print("bench: window schedule %d not available (%s)" % (wb, e), file=sys.stderr)
    if prover is None:
        raise SystemExit("bench: no table size fits this device")
    init_s = time.time() - t0
    ws, rs = config2_witnesses(B, seed=0xC0FFEE + rank)   # every rank proves a different shard
    inputs = prover.pack_inputs(ws)
    n = prover.upload(inputs, rs)                          # resident in HBM before the timed region

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    finish = args.workload == "finish"
    if finish:
        # side measurement: partial proofs of the members computed once, then only
        # finish_zk_proof_with_rs per message (protocol/proof.rs:783-849)
        partials = prover.prove_partial([{k: w[k] for k in ("identity_secret", "user_message_limit", "path_elements",
                                                             "identity_path_index")} for w in ws])
        n = prover.upload(inputs, rs)
        prover.upload_partial(partials)
    for _ in range(args.warmup):
        prover.run_async_mode(n, 2) if finish else prover.run(n)
    prover.sync()
    prover.walk_clock_mhz()   # reset the clock tap: what follows is the timed region's clock
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if finish:
            prover.run_async_mode(n, 2)
        else:
            prover.run_async(n)  # batches pipeline on the device; each ends with proofs in pinned host memory
    prover.sync()
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness spot check outside the timed region: first and last proof of the batch verify
    out = prover.download(n)
    ok = all(o["error"] == 0 for o in out) and prover.verify(out[0]["proof"], out[0]["public_inputs"]) and \
        prover.verify(out[-1]["proof"], out[-1]["public_inputs"])

    clock_mhz = prover.walk_clock_mhz()   # mean shader clock under the two walks over the timed region
    stage_ms = prover.stage_ms()   # HIP-event spans, mean over the last five launches (overlapped with their neighbours)
    # the dominant kernel by itself: single batches with nothing else in flight (outside the timed region)
    alone = []
    for _ in range(3):
        prover.run(n)
        alone.append(prover_alone_ms(prover))
    g1_alone_ms = sorted(alone)[1] if not finish else 0.0
    clock_alone_mhz = prover.walk_clock_mhz()
    if rank == 0:
        steps = max(args.steps, 1)
        proofs = world * B * args.steps
        value = proofs / elapsed
        msm_ms = stage_ms.get("msm_g1", 0.0)
        achieved = (MSM_G1_BYTES_PER_PROOF * B) / (msm_ms * 1e-3) / 1e9 if msm_ms > 0 else 0.0
        traffic = None   # HBM bytes per launch of the dominant kernel, from the committed PMC passes
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r2_pmc_k_msm_g1.json")))
            if B == 1024 and int(prover.info.window_bits) == pm.get("window_bits") and \
                    int(prover.info.windows) == pm.get("windows"):
                traffic = round(pm["traffic_bytes_per_launch"] / 1e9, 3)
        except Exception:  # noqa: BLE001
            pass
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
            "config": {"workload": "config 2: batch of %d independent RLN proofs per GPU, tree_height=20, "
                                   "inputs resident in HBM" % B,
                       "batch_per_gpu": B, "parallelism": "proof-sharded x%d, no collective" % world,
                       "glv": bool(prover.info.glv),
                       "window_bits": int(prover.info.window_bits), "windows": int(prover.info.windows),
                       "window_bits_g2": int(prover.info.window_bits_g2), "windows_g2": int(prover.info.windows_g2),
                       "table_gib": round(prover.info.table_bytes / 2**30, 2),
                       "device": name.value.decode(), "init_s": round(init_s, 2), "verified": bool(ok)},
            "achieved_GBps_whole_proof": round(value * BYTES_PER_PROOF / 1e9, 3),
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            # the walks are VALU-issue bound: additions/s = SIMDs x 64 x clock / (4 x instructions per addition), so
            # the clock the power management holds is part of the result (2.4 GHz nominal)
            "shader_clock_mhz": {"timed_region": {k: round(v, 1) for k, v in clock_mhz.items()},
                                 "walks_alone": {k: round(v, 1) for k, v in clock_alone_mhz.items()}},
            "roofline": {"bound": "hbm", "kernel": "k_msm29<G1> (G1 fixed-base table MSM, 9 x 29-bit limbs)",
                         "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic,
                         "traffic_unit": "GB per launch (2*FETCH_SIZE + WRITE_SIZE, profiles/r2_pmc_k_msm_g1.json)",
                         "launch_ms": round(msm_ms, 3), "launch_ms_alone": round(g1_alone_ms, 3),
                         # what the table-walk algorithm itself must read: one 64-byte entry per mixed addition
                         "table_walk_gb_per_launch": round(int(prover.info.g1_rows) * int(prover.info.windows) * B * 64 / 1e9, 3),
                         "table_walk_GBps_alone": round(int(prover.info.g1_rows) * int(prover.info.windows) * B * 64 / (g1_alone_ms * 1e-3) / 1e9, 1)
                         if g1_alone_ms > 0 else None,
                         "madd_per_s": round(int(prover.info.g1_rows) * int(prover.info.windows) * B / (msm_ms * 1e-3) / 1e9, 2) if msm_ms > 0 else None,
                         "madd_per_s_alone": round(int(prover.info.g1_rows) * int(prover.info.windows) * B / (g1_alone_ms * 1e-3) / 1e9, 2)
                         if g1_alone_ms > 0 else None,
                         # the bound that actually binds ("expect HBM-fraction << 1 and report VALU utilisation
                         # alongside"): VALU wave-instructions of one launch (PMC SQ_INSTS_VALU, profiles/r2_pmc_walks.json)
                         # over the launch by itself, against 1024 SIMDs x clock / 4 cycles per instruction
                         "valu": valu_view(prover, B, g1_alone_ms, clock_alone_mhz.get("g1_walk", 0.0)),
                         "note": "launch_ms = mean HIP-event span of the last five launches of the timed region, on the "
                                 "kernel's stream; the G1 and G2 walks of neighbouring batches run on two streams and share "
                                 "the SIMDs, so the span includes that sharing (launch_ms_alone: the same launch with nothing "
                                 "else in flight).  VALU-issue bound, not HBM bound: one mixed addition is ~2 020 VALU instructions "
                                 "(1 467 v_mad_u64_u32 of the field products) and the SIMDs issue one per 4 cycles for the whole "
                                 "launch (roofline.valu: fraction of the issue rate at the measured clock ~ 1); what is left is the "
                                 "clock, which the power management holds near 1.85 - 1.95 GHz under this instruction mix plus "
                                 "the table gathers (shader_clock_mhz; 2.4 GHz nominal).  madd_per_s / madd_per_s_alone are "
                                 "G additions/s; traffic: FETCH_SIZE doubled as the guide prescribes for 128-byte requests "
                                 "(calibrated on the G2 walk's 128-byte entries and on the NTT streams); the G1 walk's "
                                 "64-byte gathers may be 64-byte requests, in which case the traffic is half of it; "
                                 "see DESIGN.md section 4"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ws, rs)
        print(json.dumps(line))
    prover.close()
    if use_dist:
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
