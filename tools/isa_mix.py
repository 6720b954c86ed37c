#!/usr/bin/env python3
"""Instruction mix of one mixed addition in the two throughput walks, priced in VALU issue cycles.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=400000 -S --cuda-device-only \
          -I zerokit_amd/csrc zerokit_amd/csrc/prover_walks.hip -o /tmp/walks.s
    python tools/isa_mix.py /tmp/walks.s profiles/r5_pmc_walks.json profiles/r5_walk_isa_mix.json

The hot path of a walk kernel is found from the listing and the counters together: among the kernel's large basic
blocks, the subset whose VALU total comes closest (from below) to the measured wave-instructions per wave-addition
(SQ_INSTS_VALU / additions, PMC) -- the other large blocks are the out-of-line rare paths (same-x cases, the general
law).  Every opcode is priced with the issue rate MEASURED on this chip (tools/microbench_dfma.hip, profiles/
r5_microbench_dfma.txt): 4 cycles per wave-instruction for the quarter-rate class (v_mad_u64_u32, v_mul_lo/hi_u32,
64-bit adds and shifts, carries, three-operand integer ops, v_lshlrev_b32, FP64), 2 cycles for the plain 32-bit ops
(v_add_u32, v_sub_u32, v_and/or_b32, v_lshrrev_b32, v_mov_b32); an opcode that was not measured is priced at 4.
bench.py reads the result (gated by the walk's source hash) for the issue-slot view of its roofline object."""
import collections
import itertools
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import walk_source_hash  # noqa: E402

# measured at ~1 000 G wave-instr/s (2 cycles per wave-instruction per SIMD)
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32"}
# measured at 450 - 590 G wave-instr/s (4 cycles)
QUARTER_MEASURED = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_lshl_add_u64", "v_lshl_add_u32",
                    "v_add_co_u32", "v_addc_co_u32", "v_alignbit_b32", "v_lshlrev_b32", "v_or3_b32", "v_add3_u32",
                    "v_lshrrev_b64", "v_fma_f64", "v_add_f64"}
KERNELS = {"k_msm29<G1>": ("k_msm29INS_7G1Acc29", "Li4ELb0"), "k_msm29<G2>": ("k_msm29INS_6G2AccTINS_10Fq2LaneOps", "Li2ELb0")}


def base(op):
    return re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)


def blocks_of(lines):
    out, cur = [], collections.Counter()
    for l in lines:
        t = l.strip()
        op = t.split()[0] if t and not t.startswith((";", ".", "//")) and not t.endswith(":") else ""
        if re.match(r"^\.LBB\d+_\d+:", t) or op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
            if cur:
                out.append(cur)
            cur = collections.Counter()
        if op.startswith("v_"):
            cur[base(op)] += 1
    if cur:
        out.append(cur)
    return out


def main():
    asm, pmc_path, out_path = sys.argv[1:4]
    lines = open(asm).read().split("\n")
    pmc = json.load(open(pmc_path))
    doc = {"source": "tools/isa_mix.py over the gfx950 listing of prover_walks.hip and " + os.path.basename(pmc_path),
           "walk_source_hash": walk_source_hash(),
           "cycles_per_wave_instruction": {"quarter_rate_class": 4, "plain_32bit_class": 2, "not_measured": 4},
           "plain_32bit_class": sorted(FAST), "quarter_rate_class_measured": sorted(QUARTER_MEASURED), "kernels": {}}
    for tag, (sub, variant) in KERNELS.items():
        a = next(i for i, l in enumerate(lines) if re.match(r"^_ZN6rlnamd7" + re.escape(sub) + r".*" + variant + r".*:", l))
        b = next(i for i in range(a, len(lines)) if ".end_amdhsa_kernel" in lines[i])
        meta = dict(re.findall(r"\.amdhsa_(next_free_vgpr|accum_offset|private_segment_fixed_size)\s+(\d+)", "\n".join(lines[a:b + 1])))
        notes = {k: int(v) for l in lines[b:b + 80] for k, v in re.findall(r";\s*(NumVgprs|NumAgprs|ScratchSize|Occupancy|VGPRSpillCount|TotalNumVgprs)[^:]*:\s*(\d+)", l)}
        scratch_ops = sum(1 for l in lines[a:b] if l.strip().startswith("scratch_"))
        blks = blocks_of(lines[a:b])
        big = [c for c in blks if sum(c.values()) >= 150]
        target = pmc["kernels"][tag]["valu_per_wave_addition"]
        best = None
        for r in range(1, len(big) + 1):
            for comb in itertools.combinations(range(len(big)), r):
                tot = sum(sum(big[i].values()) for i in comb)
                if tot <= target and (best is None or tot > best[0]):
                    best = (tot, comb)
        hot = collections.Counter()
        for i in best[1]:
            hot.update(big[i])
        n = sum(hot.values())
        fast = sum(v for k, v in hot.items() if k in FAST)
        unmeasured = {k: v for k, v in hot.items() if k not in FAST and k not in QUARTER_MEASURED}
        quarter = n - fast
        # the few instructions per addition outside the large blocks (loop control, digit fetch, address arithmetic:
        # target - n of them) are priced as quarter-rate
        rest = max(0.0, target - n)
        cycles = 4 * (quarter + rest) + 2 * fast
        doc["kernels"][tag] = {
            "valu_per_wave_addition_pmc": target, "valu_in_hot_blocks_isa": n, "hot_blocks_valu": [sum(big[i].values()) for i in best[1]],
            "other_large_blocks_valu": [sum(c.values()) for j, c in enumerate(big) if j not in best[1]],
            "v_mad_u64_u32": hot["v_mad_u64_u32"], "mad_u64_u32_share": round(hot["v_mad_u64_u32"] / target, 4),
            "quarter_rate_insts": quarter, "plain_32bit_insts": fast,
            "quarter_rate_share": round((quarter + rest) / target, 4),
            "issue_cycles_per_wave_addition": round(cycles, 1),
            "mean_cycles_per_wave_instruction": round(cycles / target, 4),
            "by_opcode": dict(hot.most_common()), "priced_at_4_without_measurement": unmeasured,
            "registers": {"next_free_vgpr": int(meta.get("next_free_vgpr", 0)), "scratch_bytes_per_lane": int(meta.get("private_segment_fixed_size", 0)),
                          "scratch_instructions_in_kernel": scratch_ops, **notes}}
    json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)
    for k, v in doc["kernels"].items():
        print(k, {q: v[q] for q in ("valu_per_wave_addition_pmc", "valu_in_hot_blocks_isa", "hot_blocks_valu", "mad_u64_u32_share",
                                    "quarter_rate_share", "issue_cycles_per_wave_addition", "registers")})


if __name__ == "__main__":
    main()
