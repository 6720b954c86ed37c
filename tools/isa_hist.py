#!/usr/bin/env python3
"""Per-basic-block VALU instruction counts of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only).
Usage: isa_hist.py file.s [kernel-name-substring] [min-valu-per-block]
Prints, for every kernel, its register / scratch use and the blocks with at least `min` VALU instructions (line range,
VALU count, v_mad_u64_u32 count), then the histogram of the blocks that are not skipped rare paths when `--main a:b`
gives a line range."""
import collections
import re
import sys


def kernels(lines):
    cur, out = None, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
            out[cur] = [i, None]
        if cur and ".end_amdhsa_kernel" in l:
            out[cur][1] = i
            cur = None
    return out


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    mn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    for name, (a, b) in kernels(lines).items():
        if sub not in name or b is None:
            continue
        body = lines[a:b]
        meta = {k: v for l in body for k, v in re.findall(r"\.amdhsa_(next_free_vgpr|accum_offset|private_segment_fixed_size)\s+(\d+)", l)}
        print("==", name[:90], meta)
        blk_start, valu, mad, hist = 0, 0, 0, collections.Counter()
        tot = collections.Counter()
        for i, l in enumerate(body):
            t = l.strip()
            op = t.split()[0] if t and not t.startswith((";", ".")) else ""
            is_label = re.match(r"^\.LBB\d+_\d+:", t) is not None
            if is_label or op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                if valu >= mn:
                    print("  lines %6d-%6d  valu %5d  mad %5d  %s" % (blk_start, i, valu, mad,
                          " ".join("%s:%d" % (k.replace("v_", ""), v) for k, v in hist.most_common(8))))
                blk_start, valu, mad, hist = i, 0, 0, collections.Counter()
            if op.startswith("v_"):
                valu += 1
                hist[op] += 1
                tot[op] += 1
                if op == "v_mad_u64_u32":
                    mad += 1
        print("  total valu %d, scratch ops %d" % (sum(tot.values()), sum(1 for l in body if "scratch_" in l)))


if __name__ == "__main__":
    main()
