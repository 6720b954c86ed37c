#!/bin/bash
# kernel durations + PMC split of the lone-wave interpreter under single proofs
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/single_latency.py > $O/trace.json 2> $O/trace.err
echo "trace rc=$?"
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/single_timeline.py $f 2>&1 | tail -45
pass() { tag=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $O/$tag -- python3 tools/single_latency.py > $O/$tag.json 2> $O/$tag.err
  echo "$tag rc=$?"
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVES
pass b SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass c SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r3p/[abc]/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        for k in ("k_witness_lanes",):
            if k in n: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in agg.items():
    print(k, {c: round(sum(v)/len(v)) for c,v in sorted(cs.items())}, "launches", max(len(v) for v in cs.values()))
PY
find $O -name "*.csv" -size +4M -delete
