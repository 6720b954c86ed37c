#!/bin/bash
# PMC of the walks of a 64-proof batch (lanes = proofs over the short chunks)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ab
mkdir -p $O
export N=64
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/a -- python3 tools/single_latency.py > $O/a.json 2> $O/a.err
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES_EQ_64 --output-format csv -d $O/b -- python3 tools/single_latency.py > $O/b.json 2> $O/b.err
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r3ab/[ab]/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        k = "G1" if ("k_msm29" in n and "G1Acc29" in n) else "G2" if "k_msm29" in n else None
        if k: agg[k+":"+r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in sorted(agg.items()):
    print(k, {c: round(sum(v)/len(v)) for c,v in sorted(cs.items())}, "launches", max(len(v) for v in cs.values()))
PY
tail -2 $O/b.err
find $O -name "*.csv" -size +4M -delete
