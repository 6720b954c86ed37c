#!/bin/bash
# PMC view of the lone-wave interpreter (k_witness_lanes) under single proofs
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n
mkdir -p $O
pass() { tag=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $O/$tag -- python3 tools/ffi_latency.py > $O/$tag.json 2> $O/$tag.err
  echo "$tag rc=$?"
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVES
pass b SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass c SQ_INSTS_BRANCH SQ_INSTS_CBRANCH SQ_INSTS_CBRANCH_TAKEN SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r3n/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        for k in ("k_witness_lanes","k_msm29","k_ntt_pass","k_hquot"):
            if k in n: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in agg.items():
    print(k, {c: round(sum(v)/len(v)) for c,v in sorted(cs.items())}, "launches", max(len(v) for v in cs.values()))
PY
tail -3 $O/c.err
find $O -name "*.csv" -size +4M -delete
