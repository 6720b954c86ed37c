#!/bin/bash
# round 2, step F: memory-pipeline counters of the walks (separate PMC passes, counters only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
pass() { tag=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/r2f/$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2f/$tag.json 2> gpurun_out/r2f/$tag.err
  echo "$tag rc=$?"
}
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum
pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass utcl TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum
pass sq SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
python - <<'PY'
import csv, glob, collections
for tag in ("tcp","tcc","ea","utcl","sq","grbm"):
    for f in glob.glob("gpurun_out/r2f/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "k_msm29" in k:
                agg[("G1" if "G1Acc29" in k else "G2", r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(agg.items()):
            print(tag, k, len(v), sum(v)/len(v))
PY
# keep the csv small
find gpurun_out/r2f -name "*.csv" -size +8M -delete
