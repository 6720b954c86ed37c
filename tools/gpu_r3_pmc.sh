#!/bin/bash
# round 3: PMC passes over the bench (counters only -- never combined with tracing), then the kernel trace + stats
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3pmc
mkdir -p $O
pass() { tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-configs > $O/$tag.json 2> $O/$tag.err
  echo "$tag rc=$?"
}
pass sq SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
python3 tools/pmc_collect.py $O $O/sq.json $O/r3_pmc_walks.json
# keep the merged directory small
find $O -name "*.csv" -size +6M -delete
# kernel trace + stats of the default bench (the driver's command line)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/stats.json 2> $O/stats.err
echo "stats rc=$?"
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r3_kernel_stats.csv
find $O/stats -name "*kernel_trace.csv" -size +6M -delete
head -c 600 $O/stats.json
