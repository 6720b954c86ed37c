#!/bin/bash
# The round's measurement pass on the GPU box (one maintained script instead of one-off wrappers):
#   tools/gpu_profile.sh <tag> [what...]     what = latency | timeline | pmc | stats | finish   (default: the first four)
# Everything lands under gpurun_out/<tag>/; summaries worth keeping are copied into profiles/ by hand.
#   latency   tools/ffi_latency.py (what a caller of include/rln.h sees) and tools/single_latency.py
#   timeline  rocprofv3 --kernel-trace of ONE single proof -> tools/single_timeline.py
#   pmc       counter passes over the bench (counters only, never combined with tracing), merged by tools/pmc_collect.py
#   stats     rocprofv3 --kernel-trace --stats of the driver's bench command line restricted to FULL proofs (the finish and
#             sustained legs of the default line launch the same kernels with other durations: their averages would mix)
#   finish    the same for `bench.py --workload finish`, tools/finish_latency.py and the timeline of one finish per call
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-prof}; shift
WHAT=${*:-latency timeline pmc stats}
O=gpurun_out/$TAG
mkdir -p "$O"
for w in $WHAT; do case $w in
latency)
  timeout 200 python3 tools/ffi_latency.py > $O/ffi_latency.json 2> $O/ffi_latency.err; echo "ffi_latency rc=$?"; cat $O/ffi_latency.json
  timeout 200 python3 tools/single_latency.py > $O/single_latency.json 2> $O/single_latency.err; echo "single_latency rc=$?"; cut -c1-600 $O/single_latency.json
  ;;
timeline)
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/single_latency.py > $O/trace.json 2> $O/trace.err
  echo "trace rc=$?"
  f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
  python3 tools/single_timeline.py $f > $O/single_proof_timeline.txt 2>&1; tail -50 $O/single_proof_timeline.txt
  find $O/trace -name "*.csv" -size +4M -delete
  ;;
pmc)
  pass() { t=$1; shift
    timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/$t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-configs > $O/$t.json 2> $O/$t.err
    echo "pmc $t rc=$?"; }
  pass sq SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES
  pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
  pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
  pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
  pass valu2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_IOPS
  python3 tools/pmc_collect.py $O $O/sq.json $O/pmc_walks.json; echo "collect rc=$?"
  find $O -name "*.csv" -size +6M -delete
  ;;
stats)
  timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --side latency,config3,config5 --sustained-seconds 0 > $O/bench_under_rocprof.json 2> $O/stats.err
  echo "stats rc=$?"
  f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
  find $O/stats -name "*kernel_trace.csv" -size +6M -delete
  head -c 400 $O/bench_under_rocprof.json
  ;;
finish)
  timeout 200 python3 tools/finish_latency.py 40 > $O/finish_latency.json 2> $O/finish_latency.err; echo "finish_latency rc=$?"; cat $O/finish_latency.json
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/ftrace -- python3 tools/finish_latency.py 12 > /dev/null 2> $O/ftrace.err
  f=$(find $O/ftrace -name "*kernel_trace.csv" | head -1)
  python3 tools/single_timeline.py $f k_cone_restore > $O/finish_single_call_timeline.txt 2>&1; tail -40 $O/finish_single_call_timeline.txt
  find $O/ftrace -name "*.csv" -size +4M -delete
  timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fstats -- python3 bench.py --workload finish --steps 20 --warmup 5 > $O/bench_finish_under_rocprof.json 2> $O/fstats.err
  echo "finish stats rc=$?"
  f=$(find $O/fstats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_finish.csv
  find $O/fstats -name "*kernel_trace.csv" -size +6M -delete
  head -c 300 $O/bench_finish_under_rocprof.json
  ;;
esac; done
