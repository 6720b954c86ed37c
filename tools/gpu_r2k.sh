#!/bin/bash
# round 2, step K: kernel trace of the final build + the default bench (with the CPU baseline) twice
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2k/trace -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline > gpurun_out/r2k/trace_bench.json 2> gpurun_out/r2k/trace.err
echo "trace rc=$?"; cut -c1-200 gpurun_out/r2k/trace_bench.json
find gpurun_out/r2k -name "*kernel_trace.csv" -size +20M -delete
python bench.py --steps 20 --warmup 5 > gpurun_out/r2k/bench1.json 2> gpurun_out/r2k/bench1.err; echo "bench rc=$?"; cut -c1-250 gpurun_out/r2k/bench1.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2k/bench2.json 2> gpurun_out/r2k/bench2.err; echo "bench rc=$?"; cut -c1-250 gpurun_out/r2k/bench2.json
python bench.py --workload merkle > gpurun_out/r2k/merkle.json 2>/dev/null; cut -c1-400 gpurun_out/r2k/merkle.json
python bench.py --workload msm > gpurun_out/r2k/msm.json 2>/dev/null; cut -c1-600 gpurun_out/r2k/msm.json
python bench.py --workload finish --no-cpu-baseline > gpurun_out/r2k/finish.json 2>/dev/null; cut -c1-300 gpurun_out/r2k/finish.json
