#!/bin/bash
# round 2, step C: rocprofv3 kernel trace + PMC passes (FETCH_SIZE / WRITE_SIZE separately) of the default bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2c
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2c/trace -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline > gpurun_out/r2c/trace_bench.json 2> gpurun_out/r2c/trace.err
echo "trace rc=$?"; cut -c1-400 gpurun_out/r2c/trace_bench.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2c/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2c/fetch_bench.json 2> gpurun_out/r2c/fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2c/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2c/write_bench.json 2> gpurun_out/r2c/write.err
echo "write rc=$?"
find gpurun_out/r2c -name "*.csv" | head -30
# keep only the small summaries (the merge back is limited to 64 MiB)
find gpurun_out/r2c -name "*kernel_trace.csv" -size +20M -delete
python bench.py --steps 20 --warmup 5 > gpurun_out/r2c/bench_plain.json 2> gpurun_out/r2c/bench_plain.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r2c/bench_plain.json
