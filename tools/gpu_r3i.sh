mkdir -p gpurun_out/r3i
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r3i/gputests.log 2>&1; grep -n "passed\|failed" gpurun_out/r3i/gputests.log | tail -2
timeout 300 python tools/ffi_latency.py 2>&1 | head -1
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'], d['config3']['build_ms'], d['config3']['paths_ms'], d['config5']['ms'], d['config']['init_s'])"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3i/prof -- python3 tools/single_latency.py > gpurun_out/r3i/lat.log 2>&1
f=$(find gpurun_out/r3i/prof -name "*kernel_trace.csv" | head -1); python3 tools/single_timeline.py $f > gpurun_out/r3i/timeline.txt; cat gpurun_out/r3i/timeline.txt | grep -v ntt_pass
find gpurun_out/r3i -name "*.csv" -size +4M -delete
