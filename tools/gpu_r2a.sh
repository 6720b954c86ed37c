#!/bin/bash
# round 2, step A: parity of the GLV walk + same-box A/B against the round-1 19-window walk
set -x
mkdir -p gpurun_out/r2a
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2a/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2a/pytest.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r2a/bench_glv.json 2> gpurun_out/r2a/bench_glv.err; echo "bench rc=$?"; cat gpurun_out/r2a/bench_glv.json | cut -c1-1500
RLNAMD_GLV=0 RLNAMD_WINDOW_BITS=813 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2a/bench_r1.json 2> gpurun_out/r2a/bench_r1.err; echo "bench rc=$?"; cat gpurun_out/r2a/bench_r1.json | cut -c1-1200
