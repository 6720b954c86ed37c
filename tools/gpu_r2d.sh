#!/bin/bash
# round 2, step D: wide reduction rounds + folded subtractions: parity, then bench
mkdir -p gpurun_out/r2d
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2d/pytest.log
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2d/bench$i.json 2> gpurun_out/r2d/bench$i.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/r2d/bench$i.json"))
print(d["value"], d["ms_per_step"], d["stage_ms"], d["roofline"]["launch_ms_alone"])
PY
done
timeout 300 python bench.py --workload merkle > gpurun_out/r2d/merkle.json 2> gpurun_out/r2d/merkle.err; cut -c1-600 gpurun_out/r2d/merkle.json
