#!/bin/bash
# round 2, step E: shader clock under the walks
mkdir -p gpurun_out/r2e
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fq29 or batch or golden" > gpurun_out/r2e/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2e/pytest.log
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2e/bench$i.json 2> gpurun_out/r2e/bench$i.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/r2e/bench$i.json"))
print(d["value"], d["ms_per_step"], d["stage_ms"], d["roofline"]["launch_ms_alone"], d["shader_clock_mhz"])
PY
done
RLNAMD_MSM_SPLIT=0 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2e/nosplit.json 2> gpurun_out/r2e/nosplit.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2e/nosplit.json"))
print("nosplit", d["value"], d["ms_per_step"], d["stage_ms"], d["roofline"]["launch_ms_alone"], d["shader_clock_mhz"])
PY
