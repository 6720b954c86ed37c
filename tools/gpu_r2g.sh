#!/bin/bash
# round 2, step G: full GPU suite on the clock-tap build, bench with the VALU view, NTT29 A/B
mkdir -p gpurun_out/r2g
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2g/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2g/pytest.log
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2g/$tag.json 2> gpurun_out/r2g/$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2g/$tag.json"))
    print("$tag", d["value"], d["ms_per_step"], d["roofline"]["launch_ms_alone"], d["shader_clock_mhz"], d["roofline"]["valu"])
except Exception as e:
    print("$tag FAILED", e)
PY
}
run base A=1
run ntt29 RLNAMD_NTT29=1
run base2 A=1
run ntt29b RLNAMD_NTT29=1
