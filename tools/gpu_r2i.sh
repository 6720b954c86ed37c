#!/bin/bash
# round 2, step I: Fr29 graph interpreter (final form): full suite, smoke, bench A/B against RLNAMD_WIT29=0, FFI latency
mkdir -p gpurun_out/r2i
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2i/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2i/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2i/$tag.json 2> gpurun_out/r2i/$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2i/$tag.json"))
    print("$tag", d["value"], d["ms_per_step"], d["stage_ms"]["witness"], d["shader_clock_mhz"]["timed_region"])
except Exception as e:
    print("$tag FAILED", e)
PY
}
run wit29 A=1
run wit32 RLNAMD_WIT29=0
run wit29b A=1
run wit32b RLNAMD_WIT29=0
timeout 300 python tools/ffi_latency.py > gpurun_out/r2i/ffi_latency.txt 2>&1; tail -12 gpurun_out/r2i/ffi_latency.txt
