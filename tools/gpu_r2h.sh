#!/bin/bash
# round 2, step H: Fr29 graph interpreter: parity, stage times, bench A/B against RLNAMD_WIT29=0
mkdir -p gpurun_out/r2h
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "interpreters or witness or other_circuits or edge_case or partial" > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2h/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
RLNAMD_WIT29=0 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2h/$tag.json 2> gpurun_out/r2h/$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2h/$tag.json"))
    print("$tag", d["value"], d["ms_per_step"], d["stage_ms"])
except Exception as e:
    print("$tag FAILED", e)
PY
}
run wit29 A=1
run wit32 RLNAMD_WIT29=0
run wit29b A=1
run wit32b RLNAMD_WIT29=0
