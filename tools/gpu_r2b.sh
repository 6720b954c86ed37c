#!/bin/bash
# round 2, step B: same-box knob sweep on the GLV walk
mkdir -p gpurun_out/r2b
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2b/$tag.json 2> gpurun_out/r2b/$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2b/$tag.json"))
    print("$tag", d["value"], d["ms_per_step"], d["stage_ms"], d["roofline"]["launch_ms_alone"])
except Exception as e:
    print("$tag FAILED", e)
PY
}
run base A=1
run base2 A=1
run ntt29 RLNAMD_NTT29=1
run chunk8 RLNAMD_MSM_CHUNK=8
run chunk32 RLNAMD_MSM_CHUNK=32
run g2chunk4 RLNAMD_MSM_CHUNK_G2=4
run g2chunk16 RLNAMD_MSM_CHUNK_G2=16
run slots6 RLNAMD_SLOTS=6
run nosplit RLNAMD_MSM_SPLIT=0
run base3 A=1
