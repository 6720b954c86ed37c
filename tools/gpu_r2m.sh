#!/bin/bash
# round 2, step M: occupancy of the G1 walk against clock / throughput (RLNAMD_MSM_WAVES caps waves per SIMD through LDS)
mkdir -p gpurun_out/r2m
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2m/$tag.json 2> gpurun_out/r2m/$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2m/$tag.json"))
    print("$tag", d["value"], d["ms_per_step"], d["roofline"]["launch_ms_alone"], d["shader_clock_mhz"])
except Exception as e:
    print("$tag FAILED", e)
PY
}
run base A=1
run waves3 RLNAMD_MSM_WAVES=3
run waves2 RLNAMD_MSM_WAVES=2
run base2 A=1
run chunk32 RLNAMD_MSM_CHUNK=32 RLNAMD_MSM_CHUNK_G2=16
run chunk32b RLNAMD_MSM_CHUNK=32 RLNAMD_MSM_CHUNK_G2=16
run base3 A=1
