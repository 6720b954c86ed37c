#!/bin/bash
# round 2, step L: pipeline fill / drain: total time of K steps for several K on one box
mkdir -p gpurun_out/r2l
for K in 5 10 20 40 20 10 5; do
python bench.py --steps $K --warmup 3 --no-cpu-baseline > gpurun_out/r2l/k$K.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r2l/k$K.json"))
print("K=%2d total %.1f ms  per step %.3f  value %.0f" % ($K, d["ms_per_step"]*$K, d["ms_per_step"], d["value"]))
PY
done
