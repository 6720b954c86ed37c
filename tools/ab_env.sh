#!/bin/bash
# same-box A/B of the bench under environment switches:  tools/ab_env.sh <out-dir> "<tag> VAR=val ..." ...
# every run: python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs; prints one summary row per run
O=$1; shift; mkdir -p $O
for spec in "$@"; do
  set -- $spec; tag=$1; shift
  env "$@" timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs > $O/$tag.json 2> $O/$tag.err
  python3 - "$O/$tag" <<'PY'
import json, sys
t = sys.argv[1]
try:
    l = json.loads(open(t + ".json").read().strip().splitlines()[-1])
    print(t.split("/")[-1], l["value"], l["ms_per_step"], l["stage_ms"]["overlapped"].get("msm_g1"), l["stage_ms"]["overlapped"].get("msm_g2"),
          l["shader_clock_mhz"]["timed_region"], l["config"]["verified"])
except Exception as e:
    print(t, "failed", e, open(t + ".err").read()[-300:])
PY
done
