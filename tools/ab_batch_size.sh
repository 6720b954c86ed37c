cd $GRAFT_REPO_ROOT
for cfg in "1024 5 12" "2048 5 8" "4096 5 6" "8192 3 4"; do
  set -- $cfg
  RLNAMD_WINDOW_BITS=114 RLNAMD_SLOTS=$2 timeout 400 python bench.py --batch $1 --steps $3 --warmup 2 --no-side-configs --no-cpu-baseline --sustained-seconds 6 > gpurun_out/r6_ab_batch_$1.json 2> gpurun_out/r6_ab_batch_$1.err
  echo "batch $1 rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r6_ab_batch_$1.json").read().strip().splitlines()[-1])
    print($1, d["value"], d["ms_per_step"], d["sustained"]["proofs_per_s"], d["sustained"]["shader_clock_mhz"], d["stage_ms"]["overlapped"].get("msm_g1"), d["config"]["verified"], d["config"]["table_gib"])
except Exception as e:
    print("fail", e); print(open("gpurun_out/r6_ab_batch_$1.err").read()[-800:])
PY
done
