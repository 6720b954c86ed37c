mkdir -p gpurun_out/r5z
for c in 16 15 14 13; do
  L=zerokit_amd/lib_exp/librln_c$c.so; [ $c = 16 ] && L=zerokit_amd/lib/librln.so
  for lg in 21 24; do
    RLNAMD_LIB=$L RLNAMD_MSM_LOG2=$lg timeout 300 python3 bench.py --workload msm --steps 5 2>/dev/null > gpurun_out/r5z/c${c}_$lg.json
    python3 - <<PY
import json
d=json.loads(open("gpurun_out/r5z/c${c}_$lg.json").read().strip().splitlines()[-1])
print("c=$c 2^$lg", d["ms"], d["stage_ms_rank0"], d["correct"], "g2", d.get("g2_2^%d"%($lg-2),{}).get("ms"))
PY
  done
done
