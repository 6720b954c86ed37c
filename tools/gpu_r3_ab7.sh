# same-box A/B of the non-walk knobs (bench.py --steps 20 --warmup 2, no side configs)
run() { echo "--- $*"; env "$@" timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-side-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"; }
run A=base
run RLNAMD_NTT29=1
run A=base2
run RLNAMD_SLOTS=6
run RLNAMD_NTT29=1 RLNAMD_SLOTS=6
run RLNAMD_WSTREAMS=1
