# fabric / L2 counters of the walks per proof at 1 024 and 8 192 proofs per launch (schedule 114); counters only, no tracing
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_ab_pmc
mkdir -p $O
for cfg in "1024 5 3" "8192 3 2"; do
  set -- $cfg
  export RLNAMD_WINDOW_BITS=114 RLNAMD_SLOTS=$2
  timeout 500 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/ea_$1 -- python3 bench.py --batch $1 --steps $3 --warmup 1 --no-side-configs --no-cpu-baseline --sustained-seconds 0 > $O/ea_$1.json 2> $O/ea_$1.err
  echo "pmc ea $1 rc=$?"
  timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$1 -- python3 bench.py --batch $1 --steps $3 --warmup 1 --no-side-configs --no-cpu-baseline --sustained-seconds 0 > $O/fetch_$1.json 2> $O/fetch_$1.err
  echo "pmc fetch $1 rc=$?"
done
python3 tools/pmc_walk_per_proof.py $O/ea_1024 1024 $O/ea_8192 8192 > $O/ea_per_proof.json; cat $O/ea_per_proof.json
python3 tools/pmc_walk_per_proof.py $O/fetch_1024 1024 $O/fetch_8192 8192 > $O/fetch_per_proof.json; cat $O/fetch_per_proof.json
find $O -name "*.csv" -size +2M -delete
