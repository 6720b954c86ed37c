cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3h
for v in 1 0; do
RLNAMD_WITROWS=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3h/prof$v -- python3 tools/single_latency.py > gpurun_out/r3h/lat$v.log 2>&1
f=$(find gpurun_out/r3h/prof$v -name "*kernel_trace.csv" | head -1); python3 tools/single_timeline.py $f > gpurun_out/r3h/timeline$v.txt; grep "k_witness_lanes\|total\|k_proof_values\|k_fin_out" gpurun_out/r3h/timeline$v.txt
done
find gpurun_out/r3h -name "*.csv" -size +4M -delete
