#!/bin/bash
# kernel timeline of a 64-proof batch
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t
mkdir -p $O
export N=64
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/single_latency.py > $O/trace.json 2> $O/trace.err
echo "trace rc=$?"
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/single_timeline.py $f 2>&1 | tail -45
cut -c1-300 $O/trace.json
find $O -name "*.csv" -size +4M -delete
