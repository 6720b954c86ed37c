#!/bin/bash
# single-proof kernel timeline (rocprofv3 kernel trace of tools/single_latency.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/single_latency.py > $O/trace.json 2> $O/trace.err
echo "trace rc=$?"
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/single_timeline.py $f 2>&1 | tail -45
cat $O/trace.json | cut -c1-400
find $O -name "*.csv" -size +4M -delete
