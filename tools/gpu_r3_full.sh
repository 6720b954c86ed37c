# round-end style check: the whole GPU suite, smoke(), the default bench line
mkdir -p gpurun_out/r3full
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r3full/gputests.log 2>&1; tail -3 gpurun_out/r3full/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3full/smoke.log 2>&1; tail -1 gpurun_out/r3full/smoke.log
timeout 900 python bench.py > gpurun_out/r3full/bench.json 2> gpurun_out/r3full/bench.err; wc -l gpurun_out/r3full/bench.json; head -c 1500 gpurun_out/r3full/bench.json
