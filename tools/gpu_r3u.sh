# crossover of the small-batch shapes: walk form threshold (RLNAMD_LANECHUNK) x interpreter threshold
for lc in 0 4 8 16 32 64 128; do RLNAMD_LANECHUNK=$lc timeout 300 python tools/lanechunk_sweep.py 2>/dev/null | tail -1; done
RLNAMD_LANECHUNK=0 RLNAMD_WITLANES_MAX=0 timeout 300 python tools/lanechunk_sweep.py 2>/dev/null | tail -1
