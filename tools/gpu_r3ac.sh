# lone mid-size batches: fused plan (+25 % G1 rows) on / off, both table sizes
timeout 300 python tools/lanechunk_sweep.py 2>/dev/null | tail -1
RLNAMD_FUSED_SMUL=0 timeout 300 python tools/lanechunk_sweep.py 2>/dev/null | tail -1
MAXB=1024 WBITS=7150114 timeout 600 python tools/lanechunk_sweep.py 2>/dev/null | tail -1
RLNAMD_FUSED_SMUL=0 MAXB=1024 WBITS=7150114 timeout 600 python tools/lanechunk_sweep.py 2>/dev/null | tail -1
