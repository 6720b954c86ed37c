# small-batch chunk sizes after the 512-lane sum tree (rows per chunk: G1 / G2)
run() { echo "$*"; env "$@" timeout 100 python tools/ffi_latency.py 2>/dev/null | head -1 | cut -c1-120; }
run A=base
run RLNAMD_MSM_CHUNK_G2_SMALL=1
run RLNAMD_MSM_CHUNK_G2_SMALL=3
run RLNAMD_MSM_CHUNK_SMALL=2
run RLNAMD_MSM_CHUNK_SMALL=3
run RLNAMD_MSM_CHUNK_SMALL=3 RLNAMD_MSM_CHUNK_G2_SMALL=1
run A=base
