run() { echo "$*"; env "$@" timeout 100 python tools/ffi_latency.py 2>/dev/null | head -1 | cut -c1-150; }
run A=base
run RLNAMD_MSM_CHUNK_G2_SMALL=2
run RLNAMD_MSM_CHUNK_G2_SMALL=2 RLNAMD_MSM_CHUNK_SMALL=2
run RLNAMD_MSM_CHUNK_G2_SMALL=2 RLNAMD_WARM=4200
run RLNAMD_MSM_CHUNK_G2_SMALL=1 RLNAMD_MSM_CHUNK_SMALL=2 RLNAMD_WARM=4200
run A=base
