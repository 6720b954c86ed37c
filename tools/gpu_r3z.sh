# mid-size batches: walks with lanes = chunks up to RLNAMD_LANECHUNK_WALK proofs, lanes = proofs over the short chunks above
for t in 128 0 4 8 16 32; do echo "LANECHUNK_WALK=$t"; RLNAMD_LANECHUNK_WALK=$t timeout 300 python tools/lanechunk_sweep.py 2>/dev/null | tail -1; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "variants or lane_chunk or golden" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
