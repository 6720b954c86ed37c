# same-box A/B: does importing torch first (its bundled HIP runtime becomes the process's) change the streamed path,
# and does staging the inputs with a kernel instead of hipMemcpyAsync remove the difference?
echo "--- torch first, H2D by kernel";      RLN_AB_TORCH=1 RLNAMD_H2D_KERNEL=1 timeout 150 python -u tools/stream_ab.py 20 2>/dev/null | tail -1
echo "--- torch first, H2D by hipMemcpyAsync"; RLN_AB_TORCH=1 RLNAMD_H2D_KERNEL=0 timeout 150 python -u tools/stream_ab.py 20 2>/dev/null | tail -1
echo "--- no torch, H2D by kernel";         RLNAMD_H2D_KERNEL=1 timeout 150 python -u tools/stream_ab.py 20 2>/dev/null | tail -1
echo "--- no torch, H2D by hipMemcpyAsync"; RLNAMD_H2D_KERNEL=0 timeout 150 python -u tools/stream_ab.py 20 2>/dev/null | tail -1
echo "--- bench.py"; timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'], d.get('config3'), d.get('config5'))"
