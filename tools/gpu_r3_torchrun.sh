# the driver's N > 1 launch shape, as far as a 1-GPU box can run it: torchrun with 2 ranks, both on device 0 (small
# tables), gloo for the barrier / max-reduce (RCCL refuses two ranks on one device), config 4 shards of 8 192 per rank
mkdir -p gpurun_out/r3t
RLNAMD_WINDOW_BITS=8 RLNAMD_BENCH_DEVICE=0 RLNAMD_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --no-side-configs > gpurun_out/r3t/n2.json 2> gpurun_out/r3t/n2.err
echo "rc=$?"; wc -l gpurun_out/r3t/n2.json; python3 -c "
import json; d=json.loads(open('gpurun_out/r3t/n2.json').read()); print(d['value'], d['n_gpus'], d['ms_per_step'], d['config']['workload'], d['config']['verified'], d['config']['verified_proofs'])"; tail -3 gpurun_out/r3t/n2.err
# the torchrun path with one rank and RCCL (communicator of one rank for config 5)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3t/n1.json 2> gpurun_out/r3t/n1.err
echo "rc=$?"; python3 -c "
import json; d=json.loads(open('gpurun_out/r3t/n1.json').read()); print(d['value'], d['n_gpus'], d.get('rccl_ranks'), d['config5']['correct'], d['config3']['correct'])"
# plain --gpus 2 without torchrun on a 1-GPU box must refuse
python bench.py --gpus 2 --steps 1 --warmup 0; echo "rc=$? (non-zero expected)"
# the one-process N-device path (rlnamd_pool) with two replicas sharing device 0
RLNAMD_WINDOW_BITS=8 RLNAMD_BENCH_POOL_DEVICES=0,0 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r3t/pool2.json 2> gpurun_out/r3t/pool2.err
echo "rc=$?"; python3 -c "
import json; d=json.loads(open('gpurun_out/r3t/pool2.json').read()); print(d['value'], d['n_gpus'], d['ms_per_step'], d['config']['verified'], d['config']['verified_sample'], d['replica_ms_last_step'])"
