"""world_size-2 gloo test of the N > 1 path (proof index sharding, no data-path collective; results
returned with one all_gather).  The per-rank prover is the C oracle here (no GPU in this container); the GPU
prover plugs into the same prove_fn slot."""
import os
import socket

import torch.distributed as dist
import torch.multiprocessing as mp

from zerokit_amd.distributed import max_over_ranks, prove_sharded, shard_bounds


def test_shard_bounds():
    assert shard_bounds(65536, 8) == [(8192 * i, 8192 * (i + 1)) for i in range(8)]
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert shard_bounds(0, 2) == [(0, 0), (0, 0)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.c import binding as ob
    from oracle.pyref import workload
    c = ob.Circuit(20)

    def prove_fn(ws, rs):
        _, proofs, pub = c.prove_many(ws, rs, threads=2)
        return [dict(proof=p, public_inputs=v) for p, v in zip(proofs, pub)]

    ws, rs = workload.config2_witnesses(n, seed=777)
    out = prove_sharded(prove_fn, ws, rs)
    slowest = max_over_ranks(1.0 + rank)
    if rank == 0:
        q.put(([o["proof"] for o in out], [o["public_inputs"] for o in out], slowest))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_proving_matches_single_process():
    n, world = 5, 2   # ragged: shards of 3 and 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    proofs, pubs, slowest = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle.c import binding as ob
    from oracle.pyref import workload
    ws, rs = workload.config2_witnesses(n, seed=777)
    _, ref_proofs, ref_pub = ob.Circuit(20).prove_many(ws, rs, threads=4)
    assert proofs == ref_proofs and pubs == ref_pub
    assert slowest == 2.0


class _FakeMsm:
    """stands in for MsmG1 on CPU: 'window sums' are 16 integers (sum of k_i s_i digits would be EC points on
    the GPU); combine is the same add-then-Horner fold, over the integers"""

    def generate(self, seed, first, n):
        self.vals = [(seed + first + i) % 1009 for i in range(n)]

    def run_windows(self):
        w = [sum(v >> (3 * k) & 7 for v in self.vals) for k in range(16)]
        return b"".join(x.to_bytes(8, "little") for x in w), {}

    def combine(self, blobs):
        tot = 0
        for k in reversed(range(16)):
            tot = tot * 8 + sum(int.from_bytes(b[8 * k:8 * k + 8], "little") for b in blobs)
        return tot


def _msm_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zerokit_amd.distributed import msm_sharded
    res, _ = msm_sharded(_FakeMsm(), 5, 1000)
    if rank == 1:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_msm_gather_and_fold():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_msm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == sum((5 + i) % 1009 for i in range(1000))
