"""CPU suite: the N>1 path of bench.py (one process per sequence, throughput aggregation) with
world_size 2 over gloo.  Each rank also runs its own sequence through the CPU oracle to show the
shards are independent: different seeds, different results, no exchange."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT, pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard, synth = pkg("shard"), pkg("synth")
    from oracle import bindings as B
    assert shard.rank_info() == (rank, rank, world)
    seed = shard.sequence_seed(rank)
    seq = synth.stereo_sequence(seed, 320, 128, 3)
    m = B.CpuMatcher("oracle")
    for l, r in seq:
        m.push_back(l, r)
        m.match(2)
    n = len(m.matches())
    shard.barrier(dist)
    units, seconds = 10.0 * (rank + 1), 2.0 + rank      # rank0: 10 units in 2 s, rank1: 20 units in 3 s
    tot, tmax, ok = shard.aggregate(dist, torch, units, seconds, all_ok=True)
    # one rank reports a failed verification: every rank must see it (MIN reduction), sums and maxima unchanged
    tot2, tmax2, ok2 = shard.aggregate(dist, torch, units, seconds, all_ok=(rank == 0))
    rows = shard.gather_rows(dist, torch, [rank, 14 - 6 * rank, 110 - 30 * rank, 4.0 + rank])   # what bench.py reports per rank
    assert rows == [[0.0, 14.0, 110.0, 4.0], [1.0, 8.0, 80.0, 5.0]], rows
    out.put((rank, seed, n, tot, tmax, ok, tot2, tmax2, ok2))
    dist.destroy_process_group()


def test_two_rank_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, n0, tot0, t0, ok0, tb0, tmb0, okb0), (r1, s1, n1, tot1, t1, ok1, tb1, tmb1, okb1) = res
    assert (s0, s1) == (1234, 1235)
    assert n0 > 50 and n1 > 50 and n0 != n1          # independent sequences, independent results
    assert tot0 == tot1 == 30.0 and t0 == t1 == 3.0   # sum of units / max of seconds on every rank
    assert ok0 and ok1
    assert not okb0 and not okb1                      # rank 1 said "not verified": both ranks know
    assert tb0 == tb1 == 30.0 and tmb0 == tmb1 == 3.0
    assert abs(tot0 / t0 - 10.0) < 1e-12
