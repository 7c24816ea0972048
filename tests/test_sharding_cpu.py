"""CPU: the N>1 path (sequence sharding + barrier/max-reduce bookkeeping) with world_size 2 over gloo."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from busca_amd import sharding


def test_assign_sequences_lpt():
    # MOT17-val frame counts (7 sequences): 8 ranks -> one idles
    frames = [600, 1050, 837, 525, 654, 900, 750]
    a = sharding.assign_sequences(frames, 8)
    assert sorted(i for r in a for i in r) == list(range(7)) and sum(1 for r in a if not r) == 1
    b = sharding.assign_sequences(frames, 2)
    loads = [sum(frames[i] for i in r) for r in b]
    assert abs(loads[0] - loads[1]) <= min(frames)
    assert sharding.assign_sequences(frames, 2) == b               # deterministic


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.assign_sequences([30, 10, 20, 40, 5], world)[rank]
    dist.barrier()
    elapsed = 1.0 + rank                                          # pretend rank 1 was slower
    tmax = sharding.max_over_ranks(elapsed, dist)
    total = sharding.sum_over_ranks(sum(mine), dist)
    dist.barrier()
    q.put((rank, mine, tmax, total))
    dist.destroy_process_group()


def test_two_rank_bookkeeping_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, m0, t0, s0), (r1, m1, t1, s1) = res
    assert sorted(m0 + m1) == [0, 1, 2, 3, 4] and not set(m0) & set(m1)
    assert t0 == t1 == 2.0                                         # the bench reports the slowest rank
    assert s0 == s1 == 10.0
