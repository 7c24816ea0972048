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


def test_split_tracks_partitions_a_step():
    """SURVEY.md 8e case 2: B tracks of one step over N ranks - contiguous, disjoint, complete, balanced to one track."""
    import numpy as np
    for B, N in ((512, 8), (512, 3), (5, 8), (0, 2), (31, 2)):
        parts = sharding.split_tracks(B, N)
        assert len(parts) == N and parts[0][0] == 0 and parts[-1][1] == B
        assert all(parts[i][1] == parts[i + 1][0] for i in range(N - 1))
        sizes = [hi - lo for lo, hi in parts]
        assert max(sizes) - min(sizes) <= 1 and sharding.split_tracks(B, N, rank=N - 1) == parts[-1]
    full = np.arange(31 * 7, dtype=np.float32).reshape(31, 7)
    assert np.array_equal(sharding.concat_tracks([full[lo:hi] for lo, hi in sharding.split_tracks(31, 4)]), full)
    assert np.array_equal(sharding.concat_tracks([full[lo:hi] for lo, hi in sharding.split_tracks(31, 40)]), full)   # empty slices


def _split_worker(rank, world, port, q):
    """Each rank computes a (stand-in) per-track function on ITS slice of the step only; the gathered result must equal
    the single-process result - tracks are independent, no data-path collective."""
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, P = 37, 5
    feats = np.random.default_rng(5).standard_normal((B, 11 + P, 8)).astype(np.float32)       # same on every rank (seeded)
    per_track = lambda x: np.tanh(x.sum(axis=1))[:, :P + 2]                                     # any per-track function
    lo, hi = sharding.split_tracks(B, world, rank)
    mine = per_track(feats[lo:hi])
    full = sharding.gather_track_slices(mine, dist)
    q.put((rank, (lo, hi), bool(np.array_equal(full, per_track(feats))), full.shape))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_track_split_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_split_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == (0, 19) and res[1][1] == (19, 37)
    assert all(r[2] for r in res) and res[0][3] == (37, 7)


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


def test_bench_parent_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` without a launcher is a parent that touches no GPU at all (it does not even count devices) and starts
    the ranks as child processes; with fewer than N devices visible - none in the build container - every rank refuses, and the parent
    relays that: non-zero exit, the ranks' message, no JSON line - never a silent fall-back to fewer ranks (the N > 1 happy path is
    tests/test_bench_gpu.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    n = torch.cuda.device_count() + 2
    env = {k: v for k, v in os.environ.items() if k not in ("BUSCA_BENCH_BACKEND", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "4", "--warmup", "1", "--no-variants",
                        "--cpu-seconds", "0", "--latency-samples", "0"], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode != 0, (r.returncode, r.stderr[-500:])
    assert "refusing" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
