"""Multi-GPU: one process per GPU, whole sequences (videos) per rank, no data-path collective.

Sequences are independent (BASELINE config 3) and a BatchNorm batch must never be split, so the
only cross-rank traffic is bookkeeping: a barrier around the timed region and a max-reduction of the elapsed
time (`torch.distributed`, backend "nccl" = RCCL on the GPUs, "gloo" in the CPU tests)."""
import torch


def assign_sequences(frame_counts, world_size):
    """Static longest-processing-time assignment: sequences sorted by descending frame count go to the
    currently least-loaded rank.  Returns a list (per rank) of sequence indices.  Deterministic."""
    order = sorted(range(len(frame_counts)), key=lambda i: (-frame_counts[i], i))
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += frame_counts[i]
    return out


def max_over_ranks(value, dist=None, device="cpu"):
    """Max of a python float over all ranks (identity without a process group)."""
    if dist is None or not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device="cpu"):
    if dist is None or not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
