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


def split_tracks(n_tracks, world_size, rank=None):
    """SURVEY.md 8e case 2 (BASELINE configs[4], 512 lost x 64 proposals x d512 on 8 GPUs): the lost tracks of ONE
    association step are independent sequences of the Decision Transformer (attention never crosses tracks), so rank r
    takes the contiguous slice [lo_r, hi_r) of the B tracks, computes its [hi_r - lo_r, P + 2] outputs with replicated
    weights and returns them - no collective on the data path; the host concatenates in rank order (`concat_tracks`).
    Slices differ by at most one track.  Returns the list of (lo, hi) for all ranks, or this rank's pair.
    ReID is NOT split this way: a BatchNorm batch stays on one GPU (its statistics span the batch)."""
    base, extra = divmod(int(n_tracks), int(world_size))
    bounds, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < extra else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds if rank is None else bounds[rank]


def concat_tracks(parts):
    """Per-rank output slices (rank order; empty slices allowed) -> the full [B, ...] array."""
    import numpy as np
    parts = [np.asarray(p) for p in parts]
    keep = [p for p in parts if p.shape[0] > 0]
    return np.concatenate(keep, axis=0) if keep else parts[0]


def gather_track_slices(local, dist=None):
    """Host-side gather of every rank's slice to every rank (bookkeeping traffic: B*(P+2) floats), rank order."""
    if dist is None or not dist.is_initialized():
        return concat_tracks([local])
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local)
    return concat_tracks(parts)


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
