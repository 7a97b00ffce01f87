"""Utterance sharding for one-process-per-GPU runs.

The reference parallelises over utterances only (`#pragma omp parallel for` in Recognizer::recognize,
sietill/Recognizer.cpp:46-47); utterances are independent, so ranks need no data-path collective:
every rank holds a full model replica, decodes its shard and rank 0 gathers the (tiny) results.
Work is proportional to frames, so shards are balanced by total frames (longest-processing-time first).
"""
from __future__ import annotations

import numpy as np


def shard_utterances(frame_off, world_size):
    """-> list (one per rank) of utterance index arrays, frame-balanced by greedy LPT; deterministic."""
    lens = np.diff(np.asarray(frame_off, dtype=np.int64))
    order = np.argsort(-lens, kind="stable")
    loads = np.zeros(world_size, dtype=np.int64)
    bins = [[] for _ in range(world_size)]
    for u in order:
        r = int(np.argmin(loads))
        bins[r].append(int(u))
        loads[r] += lens[u]
    return [np.asarray(sorted(b), dtype=np.int64) for b in bins]


def take_shard(feats, frame_off, utts):
    """Sub-corpus (feats, frame_off) holding utterances `utts` in the given order."""
    frame_off = np.asarray(frame_off, dtype=np.int64)
    parts = [feats[frame_off[u]:frame_off[u + 1]] for u in utts]
    lens = [len(p) for p in parts]
    sub = np.concatenate(parts) if parts else np.zeros((0, feats.shape[1]), dtype=feats.dtype)
    return sub, np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)


def imbalance(frame_off, world_size):
    """frames in the heaviest shard / mean frames per shard (1.0 = perfect)."""
    lens = np.diff(np.asarray(frame_off, dtype=np.int64))
    shards = shard_utterances(frame_off, world_size)
    loads = np.array([lens[s].sum() for s in shards], dtype=np.float64)
    return float(loads.max() / loads.mean()) if loads.mean() > 0 else 1.0


def gather_words(words, word_off, utts, n_utts_total, dist=None, dst=0):
    """Collect per-utterance word lists on rank `dst` (torch.distributed object gather; the payload is a
    few KB).  Returns a list of n_utts_total arrays on dst, None elsewhere.  dist=None: single process."""
    local = {int(u): np.asarray(words[int(word_off[i]):int(word_off[i + 1])]) for i, u in enumerate(utts)}
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        parts = [local]
    else:
        parts = [None] * dist.get_world_size() if dist.get_rank() == dst else None
        dist.gather_object(local, parts, dst=dst)
        if dist.get_rank() != dst:
            return None
    out = [None] * n_utts_total
    for part in parts:
        for u, w in part.items():
            out[u] = w
    return out


def reduce_timing(elapsed_s, n_frames, dist=None, device=None):
    """bench.py's contract: max over ranks of the step time, sum over ranks of frames."""
    if dist is None or not dist.is_initialized():
        return float(elapsed_s), float(n_frames)
    import torch

    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    f = torch.tensor([float(n_frames)], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return float(t.item()), float(f.item())


def allreduce_accumulators(acc, dist=None, device=None):
    """EM statistics of a sharded corpus: every rank accumulated its own utterances (sr_accumulate_corpus); the model
    update needs the sums over all ranks -- the one real exchange step on this path.  One all-reduce
    (RCCL when `device` is a GPU, gloo on CPU) over the four arrays packed into a single buffer: C*(2D+2) doubles,
    81 MB at 128k densities, link-bound on xGMI at ~1 ms.  The variance rows start at 1e-4 on EVERY rank
    (reset_accumulators, Mixtures.cpp:243), so (world-1)*1e-4 is taken off again.  Summation order across ranks
    differs from the single-process frame order: equal to ~1e-16 relative, not bitwise."""
    mean_acc, mean_w, var_acc, var_w = acc
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return acc
    import torch

    flat = np.concatenate([mean_acc.ravel(), mean_w.ravel(), var_acc.ravel(), var_w.ravel()])
    t = torch.from_numpy(flat)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    flat = t.cpu().numpy()
    n0, n1, n2 = mean_acc.size, mean_w.size, var_acc.size
    out_var = flat[n0 + n1:n0 + n1 + n2].reshape(var_acc.shape) - (dist.get_world_size() - 1) * 1e-4
    return (flat[:n0].reshape(mean_acc.shape), flat[n0:n0 + n1].copy(), out_var, flat[n0 + n1 + n2:].copy())
