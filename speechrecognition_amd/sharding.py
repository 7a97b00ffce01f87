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
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), float(n_frames)
    import torch

    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    f = torch.tensor([float(n_frames)], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return float(t.item()), float(f.item())
