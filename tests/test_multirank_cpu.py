"""N>1 path on CPU: two gloo ranks shard a corpus by frames, each decodes its shard (the CPU oracle stands in
for the device call -- this test is about sharding, gathering and the bench timing reduction, which are the
only multi-rank logic the path has: there is no data-path collective), rank 0 gathers and compares with a
single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mixset, tmpdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from oracle import pyoracle
    from speechrecognition_amd import sharding, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lex = synth.make_lexicon(6, 3, 1)
    feats, off = synth.make_batch(11, 15, 60, 39, seed=5)
    shards = sharding.shard_utterances(off, world)
    mine = shards[rank]
    sub, sub_off = sharding.take_shard(feats, off, mine)
    orc = pyoracle.Oracle(mixset, 39, lex, am_threshold=200.0)
    words, woff, secs = orc.recognize_batch(sub, sub_off, n_threads=1)
    gathered = sharding.gather_words(words, woff, mine, 11, dist)
    t, f = sharding.reduce_timing(0.5 + rank, float(sub_off[-1]), dist)
    # EM statistics: each rank accumulates its shard, one all-reduce gives the corpus statistics
    states = np.concatenate([np.random.default_rng(100 + int(u)).integers(0, lex.n_states, size=int(off[u + 1] - off[u]))
                             for u in mine]).astype(np.uint16) if len(mine) else np.zeros(0, np.uint16)
    acc = sharding.allreduce_accumulators(orc.accumulate(sub, states), dist)
    if rank == 0:
        all_states = np.concatenate([np.random.default_rng(100 + u).integers(0, lex.n_states, size=int(off[u + 1] - off[u]))
                                     for u in range(11)]).astype(np.uint16)
        want = orc.accumulate(feats, all_states)
        for g, w_ in zip(acc, want):
            np.testing.assert_allclose(g, w_, rtol=1e-12, atol=1e-12)
        assert acc[1].sum() == float(off[-1])
    if rank == 0:
        assert t == 0.5 + (world - 1) and f == float(off[-1])
        for u in range(11):
            w = orc.decode(feats[int(off[u]):int(off[u + 1])])
            assert np.array_equal(gathered[u], w), u
        open(os.path.join(tmpdir, "ok"), "w").write("ok")
    else:
        assert gathered is None
    orc.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_gather(tmp_path, oracle_lib):
    from speechrecognition_amd import synth

    lex = synth.make_lexicon(6, 3, 1)
    mp_path = str(tmp_path / "m.mix")
    synth.write_mixset(mp_path, synth.make_mixset(lex.n_states, 2, 39, seed=4))
    mp.spawn(_worker, args=(2, _free_port(), mp_path, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_lpt_sharding_is_balanced_and_complete():
    from speechrecognition_amd import sharding, synth

    _, off = synth.make_batch(1000, 200, 400, 1, seed=7)
    for world in (1, 2, 4, 8):
        shards = sharding.shard_utterances(off, world)
        allu = np.sort(np.concatenate(shards))
        assert np.array_equal(allu, np.arange(1000))
        assert sharding.imbalance(off, world) < 1.002
    # ragged extreme: one giant utterance
    off2 = np.concatenate([[0], np.cumsum([5000] + [10] * 50)])
    shards = sharding.shard_utterances(off2, 4)
    assert sorted(len(s) for s in shards)[0] == 1  # the giant one sits alone
    feats = np.arange(int(off2[-1]), dtype=np.float32)[:, None]
    sub, sub_off = sharding.take_shard(feats, off2, shards[1])
    assert sub_off[-1] == len(sub)
