"""GPU parity tests at BASELINE.json's configuration sizes that round 1 left to bench.py: configs[2] through the DEFAULT
kernel pairing, configs[4]'s bigram leg at size, and global variance pooling.  All calls go through the C ABI."""
import hashlib

import numpy as np
import pytest

from speechrecognition_amd import capi, sharding, synth
from tests.util import Case

pytestmark = pytest.mark.gpu

TDP = (3.0, 0.0, 30.0)


def test_cfg3_default_pairing_prefilter_plus_fast_decoder(tmp_path, oracle_lib):
    """BASELINE.json configs[2]: 4000 tied states x 32 mixtures, P = 4000 trellis positions, >= 128 utterances so that the
    fast decoder runs in its throughput slot layout -- the pairing bench.py times (SR_GMM_PREFILTER + decode_fast_kernel).
    Words AND traceback must equal (a) the other exact pairing, SR_GMM_EXACT + the general kernel, on all 288 utterances
    and (b) the CPU oracle on 8 of them."""
    lex = synth.make_lexicon(1333, 3, 1)
    assert lex.n_states == 4000
    spec = synth.make_mixset(lex.n_states, 32, 39, seed=5)
    mp = str(tmp_path / "cfg3.mix")
    synth.write_mixset(mp, spec)
    feats, off = synth.make_batch(288, 200, 400, 39, seed=7)   # T_u ~ U{200..400}, seed 7
    word_off, automaton, sil_state = lex.flatten()
    with capi.Model.from_mixset(mp, 39) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, TDP, sil_state)
        assert lexh.describe() == "words 3 x 512 plain 3"  # 1333 plain words + silence: 22 groups of 64 on 8 waves, three per lane
        sl = synth.sietill_lexicon()                        # the reference's own lexicon (18-24 positions per word): the slot kernel
        swo, sau, ssil = sl.flatten()
        slh = m.lexicon(swo, sau, sl.silence_idx, TDP, ssil)
        assert slh.describe().startswith("slots (")
        slh.close()
        corpus = m.upload(feats, off)
        words, woff, (tbs, tbw, tbb) = corpus.recognize(lexh, 200.0, 10.0, capi.GMM_PREFILTER, traceback=True)
        w2, o2, (s2, ww2, b2) = corpus.recognize(lexh, 200.0, 10.0, capi.GMM_EXACT, traceback=True, general_kernel=True)
        # three-state words: the default search kernel is the word-per-lane one (viterbi_words.hip); the slot-per-lane kernel
        # (viterbi_fast.hip, P = 4000 in its 1024 x 4 layout) must agree as well
        w3, o3, (s3, ww3, b3) = corpus.recognize(lexh, 200.0, 10.0, capi.GMM_PREFILTER, traceback=True, slot_kernel=True)
        corpus.close()
        lexh.close()
    assert len(words) > 288
    assert np.array_equal(words, w2) and np.array_equal(woff, o2)
    assert np.array_equal(tbw, ww2) and np.array_equal(tbb, b2) and np.array_equal(tbs.view(np.uint64), s2.view(np.uint64))
    assert np.array_equal(words, w3) and np.array_equal(woff, o3)
    assert np.array_equal(tbw, ww3) and np.array_equal(tbb, b3) and np.array_equal(tbs.view(np.uint64), s3.view(np.uint64))
    o = oracle_lib.Oracle(mp, 39, lex, am_threshold=200.0)
    for u in (0, 1, 37, 100, 143, 200, 286, 287):
        x = feats[int(off[u]):int(off[u + 1])]
        w, (os_, ow, ob) = o.decode(x, traceback=True)
        assert np.array_equal(w, words[int(woff[u]):int(woff[u + 1])]), u
        a = int(off[u]) + u
        assert np.array_equal(tbw[a:a + len(x) + 1], ow) and np.array_equal(tbb[a:a + len(x) + 1], ob), u
        assert np.array_equal(tbs[a:a + len(x) + 1].view(np.uint64), os_.view(np.uint64)), u
    o.close()


def _cfg5_setup(tmp_path):
    lex = synth.make_lexicon(2666, 3, 1, extra_states_last=1)   # 2666 words x 3 states + one 4-state word
    assert lex.n_states == 8000 and lex.n_words == 2667
    spec = synth.make_mixset(lex.n_states, 64, 39, seed=29)
    mp = str(tmp_path / "cfg5.mix")
    synth.write_mixset(mp, spec)
    word_off, mixtures, _ = lex.flatten()
    W = lex.n_words
    rng = np.random.default_rng(31)
    p = rng.dirichlet(np.ones(W), size=W)                       # rows of a symmetric Dirichlet(1): p[h, w]
    lm = (-np.log(p)).T.astype(np.float32).copy()                # lm[w, h] = -log p(w | h)
    # rwth-asr example values (src/example-setup/config/recognition-triphones-lda-pruned.config:47-58), as bench.py uses
    tdp = np.array([[3.0, 0.0, 3.0, 150.0], [0.0001, 3.0, np.inf, 15.0]], np.float32)
    return lex, spec, mp, word_off, mixtures, lm, tdp


def test_cfg5_bigram_at_size_vs_restatement(tmp_path, oracle_lib):
    """BASELINE.json configs[4]'s bigram leg at its real size -- 8000 states x 64 mixtures (512 000 densities), 2667 words,
    dense Dirichlet bigram (28 MB), acoustic beam 200 -- against orc_bigram_decode (PARITY UNPINNED: the restatement of
    Teaching::LinearSearch is the specification, rwth-asr cannot be built here), three utterances of 30-60 frames; then a
    100-utterance batch through size-independent properties: permutation of the utterances and chunking of the score
    table (SRGPU chunk limit) must not change any utterance's traceback."""
    lex, spec, mp, word_off, mixtures, lm, tdp = _cfg5_setup(tmp_path)
    rng = np.random.default_rng(32)
    utts = [synth.make_features(int(n), 39, seed=40 + i) for i, n in enumerate((30, 47, 60))]
    # one of them drawn from the model, so that the acoustic beam really prunes
    utts[1] = synth.sample_utterance(spec, lex, rng.integers(1, lex.n_words, size=4), seed=33, frames_per_state=(2, 4))[:47]
    off = np.concatenate([[0], np.cumsum([len(x) for x in utts])]).astype(np.uint64)
    o = oracle_lib.Oracle(mp, 39, lex)
    want = []
    for x in utts:
        dense = o.score_matrix(x, n_threads=16)
        want.append(oracle_lib.bigram_decode(dense, word_off, mixtures, lex.silence_idx, lm, tdp, 200.0, capi.FLT_MAX))
    o.close()
    feats100, off100 = synth.make_batch(100, 30, 60, 39, seed=34)
    for u in range(0, 100, 4):  # every fourth utterance drawn from the model: several words in its traceback
        x = synth.sample_utterance(spec, lex, rng.integers(1, lex.n_words, size=5), seed=200 + u, frames_per_state=(2, 4))
        n = min(len(x), int(off100[u + 1] - off100[u]))
        feats100[int(off100[u]):int(off100[u]) + n] = x[:n]
    with capi.Model.from_mixset(mp, 39) as m:
        bg = m.bigram(word_off, mixtures, lex.silence_idx, lm, tdp)
        corpus = m.upload(np.concatenate(utts), off)
        gw, gs, gt, goff = corpus.recognize_bigram(bg, 200.0, capi.FLT_MAX)
        corpus.close()
        for u, (w, s, t) in enumerate(want):
            a, b = int(goff[u]), int(goff[u + 1])
            assert len(w) > 0
            assert np.array_equal(gw[a:b], w), u
            assert np.array_equal(gt[a:b], t), u
            assert np.array_equal(gs[a:b].view(np.uint32), s.view(np.uint32)), u
        # ---- 100 utterances: permutation invariance --------------------------------------------------------------
        c1 = m.upload(feats100, off100)
        w1, s1, t1, o1 = c1.recognize_bigram(bg, 200.0, capi.FLT_MAX)
        # (three-state words, one four-state word: the state hypotheses live in registers; the dense LDS layout must agree)
        wd, sd, td, od = c1.recognize_bigram(bg, 200.0, capi.FLT_MAX, dense_states=True)
        assert np.array_equal(w1, wd) and np.array_equal(t1, td) and np.array_equal(o1, od) and np.array_equal(s1.view(np.uint32), sd.view(np.uint32))
        c1.close()
        perm = np.random.default_rng(35).permutation(100)
        lens = np.diff(off100.astype(np.int64))
        pf = np.concatenate([feats100[int(off100[u]):int(off100[u + 1])] for u in perm])
        poff = np.concatenate([[0], np.cumsum(lens[perm])]).astype(np.uint64)
        c2 = m.upload(pf, poff)
        w2, s2, t2, o2 = c2.recognize_bigram(bg, 200.0, capi.FLT_MAX)
        c2.close()
        bg.close()
    # ---- chunking invariance: a 64 MiB score workspace holds 1000 frames of this model -> five chunks, two buffers ----
    import os
    os.environ["SRGPU_SCORE_CHUNK_MB"] = "64"
    try:
        with capi.Model.from_mixset(mp, 39) as m:
            bg = m.bigram(word_off, mixtures, lex.silence_idx, lm, tdp)
            c3 = m.upload(feats100, off100)
            w3, s3, t3, o3 = c3.recognize_bigram(bg, 200.0, capi.FLT_MAX)
            c3.close()
            bg.close()
    finally:
        del os.environ["SRGPU_SCORE_CHUNK_MB"]
    assert np.array_equal(w1, w3) and np.array_equal(t1, t3) and np.array_equal(o1, o3)
    assert np.array_equal(s1.view(np.uint32), s3.view(np.uint32))
    assert int(o1[-1]) > 100  # (a random-feature utterance ends in one item, a sampled one in several)
    for k, u in enumerate(perm):
        a, b, a2, b2 = int(o1[u]), int(o1[u + 1]), int(o2[k]), int(o2[k + 1])
        assert np.array_equal(w1[a:b], w2[a2:b2]) and np.array_equal(t1[a:b], t2[a2:b2]), u
        assert np.array_equal(s1[a:b].view(np.uint32), s2[a2:b2].view(np.uint32)), u


def test_cfg5_bigram_lm_beam_on_at_size(tmp_path, oracle_lib):
    """configs[4] with the LM beam ON (LinearSearch.cc:499-503: start hypotheses beyond best_start + lm-pruning are dropped;
    rounds 1-2 only ran it up to 2200 words): 8000 states x 64, 2667 words, acoustic beam 200, lm-pruning 4 and 9 -- against
    orc_bigram_decode on three utterances each (words, times, scores bit for bit).  PARITY UNPINNED: the restatement of
    Teaching::LinearSearch is the specification (rwth-asr cannot be built here); the test also checks that the beam bites,
    i.e. that the tight beam activates fewer start hypotheses than no beam."""
    lex, spec, mp, word_off, mixtures, lm, tdp = _cfg5_setup(tmp_path)
    rng = np.random.default_rng(52)
    utts = [synth.make_features(int(n), 39, seed=60 + i) for i, n in enumerate((28, 41, 55))]
    utts[2] = synth.sample_utterance(spec, lex, rng.integers(1, lex.n_words, size=5), seed=53, frames_per_state=(2, 4))[:55]
    off = np.concatenate([[0], np.cumsum([len(x) for x in utts])]).astype(np.uint64)
    o = oracle_lib.Oracle(mp, 39, lex)
    dense = [o.score_matrix(x, n_threads=16) for x in utts]
    o.close()
    starts = {}
    with capi.Model.from_mixset(mp, 39) as m:
        bg = m.bigram(word_off, mixtures, lex.silence_idx, lm, tdp)
        corpus = m.upload(np.concatenate(utts), off)
        for lm_beam in (4.0, 9.0, capi.FLT_MAX):
            gw, gs, gt, goff = corpus.recognize_bigram(bg, 200.0, lm_beam)
            n_started = 0
            for u, d in enumerate(dense):
                w, sc, t, st = oracle_lib.bigram_decode(d, word_off, mixtures, lex.silence_idx, lm, tdp, 200.0, lm_beam, stats=True)
                n_started += int(st[1])
                a, b = int(goff[u]), int(goff[u + 1])
                assert len(w) > 0
                assert np.array_equal(gw[a:b], w), (lm_beam, u)
                assert np.array_equal(gt[a:b], t), (lm_beam, u)
                assert np.array_equal(gs[a:b].view(np.uint32), sc.view(np.uint32)), (lm_beam, u)
            starts[lm_beam] = n_started
        corpus.close()
        bg.close()
    assert starts[4.0] < starts[9.0] < starts[capi.FLT_MAX], starts   # the LM beam prunes start hypotheses


def test_global_pooling_em_iteration_on_device(tmp_path):
    """pooling = 0 (MixtureModel::GLOBAL_POOLING, Mixtures.cpp:431-450) on the EM side, against what the REFERENCE wrote
    (tests/golden/global_pooling.npz): accumulators bit for bit, the MIXSET file byte for byte (sha256 of the reference's
    own file), and the scores of the model finalised from those statistics with global pooling."""
    c = Case("global_pooling", tmp_path)
    z = c.z
    dens_off = z["model_mix_off"].astype(np.uint32)
    flat = z["model_mix_dens"].astype(np.int64)
    dens_mean, dens_var = c.spec.dens_mean[flat], c.spec.dens_var[flat]
    with capi.Model.from_mixset(c.mixset_path, c.dim, capi.POOL_GLOBAL) as m:
        corpus = m.upload(c.feats, np.array([0, len(c.feats)], np.uint64))
        acc = corpus.accumulate(z["em_states"])
        corpus.close()
    keep = z["em_var_keep"]
    assert np.array_equal(acc[0].view(np.uint64), z["em_mean_acc"].view(np.uint64)) and np.array_equal(acc[1], z["em_mean_w"])
    assert np.array_equal(acc[2][keep].view(np.uint64), z["em_var_acc"].view(np.uint64)) and np.array_equal(acc[3][keep], z["em_var_w"])
    out = str(tmp_path / "em.mix")
    capi.mixset_write(out, c.dim, dens_off, dens_mean, dens_var, acc)
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == str(z["em_file_sha256"])
    for make in (lambda: capi.Model.from_statistics(c.dim, dens_off, dens_mean, dens_var, acc, pooling=capi.POOL_GLOBAL),
                 lambda: capi.Model.from_mixset(out, c.dim, capi.POOL_GLOBAL)):
        with make() as m2:
            for kernel in (capi.GMM_EXACT, capi.GMM_PREFILTER):
                got = m2.score_frames(c.feats[:32], kernel)
                assert np.array_equal(got.view(np.uint64), z["em_scores_after"].view(np.uint64))


@pytest.mark.parametrize("name", ["tied_variances", "mixture_pooling", "global_pooling", "nan_variance_floor", "sietill_lexicon_d25"])
def test_device_finalize_equals_host_finalize(name, tmp_path, monkeypatch):
    """MixtureModel::finalize on the device (em_finalize.hip, the default) against the host-only version of the same
    arithmetic (SRGPU_HOST_FINALIZE=1) and against the reference's golden scores: every pooling mode, tied variance rows
    (last writer wins), rows nobody finalises, NaN / non-positive variances -- bit for bit."""
    c = Case(name, tmp_path)
    with capi.Model.from_mixset(c.mixset_path, c.dim, c.pooling, c.max_approx) as m:
        dev = m.score_frames(c.feats, capi.GMM_EXACT)
        topo = m.topology()
    monkeypatch.setenv("SRGPU_HOST_FINALIZE", "1")
    with capi.Model.from_mixset(c.mixset_path, c.dim, c.pooling, c.max_approx) as m:
        host = m.score_frames(c.feats, capi.GMM_EXACT)
    monkeypatch.delenv("SRGPU_HOST_FINALIZE")
    assert np.array_equal(dev.view(np.uint64), host.view(np.uint64))
    if c.max_approx:
        c.check_scores(dev, exact=True)
    # the same statistics through sr_model_create_from_statistics (the EM path)
    acc = (c.spec.mean_acc, c.spec.mean_w, c.spec.var_acc, c.spec.var_w)
    with capi.Model.from_statistics(c.dim, topo[0], topo[1], topo[2], acc, pooling=c.pooling, max_approx=c.max_approx) as m2:
        assert np.array_equal(m2.score_frames(c.feats, capi.GMM_EXACT).view(np.uint64), dev.view(np.uint64))
        assert np.array_equal(m2.score_frames(c.feats, capi.GMM_PREFILTER).view(np.uint64), dev.view(np.uint64))


def test_cfg4_ten_thousand_utterances_eight_replicas(tmp_path, oracle_lib):
    """BASELINE configs[3]: 4000 states x 32, ONE batch of 10 000 utterances (U{200..400} frames, 3.0 M frames) sharded eight
    ways -- sr_recognize_batch_multi with eight (model, lexicon) replicas, all on device 0 here, one per GPU in production
    (Recognizer.cpp:43-56: the utterance loop, parallel over segments).  Against the single-handle path (three score chunks)
    on every word, against sr_shard_utterances on the deal, and against the CPU oracle on sampled utterances including the
    longest, the shortest and the last utterance of a shard."""
    lex = synth.make_lexicon(1333, 3, 1)
    spec = synth.make_mixset(lex.n_states, 32, 39, seed=23)
    mp = str(tmp_path / "cfg4.mix")
    synth.write_mixset(mp, spec)
    feats, off = synth.make_batch(10000, 200, 400, 39, seed=7)
    assert int(off[-1]) > 2_900_000
    word_off, automaton, sil = lex.flatten()
    R = 8
    models = [capi.Model.from_mixset(mp, 39) for _ in range(R)]
    lexica = [m.lexicon(word_off, automaton, lex.silence_idx, TDP, sil) for m in models]
    try:
        w8, o8, load = capi.recognize_batch_multi(models, lexica, feats, off, 200.0, 10.0)
        shard_of, want_load = capi.shard_utterances(off, R)
        assert np.array_equal(load, want_load) and int(load.sum()) == int(off[-1])
        assert load.max() / load.mean() < 1.002
        assert np.bincount(shard_of, minlength=R).min() >= 1200   # ~1250 utterances per shard
        w1, o1 = models[0].recognize_batch(lexica[0], feats, off, 200.0, 10.0)
        assert np.array_equal(o8, o1) and np.array_equal(w8, w1) and len(w1) > 500_000
    finally:
        for l in lexica:
            l.close()
        for m in models:
            m.close()
    lens = np.diff(off.astype(np.int64))
    rng = np.random.default_rng(1)
    sample = {int(np.argmax(lens)), int(np.argmin(lens)), 0, 9999, int(np.flatnonzero(shard_of == 0)[-1]),
              int(np.flatnonzero(shard_of == R - 1)[-1])} | {int(u) for u in rng.integers(0, 10000, size=4)}
    sample = sorted(sample)
    sub, sub_off = sharding.take_shard(feats, off, sample)
    o = oracle_lib.Oracle(mp, 39, lex, tdp=TDP, am_threshold=200.0, word_penalty=10.0)
    ow, ooff, _ = o.recognize_batch(sub, sub_off, n_threads=min(16, len(sample)))
    o.close()
    for i, u in enumerate(sample):
        assert np.array_equal(ow[int(ooff[i]):int(ooff[i + 1])], w8[int(o8[u]):int(o8[u + 1])]), f"utterance {u}"
