"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI
(speechrecognition_amd/capi.py -> libsrgpu.so), against the golden vectors the real reference
produced and against the CPU oracle on the same seeded inputs.

Tolerances: integer/index results (words, state sequences, back pointers) bit-exact; SR_GMM_EXACT
scores bit-exact; SR_GMM_MFMA scores within 1e-9 relative (north_star asks 1e-4; the FP64 GEMM form
is good to ~1e-13)."""
import numpy as np
import pytest

from speechrecognition_amd import capi, synth
from tests.util import Case, golden_names

pytestmark = pytest.mark.gpu

MFMA_RTOL = 1e-9


def _assert_scores_close(got, want, rtol=MFMA_RTOL):
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    denom = np.maximum(np.abs(want[fin]), 1.0)
    err = np.abs(got[fin] - want[fin]) / denom
    assert err.max(initial=0.0) <= rtol, f"max rel err {err.max():.3e}"


def _lex_handle(model, c):
    word_off, automaton, sil_state = c.lex.flatten()
    return model.lexicon(word_off, automaton, c.lex.silence_idx, c.tdp, sil_state), sil_state


@pytest.mark.parametrize("name", golden_names())
def test_golden_scores(name, tmp_path):
    c = Case(name, tmp_path)
    with capi.Model.from_mixset(c.mixset_path, c.dim, c.pooling, c.max_approx) as m:
        exact = m.score_frames(c.feats, capi.GMM_EXACT)
        mfma = m.score_frames(c.feats, capi.GMM_MFMA)
        pref = m.score_frames(c.feats, capi.GMM_PREFILTER)
    # bf16 prefilter + FP64 refinement: the same bits as the exact kernel, whatever the model (NaN variances, empty
    # mixtures, sum mode -> falls through to the exact kernel)
    assert np.array_equal(pref.view(np.uint64), exact.view(np.uint64))
    if c.max_approx:
        c.check_scores(exact, exact=True)  # bit-identical to MixtureModel::score
    else:
        c.check_scores(exact, exact=False, rtol=1e-12)  # sum mode: device exp/log
    # SR_GMM_DEFAULT: the prefilter path for max-approx models (those bits), the FP64-MFMA kernel for sum scoring (srgpu.h)
    with capi.Model.from_mixset(c.mixset_path, c.dim, c.pooling, c.max_approx) as m:
        dflt = m.score_frames(c.feats, capi.GMM_DEFAULT)
    assert np.array_equal(dflt.view(np.uint64), (pref if c.max_approx else mfma).view(np.uint64))
    want = c.z["scores"] if "scores" in c.z else None
    if want is not None:
        _assert_scores_close(mfma, want)
    else:
        _assert_scores_close(mfma.reshape(-1)[c.z["score_idx"]], c.z["score_val"])


EXACT_KERNELS = (capi.GMM_EXACT, capi.GMM_PREFILTER, capi.GMM_DEFAULT)  # (DEFAULT: bit-exact for max-approx models; every use below also asks c.max_approx)


@pytest.mark.parametrize("kernel", [capi.GMM_EXACT, capi.GMM_PREFILTER, capi.GMM_MFMA, capi.GMM_DEFAULT])
@pytest.mark.parametrize("name", golden_names())
def test_golden_decode_and_align(name, kernel, tmp_path, oracle_lib):
    c = Case(name, tmp_path)
    T = c.feats.shape[0]
    off = np.array([0, T], dtype=np.uint64)
    with capi.Model.from_mixset(c.mixset_path, c.dim, c.pooling, c.max_approx) as m:
        lex, sil_state = _lex_handle(m, c)
        corpus = m.upload(c.feats, off)
        words, woff, (tbs, tbw, tbb) = corpus.recognize(lex, c.beam, c.wp, kernel, traceback=True)
        assert np.array_equal(words, c.z["words"]), (words, c.z["words"])
        assert woff[-1] == len(c.z["words"])
        # traceback array against the oracle's (scores bit-exact only with the exact kernel)
        o = c.oracle(oracle_lib)
        _, (os_, ow, ob) = o.decode(c.feats, traceback=True)
        o.close()
        assert np.array_equal(tbw, ow) and np.array_equal(tbb, ob)
        if kernel in EXACT_KERNELS and c.max_approx:
            assert np.array_equal(tbs.view(np.uint64), os_.view(np.uint64))
        else:
            _assert_scores_close(tbs, os_)
        if "align_ref" in c.z:
            st, cost = corpus.align([c.z["align_ref"]], c.tdp, sil_state, kernel)
            assert np.array_equal(st, c.z["align_full_states"])
            if kernel in EXACT_KERNELS and c.max_approx:
                assert cost[0] == float(c.z["align_full_cost"])
            else:
                assert abs(cost[0] - float(c.z["align_full_cost"])) <= MFMA_RTOL * max(1.0, abs(float(c.z["align_full_cost"])))
            i = 0
            while f"align_pruned_thr{i}" in c.z:
                st, cost = corpus.align([c.z["align_ref"]], c.tdp, sil_state, kernel,
                                        pruning_threshold=float(c.z[f"align_pruned_thr{i}"]))
                assert np.array_equal(st, c.z[f"align_pruned_states{i}"])
                want = float(c.z[f"align_pruned_cost{i}"])
                if kernel in EXACT_KERNELS and c.max_approx:
                    assert cost[0] == want
                else:
                    assert abs(cost[0] - want) <= MFMA_RTOL * max(1.0, abs(want))
                i += 1
        corpus.close()
        lex.close()


def _random_setup(tmp_path, seed, W, spw, reps, M, D, var_floor=0.5):
    lex = synth.make_lexicon(W, spw, reps)
    rng = np.random.default_rng(seed)
    nm = M if np.isscalar(M) else rng.integers(M[0], M[1] + 1, size=lex.n_states)
    spec = synth.make_mixset(lex.n_states, nm, D, seed=seed, var_floor=var_floor)
    mp = str(tmp_path / f"m{seed}.mix")
    synth.write_mixset(mp, spec)
    return lex, spec, mp


@pytest.mark.parametrize("seed,W,spw,reps,M,D,beam", [
    (201, 40, 3, 1, (1, 6), 39, 200.0),    # ragged mixture sizes -> padded 4-state groups
    (202, 25, 4, 2, 3, 25, 40.0),          # D = 25 (KSTEPS 14), repetitions 2
    (203, 10, 3, 1, 5, 12, 30.0),          # D = 12 (KSTEPS 8)
    (204, 15, 3, 1, 2, 50, 100.0),         # D = 50 (KSTEPS 32, generic-dim exact kernel)
    (205, 200, 3, 1, 4, 39, 150.0),        # P = 601 -> 4 slots per thread
    (206, 700, 3, 1, 2, 39, 200.0),        # P = 2101 -> 1024 threads x 4
])
def test_batch_vs_oracle(tmp_path, oracle_lib, seed, W, spw, reps, M, D, beam):
    """A ragged batch (including a 1-frame utterance) through score + decode + align, vs the oracle."""
    lex, spec, mp = _random_setup(tmp_path, seed, W, spw, reps, M, D)
    o = oracle_lib.Oracle(mp, D, lex, am_threshold=beam)
    rng = np.random.default_rng(seed + 1)
    lens = [1, 2] + list(rng.integers(30, 90, size=6))
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    feats = rng.standard_normal((int(off[-1]), D)).astype(np.float32)
    # two utterances drawn from the model so that the beam prunes like on speech
    u4 = synth.sample_utterance(spec, lex, rng.integers(1, lex.n_words, size=3), seed=seed + 2)[: lens[4]]
    feats[int(off[4]):int(off[4]) + len(u4)] = u4
    word_off, automaton, sil_state = lex.flatten()
    with capi.Model.from_mixset(mp, D) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil_state)
        corpus = m.upload(feats, off)
        want = o.score_matrix(feats)
        got_exact = corpus.score(capi.GMM_EXACT)
        assert np.array_equal(got_exact.view(np.uint64), want.view(np.uint64))
        assert np.array_equal(corpus.score(capi.GMM_PREFILTER).view(np.uint64), want.view(np.uint64))
        _assert_scores_close(corpus.score(capi.GMM_MFMA), want)
        for kernel in (capi.GMM_EXACT, capi.GMM_PREFILTER, capi.GMM_MFMA):
            words, woff = corpus.recognize(lexh, beam, 10.0, kernel)
            for u in range(len(lens)):
                w = o.decode(feats[int(off[u]):int(off[u + 1])])
                assert np.array_equal(w, words[int(woff[u]):int(woff[u + 1])]), (u, kernel)
        # aligner: utterances long enough for `sil w sil w sil`
        auts, keep = [], []
        for u in range(len(lens)):
            ws = rng.integers(1, lex.n_words, size=2)
            a = [sil_state]
            for w in ws:
                a += list(automaton[word_off[w]:word_off[w + 1]]) + [sil_state]
            if len(a) > lens[u]:
                a = [sil_state]
            auts.append(np.asarray(a, dtype=np.uint16))
        st, cost = corpus.align(auts, (3.0, 0.0, 30.0), sil_state, capi.GMM_EXACT)
        stp, costp = corpus.align(auts, (3.0, 0.0, 30.0), sil_state, capi.GMM_EXACT, pruning_threshold=25.0)
        for u in range(len(lens)):
            f = feats[int(off[u]):int(off[u + 1])]
            ws, wc = o.align_full(f, auts[u])
            assert np.array_equal(st[int(off[u]):int(off[u + 1])], ws) and (cost[u] == wc or (np.isinf(wc) and np.isinf(cost[u])))
            ws, wc = o.align_pruned(f, auts[u], 25.0)
            assert np.array_equal(stp[int(off[u]):int(off[u + 1])], ws) and costp[u] == wc
        corpus.close()
        lexh.close()
    o.close()


def test_negative_emission_costs_take_the_sequential_path(tmp_path, oracle_lib):
    """Tight variances make emission costs negative, where the reference's pre-AM early-out
    (Recognizer.cpp:143,173) is live: the decoder must replay it, not shortcut it."""
    lex = synth.make_lexicon(30, 3, 2)
    spec = synth.make_mixset(lex.n_states, 2, 39, seed=301, var_floor=0.002)
    mu = spec.mean_acc / spec.mean_w[:, None]
    var = 0.004 * (spec.var_acc / spec.var_w[:, None] - mu ** 2)
    spec.var_acc = (var + mu ** 2) * spec.var_w[:, None]
    mp = str(tmp_path / "neg.mix")
    synth.write_mixset(mp, spec)
    rng = np.random.default_rng(302)
    utts = [synth.sample_utterance(spec, lex, rng.integers(1, lex.n_words, size=4), seed=310 + i, frames_per_state=(1, 3), noise=0.8)
            for i in range(6)]
    off = np.concatenate([[0], np.cumsum([len(x) for x in utts])]).astype(np.uint64)
    feats = np.concatenate(utts)
    word_off, automaton, sil_state = lex.flatten()
    for beam, wp in ((150.0, 2.0), (40.0, 0.0)):
        o = oracle_lib.Oracle(mp, 39, lex, am_threshold=beam, word_penalty=wp)
        assert (o.score_matrix(feats) < 0).any()
        with capi.Model.from_mixset(mp, 39) as m:
            lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil_state)
            corpus = m.upload(feats, off)
            words, woff, (tbs, tbw, tbb) = corpus.recognize(lexh, beam, wp, capi.GMM_EXACT, traceback=True)
            for u in range(len(utts)):
                w, (os_, ow, ob) = o.decode(utts[u], traceback=True)
                assert np.array_equal(w, words[int(woff[u]):int(woff[u + 1])])
                b = int(off[u]) + u
                assert np.array_equal(tbs[b:b + len(os_)].view(np.uint64), os_.view(np.uint64))
                assert np.array_equal(tbw[b:b + len(ow)], ow) and np.array_equal(tbb[b:b + len(ob)], ob)
            corpus.close()
            lexh.close()
        o.close()


@pytest.mark.parametrize("seed,W,spw,reps,M,D,var_scale,beam,wp,ragged", [
    (321, 30, 3, 1, 2, 39, 0.004, 150.0, 2.0, False),    # the set-up of the test above on three-position words: the word-per-lane kernel
    (322, 30, 3, 1, 2, 39, 0.004, 40.0, 0.0, False),
    (323, 200, 2, 1, 3, 4, 0.05, 1e9, 10.0, False),      # low dimension, no pruning: every word end alive, many negative entry costs
    (324, 120, 4, 1, 2, 6, 0.05, 60.0, 5.0, False),      # four-position words
    (325, 64, 1, 2, 2, 4, 0.05, 1e9, 3.0, False),        # one state twice: position 1 is the word end, loop-free
    (326, 90, 0, 0, 3, 5, 0.05, 200.0, 10.0, True),      # ragged short lexicon: silence anywhere, one-position words, words that begin in silence
    (327, 150, 0, 0, 2, 4, 0.03, 1e9, 0.0, True),
    (328, 1400, 3, 1, 1, 4, 0.05, 80.0, 10.0, False),    # three words per lane
    (329, 2900, 3, 1, 1, 4, 0.05, 30.0, 10.0, False),    # more than 8192 positions: the word-per-lane kernel of a "big" lexicon
])
def test_negative_emission_costs_on_short_word_lexica(tmp_path, oracle_lib, seed, W, spw, reps, M, D, var_scale, beam, wp, ragged):
    """Round 5: models whose emission costs can be negative are searched by the word-per-lane kernel's NEG variant (viterbi_words.hip),
    which replays the reference's pre-AM early-out (Recognizer.cpp:143,173) frame by frame -- the boundary loop only for the words
    whose entry costs are negative -- instead of handing the whole utterance to the general kernel.  Words and the traceback arrays
    must be the oracle's, bit for bit."""
    rng = np.random.default_rng(seed)
    lex = synth.make_ragged_lexicon(W, rng, short=True) if ragged else synth.make_lexicon(W, spw, reps)
    spec = synth.make_mixset(lex.n_states, M, D, seed=seed, var_floor=0.05)
    synth.scale_variances(spec, var_scale)
    mp = str(tmp_path / "negw.mix")
    synth.write_mixset(mp, spec)
    speech = [w for w in range(lex.n_words) if w != lex.silence_idx]
    n_utts = 4 if W >= 1000 else 8
    utts = []
    for i in range(n_utts):
        x = synth.sample_utterance(spec, lex, rng.choice(speech, size=3), seed=seed * 10 + i, frames_per_state=(1, 3), noise=0.9)[:60]
        if i % 3 == 2:  # frames near NO mean in between: every cost large and positive for a while
            x[len(x) // 2: len(x) // 2 + 4] += 3.0
        utts.append(x.astype(np.float32))
    off = np.concatenate([[0], np.cumsum([len(x) for x in utts])]).astype(np.uint64)
    feats = np.concatenate(utts)
    word_off, automaton, sil_state = lex.flatten()
    o = oracle_lib.Oracle(mp, D, lex, am_threshold=beam, word_penalty=wp)
    sm = o.score_matrix(feats)
    assert (sm < 0).mean() > 1e-4, "the case has no negative emission costs"
    with capi.Model.from_mixset(mp, D) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil_state)
        corpus = m.upload(feats, off)
        words, woff, (tbs, tbw, tbb) = corpus.recognize(lexh, beam, wp, capi.GMM_EXACT, traceback=True)
        for u in range(len(utts)):
            w, (os_, ow, ob) = o.decode(utts[u], traceback=True)
            assert np.array_equal(w, words[int(woff[u]):int(woff[u + 1])]), (u, w, words[int(woff[u]):int(woff[u + 1])])
            b = int(off[u]) + u
            assert np.array_equal(tbs[b:b + len(os_)].view(np.uint64), os_.view(np.uint64)), u
            assert np.array_equal(tbw[b:b + len(ow)], ow) and np.array_equal(tbb[b:b + len(ob)], ob), u
        if max(np.diff(word_off.astype(np.int64))) <= 4 and lex.n_states * 8 <= 60000:
            # the slot-per-lane kernel + general replay (rounds 1-4's route) must agree
            w_s, o_s, (s_s, tw_s, tb_s) = corpus.recognize(lexh, beam, wp, capi.GMM_EXACT, traceback=True, slot_kernel=True)
            assert np.array_equal(w_s, words) and np.array_equal(o_s, woff) and np.array_equal(tw_s, tbw) and np.array_equal(tb_s, tbb)
            assert np.array_equal(s_s.view(np.uint64), tbs.view(np.uint64))
        corpus.close()
        lexh.close()
    o.close()


def test_chunked_pipeline_matches_single_chunk(tmp_path, oracle_lib, monkeypatch):
    """A tiny score-workspace budget forces many chunks through the two-stream pipeline."""
    lex, spec, mp = _random_setup(tmp_path, 401, 60, 3, 1, 4, 39)
    feats, off = synth.make_batch(40, 20, 60, 39, seed=402)
    word_off, automaton, sil_state = lex.flatten()
    results = []
    for mb in ("4096", "1"):
        monkeypatch.setenv("SRGPU_SCORE_CHUNK_MB", mb)
        with capi.Model.from_mixset(mp, 39) as m:
            lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil_state)
            corpus = m.upload(feats, off)
            results.append((corpus.recognize(lexh, 200.0, 10.0, capi.GMM_MFMA), corpus.score(capi.GMM_MFMA)))
            corpus.close()
            lexh.close()
    (w0, o0), s0 = results[0]
    (w1, o1), s1 = results[1]
    assert np.array_equal(w0, w1) and np.array_equal(o0, o1) and np.array_equal(s0.view(np.uint64), s1.view(np.uint64))
    o = oracle_lib.Oracle(mp, 39, lex, am_threshold=200.0)
    for u in (0, 17, 39):
        assert np.array_equal(o.decode(feats[int(off[u]):int(off[u + 1])]), w0[int(o0[u]):int(o0[u + 1])])
    o.close()


def test_full_size_properties(tmp_path):
    """BASELINE.json configs[2] model size (4000 states x 32 mixtures = 128000 densities, beyond what
    the reference's 16-bit density index can load): size-independent properties instead of an oracle run.
      * MFMA scores == EXACT scores to 1e-9 relative on a frame sample,
      * permuting utterances permutes results (independence / sharding invariance),
      * decoding the concatenated batch equals decoding each half (chunk invariance)."""
    lex = synth.make_lexicon(1333, 3, 1)
    spec = synth.make_mixset(lex.n_states, 32, 39, seed=23)
    mp = str(tmp_path / "big.mix")
    synth.write_mixset(mp, spec)
    feats, off = synth.make_batch(24, 200, 400, 39, seed=7)
    word_off, automaton, sil_state = lex.flatten()
    with capi.Model.from_mixset(mp, 39) as m:
        assert m.n_densities == 128000
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil_state)
        sample = feats[:300]
        _assert_scores_close(m.score_frames(sample, capi.GMM_MFMA), m.score_frames(sample, capi.GMM_EXACT))
        corpus = m.upload(feats, off)
        words, woff = corpus.recognize(lexh, 200.0, 10.0, capi.GMM_MFMA)
        wexact, woff_exact = corpus.recognize(lexh, 200.0, 10.0, capi.GMM_EXACT)
        assert np.array_equal(words, wexact) and np.array_equal(woff, woff_exact)
        corpus.close()
        per_utt = [words[int(woff[u]):int(woff[u + 1])] for u in range(24)]
        assert sum(len(x) for x in per_utt) > 0
        perm = np.random.default_rng(5).permutation(24)
        pf = np.concatenate([feats[int(off[u]):int(off[u + 1])] for u in perm])
        poff = np.concatenate([[0], np.cumsum([int(off[u + 1] - off[u]) for u in perm])]).astype(np.uint64)
        c2 = m.upload(pf, poff)
        w2, o2 = c2.recognize(lexh, 200.0, 10.0, capi.GMM_MFMA)
        c2.close()
        for i, u in enumerate(perm):
            assert np.array_equal(w2[int(o2[i]):int(o2[i + 1])], per_utt[u])
        lexh.close()


def test_error_paths():
    with pytest.raises(capi.SrError) as e:
        capi.Model.from_mixset("/nonexistent.mix", 39)
    assert e.value.code == -1 and "cannot open" in str(e.value)
    means = np.zeros((2, 161))
    with pytest.raises(capi.SrError) as e:
        capi.Model.from_tables([0, 1, 2], means, means + 1, np.zeros(2), np.zeros(2))
    assert e.value.code == -4  # dim > 160 (round 5: 64 .. 160 run the exact kernel, test_dimensions_beyond_the_matrix_core_kernels)


def test_cfg2_single_long_utterance(tmp_path, oracle_lib):
    """BASELINE.json configs[1] at full size: 1000 tied states x 8 mixtures, ONE 10 000-frame utterance (the
    latency-bound case: a single workgroup walks 10k frames).  Direct comparison with the oracle."""
    lex = synth.make_lexicon(333, 3, 1)
    spec = synth.make_mixset(lex.n_states, 8, 39, seed=23)
    mp = str(tmp_path / "cfg2.mix")
    synth.write_mixset(mp, spec)
    feats = synth.make_features(10000, 39, seed=24)
    off = np.array([0, 10000], dtype=np.uint64)
    word_off, automaton, sil_state = lex.flatten()
    o = oracle_lib.Oracle(mp, 39, lex, am_threshold=200.0)
    want_words, (ws, ww, wb) = o.decode(feats, traceback=True)
    with capi.Model.from_mixset(mp, 39) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil_state)
        corpus = m.upload(feats, off)
        words, woff, (tbs, tbw, tbb) = corpus.recognize(lexh, 200.0, 10.0, capi.GMM_EXACT, traceback=True)
        assert np.array_equal(words, want_words)
        assert np.array_equal(tbw, ww) and np.array_equal(tbb, wb) and np.array_equal(tbs.view(np.uint64), ws.view(np.uint64))
        words, woff = corpus.recognize(lexh, 200.0, 10.0, capi.GMM_MFMA)
        assert np.array_equal(words, want_words)
        # the DEFAULT scorer on this geometry: the 8-density-slot refinement variant (gmm_refine_kernel<39, 8, 8, 1>) over one
        # long utterance, scores against the exact kernel bit for bit, words + traceback against the oracle
        pref = corpus.score(capi.GMM_PREFILTER)
        assert np.array_equal(pref.view(np.uint64), corpus.score(capi.GMM_EXACT).view(np.uint64))
        words, woff, (tbs, tbw, tbb) = corpus.recognize(lexh, 200.0, 10.0, capi.GMM_DEFAULT, traceback=True)
        assert np.array_equal(words, want_words)
        assert np.array_equal(tbw, ww) and np.array_equal(tbb, wb) and np.array_equal(tbs.view(np.uint64), ws.view(np.uint64))
        # aligner over the long utterance: 40 words
        rng = np.random.default_rng(25)
        aut = [sil_state]
        for w in rng.integers(1, lex.n_words, size=40):
            aut += list(automaton[word_off[w]:word_off[w + 1]]) + [sil_state]
        aut = np.asarray(aut, dtype=np.uint16)
        dense = o.score_matrix(feats, n_threads=8)
        st, cost = corpus.align([aut], (3.0, 0.0, 30.0), sil_state, capi.GMM_EXACT)
        wst, wcost = o.align_full(feats, aut, dense=dense)
        assert np.array_equal(st, wst) and cost[0] == wcost
        st, cost = corpus.align([aut], (3.0, 0.0, 30.0), sil_state, capi.GMM_EXACT, pruning_threshold=60.0)
        wst, wcost = o.align_pruned(feats, aut, 60.0, dense=dense)
        assert np.array_equal(st, wst) and cost[0] == wcost
        corpus.close()
        lexh.close()
    o.close()


def test_cfg5_model_size(tmp_path, oracle_lib):
    """BASELINE.json configs[4] model size: 8000 states x 64 mixtures (512 000 densities), 2666 three-state words
    plus one four-state word -> 8003 trellis positions (the 1024 x 8 decoder geometry).  A short utterance
    directly against the oracle, a longer batch through exact-vs-MFMA equality."""
    lex = synth.make_lexicon(2666, 3, 1, extra_states_last=1)
    assert lex.n_states == 8000
    spec = synth.make_mixset(lex.n_states, 64, 39, seed=29)
    mp = str(tmp_path / "cfg5.mix")
    synth.write_mixset(mp, spec)
    word_off, automaton, sil_state = lex.flatten()
    short = synth.make_features(24, 39, seed=30)
    o = oracle_lib.Oracle(mp, 39, lex, am_threshold=200.0)
    want_scores = o.score_matrix(short, n_threads=8)
    want_words = o.decode(short, dense=want_scores)
    o.close()
    feats, off = synth.make_batch(6, 120, 200, 39, seed=31)
    with capi.Model.from_mixset(mp, 39) as m:
        assert m.n_densities == 512000
        got = m.score_frames(short, capi.GMM_EXACT)
        assert np.array_equal(got.view(np.uint64), want_scores.view(np.uint64))
        _assert_scores_close(m.score_frames(short, capi.GMM_MFMA), want_scores)
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil_state)
        c1 = m.upload(short, np.array([0, len(short)], dtype=np.uint64))
        w, _ = c1.recognize(lexh, 200.0, 10.0, capi.GMM_EXACT)
        assert np.array_equal(w, want_words)
        c1.close()
        corpus = m.upload(feats, off)
        wa, oa = corpus.recognize(lexh, 200.0, 10.0, capi.GMM_MFMA)
        wb, ob = corpus.recognize(lexh, 200.0, 10.0, capi.GMM_EXACT)
        assert np.array_equal(wa, wb) and np.array_equal(oa, ob) and len(wa) > 0
        # round 4: the two-chunk refinement (one evaluation per state, list batches through atomic minima) at this size: every
        # score of the batch against the exact kernel, bit for bit, and the short utterance against the oracle
        assert np.array_equal(m.score_frames(short, capi.GMM_PREFILTER).view(np.uint64), want_scores.view(np.uint64))
        assert np.array_equal(corpus.score(capi.GMM_PREFILTER).view(np.uint64), corpus.score(capi.GMM_EXACT).view(np.uint64))
        wc, oc = corpus.recognize(lexh, 200.0, 10.0, capi.GMM_DEFAULT)
        assert np.array_equal(wc, wb) and np.array_equal(oc, ob)
        corpus.close()
        lexh.close()


def test_path_scores_are_calc_am_score_summands(tmp_path, oracle_lib):
    """sr_path_scores_corpus: MixtureModel::score along an alignment (Trainer::calc_am_score, Training.cpp:605)."""
    lex, spec, mp = _random_setup(tmp_path, 501, 30, 3, 2, (1, 5), 39)
    feats, off = synth.make_batch(7, 20, 50, 39, seed=502)
    o = oracle_lib.Oracle(mp, 39, lex)
    rng = np.random.default_rng(503)
    states = rng.integers(0, lex.n_states, size=len(feats)).astype(np.uint16)
    dense = o.score_matrix(feats)
    want = dense[np.arange(len(feats)), states]
    with capi.Model.from_mixset(mp, 39) as m:
        corpus = m.upload(feats, off)
        got = corpus.path_scores(states, capi.GMM_EXACT)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
        _assert_scores_close(corpus.path_scores(states, capi.GMM_MFMA), want)
        with pytest.raises(capi.SrError):
            corpus.path_scores(np.full(len(feats), lex.n_states, dtype=np.uint16))
        corpus.close()
    o.close()


@pytest.mark.parametrize("tag,first_pass,max_approx", [("max", False, True), ("first", True, True), ("soft", False, False)])
def test_em_accumulate_golden(tag, first_pass, max_approx, tmp_path):
    """sr_accumulate_corpus against the REFERENCE's accumulators (tests/golden/accumulate.npz: tied variances, ragged
    mixtures): bit-identical in max-approx and first-pass mode, 1e-12 with soft memberships (device exp)."""
    from tests.test_oracle_golden import _accumulate_case
    z, lex, spec, mp = _accumulate_case(tmp_path)
    off = np.array([0, 200, 500], dtype=np.uint64)  # two "utterances"; accumulation ignores the boundary
    with capi.Model.from_mixset(mp, 39, capi.POOL_NONE, max_approx) as m:
        corpus = m.upload(z["feats"], off)
        a, w, v, vw = corpus.accumulate(z["states"], first_pass=first_pass, max_approx=max_approx)
        corpus.close()
    keep = z["var_keep"]
    if tag == "soft":
        np.testing.assert_allclose(a, z[f"{tag}_mean_acc"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(w, z[f"{tag}_mean_w"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(v[keep], z[f"{tag}_var_acc"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(vw[keep], z[f"{tag}_var_w"], rtol=1e-12, atol=1e-12)
    else:
        assert np.array_equal(a.view(np.uint64), z[f"{tag}_mean_acc"].view(np.uint64)) and np.array_equal(w, z[f"{tag}_mean_w"])
        assert np.array_equal(v[keep].view(np.uint64), z[f"{tag}_var_acc"].view(np.uint64)) and np.array_equal(vw[keep], z[f"{tag}_var_w"])
        unused = np.setdiff1d(np.arange(len(vw)), keep)
        assert np.all(vw[unused] == 0) and np.all(v[unused] == 1e-4)  # untouched rows = reset_accumulators()


def test_em_accumulate_after_alignment_vs_oracle(tmp_path, oracle_lib):
    """The training inner loop on the device: pruned re-alignment of a batch, then max-approx accumulation of that
    alignment, against the oracle doing the same frame by frame."""
    lex, spec, mp = _random_setup(tmp_path, 601, 25, 3, 1, (1, 6), 39)
    rng = np.random.default_rng(602)
    utts, auts = [], []
    word_off, automaton, sil_state = lex.flatten()
    for i in range(12):
        ws = rng.integers(1, lex.n_words, size=3)
        utts.append(synth.sample_utterance(spec, lex, ws, seed=610 + i))
        a = [sil_state]
        for w in ws:
            a += list(automaton[word_off[w]:word_off[w + 1]]) + [sil_state]
        auts.append(np.asarray(a, dtype=np.uint16))
    off = np.concatenate([[0], np.cumsum([len(u) for u in utts])]).astype(np.uint64)
    feats = np.concatenate(utts)
    o = oracle_lib.Oracle(mp, 39, lex)
    with capi.Model.from_mixset(mp, 39) as m:
        corpus = m.upload(feats, off)
        states, _ = corpus.align(auts, (3.0, 0.0, 30.0), sil_state, capi.GMM_EXACT, pruning_threshold=50.0)
        want_states = np.concatenate([o.align_pruned(u, a, 50.0)[0] for u, a in zip(utts, auts)])
        assert np.array_equal(states, want_states)
        got = corpus.accumulate(states)
        want = o.accumulate(feats, want_states)
        for g, wv in zip(got, want):
            assert np.array_equal(g.view(np.uint64), wv.view(np.uint64))
        assert got[1].sum() == len(feats)
        corpus.close()
    o.close()


def test_em_iteration_matches_reference_model_file(tmp_path, oracle_lib):
    """One EM iteration on the device -- accumulate an alignment, write the statistics as a MIXSET file
    (MixtureModel::write), finalise them into a new device model (MixtureModel::finalize) -- against the
    REFERENCE's accumulate + write golden: the files must be byte-identical and the updated model's scores
    bit-identical to what the oracle computes after loading that file."""
    from tests.test_oracle_golden import _accumulate_case
    z, lex, spec, mp = _accumulate_case(tmp_path)
    dens_off = np.concatenate([[0], np.cumsum([len(m) for m in spec.mixtures])]).astype(np.uint32)
    flat = np.asarray([d for m in spec.mixtures for d in m], dtype=np.int64)
    dens_mean, dens_var = spec.dens_mean[flat], spec.dens_var[flat]
    with capi.Model.from_mixset(mp, 39) as m:
        corpus = m.upload(z["feats"], np.array([0, 500], dtype=np.uint64))
        acc = corpus.accumulate(z["states"])
        corpus.close()
    out = str(tmp_path / "updated.mix")
    capi.mixset_write(out, 39, dens_off, dens_mean, dens_var, acc)
    got = synth.read_mixset(out)
    assert np.array_equal(got.mean_acc, z["max_mean_acc"]) and np.array_equal(got.mean_w, z["max_mean_w"])
    assert np.array_equal(got.var_acc, z["max_var_acc"]) and np.array_equal(got.var_w, z["max_var_w"])
    # states that received no frames have zero weight -> NaN parameters -> 1e10 scores, same as the reference
    o = oracle_lib.Oracle(out, 39, lex)
    want = o.score_matrix(z["feats"][:64])
    o.close()
    with capi.Model.from_statistics(39, dens_off, dens_mean, dens_var, acc) as m2:
        got_scores = m2.score_frames(z["feats"][:64], capi.GMM_EXACT)
        assert np.array_equal(got_scores.view(np.uint64), want.view(np.uint64))
        _assert_scores_close(m2.score_frames(z["feats"][:64], capi.GMM_MFMA), want)
    # the same iteration with the statistics kept on the device (no PCIe round trip): sr_accumulate_corpus with NULL outputs,
    # then sr_model_create_from_accumulated
    with capi.Model.from_mixset(mp, 39) as m:
        corpus = m.upload(z["feats"], np.array([0, 500], dtype=np.uint64))
        with pytest.raises(capi.SrError):
            corpus.next_model()                      # nothing accumulated yet
        corpus.accumulate_on_device(z["states"])
        with corpus.next_model() as m3:
            assert np.array_equal(m3.score_frames(z["feats"][:64], capi.GMM_EXACT).view(np.uint64), want.view(np.uint64))
            assert np.array_equal(m3.score_frames(z["feats"][:64], capi.GMM_PREFILTER).view(np.uint64), want.view(np.uint64))
            t3 = m3.topology()
            assert np.array_equal(t3[0], dens_off) and np.array_equal(t3[1], dens_mean) and np.array_equal(t3[2], dens_var)
        corpus.close()


@pytest.mark.parametrize("seed,S,M,D,scale,var_floor,dup", [
    (901, 61, 32, 39, 1.0, 0.5, False),     # full 32-density mixtures, ragged last group (61 = 15*4 + 1)
    (902, 40, (1, 32), 39, 1.0, 0.5, False),  # ragged mixture sizes incl. single-density states
    (903, 24, 16, 39, 30.0, 0.5, False),    # features far from every mean: |b| large, scores ~ 1e4..1e5
    (904, 24, 16, 39, 1.0, 1e-4, False),    # tiny variances: coefficients ~ 1e4, GEMM-form cancellation
    (905, 24, 8, 46, 1.0, 0.5, True),       # dim 46 (K = 95, the limit) and duplicated densities (exact ties)
    (906, 24, 8, 12, 1.0, 0.5, False),      # dim 12 -> one 32-wide k-step
    (907, 16, 40, 39, 1.0, 0.5, False),     # 40 densities per mixture: two 32-slot chunks per state
    (909, 13, 64, 39, 1.0, 0.5, False),     # 64 (BASELINE configs[4] mixtures), odd state count
    (910, 11, (1, 100), 25, 1.0, 0.5, False),  # ragged up to 100 densities: four chunks, some of them empty
    (911, 6, 130, 12, 1.0, 0.5, False),     # 130 densities: two halves of four chunks (round 5; the second half has 2 densities)
    (908, 16, 4, 63, 1.0, 0.5, False),      # dim 63 (K = 129): not eligible either
    (912, 9, 64, 39, 1.0, 0.5, True),       # round 4 (one evaluation per STATE): exact ties ACROSS a state's two chunks
    (913, 7, (33, 128), 39, 1.0, 0.5, True),  # three / four chunks at dim 39, ties across chunks, non-finite frames below
    # round 5: every dimension <= 62 on the prefilter path -- the refinement runs in a padded odd dimension (9, 17, 25, 33, 39 on 768
    # threads; 47, 55, 63 on 512), the fp16 pass with four k-steps (K = 128) from dimension 47
    (920, 20, 8, 1, 1.0, 0.5, False),       # dim 1: no pair at all, only the odd tail          (padded 9)
    (921, 20, 8, 2, 1.0, 0.5, True),        # dim 2: one pair, empty tail
    (922, 24, 16, 13, 1.0, 0.5, False),     # dim 13 -> padded 17 (odd: tail moves to slot 16)
    (923, 24, 32, 26, 1.0, 0.5, True),      # dim 26 -> padded 33 (even: empty tail)
    (924, 24, 32, 33, 1.0, 0.5, False),     # dim 33: exact fit
    (925, 24, 32, 38, 1.0, 0.5, False),     # dim 38 -> padded 39, empty tail
    (926, 24, 32, 40, 1.0, 0.5, True),      # dim 40 -> padded 47, 512 threads, K = 83
    (927, 21, 8, 45, 1.0, 0.5, False),      # dim 45 -> 47, small mixtures in 32-slot panels
    (928, 24, 32, 47, 1.0, 0.5, False),     # dim 47: K = 97 -> four k-steps
    (929, 13, (1, 70), 50, 1.0, 0.5, True),  # dim 50 -> 55, up to three chunks, ties across chunks, non-finite frames
    (930, 24, 32, 62, 3.0, 0.5, False),     # dim 62 (K = 127, the limit) -> 63, features far from the means
    (931, 9, 128, 61, 1.0, 1e-3, True),     # dim 61 -> 63 (odd tail in slot 62), four full chunks, small variances
    # round 5: 129 .. 256 densities per mixture = two halves of four chunks, each half with a candidate limit of its own
    (932, 7, (100, 256), 39, 1.0, 0.5, True),   # ragged: states with one half and with two, ties across chunks and halves, non-finite frames
    (933, 5, 160, 25, 1.0, 0.5, False),     # 160 everywhere (the cliff ledger's shape), dim 25
    (934, 6, 200, 45, 1.0, 0.5, False),     # 200 densities at dim 45 (padded 47: eight panels do not fit the LDS): exact kernel
    (935, 5, 257, 12, 1.0, 0.5, False),     # 257: beyond two halves -> exact kernel
])
def test_prefilter_scores_are_bit_identical(tmp_path, oracle_lib, seed, S, M, D, scale, var_floor, dup):
    """SR_GMM_PREFILTER must return MixtureModel::score's bits: the fp16 stage may only over-select candidates."""
    rng = np.random.default_rng(seed)
    nm = M if np.isscalar(M) else rng.integers(M[0], M[1] + 1, size=S)
    spec = synth.make_mixset(S, nm, D, seed=seed, var_floor=var_floor)
    if dup:  # second density of every mixture := copy of the first (same accumulators, same weight)
        for dl in spec.mixtures:
            if len(dl) > 1:
                spec.mean_acc[dl[1]] = spec.mean_acc[dl[0]]; spec.var_acc[dl[1]] = spec.var_acc[dl[0]]
                spec.mean_w[dl[1]] = spec.mean_w[dl[0]]; spec.var_w[dl[1]] = spec.var_w[dl[0]]
            for k in range(32, len(dl), 32):  # ... and a copy of the first density in every further 32-slot chunk of the state
                src, dst = dl[0], dl[min(k + 5, len(dl) - 1)]
                spec.mean_acc[dst] = spec.mean_acc[src]; spec.var_acc[dst] = spec.var_acc[src]
                spec.mean_w[dst] = spec.mean_w[src]; spec.var_w[dst] = spec.var_w[src]
    mp = str(tmp_path / "pf.mix")
    synth.write_mixset(mp, spec)
    T = 700  # not a multiple of the 128-frame tile nor of 256
    feats = (scale * rng.standard_normal((T, D))).astype(np.float32)
    feats[5] = 0.0
    feats[6] = 1e-20     # squares underflow in bf16/fp32
    feats[7, 0] = 250.0  # one dominant component
    if seed in (912, 913, 929, 931, 932):  # every density of every chunk stays a candidate: all lists, both levels, all chunks
        feats[9] = np.nan
        feats[10, 3] = np.inf
        feats[11, 1] = 1e30
    lex = synth.make_lexicon(max(1, (S - 1) // 3), 3, 1, extra_states_last=(S - 1) % 3)
    o = oracle_lib.Oracle(mp, D, lex)
    want = o.score_matrix(feats)
    o.close()
    with capi.Model.from_mixset(mp, D) as m:
        m.profile(True)
        got = m.score_frames(feats, capi.GMM_PREFILTER)
        prof = m.profile_read()
        m.profile(False)
        exact = m.score_frames(feats, capi.GMM_EXACT)
    assert np.array_equal(exact.view(np.uint64), want.view(np.uint64))
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    # which kernels produced `got`: the fp16 pass + refinement for every eligible model, the exact kernel otherwise
    mx, padded = int(np.max(nm)), next(b for b in (9, 17, 25, 33, 39, 47, 55, 63, 999) if (D | 1) <= b)
    eligible = D <= 62 and (mx <= 128 or (mx <= 256 and padded <= 39))
    assert (prof["prefilter_ms"] > 0 and prof["refined_densities"] >= T * S) == eligible, prof


def test_prefilter_full_size_matches_exact_kernel(tmp_path):
    """4000 states x 32 densities (bench model): 12 000 frames (48 M (frame, state) pairs, 1.5 G density scores
    behind them) through both exact paths, compared bit for bit -- a miss of the true arg-min by the fp16 stage's
    error bound would show up here."""
    lex = synth.make_lexicon(1333, 3, 1)
    spec = synth.make_mixset(lex.n_states, 32, 39, seed=5)
    mp = str(tmp_path / "big.mix")
    synth.write_mixset(mp, spec)
    feats, _ = synth.make_batch(40, 200, 400, 39, seed=11)
    with capi.Model.from_mixset(mp, 39) as m:
        got = m.score_frames(feats, capi.GMM_PREFILTER)
        exact = m.score_frames(feats, capi.GMM_EXACT)
    assert np.array_equal(got.view(np.uint64), exact.view(np.uint64))


def test_prefilter_two_chunk_states_at_scale_match_exact_kernel(tmp_path):
    """BASELINE configs[4]'s model (8000 states x 64 densities = two 32-slot chunks per state): 12 000 frames = 96 M (frame, state)
    pairs through the prefilter path and the exact kernel, bit for bit.  Round 4's refinement evaluates one candidate per STATE and lets
    its list batches lower the shared table entry with floating-point atomic minima issued behind the row stores of the same wave: about
    ten million such atomics here, every one of which must land after the plain store it follows."""
    lex = synth.make_lexicon(2666, 3, 1, extra_states_last=1)
    spec = synth.make_mixset(lex.n_states, 64, 39, seed=6)
    mp = str(tmp_path / "big64.mix")
    synth.write_mixset(mp, spec)
    feats, _ = synth.make_batch(40, 200, 400, 39, seed=12)
    with capi.Model.from_mixset(mp, 39) as m:
        got = m.score_frames(feats, capi.GMM_PREFILTER)
        exact = m.score_frames(feats, capi.GMM_EXACT)
    assert np.array_equal(got.view(np.uint64), exact.view(np.uint64))


def test_fp16_matrix_pipe_keeps_subnormals():
    """An assumption of the prefilter's error bound (gmm_prefilter.hip): fp16 subnormal inputs are not flushed.  If this
    ever fails the library falls back to the exact kernel by itself; the test makes the change visible."""
    import ctypes as C

    ok = C.c_int(0)
    assert capi.lib().sr_probe_fp16_denormals(0, C.byref(ok)) == 0
    assert ok.value == 1


@pytest.mark.parametrize("W,beam", [(60, 60.0), (700, 150.0)])
def test_many_utterances_use_the_throughput_slot_layout(tmp_path, oracle_lib, W, beam):
    """With >= 128 utterances in a launch the decoder gives every wave consecutive slot chunks (only word-end waves run
    the word-end reduction) instead of dealing chunks round-robin: same words, utterance for utterance."""
    lex, spec, mp = _random_setup(tmp_path, 700 + W, W, 3, 1, 2, 12)
    o = oracle_lib.Oracle(mp, 12, lex, am_threshold=beam)
    rng = np.random.default_rng(W)
    lens = rng.integers(12, 30, size=160)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    feats = rng.standard_normal((int(off[-1]), 12)).astype(np.float32)
    for u in range(0, 160, 7):  # some utterances drawn from the model, so that the beam prunes
        x = synth.sample_utterance(spec, lex, rng.integers(1, lex.n_words, size=2), seed=u)[: lens[u]]
        feats[int(off[u]):int(off[u]) + len(x)] = x
    word_off, automaton, sil_state = lex.flatten()
    with capi.Model.from_mixset(mp, 12) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil_state)
        corpus = m.upload(feats, off)
        words, woff, (tbs, tbw, tbb) = corpus.recognize(lexh, beam, 10.0, capi.GMM_PREFILTER, traceback=True)
        few = m.upload(feats[: int(off[5])], off[:6])  # the same first utterances through the round-robin layout
        wf, of = few.recognize(lexh, beam, 10.0, capi.GMM_PREFILTER)
        assert np.array_equal(wf, words[: int(woff[5])]) and np.array_equal(of, woff[:6])
        few.close()
        for u in range(160):
            x = feats[int(off[u]):int(off[u + 1])]
            w, (os_, ow, ob) = o.decode(x, traceback=True)
            assert np.array_equal(w, words[int(woff[u]):int(woff[u + 1])]), u
            a = int(off[u]) + u
            assert np.array_equal(tbw[a:a + len(x) + 1], ow) and np.array_equal(tbb[a:a + len(x) + 1], ob)
            assert np.array_equal(tbs[a:a + len(x) + 1].view(np.uint64), os_.view(np.uint64))
        corpus.close()
        lexh.close()
    o.close()


def test_handles_give_their_device_memory_back(tmp_path):
    """create / use / destroy every handle type repeatedly: free device memory must not creep.  (Free memory is read from
    the HIP runtime the library itself is linked against: a second runtime -- the one bundled with torch -- fails to find the
    GPU when it is initialised after the library has been at work, which made this test depend on the test order.)"""
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        assert hip.hipDeviceSynchronize() == 0
        fr, tot = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot)) == 0
        return fr.value

    lex, spec, mp = _random_setup(tmp_path, 801, 20, 3, 1, 4, 39)
    feats, off = synth.make_batch(12, 30, 60, 39, seed=2)
    word_off, aut, sil = lex.flatten()
    W = lex.n_words
    lm = np.full((W, W), 3.0, np.float32)
    btdp = np.array([[3, 0, 30, 5], [1, 0, 40, 2]], np.float32)
    free = []
    for it in range(8):
        with capi.Model.from_mixset(mp, 39) as m:
            c = m.upload(feats, off)
            lexh = m.lexicon(word_off, aut, lex.silence_idx, (3.0, 0.0, 30.0), sil)
            bg = m.bigram(word_off, aut, lex.silence_idx, lm, btdp)
            c.recognize(lexh, 100.0, 10.0)
            c.score(capi.GMM_MFMA)
            c.recognize_bigram(bg, 100.0, capi.FLT_MAX)
            st, _ = c.align([np.array([sil, 1, 2, 3, sil], np.uint16)] * 12, (3.0, 0.0, 30.0), sil)
            c.accumulate(st)
            c.path_scores(st)
            bg.close(); lexh.close(); c.close()
        free.append(free_bytes())
    assert free[-1] >= free[2] - (1 << 20), free  # (the first iterations warm up runtime pools)


@pytest.mark.parametrize("S,M,D", [(24, 16, 39), (13, 64, 39), (9, 8, 25), (12, 32, 47), (9, 16, 62), (10, 8, 6)])
def test_prefilter_overflow_and_nonfinite_features(tmp_path, oracle_lib, S, M, D):
    """Features the fp16 stage cannot represent: |x| = 256 and 300 (the square leaves fp16's range: inf), 1e4, 7e4 (x itself
    does), +-inf and NaN.  Their frames' prefilter scores are inf/NaN, every density stays a candidate, and the FP64 stage
    decides alone: SR_GMM_PREFILTER == SR_GMM_EXACT == MixtureModel::score bit for bit (NaN scores never win: 1e10)."""
    rng = np.random.default_rng(S * 100 + M)
    spec = synth.make_mixset(S, M, D, seed=S + M)
    mp = str(tmp_path / "ovf.mix")
    synth.write_mixset(mp, spec)
    T = 300
    feats = rng.standard_normal((T, D)).astype(np.float32)
    specials = [255.0, 256.0, -256.0, 300.0, 1e4, -1e4, 65504.0, 7e4, 3e38, np.inf, -np.inf, np.nan]
    for i, v in enumerate(specials):
        feats[10 + i, (3 * i) % D] = v
        feats[40 + i, :] = v            # the whole frame
    feats[70, 0], feats[70, 1] = np.inf, -np.inf
    feats[71, 0], feats[71, 1] = np.nan, 300.0
    lex = synth.make_lexicon(max(1, (S - 1) // 3), 3, 1, extra_states_last=(S - 1) % 3)
    o = oracle_lib.Oracle(mp, D, lex)
    want = o.score_matrix(feats)
    o.close()
    with capi.Model.from_mixset(mp, D) as m:
        m.profile(True)
        got = m.score_frames(feats, capi.GMM_PREFILTER)
        prof = m.profile_read()
        m.profile(False)
        exact = m.score_frames(feats, capi.GMM_EXACT)
    assert np.array_equal(exact.view(np.uint64), want.view(np.uint64))
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    # which kernels produced `got`: the fp16 pass + refinement for every eligible model, the exact kernel otherwise
    eligible = D <= 62 and M <= 128
    assert (prof["prefilter_ms"] > 0 and prof["refined_densities"] >= T * S) == eligible, prof
    assert np.all(want[40 + specials.index(np.nan)] == 1e10)  # an all-NaN frame: every state keeps min_score's seed


def test_fp16_accumulation_stays_inside_the_bounds_model():
    """The other hardware assumption of the prefilter's error bound: fp32 accumulation inside the MFMA chain errs by at most
    2^-24 of the running magnitude per addition (87 additions at K = 96).  Adversarial dot products with known exact sums;
    the library runs the same probe at model creation and falls back to the exact kernel if it fails."""
    import ctypes as C

    ok, worst = C.c_int(0), C.c_double(0.0)
    assert capi.lib().sr_probe_fp16_accumulation(0, C.byref(ok), C.byref(worst)) == 0
    print(f"worst |error| / (2^-24 sum|ab|) = {worst.value:.3f} (model: 87)")
    assert ok.value == 1 and worst.value <= 87.0


def test_wave_handoff_stress():
    """The memory ordering the refinement kernel's candidate lists rest on (csrc/gmm_prefilter.hip: entries stored by one lane and
    loaded by another lane of the same wave with no wait in between; a no-return atomic minimum behind another lane's store to the
    same address), hammered on this device: tools/wave_handoff_stress.hip, 1e9 hand-offs under load, none stale."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "wave_handoff_stress")
    assert os.path.exists(exe), "tools/wave_handoff_stress is built by __graft_entry__.build()"
    r = subprocess.run([exe, "2000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "0 stale entries, 0 wrong minima" in r.stdout, (r.stdout, r.stderr)


@pytest.mark.parametrize("route", ["deferred", "full segments", "in-kernel lists"])
@pytest.mark.parametrize("S,M,D", [(13, 64, 39), (7, (33, 128), 39), (5, 160, 25), (9, (40, 64), 12)])
def test_deferred_leftovers_route_matches_the_in_kernel_lists(tmp_path, oracle_lib, monkeypatch, route, S, M, D):
    """Round 5: for mixtures of more than 32 densities the refinement's main pass appends the pairs with candidates left over to per-wave
    segments and gmm_drain_kernel works them off (default); a segment that is full makes the lane evaluate on the spot (forced here with
    a capacity of 3 entries); SRGPU_DEFER_MB=0 keeps rounds 2-4's in-kernel lists.  All three: MixtureModel::score's bits."""
    if route == "full segments":
        monkeypatch.setenv("SRGPU_DEFER_CAP", "3")
    elif route == "in-kernel lists":
        monkeypatch.setenv("SRGPU_DEFER_MB", "0")
    rng = np.random.default_rng(S * 7 + D)
    nm = M if np.isscalar(M) else rng.integers(M[0], M[1] + 1, size=S)
    spec = synth.make_mixset(S, nm, D, seed=S + D)
    for dl in spec.mixtures:  # exact ties across chunks: a copy of the first density in every further chunk
        for k in range(32, len(dl), 32):
            src, dst = dl[0], dl[min(k + 5, len(dl) - 1)]
            spec.mean_acc[dst] = spec.mean_acc[src]; spec.var_acc[dst] = spec.var_acc[src]
            spec.mean_w[dst] = spec.mean_w[src]; spec.var_w[dst] = spec.var_w[src]
    mp = str(tmp_path / "df.mix")
    synth.write_mixset(mp, spec)
    T = 3000
    feats = rng.standard_normal((T, D)).astype(np.float32)
    feats[9] = np.nan          # every density of every chunk a candidate
    feats[10, 3] = np.inf
    feats[100:164] *= 40.0     # a run of frames whose fp16 bound is loose: many candidates per pair, lists fill at once
    lex = synth.make_lexicon(max(1, (S - 1) // 3), 3, 1, extra_states_last=(S - 1) % 3)
    o = oracle_lib.Oracle(mp, D, lex)
    want = o.score_matrix(feats)
    o.close()
    with capi.Model.from_mixset(mp, D) as m:
        m.profile(True)
        got = m.score_frames(feats, capi.GMM_PREFILTER)
        prof = m.profile_read()
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert prof["refined_densities"] >= T * S


@pytest.mark.parametrize("S,M,D", [(40, 32, 39), (24, (1, 32), 25), (21, 8, 12), (16, 16, 38), (9, 5, 1)])
def test_shared_variances_score_bit_identically(tmp_path, oracle_lib, S, M, D):
    """A model whose states each share ONE variance vector among their densities (mixture / global pooling, tied variances: var_idx !=
    mean_idx): MixtureModel::score's bits from the prefilter path and the exact kernel.  (Round 5 tried a scalar-operand refinement for
    such models: bit-identical and 7 % slower, profiles/r5_pooled_scalar_operands_experiment.txt.)"""
    rng = np.random.default_rng(S + D)
    nm = M if np.isscalar(M) else rng.integers(M[0], M[1] + 1, size=S)
    spec = synth.make_mixset(S, nm, D, seed=S * 3 + D, tie_vars=True)
    mp = str(tmp_path / "tied.mix")
    synth.write_mixset(mp, spec)
    T = 1500
    feats = rng.standard_normal((T, D)).astype(np.float32)
    feats[7] = np.nan
    feats[8, 0] = 300.0
    lex = synth.make_lexicon(max(1, (S - 1) // 3), 3, 1, extra_states_last=(S - 1) % 3)
    o = oracle_lib.Oracle(mp, D, lex)
    want = o.score_matrix(feats)
    o.close()
    with capi.Model.from_mixset(mp, D) as m:
        got = m.score_frames(feats, capi.GMM_PREFILTER)
        exact = m.score_frames(feats, capi.GMM_EXACT)
    assert np.array_equal(exact.view(np.uint64), want.view(np.uint64))
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_deferred_leftovers_across_score_chunks_and_the_feeder(tmp_path, oracle_lib, monkeypatch):
    """Mixtures of 40 densities (two chunks per state: the deferred-leftover route) with a score-table budget that cuts the corpus into
    many chunks -- the segments are one workspace reused launch after launch on the scoring stream while the search of the previous chunk
    runs beside it -- and through sr_recognize_batch's asynchronous feeder (four scoring launches per chunk): words as with one chunk."""
    lex, spec, mp = _random_setup(tmp_path, 451, 40, 3, 1, 40, 39)
    feats, off = synth.make_batch(60, 30, 90, 39, seed=452)
    word_off, automaton, sil_state = lex.flatten()
    results = []
    for mb in ("4096", "1"):
        monkeypatch.setenv("SRGPU_SCORE_CHUNK_MB", mb)
        with capi.Model.from_mixset(mp, 39) as m:
            lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil_state)
            corpus = m.upload(feats, off)
            results.append((corpus.recognize(lexh, 200.0, 10.0, capi.GMM_PREFILTER), corpus.score(capi.GMM_PREFILTER),
                            m.recognize_batch(lexh, feats, off, 200.0, 10.0, capi.GMM_PREFILTER)))
            corpus.close()
            lexh.close()
    (w0, o0), s0, (b0, bo0) = results[0]
    (w1, o1), s1, (b1, bo1) = results[1]
    assert np.array_equal(w0, w1) and np.array_equal(o0, o1) and np.array_equal(b0, w0) and np.array_equal(b1, w0) and np.array_equal(bo1, o0)
    assert np.array_equal(s0.view(np.uint64), s1.view(np.uint64))
    o = oracle_lib.Oracle(mp, 39, lex)
    assert np.array_equal(o.score_matrix(feats[:500]).view(np.uint64), s0[:500].view(np.uint64))
    o.close()


@pytest.mark.parametrize("D,max_approx", [(64, True), (100, True), (160, True), (77, False)])
def test_dimensions_beyond_the_matrix_core_kernels(tmp_path, oracle_lib, D, max_approx):
    """MixtureModel::density_score_sse is dimension-generic (Mixtures.cpp:645-690).  Round 5: dimensions 64 .. 160 are taken (rounds 1-4
    refused dim > 63); every scoring selector resolves to the exact kernel there -- SR_GMM_MFMA has no instantiation, the prefilter path ends
    at 62 -- so scores are MixtureModel::score's bits (1e-12 with sum scoring: device exp / log), words, tracebacks and alignments the oracle's."""
    lex = synth.make_lexicon(12, 3, 1)
    spec = synth.make_mixset(lex.n_states, 3, D, seed=600 + D)
    mp = str(tmp_path / "wide.mix")
    synth.write_mixset(mp, spec)
    rng = np.random.default_rng(D)
    utts = [synth.sample_utterance(spec, lex, rng.integers(1, lex.n_words, size=3), seed=610 + i)[:50].astype(np.float32) for i in range(3)]
    off = np.concatenate([[0], np.cumsum([len(x) for x in utts])]).astype(np.uint64)
    feats = np.concatenate(utts)
    word_off, automaton, sil_state = lex.flatten()
    o = oracle_lib.Oracle(mp, D, lex, am_threshold=120.0, max_approx=max_approx)
    want = o.score_matrix(feats)
    with capi.Model.from_mixset(mp, D, capi.POOL_NONE, max_approx) as m:
        for kernel in (capi.GMM_DEFAULT, capi.GMM_MFMA, capi.GMM_PREFILTER, capi.GMM_EXACT):
            got = m.score_frames(feats, kernel)
            if max_approx:
                assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), kernel
            else:
                _assert_scores_close(got, want, rtol=1e-12)
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil_state)
        corpus = m.upload(feats, off)
        words, woff, (tbs, tbw, tbb) = corpus.recognize(lexh, 120.0, 10.0, capi.GMM_DEFAULT, traceback=True)
        aut = np.asarray([sil_state] + list(automaton[word_off[1]:word_off[2]]) + [sil_state], np.uint16)
        st, cost = corpus.align([aut] * 3, (3.0, 0.0, 30.0), sil_state, capi.GMM_DEFAULT)
        for u, x in enumerate(utts):
            w, (os_, ow, ob) = o.decode(x, traceback=True)
            assert np.array_equal(w, words[int(woff[u]):int(woff[u + 1])])
            b = int(off[u]) + u
            assert np.array_equal(tbw[b:b + len(ow)], ow) and np.array_equal(tbb[b:b + len(ob)], ob)
            if max_approx:
                assert np.array_equal(tbs[b:b + len(os_)].view(np.uint64), os_.view(np.uint64))
                s_, c_ = o.align_full(x, aut)
                assert np.array_equal(st[int(off[u]):int(off[u + 1])], s_) and cost[u] == c_
        corpus.close()
        lexh.close()
    o.close()
