"""Lexicons beyond the LDS-resident decoders (more than 8192 type-padded trellis positions): decode_big_kernel keeps the
hypothesis arrays in device memory.  Words and traceback must equal Recognizer::recognizeSequence_pruned's
(sietill/Recognizer.cpp:103-232) through the oracle, with and without negative emission costs."""
import numpy as np
import pytest

from speechrecognition_amd import capi, synth

pytestmark = pytest.mark.gpu


def _model(tmp_path, lex, dim, seed, tight=False):
    spec = synth.make_mixset(lex.n_states, 2, dim, seed=seed, var_floor=0.002 if tight else 0.5)
    if tight:  # variances small enough for negative emission costs: the pre-AM early-out (Recognizer.cpp:143,173) is live
        mu = spec.mean_acc / spec.mean_w[:, None]
        var = 0.004 * (spec.var_acc / spec.var_w[:, None] - mu ** 2)
        spec.var_acc = (var + mu ** 2) * spec.var_w[:, None]
    mp = str(tmp_path / f"big{seed}.mix")
    synth.write_mixset(mp, spec)
    return spec, mp


@pytest.mark.parametrize("n_words,tight", [(1400, False), (1400, True), (3000, False)])
def test_big_lexicon_matches_oracle(n_words, tight, tmp_path, oracle_lib):
    dim = 12
    lex = synth.make_lexicon(n_words, 3, 2)  # 1 + 6 n_words positions: 8401 / 18001
    word_off, automaton, sil_state = lex.flatten()
    assert int(word_off[-1]) > 8192
    spec, mp = _model(tmp_path, lex, dim, 900 + n_words + int(tight), tight)
    rng = np.random.default_rng(77)
    utts = [synth.sample_utterance(spec, lex, rng.integers(1, lex.n_words, size=2), seed=920 + i, frames_per_state=(1, 2),
                                   noise=0.8 if tight else 1.0) for i in range(3)]
    utts.append(synth.make_features(9, dim, seed=931))  # nothing to recognise: the beam empties or keeps silence
    off = np.concatenate([[0], np.cumsum([len(x) for x in utts])]).astype(np.uint64)
    feats = np.concatenate(utts)
    for beam, wp in ((120.0, 4.0), (25.0, 0.0)):
        o = oracle_lib.Oracle(mp, dim, lex, am_threshold=beam, word_penalty=wp)
        if tight:
            assert (o.score_matrix(feats) < 0).any()
        with capi.Model.from_mixset(mp, dim) as m:
            lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil_state)
            corpus = m.upload(feats, off)
            words, woff, (tbs, tbw, tbb) = corpus.recognize(lexh, beam, wp, capi.GMM_EXACT, traceback=True)
            n_items = 0
            for u in range(len(utts)):
                w, (os_, ow, ob) = o.decode(utts[u], traceback=True)
                assert np.array_equal(w, words[int(woff[u]):int(woff[u + 1])])
                b = int(off[u]) + u
                assert np.array_equal(tbs[b:b + len(os_)].view(np.uint64), os_.view(np.uint64))
                assert np.array_equal(tbw[b:b + len(ow)], ow) and np.array_equal(tbb[b:b + len(ob)], ob)
                n_items += len(w)
            if beam > 100:
                assert n_items > 0
            corpus.close()
            lexh.close()
        o.close()


@pytest.mark.parametrize("tight", [False, True])
def test_big_short_word_lexicon_runs_the_word_per_lane_kernel(tight, tmp_path, oracle_lib):
    """about 2 700 .. 3 072 three-state words are more than 8 192 type-padded positions, so the lexicon is `big`, but the
    word-per-lane kernel does not depend on the slot count: it runs first, and what it flags (negative emission costs: the `tight`
    model) is redone by decode_big_kernel, which exits at once for every other utterance.  Words and traceback against the oracle
    and against decode_big_kernel on everything (SR_SEARCH_GENERAL_KERNEL)."""
    dim = 12
    lex = synth.make_lexicon(2900, 3, 1)  # 1 + 3 * 2900 = 8701 positions
    word_off, automaton, sil_state = lex.flatten()
    assert int(word_off[-1]) > 8192 and lex.n_words <= 3072
    spec, mp = _model(tmp_path, lex, dim, 940 + int(tight), tight)
    rng = np.random.default_rng(78)
    utts = [synth.sample_utterance(spec, lex, rng.integers(1, lex.n_words, size=2), seed=950 + i, frames_per_state=(1, 2),
                                   noise=0.8 if tight else 1.0) for i in range(3)]
    utts.append(synth.make_features(7, dim, seed=955))
    off = np.concatenate([[0], np.cumsum([len(x) for x in utts])]).astype(np.uint64)
    feats = np.concatenate(utts)
    beam, wp = 120.0, 4.0
    o = oracle_lib.Oracle(mp, dim, lex, am_threshold=beam, word_penalty=wp)
    if tight:
        assert (o.score_matrix(feats) < 0).any()
    with capi.Model.from_mixset(mp, dim) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil_state)
        assert lexh.describe().startswith("words ") and "replay: big" in lexh.describe(), lexh.describe()
        corpus = m.upload(feats, off)
        words, woff, (tbs, tbw, tbb) = corpus.recognize(lexh, beam, wp, capi.GMM_EXACT, traceback=True)
        gw, goff, (gs, gtw, gtb) = corpus.recognize(lexh, beam, wp, capi.GMM_EXACT, traceback=True, general_kernel=True)
        assert np.array_equal(words, gw) and np.array_equal(woff, goff) and np.array_equal(tbw, gtw) and np.array_equal(tbb, gtb)
        assert np.array_equal(tbs.view(np.uint64), gs.view(np.uint64))
        n_items = 0
        for u in range(len(utts)):
            w, (os_, ow, ob) = o.decode(utts[u], traceback=True)
            assert np.array_equal(w, words[int(woff[u]):int(woff[u + 1])])
            b = int(off[u]) + u
            assert np.array_equal(tbs[b:b + len(os_)].view(np.uint64), os_.view(np.uint64))
            assert np.array_equal(tbw[b:b + len(ow)], ow) and np.array_equal(tbb[b:b + len(ob)], ob)
            n_items += len(w)
        assert n_items > 0
        corpus.close()
        lexh.close()
    o.close()
