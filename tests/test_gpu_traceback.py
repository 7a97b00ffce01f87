"""GPU tests of the guarded traceback walk (csrc/traceback.h): the device walker that ends every search kernel, run alone
through sr_traceback_corpus on dumps the search itself produced -- and on corrupted ones, which must come back as
SR_ECORRUPT instead of being followed (round 2's unexplained GPU memory fault, DESIGN.md section 8 and DESIGN_HISTORY.md section 8.1)."""
import numpy as np
import pytest

from speechrecognition_amd import capi, synth

pytestmark = pytest.mark.gpu

TDP = (3.0, 0.0, 30.0)


def test_device_walk_equals_search_and_host_walk_and_rejects_corruption(tmp_path):
    lex = synth.make_lexicon(40, 3, 1)
    spec = synth.make_mixset(lex.n_states, 4, 39, seed=5)
    mp = str(tmp_path / "m.mix")
    synth.write_mixset(mp, spec)
    feats, off = synth.make_batch(70, 30, 120, 39, seed=6)
    word_off, automaton, sil = lex.flatten()
    with capi.Model.from_mixset(mp, 39) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, TDP, sil)
        c = m.upload(feats, off)
        for general, slots in ((False, False), (False, True), (True, False)):  # word-per-lane, slot-per-lane, general kernel
            words, woff, (tbs, tbw, tbb) = c.recognize(lexh, 150.0, 10.0, capi.GMM_PREFILTER, traceback=True, general_kernel=general, slot_kernel=slots)
            assert len(words) > 70
            w2, o2 = c.retrace(lexh, tbw, tbb)          # the device walker alone
            assert np.array_equal(w2, words) and np.array_equal(o2, woff)
            for u in range(70):                         # the host walker, utterance by utterance
                b, e = int(off[u]) + u, int(off[u + 1]) + u + 1
                assert np.array_equal(capi.traceback_words(tbw[b:e], tbb[b:e], lex.silence_idx, lex.n_words),
                                      words[int(woff[u]):int(woff[u + 1])])
        # corrupt the last entry of one utterance in three ways; each call must return the status, none may fault
        u = 17
        T = int(off[u + 1] - off[u])
        last = int(off[u]) + u + T
        for what in ("bkp == t", "bkp > t", "word outside the lexicon"):
            w, b = tbw.copy(), tbb.copy()
            if what == "bkp == t":
                b[last] = T
            elif what == "bkp > t":
                b[last] = 65535
            else:
                w[last] = lex.n_words
            with pytest.raises(capi.SrError) as e:
                c.retrace(lexh, w, b)
            assert e.value.code == capi.SR_ECORRUPT and f"utterance {u}" in str(e.value)
        # the handle is still good afterwards
        w3, o3 = c.recognize(lexh, 150.0, 10.0)
        assert np.array_equal(w3, words) and np.array_equal(o3, woff)
        c.close(); lexh.close()


def test_all_words_tie_exactly(tmp_path, oracle_lib):
    """Every word a clone of the first (its states carry the same densities, accumulator for accumulator): all word hypotheses
    tie bit for bit in every frame, so everything that is decided by source ORDER is exercised at once -- the boundary candidate
    against the in-word candidates (first word-end index per class against the slot's own word, Recognizer.cpp:126,143-157), the
    FIRST minimal surviving word end of traceback[t] (:199-205) when a lane holds several of them (round 3: the lane must
    keep the one with the smallest original index), padding lanes, type-boundary waves.  300 utterances -> the throughput
    geometry (256 threads x 4 slots: one lane owns four tied word ends); 3 utterances -> the wide one."""
    W, M, D = 200, 2, 39
    lex = synth.make_lexicon(W, 3, 1)
    spec = synth.make_mixset(lex.n_states, M, D, seed=77)
    # states: 0 = silence, word w (1..W) owns states 1 + 3 (w - 1) + k; density rows are state-major, M per state
    for w in range(2, W + 1):
        for k in range(3):
            src, dst = (1 + k) * M, (1 + 3 * (w - 1) + k) * M
            for arr in (spec.mean_acc, spec.var_acc):
                arr[dst:dst + M] = arr[src:src + M]
            spec.mean_w[dst:dst + M] = spec.mean_w[src:src + M]
            spec.var_w[dst:dst + M] = spec.var_w[src:src + M]
    mp = str(tmp_path / "clones.mix")
    synth.write_mixset(mp, spec)
    word_off, automaton, sil = lex.flatten()
    o = oracle_lib.Oracle(mp, D, lex, tdp=TDP, am_threshold=60.0, word_penalty=10.0)
    with capi.Model.from_mixset(mp, D) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, TDP, sil)
        for n_utts in (300, 3):
            feats, off = synth.make_batch(n_utts, 8, 30, D, seed=78 + n_utts)
            for u in range(0, n_utts, 2):  # half of them speech-like: a few cloned words in a row
                x = synth.sample_utterance(spec, lex, [3, 150, 77], seed=u)
                n = min(len(x), int(off[u + 1] - off[u]))
                feats[int(off[u]):int(off[u]) + n] = x[:n]
            c = m.upload(feats, off)
            sc = c.score(capi.GMM_EXACT)
            assert np.array_equal(sc[:, 1].view(np.uint64), sc[:, 1 + 3 * 57].view(np.uint64))  # the clones do tie
            for general, slots in ((False, False), (False, True), (True, False)):  # word-per-lane, slot-per-lane, general kernel
                words, woff, (tbs, tbw, tbb) = c.recognize(lexh, 60.0, 10.0, capi.GMM_EXACT, traceback=True, general_kernel=general, slot_kernel=slots)
                for u in range(0, n_utts, max(1, n_utts // 40)):
                    x = feats[int(off[u]):int(off[u + 1])]
                    w, (os_, ow, ob) = o.decode(x, traceback=True)
                    a = int(off[u]) + u
                    assert np.array_equal(w, words[int(woff[u]):int(woff[u + 1])]), (n_utts, general, slots, u)
                    assert np.array_equal(tbw[a:a + len(x) + 1], ow) and np.array_equal(tbb[a:a + len(x) + 1], ob), (n_utts, general, slots, u)
                    assert np.array_equal(tbs[a:a + len(x) + 1].view(np.uint64), os_.view(np.uint64)), (n_utts, general, slots, u)
            c.close()
        lexh.close()
    o.close()


def test_silence_last_in_the_lexicon_padding_lanes_stay_out_of_best(tmp_path, oracle_lib):
    """Silence as the LAST word: state 0 belongs to an ordinary word.  The padding lanes that fill the one-position-word
    chunk of the type-sorted net read state 0's emission; their 'dead position-1 slot' candidate (Recognizer.cpp:139,155)
    must not reach best_score -- it would be a word entry without the word penalty, up to word_penalty below the true best,
    and the beam of that frame would shrink by as much.  Narrow beam, large penalty: any such shift changes what survives."""
    W, M, D = 60, 2, 39
    ws = np.full(W + 1, 3, dtype=np.uint16); ws[-1] = 1
    wr = np.ones(W + 1, dtype=np.uint16)
    lex = synth.LexiconSpec(ws, wr, W)
    spec = synth.make_mixset(lex.n_states, M, D, seed=31)
    mp = str(tmp_path / "sil_last.mix")
    synth.write_mixset(mp, spec)
    word_off, automaton, sil = lex.flatten()
    assert sil == lex.n_states - 1
    beam, wp = 25.0, 20.0
    o = oracle_lib.Oracle(mp, D, lex, tdp=TDP, am_threshold=beam, word_penalty=wp)
    n_utts = 300
    feats, off = synth.make_batch(n_utts, 20, 60, D, seed=32)
    rng = np.random.default_rng(33)
    for u in range(n_utts):  # word after word without silence in between, word 0 (state 0) often among them
        seq = [int(w) for w in rng.integers(0, W, size=4)]
        seq[int(rng.integers(1, 4))] = 0
        x = np.concatenate([synth.sample_utterance(spec, lex, [w], seed=1000 * u + i, noise=0.6)[2:-2] for i, w in enumerate(seq)])
        n = min(len(x), int(off[u + 1] - off[u]))
        feats[int(off[u]):int(off[u]) + n] = x[:n]
    with capi.Model.from_mixset(mp, D) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, TDP, sil)
        c = m.upload(feats, off)
        got = {}
        for which, (general, slots) in enumerate(((False, False), (False, True), (True, False))):  # word-per-lane, slot-per-lane, general
            got[which] = c.recognize(lexh, beam, wp, capi.GMM_EXACT, traceback=True, general_kernel=general, slot_kernel=slots)
        w0, o0, (s0, tw0, tb0) = got[0]
        for which in (1, 2):
            w1, o1, (s1, tw1, tb1) = got[which]
            assert np.array_equal(s0.view(np.uint64), s1.view(np.uint64)) and np.array_equal(tw0, tw1) and np.array_equal(tb0, tb1), which
            assert np.array_equal(w0, w1) and np.array_equal(o0, o1), which
        for u in range(0, n_utts, 6):
            x = feats[int(off[u]):int(off[u + 1])]
            w, (os_, ow, ob) = o.decode(x, traceback=True)
            a = int(off[u]) + u
            assert np.array_equal(w, w0[int(o0[u]):int(o0[u + 1])]), u
            assert np.array_equal(tw0[a:a + len(x) + 1], ow) and np.array_equal(tb0[a:a + len(x) + 1], ob), u
            assert np.array_equal(s0[a:a + len(x) + 1].view(np.uint64), os_.view(np.uint64)), u
        c.close(); lexh.close()
    o.close()


def test_ragged_lexica_sample_of_the_soak(tmp_path, oracle_lib):
    """A fixed sample of tools/soak_parity.py's ragged cases (synth.make_ragged_lexicon: silence anywhere in the word list,
    one-position words beside 40-state ones, cloned words, words that begin in silence) through scoring, search with
    traceback, both aligners and the bigram search, everything against the oracle.  The full soak (hundreds of cases) is a
    tool, not a test: DESIGN_HISTORY.md section 8 records its counts."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("soak_parity", os.path.join(os.path.dirname(__file__), "..", "tools", "soak_parity.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    for case in range(16):
        soak.run_case(case, seed0=41, ragged=True, tmp=str(tmp_path))
    for case in range(16):  # words of one to four positions only: the word-per-lane kernel, checked against the slot kernel too
        soak.run_case(case, seed0=47, ragged="short", tmp=str(tmp_path))


def test_empty_and_one_frame_utterances_in_a_batch(tmp_path, oracle_lib):
    """Utterances of 0, 1 and 2 frames between ordinary ones (Recognizer.cpp:103-232 with begin == end: no frame loop, an empty
    word sequence): all three search kernels and the bigram search, next to the oracle."""
    D = 12
    lex = synth.make_lexicon(30, 3, 1)
    spec = synth.make_mixset(lex.n_states, 2, D, seed=3)
    mp = str(tmp_path / "m.mix")
    synth.write_mixset(mp, spec)
    lens = [0, 25, 1, 0, 2, 40, 0]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    feats = np.random.default_rng(4).standard_normal((int(off[-1]), D)).astype(np.float32)
    x = synth.sample_utterance(spec, lex, [3, 7], seed=5)[:40]
    feats[int(off[5]):int(off[5]) + len(x)] = x
    word_off, automaton, sil = lex.flatten()
    o = oracle_lib.Oracle(mp, D, lex, tdp=TDP, am_threshold=80.0, word_penalty=10.0)
    W = lex.n_words
    lm = (-np.log(np.random.default_rng(6).dirichlet(np.ones(W), size=W))).T.astype(np.float32).copy()
    btdp = np.array([[3.0, 0.0, 30.0, 0.0], [1.0, 0.0, 40.0, 2.0]], np.float32)
    with capi.Model.from_mixset(mp, D) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, TDP, sil)
        c = m.upload(feats, off)
        for general, slots in ((False, False), (False, True), (True, False)):
            words, woff, (tbs, tbw, tbb) = c.recognize(lexh, 80.0, 10.0, capi.GMM_PREFILTER, traceback=True, general_kernel=general, slot_kernel=slots)
            for u, n in enumerate(lens):
                xs = feats[int(off[u]):int(off[u + 1])]
                w, (os_, ow, ob) = o.decode(xs, traceback=True)
                a = int(off[u]) + u
                assert np.array_equal(w, words[int(woff[u]):int(woff[u + 1])]), (general, slots, u)
                assert np.array_equal(tbw[a:a + n + 1], ow) and np.array_equal(tbb[a:a + n + 1], ob), (general, slots, u)
                assert np.array_equal(tbs[a:a + n + 1].view(np.uint64), os_.view(np.uint64)), (general, slots, u)
                if n == 0:
                    assert len(w) == 0
        bg = m.bigram(word_off, automaton, lex.silence_idx, lm, btdp)
        for dense_states in (False, True):
            gw, gs, gt, goff = c.recognize_bigram(bg, 60.0, 8.0, dense_states=dense_states)
            for u, n in enumerate(lens):
                dense = o.score_matrix(feats[int(off[u]):int(off[u + 1])]) if n else np.zeros((0, lex.n_states))
                w, s_, t_ = oracle_lib.bigram_decode(dense, word_off, automaton, lex.silence_idx, lm, btdp, 60.0, 8.0)
                a, b = int(goff[u]), int(goff[u + 1])
                assert np.array_equal(gw[a:b], w) and np.array_equal(gt[a:b], t_), (dense_states, u)
                assert np.array_equal(gs[a:b].view(np.uint32), s_.view(np.uint32)), (dense_states, u)
        bg.close(); c.close(); lexh.close()
    o.close()


def test_longest_utterance_the_back_pointers_allow(tmp_path, oracle_lib):
    """Book::bkp is a uint16_t (Recognizer.hpp:75-89): the library takes utterances of up to 65 535 frames (one frame more is
    SR_ELIMIT at upload: the reference's back pointers would wrap).  The maximum itself, a tiny lexicon, the three kernels against
    the oracle -- back pointers up to 65 534."""
    D = 4
    lex = synth.make_lexicon(2, 3, 1)
    spec = synth.make_mixset(lex.n_states, 2, D, seed=8)
    mp = str(tmp_path / "m.mix")
    synth.write_mixset(mp, spec)
    T = 65_535
    feats = np.random.default_rng(9).standard_normal((T, D)).astype(np.float32)
    for i, start in enumerate(range(0, T - 400, 1500)):  # a word now and then
        x = synth.sample_utterance(spec, lex, [1 + i % 2], seed=100 + i)
        feats[start:start + len(x)] = x
    word_off, automaton, sil = lex.flatten()
    o = oracle_lib.Oracle(mp, D, lex, tdp=TDP, am_threshold=100.0, word_penalty=10.0)
    w, (os_, ow, ob) = o.decode(feats, traceback=True)
    o.close()
    assert ob.max() > 65_000 and len(w) > 20
    with capi.Model.from_mixset(mp, D) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, TDP, sil)
        with pytest.raises(capi.SrError) as e:
            m.upload(np.zeros((T + 1, D), np.float32), np.array([0, T + 1], np.uint64))
        assert e.value.code == -4  # SR_ELIMIT
        c = m.upload(feats, np.array([0, T], np.uint64))
        for general, slots in ((False, False), (False, True), (True, False)):
            words, woff, (tbs, tbw, tbb) = c.recognize(lexh, 100.0, 10.0, capi.GMM_EXACT, traceback=True, general_kernel=general, slot_kernel=slots)
            assert np.array_equal(words, w), (general, slots)
            assert np.array_equal(tbw, ow) and np.array_equal(tbb, ob) and np.array_equal(tbs.view(np.uint64), os_.view(np.uint64)), (general, slots)
        c.close(); lexh.close()
