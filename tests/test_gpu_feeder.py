"""GPU tests of the frame-batch feeder (sr_corpus_upload_async, pinned double-buffered staging on its own stream) and of the
multi-device driver (sr_recognize_batch_multi: LPT shard, one host thread per replica, host gather) -- on the one GPU of
the test box with two replicas on device 0, against the single-handle path."""
import numpy as np
import pytest

from speechrecognition_amd import capi, synth

pytestmark = pytest.mark.gpu

TDP = (3.0, 0.0, 30.0)


def _setup(tmp_path, W=60, M=4, D=39, seed=11):
    lex = synth.make_lexicon(W, 3, 1)
    spec = synth.make_mixset(lex.n_states, M, D, seed=seed)
    mp = str(tmp_path / "m.mix")
    synth.write_mixset(mp, spec)
    return lex, spec, mp


def test_async_upload_equals_sync_upload(tmp_path):
    """Every entry point on an asynchronously fed corpus, called while the feeder may still be copying (40 MB = 5 pieces):
    scores, words + traceback, alignments, path scores and EM statistics equal the synchronously uploaded corpus'."""
    lex, spec, mp = _setup(tmp_path)
    feats, off = synth.make_batch(600, 300, 520, 39, seed=3)      # ~250k frames x 156 B = 39 MB
    word_off, automaton, sil = lex.flatten()
    rng = np.random.default_rng(4)
    auts = []
    for u in range(600):
        a = [sil]
        for w in rng.integers(1, lex.n_words, size=3):
            a += list(automaton[word_off[w]:word_off[w + 1]]) + [sil]
        auts.append(np.asarray(a, np.uint16))
    with capi.Model.from_mixset(mp, 39) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, TDP, sil)
        ref = m.upload(feats, off)
        want_rec = ref.recognize(lexh, 120.0, 10.0, traceback=True)
        want_al = ref.align(auts, TDP, sil, pruning_threshold=80.0)
        want_ps = ref.path_scores(want_al[0])
        want_acc = ref.accumulate(want_al[0])
        want_sc = ref.score(capi.GMM_PREFILTER)[:5000]
        ref.close()
        for first in ("recognize", "align", "path", "accumulate", "score"):
            c = m.upload(feats, off, asynchronous=True)     # returns at once; the call below races the feeder
            if first == "recognize":
                got = c.recognize(lexh, 120.0, 10.0, traceback=True)
                assert np.array_equal(got[0], want_rec[0]) and np.array_equal(got[1], want_rec[1])
                for g, w in zip(got[2], want_rec[2]):
                    assert np.array_equal(g.view(np.uint8), w.view(np.uint8))
            elif first == "align":
                got = c.align(auts, TDP, sil, pruning_threshold=80.0)
                assert np.array_equal(got[0], want_al[0]) and np.array_equal(got[1], want_al[1])
            elif first == "path":
                assert np.array_equal(c.path_scores(want_al[0]).view(np.uint64), want_ps.view(np.uint64))
            elif first == "accumulate":
                for g, w in zip(c.accumulate(want_al[0]), want_acc):
                    assert np.array_equal(g.view(np.uint64), w.view(np.uint64))
            else:
                assert np.array_equal(c.score(capi.GMM_PREFILTER)[:5000].view(np.uint64), want_sc.view(np.uint64))
            c.wait()
            c.close()
        # destroying a corpus whose feeder is still running must be safe (it joins the thread first)
        c = m.upload(feats, off, asynchronous=True)
        c.close()
        # the one-shot boundary call feeds asynchronously by itself
        w, o = m.recognize_batch(lexh, feats, off, 120.0, 10.0)
        assert np.array_equal(w, want_rec[0]) and np.array_equal(o, want_rec[1])
        lexh.close()


def test_async_upload_edge_cases(tmp_path):
    lex, spec, mp = _setup(tmp_path, W=5)
    word_off, automaton, sil = lex.flatten()
    with capi.Model.from_mixset(mp, 39) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, TDP, sil)
        # empty corpus, single frame
        for feats, off in ((np.zeros((0, 39), np.float32), np.array([0], np.uint64)),
                           (synth.make_features(1, 39, 5), np.array([0, 1], np.uint64))):
            c = m.upload(feats, off, asynchronous=True)
            c.wait()
            w, o = c.recognize(lexh, 100.0, 10.0)
            s = m.upload(feats, off)
            w2, o2 = s.recognize(lexh, 100.0, 10.0)
            assert np.array_equal(w, w2) and np.array_equal(o, o2)
            c.close(); s.close()
        with pytest.raises(capi.SrError):  # over-long utterance: rejected before any thread starts
            m.upload(np.zeros((70000, 39), np.float32), np.array([0, 70000], np.uint64), asynchronous=True)
        lexh.close()


@pytest.mark.parametrize("n_replicas", [2, 3])
def test_multi_device_driver_equals_single_handle(tmp_path, n_replicas):
    """sr_recognize_batch_multi with `n_replicas` (model, lexicon) replicas -- all on device 0 here, one per GPU in production --
    against sr_recognize_batch on one handle: same words in corpus order; shard loads as sr_shard_utterances deals them."""
    lex, spec, mp = _setup(tmp_path, W=120, M=3)
    feats, off = synth.make_batch(301, 40, 160, 39, seed=9)
    rng = np.random.default_rng(10)
    for u in range(0, 301, 5):  # some utterances drawn from the model: several words each
        x = synth.sample_utterance(spec, lex, rng.integers(1, lex.n_words, size=3), seed=u)
        n = min(len(x), int(off[u + 1] - off[u]))
        feats[int(off[u]):int(off[u]) + n] = x[:n]
    word_off, automaton, sil = lex.flatten()
    models = [capi.Model.from_mixset(mp, 39) for _ in range(n_replicas)]
    lexica = [m.lexicon(word_off, automaton, lex.silence_idx, TDP, sil) for m in models]
    try:
        want_w, want_o = models[0].recognize_batch(lexica[0], feats, off, 150.0, 10.0)
        w, o, load = capi.recognize_batch_multi(models, lexica, feats, off, 150.0, 10.0)
        assert np.array_equal(w, want_w) and np.array_equal(o, want_o) and len(w) > 301
        shard_of, want_load = capi.shard_utterances(off, n_replicas)
        assert np.array_equal(load, want_load) and int(load.sum()) == int(off[-1])
        assert load.max() / load.mean() < 1.01      # LPT on 301 utterances
        # an error in one device thread comes back as an error of the call
        with pytest.raises(capi.SrError, match="share one handle"):
            capi.recognize_batch_multi([models[0], models[0]], [lexica[0], lexica[0]], feats, off, 150.0, 10.0)
    finally:
        for l in lexica:
            l.close()
        for m in models:
            m.close()


def test_corpus_buffers_recycled_across_calls(tmp_path):
    """A model keeps the device buffers of its last destroyed corpus for the next one (round 3: the nine allocations per
    sr_recognize_batch call were a millisecond of the boundary step).  Batches of different sizes through ONE model, blocking and
    asynchronous uploads interleaved, a second corpus alive at the same time: every result equals a fresh model's."""
    lex, spec, mp = _setup(tmp_path, W=40, M=3)
    word_off, automaton, sil = lex.flatten()
    batches = [synth.make_batch(n, 20, 90, 39, seed=20 + i) for i, n in enumerate((120, 7, 300, 1, 64))]
    want = []
    for feats, off in batches:
        with capi.Model.from_mixset(mp, 39) as m:
            lexh = m.lexicon(word_off, automaton, lex.silence_idx, TDP, sil)
            want.append(m.recognize_batch(lexh, feats, off, 150.0, 10.0))
            lexh.close()
    with capi.Model.from_mixset(mp, 39) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, TDP, sil)
        for rnd in range(2):
            for i, (feats, off) in enumerate(batches):
                if (i + rnd) % 2:
                    w, o = m.recognize_batch(lexh, feats, off, 150.0, 10.0)          # asynchronous feeder, corpus created + destroyed
                else:
                    c = m.upload(feats, off)                                           # blocking upload
                    keep = m.upload(batches[1][0], batches[1][1], asynchronous=True)   # a second corpus at the same time
                    w, o = c.recognize(lexh, 150.0, 10.0)
                    w2, o2 = keep.recognize(lexh, 150.0, 10.0)
                    assert np.array_equal(w2, want[1][0]) and np.array_equal(o2, want[1][1])
                    c.close(); keep.close()
                assert np.array_equal(w, want[i][0]) and np.array_equal(o, want[i][1]), (rnd, i)
        lexh.close()


def test_spare_buffers_are_capped_trimmed_and_survive_the_wrong_destroy_order(tmp_path):
    """the buffers a model parks for its next corpus (a) can be released (sr_model_trim), (b) are only kept up to
    SRGPU_SPARE_MB, and (c) a corpus destroyed AFTER its model -- against srgpu.h -- must not reach into the freed model."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        assert hip.hipDeviceSynchronize() == 0
        fr, tot = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot)) == 0
        return fr.value

    lex, spec, mp = _setup(tmp_path, W=20, M=2)
    word_off, automaton, sil = lex.flatten()
    feats, off = synth.make_batch(400, 300, 500, 39, seed=31)   # ~160k frames x 156 B = 25 MB of features
    with capi.Model.from_mixset(mp, 39) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, TDP, sil)
        want = m.recognize_batch(lexh, feats, off, 150.0, 10.0)          # creates and destroys a corpus: its buffers are parked
        parked = free_bytes()
        m.trim()
        trimmed = free_bytes()
        assert trimmed - parked >= feats.nbytes, (parked, trimmed)        # the feature buffer (at least) came back
        got = m.recognize_batch(lexh, feats, off, 150.0, 10.0)            # and the model works on without it
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        lexh.close()
    # (c): the corpus outlives its model
    m = capi.Model.from_mixset(mp, 39)
    c = m.upload(feats[: int(off[3])], off[:4])
    m.close()
    c.close()
    assert free_bytes() > 0
