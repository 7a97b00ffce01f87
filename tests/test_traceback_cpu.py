"""The guarded traceback walk (speechrecognition_amd/csrc/traceback.h) from the host side: sr_traceback_words runs the same
`walk_traceback` the three search kernels and sr_traceback_corpus run on the device.  Recognizer.cpp:222-231."""
import numpy as np
import pytest

from tests.util import Case

FIXTURES = ["cfg1_monophone", "sietill_lexicon_d25", "sietill_lexicon_d25_tightbeam", "ragged_words", "two_frames", "one_frame"]


@pytest.fixture(scope="module")
def capi():
    from speechrecognition_amd import build, capi
    build.build()
    return capi


@pytest.mark.parametrize("name", FIXTURES)
def test_host_walk_reproduces_the_reference_words(name, capi, oracle_lib, tmp_path):
    """traceback[] of the oracle (== the reference's, tests/test_oracle_golden.py) -> the reference's golden word sequence."""
    c = Case(name, tmp_path)
    o = c.oracle(oracle_lib)
    words, (_, tw, tb) = o.decode(c.feats, traceback=True)
    o.close()
    assert np.array_equal(words, c.z["words"])
    got = capi.traceback_words(tw, tb, c.lex.silence_idx, c.lex.n_words)
    assert np.array_equal(got, c.z["words"])


def _valid(T=50, n_words=7, seed=3):
    rng = np.random.default_rng(seed)
    tw = rng.integers(0, n_words, T + 1).astype(np.uint16)
    tb = np.array([0] + [int(rng.integers(0, t)) for t in range(1, T + 1)], np.uint16)  # bkp < t everywhere
    return tw, tb


def test_corrupted_traceback_yields_the_status(capi):
    tw, tb = _valid()
    T, nW = len(tw) - 1, 7
    ok = capi.traceback_words(tw, tb, 0, nW)
    # the walk by hand
    want, t = [], T
    while t > 0:
        if tw[t] != 0:
            want.append(int(tw[t]))
        t = int(tb[t])
    assert list(ok) == want[::-1]
    # a back pointer that does not fall: the unguarded loop would never reach t = 0 and push a word per round (the
    # round-2 fault)
    for bad_bkp in (T, T + 5, 65535):
        b2 = tb.copy(); b2[T] = bad_bkp
        with pytest.raises(capi.SrError) as e:
            capi.traceback_words(tw, b2, 0, nW)
        assert e.value.code == capi.SR_ECORRUPT
    # ... also in the middle of the chain
    mid = int(tb[T])
    if mid > 0:
        b3 = tb.copy(); b3[mid] = mid
        with pytest.raises(capi.SrError) as e:
            capi.traceback_words(tw, b3, 0, nW)
        assert e.value.code == capi.SR_ECORRUPT
    # a word outside the lexicon (a stale slot id)
    w2 = tw.copy(); w2[T] = nW
    with pytest.raises(capi.SrError) as e:
        capi.traceback_words(w2, tb, 0, nW)
    assert e.value.code == capi.SR_ECORRUPT
    # entries the walk never visits may hold anything
    visited, t = set(), T
    while t > 0:
        visited.add(t); t = int(tb[t])
    w3, b3 = tw.copy(), tb.copy()
    for t in range(1, T + 1):
        if t not in visited:
            w3[t], b3[t] = 65535, 65535
    assert np.array_equal(capi.traceback_words(w3, b3, 0, nW), ok)


def test_empty_and_16_bit_edges(capi):
    assert len(capi.traceback_words(np.zeros(1, np.uint16), np.zeros(1, np.uint16), 0, 3)) == 0  # T = 0
    # T = 65535 with bkp = t - 1 everywhere: the longest chain 16-bit back pointers can express, one word per frame
    T = 65535
    tw = np.full(T + 1, 1, np.uint16)
    tb = np.arange(-1, T, dtype=np.int64).clip(0).astype(np.uint16)
    got = capi.traceback_words(tw, tb, 0, 2)
    assert len(got) == T and (got == 1).all()
