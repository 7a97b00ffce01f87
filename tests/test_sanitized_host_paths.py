"""Host-side paths that parse untrusted bytes or index by caller-supplied offsets, exercised without a GPU.  They run in the
ordinary CPU suite and again under ASan + UBSan (tools/sanitize_host.py, tests/test_sanitizers_cpu.py), where a read past a
buffer is a failure, not luck.  Reference bugs this code must stay compatible with but not share: the 2-float over-read
of density_score_sse (Mixtures.cpp:653) and the T-sized cost arrays indexed by position (Alignment.cpp:62-63)."""
import os
import subprocess

import numpy as np
import pytest

from speechrecognition_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from speechrecognition_amd import build, capi
    build.build()
    return capi


def test_every_truncation_of_a_model_file_is_an_error_not_a_crash(capi, tmp_path):
    """MIXSET v2 reader (Mixtures.cpp:748-830): every proper prefix of a valid file, and the file with each count field
    blown up, must come back with a status.  The whole file itself parses (then: SR_ENODEV here, a model on a GPU box)."""
    lex = synth.make_lexicon(3, 3, 1)
    spec = synth.make_mixset(lex.n_states, 2, 5, seed=4)
    good = tmp_path / "good.mix"
    synth.write_mixset(str(good), spec)
    blob = good.read_bytes()
    p = tmp_path / "cut.mix"
    for n in list(range(0, 64)) + list(range(64, len(blob), 37)) + [len(blob) - 1]:
        p.write_bytes(blob[:n])
        with pytest.raises(capi.SrError) as e:
            capi.Model.from_mixset(str(p), 5)
        assert e.value.code == -1, (n, str(e.value))
    for pos in range(16, min(len(blob) - 4, 4000), 4):   # every aligned u32 in the file set to 0xFFFFFFF0
        q = bytearray(blob)
        q[pos:pos + 4] = (0xFFFFFFF0).to_bytes(4, "little")
        p.write_bytes(bytes(q))
        try:
            capi.Model.from_mixset(str(p), 5).close()
        except capi.SrError as e:
            assert e.code in (-1, -3, -4, -5), (pos, str(e))   # a message, never an abort
    try:
        capi.Model.from_mixset(str(good), 5).close()
    except capi.SrError as e:
        assert e.code == -3


def test_mixset_writer_round_trip_and_unreferenced_rows(capi, tmp_path):
    """sr_mixset_write == MixtureModel::write (Mixtures.cpp:834-878): unreferenced accumulator rows are dropped and the rest
    renumbered; the bytes parse back to the referenced statistics."""
    D, S = 3, 2
    dens_off = np.array([0, 2, 3], np.uint32)
    dens_mean = np.array([4, 1, 2], np.uint32)     # rows 0 and 3 are unreferenced
    dens_var = np.array([0, 0, 2], np.uint32)      # tied variance row 0; row 1 unreferenced
    rng = np.random.default_rng(5)
    ma, mw = rng.normal(size=(5, D)), rng.uniform(1, 9, size=5)
    va, vw = rng.uniform(1, 2, size=(3, D)), rng.uniform(1, 9, size=3)
    path = tmp_path / "w.mix"
    capi.mixset_write(str(path), D, dens_off, dens_mean, dens_var, (ma, mw, va, vw))
    b = path.read_bytes()
    assert b[:8] == b"MIXSET\0\0"
    ver, dim, n_mean = np.frombuffer(b, "<u4", 3, 8)
    assert (ver, dim, n_mean) == (2, D, 3)
    rec = 4 + 8 * D + 8
    rows = [np.frombuffer(b, "<f8", D + 1, 20 + i * rec + 4) for i in range(3)]
    kept = sorted(set(dens_mean.tolist()))
    for r, src in zip(rows, kept):
        assert np.array_equal(r[:D], ma[src]) and r[D] == mw[src]
    with pytest.raises(capi.SrError):   # a density that points past the accumulators
        capi.mixset_write(str(path), D, dens_off, np.array([4, 1, 7], np.uint32), dens_var, (ma, mw, va, vw))


def test_shard_and_driver_argument_checks(capi):
    """sr_shard_utterances on ragged inputs (empty utterances, one shard, more shards than utterances) and the multi-device
    driver's argument errors, which run before any device is touched."""
    for lens, k in (([], 1), ([0, 0, 0], 2), ([5], 8), ([3, 0, 7, 7, 1], 3), ([1] * 100, 7)):
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        shard_of, load = capi.shard_utterances(off, k)
        assert len(shard_of) == len(lens) and int(load.sum()) == int(off[-1])
        assert (shard_of < k).all()
    import ctypes as C
    sp = capi.SearchParams(100.0, 10.0, capi.GMM_PREFILTER, 0)
    off = np.array([0, 4], np.uint64)
    out = np.zeros(4, np.uint32); woff = np.zeros(2, np.uint64)
    rc = capi.lib().sr_recognize_batch_multi(None, None, 0, C.byref(sp), None, off.ctypes.data, 1, out.ctypes.data, woff.ctypes.data, None)
    assert rc == -1


def _driver():
    asan = os.environ.get("SR_ASAN_DRIVER")
    if asan:
        return asan
    from tests.test_host_mirror import DRIVER
    from speechrecognition_amd import build
    build.build()
    src = os.path.join(ROOT, "tests", "cpp", "host_mirror_driver.cpp")
    hdr = os.path.join(ROOT, "include", "sr_sietill.hpp")
    if not os.path.exists(DRIVER) or os.path.getmtime(DRIVER) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), src, "-o", DRIVER,
                               "-L" + os.path.join(ROOT, "speechrecognition_amd"), "-lsrgpu",
                               "-Wl,-rpath,$ORIGIN/../../speechrecognition_amd", "-Wl,-rpath,/opt/rocm/lib"])
    return DRIVER


def test_host_mirror_cpu_modes(tmp_path):
    """include/sr_sietill.hpp through tests/cpp/host_mirror_driver: lexicon + TDP, edit distance on ragged and empty
    sequences, .mm2 reader + delta/normalisation post-processing + the alignment dump round trip (IO.cpp:48-69,
    SignalAnalysis.cpp:320-399, Alignment.cpp:303-342) on short files: 1, 2, 3 and 50 frames."""
    drv = _driver()
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}   # an ASan-built driver brings its own runtime
    out = subprocess.check_output([drv, "lexicon"], text=True, env=env).splitlines()
    assert out[0].split()[:2] == ["12", "106"]
    ed = tmp_path / "ed.txt"
    ed.write_text("1 2 3 | 1 2 3\n | 1 2\n4 5 | \n | \n7 | 8\n")
    got = np.asarray(subprocess.check_output([drv, "edit", str(ed)], text=True, env=env).split(), dtype=np.int64).reshape(-1, 4)
    # (rows with an empty reference follow the reference's stale row-0 counter: pinned by tests/golden/edit_distance.npz, not here)
    assert got[0].tolist() == [0, 0, 0, 0] and got[2].tolist() == [2, 0, 0, 2] and got[4].tolist() == [1, 1, 0, 0] and len(got) == 5
    rng = np.random.default_rng(8)
    for frames in (1, 2, 3, 50):
        mm2 = tmp_path / f"f{frames}.mm2"
        rng.normal(size=(frames, 12)).astype("<f4").tofile(str(mm2))
        o = tmp_path / f"f{frames}.f32"
        line = subprocess.check_output([drv, "features", str(mm2), "-", str(o)], text=True, env=env)
        assert line.startswith(f"{frames} frames x 25") and "dump ok" in line
        assert os.path.getsize(o) == frames * 25 * 4
