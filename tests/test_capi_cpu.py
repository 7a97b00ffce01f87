"""CPU suite for the boundary: libsrgpu.so builds, loads and exports every symbol include/srgpu.h
declares, and the product path fails loudly without a GPU (no CPU fallback, no oracle behind it)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from speechrecognition_amd import build
    return build.build()


def test_header_symbols_are_exported(built_lib):
    hdr = open(os.path.join(ROOT, "include", "srgpu.h")).read()
    declared = set(re.findall(r"SR_API\s+(?:const\s+char\*|int)\s+(sr_\w+)\s*\(", hdr))
    assert len(declared) >= 19
    from speechrecognition_amd import capi
    assert declared == set(capi.SYMBOLS)
    L = ctypes.CDLL(built_lib)
    for sym in declared:
        assert hasattr(L, sym), sym


def test_library_does_not_link_the_oracle(built_lib):
    out = subprocess.run(["ldd", built_lib], capture_output=True, text=True).stdout
    assert "oracle" not in out and "sietill" not in out
    src = os.path.join(ROOT, "speechrecognition_amd")
    for dirpath, _, files in os.walk(src):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in text and "sr_oracle" not in text and "liboracle" not in text, f


def test_fails_loudly_without_gpu(built_lib, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from speechrecognition_amd import capi, synth
    lex = synth.make_lexicon(2, 3, 1)
    mp = str(tmp_path / "m.mix")
    synth.write_mixset(mp, synth.make_mixset(lex.n_states, 2, 39, seed=1))
    with pytest.raises(capi.SrError) as e:
        capi.Model.from_mixset(mp, 39)
    assert e.value.code == -3  # SR_ENODEV: no silent CPU path


def test_mixset_loader_rejects_malformed_files(built_lib, tmp_path):
    from speechrecognition_amd import capi
    p = tmp_path / "bad.mix"
    p.write_bytes(b"NOTAMIX\0" + b"\0" * 32)
    with pytest.raises(capi.SrError, match="Invalid magic header"):
        capi.Model.from_mixset(str(p), 39)
    p.write_bytes(b"MIXSET\0\0" + np.array([2, 38], dtype="<u4").tobytes())
    with pytest.raises(capi.SrError, match="Invalid dimension"):
        capi.Model.from_mixset(str(p), 39)


def _mixset_header(dim):
    return b"MIXSET\0\0" + np.array([2, dim], dtype="<u4").tobytes()


def test_absurd_counts_in_a_model_file_are_errors_not_aborts(built_lib, tmp_path):
    """include/srgpu.h: "No exceptions cross this boundary".  A corrupt count used to size a std::vector straight from
    the file (4 Gi accumulators x 39 doubles -> std::bad_alloc / std::length_error -> std::terminate -> SIGABRT in the
    host process).  Counts are now checked against the bytes left in the file, and every entry point has a catch-all."""
    from speechrecognition_amd import capi
    p = tmp_path / "huge.mix"
    # (a) mean-accumulator count 0xFFFFFFFF with nothing behind it
    p.write_bytes(_mixset_header(39) + np.array([0xFFFFFFFF], dtype="<u4").tobytes())
    with pytest.raises(capi.SrError, match="Error reading features") as e:
        capi.Model.from_mixset(str(p), 39)
    assert e.value.code == -1
    # (b) both accumulator blocks empty, density count absurd
    p.write_bytes(_mixset_header(39) + np.array([0, 0, 0xFFFFFFF0], dtype="<u4").tobytes())
    with pytest.raises(capi.SrError, match="Error reading mean_idx"):
        capi.Model.from_mixset(str(p), 39)
    # (c) absurd mixture count, (d) absurd per-mixture density count
    p.write_bytes(_mixset_header(39) + np.array([0, 0, 0, 0xFFFFFFF0], dtype="<u4").tobytes())
    with pytest.raises(capi.SrError, match="Error reading density count for mixture"):
        capi.Model.from_mixset(str(p), 39)
    p.write_bytes(_mixset_header(39) + np.array([0, 0, 0, 1, 0xFFFFFFF0], dtype="<u4").tobytes())
    with pytest.raises(capi.SrError, match="Error reading density idx"):
        capi.Model.from_mixset(str(p), 39)


def test_exception_barrier_turns_bad_alloc_into_a_status(built_lib):
    """Forces a host allocation that cannot succeed through an entry point that needs no GPU (sr_mixset_write): statistics with
    2^32-1 accumulator rows x 63 dimensions (2 TB of doubles).  The call must come back with SR_ENOMEM / SR_ELIMIT and a
    message -- in a child process, so that a regression (SIGABRT) fails this test instead of killing pytest."""
    import sys
    code = r"""
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
from speechrecognition_amd import capi
L = capi.lib()
off = np.zeros(2, np.uint32); one = np.zeros(1, np.uint32); d = np.zeros(64, np.float64)
out = C.c_void_p()
rc = L.sr_mixset_write(b"/tmp/never_written.mix", 63, 1, off.ctypes.data, 0xFFFFFFFF, 1, one.ctypes.data, one.ctypes.data,
                       d.ctypes.data, d.ctypes.data, d.ctypes.data, d.ctypes.data)
print(rc, L.sr_last_error().decode())
sys.exit(0 if rc in (-4, -5) else 3)
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "sr_mixset_write" in r.stdout


def test_build_stamps_follow_content_not_mtime(built_lib):
    """An object compiled while a header was being edited must be rebuilt: staleness is decided by a content hash taken
    before the compiler starts (speechrecognition_amd/build.py), never by mtime."""
    from speechrecognition_amd import build
    obj = os.path.join(build.CSRC, "build", "mixset.cpp.o")
    stamp = open(obj + ".stamp").read()
    assert len(stamp) == 64
    lib_stamp = open(build.LIB + ".stamp").read()
    os.utime(os.path.join(build.CSRC, "kernels.h"))           # newer mtime, same bytes: nothing to do
    before = os.path.getmtime(obj)
    build.build()
    assert os.path.getmtime(obj) == before and open(build.LIB + ".stamp").read() == lib_stamp


def test_shard_utterances_matches_the_python_helper(built_lib):
    """sr_shard_utterances (the C ABI's LPT deal, used by sr_recognize_batch_multi) == sharding.shard_utterances (used by
    bench.py's ranks): same shards, so an in-process multi-device run and a one-process-per-GPU run decode the same
    utterances on the same device index."""
    from speechrecognition_amd import capi, sharding
    rng = np.random.default_rng(3)
    for n_utts, n_shards in ((0, 3), (1, 4), (17, 2), (1000, 8), (257, 5), (64, 64)):
        lens = rng.integers(1, 400, size=n_utts)
        if n_utts > 10:
            lens[3:9] = 77  # ties
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        shard_of, load = capi.shard_utterances(off, n_shards)
        want = sharding.shard_utterances(off, n_shards)
        for r in range(n_shards):
            assert np.array_equal(np.nonzero(shard_of == r)[0], want[r]), (n_utts, n_shards, r)
            assert int(load[r]) == int(lens[want[r]].sum()) if n_utts else int(load[r]) == 0
    with pytest.raises(capi.SrError):
        capi.shard_utterances(np.array([0, 5], np.uint64), 0)


def test_abi_version_matches_the_header_and_the_binding(built_lib):
    """a struct of the ABI grew without a version to tell callers.  The header's SR_ABI_VERSION, the library's
    sr_abi_version() and the ctypes binding's constant are one number; capi.lib() refuses a library that disagrees."""
    from speechrecognition_amd import capi
    hdr = open(os.path.join(ROOT, "include", "srgpu.h")).read()
    want = int(re.search(r"#define\s+SR_ABI_VERSION\s+(\d+)", hdr).group(1))
    L = ctypes.CDLL(built_lib)
    assert L.sr_abi_version() == want == capi.SR_ABI_VERSION
    assert capi.lib().sr_abi_version() == want
