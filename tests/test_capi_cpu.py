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
