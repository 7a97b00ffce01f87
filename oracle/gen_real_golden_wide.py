#!/usr/bin/env python3
"""A WIDER real-speech pin (round 5, VERDICT r4 "Next" #9): 64 more test utterances of the reference's own SieTill test list,
decoded by the REFERENCE (Recognizer::recognizeSequence_pruned, Recognizer.cpp:103-232, through oracle/_ref) with the two models
the reference's trainer produced for tests/golden_real/sietill_real.npz (oracle/gen_real_golden.py; the model bytes are taken from
that fixture, nothing is trained again).  Build container only (/root/reference).

Kept small: per utterance the RAW 12-dimensional cepstra as they sit in the .mm2 file (IO.cpp:48-69) -- the tests put them through
sr::FeaturePostProcessor (include/sr_sietill.hpp: delta, delta-delta, normalisation) themselves --, the reference's words at the tight
and the wide beam for both models, and the SHA-256 of the three traceback arrays of the CPU restatement (whose words were checked
against the reference's here, utterance by utterance; the reference keeps its traceback in a local and does not hand it out).
-> tests/golden_real/sietill_real_wide.npz"""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from oracle.gen_real_golden import FEATS, REF, TDP, base_config  # noqa: E402
from speechrecognition_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden_real")
N_UTTS = 64
BEAMS = (("wide", 200.0, 80.0), ("tight", 40.0, 30.0))  # (tag, am-threshold, word-penalty): the first fixture's two settings
POOL = {"mixture": po.POOL_MIXTURE, "none": po.POOL_NONE}


def tb_digest(score, word, bkp):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(score, dtype="<f8").tobytes())
    h.update(np.ascontiguousarray(word, dtype="<u2").tobytes())
    h.update(np.ascontiguousarray(bkp, dtype="<u2").tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8)


def main():
    assert po.reference_available()
    first = np.load(os.path.join(OUT, "sietill_real.npz"))
    taken = set(str(n) for n in first["names"])
    test_all = json.load(open(os.path.join(REF, "corpora", "corpus_test.json")))["segments"]
    rng = np.random.default_rng(12)
    pick = [test_all[i] for i in rng.permutation(len(test_all))
            if test_all[i]["name"] not in taken and os.path.exists(FEATS + test_all[i]["name"] + ".mm2")][:N_UTTS]
    pick.sort(key=lambda s: s["name"])
    print(f"{len(pick)} test utterances (none of the first fixture's 16)")
    tmp = tempfile.mkdtemp()
    ej = os.path.join(tmp, "test.json")
    json.dump({"segments": pick}, open(ej, "w"))

    # the reference's own corpus reader: processed features (what the reference decodes) + the transcriptions
    L = C.CDLL(po.REF_SO)
    L.ref_corpus_open.restype = C.c_void_p
    L.ref_corpus_open.argtypes = [C.c_char_p]
    for f in ("ref_corpus_size", "ref_corpus_dim"):
        getattr(L, f).restype, getattr(L, f).argtypes = C.c_size_t, [C.c_void_p]
    L.ref_corpus_frames.restype, L.ref_corpus_frames.argtypes = C.c_size_t, [C.c_void_p, C.c_size_t]
    L.ref_corpus_get.restype, L.ref_corpus_get.argtypes = C.c_size_t, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.ref_corpus_close.argtypes = [C.c_void_p]
    cp = os.path.join(tmp, "test-corpus.json")
    json.dump(base_config(ej, {"action": "recognize"}), open(cp, "w"))
    h = L.ref_corpus_open(cp.encode())
    dim = L.ref_corpus_dim(h)
    assert L.ref_corpus_size(h) == len(pick)
    feats, refs, raws = [], [], []
    for s in range(len(pick)):
        T = L.ref_corpus_frames(h, s)
        f = np.zeros((T, dim), dtype=np.float32)
        w = np.zeros(64, dtype=np.uint64)
        nw = L.ref_corpus_get(h, s, f.ctypes.data, w.ctypes.data)
        feats.append(f)
        refs.append(w[:nw].astype(np.uint32))
        raw = np.fromfile(FEATS + pick[s]["name"] + ".mm2", dtype="<f4")
        assert raw.size == 12 * T
        raws.append(raw)
    L.ref_corpus_close(h)

    lex = synth.sietill_lexicon()
    out = dict(dim=dim, names=np.asarray([p["name"] for p in pick]), tdp=np.asarray(TDP),
               raw_mm2=np.concatenate(raws), raw_off=np.cumsum([0] + [r.size for r in raws]).astype(np.uint64),
               frame_off=np.cumsum([0] + [len(f) for f in feats]).astype(np.uint64),
               ref_flat=np.concatenate(refs), ref_off=np.cumsum([0] + [len(r) for r in refs]).astype(np.uint32),
               # checksum of the processed features the reference decoded (the tests' own post-processing must reproduce it)
               feats_sha256=np.frombuffer(hashlib.sha256(np.concatenate(feats).astype("<f4").tobytes()).digest(), dtype=np.uint8))
    for pname, pool in POOL.items():
        mix = os.path.join(tmp, f"{pname}.mix")
        open(mix, "wb").write(first[f"model_{pname}"].tobytes())
        for tag, beam, wp in BEAMS:
            rc = os.path.join(tmp, f"rec-{pname}-{tag}.json")
            synth.write_config(rc, mix, tdp=TDP, am_threshold=beam, word_penalty=wp)
            ref = po.Reference(rc, dim, lex, pooling=pool)
            orc = po.Oracle(mix, dim, lex, tdp=TDP, am_threshold=beam, word_penalty=wp, pooling=pool)
            words, digests, errs, neg_frames = [], [], np.zeros(4, np.int64), 0
            for f, r in zip(feats, refs):
                w = ref.decode(f)
                ow, (ts, tw, tb) = orc.decode(f, traceback=True)
                assert np.array_equal(w, ow), (pname, tag)
                words.append(np.asarray(w, np.uint32))
                digests.append(tb_digest(ts, tw, tb))
                errs += ref.edit_distance(r, w).astype(np.int64)
            key = f"{pname}_{tag}"
            out[f"{key}_beam"], out[f"{key}_wp"] = beam, wp
            out[f"{key}_words"] = np.concatenate(words) if sum(map(len, words)) else np.zeros(0, np.uint32)
            out[f"{key}_word_off"] = np.cumsum([0] + [len(w) for w in words]).astype(np.uint32)
            out[f"{key}_tb_sha256"] = np.stack(digests)
            out[f"{key}_errors"] = errs
            print(f"{key}: WER {100.0 * errs[0] / max(1, len(out['ref_flat'])):.1f}% (S/I/D {errs[1]}/{errs[2]}/{errs[3]})")
            ref.close(); orc.close()
    path = os.path.join(OUT, "sietill_real_wide.npz")
    np.savez_compressed(path, **out)
    print("wrote", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
