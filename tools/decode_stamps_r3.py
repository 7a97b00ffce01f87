"""Diagnostic: per-phase cycle sums of decode_fast_kernel waves 0/5/10/15 (needs the -DSR_DEC_STAMPS variant)."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from speechrecognition_amd import capi, synth
lex = synth.make_lexicon(1333, 3, 1)
spec = synth.make_mixset(lex.n_states, 4, 39, seed=23)
mp = os.path.join(tempfile.mkdtemp(), "m.mix"); synth.write_mixset(mp, spec)
feats, off = synth.make_batch(1000, 200, 400, 39, seed=7)
word_off, automaton, sil = lex.flatten()
m = capi.Model.from_mixset(mp, 39)
lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil)
c = m.upload(feats, off)
import ctypes as C
for _ in range(2):
    words = np.zeros(c.n_frames, np.uint32); woff = np.zeros(c.n_utts + 1, np.uint64)
    n = c.n_frames + c.n_utts
    tbs, tbw, tbb = np.zeros(n, np.float64), np.zeros(n, np.uint16), np.zeros(n, np.uint16)
    sp = capi.SearchParams(200.0, 10.0, capi.GMM_MFMA, 0)
    rc = capi.lib().sr_recognize_corpus(m.h, c.h, lexh.h, C.byref(sp), words.ctypes.data, woff.ctypes.data, tbs.ctypes.data, tbw.ctypes.data, tbb.ctypes.data)
names = ["top", "flush", "A g0 loads(wait)", "A g0 compute", "A g1 loads(wait)", "A g1 compute", "A rest (uniform waves: all of A)", "B", "barrier 1", "C read", "C", "barrier 2"]
for wi, wv in enumerate((0, 7, 10, 15)):
    sums = np.zeros(12); frames = 0
    for u in range(len(off) - 1):
        b = int(off[u]) + u
        sums += tbs[b + 1 + 12 * wi: b + 13 + 12 * wi]; frames += int(off[u + 1] - off[u])
    print(f"wave {wv}: cycles per frame:", {n: int(round(v / frames)) for n, v in zip(names, sums)}, "total", int(round(sums.sum() / frames)))
