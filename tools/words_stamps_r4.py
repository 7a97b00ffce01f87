"""Diagnostic: per-phase s_memtime sums of decode_words_kernel's frame loop, waves 0..3 (needs tools/make_words_stamps_variant.py's
library: SRGPU_LIB=.../libsrgpu_wstamps.so).  usage: python tools/words_stamps_r4.py [cfg2|cfg3]
cfg2 = BASELINE configs[1] (334 words, one 10 000-frame utterance: one workgroup on one CU), cfg3 = configs[2] (1334 words, 1000 utterances)."""
import os, sys, tempfile
import ctypes as C
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from speechrecognition_amd import capi, synth
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
if which == "cfg2":
    lex = synth.make_lexicon(333, 3, 1); M = 8
    feats = synth.make_features(10000, 39, seed=4); off = np.array([0, 10000], np.uint64)
else:
    lex = synth.make_lexicon(1333, 3, 1); M = 4
    feats, off = synth.make_batch(1000, 200, 400, 39, seed=7)
spec = synth.make_mixset(lex.n_states, M, 39, seed=23)
mp = os.path.join(tempfile.mkdtemp(), "m.mix"); synth.write_mixset(mp, spec)
word_off, automaton, sil = lex.flatten()
m = capi.Model.from_mixset(mp, 39)
lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil)
print("network:", lexh.describe())
c = m.upload(feats, off)
for _ in range(2):
    words = np.zeros(c.n_frames, np.uint32); woff = np.zeros(c.n_utts + 1, np.uint64)
    n = c.n_frames + c.n_utts
    tbs, tbw, tbb = np.zeros(n, np.float64), np.zeros(n, np.uint16), np.zeros(n, np.uint16)
    sp = capi.SearchParams(200.0, 10.0, capi.GMM_MFMA, 0)
    rc = capi.lib().sr_recognize_corpus(m.h, c.h, lexh.h, C.byref(sp), words.ctypes.data, woff.ctypes.data, tbs.ctypes.data, tbw.ctypes.data, tbb.ctypes.data)
    assert rc == 0, capi.lib().sr_last_error()
names = ["loop back edge", "issue row t+1 (LDS-DMA) + boundary candidates", "A: emission reads + hypotheses", "B: DPP row minima + LDS atomics (incl. their wait)",
         "vmcnt(0): next row landed", "barrier", "C: read minima, limit, tie patch", "C: prune, word ends, resets"]
for wv in range(4):
    sums = np.zeros(8); frames = 0
    for u in range(len(off) - 1):
        b = int(off[u]) + u
        sums += tbs[b + 1 + 8 * wv: b + 9 + 8 * wv]; frames += int(off[u + 1] - off[u])
    print(f"wave {wv}: s_memtime ticks per frame:", {nm: round(v / frames, 1) for nm, v in zip(names, sums)}, "total", round(sums.sum() / frames, 1))
