"""Diagnostic: per-phase cycle shares of the decode kernel (needs a -DSR_DECODE_STAMPS build of libsrgpu)."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechrecognition_amd import capi, synth
lex = synth.make_lexicon(1333, 3, 1)
spec = synth.make_mixset(lex.n_states, 4, 39, seed=23)
mp = os.path.join(tempfile.mkdtemp(), "m.mix"); synth.write_mixset(mp, spec)
feats, off = synth.make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 200, 400, 39, seed=7)
word_off, automaton, sil = lex.flatten()
m = capi.Model.from_mixset(mp, 39)
lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil)
c = m.upload(feats, off)
for _ in range(2):
    words, woff, (tbs, tbw, tbb) = c.recognize(lexh, 200.0, 10.0, capi.GMM_MFMA, traceback=True)
names = ["top+gather issue", "phase A", "wave reductions", "barrier 1", "combine+phase C", "barrier 2"]
sums = np.zeros(6); frames = 0
for u in range(len(off) - 1):
    b = int(off[u]) + u
    sums += tbs[b + 1:b + 7]; frames += int(off[u + 1] - off[u])
print("cycles per frame (wave 0):", {n: round(v / frames) for n, v in zip(names, sums)}, "total", round(sums.sum() / frames))
