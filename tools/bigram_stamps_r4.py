"""Diagnostic: per-step s_memtime ticks per frame of bigram_kernel (register layout), thread 0 of every workgroup (needs the stamps variant of
the library, tools/make_bigram_stamps_variant.py; the stamps overwrite the output scores, so the words are garbage)."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from speechrecognition_amd import capi, synth
lex = synth.make_lexicon(2666, 3, 1, extra_states_last=1)
spec = synth.make_mixset(lex.n_states, 4, 39, seed=23)
mp = os.path.join(tempfile.mkdtemp(), "m.mix"); synth.write_mixset(mp, spec)
feats, off = synth.make_batch(512, 200, 400, 39, seed=7)
word_off, mixtures, _ = lex.flatten()
W = lex.n_words
rng = np.random.default_rng(99)
lm = np.empty((W, W), np.float32)
for h0 in range(0, W, 256):
    p = rng.dirichlet(np.ones(W), size=min(256, W - h0))
    lm[:, h0:h0 + p.shape[0]] = (-np.log(np.maximum(p, 1e-30))).T
tdp = np.array([[3.0, 0.0, 3.0, 150.0], [0.0001, 3.0, np.inf, 15.0]], np.float32)
m = capi.Model.from_mixset(mp, 39)
bg = m.bigram(word_off, mixtures, lex.silence_idx, lm, tdp)
c = m.upload(feats, off)
for _ in range(2):
    ow, osc, ot, o = c.recognize_bigram(bg, 200.0, capi.FLT_MAX, capi.GMM_MFMA)
names = ["list read + U bound (wg_min)", "1a silence-copy entries, skip test, keep scan", "1b staging + LM rows + entries + best start (wg_min)", "2 activation (scan, list appends)",
         "3 expand + emission (registers)", "3 wg_min best score", "4 prune (registers) + flags + barrier", "4 compaction (scan, list writes, barrier)",
         "5 positions + barrier", "5-6 merge + book entries", "final barrier"]
acc = np.zeros(11)
for u in range(len(off) - 1):
    a = int(o[u])
    acc += osc[a + 1:a + 12]
acc /= (len(off) - 1)
print({n: int(v) for n, v in zip(names, acc)}, "total", int(acc.sum()))
