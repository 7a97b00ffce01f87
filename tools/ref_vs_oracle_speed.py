"""Restatement vs the compiled reference (oracle/_ref, this container only): single-thread decode speed of
Recognizer::recognizeSequence_pruned on a model the unmodified reference can load (< 65 536 densities).
usage: python tools/ref_vs_oracle_speed.py [words] [mix] [frames]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from speechrecognition_amd import synth
from oracle import pyoracle

W = int(sys.argv[1]) if len(sys.argv) > 1 else 333
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
pyoracle.build()
lex = synth.make_lexicon(W, 3, 1)
spec = synth.make_mixset(lex.n_states, M, 39, seed=23)
tmp = tempfile.mkdtemp()
mp = os.path.join(tmp, "m.mix")
synth.write_mixset(mp, spec)
cfgp = os.path.join(tmp, "c.json")
synth.write_config(cfgp, mp, am_threshold=200.0, word_penalty=10.0)
feats = synth.make_features(T, 39, seed=4)
o = pyoracle.Oracle(mp, 39, lex, am_threshold=200.0)
r = pyoracle.Reference(cfgp, 39, lex)
for name, dec in (("restatement", o.decode), ("reference", r.decode)):
    dec(feats[:200])
    t0 = time.perf_counter()
    w = dec(feats)
    dt = time.perf_counter() - t0
    print(f"{name:12s} {T/dt:10,.0f} frames/s  ({len(w)} words)")
print("equal words:", np.array_equal(o.decode(feats), r.decode(feats)))
