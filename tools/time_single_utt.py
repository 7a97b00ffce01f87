"""BASELINE configs[1]: 1000 states x 8-mix, one 10 000-frame utterance -- wall time of sr_recognize_corpus (features
resident) and of the CPU oracle on the same input.  usage: python tools/time_single_utt.py [T] [words] [mix]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from speechrecognition_amd import capi, synth
from oracle import pyoracle

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 333
M = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lex = synth.make_lexicon(W, 3, 1)
spec = synth.make_mixset(lex.n_states, M, 39, seed=23)
mp = os.path.join(tempfile.mkdtemp(), "m.mix")
synth.write_mixset(mp, spec)
feats = synth.make_features(T, 39, seed=4)
off = np.array([0, T], np.uint64)
word_off, automaton, sil = lex.flatten()
with capi.Model.from_mixset(mp, 39) as m:
    lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil)
    c = m.upload(feats, off)
    c.recognize(lexh, 200.0, 10.0)
    m.profile(True)
    t0 = time.perf_counter()
    for _ in range(3):
        words, woff = c.recognize(lexh, 200.0, 10.0)
    dt = (time.perf_counter() - t0) / 3
    p = m.profile_read()
    print(f"GPU: {dt*1e3:.2f} ms per pass = {T/dt:,.0f} frames/s; scoring {p['gmm_ms']/3:.2f} ms, search {p['search_ms']/3:.2f} ms "
          f"({p['search_ms']/3/T*1e3:.2f} us/frame), {len(words)} words")
    c.close(); lexh.close()
o = pyoracle.Oracle(mp, 39, lex, am_threshold=200.0)
t0 = time.perf_counter()
w = o.decode(feats)
dt = time.perf_counter() - t0
print(f"CPU oracle (1 thread, lazy scoring): {dt*1e3:.1f} ms = {T/dt:,.0f} frames/s; words equal: {np.array_equal(w, words)}")
