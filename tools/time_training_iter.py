"""One EM iteration of the trainer-side callers on the bench model: pruned forced alignment of
every utterance against `sil w sil w sil w sil`, max-approx accumulation, finalize into a new device model.
Times each ABI call (features resident) and the CPU oracle on a sample.
usage: python tools/time_training_iter.py [utts] [words] [mix]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from speechrecognition_amd import capi, synth
from oracle import pyoracle

U = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1333
M = int(sys.argv[3]) if len(sys.argv) > 3 else 32
lex = synth.make_lexicon(W, 3, 1)
spec = synth.make_mixset(lex.n_states, M, 39, seed=23)
mp = os.path.join(tempfile.mkdtemp(), "m.mix")
synth.write_mixset(mp, spec)
feats, off = synth.make_batch(U, 200, 400, 39, seed=7)
word_off, automaton, sil = lex.flatten()
rng = np.random.default_rng(5)
auts = []
for u in range(U):
    a = [sil]
    for w in rng.integers(1, lex.n_words, size=3):
        a += list(automaton[word_off[w]:word_off[w + 1]]) + [sil]
    auts.append(np.asarray(a, np.uint16))
tdp, thr = (3.0, 0.0, 30.0), 200.0
dens_off = np.concatenate([[0], np.cumsum([len(m_) for m_ in spec.mixtures])]).astype(np.uint32)

def timed(f, n=3):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    return r, (time.perf_counter() - t0) / n

with capi.Model.from_mixset(mp, 39) as m:
    c = m.upload(feats, off)
    (states, cost), t_align = timed(lambda: c.align(auts, tdp, sil, pruning_threshold=thr))
    acc, t_acc = timed(lambda: c.accumulate(states))
    def fin():
        m2 = capi.Model.from_statistics(39, dens_off, spec.dens_mean, spec.dens_var, acc)
        m2.close()
    _, t_fin = timed(fin, 2)
    _, t_acc_dev = timed(lambda: c.accumulate_on_device(states))
    def fin_dev():
        m3 = c.next_model()
        m3.close()
    _, t_fin_dev = timed(fin_dev, 2)
    F = len(feats)
    print(f"GPU ({U} utterances, {F} frames, {lex.n_states} states x {M}): align_pruned {t_align*1e3:.1f} ms "
          f"({F/t_align:,.0f} frames/s), accumulate {t_acc*1e3:.1f} ms ({F/t_acc:,.0f} frames/s), "
          f"finalize from host statistics {t_fin*1e3:.1f} ms; statistics kept on the device: accumulate {t_acc_dev*1e3:.1f} ms + "
          f"finalize {t_fin_dev*1e3:.1f} ms -> EM iteration {1e3*(t_align + t_acc_dev + t_fin_dev):.1f} ms")
    c.close()
# CPU oracle on a sample: lazy scoring inside the aligner (the reference's cost profile), then accumulate
n = min(U, 16)
o = pyoracle.Oracle(mp, 39, lex)
t0 = time.perf_counter()
ok = True
for u in range(n):
    x = feats[int(off[u]):int(off[u + 1])]
    st, cs = o.align_pruned(x, auts[u], thr)
    ok = ok and np.array_equal(st, states[int(off[u]):int(off[u + 1])]) and cs == cost[u]
ta = time.perf_counter() - t0
fs = int(off[n])
t0 = time.perf_counter()
o.accumulate(feats[:fs], states[:fs])
tc = time.perf_counter() - t0
print(f"CPU oracle (1 thread, {n} utterances, {fs} frames): align_pruned {fs/ta:,.0f} frames/s, accumulate {fs/tc:,.0f} frames/s; "
      f"alignments and costs equal: {ok}")
