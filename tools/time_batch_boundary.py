"""PCIe-inclusive rate at the boundary that hands over host buffers, against the resident-features rate bench.py reports:
(a) sr_corpus_upload (blocking copy of pageable memory) + recognise + destroy per step, (b) sr_recognize_batch, which feeds
through sr_corpus_upload_async (pinned staging on a copy stream, scoring starts on the first chunk while the rest is in flight).
usage: python tools/time_batch_boundary.py"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from speechrecognition_amd import capi, synth

lex = synth.make_lexicon(1333, 3, 1)
spec = synth.make_mixset(lex.n_states, 32, 39, seed=23)
mp = os.path.join(tempfile.mkdtemp(), "m.mix")
synth.write_mixset(mp, spec)
feats, off = synth.make_batch(1000, 200, 400, 39, seed=7)
word_off, automaton, sil = lex.flatten()
with capi.Model.from_mixset(mp, 39) as m:
    lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil)
    def host_step():
        c = m.upload(feats, off)
        r = c.recognize(lexh, 200.0, 10.0)
        c.close()
        return r
    host_step()
    t0 = time.perf_counter()
    for _ in range(6):
        host_step()
    t_host = (time.perf_counter() - t0) / 6
    m.recognize_batch(lexh, feats, off, 200.0, 10.0)
    t0 = time.perf_counter()
    for _ in range(6):
        wb, ob = m.recognize_batch(lexh, feats, off, 200.0, 10.0)
    t_batch = (time.perf_counter() - t0) / 6
    c = m.upload(feats, off)
    c.recognize(lexh, 200.0, 10.0)
    t0 = time.perf_counter()
    for _ in range(6):
        wr, orr = c.recognize(lexh, 200.0, 10.0)
    t_res = (time.perf_counter() - t0) / 6
    assert np.array_equal(wb, wr) and np.array_equal(ob, orr)
    c.close(); lexh.close()
print(f"resident features: {t_res*1e3:.1f} ms/step = {len(feats)/t_res:,.0f} frames/s; "
      f"host buffers, blocking upload (47 MB pageable H2D + alloc/free per step): {t_host*1e3:.1f} ms/step = {len(feats)/t_host:,.0f} frames/s; "
      f"host buffers, sr_recognize_batch (asynchronous feeder, 2 MiB pieces, scoring starts on the first 1/24): {t_batch*1e3:.1f} ms/step = {len(feats)/t_batch:,.0f} frames/s")
