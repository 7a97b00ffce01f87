"""Cliff ledger (round 5, VERDICT r4 "Next" #2c): what the scoring path costs AWAY from the benchmark's shape.

One line per model shape at the headline's size (S = 4000 states, 302 685 frames in 1000 utterances): kernel times from the library's own
HIP events (sr_profile_*), per launch, median of `--reps` passes, and the cost per density-dimension relative to the 39-dimensional /
32-density headline model measured in the same process.  Lines:
  dim 13 / 26 / 33 / 39 / 40 / 45 / 50 / 62 at 32 densities per mixture   (the refinement's padded dimensions; K = 128 from dim 47)
  160 densities per mixture at dim 39                                     (> 128: beyond the fp16 pass' four chunks)
  the zerogram search on the headline lexicon with negative emission costs (variances x 0.004: entry-slot costs below zero)
usage (GPU box): python tools/cliffs.py [--frames-scale 1.0] [--reps 3] [--only dims|m160|neg] > gpurun_out/r5_cliffs.txt"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from speechrecognition_amd import capi, synth  # noqa: E402


def time_scoring(mp, D, feats, off, reps, kernel=capi.GMM_PREFILTER):
    """(P ms, R ms, gmm ms, evaluations per pair) per pass over the corpus, median of reps."""
    with capi.Model.from_mixset(mp, D) as m:
        corpus = m.upload(feats, off)
        lex = synth.make_lexicon(1333, 3, 1)
        word_off, automaton, sil_state = lex.flatten()
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil_state)
        corpus.recognize(lexh, 200.0, 10.0, kernel)  # warm-up: packings, workspaces
        rows = []
        for _ in range(reps):
            m.profile(True)
            corpus.recognize(lexh, 200.0, 10.0, kernel)
            p = m.profile_read()
            m.profile(False)
            rows.append((p["prefilter_ms"], p["refine_ms"], p["gmm_ms"], p["search_ms"],
                         p["refined_densities"] / max(1, p["refined_pairs"])))
        lexh.close()
        corpus.close()
    return tuple(float(np.median([r[i] for r in rows])) for i in range(5))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--utts", type=int, default=1000)
    ap.add_argument("--only", default="")
    ap.add_argument("--dims", default="13,26,33,40,45,50,62")
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="cliffs_")
    S = 4000
    print(f"# cliff ledger: S = {S}, {args.utts} utterances U{{200..400}} frames, median of {args.reps} passes; kernel times from sr_profile_* (HIP events)")
    print(f"# {'model':34s} {'P ms':>8s} {'R ms':>8s} {'GMM ms':>8s} {'eval/pair':>9s} {'ps per dens*dim':>16s} {'vs headline':>11s}")
    base = None

    def line(tag, D, M, kernel=capi.GMM_PREFILTER, mix_seed=23, tie_vars=False):
        nonlocal base
        spec = synth.make_mixset(S, M, D, seed=mix_seed, tie_vars=tie_vars)
        mp = os.path.join(tmp, f"m_{D}_{M}.mix")
        synth.write_mixset(mp, spec)
        feats, off = synth.make_batch(args.utts, 200, 400, D, seed=7)
        n = int(off[-1])
        p, r, g, s_, ev = time_scoring(mp, D, feats, off, args.reps, kernel)
        per = g * 1e-3 / (n * float(S) * M * D) * 1e12  # ps per (frame, density, dimension)
        if base is None:
            base = per
        print(f"  {tag:34s} {p:8.2f} {r:8.2f} {g:8.2f} {ev:9.3f} {per:16.4f} {per / base:11.2f}", flush=True)
        os.remove(mp)

    if args.only in ("", "dims", "m160"):
        line("dim 39 x 32 (headline)", 39, 32)
    if args.only in ("", "dims"):
        for D in [int(x) for x in args.dims.split(",") if x]:
            line(f"dim {D} x 32", D, 32)
    if args.only in ("", "pooled"):
        # one variance vector per state (mixture pooling / tied variances): the per-density route at the untied model's speed
        if args.only == "pooled":
            line("dim 39 x 32 (headline)", 39, 32)
        line("dim 39 x 32, variances tied per state", 39, 32, tie_vars=True)
    if args.only in ("", "m160"):
        line("dim 39 x 64 (configs[4] mixtures)", 39, 64)
        line("dim 39 x 128", 39, 128)
        line("dim 39 x 160 (> 128 densities)", 39, 160)
    if args.only in ("", "neg"):
        # zerogram search with negative emission costs on the headline lexicon: variances x 0.004 as in
        # tests/test_gpu_parity.py::test_negative_emission_costs..., features drawn near the means so that costs do go negative
        D, M = 39, 32
        for scale, tag, wl in ((1.0, "search, costs >= 0 (headline)", (1333, 3)), (0.004, "search, negative emission costs", (1333, 3)),
                               (1.0, "search, 12-position words, >= 0", (333, 12)), (0.004, "search, 12-position words, negative", (333, 12))):
            lex = synth.make_lexicon(wl[0], wl[1], 1, extra_states_last=S - 1 - wl[0] * wl[1])
            spec = synth.make_mixset(S, M, D, seed=23, var_floor=0.5 if scale == 1.0 else 0.002)
            if scale != 1.0:
                synth.scale_variances(spec, scale)
            mp = os.path.join(tmp, f"neg_{scale}.mix")
            synth.write_mixset(mp, spec)
            rng = np.random.default_rng(5)
            utts, lens = [], []
            for u in range(args.utts):
                T = int(rng.integers(200, 401))
                x = synth.sample_utterance(spec, lex, rng.integers(1, lex.n_words, size=12 if wl[1] == 3 else 4), seed=1000 + u, frames_per_state=(3, 7), noise=0.8)
                while len(x) < T:
                    x = np.concatenate([x, x])
                x = x[:T]
                utts.append(x.astype(np.float32)); lens.append(T)
            feats = np.concatenate(utts)
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
            with capi.Model.from_mixset(mp, D) as m:
                word_off, automaton, sil_state = lex.flatten()
                lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil_state)
                corpus = m.upload(feats, off)
                corpus.recognize(lexh, 200.0, 10.0, capi.GMM_PREFILTER)
                ts = []
                for _ in range(args.reps):
                    m.profile(True)
                    t0 = time.perf_counter()
                    words, woff = corpus.recognize(lexh, 200.0, 10.0, capi.GMM_PREFILTER)
                    wall = (time.perf_counter() - t0) * 1e3
                    p = m.profile_read()
                    m.profile(False)
                    ts.append((p["search_ms"], p["gmm_ms"], wall))
                sc = m.score_frames(feats[:3000], capi.GMM_PREFILTER)  # (a sample: the fraction of negative emission costs)
                neg = float((sc < 0).mean())
                print(f"  {tag:34s} search {np.median([t[0] for t in ts]):8.2f} ms  gmm {np.median([t[1] for t in ts]):8.2f} ms  "
                      f"step wall {np.median([t[2] for t in ts]):8.2f} ms  words {int(woff[-1])}  frames {int(off[-1])}  negative scores {neg:.2e}", flush=True)
                corpus.close(); lexh.close()


if __name__ == "__main__":
    main()
