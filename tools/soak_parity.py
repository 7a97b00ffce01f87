"""Randomised soak of the GPU path against the CPU oracle: random lexica / mixtures / dims / beams / utterance sets,
scores (prefilter, exact) bit-identical, words + tracebacks + alignments identical, bigram search identical.
usage: python tools/soak_parity.py [n_cases] [seed] [ragged|short] [--ledger FILE]
(ragged: synth.make_ragged_lexicon instead of the uniform lexicon; short: words of one to four positions -- the word-per-lane search
kernel, cross-checked against the slot-per-lane kernel.)  --ledger appends ONE JSON line per run -- seed, generator, cases run, the
first failing case if any, wall time, the commit and kernel-source hash of the library under test -- to FILE; the lines judged are
kept under profiles/r4_soak.jsonl, profiles/r5_soak.jsonl."""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from speechrecognition_amd import capi, synth
from oracle import pyoracle



def run_case(case, seed0=0, ragged=False, tmp=None):
    """One randomised case; raises AssertionError on the first mismatch, returns the case's description."""
    tmp = tmp or tempfile.mkdtemp()
    if ragged is True:
        ragged = "ragged"
    rng = np.random.default_rng(seed0 * 100003 + case)
    W = int(rng.choice([2, 5, 17, 60, 300]))
    if ragged == "short" and rng.random() < 0.15:
        # two and three words per lane; 2900 three- and four-position words: more than 8192 type-padded positions, the word-per-lane
        # kernel with decode_big_kernel as its replay (round 4)
        W = int(rng.choice([1100, 2300, 2900]))
    spw = int(rng.integers(1, 5))
    reps = int(rng.integers(1, 3))
    if spw * reps < 2:
        reps = 2  # (the decoder wants a word with two or more positions: sr_lexicon_create's documented limit)
    D = int(rng.choice([1, 4, 12, 25, 26, 33, 39, 40, 46, 47, 50, 62, 63, 64, 90]))  # (round 5: every dimension <= 62 on the prefilter path; 63 .. 160: exact kernel)
    Mhi = int(rng.choice([1, 3, 8, 33, 70, 100]))  # (33 / 70 / 100: two, three and four 32-slot chunks per state in the refinement)
    if W >= 1000:
        Mhi = min(Mhi, 3)
    lex = synth.make_ragged_lexicon(W, rng, short=ragged == "short") if ragged else synth.make_lexicon(W, spw, reps)
    speech = [w for w in range(lex.n_words) if w != lex.silence_idx]
    nm = rng.integers(1, Mhi + 1, size=lex.n_states)
    spec = synth.make_mixset(lex.n_states, nm, D, seed=case, var_floor=float(rng.choice([0.5, 1e-3])))
    if rng.random() < 0.3:
        # round 5: tight variances -> negative emission costs, where the reference's pre-AM early-out is live (Recognizer.cpp:143,173):
        # the word-per-lane kernel's NEG variant on short-word lexica, the flag + replay route elsewhere
        synth.scale_variances(spec, float(rng.choice([0.004, 0.05])))
    mp = os.path.join(tmp, "m.mix")
    synth.write_mixset(mp, spec)
    beam = float(rng.choice([15.0, 60.0, 200.0, 1e9]))
    n_utts = int(rng.choice([1, 3, 9, 140, 300]))
    if W >= 1000:  # (keeps the oracle's dense score matrix of the case within seconds)
        n_utts = min(n_utts, 9)
    lens = rng.integers(1, 40, size=n_utts)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    feats = (float(rng.choice([1.0, 3.0])) * rng.standard_normal((int(off[-1]), D))).astype(np.float32)
    for u in range(0, n_utts, 3):
        x = synth.sample_utterance(spec, lex, rng.choice(speech, size=2), seed=case + u)[: lens[u]]
        feats[int(off[u]):int(off[u]) + len(x)] = x
    word_off, automaton, sil = lex.flatten()
    o = pyoracle.Oracle(mp, D, lex, am_threshold=beam)
    want = o.score_matrix(feats)
    tag = f"case {case}: {(ragged + ' ') if ragged else ''}W={W} spw={spw} reps={reps} sil={lex.silence_idx} D={D} M<={Mhi} beam={beam} utts={n_utts}"
    with capi.Model.from_mixset(mp, D) as m:
        corpus = m.upload(feats, off)
        for k in (capi.GMM_PREFILTER, capi.GMM_EXACT):
            got = corpus.score(k)
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (tag, "scores", k)
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, (3.0, 0.0, 30.0), sil)
        words, woff, (tbs, tbw, tbb) = corpus.recognize(lexh, beam, 10.0, capi.GMM_PREFILTER, traceback=True)
        if max(np.diff(word_off.astype(np.int64))) <= 4:  # the word-per-lane kernel ran: the slot-per-lane kernel must agree
            w_s, o_s, (s_s, tw_s, tb_s) = corpus.recognize(lexh, beam, 10.0, capi.GMM_PREFILTER, traceback=True, slot_kernel=True)
            assert np.array_equal(w_s, words) and np.array_equal(o_s, woff) and np.array_equal(tw_s, tbw) and np.array_equal(tb_s, tbb), (tag, "slot kernel")
            assert np.array_equal(s_s.view(np.uint64), tbs.view(np.uint64)), (tag, "slot kernel scores")
        auts, ok_al = [], True
        for u in range(n_utts):
            x = feats[int(off[u]):int(off[u + 1])]
            w, (os_, ow, ob) = o.decode(x, traceback=True)
            assert np.array_equal(w, words[int(woff[u]):int(woff[u + 1])]), (tag, "words", u)
            a = int(off[u]) + u
            assert np.array_equal(tbw[a:a + len(x) + 1], ow) and np.array_equal(tbb[a:a + len(x) + 1], ob), (tag, "tb", u)
            assert np.array_equal(tbs[a:a + len(x) + 1].view(np.uint64), os_.view(np.uint64)), (tag, "tb score", u)
            ws = rng.choice(speech, size=2) if speech else []
            aut = [sil]
            for w_ in ws:
                aut += list(automaton[word_off[w_]:word_off[w_ + 1]]) + [sil]
            if len(aut) > len(x):
                aut = aut[: len(x)]
            auts.append(np.asarray(aut, np.uint16))
        st, cost = corpus.align(auts, (3.0, 0.0, 30.0), sil)
        stp, costp = corpus.align(auts, (3.0, 0.0, 30.0), sil, pruning_threshold=50.0)
        for u in range(n_utts):
            x = feats[int(off[u]):int(off[u + 1])]
            s_, c_ = o.align_full(x, auts[u])
            assert np.array_equal(st[int(off[u]):int(off[u + 1])], s_) and cost[u] == c_, (tag, "align", u)
            s_, c_ = o.align_pruned(x, auts[u], 50.0)
            assert np.array_equal(stp[int(off[u]):int(off[u + 1])], s_) and costp[u] == c_, (tag, "align pruned", u)
        # bigram search on the same model (the automaton positions are the lexicon's mixtures)
        nW = lex.n_words
        if nW <= 400 or n_utts <= 9:  # (big lexica -- two and three words per lane in the register layout -- on the small batches only: the oracle's cost)
            lm = (-np.log(rng.dirichlet(np.ones(nW), size=nW))).T.astype(np.float32).copy()
            tdp = np.array([[3.0, 0.0, 30.0, float(rng.choice([0.0, 5.0]))], [1.0, 0.0, 40.0, 2.0]], np.float32)
            bg_off, bg_aut = word_off, automaton
            if rng.random() < 0.3 and nW <= 400:  # (a big lexicon runs in the register layout only, which wants a one-state silence word)
                # round 5: a silence word of two to four states with forward / skip penalties of its own -- the dense
                # state layout, where tdp[isSilence][1..2] apply inside silence and its copies (LinearSearch.cc:296-326)
                tdp[1] = [float(rng.choice([0.0, 1.0])), float(rng.choice([0.0, 7.0])), float(rng.choice([3.0, 40.0])), 2.0]
                si, extra = lex.silence_idx, rng.integers(0, lex.n_states, size=int(rng.integers(1, 4))).astype(np.uint16)
                cut = int(word_off[si + 1])
                bg_aut = np.concatenate([automaton[:cut], extra, automaton[cut:]]).astype(np.uint16)
                bg_off = word_off.copy()
                bg_off[si + 1:] += len(extra)
            acp = float(rng.choice([30.0, 120.0, pyoracle.FLT_MAX]))
            lmp = float(rng.choice([5.0, 25.0, pyoracle.FLT_MAX]))
            bg = m.bigram(bg_off, bg_aut, lex.silence_idx, lm, tdp)
            gw, gs, gt, goff = corpus.recognize_bigram(bg, acp, lmp)
            try:
                dw, ds, dt, doff = corpus.recognize_bigram(bg, acp, lmp, dense_states=True)  # register layout (short words) vs dense LDS image
            except capi.SrError as e:
                if e.code != -4:  # SR_ELIMIT: a lexicon whose dense image does not fit the LDS runs in the register layout only
                    raise
            else:
                assert np.array_equal(gw, dw) and np.array_equal(gt, dt) and np.array_equal(goff, doff) and np.array_equal(gs.view(np.uint32), ds.view(np.uint32)), (tag, "bigram layouts")
            for u in range(n_utts):
                x = feats[int(off[u]):int(off[u + 1])]
                w, s_, t_ = pyoracle.bigram_decode(want[int(off[u]):int(off[u + 1])], bg_off, bg_aut, lex.silence_idx, lm, tdp, acp, lmp)
                a, b = int(goff[u]), int(goff[u + 1])
                assert np.array_equal(gw[a:b], w) and np.array_equal(gt[a:b], t_), (tag, "bigram", u)
                assert np.array_equal(gs[a:b].view(np.uint32), s_.view(np.uint32)), (tag, "bigram score", u)
            bg.close()
        lexh.close()
        corpus.close()
    o.close()
    # sum scoring (max-approx false, Mixtures.cpp:719-728) through SR_GMM_DEFAULT = the FP64-MFMA kernel (round 4) and through the
    # direct-form kernel: 1e-9 / 1e-12 of the oracle (device exp / log); on the small cases only (the oracle's dense sum-mode matrix)
    if lex.n_states * int(off[-1]) <= 400000:
        o = pyoracle.Oracle(mp, D, lex, am_threshold=beam, max_approx=False)
        want_sum = o.score_matrix(feats)
        o.close()
        with capi.Model.from_mixset(mp, D, max_approx=False) as m:
            for k, tol in ((capi.GMM_DEFAULT, 1e-9), (capi.GMM_EXACT, 1e-12)):
                got = m.score_frames(feats, k)
                fin = np.isfinite(want_sum)
                assert np.array_equal(np.isfinite(got), fin), (tag, "sum mode finiteness", k)
                err = np.abs(got[fin] - want_sum[fin]) / np.maximum(np.abs(want_sum[fin]), 1.0)
                assert err.max(initial=0.0) <= tol, (tag, "sum mode", k, float(err.max(initial=0.0)))
    return tag


def ledger_line(seed0, generator, n_cases, done, failure, secs):
    import hashlib
    from speechrecognition_amd import build as B
    csrc = os.path.join(os.path.dirname(os.path.abspath(B.__file__)), "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode()); h.update(open(os.path.join(csrc, name), "rb").read())
    info = B.build_info()
    return {"tool": "tools/soak_parity.py", "seed": seed0, "generator": generator or "uniform", "cases_asked": n_cases, "cases_passed": done,
            "first_failure": failure, "wall_s": round(secs, 1), "git_head": info.get("git_head", "unknown") + ("+dirty" if info.get("dirty") else ""),
            "kernel_sources_sha16": h.hexdigest()[:16], "library": os.environ.get("SRGPU_LIB", "in-tree libsrgpu.so"),
            "checks": "scores prefilter+exact == oracle (uint64); words, traceback, full+pruned alignments == oracle; word-per-lane == slot kernel; "
                      "bigram register layout == dense layout == oracle restatement; sum mode (SR_GMM_DEFAULT 1e-9, exact kernel 1e-12) on the small cases"}


if __name__ == "__main__":
    argv = list(sys.argv[1:])
    ledger = None
    if "--ledger" in argv:
        i = argv.index("--ledger"); ledger = argv[i + 1]; del argv[i:i + 2]
    n_cases = int(argv[0]) if len(argv) > 0 else 50
    seed0 = int(argv[1]) if len(argv) > 1 else 0
    ragged = argv[2] if len(argv) > 2 and argv[2] in ("ragged", "short") else False
    tmp = tempfile.mkdtemp()
    t_start = time.time()
    done, failure = 0, None
    import signal

    def _stop(signum, frame):  # a run cut by `timeout` still writes its ledger line
        raise KeyboardInterrupt(f"signal {signum}")
    signal.signal(signal.SIGTERM, _stop)
    try:
        for case in range(n_cases):
            tag = run_case(case, seed0, ragged, tmp)
            done += 1
            print(f"ok {tag}  [{time.time() - t_start:.0f} s]", flush=True)
    except KeyboardInterrupt as e:  # cut short from outside (time limit): not a failure, the line says how far it got
        print(f"stopped after {done} cases ({e})", flush=True)
    except Exception as e:  # a mismatch (AssertionError) or an error status of the library
        failure = {"case": done, "what": f"{type(e).__name__}: {e}"[:400]}
        raise
    finally:
        if ledger:
            with open(ledger, "a") as f:
                f.write(json.dumps(ledger_line(seed0, ragged, n_cases, done, failure, time.time() - t_start)) + "\n")
    print("soak passed:", done, "cases" + ("" if done == n_cases else f" (of {n_cases} asked: stopped by the time limit)"))
