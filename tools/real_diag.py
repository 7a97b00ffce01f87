import sys, os, numpy as np, tempfile
sys.path.insert(0, '/root/repo')
from speechrecognition_amd import capi, synth
z = np.load('/root/repo/tests/golden_real/sietill_real.npz')
lex = synth.sietill_lexicon(); word_off, automaton, sil = lex.flatten()
tdp = tuple(float(x) for x in z["tdp"]); off = z["frame_off"].astype(np.int64)
auts = []
for i in range(len(z["ref_off"]) - 1):
    a = [sil]
    for w in z["ref_flat"][z["ref_off"][i]:z["ref_off"][i + 1]]:
        a += list(automaton[word_off[w]:word_off[w + 1]]) + [sil]
    auts.append(np.asarray(a, dtype=np.uint16))
for pname, pool in (("mixture", 1), ("none", 2)):
    mp = os.path.join(tempfile.mkdtemp(), "m.mix"); open(mp, "wb").write(z[f"model_{pname}"].tobytes())
    with capi.Model.from_mixset(mp, int(z["dim"]), pool, True) as m:
        lexh = m.lexicon(word_off, automaton, lex.silence_idx, tdp, sil)
        corpus = m.upload(z["feats"], z["frame_off"])
        for tag in ("wide", "tight"):
            key = f"{pname}_{tag}"
            for kernel in (1, 0):
                words, woff = corpus.recognize(lexh, float(z[f"{key}_beam"]), float(z[f"{key}_wp"]), kernel)
                st, cost = corpus.align(auts, tdp, sil, kernel)
                st2, cost2 = corpus.align(auts, tdp, sil, kernel, pruning_threshold=float(z[f"{key}_athr"]))
                print(key, "kernel", kernel, "words eq", np.array_equal(words, z[f"{key}_words"]),
                      "full: frames differing", int((st != z[f"{key}_align_full"]).sum()), "max cost rel", float(np.max(np.abs(cost - z[f"{key}_align_full_cost"]) / np.abs(cost))),
                      "pruned: frames differing", int((st2 != z[f"{key}_align_pruned"]).sum()), "max cost rel", float(np.max(np.abs(cost2 - z[f"{key}_align_pruned_cost"]) / np.abs(cost2))))
        es = m.score_frames(z["feats"][:2000], 1); ms = m.score_frames(z["feats"][:2000], 0)
        print(pname, "mfma vs exact max rel", float(np.max(np.abs(es - ms) / np.maximum(1, np.abs(es)))))
