"""Helpers shared by the tests: fixture loading and model-file materialisation."""
from __future__ import annotations

import glob
import os

import numpy as np

from speechrecognition_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
                  if not p.endswith(("edit_distance.npz", "accumulate.npz")))


class Case:
    """One golden fixture: inputs + outputs of the real reference (oracle/gen_golden.py)."""

    def __init__(self, name, tmpdir):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.z = z
        self.name = name
        self.lex = synth.LexiconSpec(z["lex_word_states"], z["lex_word_reps"], int(z["lex_silence"]))
        if "model_dim" in z:
            mixtures = [list(z["model_mix_dens"][z["model_mix_off"][s]:z["model_mix_off"][s + 1]])
                        for s in range(len(z["model_mix_off"]) - 1)]
            self.spec = synth.MixsetSpec(int(z["model_dim"]), z["model_mean_acc"], z["model_mean_w"], z["model_var_acc"],
                                         z["model_var_w"], z["model_dens_mean"], z["model_dens_var"], mixtures)
        else:
            S, M, D, seed = [int(x) for x in z["model_seed"]]
            self.spec = synth.make_mixset(S, M, D, seed=seed)
        self.dim = self.spec.dim
        self.feats = z["feats"]
        self.tdp = tuple(float(x) for x in z["tdp"])
        self.beam = float(z["beam"])
        self.wp = float(z["word_penalty"])
        self.pooling = int(z["pooling"])
        self.max_approx = bool(int(z["max_approx"]))
        self.mixset_path = os.path.join(str(tmpdir), name + ".mix")
        synth.write_mixset(self.mixset_path, self.spec)

    def oracle(self, po):
        return po.Oracle(self.mixset_path, self.dim, self.lex, tdp=self.tdp, am_threshold=self.beam,
                         word_penalty=self.wp, pooling=self.pooling, max_approx=self.max_approx)

    def check_scores(self, scores, exact=True, rtol=0.0):
        z = self.z
        if "scores" in z:
            want, got = z["scores"], scores
        else:
            want, got = z["score_val"], scores.reshape(-1)[z["score_idx"]]
        if exact:
            assert np.array_equal(want.view(np.uint64), np.ascontiguousarray(got).view(np.uint64))
            if "score_xor" in z:
                assert np.bitwise_xor.reduce(np.ascontiguousarray(scores).view(np.uint64).reshape(-1)) == z["score_xor"]
        else:
            np.testing.assert_allclose(got, want, rtol=rtol, atol=0)
