"""Synthetic workloads for the GMM-scoring + Viterbi path.

Everything here is plain numpy and deterministic in its seeds: MIXSET-v2 model files in the
reference's on-disk format (sietill/Mixtures.cpp:748-878), linear whole-word lexica
(sietill/Lexicon.cpp:11-22), feature batches in the Corpus layout (one contiguous float32
buffer + offsets, sietill/Corpus.cpp:89-111).
"""
from __future__ import annotations

import json
import struct
from dataclasses import dataclass, field

import numpy as np

MAGIC = b"MIXSET\0\0"  # sietill/Mixtures.cpp:153


@dataclass
class MixsetSpec:
    """Accumulator-level content of a MIXSET v2 file (what MixtureModel::read consumes)."""

    dim: int
    mean_acc: np.ndarray  # [n_mean, D] f64  accumulated sums
    mean_w: np.ndarray  # [n_mean]    f64  counts
    var_acc: np.ndarray  # [n_var, D]  f64  accumulated sums of squares
    var_w: np.ndarray  # [n_var]     f64
    dens_mean: np.ndarray  # [n_dens] u32
    dens_var: np.ndarray  # [n_dens] u32
    mixtures: list = field(default_factory=list)  # per state: list of density indices


def make_mixset(n_states, n_mix, dim=39, seed=1, tie_vars=False, var_floor=0.5) -> MixsetSpec:
    """Random model: mu ~ N(0,1), sigma^2 = var_floor + |N(0,1)|, count = 10 + (i mod 7).

    n_mix: int (same for every state) or a sequence of per-state density counts.
    tie_vars: all densities of one mixture share the variance of its first density
              (exercises var_idx != mean_idx).
    """
    rng = np.random.default_rng(seed)
    per_state = [int(n_mix)] * n_states if np.isscalar(n_mix) else [int(x) for x in n_mix]
    assert len(per_state) == n_states
    C = int(sum(per_state))
    mu = rng.standard_normal((C, dim))
    var = var_floor + np.abs(rng.standard_normal((C, dim)))
    cnt = 10.0 + (np.arange(C) % 7)
    dens_mean = np.arange(C, dtype=np.uint32)
    dens_var = np.arange(C, dtype=np.uint32)
    mixtures, k = [], 0
    for m in per_state:
        mixtures.append(list(range(k, k + m)))
        if tie_vars and m > 0:
            dens_var[k : k + m] = k
        k += m
    mean_acc = mu * cnt[:, None]
    if tie_vars:
        # the tied variance accumulator must stay positive against the LAST density's mean
        # (calculate_variance is re-run per referencing density; the last one wins, Mixtures.cpp:396-398)
        var_acc = np.empty_like(mu)
        for dl in mixtures:
            if not dl:
                continue
            last = dl[-1]
            var_acc[dl[0]] = (var[dl[0]] + mu[last] ** 2) * cnt[dl[0]]
            for j in dl[1:]:
                var_acc[j] = (var[j] + mu[j] ** 2) * cnt[j]
    else:
        var_acc = (var + mu**2) * cnt[:, None]
    return MixsetSpec(dim, mean_acc, cnt.copy(), var_acc, cnt.copy(), dens_mean, dens_var, mixtures)


def write_mixset(path, spec: MixsetSpec):
    """Serialise in MIXSET v2 layout (sietill/Mixtures.cpp:834-878 as read back by :748-827)."""
    D = spec.dim
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<II", 2, D))
        for acc, w in ((spec.mean_acc, spec.mean_w), (spec.var_acc, spec.var_w)):
            n = acc.shape[0]
            f.write(struct.pack("<I", n))
            rec = np.zeros(n, dtype=np.dtype([("dim", "<u4"), ("acc", "<f8", (D,)), ("w", "<f8")], align=False))
            rec["dim"] = D
            rec["acc"] = acc
            rec["w"] = w
            f.write(rec.tobytes())
        n_dens = len(spec.dens_mean)
        f.write(struct.pack("<I", n_dens))
        f.write(np.stack([spec.dens_mean, spec.dens_var], axis=1).astype("<u4").tobytes())
        f.write(struct.pack("<I", len(spec.mixtures)))
        for dens in spec.mixtures:
            f.write(struct.pack("<I", len(dens)))
            rec = np.zeros(len(dens), dtype=np.dtype([("idx", "<u4"), ("w", "<f8")], align=False))
            rec["idx"] = dens
            rec["w"] = spec.mean_w[spec.dens_mean[dens]] if len(dens) else []
            f.write(rec.tobytes())


@dataclass
class LexiconSpec:
    """Linear whole-word lexicon as Lexicon::add_word builds it (fresh consecutive state ids)."""

    word_states: np.ndarray  # [W] u16 distinct states per word
    word_reps: np.ndarray  # [W] u16 state repetitions
    silence_idx: int

    @property
    def n_words(self):
        return len(self.word_states)

    @property
    def n_states(self):
        return int(self.word_states.astype(np.int64).sum())

    def flatten(self):
        """-> (word_off[W+1] u32, automaton u16[], silence_state) with repetitions expanded."""
        off, aut, s = [0], [], 0
        for n, r in zip(self.word_states, self.word_reps):
            for k in range(int(n)):
                aut.extend([s + k] * int(r))
            s += int(n)
            off.append(len(aut))
        word_off = np.asarray(off, dtype=np.uint32)
        automaton = np.asarray(aut, dtype=np.uint16)
        return word_off, automaton, int(automaton[word_off[self.silence_idx]])


def make_lexicon(n_words, states_per_word=3, reps=1, extra_states_last=0) -> LexiconSpec:
    """silence (1 state x1) + n_words words (S = 1 + 3W)."""
    ws = np.full(n_words + 1, states_per_word, dtype=np.uint16)
    wr = np.full(n_words + 1, reps, dtype=np.uint16)
    ws[0], wr[0] = 1, 1
    if extra_states_last:
        ws[-1] += extra_states_last
    return LexiconSpec(ws, wr, 0)


@dataclass
class ExplicitLexicon:
    """A lexicon given as the arrays sr_lexicon_create takes: any word may be silence, words may share states."""

    word_off: np.ndarray  # [W + 1] u32
    automaton: np.ndarray  # u16 emission state per position
    silence_idx: int

    @property
    def n_words(self):
        return len(self.word_off) - 1

    @property
    def n_states(self):
        return int(self.automaton.max()) + 1

    def flatten(self):
        return self.word_off, self.automaton, int(self.automaton[self.word_off[self.silence_idx]])


def make_ragged_lexicon(n_words, rng, short=False) -> ExplicitLexicon:
    """What Lexicon::add_word permits and make_lexicon never draws: silence anywhere in the word list, words of 1 .. 40 states
    with or without repetitions side by side (several one-position words among them), now and then a word that is a clone of
    an earlier one (the same states: every hypothesis of the two ties) or that starts with the silence state.
    short: every word has at most four positions (the lexica the word-per-lane search kernel takes)."""
    sil_idx = int(rng.integers(0, n_words + 1))
    off, aut, s = [0], [], 0
    words = []
    for w in range(n_words + 1):
        if w == sil_idx:
            st = [s]; s += 1
            sil_state = st[0]
            reps = 1
        else:
            r = rng.random()
            if r < 0.12 and any(i != sil_idx for i in range(len(words))):
                src = int(rng.choice([i for i in range(len(words)) if i != sil_idx]))
                st, reps = words[src]
            else:
                n = int(rng.choice([1, 2, 3, 4] if short else [1, 1, 2, 3, 3, 4, 7, 40]))
                st = list(range(s, s + n)); s += n
                reps = int(rng.integers(1, 3)) if not short or n <= 2 else 1  # short: at most four positions per word
        words.append((st, reps))
    for w, (st, reps) in enumerate(words):
        if w != sil_idx and w > sil_idx and rng.random() < 0.05 and (not short or len(st) * reps < 4):
            st = [sil_state] + list(st)  # a word that begins in silence (the tdps are keyed on the state, TdpModel.cpp:19-29)
        for x in st:
            aut.extend([x] * reps)
        off.append(len(aut))
    if max(off[i + 1] - off[i] for i in range(n_words + 1)) < 2:  # sr_lexicon_create: some word with two or more positions
        aut.append(s); off[-1] += 1
        if sil_idx == n_words:  # (never lengthen silence: give the word before it the position instead)
            aut = aut[:-1]; aut.insert(off[-2], s); off[-2] += 1
    return ExplicitLexicon(np.asarray(off, dtype=np.uint32), np.asarray(aut, dtype=np.uint16), sil_idx)


def sietill_lexicon() -> LexiconSpec:
    """The reference's 12-word / 106-state digit lexicon (sietill/Lexicon.cpp:70-85)."""
    ws = np.array([1, 9, 9, 9, 9, 12, 9, 12, 9, 9, 9, 9], dtype=np.uint16)
    wr = np.array([1] + [2] * 11, dtype=np.uint16)
    return LexiconSpec(ws, wr, 0)


def make_features(n_frames, dim=39, seed=2) -> np.ndarray:
    """i.i.d. N(0,1) float32 frames (mean/var-normalised MFCC stand-in)."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n_frames, dim)).astype(np.float32)


def make_batch(n_utts, t_min=200, t_max=400, dim=39, seed=7):
    """Corpus layout: feats [sum T, D] float32 + frame_off [n_utts+1] uint64 (lengths ~ U{t_min..t_max})."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(t_min, t_max + 1, size=n_utts)
    frame_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    feats = rng.standard_normal((int(frame_off[-1]), dim)).astype(np.float32)
    return feats, frame_off


def sample_utterance(spec: MixsetSpec, lex: LexiconSpec, words, seed=3, frames_per_state=(2, 5), noise=1.0):
    """Draw frames from the model along `sil w1 sil w2 ... sil` so that the decoder has something to
    recognise (beam pruning then behaves like on real speech). Returns float32 [T, D]."""
    rng = np.random.default_rng(seed)
    word_off, automaton, _ = lex.flatten()
    mu = spec.mean_acc / spec.mean_w[:, None]
    var = spec.var_acc[spec.dens_var] / spec.var_w[spec.dens_var][:, None] - mu[spec.dens_mean] ** 2
    seq = [lex.silence_idx]
    for w in words:
        seq += [int(w), lex.silence_idx]
    frames = []
    for w in seq:
        for p in range(word_off[w], word_off[w + 1]):
            s = int(automaton[p])
            dens = spec.mixtures[s]
            n = int(rng.integers(frames_per_state[0], frames_per_state[1] + 1))
            for _ in range(n):
                d = dens[int(rng.integers(len(dens)))]
                frames.append(mu[spec.dens_mean[d]] + noise * np.sqrt(np.abs(var[d])) * rng.standard_normal(spec.dim))
    return np.asarray(frames, dtype=np.float32)


def scale_variances(spec: MixsetSpec, factor: float):
    """Multiplies every variance by `factor` in place (the accumulators keep their means): tight variances (0.004) give negative
    emission costs, the case the reference's pre-AM early-out is live in (Recognizer.cpp:143,173)."""
    mu_v = spec.mean_acc[spec.dens_mean] / spec.mean_w[spec.dens_mean][:, None]      # per density: its mean
    var = spec.var_acc[spec.dens_var] / spec.var_w[spec.dens_var][:, None] - mu_v ** 2
    spec.var_acc[spec.dens_var] = (factor * var + mu_v ** 2) * spec.var_w[spec.dens_var][:, None]


def write_config(path, mixset_path, tdp=(3.0, 0.0, 30.0), am_threshold=200.0, word_penalty=10.0, extra=None):
    """JSON config for the reference classes (double parameters need a decimal point,
    sietill/Config.cpp:114-126; verbosity must be set, Mixtures.cpp:150,164)."""
    cfg = {
        "action": "recognize",
        "verbosity": "noLog",
        "load-mixtures-from": str(mixset_path),
        "tdp-loop": float(tdp[0]),
        "tdp-forward": float(tdp[1]),
        "tdp-skip": float(tdp[2]),
        "am-threshold": float(am_threshold),
        "word-penalty": float(word_penalty),
    }
    if extra:
        cfg.update(extra)
    with open(path, "w") as f:
        json.dump(cfg, f)
    return cfg


def read_mixset(path) -> MixsetSpec:
    """Parse a MIXSET v2 file back into accumulators + topology (inverse of write_mixset)."""
    buf = open(path, "rb").read()
    assert buf[:8] == MAGIC
    version, D = struct.unpack_from("<II", buf, 8)
    assert version == 2
    pos = 16
    blocks = []
    for _ in range(2):
        (n,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        rec = np.frombuffer(buf, dtype=np.dtype([("dim", "<u4"), ("acc", "<f8", (D,)), ("w", "<f8")]), count=n, offset=pos)
        pos += rec.nbytes
        blocks.append((rec["acc"].copy().reshape(n, D), rec["w"].copy()))
    (n_dens,) = struct.unpack_from("<I", buf, pos)
    pos += 4
    dens = np.frombuffer(buf, dtype="<u4", count=2 * n_dens, offset=pos).reshape(n_dens, 2)
    pos += 8 * n_dens
    (n_mix,) = struct.unpack_from("<I", buf, pos)
    pos += 4
    mixtures = []
    for _ in range(n_mix):
        (nd,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        rec = np.frombuffer(buf, dtype=np.dtype([("idx", "<u4"), ("w", "<f8")]), count=nd, offset=pos)
        pos += rec.nbytes
        mixtures.append([int(x) for x in rec["idx"]])
    return MixsetSpec(D, blocks[0][0], blocks[0][1], blocks[1][0], blocks[1][1], dens[:, 0].copy(), dens[:, 1].copy(), mixtures)
