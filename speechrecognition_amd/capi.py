"""ctypes binding of libsrgpu.so (include/srgpu.h) -- the product's C ABI.

Used by tests/ and bench.py to drive the HIP path exactly as a foreign-language host would; no
compute happens in Python and there is no fallback: if the library or a gfx950 device is missing
the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# SRGPU_LIB: a differently tuned build of the same library (tools/build_variant.py), for A/B timing only
LIB_PATH = os.environ.get("SRGPU_LIB") or os.path.join(HERE, "libsrgpu.so")

GMM_MFMA, GMM_EXACT, GMM_PREFILTER, GMM_DEFAULT = 0, 1, 2, 3
POOL_GLOBAL, POOL_MIXTURE, POOL_NONE = 0, 1, 2
SEARCH_GENERAL_KERNEL = 1
SEARCH_SLOT_KERNEL = 2
BIGRAM_DENSE_STATES = 1
SR_ECORRUPT = -7
SR_ABI_VERSION = 4

# every symbol include/srgpu.h declares
SYMBOLS = [
    "sr_abi_version", "sr_model_trim",
    "sr_last_error", "sr_device_count", "sr_model_create", "sr_model_load_mixset", "sr_model_destroy", "sr_model_info",
    "sr_corpus_upload", "sr_corpus_upload_async", "sr_corpus_wait", "sr_corpus_destroy", "sr_shard_utterances", "sr_recognize_batch_multi", "sr_score_corpus", "sr_score_frames", "sr_lexicon_create",
    "sr_lexicon_destroy", "sr_lexicon_describe", "sr_recognize_corpus", "sr_traceback_corpus", "sr_traceback_words", "sr_recognize_batch", "sr_align_corpus", "sr_align_corpus_pruned", "sr_path_scores_corpus", "sr_model_create_from_statistics", "sr_model_create_from_accumulated", "sr_mixset_write", "sr_model_set_tying", "sr_model_tying_info", "sr_model_topology", "sr_accumulate_corpus",
    "sr_bigram_create", "sr_bigram_destroy", "sr_recognize_bigram_corpus",
    "sr_probe_fp16_denormals", "sr_probe_fp16_accumulation",
    "sr_profile_enable", "sr_profile_reset", "sr_profile_read",
]


class SrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsrgpu error {code}: {msg}")
        self.code = code


class SearchParams(C.Structure):
    _fields_ = [("am_threshold", C.c_double), ("word_penalty", C.c_double), ("gmm_kernel", C.c_int), ("flags", C.c_int)]


class BigramParams(C.Structure):
    _fields_ = [("acoustic_pruning", C.c_float), ("lm_pruning", C.c_float), ("gmm_kernel", C.c_int), ("max_word_ends", C.c_uint32), ("flags", C.c_int)]


class Profile(C.Structure):
    _fields_ = [("gmm_ms", C.c_double), ("gmm_launches", C.c_uint64), ("gmm_flops", C.c_double),
                ("search_ms", C.c_double), ("search_launches", C.c_uint64), ("search_bytes", C.c_double),
                ("frames", C.c_uint64), ("refined_pairs", C.c_uint64), ("refined_densities", C.c_uint64),
                ("prefilter_ms", C.c_double), ("refine_ms", C.c_double)]


_lib = None


def lib():
    """Loads libsrgpu.so (built in-tree by speechrecognition_amd/build.py). Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        if L.sr_abi_version() != SR_ABI_VERSION:  # the struct layouts below are those of include/srgpu.h version SR_ABI_VERSION
            raise RuntimeError(f"{LIB_PATH} has ABI version {L.sr_abi_version()}, this binding was written for {SR_ABI_VERSION}")
        L.sr_last_error.restype = C.c_char_p
        vp, u32, u64, i32, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_double
        L.sr_device_count.argtypes = [C.POINTER(i32)]
        L.sr_model_create.argtypes = [i32, u32, u32, vp, vp, vp, vp, vp, i32, C.POINTER(vp)]
        L.sr_model_load_mixset.argtypes = [C.c_char_p, u32, i32, i32, i32, C.POINTER(vp)]
        L.sr_model_destroy.argtypes = [vp]
        L.sr_model_trim.argtypes = [vp]
        L.sr_model_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u64)]
        L.sr_corpus_upload.argtypes = [vp, vp, vp, u32, C.POINTER(vp)]
        L.sr_corpus_destroy.argtypes = [vp]
        L.sr_corpus_upload_async.argtypes = [vp, vp, vp, u32, C.POINTER(vp)]
        L.sr_corpus_wait.argtypes = [vp]
        L.sr_shard_utterances.argtypes = [vp, u32, u32, vp, vp]
        L.sr_recognize_batch_multi.argtypes = [vp, vp, u32, C.POINTER(SearchParams), vp, vp, u32, vp, vp, vp]
        L.sr_score_corpus.argtypes = [vp, vp, i32, vp]
        L.sr_score_frames.argtypes = [vp, vp, u64, i32, vp]
        L.sr_lexicon_create.argtypes = [vp, u32, vp, vp, u32, C.POINTER(dbl * 3), C.c_uint16, C.POINTER(vp)]
        L.sr_lexicon_destroy.argtypes = [vp]
        L.sr_lexicon_describe.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.sr_recognize_corpus.argtypes = [vp, vp, vp, C.POINTER(SearchParams), vp, vp, vp, vp, vp]
        L.sr_recognize_batch.argtypes = [vp, vp, C.POINTER(SearchParams), vp, vp, u32, vp, vp]
        L.sr_traceback_corpus.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.sr_traceback_words.argtypes = [u32, vp, vp, u32, u32, vp, C.POINTER(u32)]
        L.sr_align_corpus.argtypes = [vp, vp, vp, vp, C.POINTER(dbl * 3), C.c_uint16, i32, vp, vp]
        L.sr_align_corpus_pruned.argtypes = [vp, vp, vp, vp, C.POINTER(dbl * 3), C.c_uint16, dbl, i32, vp, vp]
        L.sr_path_scores_corpus.argtypes = [vp, vp, vp, i32, vp]
        L.sr_model_create_from_statistics.argtypes = [i32, u32, u32, vp, u32, u32, vp, vp, vp, vp, vp, vp, i32, i32, C.POINTER(vp)]
        L.sr_model_create_from_accumulated.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
        L.sr_mixset_write.argtypes = [C.c_char_p, u32, u32, vp, u32, u32, vp, vp, vp, vp, vp, vp]
        L.sr_model_set_tying.argtypes = [vp, u32, u32, vp, vp]
        L.sr_model_tying_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
        L.sr_model_topology.argtypes = [vp, vp, vp, vp]
        L.sr_accumulate_corpus.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp]
        L.sr_bigram_create.argtypes = [vp, u32, vp, vp, u32, vp, vp, C.POINTER(vp)]
        L.sr_bigram_destroy.argtypes = [vp]
        L.sr_recognize_bigram_corpus.argtypes = [vp, vp, vp, C.POINTER(BigramParams), vp, vp, vp, vp]
        L.sr_probe_fp16_denormals.argtypes = [i32, C.POINTER(i32)]
        L.sr_probe_fp16_accumulation.argtypes = [i32, C.POINTER(i32), C.POINTER(dbl)]
        L.sr_profile_enable.argtypes = [vp, i32]
        L.sr_profile_reset.argtypes = [vp]
        L.sr_profile_read.argtypes = [vp, C.POINTER(Profile)]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise SrError(rc, lib().sr_last_error().decode(errors="replace"))


def device_count():
    n = C.c_int(0)
    _check(lib().sr_device_count(C.byref(n)))
    return n.value


def _ptr(a):
    return a.ctypes.data if a is not None else None


class Model:
    """sr_model handle: the GPU FeatureScorer (MixtureModel, sietill/Mixtures.hpp:18)."""

    def __init__(self, handle):
        self.h = handle
        d, s, c = C.c_uint32(), C.c_uint32(), C.c_uint64()
        _check(lib().sr_model_info(self.h, C.byref(d), C.byref(s), C.byref(c)))
        self.dim, self.n_states, self.n_densities = d.value, s.value, c.value

    @classmethod
    def from_mixset(cls, path, dim, pooling=POOL_NONE, max_approx=True, device=0):
        h = C.c_void_p()
        _check(lib().sr_model_load_mixset(str(path).encode(), dim, pooling, int(max_approx), device, C.byref(h)))
        return cls(h)

    @classmethod
    def from_tables(cls, dens_off, means, inv_vars, norm, logw, max_approx=True, device=0):
        dens_off = np.ascontiguousarray(dens_off, dtype=np.uint32)
        means = np.ascontiguousarray(means, dtype=np.float64)
        inv_vars = np.ascontiguousarray(inv_vars, dtype=np.float64)
        norm = np.ascontiguousarray(norm, dtype=np.float64)
        logw = np.ascontiguousarray(logw, dtype=np.float64)
        h = C.c_void_p()
        _check(lib().sr_model_create(device, means.shape[1], len(dens_off) - 1, _ptr(dens_off), _ptr(means), _ptr(inv_vars),
                                     _ptr(norm), _ptr(logw), int(max_approx), C.byref(h)))
        return cls(h)

    @classmethod
    def from_statistics(cls, dim, dens_off, dens_mean, dens_var, acc, pooling=POOL_NONE, max_approx=True, device=0):
        """MixtureModel::finalize on EM statistics `acc` = (mean_acc, mean_w, var_acc, var_w) -> new device model."""
        dens_off, dens_mean, dens_var = (np.ascontiguousarray(x, dtype=np.uint32) for x in (dens_off, dens_mean, dens_var))
        ma, mw, va, vw = (np.ascontiguousarray(x, dtype=np.float64) for x in acc)
        h = C.c_void_p()
        _check(lib().sr_model_create_from_statistics(device, dim, len(dens_off) - 1, _ptr(dens_off), len(mw), len(vw), _ptr(dens_mean),
                                                     _ptr(dens_var), _ptr(ma), _ptr(mw), _ptr(va), _ptr(vw), pooling, int(max_approx),
                                                     C.byref(h)))
        return cls(h)

    def topology(self):
        """-> (dens_off u32[S+1], dens_mean u32[C], dens_var u32[C]): what mixset_write / from_statistics take."""
        off = np.zeros(self.n_states + 1, np.uint32)
        dm, dv = np.zeros(self.n_densities, np.uint32), np.zeros(self.n_densities, np.uint32)
        _check(lib().sr_model_topology(self.h, _ptr(off), _ptr(dm), _ptr(dv)))
        return off, dm, dv

    def close(self):
        if self.h:
            lib().sr_model_destroy(self.h)
            self.h = None

    def trim(self):
        """sr_model_trim: releases the device buffers the model keeps between calls."""
        _check(lib().sr_model_trim(self.h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- scoring -----------------------------------------------------------------------------------
    def score_frames(self, feats, kernel=GMM_PREFILTER):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        out = np.empty((feats.shape[0], self.n_states), dtype=np.float64)
        _check(lib().sr_score_frames(self.h, _ptr(feats), feats.shape[0], kernel, _ptr(out)))
        return out

    def upload(self, feats, frame_off, asynchronous=False):
        return Corpus(self, feats, frame_off, asynchronous)

    def recognize_batch(self, lexicon, feats, frame_off, am_threshold, word_penalty, kernel=GMM_PREFILTER):
        """sr_recognize_batch: host buffers in, words out (fed asynchronously while the first chunks are scored)."""
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        frame_off = np.ascontiguousarray(frame_off, dtype=np.uint64)
        n = len(frame_off) - 1
        words = np.zeros(max(int(frame_off[-1]), 1), dtype=np.uint32)
        woff = np.zeros(n + 1, dtype=np.uint64)
        sp = SearchParams(am_threshold, word_penalty, kernel, 0)
        _check(lib().sr_recognize_batch(self.h, lexicon.h, C.byref(sp), _ptr(feats), _ptr(frame_off), n, _ptr(words), _ptr(woff)))
        return words[: int(woff[-1])].copy(), woff

    def lexicon(self, word_off, automaton, silence_idx, tdp, silence_state):
        return Lexicon(self, word_off, automaton, silence_idx, tdp, silence_state)

    def bigram(self, word_off, mixtures, silence_word, lm, tdp):
        return Bigram(self, word_off, mixtures, silence_word, lm, tdp)

    # -- profiling ---------------------------------------------------------------------------------
    def profile(self, on=True):
        _check(lib().sr_profile_enable(self.h, int(on)))
        _check(lib().sr_profile_reset(self.h))

    def profile_read(self):
        p = Profile()
        _check(lib().sr_profile_read(self.h, C.byref(p)))
        return {k: getattr(p, k) for k, _ in Profile._fields_}


FLT_MAX = float(np.finfo(np.float32).max)


class Bigram:
    """sr_bigram handle: linear lexicon + dense bigram table + transition scores (Teaching::LinearSearch)."""

    def __init__(self, model, word_off, mixtures, silence_word, lm, tdp):
        self.model = model
        word_off = np.ascontiguousarray(word_off, dtype=np.uint32)
        mixtures = np.ascontiguousarray(mixtures, dtype=np.uint16)
        W = len(word_off) - 1
        lm = np.ascontiguousarray(lm, dtype=np.float32)
        assert lm.shape == (W, W)
        tdp = np.ascontiguousarray(tdp, dtype=np.float32)
        assert tdp.size == 8
        self.h = C.c_void_p()
        _check(lib().sr_bigram_create(model.h, W, _ptr(word_off), _ptr(mixtures), silence_word, _ptr(lm), _ptr(tdp), C.byref(self.h)))

    def close(self):
        if self.h:
            lib().sr_bigram_destroy(self.h)
            self.h = None


class Corpus:
    """sr_corpus handle: device-resident utterance batch (Corpus layout, sietill/Corpus.cpp:89-111)."""

    def __init__(self, model, feats, frame_off, asynchronous=False):
        self.model = model
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        self.frame_off = np.ascontiguousarray(frame_off, dtype=np.uint64)
        self.n_utts = len(self.frame_off) - 1
        self.n_frames = int(self.frame_off[-1])
        self.h = C.c_void_p()
        if asynchronous:  # the feeder borrows `feats` until wait() / close()
            self._borrowed = feats
            _check(lib().sr_corpus_upload_async(model.h, _ptr(feats), _ptr(self.frame_off), self.n_utts, C.byref(self.h)))
        else:
            _check(lib().sr_corpus_upload(model.h, _ptr(feats), _ptr(self.frame_off), self.n_utts, C.byref(self.h)))

    def wait(self):
        _check(lib().sr_corpus_wait(self.h))
        self._borrowed = None

    def close(self):
        if self.h:
            lib().sr_corpus_destroy(self.h)
            self.h = None

    def score(self, kernel=GMM_PREFILTER):
        out = np.empty((self.n_frames, self.model.n_states), dtype=np.float64)
        _check(lib().sr_score_corpus(self.model.h, self.h, kernel, _ptr(out)))
        return out

    def recognize(self, lexicon, am_threshold, word_penalty, kernel=GMM_PREFILTER, traceback=False, general_kernel=False, slot_kernel=False):
        """-> (words u32[], word_off u64[n_utts+1]) [, (tb_score, tb_word, tb_bkp)]"""
        words = np.zeros(max(self.n_frames, 1), dtype=np.uint32)
        woff = np.zeros(self.n_utts + 1, dtype=np.uint64)
        sp = SearchParams(am_threshold, word_penalty, kernel, (SEARCH_GENERAL_KERNEL if general_kernel else 0) | (SEARCH_SLOT_KERNEL if slot_kernel else 0))
        tbs = tbw = tbb = None
        if traceback:
            n = self.n_frames + self.n_utts
            tbs, tbw, tbb = np.zeros(n, np.float64), np.zeros(n, np.uint16), np.zeros(n, np.uint16)
        _check(lib().sr_recognize_corpus(self.model.h, self.h, lexicon.h, C.byref(sp), _ptr(words), _ptr(woff), _ptr(tbs),
                                         _ptr(tbw), _ptr(tbb)))
        words = words[: int(woff[-1])].copy()
        if traceback:
            return words, woff, (tbs, tbw, tbb)
        return words, woff

    def retrace(self, lexicon, tb_word, tb_bkp):
        """sr_traceback_corpus: the device's traceback walk alone on dumps in recognize(traceback=True)'s layout."""
        tb_word = np.ascontiguousarray(tb_word, dtype=np.uint16)
        tb_bkp = np.ascontiguousarray(tb_bkp, dtype=np.uint16)
        assert len(tb_word) == len(tb_bkp) == self.n_frames + self.n_utts
        words = np.zeros(max(self.n_frames, 1), dtype=np.uint32)
        woff = np.zeros(self.n_utts + 1, dtype=np.uint64)
        _check(lib().sr_traceback_corpus(self.model.h, self.h, lexicon.h, _ptr(tb_word), _ptr(tb_bkp), _ptr(words), _ptr(woff)))
        return words[: int(woff[-1])].copy(), woff

    def _aut(self, automata):
        flat = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.uint16) for a in automata]), dtype=np.uint16)
        off = np.concatenate([[0], np.cumsum([len(a) for a in automata])]).astype(np.uint64)
        return flat, off

    def align(self, automata, tdp, silence_state, kernel=GMM_PREFILTER, pruning_threshold=None):
        """automata: one state-id sequence per utterance -> (states u16[total_frames], cost f64[n_utts])"""
        flat, off = self._aut(automata)
        states = np.zeros(max(self.n_frames, 1), dtype=np.uint16)
        cost = np.zeros(max(self.n_utts, 1), dtype=np.float64)
        t3 = (C.c_double * 3)(*tdp)
        if pruning_threshold is None:
            _check(lib().sr_align_corpus(self.model.h, self.h, _ptr(flat), _ptr(off), C.byref(t3), silence_state, kernel,
                                         _ptr(states), _ptr(cost)))
        else:
            _check(lib().sr_align_corpus_pruned(self.model.h, self.h, _ptr(flat), _ptr(off), C.byref(t3), silence_state,
                                                float(pruning_threshold), kernel, _ptr(states), _ptr(cost)))
        return states[: self.n_frames], cost[: self.n_utts]

    def accumulate_on_device(self, states, first_pass=False, max_approx=True):
        """The same, but the statistics stay in this corpus handle on the device (for next_model())."""
        states = np.ascontiguousarray(states, dtype=np.uint16)
        _check(lib().sr_accumulate_corpus(self.model.h, self.h, _ptr(states), int(first_pass), int(max_approx), None, None, None, None))

    def next_model(self, pooling=POOL_NONE, max_approx=True):
        """MixtureModel::finalize of the statistics accumulate_on_device() left on the device -> new Model."""
        h = C.c_void_p()
        _check(lib().sr_model_create_from_accumulated(self.model.h, self.h, pooling, int(max_approx), C.byref(h)))
        return Model(h)

    def accumulate(self, states, first_pass=False, max_approx=True):
        """EM statistics of an alignment (MixtureModel::accumulate) -> (mean_acc, mean_w, var_acc, var_w)."""
        states = np.ascontiguousarray(states, dtype=np.uint16)
        nm, nv = C.c_uint32(), C.c_uint32()
        _check(lib().sr_model_tying_info(self.model.h, C.byref(nm), C.byref(nv)))
        D = self.model.dim
        ma, mw = np.zeros((nm.value, D)), np.zeros(nm.value)
        va, vw = np.zeros((nv.value, D)), np.zeros(nv.value)
        _check(lib().sr_accumulate_corpus(self.model.h, self.h, _ptr(states), int(first_pass), int(max_approx), _ptr(ma), _ptr(mw),
                                          _ptr(va), _ptr(vw)))
        return ma, mw, va, vw

    def recognize_bigram(self, bigram, acoustic_pruning=FLT_MAX, lm_pruning=FLT_MAX, kernel=GMM_PREFILTER, max_word_ends=0, dense_states=False):
        """-> (words u32[], scores f32[], times u32[], off u64[n_utts+1]): LinearSearch::getResult per utterance"""
        cap = max(self.n_frames + self.n_utts, 1)
        ow, osc, ot = np.zeros(cap, np.uint32), np.zeros(cap, np.float32), np.zeros(cap, np.uint32)
        off = np.zeros(self.n_utts + 1, np.uint64)
        p = BigramParams(acoustic_pruning, lm_pruning, kernel, max_word_ends, BIGRAM_DENSE_STATES if dense_states else 0)
        _check(lib().sr_recognize_bigram_corpus(self.model.h, self.h, bigram.h, C.byref(p), _ptr(ow), _ptr(osc), _ptr(ot), _ptr(off)))
        n = int(off[-1])
        return ow[:n], osc[:n], ot[:n], off

    def path_scores(self, states, kernel=GMM_PREFILTER):
        """Emission cost along a state path (one state per frame): Trainer::calc_am_score's summands."""
        states = np.ascontiguousarray(states, dtype=np.uint16)
        out = np.zeros(max(self.n_frames, 1), dtype=np.float64)
        _check(lib().sr_path_scores_corpus(self.model.h, self.h, _ptr(states), kernel, _ptr(out)))
        return out[: self.n_frames]


class Lexicon:
    """sr_lexicon handle: flattened Lexicon + TdpModel (sietill/Lexicon.hpp:16-33, TdpModel.hpp:13-29)."""

    def __init__(self, model, word_off, automaton, silence_idx, tdp, silence_state):
        word_off = np.ascontiguousarray(word_off, dtype=np.uint32)
        automaton = np.ascontiguousarray(automaton, dtype=np.uint16)
        self.h = C.c_void_p()
        t3 = (C.c_double * 3)(*tdp)
        _check(lib().sr_lexicon_create(model.h, len(word_off) - 1, _ptr(word_off), _ptr(automaton), silence_idx, C.byref(t3),
                                       silence_state, C.byref(self.h)))

    def describe(self):
        """Which search kernel sr_recognize_corpus runs on this lexicon (sr_lexicon_describe)."""
        buf = C.create_string_buffer(128)
        _check(lib().sr_lexicon_describe(self.h, buf, len(buf)))
        return buf.value.decode()

    def close(self):
        if self.h:
            lib().sr_lexicon_destroy(self.h)
            self.h = None


def traceback_words(tb_word, tb_bkp, silence_word, n_words):
    """sr_traceback_words: Recognizer.cpp:222-231 on one utterance's traceback[0..T] (host side, guarded)."""
    tb_word = np.ascontiguousarray(tb_word, dtype=np.uint16)
    tb_bkp = np.ascontiguousarray(tb_bkp, dtype=np.uint16)
    T = len(tb_word) - 1
    out = np.zeros(max(T, 1), dtype=np.uint32)
    n = C.c_uint32(0)
    _check(lib().sr_traceback_words(T, _ptr(tb_word), _ptr(tb_bkp), silence_word, n_words, _ptr(out), C.byref(n)))
    return out[: n.value].copy()


def mixset_write(path, dim, dens_off, dens_mean, dens_var, acc):
    """MixtureModel::write: EM statistics + topology -> MIXSET v2 file."""
    dens_off, dens_mean, dens_var = (np.ascontiguousarray(x, dtype=np.uint32) for x in (dens_off, dens_mean, dens_var))
    ma, mw, va, vw = (np.ascontiguousarray(x, dtype=np.float64) for x in acc)
    _check(lib().sr_mixset_write(str(path).encode(), dim, len(dens_off) - 1, _ptr(dens_off), len(mw), len(vw), _ptr(dens_mean),
                                 _ptr(dens_var), _ptr(ma), _ptr(mw), _ptr(va), _ptr(vw)))


def shard_utterances(frame_off, n_shards):
    """sr_shard_utterances: greedy LPT deal by frames -> (shard_of_utt u32[n_utts], shard_frames u64[n_shards])."""
    frame_off = np.ascontiguousarray(frame_off, dtype=np.uint64)
    n = len(frame_off) - 1
    shard = np.zeros(max(n, 1), dtype=np.uint32)
    load = np.zeros(n_shards, dtype=np.uint64)
    _check(lib().sr_shard_utterances(_ptr(frame_off), n, n_shards, _ptr(shard), _ptr(load)))
    return shard[:n], load


def recognize_batch_multi(models, lexica, feats, frame_off, am_threshold, word_penalty, kernel=GMM_PREFILTER):
    """sr_recognize_batch_multi over (model, lexicon) replicas, one host thread each -> (words, word_off, shard_frames)."""
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    frame_off = np.ascontiguousarray(frame_off, dtype=np.uint64)
    n, nd = len(frame_off) - 1, len(models)
    mh = (C.c_void_p * nd)(*[m.h for m in models])
    lh = (C.c_void_p * nd)(*[l.h for l in lexica])
    words = np.zeros(max(int(frame_off[-1]), 1), dtype=np.uint32)
    woff = np.zeros(n + 1, dtype=np.uint64)
    load = np.zeros(nd, dtype=np.uint64)
    sp = SearchParams(am_threshold, word_penalty, kernel, 0)
    _check(lib().sr_recognize_batch_multi(mh, lh, nd, C.byref(sp), _ptr(feats), _ptr(frame_off), n, _ptr(words), _ptr(woff), _ptr(load)))
    return words[: int(woff[-1])].copy(), woff, load
