"""ctypes bindings for the CPU checkers (TEST INFRASTRUCTURE ONLY -- see oracle/sr_oracle.h).

`Oracle`    -> oracle/liboracle.so           (our C restatement; travels to the GPU box)
`Reference` -> oracle/_ref/libsietill_ref.so (the real reference compiled by oracle/Makefile;
               exists only where /root/reference was present at build time)

Importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg. Never the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsietill_ref.so")

POOL_GLOBAL, POOL_MIXTURE, POOL_NONE = 0, 1, 2

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the checkers (liboracle.so always; _ref only when the reference sources exist)."""
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
        os.path.join(HERE, "sr_oracle.c")
    ):
        subprocess.check_call(["make", "-C", HERE, os.path.join(HERE, "liboracle.so")], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src/sietill"):
        subprocess.check_call(["make", "-C", HERE, "-j4", "ref"], stdout=subprocess.DEVNULL)


class _Lex(C.Structure):
    _fields_ = [
        ("n_words", C.c_uint32),
        ("n_states", C.c_uint32),
        ("silence_idx", C.c_uint32),
        ("word_off", C.c_void_p),
        ("automaton", C.c_void_p),
    ]


class _Tdp(C.Structure):
    _fields_ = [("loop", C.c_double), ("forward", C.c_double), ("skip", C.c_double), ("silence_state", C.c_uint16)]


class _Search(C.Structure):
    _fields_ = [("am_threshold", C.c_double), ("word_penalty", C.c_double)]


def _lib():
    if not os.path.exists(ORACLE_SO):
        build()
    L = C.CDLL(ORACLE_SO)
    L.orc_model_load.restype = C.c_void_p
    L.orc_model_load.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.c_int]
    L.orc_model_free.argtypes = [C.c_void_p]
    L.orc_last_error.restype = C.c_char_p
    for name in ("dim", "num_states", "num_means", "num_vars", "num_densities"):
        fn = getattr(L, "orc_model_" + name)
        fn.restype, fn.argtypes = C.c_uint32, [C.c_void_p]
    for name in ("means", "vars_inv", "norm", "logw", "mix_offsets", "mix_mean_idx", "mix_var_idx"):
        fn = getattr(L, "orc_model_" + name)
        fn.restype, fn.argtypes = C.c_void_p, [C.c_void_p]
    L.orc_score.restype = C.c_double
    L.orc_score.argtypes = [C.c_void_p, _f32p, C.c_uint32]
    L.orc_score_matrix.argtypes = [C.c_void_p, _f32p, C.c_size_t, _f64p, C.c_int]
    L.orc_score_argmin.restype = C.c_double
    L.orc_score_argmin.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.orc_decode_pruned.restype = C.c_size_t
    L.orc_decode_pruned.argtypes = [
        C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(_Lex), C.POINTER(_Tdp), C.POINTER(_Search),
        _f32p, C.c_size_t, C.c_uint32, _u32p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64),
    ]
    L.orc_align_full.restype = C.c_double
    L.orc_align_full.argtypes = [
        C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(_Tdp), _u16p, C.c_size_t, _f32p, C.c_size_t, C.c_uint32, _u16p,
    ]
    L.orc_align_pruned.restype = C.c_double
    L.orc_align_pruned.argtypes = [
        C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(_Tdp), _u16p, C.c_size_t, _f32p, C.c_size_t, C.c_uint32,
        C.c_double, _u16p,
    ]
    L.orc_edit_distance.argtypes = [_u64p, C.c_size_t, _u64p, C.c_size_t, _u16p]
    L.orc_bigram_decode.restype = C.c_size_t
    L.orc_bigram_decode.argtypes = [_f64p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, _u32p, _u16p, _f32p, _f32p,
                                    C.c_float, C.c_float, _u32p, _f32p, _u32p, C.c_size_t, _u64p]
    L.orc_accumulate.argtypes = [C.c_void_p, _f32p, C.c_size_t, _u16p, C.c_int, C.c_int, _f64p, _f64p, _f64p, _f64p]
    L.orc_recognize_batch.restype = C.c_double
    L.orc_recognize_batch.argtypes = [
        C.c_void_p, C.POINTER(_Lex), C.POINTER(_Tdp), C.POINTER(_Search), _f32p, _u64p, C.c_size_t, C.c_uint32,
        C.c_int, _u32p, _u64p,
    ]
    return L


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).copy()


class Oracle:
    """Our C restatement of the reference path, loaded from a MIXSET file."""

    def __init__(self, mixset_path, dim, lex, tdp=(3.0, 0.0, 30.0), am_threshold=200.0, word_penalty=10.0,
                 pooling=POOL_NONE, max_approx=True):
        self.L = _lib()
        self.h = self.L.orc_model_load(str(mixset_path).encode(), dim, pooling, int(max_approx))
        if not self.h:
            raise RuntimeError("oracle: " + self.L.orc_last_error().decode())
        self.dim = dim
        self.S = self.L.orc_model_num_states(self.h)
        self.lex = lex
        self.word_off, self.automaton, self.silence_state = lex.flatten()
        self._lex = _Lex(lex.n_words, lex.n_states, lex.silence_idx, self.word_off.ctypes.data, self.automaton.ctypes.data)
        self._tdp = _Tdp(tdp[0], tdp[1], tdp[2], self.silence_state)
        self._sp = _Search(am_threshold, word_penalty)

    def close(self):
        if self.h:
            self.L.orc_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- model tables (finalize) ---------------------------------------------------------------
    def tables(self):
        L, h, D = self.L, self.h, self.dim
        nm, nv, S, C_ = (L.orc_model_num_means(h), L.orc_model_num_vars(h), L.orc_model_num_states(h),
                         L.orc_model_num_densities(h))
        return dict(
            means=_arr(L.orc_model_means(h), nm * D, np.float64).reshape(nm, D),
            vars_inv=_arr(L.orc_model_vars_inv(h), nv * D, np.float64).reshape(nv, D),
            norm=_arr(L.orc_model_norm(h), nv, np.float64),
            logw=_arr(L.orc_model_logw(h), nm, np.float64),
            mix_off=_arr(L.orc_model_mix_offsets(h), S + 1, np.uint32),
            mix_mean=_arr(L.orc_model_mix_mean_idx(h), C_, np.uint32),
            mix_var=_arr(L.orc_model_mix_var_idx(h), C_, np.uint32),
        )

    # -- scoring -------------------------------------------------------------------------------
    def score_matrix(self, feats, n_threads=1):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        out = np.empty((feats.shape[0], self.S), dtype=np.float64)
        self.L.orc_score_matrix(self.h, feats, feats.shape[0], out, n_threads)
        return out

    def argmin_matrix(self, feats):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        out = np.empty((feats.shape[0], self.S), dtype=np.uint32)
        d = C.c_uint32(0)
        for t in range(feats.shape[0]):
            for s in range(self.S):
                self.L.orc_score_argmin(self.h, feats[t].ctypes.data, s, C.byref(d))
                out[t, s] = d.value
        return out

    # -- search --------------------------------------------------------------------------------
    def decode(self, feats, dense=None, traceback=False):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        T = feats.shape[0]
        words = np.zeros(max(T, 1), dtype=np.uint32)
        tbs = np.zeros(T + 1, dtype=np.float64)
        tbw = np.zeros(T + 1, dtype=np.uint16)
        tbb = np.zeros(T + 1, dtype=np.uint16)
        nsc = C.c_uint64(0)
        dptr, dstride = (None, 0)
        if dense is not None:
            dense = np.ascontiguousarray(dense, dtype=np.float64)
            dptr, dstride = dense.ctypes.data, dense.shape[1]
        n = self.L.orc_decode_pruned(self.h, dptr, dstride, C.byref(self._lex), C.byref(self._tdp), C.byref(self._sp),
                                     feats, T, self.dim, words, tbs.ctypes.data, tbw.ctypes.data, tbb.ctypes.data,
                                     C.byref(nsc))
        self.last_n_scored = nsc.value
        if traceback:
            return words[:n].copy(), (tbs, tbw, tbb)
        return words[:n].copy()

    def _dense(self, dense):
        if dense is None:
            return None, 0, None
        dense = np.ascontiguousarray(dense, dtype=np.float64)
        return dense.ctypes.data, dense.shape[1], dense

    def align_full(self, feats, ref, dense=None):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        ref = np.ascontiguousarray(ref, dtype=np.uint16)
        out = np.zeros(feats.shape[0], dtype=np.uint16)
        dptr, dstride, _keep = self._dense(dense)
        cost = self.L.orc_align_full(self.h, dptr, dstride, C.byref(self._tdp), ref, len(ref), feats, feats.shape[0],
                                     self.dim, out)
        return out, cost

    def align_pruned(self, feats, ref, threshold, dense=None):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        ref = np.ascontiguousarray(ref, dtype=np.uint16)
        out = np.zeros(feats.shape[0], dtype=np.uint16)
        dptr, dstride, _keep = self._dense(dense)
        cost = self.L.orc_align_pruned(self.h, dptr, dstride, C.byref(self._tdp), ref, len(ref), feats, feats.shape[0],
                                       self.dim, float(threshold), out)
        return out, cost

    def accumulate(self, feats, states, first_pass=False, max_approx=True):
        """-> (mean_acc [n_mean, D], mean_w [n_mean], var_acc [n_var, D], var_w [n_var])"""
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        states = np.ascontiguousarray(states, dtype=np.uint16)
        nm, nv, D = self.L.orc_model_num_means(self.h), self.L.orc_model_num_vars(self.h), self.dim
        ma, mw = np.zeros((nm, D)), np.zeros(nm)
        va, vw = np.zeros((nv, D)), np.zeros(nv)
        self.L.orc_accumulate(self.h, feats, feats.shape[0], states, int(first_pass), int(max_approx), ma, mw, va, vw)
        return ma, mw, va, vw

    def edit_distance(self, ref, hyp):
        out = np.zeros(4, dtype=np.uint16)
        self.L.orc_edit_distance(np.ascontiguousarray(ref, dtype=np.uint64), len(ref),
                                 np.ascontiguousarray(hyp, dtype=np.uint64), len(hyp), out)
        return out

    def recognize_batch(self, feats, frame_off, n_threads=1):
        """-> (words u32[], word_off u64[n_utts+1], seconds of the utterance loop)."""
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        frame_off = np.ascontiguousarray(frame_off, dtype=np.uint64)
        n = len(frame_off) - 1
        words = np.zeros(max(int(frame_off[-1]), 1), dtype=np.uint32)
        woff = np.zeros(n + 1, dtype=np.uint64)
        secs = self.L.orc_recognize_batch(self.h, C.byref(self._lex), C.byref(self._tdp), C.byref(self._sp), feats,
                                          frame_off, n, self.dim, n_threads, words, woff)
        return words[: int(woff[-1])].copy(), woff, secs


def reference_available():
    return os.path.exists(REF_SO)


class Reference:
    """The real reference classes behind oracle/ref_driver.cpp (only where oracle/_ref was built)."""

    def __init__(self, config_path, dim, lex, pooling=POOL_NONE, max_approx=True):
        self.L = C.CDLL(REF_SO)
        L = self.L
        L.ref_create.restype = C.c_void_p
        L.ref_create.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, _u16p, _u16p, C.c_uint32, C.c_int, C.c_int]
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_num_states.restype = C.c_uint32
        L.ref_num_states.argtypes = [C.c_void_p]
        L.ref_score_matrix.argtypes = [C.c_void_p, _f32p, C.c_size_t, _f64p]
        L.ref_argmin_matrix.argtypes = [C.c_void_p, _f32p, C.c_size_t, _u16p]
        L.ref_decode_pruned.restype = C.c_size_t
        L.ref_decode_pruned.argtypes = [C.c_void_p, _f32p, C.c_size_t, _u64p]
        L.ref_edit_distance.argtypes = [C.c_void_p, _u64p, C.c_size_t, _u64p, C.c_size_t, _u16p]
        L.ref_align_full.restype = C.c_double
        L.ref_align_full.argtypes = [C.c_void_p, _f32p, C.c_size_t, _u16p, C.c_size_t, _u16p]
        L.ref_align_pruned.restype = C.c_double
        L.ref_align_pruned.argtypes = [C.c_void_p, _f32p, C.c_size_t, _u16p, C.c_size_t, C.c_double, _u16p]
        L.ref_accumulate_and_write.argtypes = [C.c_void_p, _f32p, C.c_size_t, _u16p, C.c_int, C.c_int, C.c_char_p]
        self.dim = dim
        self.h = L.ref_create(str(config_path).encode(), dim, lex.n_words,
                              np.ascontiguousarray(lex.word_states, dtype=np.uint16),
                              np.ascontiguousarray(lex.word_reps, dtype=np.uint16), lex.silence_idx, pooling,
                              int(max_approx))
        self.S = L.ref_num_states(self.h)

    def close(self):
        if self.h:
            self.L.ref_destroy(self.h)
            self.h = None

    def score_matrix(self, feats):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        out = np.empty((feats.shape[0], self.S), dtype=np.float64)
        self.L.ref_score_matrix(self.h, feats, feats.shape[0], out)
        return out

    def argmin_matrix(self, feats):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        out = np.empty((feats.shape[0], self.S), dtype=np.uint16)
        self.L.ref_argmin_matrix(self.h, feats, feats.shape[0], out)
        return out

    def decode(self, feats):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        out = np.zeros(max(feats.shape[0], 1), dtype=np.uint64)
        n = self.L.ref_decode_pruned(self.h, feats, feats.shape[0], out)
        return out[:n].astype(np.uint32)

    def edit_distance(self, ref, hyp):
        out = np.zeros(4, dtype=np.uint16)
        self.L.ref_edit_distance(self.h, np.ascontiguousarray(ref, dtype=np.uint64), len(ref),
                                 np.ascontiguousarray(hyp, dtype=np.uint64), len(hyp), out)
        return out

    def align_full(self, feats, ref):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        ref = np.ascontiguousarray(ref, dtype=np.uint16)
        out = np.zeros(feats.shape[0], dtype=np.uint16)
        cost = self.L.ref_align_full(self.h, feats, feats.shape[0], ref, len(ref), out)
        return out, cost

    def accumulate_and_write(self, feats, states, out_path, first_pass=False, max_approx=True):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        states = np.ascontiguousarray(states, dtype=np.uint16)
        self.L.ref_accumulate_and_write(self.h, feats, feats.shape[0], states, int(first_pass), int(max_approx),
                                        str(out_path).encode())

    def align_pruned(self, feats, ref, threshold):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        ref = np.ascontiguousarray(ref, dtype=np.uint16)
        out = np.zeros(feats.shape[0], dtype=np.uint16)
        cost = self.L.ref_align_pruned(self.h, feats, feats.shape[0], ref, len(ref), float(threshold), out)
        return out, cost


FLT_MAX = float(np.finfo(np.float32).max)


def bigram_decode(dense, word_off, mixtures, silence, lm, tdp, acoustic_pruning=FLT_MAX, lm_pruning=FLT_MAX, stats=False):
    """Teaching::LinearSearch restated (oracle/sr_oracle.c, PARITY UNPINNED): dense [T x S] f64 scores, linear lexicon
    (word_off [W+1], mixtures = emission state per position), lm [W x W] f32 (-log p(w|h) at [w, h]),
    tdp [2 x 4] f32 ([isSilence][loop, forward, skip, exit]).  -> (words u32[], scores f32[], times u32[])"""
    L = _lib()
    dense = np.ascontiguousarray(dense, dtype=np.float64)
    T, ld = dense.shape
    word_off = np.ascontiguousarray(word_off, dtype=np.uint32)
    mixtures = np.ascontiguousarray(mixtures, dtype=np.uint16)
    W = len(word_off) - 1
    lm = np.ascontiguousarray(lm, dtype=np.float32)
    assert lm.shape == (W, W)
    tdp = np.ascontiguousarray(tdp, dtype=np.float32)
    assert tdp.shape == (2, 4)
    cap = T + 1
    ow, osc, ot = np.zeros(cap, np.uint32), np.zeros(cap, np.float32), np.zeros(cap, np.uint32)
    st = np.zeros(4, np.uint64)
    n = L.orc_bigram_decode(dense, ld, T, W, silence, word_off, mixtures, lm, tdp, acoustic_pruning, lm_pruning, ow, osc, ot, cap, st)
    assert n <= cap
    res = (ow[:n].copy(), osc[:n].copy(), ot[:n].copy())
    return res + (st,) if stats else res
