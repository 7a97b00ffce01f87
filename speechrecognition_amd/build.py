"""Builds libsrgpu.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the .so lands next to the sources so it travels with the repo
snapshot to the GPU box (it is git-ignored, not gpurun-ignored)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsrgpu.so")
SOURCES = ["srgpu_api.cpp", "mixset.cpp", "gmm_mfma.hip", "gmm_exact.hip", "gmm_prefilter.hip", "viterbi_decode.hip", "viterbi_fast.hip", "viterbi_align.hip", "viterbi_bigram.hip", "em_accumulate.hip"]
HEADERS = ["kernels.h", "host_util.h", os.path.join("..", "..", "include", "srgpu.h")]
FLAGS = (["-DSR_DECODE_STAMPS"] if __import__("os").environ.get("SR_DECODE_STAMPS") else []) + ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# gmm_exact.hip must not contract a*b+c into an FMA: it replays the reference's SSE2 operation order
PER_FILE = {"gmm_exact.hip": ["-ffp-contract=off"], "gmm_prefilter.hip": ["-ffp-contract=off"], "em_accumulate.hip": ["-ffp-contract=off"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, procs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + hdrs):
            cmd = ["hipcc", "-x", "hip", "-c", sp, "-o", obj] + FLAGS + PER_FILE.get(src, [])
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    if force or procs or _stale(LIB, objs):
        cmd = ["hipcc", "-shared", "-o", LIB] + objs + ["--offload-arch=gfx950"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
