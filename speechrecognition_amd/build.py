"""Builds libsrgpu.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the .so lands next to the sources so it travels with the repo
snapshot to the GPU box (it is git-ignored, not gpurun-ignored).

Staleness is decided by CONTENT, not by mtime: every object carries a stamp = sha256(its source, every shared header,
its flags), taken from the bytes read BEFORE the compiler starts.  An object compiled while a header was being edited
therefore keeps the stamp of the old header and is rebuilt next time (an mtime rule would call it fresh, and the
library could be linked from objects that disagree about a struct in kernels.h).  The library is relinked only after
all objects are current, and records the stamps it was linked from."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsrgpu.so")
SOURCES = ["srgpu_api.cpp", "mixset.cpp", "feeder.cpp", "gmm_mfma.hip", "gmm_exact.hip", "gmm_prefilter.hip", "viterbi_decode.hip",
           "viterbi_fast.hip", "viterbi_words.hip", "viterbi_align.hip", "viterbi_bigram.hip", "em_accumulate.hip", "em_finalize.hip"]
HEADERS = ["kernels.h", "host_util.h", "handles.h", "traceback.h", "dpp_util.h", os.path.join("..", "..", "include", "srgpu.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# these replay the reference's SSE2 operation order and must not contract a*b+c into an FMA
PER_FILE = {"gmm_exact.hip": ["-ffp-contract=off"], "gmm_prefilter.hip": ["-ffp-contract=off"], "em_accumulate.hip": ["-ffp-contract=off"],
            "em_finalize.hip": ["-ffp-contract=off"]}
MAX_PARALLEL = 8


def _read(path):
    with open(path, "rb") as f:
        return f.read()


def _stamp_of(parts):
    h = hashlib.sha256()
    for p in parts:
        h.update(len(p).to_bytes(8, "little"))
        h.update(p)
    return h.hexdigest()


def _stored(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return ""


def build(force=False, verbose=False):
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    sources = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdr_bytes = [_read(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h))]
    objs, todo, stamps = [], [], {}
    for src in sources:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        flags = FLAGS + PER_FILE.get(src, [])
        stamps[src] = _stamp_of([_read(sp)] + hdr_bytes + [" ".join(flags).encode()])
        if force or not os.path.exists(obj) or _stored(obj + ".stamp") != stamps[src]:
            todo.append((src, ["hipcc", "-x", "hip", "-c", sp, "-o", obj] + flags, obj))
    failed = None
    for i in range(0, len(todo), MAX_PARALLEL):
        procs = []
        for src, cmd, obj in todo[i:i + MAX_PARALLEL]:
            if os.path.exists(obj + ".stamp"):
                os.remove(obj + ".stamp")
            if verbose:
                print(" ".join(cmd))
            procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for src, obj, p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                failed = failed or f"hipcc failed on {src}:\n{out}"
                continue
            with open(obj + ".stamp", "w") as f:
                f.write(stamps[src])
            if verbose and out.strip():
                print(out)
    if failed:
        raise RuntimeError(failed)
    link_stamp = _stamp_of([stamps[s].encode() for s in sources])
    if force or todo or not os.path.exists(LIB) or _stored(LIB + ".stamp") != link_stamp:
        if os.path.exists(LIB + ".stamp"):
            os.remove(LIB + ".stamp")
        cmd = ["hipcc", "-shared", "-o", LIB] + objs + ["--offload-arch=gfx950", "-lpthread"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout)
        with open(LIB + ".stamp", "w") as f:
            f.write(link_stamp)
    build_tools(force)
    write_build_info()
    return LIB


TOOLS = os.path.join(os.path.dirname(HERE), "tools")
DEVICE_TOOLS = ["wave_handoff_stress"]  # stand-alone device programs the GPU tests run (tools/<name>.hip -> tools/<name>)


def build_tools(force=False):
    """tools/<name>.hip -> tools/<name> for the device programs a GPU test runs; same content stamps as the library's objects."""
    out = []
    for name in DEVICE_TOOLS:
        src, exe = os.path.join(TOOLS, name + ".hip"), os.path.join(TOOLS, name)
        stamp = _stamp_of([_read(src)])
        if force or not os.path.exists(exe) or _stored(exe + ".stamp") != stamp:
            r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-Wno-unused-result", src, "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}")
            with open(exe + ".stamp", "w") as f:
                f.write(stamp)
        out.append(exe)
    return out


BUILD_INFO = os.path.join(HERE, "BUILD_INFO.json")


def write_build_info():
    """Records the commit the tree was at (and whether it was dirty) next to the library: the GPU box gets the tree without
    .git, and profile summaries / bench lines taken there must still name their commit (tools/summarize_profile.py, bench.py).
    Only written where git answers; the file is git-ignored and travels with the snapshot like the .so."""
    import json
    try:
        root = os.path.dirname(HERE)
        head = subprocess.check_output(["git", "rev-parse", "--short=12", "HEAD"], cwd=root, text=True, stderr=subprocess.DEVNULL).strip()
        dirty = bool(subprocess.check_output(["git", "status", "--porcelain", "--untracked-files=no"], cwd=root, text=True,
                                             stderr=subprocess.DEVNULL).strip())
    except (OSError, subprocess.CalledProcessError):
        return
    with open(BUILD_INFO, "w") as f:
        json.dump({"git_head": head, "dirty": dirty}, f)


def build_info():
    import json
    try:
        with open(BUILD_INFO) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
