"""Builds libsrgpu variants with different compile-time tuning constants for A/B timing on the GPU box.

usage: python tools/build_variant.py NAME [--src gmm_prefilter.hip=/path/to/other_version.hip] -DSR_R_THREADS=512 -DSR_R_BATCH=8 ...
  -> speechrecognition_amd/csrc/build/variants/libsrgpu_NAME.so   (select with SRGPU_LIB=<path>)
--src FILE=PATH compiles PATH in place of csrc/FILE (e.g. an earlier commit's version: `git show REV:speechrecognition_amd/csrc/FILE > PATH`);
the substitute is copied next to the real sources first so that its relative includes resolve.
"""
import os, subprocess, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from speechrecognition_amd import build as B

name, defs = sys.argv[1], sys.argv[2:]
subst = {}
while "--src" in defs:
    i = defs.index("--src")
    f, path = defs[i + 1].split("=", 1)
    subst[f] = path
    del defs[i:i + 2]
out_dir = os.path.join(B.CSRC, "build", "variants", name)
os.makedirs(out_dir, exist_ok=True)
objs = []
procs = []
for src in [x for x in B.SOURCES if os.path.exists(os.path.join(B.CSRC, x))]:
    obj = os.path.join(out_dir, src + ".o")
    objs.append(obj)
    path = os.path.join(B.CSRC, src)
    if src in subst:
        path = os.path.join(B.CSRC, f".variant_{name}_{src}")
        with open(subst[src], "rb") as fi, open(path, "wb") as fo:
            fo.write(fi.read())
    cmd = ["hipcc", "-x", "hip", "-c", path, "-o", obj] + B.FLAGS + B.PER_FILE.get(src, []) + defs
    procs.append(subprocess.Popen(cmd))
for p in procs:
    if p.wait() != 0:
        sys.exit(1)
lib = os.path.join(B.CSRC, "build", "variants", f"libsrgpu_{name}.so")
subprocess.check_call(["hipcc", "-shared", "-o", lib] + objs + ["--offload-arch=gfx950"])
for src in subst:
    os.remove(os.path.join(B.CSRC, f".variant_{name}_{src}"))
print(lib)
