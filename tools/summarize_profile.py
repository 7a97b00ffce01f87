"""Condenses a tools/profile_bench.sh run into one JSON: per-kernel average durations (rocprofv3 --stats) and PMC
counters per dispatch, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section): FETCH_SIZE doubled (wide
coalesced reads are tallied at half their bytes), WRITE_SIZE as is, both in KB.

usage: python3 tools/summarize_profile.py gpurun_out/profile_TAG  -> gpurun_out/profile_TAG/summary.json (+ stdout)
"""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

root = sys.argv[1]


def kernel_sources_sha16():
    """Identity of the kernels the counters belong to: sha256 over the device sources (sorted by name), first 16 hex digits.
    bench.py prints `traffic` only while this still matches the tree it runs from."""
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speechrecognition_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


short = lambda n: n.split("(")[0].replace("void ", "").replace("srgpu::", "")
out = {"source": "tools/profile_bench.sh: rocprofv3 --kernel-trace --stats (3 timed steps + 1 warm-up) and one "
                 "rocprofv3 --pmc pass per counter group (1 step + 1 warm-up) of python3 bench.py, MI355X",
       "kernel_sources_sha16": kernel_sources_sha16(), "kernels": {}}
try:
    out["git_head"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL,
                                              cwd=os.path.dirname(os.path.abspath(__file__))).strip()
except (OSError, subprocess.CalledProcessError):
    # the GPU box has no .git: the build recorded the commit next to the library (speechrecognition_amd/build.py, BUILD_INFO.json)
    try:
        info = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speechrecognition_amd", "BUILD_INFO.json")))
        out["git_head"] = info["git_head"] + ("+dirty" if info.get("dirty") else "")
    except (OSError, ValueError, KeyError):
        pass
try:
    out["bench_line_under_profiler"] = json.loads(open(os.path.join(root, "bench_under_profiler.json")).read())
    cfg = out["bench_line_under_profiler"]["config"]
    out["workload_frames_per_launch"] = cfg.get("frames_rank0", cfg.get("frames_per_gpu_rank0"))
except (OSError, ValueError, KeyError):
    pass
for r in csv.DictReader(open(os.path.join(root, "stats", "s_kernel_stats.csv"))):
    if float(r["AverageNs"]) >= 2e4:
        out["kernels"][short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                             "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6,
                                             "share_pct": float(r["Percentage"])}
for path in sorted(glob.glob(os.path.join(root, "pmc*", "p_counter_collection.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    per_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        per_dispatch[(short(r["Kernel_Name"]), r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, _, c), v in per_dispatch.items():
        agg[k][c].append(v)
    for k, cs in agg.items():
        if k in out["kernels"]:
            for c, vals in cs.items():
                out["kernels"][k].setdefault("pmc_mean_per_dispatch", {})[c] = sum(vals) / len(vals)
for k, z in out["kernels"].items():
    p = z.get("pmc_mean_per_dispatch", {})
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        z["hbm_bytes_per_launch_corrected"] = (2.0 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024.0
        z["correction"] = "gfx950: FETCH_SIZE x2 (wide coalesced reads tallied at half their bytes), WRITE_SIZE exact; unit KB"
    if "GRBM_GUI_ACTIVE" in p:
        z["clock_GHz_from_GRBM_GUI_ACTIVE"] = p["GRBM_GUI_ACTIVE"] / 8.0 / (z["avg_ms"] * 1e-3) / 1e9
    if "TCC_HIT_sum" in p:
        z["l2_hit_rate"] = p["TCC_HIT_sum"] / max(1.0, p["TCC_HIT_sum"] + p["TCC_MISS_sum"])
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "pmc_mean_per_dispatch"} for k, v in out["kernels"].items()}, indent=1))
