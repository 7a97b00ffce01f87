"""Host-side pieces of bench.py's JSON line that need no GPU: the `traffic` figure is reported only from a PMC summary whose
kernel-source hash matches the tree, the CPU model string, the decoder-geometry report."""
import json
import os
import types

import bench


def _args(kernel="prefilter", words=1333, mix=32):
    return types.SimpleNamespace(kernel=kernel, words=words, mix=mix)


def test_traffic_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    sha = bench.kernel_sources_sha16()
    assert len(sha) == 16 and int(sha, 16) >= 0
    prof = tmp_path / "profiles"
    prof.mkdir()
    good = {"workload_frames_per_launch": 1000, "kernel_sources_sha16": sha, "git_head": "abc1234",
            "kernels": {"gmm_refine_kernel<39, 32, 8>": {"hbm_bytes_per_launch_corrected": 123.0}}}
    (prof / "r3_prefilter_summary.json").write_text(json.dumps(good))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_sources_sha16", lambda: sha)
    assert bench.pmc_traffic(_args(), 1000, "gmm_refine_kernel") == 123.0
    src = bench.traffic_source(_args(), 1000)
    assert src["git_head"] == "abc1234" and src["kernel_sources_sha16"] == sha and src["file"].endswith("r3_prefilter_summary.json")
    assert bench.pmc_traffic(_args(), 999, "gmm_refine_kernel") is None           # another workload
    assert bench.pmc_traffic(_args(mix=16), 1000, "gmm_refine_kernel") is None     # another model
    monkeypatch.setattr(bench, "kernel_sources_sha16", lambda: "0" * 16)          # a kernel changed since the counters were taken
    assert bench.pmc_traffic(_args(), 1000, "gmm_refine_kernel") is None
    assert isinstance(bench.traffic_source(_args(), 1000), str)


def test_committed_summary_matches_the_tree_or_is_silent():
    """Whatever is committed under profiles/ either belongs to these kernel sources (then bench.py prints its traffic figure) or
    is ignored; it is never printed for other sources."""
    z = bench.pmc_summary(_args(), 302685)
    if z is not None:
        assert z["kernel_sources_sha16"] == bench.kernel_sources_sha16()
        assert bench.pmc_traffic(_args(), 302685, "gmm_refine_kernel") > 1e9


def test_cpu_model_and_core_count():
    name, n = bench.cpu_model()
    assert isinstance(name, str) and name and n >= 1
    assert 1 <= bench.usable_cores() <= 16


def test_decode_geometry_report():
    assert bench.decode_geometry(4096, 1000) == "1024, 4"
    assert bench.decode_geometry(1216, 1) == "1024, 2"
    assert bench.decode_geometry(9000, 10) is None
