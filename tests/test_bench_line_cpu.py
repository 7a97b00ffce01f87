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


def test_roofline_fractions_do_not_depend_on_the_number_of_score_chunks():
    """VERDICT r3 #1 / ADVICE r3: a corpus whose score table is cut into chunks takes several scoring launches per step; the
    step's flops must go over the step's kernel time, not over ONE launch's (profiles/r3_bench_cfg4_one_gpu.json showed the
    refinement at 0.62 of its peak and the prefilter at 0.77 that way, twice their real figures)."""
    D, S, frames = 39, 4000, 300000

    def prof(launches_per_step, steps):
        # the same kernels at the same rate: the work of a step takes the same time however it is cut
        return {"gmm_launches": launches_per_step * steps, "refine_ms": 14.0 * steps, "prefilter_ms": 7.0 * steps, "gmm_ms": 21.0 * steps,
                "gmm_flops": 4.0 * D * S * 32 * frames * steps, "refined_densities": 11, "refined_pairs": 10,
                "search_ms": 2.0 * steps, "search_bytes": 48e3 * frames * steps}

    lines = []
    for lps, steps in ((1, 3), (2, 3), (3, 5)):
        a = types.SimpleNamespace(kernel="prefilter", words=1333, mix=32, steps=steps)
        r = bench.gmm_roofline(a, prof(lps, steps), frames, D, S)
        p = bench.prefilter_report(a, prof(lps, steps), frames, D, S)
        assert r["chunks_per_step"] == lps and abs(r["avg_launch_ms"] - 14.0 / lps) < 1e-12 and abs(r["ms_per_step"] - 14.0) < 1e-12
        assert r["frames_per_launch"] == frames / lps
        lines.append((r["frac"], p["roofline_prefilter"]["frac"], p["roofline_prefilter"]["frac_useful"], p["gmm_step"]["ms"],
                      p["gmm_step"]["dense_fp64_equiv_tflops"]))
    for other in lines[1:]:
        for x, y in zip(lines[0], other):
            assert abs(x - y) <= 1e-12 * abs(x)
    # the figures themselves: one exact density per (frame, state) in 14 ms; K = 96 executed on the matrix cores in 7 ms
    assert abs(lines[0][0] - 4.0 * D * S * frames / 14e-3 / 1e12 / bench.FP64_VALU_UNFUSED_PEAK) < 1e-12
    assert abs(lines[0][1] - 2.0 * 96 * 128000 * frames / 7e-3 / 1e12 / bench.F16_MFMA_PEAK_TFLOPS) < 1e-12
    # the dense kernels' line: gmm_flops and gmm_ms are both sums over the launches
    a = types.SimpleNamespace(kernel="mfma", words=1333, mix=32, steps=3)
    one, two = bench.gmm_roofline(a, prof(1, 3), frames, D, S), bench.gmm_roofline(a, prof(2, 3), frames, D, S)
    assert one["frac"] == two["frac"] and two["chunks_per_step"] == 2
