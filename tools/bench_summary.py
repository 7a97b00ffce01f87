"""Prints the handful of numbers of a bench.py JSON line that an optimisation round looks at."""
import json
import sys

d = json.load(open(sys.argv[1]))
r, p, s, g = d["roofline"], d.get("roofline_prefilter", {}), d["search"], d.get("gmm_step", {})
print(f"step {d['ms_per_step']:.2f} ms  value {d['value'] / 1e6:.2f} M frames/s | {r['kernel']} {r['avg_launch_ms']:.2f} ms frac {r['frac']:.3f} | "
      f"prefilter {p.get('avg_launch_ms', 0):.2f} ms frac {p.get('frac', 0):.3f} | search {s['ms_per_step']:.2f} ms | "
      f"dens/pair {g.get('densities_refined_per_pair', 0):.4f} | cpu words match {d.get('cpu_baseline', {}).get('words_match_gpu')}")
