#!/usr/bin/env python3
"""Print the figures of a bench.py / tools/bench_legs.py JSON line that the docs quote.  usage: tools/show_bench.py file.json"""
import json
import sys

raw = open(sys.argv[1]).read().strip().splitlines()[-1]
d = json.loads(raw)
print("line bytes", len(raw) + 1, "| value", f"{d['value']:.4g}", d["unit"], "| ms/step", round(d["ms_per_step"], 5), "| n_gpus",
      d["n_gpus"], "| steps", d["steps"])
print("config", {k: v for k, v in d["config"].items() if k not in ("workload", "force_sums_note")})
r = d.get("roofline")
if r:
    print("roofline frac", round(r["frac"], 4), "achieved", round(r["achieved"], 1), r["unit"], "avg_launch_us",
          round(r.get("avg_launch_us", 0), 2), "kernel", r.get("kernel"), "traffic", r.get("traffic"), "valu_frac_of_datasheet",
          r.get("valu_frac_of_datasheet"), "profile", r.get("profile"), "long", r.get("long"))
    print("  64 * value / n_gpus / 8e12 =", round(64.0 * d["value"] / d["n_gpus"] / 8e12, 4))
keep = ("value", "us_per_step", "ms_per_step", "us_per_step_per_1e6_bots", "cores", "sims_per_s", "value_1_thread",
        "finite_at_end", "bots", "members", "frac_at_56B", "kernel", "wall_s", "value_end_to_end", "members_total", "backend",
        "ranks", "traffic", "valu_frac_of_datasheet")
for k in ("default_form", "both_sums", "large_arena", "random_blob", "streamlined", "cpu_baseline", "ensemble", "ensemble_leg",
          "survey_literal_lattice", "collective"):
    x = d.get(k)
    if x:
        print(k, {kk: x[kk] for kk in x if kk in keep})
