#!/usr/bin/env python3
"""Print the figures of a bench.py JSON line that the docs quote.  usage: tools/show_bench.py file.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", f"{d['value']:.4g}", d["unit"], "| ms/step", round(d["ms_per_step"], 5), "| n_gpus", d["n_gpus"], "| steps", d["steps"])
print("config", {k: v for k, v in d["config"].items() if k not in ("workload", "force_sums_note")})
r = d.get("roofline")
if r:
    print("roofline frac", round(r["frac"], 4), "achieved", round(r["achieved"], 1), r["unit"], "avg_launch_us",
          round(r.get("avg_launch_us", 0), 2), "clock", r.get("shader_clock_mhz"), "traffic", r.get("traffic"))
    if "frac_dead_sum" in r:
        ds = r.get("dead_sum", {})
        print("  frac (both sums, 64 B)", round(r["frac"], 4), "| frac_dead_sum (56 B)", round(r["frac_dead_sum"], 4),
              "(long", round(r.get("frac_dead_sum_long", 0), 4), ") avg_launch_us", round(ds.get("avg_launch_us", 0), 2),
              "long", round(ds.get("avg_launch_us_long", 0), 2), "| priced at 64 as rounds 1-4:",
              round(r.get("frac_dead_sum_priced_at_64", 0), 4), "| valu_frac_of_datasheet",
              r.get("valu_frac_of_datasheet") and round(r["valu_frac_of_datasheet"], 3), "both sums",
              r.get("valu_frac_of_datasheet_both_sums") and round(r["valu_frac_of_datasheet_both_sums"], 3))
    v = r.get("valu")
    if v:
        print("valu: per wave", round(v["valu_insts_per_wave"]), "trans", round(v["trans_per_wave"]),
              "frac datasheet@measured", v.get("frac_datasheet_at_measured_clock"), "frac microbench", v.get("frac_microbenchmark_rate"))
keep = ("value", "us_per_step", "ms_per_step", "us_per_step_per_1e6_bots", "cores", "sims_per_s", "value_1_thread",
        "finite_at_end", "bots", "members")
for k in ("both_sums", "large_arena", "random_blob", "streamlined", "cpu_baseline", "ensemble_leg", "survey_literal_lattice"):
    x = d.get(k)
    if x:
        print(k, {kk: x[kk] for kk in x if kk in keep})
for k in ("sims_per_s", "members_per_rank", "placement_s"):
    if k in d:
        print(k, d[k])
