#!/usr/bin/env python3
"""Copy what is judged of a tools/profile.sh run from gpurun_out/prof_<tag>/ (scratch) into profiles/ (tracked):
<tag>.kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>.summary.md, <tag>.traffic.json, <tag>.valu_rate.txt,
<tag>.bench_unprofiled.json.   usage: tools/collect_profile.py <tag> [latest_traffic.json | latest_traffic_both_sums.json]
(the optional name also installs traffic.json as the profile bench.py quotes for that kernel)."""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
pairs = [(stats[0] if stats else None, f"{tag}.kernel_stats.csv"), (os.path.join(src, "summary.md"), f"{tag}.summary.md"),
         (os.path.join(src, "traffic.json"), f"{tag}.traffic.json"), (os.path.join(src, "valu_rate.txt"), f"{tag}.valu_rate.txt"),
         (os.path.join(src, "bench_unprofiled.json"), f"{tag}.bench_unprofiled.json")]
for a, b in pairs:
    if a and os.path.exists(a) and os.path.getsize(a) > 0:
        shutil.copyfile(a, os.path.join(dst, b))
        print("profiles/" + b)
if len(sys.argv) > 2:
    shutil.copyfile(os.path.join(src, "traffic.json"), os.path.join(dst, sys.argv[2]))
    print("profiles/" + sys.argv[2])
