#!/usr/bin/env python3
"""Launch cadence of a rocprofv3 --kernel-trace: per kernel name, the launch duration and the GAP between the end of
one launch and the start of the next one on the same queue (what the per-step forms of small ensembles pay between
dependent kernels).  usage: tools/launch_gap.py <dir with *kernel_trace.csv> [name filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "k_force"
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
by_q = defaultdict(list)
for r in rows:
    by_q[(r.get("Queue_Id", "?"), r.get("Stream_Id", "?"))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, v in sorted(by_q.items()):
    v.sort()
    sel = [(a, b, c) for a, b, c in zip(v, v[1:], range(len(v))) if flt in a[2] and flt in b[2]]
    if len(sel) < 100:
        continue
    import statistics as st
    dur = [a[1] - a[0] for a, b, _ in sel]
    gap = [b[0] - a[1] for a, b, _ in sel]
    period = [b[0] - a[0] for a, b, _ in sel]
    name = sel[0][0][2].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
    q50 = lambda x: st.median(x) / 1e3
    print(f"queue {q}: {len(sel)} consecutive pairs of {name}: launch median {q50(dur):.2f} us, gap end->next start median "
          f"{q50(gap):.2f} us (p10 {sorted(gap)[len(gap)//10]/1e3:.2f}, p90 {sorted(gap)[9*len(gap)//10]/1e3:.2f}), "
          f"start->start median {q50(period):.2f} us")
