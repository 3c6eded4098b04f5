"""Experiment (round 4): do two placement/stepping pipelines on one GPU, each stepping its own sub-batches on its
own stream, beat one pipeline with sub-batches twice the size?  (A launch's ramp and drain, ~13 us of every step of
a sub-batch, would overlap the other pipeline's steady state; both sub-batches together still fit the Infinity
Cache.)  BASELINE configs[4] members, a slice of M members, full length.
usage: python tools/experiments/two_pipelines.py [M=240]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from particlerobotsimulations_amd import ensemble  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 240
cfg = os.path.join(ROOT, "examples", "example_dead_cells.cfg")
common = {"max_time": "120", "dump_interval": "6", "nCells": "100000", "light_x": "-40", "light_y": "0"}
over = []
for k in range(M):
    f = 0.40 * (k % 64) / 63.0
    over.append(f"seed\n{1000 + k // 64}\nnDead\n{int(round(f * 100000))}")


def run(parts, sub, threads):
    t0 = time.perf_counter()
    pipes = [ensemble.PipelinedEnsemble(cfg, over[i::parts], common, sub_batch=sub, host_threads=threads) for i in range(parts)]
    done = [0] * parts

    def one(i):
        done[i] = pipes[i].run(12001)
    th = [threading.Thread(target=one, args=(i,)) for i in range(1, parts)]
    for t in th:
        t.start()
    one(0)
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    tm = [p.timings for p in pipes]
    rows = [p.rows for p in pipes]
    for p in pipes:
        p.close()
    print(f"{parts} pipeline(s), sub-batch {sub}, {threads} producers each: wall {wall:.2f} s, device_s "
          f"{[round(t['device_s'], 2) for t in tm]}, waited {[round(t['placement_wait_s'], 2) for t in tm]}", flush=True)
    return rows


a = run(1, 30, 15)
b = run(2, 15, 7)
c = run(2, 30, 7)
d = run(3, 10, 5)
import numpy as np  # noqa: E402
# same members, same rows whatever the split
for parts, rows in ((2, b), (2, c), (3, d)):
    for i in range(parts):
        assert np.array_equal(np.asarray(a[0])[i::parts], np.asarray(rows[i])), (parts, i)
print("rows identical in every split")
