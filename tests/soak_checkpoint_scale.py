#!/usr/bin/env python3
"""Manual soak (not collected by pytest): exact sweep checkpoints at BASELINE config 5's member size.
M members of examples/example_dead_cells.cfg at nCells 100000 (dead fractions swept) through the ensemble pipeline:
(a) uninterrupted for S steps; (b) stopped after S/2 steps with --checkpoint semantics, resumed from the directory,
run to S: the summary rows and the final states must be bit-identical, and (b)'s resume must restore, not re-place.
    python tests/soak_checkpoint_scale.py [members=24] [steps=2400] [dir=/tmp/pb_ckpt_scale]"""
import os
import shutil
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from particlerobotsimulations_amd import ensemble

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = int(sys.argv[1]) if len(sys.argv) > 1 else 24
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2400
D = sys.argv[3] if len(sys.argv) > 3 else "/tmp/pb_ckpt_scale"
cfg = os.path.join(ROOT, "examples", "example_dead_cells.cfg")
members = [f"seed\n{7000 + k}\nnDead\n{(k % 8) * 5000}" for k in range(M)]
common = {"nCells": "100000", "max_time": "1e9", "dump_interval": "6", "light_x": "-40", "light_y": "0"}
shutil.rmtree(D, ignore_errors=True)

t0 = time.perf_counter()
a = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=8, host_threads=16, keep_final_states=True)
assert a.run(S) == S
ta = time.perf_counter() - t0
t0 = time.perf_counter()
b = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=8, host_threads=16, checkpoint_dir=D)
assert b.run(S // 2) == S // 2
b.close()
tb1 = time.perf_counter() - t0
size = sum(os.path.getsize(os.path.join(D, f)) for f in os.listdir(D))
t0 = time.perf_counter()
b2 = ensemble.PipelinedEnsemble(cfg, members, common, sub_batch=8, host_threads=16, keep_final_states=True,
                                checkpoint_dir=D, resume=True)
assert b2.run(S) == S
tb2 = time.perf_counter() - t0
assert np.array_equal(a.rows.view(np.uint32), b2.rows.view(np.uint32)), "summary rows differ"
sa, sb = a.final_states(), b2.final_states()
for k in range(M):
    for key in ("pos", "vel", "rad"):
        assert np.array_equal(np.asarray(sa[k][key]).view(np.uint32), np.asarray(sb[k][key]).view(np.uint32)), (k, key)
print(f"OK {M} members x 100000 bots: stopped after {S // 2} of {S} steps and resumed = uninterrupted, bit for bit "
      f"(rows {a.rows.shape}, final pos/vel/rad of every member); uninterrupted {ta:.1f} s, first half with checkpoints "
      f"{tb1:.1f} s ({size / 1e6:.0f} MB under {D}), resumed half {tb2:.1f} s; placement CPU-seconds "
      f"{a.timings['placement_cpu_s']:.1f} uninterrupted / {b2.timings['placement_cpu_s']:.1f} resumed")
shutil.rmtree(D, ignore_errors=True)
