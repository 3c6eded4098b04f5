"""Soak: the WHOLE of BASELINE configs[4] (1 024 members of 10^5 bots, 12 000 steps, dead fraction 0 ... 0.40 in 64
values x 16 seeds) through bin/particlebot_ensemble with --csv-dir, i.e. the product path a user of the reference would
run instead of 1 024 runs of it: two lanes of sub-batches, RCCL gather (world of one), every member's own CSV.
Checks: 1 024 files of 21 rows each, finite; the gathered rows agree with the CSVs to the fp32 centroid's accuracy; the
speed toward the light falls with the dead fraction.  usage: python tools/soak_cfg5_csv.py OUTDIR"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/cfg5_csv"
os.makedirs(out, exist_ok=True)
fractions = [0.40 * i / 63.0 for i in range(64)]
cmd = [os.path.join(ROOT, "particlerobotsimulations_amd", "bin", "particlebot_ensemble"),
       os.path.join(ROOT, "examples", "example_dead_cells.cfg"), "--members", "1024", "--seed0", "1000", "--sub-batch", "-1",
       "--set", "nCells", "100000", "--set", "light_x", "-40", "--set", "light_y", "0", "--set", "max_time", "120",
       "--set", "dump_interval", "6", "--csv-dir", os.path.join(out, "csv"), "--out", os.path.join(out, "rows.f32"),
       "--sweep", "nDead"] + [str(int(round(f * 100000))) for f in fractions]
env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_PORT="29471")
t0 = time.time()
r = subprocess.run(cmd, env=env, capture_output=True, text=True)
wall = time.time() - t0
assert r.returncode == 0, r.stderr[-2000:]
line = json.loads(r.stdout.strip().splitlines()[-1])
files = sorted(os.listdir(os.path.join(out, "csv")))
assert len(files) == 1024, len(files)
rows = np.fromfile(os.path.join(out, "rows.f32"), np.float32).reshape(1024, -1, 4)
nrows = rows.shape[1]
progress = np.zeros(1024)
worst = 0.0
for k, name in enumerate(files):
    text = open(os.path.join(out, "csv", name)).read().splitlines()
    assert text[0] == f"Seed, {1000 + k}" and text[1] == "Time,Centroid X, Centroid Y, Distance", text[:2]
    body = np.array([[float(x) for x in ln.rstrip(",").split(",")] for ln in text[2:]])
    assert body.shape == (nrows, 4) and np.isfinite(body).all(), (name, body.shape)
    worst = max(worst, float(np.abs(body[:, 1:3] - rows[k, :, 1:3]).max()))
    progress[k] = body[0, 3] - body[-1, 3]
by_fraction = progress.reshape(16, 64).mean(0)
trend = np.polyfit(np.arange(64), by_fraction, 1)[0]
summary = {"wall_s_runner": line["wall_s"], "wall_s_command": wall, "members": 1024, "rows_per_member": int(nrows),
           "csv_vs_gathered_rows_max_abs": worst, "progress_dead_0": float(by_fraction[0]),
           "progress_dead_0.40": float(by_fraction[-1]), "progress_slope_per_fraction_step": float(trend),
           "monotone_pairs": int((np.diff(by_fraction) < 0).sum()), "pipeline_rank0": line["pipeline_rank0"]}
assert trend < 0 and by_fraction[0] > by_fraction[-1] > 0
print(json.dumps(summary))
