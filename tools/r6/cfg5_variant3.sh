#!/bin/bash
# BASELINE configs[4] whole (Cartesian: 64 fractions x 16 seeds) with the opt-in tolerance kernel, whose flattened walk is
# chosen automatically on these blobs (round 6), and pinned row by row for comparison
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for w in auto 0; do
  if [ $w = 0 ]; then export PB_ALLOW_ENV_OVERRIDES=1 PB_STREAM_WALK=0; fi
  timeout 1500 python3 tools/bench_legs.py --workload ensemble5 --members-total 1024 --force-variant 3 --steps 50 --no-cpu-baseline > gpurun_out/r6/cfg5_variant3_walk_$w.json 2> gpurun_out/r6/cfg5_variant3_walk_$w.err
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r6/cfg5_variant3_walk_$w.json").read().strip().splitlines()[-1])
e = d["end_to_end"]; p = e["pipeline_rank0"][0]
print("configs[4] whole, force variant 3, walk $w: wall", round(e["wall_s"], 2), "device_s", round(p["device_s"], 2), "bound", e["bound_rank0"][0]["bound"], "placements", p["placements_run"], "producers", p["host_threads"], "steady us/step", round(d["ms_per_step"]*1e3, 1))
PY
done
