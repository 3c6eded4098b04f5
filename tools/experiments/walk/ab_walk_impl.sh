#!/bin/bash
# A/B of two builds of the flattened walk (lib = PB_WALK_IMPL 1, lib_w0 = 0) for force variant 3
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in lib lib_w0; do
  echo "== $lib rep $rep"
  python tools/ab_bench.py --libdir particlerobotsimulations_amd/$lib --variants 3,3w1 --bots 1000000 --rounds 4 --steps 300 --skip 300 --lattice blob 2>&1 | sed -E 's/=> .*(bit-identical)/\1/' | cut -c1-200
  for w in 0 1; do
  PB_ALLOW_ENV_OVERRIDES=1 PB_WALK=$w python tools/bench_with_lib.py $lib --workload ensemble5 --members-per-gpu 16 --force-variant 3 --steps 300 --warmup 200 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ensemble5 x16 walk $w variant 3: us/step %.2f (long %.2f)' % (d['ms_per_step']*1e3, d.get('ms_per_step_long',0)*1e3))"
  done
done
done
python tools/ab_bench.py --variants 3,3w1 --bots 1000000 --rounds 4 --steps 300 --skip 300 2>&1 | sed -E 's/=> .*(bit-identical)/\1/' | cut -c1-200
