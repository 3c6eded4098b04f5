#!/bin/bash
# A/B: neighbour velocities fetched inside the contact block (ships) against prefetched with every
# posrad (0, round 2's form) on blobs, where the contact block runs in 36 % of the trips instead of 22 %
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in lib lib_ev; do
  echo "== $lib rep $rep"
  python tools/ab_bench.py --libdir particlerobotsimulations_amd/$lib --variants 2 --bots 1000000 --rounds 3 --steps 300 --skip 300 --lattice blob 2>&1 | sed -E 's/=> .*(bit-identical)/\1/' | cut -c1-160
  python tools/ab_bench.py --libdir particlerobotsimulations_amd/$lib --variants 2 --bots 1000000 --rounds 3 --steps 300 --skip 300 2>&1 | sed -E 's/=> .*(bit-identical)/\1/' | cut -c1-160
  python tools/bench_with_lib.py $lib --workload ensemble5 --members-per-gpu 16 --steps 300 --warmup 200 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ensemble5 x16: us/step %.2f (long %.2f)' % (d['ms_per_step']*1e3, d.get('ms_per_step_long',0)*1e3))"
done; done
