#!/bin/bash
# A/B: every parameter the sweep reads requested in the kernel's first block (lib_ep, -DPB_EARLY_PARAMS=1) against the
# compiler's own placement (lib): single small simulations (event-timed) and configs[3]
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in lib lib_ep; do
  echo "== $lib rep $rep"
  python tools/lanes_sweep.py --libdir particlerobotsimulations_amd/$lib --sizes 300,4000,30000,100000 --forms 8,16 --steps 1500 | tail -4
  python tools/ab_bench.py --libdir particlerobotsimulations_amd/$lib --variants 2 --bots 1000000 --rounds 3 --steps 300 --skip 300 2>&1 | sed -E 's/=> .*(bit-identical)/\1/' | cut -c1-120
  for i in 1 2; do python tools/bench_with_lib.py $lib --workload ensemble4 --steps 12000 --warmup 200 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  configs[3] 32+32: us/step %.3f' % (d['ms_per_step']*1e3))"; done
done; done
