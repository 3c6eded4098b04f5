#!/bin/bash
# A/B: k_force's arguments reordered so that everything the start of a launch needs is inside the 16 preloaded dwords
# (lib_args) against the former order (lib): small batches (latency-bound), single small simulations, the headline
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in lib lib_args; do
  echo "== $lib rep $rep"
  for m in 8 32; do echo -n "  members $m: "; python tools/ens_step_cost.py --libdir particlerobotsimulations_amd/$lib --members $m --steps 4000 | tail -1; done
  python tools/lanes_sweep.py --libdir particlerobotsimulations_amd/$lib --sizes 300,4000,30000 --forms 8,16 --steps 1500 | tail -3
  python tools/ab_bench.py --libdir particlerobotsimulations_amd/$lib --variants 2 --bots 1000000 --rounds 3 --steps 300 --skip 300 2>&1 | sed -E 's/=> .*(bit-identical)/\1/' | cut -c1-120
done; done
