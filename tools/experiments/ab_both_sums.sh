#!/bin/bash
# A/B of the both-sums throughput form: lib = PB_ASUM_XY 1 (the dead-sum trip + the attraction magnitude), lib_asumk =
# PB_ASUM_XY 0 (pbPairEvalK + pbPairAdd, rounds 1-4); each in its own process, the list twice; bit-identity vs the
# dead-sum form on every array both have is checked by ab_bench.py
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in lib lib_asumk; do
  echo "== $lib rep $rep"
  python tools/ab_bench.py --libdir particlerobotsimulations_amd/$lib --variants 2,2s1 --bots 1000000 --rounds 4 --steps 300 --skip 300 2>&1 | sed -E 's/=> .*(bit-identical)/\1/' | cut -c1-200
done; done
python tools/ab_bench.py --libdir particlerobotsimulations_amd/lib --variants 2,2s1 --bots 1000000 --rounds 3 --steps 300 --skip 300 --lattice blob 2>&1 | sed -E 's/=> .*(bit-identical)/\1/' | cut -c1-200
python tools/ab_bench.py --libdir particlerobotsimulations_amd/lib_asumk --variants 2,2s1 --bots 1000000 --rounds 3 --steps 300 --skip 300 --lattice blob 2>&1 | sed -E 's/=> .*(bit-identical)/\1/' | cut -c1-200
