#!/bin/bash
# per-step forms of BATCHES of small simulations (BASELINE configs[3]'s members): us per step by lanes per bot and
# members per batch, each batch alone and both together (tools/ens_step_cost.py); 0 = the automatic choice
cd $GRAFT_REPO_ROOT
export PB_ALLOW_ENV_OVERRIDES=1
for m in 8 16 32 64 128; do
  echo "== $m members, automatic:"; python tools/ens_step_cost.py --members $m --steps 3000 | tail -1
  for L in 4 8 16 32; do
    echo -n "   L=$L per-step: "; PB_LANES_PER_BOT=$L PB_RESIDENT=1 python tools/ens_step_cost.py --members $m --steps 3000 | tail -1
  done
done
