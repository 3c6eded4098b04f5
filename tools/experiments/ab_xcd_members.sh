#!/bin/bash
# A/B: batches of small simulations with all tiles of a member on ONE XCD (PB_XCD_MEMBERS=1) against the plain
# (tile, member) grid; BASELINE configs[3] at 32 + 32 and at 8 + 8 members per GPU; the summaries must not change
cd $GRAFT_REPO_ROOT
export PB_ALLOW_ENV_OVERRIDES=1
for m in 32 8 64; do for rep in 1 2; do for x in 0 1; do
  PB_XCD_MEMBERS=$x python tools/bench_legs.py --workload ensemble4 --members-per-gpu $m --steps 8000 --warmup 200 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys,hashlib
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
h=d['steps'] + d.get('steps_long', 0)
print('members $m xcd $x: us/step %.3f (long %.3f) steps in all %s' % (d['ms_per_step']*1e3, d.get('ms_per_step_long',0)*1e3, h))"
done; done; done
