#!/bin/bash
# runtime knobs that touch the dependent-launch boundary of the per-step forms (BASELINE configs[3], 32 + 32 members):
# kernel arguments in device memory (HIP_FORCE_DEV_KERNARG), and the launch cadence microbenchmark beside it
cd $GRAFT_REPO_ROOT
for kv in "X=0" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "ROC_ACTIVE_WAIT_TIMEOUT=100" "X=0"; do
  env $kv python tools/bench_legs.py --workload ensemble4 --steps 12000 --warmup 200 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$kv: us/step %.3f' % (d['ms_per_step']*1e3))"
done
