#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in 8 32; do for rep in 1 2 3; do for lib in lib lib_args; do
  python tools/bench_with_lib.py $lib --workload ensemble4 --members-per-gpu $m --steps 12000 --warmup 200 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib members $m: us/step %.3f' % (d['ms_per_step']*1e3))"
done; done; done
