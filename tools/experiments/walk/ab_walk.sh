#!/bin/bash
# Round 5 A/B of the neighbour-walk forms (pbSimSetWalk) on one MI355X: the bench lattice and a 10^6-bot random blob in
# one process each (interleaved rounds, bit-identity checked), then BASELINE configs[4]'s own blobs (16 and 64 members
# of 10^5 bots through bench.py, one process per form, the list twice).
#   bash tools/ab_walk.sh "3,3w1,3w2" 3        (variants for ab_bench.py; force variant of the ensemble runs)
cd $GRAFT_REPO_ROOT
V=${1:-3,3w1,3w2}; FV=${2:-3}
python tools/ab_bench.py --variants $V --bots 1000000 --rounds 4 --steps 300 --skip 300 2>&1 | sed -E 's/=> .*(bit-identical)/\1/' | cut -c1-220
python tools/ab_bench.py --variants $V --bots 1000000 --rounds 4 --steps 300 --skip 300 --lattice blob 2>&1 | sed -E 's/=> .*(bit-identical)/\1/' | cut -c1-220
for rep in 1 2; do
for w in 0 1 2; do
  PB_ALLOW_ENV_OVERRIDES=1 PB_WALK=$w python bench.py --workload ensemble5 --members-per-gpu 16 --force-variant $FV --steps 300 --warmup 200 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ensemble5 x16 walk $w variant $FV: us/step %.2f (long %.2f)' % (d['ms_per_step']*1e3, d.get('ms_per_step_long',0)*1e3))"
done
done
for w in 0 1 2; do
  PB_ALLOW_ENV_OVERRIDES=1 PB_WALK=$w python bench.py --workload ensemble5 --members-per-gpu 64 --force-variant $FV --steps 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d['end_to_end']; b=e['bound_rank0'][0]
print('ensemble5 x64 end to end walk $w variant $FV: wall %.2f s device_s %.3f host_s %.2f bound %s' % (e['wall_s'], b['device_s'], b['host_s'], b['bound']))"
done
