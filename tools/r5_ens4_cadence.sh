#!/bin/bash
# configs[3] on one GPU (32 + 32 members): launch durations and the gap between dependent launches, from the kernel trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/ens4_cadence; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 bench.py --workload ensemble4 --steps 3000 --warmup 100 --no-cpu-baseline --no-end-to-end > $OUT/trace.log 2>&1
python3 tools/launch_gap.py $OUT/trace | tee $OUT/cadence.txt
python3 bench.py --workload ensemble4 --steps 3000 --warmup 100 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('un-profiled: us per step (both batches, two streams)', d['ms_per_step']*1e3, 'long', d.get('ms_per_step_long',0)*1e3)" | tee -a $OUT/cadence.txt
rm -rf $OUT/trace
