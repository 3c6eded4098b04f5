#!/bin/bash
# round-6 build: configs[3] (32 + 32 members per GPU) and configs[4]'s blobs under the exact kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
PB_PROFILE_LARGE=0 PB_TRAFFIC_KERNEL="k_" bash tools/profile.sh r6_ens4 --workload ensemble4 --steps 3000 --warmup 100 --no-cpu-baseline --no-end-to-end > gpurun_out/r6/prof_ens4.log 2>&1
PB_PROFILE_LARGE=0 bash tools/profile.sh r6_blob --workload ensemble5 --members-per-gpu 16 --steps 300 --warmup 200 --prewarm-ms 0 --no-cpu-baseline --no-end-to-end > gpurun_out/r6/prof_blob.log 2>&1
for t in r6_ens4 r6_blob; do echo "== $t"; cat gpurun_out/prof_$t/status.txt | tr '\n' ' '; echo; grep -E "^\| k_|lane utilisation|HBM-side|VALU instructions per wave|L2 hit|share of a wave" gpurun_out/prof_$t/summary.md | head -14 | cut -c1-230; python3 -c "
import json; d=json.load(open('gpurun_out/prof_$t/bench_unprofiled.json')); print('un-profiled us/step', round(d['ms_per_step']*1e3,2))"; done
find gpurun_out/prof_r6_* -name '*counter_collection.csv' -delete
find gpurun_out/prof_r6_* -name '*kernel_trace.csv' -delete
