#!/bin/bash
# rocprofv3 counters of a neighbour-walk form on BASELINE configs[4]'s blobs: tools/r5_walk_prof.sh <tag> <walk> <variant>
cd $GRAFT_REPO_ROOT
TAG=$1; W=$2; FV=$3
export PB_ALLOW_ENV_OVERRIDES=1 PB_WALK=$W PB_PROFILE_LARGE=0
[ "$FV" = "3" ] && export PB_TRAFFIC_KERNEL="k_force_stream"
bash tools/profile.sh $TAG --workload ensemble5 --members-per-gpu 16 --force-variant $FV --steps 300 --warmup 200 --prewarm-ms 0 --no-cpu-baseline --no-end-to-end > gpurun_out/$TAG.log 2>&1
find gpurun_out/prof_$TAG -name '*counter_collection.csv' -delete
find gpurun_out/prof_$TAG -name '*kernel_trace.csv' -delete
grep -E "^###|derived" gpurun_out/prof_$TAG/summary.md | head -12
