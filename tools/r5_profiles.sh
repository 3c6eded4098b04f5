#!/bin/bash
# One gpurun call (round 5): rocprofv3 summaries of
#   r5_both_sums  the kernel that writes everything collideD writes (pbSimSetForceSums 1), 10^6-bot arena
#   r5_blob       BASELINE configs[4]'s blobs: 16 members of 10^5 bots, exact kernel (prewarm off: the scratch arena
#                 is a lattice and runs the same kernel name)
#   r5_blob_v3    the same with force variant 3
#   r5_ens4       BASELINE configs[3], 32 + 32 members per GPU
# Raw output under gpurun_out/prof_<tag>/; copy summary.md / trace stats into profiles/ afterwards.
cd $GRAFT_REPO_ROOT
COMMON="--no-cpu-baseline --no-survey-literal --no-streamlined --no-large-arena --no-clock --no-blob --no-ensemble-leg --no-both-sums --no-host-round-trip"
PB_PROFILE_LARGE=0 PB_TRAFFIC_KERNEL="1, 1, false, true>" bash tools/profile.sh r5_both_sums --force-sums 1 --steps 400 --warmup 100 $COMMON > gpurun_out/r5_both_sums.log 2>&1
PB_PROFILE_LARGE=0 bash tools/profile.sh r5_blob --workload ensemble5 --members-per-gpu 16 --steps 300 --warmup 200 --prewarm-ms 0 --no-cpu-baseline --no-end-to-end > gpurun_out/r5_blob.log 2>&1
PB_PROFILE_LARGE=0 PB_TRAFFIC_KERNEL="k_force_stream" bash tools/profile.sh r5_blob_v3 --workload ensemble5 --members-per-gpu 16 --force-variant 3 --steps 300 --warmup 200 --prewarm-ms 0 --no-cpu-baseline --no-end-to-end > gpurun_out/r5_blob_v3.log 2>&1
PB_PROFILE_LARGE=0 PB_TRAFFIC_KERNEL="k_" bash tools/profile.sh r5_ens4 --workload ensemble4 --steps 3000 --warmup 100 --no-cpu-baseline --no-end-to-end > gpurun_out/r5_ens4.log 2>&1
for t in r5_both_sums r5_blob r5_blob_v3 r5_ens4; do echo "== $t"; cat gpurun_out/prof_$t/status.txt; head -12 gpurun_out/prof_$t/summary.md; done
# gpurun copies back at most 64 MiB: drop the raw per-dispatch CSVs once they are summarised (the per-kernel stats
# table, the summaries, traffic.json and the logs stay)
find gpurun_out/prof_r5_* -name '*counter_collection.csv' -delete
find gpurun_out/prof_r5_* -name '*kernel_trace.csv' -delete
du -sh gpurun_out
