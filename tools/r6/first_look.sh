#!/bin/bash
# one gpurun call: the split bench's contract tests, the driver's form of bench.py (timed from outside), the default
# line, and this round's profiles of the two headline kernels (-> profiles/r6_*, latest_traffic*.json)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q 2>&1 | tail -8
t0=$(date +%s.%N)
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/bench_steps20.json 2> gpurun_out/r6/bench_steps20.err
t1=$(date +%s.%N); echo "bench.py --steps 20: rc=$? wall $(echo "$t1 - $t0" | bc) s"
python tools/show_bench.py gpurun_out/r6/bench_steps20.json
cp bench_detail.json gpurun_out/r6/bench_steps20.detail.json
timeout 900 python3 bench.py > gpurun_out/r6/bench_default.json 2> gpurun_out/r6/bench_default.err; python tools/show_bench.py gpurun_out/r6/bench_default.json
cp bench_detail.json gpurun_out/r6/bench_default.detail.json
COMMON="--no-cpu-baseline --no-survey-literal --no-streamlined --no-large-arena --no-clock --no-blob --no-ensemble-leg --no-both-sums --no-host-round-trip"
PB_PROFILE_LARGE=0 PB_TRAFFIC_KERNEL="1, 1, false, true>" bash tools/profile.sh r6_both_sums --force-sums 1 --steps 400 --warmup 100 $COMMON > gpurun_out/r6/prof_both_sums.log 2>&1
bash tools/profile.sh r6_v19 > gpurun_out/r6/prof_v19.log 2>&1
for t in r6_both_sums r6_v19; do echo "== $t"; cat gpurun_out/prof_$t/status.txt; grep -E "^\| k_force|derived" gpurun_out/prof_$t/summary.md | head -12; done
find gpurun_out/prof_r6_* -name '*counter_collection.csv' -delete
find gpurun_out/prof_r6_* -name '*kernel_trace.csv' -delete
du -sh gpurun_out
