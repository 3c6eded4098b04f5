#!/bin/bash
# After the last kernel-source change of the round: the tests touched since the full GPU run, the bench lines and the
# profiles DESIGN/README quote, on the build whose stamp the profiles then carry.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
cat particlerobotsimulations_amd/lib/build_stamp.json
timeout 2400 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_ensemble_pipeline.py tests/test_gpu_full_size.py tests/test_gpu_stream_walk.py -m gpu -q -x 2>&1 | tail -4
for i in 1 2; do
  s=$(date +%s%N); timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/bench_steps20_$i.json 2> gpurun_out/r6/bench_steps20_$i.err; e=$(date +%s%N)
  echo "bench.py --gpus 1 --steps 20 --warmup 5 (run $i): rc=$? wall $(( (e - s) / 1000000 )) ms"; python tools/show_bench.py gpurun_out/r6/bench_steps20_$i.json | sed -n '1p;3,4p' | cut -c1-420
done
cp bench_detail.json gpurun_out/r6/bench_steps20.detail.json
timeout 900 python3 bench.py > gpurun_out/r6/bench_default.json 2> gpurun_out/r6/bench_default.err; python tools/show_bench.py gpurun_out/r6/bench_default.json | cut -c1-420
cp bench_detail.json gpurun_out/r6/bench_default.detail.json
COMMON="--no-cpu-baseline --no-survey-literal --no-streamlined --no-large-arena --no-clock --no-blob --no-ensemble-leg --no-both-sums --no-host-round-trip"
PB_PROFILE_LARGE=0 PB_TRAFFIC_KERNEL="1, 1, false, true>" bash tools/profile.sh r6_both_sums --force-sums 1 --steps 400 --warmup 100 $COMMON > gpurun_out/r6/prof_both_sums.log 2>&1
bash tools/profile.sh r6_v19 > gpurun_out/r6/prof_v19.log 2>&1
PB_PROFILE_LARGE=0 PB_PROFILE_VARIANT=3 PB_TRAFFIC_KERNEL="k_force_stream" bash tools/profile.sh r6_stream > gpurun_out/r6/prof_stream.log 2>&1
PB_PROFILE_LARGE=0 PB_TRAFFIC_KERNEL="k_force_stream" bash tools/profile.sh r6_blob_v3 --workload ensemble5 --members-per-gpu 16 --force-variant 3 --steps 300 --warmup 200 --prewarm-ms 0 --no-cpu-baseline --no-end-to-end > gpurun_out/r6/prof_blob_v3.log 2>&1
for t in r6_both_sums r6_v19 r6_stream r6_blob_v3; do echo "== $t"; cat gpurun_out/prof_$t/status.txt | tr '\n' ' '; echo; grep -E "^\| k_force|lane utilisation|HBM-side|VALU instructions per wave" gpurun_out/prof_$t/summary.md | head -5 | cut -c1-200; done
find gpurun_out/prof_r6_* -name '*counter_collection.csv' -delete
find gpurun_out/prof_r6_* -name '*kernel_trace.csv' -delete
# after the profiles exist the line quotes them: the same two commands again
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/bench_steps20_after.json 2>/dev/null; python tools/show_bench.py gpurun_out/r6/bench_steps20_after.json | sed -n '1p;3,4p' | cut -c1-420
