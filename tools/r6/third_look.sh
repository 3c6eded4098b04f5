#!/bin/bash
# one gpurun call: the tests round 6 added or touched; where the host's time around bench.py's 20-step region goes;
# the flattened walk of k_force_stream A/B (lattice, 10^6-bot blob, configs[4] slice) and its blob profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests/test_gpu_stream_walk.py tests/test_gpu_streamlined.py tests/test_gpu_fma_bracket.py tests/test_gpu_bench_contract.py tests/test_gpu_ensemble_pipeline.py tests/test_gpu_cli_resume.py -m gpu -q 2>&1 | tail -15
PB_TIMED_TRACE=1 timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ensemble > gpurun_out/r6/bench_trace.json 2> gpurun_out/r6/bench_trace.err
grep "pbSimStepTimed: 20 steps" gpurun_out/r6/bench_trace.err | head -3; python tools/show_bench.py gpurun_out/r6/bench_trace.json | head -4 | cut -c1-300
export PB_ALLOW_ENV_OVERRIDES=1
for w in 0 1 -1; do
  echo "== PB_STREAM_WALK=$w"
  PB_STREAM_WALK=$w timeout 600 python3 tools/bench_legs.py --steps 600 --warmup 100 --no-cpu-baseline --no-survey-literal --no-large-arena --no-clock --no-ensemble-leg --no-both-sums --no-host-round-trip 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  lattice v3 us/step', round(d['streamlined']['us_per_step'],2), '| exact', round(d['ms_per_step']*1e3,2), '| blob exact', round(d['random_blob']['us_per_step'],2))"
  PB_STREAM_WALK=$w PB_FORCE_VARIANT=3 timeout 600 python3 tools/bench_legs.py --steps 600 --warmup 100 --no-cpu-baseline --no-survey-literal --no-large-arena --no-clock --no-ensemble-leg --no-both-sums --no-host-round-trip --no-streamlined 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  10^6-bot blob v3 us/step', round(d['random_blob']['us_per_step'],2))"
  PB_STREAM_WALK=$w timeout 900 python3 tools/bench_legs.py --workload ensemble5 --members-total 64 --force-variant 3 --steps 300 --warmup 100 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d['end_to_end']
print('  configs[4] 64-member slice v3: steady us/step', round(d['ms_per_step']*1e3,2), '| end to end device_s', round(e['pipeline_rank0'][0]['device_s'],3), 'wall', round(e['wall_s'],3), 'sub_batch', e['pipeline_rank0'][0]['sub_batch'])"
done
unset PB_ALLOW_ENV_OVERRIDES
PB_PROFILE_LARGE=0 PB_TRAFFIC_KERNEL="k_force_stream" bash tools/profile.sh r6_blob_v3 --workload ensemble5 --members-per-gpu 16 --force-variant 3 --steps 300 --warmup 200 --prewarm-ms 0 --no-cpu-baseline --no-end-to-end > gpurun_out/r6/prof_blob_v3.log 2>&1
echo "== r6_blob_v3"; cat gpurun_out/prof_r6_blob_v3/status.txt; grep -E "^\| k_force|derived" gpurun_out/prof_r6_blob_v3/summary.md | head -12
find gpurun_out/prof_r6_* -name '*counter_collection.csv' -delete
find gpurun_out/prof_r6_* -name '*kernel_trace.csv' -delete
du -sh gpurun_out
