#!/bin/bash
# one gpurun call: the bench contract test, the default line, the driver's --steps 20 form, and the headline kernel's
# profile on this round's build (-> profiles/r5_v18.*, latest_traffic.json)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python bench.py > gpurun_out/arena_default.json 2> gpurun_out/arena_default.err; python tools/show_bench.py gpurun_out/arena_default.json | head -40
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/arena_steps20.json 2> gpurun_out/arena_steps20.err; python tools/show_bench.py gpurun_out/arena_steps20.json | head -12
bash tools/profile.sh r5_v18 > gpurun_out/r5_v18.log 2>&1
find gpurun_out/prof_r5_v18* -name '*counter_collection.csv' -delete
find gpurun_out/prof_r5_v18* -name '*kernel_trace.csv' -delete
grep -E "^###|derived" gpurun_out/prof_r5_v18/summary.md | head -10
du -sh gpurun_out
