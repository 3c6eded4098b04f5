#!/bin/bash
# The round's closing gpurun call on the final build: the whole GPU suite + smoke, the bench lines that DESIGN/README quote
# (-> results/r6_bench_lines/), the profiles of the headline kernels and of the tolerance kernel (-> profiles/r6_*), and
# BASELINE configs[4] whole as a Cartesian sweep on one producer thread.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r6/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r6/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for i in 1 2; do
  s=$(date +%s%N); timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/bench_steps20_$i.json 2> gpurun_out/r6/bench_steps20_$i.err; e=$(date +%s%N)
  echo "bench.py --gpus 1 --steps 20 --warmup 5 (run $i): rc=$? wall $(( (e - s) / 1000000 )) ms"; python tools/show_bench.py gpurun_out/r6/bench_steps20_$i.json | head -4 | cut -c1-420
done
cp bench_detail.json gpurun_out/r6/bench_steps20.detail.json
timeout 900 python3 bench.py > gpurun_out/r6/bench_default.json 2> gpurun_out/r6/bench_default.err; python tools/show_bench.py gpurun_out/r6/bench_default.json | cut -c1-420
cp bench_detail.json gpurun_out/r6/bench_default.detail.json
timeout 900 python3 tools/bench_legs.py > gpurun_out/r6/legs_default.json 2> gpurun_out/r6/legs_default.err; python tools/show_bench.py gpurun_out/r6/legs_default.json | cut -c1-420
COMMON="--no-cpu-baseline --no-survey-literal --no-streamlined --no-large-arena --no-clock --no-blob --no-ensemble-leg --no-both-sums --no-host-round-trip"
PB_PROFILE_LARGE=0 PB_TRAFFIC_KERNEL="1, 1, false, true>" bash tools/profile.sh r6_both_sums --force-sums 1 --steps 400 --warmup 100 $COMMON > gpurun_out/r6/prof_both_sums.log 2>&1
bash tools/profile.sh r6_v19 > gpurun_out/r6/prof_v19.log 2>&1
PB_PROFILE_LARGE=0 PB_PROFILE_VARIANT=3 PB_TRAFFIC_KERNEL="k_force_stream" bash tools/profile.sh r6_stream > gpurun_out/r6/prof_stream.log 2>&1
PB_PROFILE_LARGE=0 PB_TRAFFIC_KERNEL="k_force_stream" bash tools/profile.sh r6_blob_v3 --workload ensemble5 --members-per-gpu 16 --force-variant 3 --steps 300 --warmup 200 --prewarm-ms 0 --no-cpu-baseline --no-end-to-end > gpurun_out/r6/prof_blob_v3.log 2>&1
for t in r6_both_sums r6_v19 r6_stream r6_blob_v3; do echo "== $t"; cat gpurun_out/prof_$t/status.txt | tr '\n' ' '; echo; grep -E "^\| k_force|lane utilisation|HBM-side|VALU instructions per wave" gpurun_out/prof_$t/summary.md | head -6 | cut -c1-260; done
find gpurun_out/prof_r6_* -name '*counter_collection.csv' -delete
find gpurun_out/prof_r6_* -name '*kernel_trace.csv' -delete
PB_HOST_THREADS=2 timeout 1500 python3 tools/bench_legs.py --workload ensemble5 --members-total 1024 --steps 50 --no-cpu-baseline > gpurun_out/r6/cfg5_cartesian_one_producer.json 2> gpurun_out/r6/cfg5_cartesian_one_producer.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r6/cfg5_cartesian_one_producer.json").read().strip().splitlines()[-1])
e = d["end_to_end"]
b, p = e["bound_rank0"][0], e["pipeline_rank0"][0]
print("configs[4] Cartesian, one producer: wall", round(e["wall_s"], 2), "bound", b["bound"], "host_s", round(b["host_s"], 2), "device_s", round(b["device_s"], 2),
      "placements", p["placements_run"], "shared", p["placements_shared"], "sub_batch", p["sub_batch"], "lanes", p["lanes"], "waited", round(p["placement_wait_s"], 2))
PY
du -sh gpurun_out
