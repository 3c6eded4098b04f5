#!/bin/bash
# one gpurun call: bench contract + the tests this round touched; the driver's form of bench.py (wall vs device gap after
# the polling wait); BASELINE configs[4] as a Cartesian sweep (64 fractions x 16 seeds = 1024 members of 10^5 bots) on ONE
# producer thread with each distinct blob placed once
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 1800 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_ensemble_pipeline.py tests/test_gpu_cli_resume.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -8
for i in 1 2 3; do
  s=$(date +%s%N); timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/bench_steps20_$i.json 2> gpurun_out/r6/bench_steps20_$i.err; e=$(date +%s%N)
  echo "bench.py --steps 20 run $i: rc=$? wall $(( (e - s) / 1000000 )) ms"; python tools/show_bench.py gpurun_out/r6/bench_steps20_$i.json | head -5 | cut -c1-400
done
PB_HOST_THREADS=2 timeout 1500 python3 tools/bench_legs.py --workload ensemble5 --members-total 1024 --steps 50 --no-cpu-baseline > gpurun_out/r6/cfg5_cartesian_one_producer.json 2> gpurun_out/r6/cfg5_cartesian_one_producer.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r6/cfg5_cartesian_one_producer.json").read().strip().splitlines()[-1])
e = d["end_to_end"]
print("configs[4] Cartesian, one producer:", "wall", round(e["wall_s"], 2), "bound", e["bound_rank0"], "pipeline", e["pipeline_rank0"])
PY
PB_HOST_THREADS=2 PB_SHARE_PLACEMENTS=0 timeout 1500 python3 tools/bench_legs.py --workload ensemble5 --members-total 128 --steps 50 --no-cpu-baseline > gpurun_out/r6/cfg5_128_unshared_one_producer.json 2> gpurun_out/r6/cfg5_128_unshared_one_producer.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r6/cfg5_128_unshared_one_producer.json").read().strip().splitlines()[-1])
e = d["end_to_end"]
print("configs[4] first 128 members, PB_SHARE_PLACEMENTS=0, one producer:", "wall", round(e["wall_s"], 2), "bound", e["bound_rank0"])
PY
du -sh gpurun_out
