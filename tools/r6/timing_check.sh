#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for i in 1 2; do
s=$(date +%s%N); timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/timing_$i.json 2> gpurun_out/r6/timing_$i.err; e=$(date +%s%N)
echo "bench.py --gpus 1 --steps 20 --warmup 5: rc=$? wall $(( (e - s) / 1000000 )) ms"; python3 -c "
import json; d=json.load(open('bench_detail.json')); print(d['phases_end_s']); l=d['line']; print(l['value'], l['roofline']['frac'], l['roofline']['profile'], l['cpu_baseline']['sample'])"
done
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -m gpu -q -x 2>&1 | tail -2
