#!/bin/bash
# the round's closing check: what the driver runs at round end, on the final tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
cat particlerobotsimulations_amd/lib/build_stamp.json
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r6/pytest_gpu_closing.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r6/pytest_gpu_closing.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
s=$(date +%s%N); timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/closing_steps20.json 2> gpurun_out/r6/closing_steps20.err; e=$(date +%s%N)
echo "bench.py --gpus 1 --steps 20 --warmup 5: rc=$? wall $(( (e - s) / 1000000 )) ms"; python tools/show_bench.py gpurun_out/r6/closing_steps20.json | cut -c1-420
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --force-dist --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force-dist collective', d['collective'])"
