# scratch script for one gpurun call (`gpurun -- bash tools/gpu_round.sh`): the round's closing check
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/arena_default.json 2> gpurun_out/arena_default.err; tail -c 300 gpurun_out/arena_default.json
