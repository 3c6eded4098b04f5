cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soaks
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 3000 python tests/soak_fuzz.py 1000 61 > gpurun_out/soaks/fuzz_r3_v16_1000.txt 2>&1; tail -2 gpurun_out/soaks/fuzz_r3_v16_1000.txt | cut -c1-300
