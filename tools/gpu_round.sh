set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ensemble_pipeline.py tests/test_gpu_bench_contract.py -m gpu -x -q -s 2>&1 | grep -v "^[0-9-]*\.[0-9]* [0-9-]*\.[0-9]* [0-9-]*\.[0-9]* $" | tail -30
(time python bench.py --steps 20 --warmup 5) > gpurun_out/r3_bench_b20.json 2> gpurun_out/r3_bench_b20.err; tail -c 400 gpurun_out/r3_bench_b20.err
(time python bench.py) > gpurun_out/r3_bench_b.json 2> gpurun_out/r3_bench_b.err; tail -c 400 gpurun_out/r3_bench_b.err
