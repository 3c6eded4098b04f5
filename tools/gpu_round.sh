set -x
cd $GRAFT_REPO_ROOT
nproc; free -g | head -2
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python bench.py --steps 400 --warmup 100 > gpurun_out/r3_bench_a.json 2> gpurun_out/r3_bench_a.err; tail -c 600 gpurun_out/r3_bench_a.err
PB_PROFILE_LARGE=0 PB_PROFILE_VARIANT=3 bash tools/profile.sh r3_stream0 > gpurun_out/r3_stream0.log 2>&1; tail -30 gpurun_out/r3_stream0.log
