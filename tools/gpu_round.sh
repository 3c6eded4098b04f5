cd $GRAFT_REPO_ROOT
PB_PROFILE_LARGE=1 bash tools/profile.sh r3_v15 > gpurun_out/r3_v15.log 2>&1; tail -3 gpurun_out/r3_v15.log
(time python bench.py --steps 20 --warmup 5) > gpurun_out/r3_bench_d20.json 2> gpurun_out/r3_bench_d20.err
(time python bench.py) > gpurun_out/r3_bench_d.json 2> gpurun_out/r3_bench_d.err
tail -3 gpurun_out/r3_bench_d.err
python tools/lanes_sweep.py > gpurun_out/r3_lanes_sweep.txt 2>&1; tail -12 gpurun_out/r3_lanes_sweep.txt
