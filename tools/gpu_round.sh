cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soaks
timeout 300 python -m pytest tests/test_gpu_ensemble_pipeline.py -x -q 2>&1 | tail -2
( timeout 2400 python tests/soak_fuzz.py 300 53 > gpurun_out/soaks/fuzz_r3_v16.txt 2>&1; tail -2 gpurun_out/soaks/fuzz_r3_v16.txt ) &
timeout 900 python tests/soak_long_run.py examples/example.cfg 720000 120000 0 > gpurun_out/soaks/long_example_r3_v16.txt 2>&1; tail -1 gpurun_out/soaks/long_example_r3_v16.txt
timeout 900 python tests/soak_long_run.py examples/example_object_transport.cfg 300000 60000 1 > gpurun_out/soaks/long_transport_r3_v16.txt 2>&1; tail -1 gpurun_out/soaks/long_transport_r3_v16.txt
timeout 900 python tests/soak_long_run.py examples/example_obstacle.cfg 240000 60000 2 > gpurun_out/soaks/long_obstacle_r3_v16.txt 2>&1; tail -1 gpurun_out/soaks/long_obstacle_r3_v16.txt
timeout 900 python tests/soak_bench_parity.py > gpurun_out/soaks/bench_parity_r3_v16.txt 2>&1; tail -1 gpurun_out/soaks/bench_parity_r3_v16.txt
wait
