#!/bin/bash
# Round-6 soaks on the final build (-> results/r6_soaks/): randomised differential runs of every form against the oracle,
# the headline workload in the headline (both-sums) form for its 2 500 steps, and the reference's obstacle example whole.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_soaks
timeout 1500 python tests/soak_fuzz.py 300 6 > gpurun_out/r6_soaks/fuzz_r6.txt 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r6_soaks/fuzz_r6.txt
timeout 900 python tests/soak_bench_parity.py 1000000 2500 1 > gpurun_out/r6_soaks/bench_parity_2500_steps_both_sums_r6.txt 2>&1; echo "bench parity rc=$?"; tail -2 gpurun_out/r6_soaks/bench_parity_2500_steps_both_sums_r6.txt
timeout 1500 python tests/soak_long_run.py examples/example_obstacle.cfg 1200000 200000 > gpurun_out/r6_soaks/long_obstacle_whole_r6.txt 2>&1; echo "long obstacle rc=$?"; tail -2 gpurun_out/r6_soaks/long_obstacle_whole_r6.txt
