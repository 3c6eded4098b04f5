cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak
(timeout 1500 python tests/soak_fuzz.py 300 47 2>&1 | grep -v "^[0-9-]*\.[0-9]* [0-9-]*\.[0-9]* [0-9-]*\.[0-9]* $" | tail -3) > gpurun_out/soak/fuzz_r3_final.txt 2>&1 &
F=$!
(timeout 900 python tests/soak_long_run.py examples/example.cfg 720000 120000 2>&1 | grep -v "^[0-9-]*\.[0-9]* [0-9-]*\.[0-9]* [0-9-]*\.[0-9]* $" | tail -2) > gpurun_out/soak/long_example_r3_final.txt 2>&1
(timeout 900 python tests/soak_long_run.py examples/example_dead_cells.cfg 360000 60000 2>&1 | grep -v "^[0-9-]*\.[0-9]* [0-9-]*\.[0-9]* [0-9-]*\.[0-9]* $" | tail -2) > gpurun_out/soak/long_dead_r3_final.txt 2>&1
(timeout 900 python tests/soak_long_run.py examples/example_gap.cfg 240000 60000 2 2>&1 | grep -v "^[0-9-]*\.[0-9]* [0-9-]*\.[0-9]* [0-9-]*\.[0-9]* $" | tail -2) > gpurun_out/soak/long_gap_r3_final.txt 2>&1
(timeout 600 python tests/soak_bench_parity.py 2>&1 | tail -3) > gpurun_out/soak/bench_parity_r3_final.txt 2>&1
wait $F
cat gpurun_out/soak/*_final.txt
