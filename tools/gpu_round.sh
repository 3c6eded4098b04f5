cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak
(timeout 1500 python tests/soak_fuzz.py 300 31 2>&1 | grep -v "^[0-9-]*\.[0-9]* [0-9-]*\.[0-9]* [0-9-]*\.[0-9]* $" | tail -4) > gpurun_out/soak/fuzz_r3.txt 2>&1 &
F=$!
(timeout 900 python tests/soak_long_run.py examples/example_obstacle.cfg 360000 60000 2>&1 | grep -v "^[0-9-]*\.[0-9]* [0-9-]*\.[0-9]* [0-9-]*\.[0-9]* $" | tail -3) > gpurun_out/soak/long_obstacle_r3.txt 2>&1
(timeout 900 python tests/soak_long_run.py examples/example_object_transport.cfg 300000 60000 1 2>&1 | grep -v "^[0-9-]*\.[0-9]* [0-9-]*\.[0-9]* [0-9-]*\.[0-9]* $" | tail -3) > gpurun_out/soak/long_transport_r3.txt 2>&1
wait $F
cat gpurun_out/soak/*.txt
PB_PROFILE_LARGE=1 bash tools/profile.sh r3_v14b > gpurun_out/r3_v14b.log 2>&1; tail -4 gpurun_out/r3_v14b.log
