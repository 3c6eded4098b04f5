cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) 2>&1
timeout 600 python bench.py > gpurun_out/arena_default.json 2> gpurun_out/arena_default.err; tail -c 600 gpurun_out/arena_default.json
PB_HOST_THREADS=32 timeout 1200 python bench.py --workload ensemble5 --members-per-gpu 1024 --steps 20 --warmup 5 > gpurun_out/r3_cfg5_full_pipeline.json 2> gpurun_out/r3_cfg5_full_pipeline.err; tail -c 1500 gpurun_out/r3_cfg5_full_pipeline.json
