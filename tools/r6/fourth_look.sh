#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests/test_gpu_stream_walk.py tests/test_gpu_streamlined.py tests/test_gpu_fma_bracket.py -m gpu -q 2>&1 | tail -8
python tools/ab_bench.py --variants 3w0,3w1,3 --bots 1000000 --rounds 4 --steps 300 --skip 300 --lattice blob 2>&1 | cut -c1-260 | tee gpurun_out/r6/ab_walk_blob.txt
python tools/ab_bench.py --variants 3w0,3w1,3 --bots 1000000 --rounds 4 --steps 300 --skip 300 2>&1 | cut -c1-260 | tee gpurun_out/r6/ab_walk_lattice.txt
