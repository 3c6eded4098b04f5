cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for d in 14 10 12; do echo "--- 7 waves debug $d"; PB_PIPE_WAVES=7 PB_PIPE_DEBUG=$d timeout 300 python tools/pipelined_ab.py 1000000 400 --time-only 2>&1 | tail -3 | cut -c1-330; done
