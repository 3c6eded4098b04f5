cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soaks
free -g | head -2
timeout 1200 python tests/soak_huge_arena.py 1000000000 > gpurun_out/soaks/huge_arena_1e9_r3_v16.txt 2>&1; tail -4 gpurun_out/soaks/huge_arena_1e9_r3_v16.txt | cut -c1-400
timeout 600 python tests/soak_huge_arena.py 250000000 > gpurun_out/soaks/huge_arena_2p5e8_r3_v16.txt 2>&1; tail -2 gpurun_out/soaks/huge_arena_2p5e8_r3_v16.txt | cut -c1-400
