cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soaks
timeout 900 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_dead_sum.py tests/test_gpu_forms.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-survey-literal --no-streamlined --no-large-arena --no-blob --no-ensemble-leg --no-both-sums 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline us/step', d['ms_per_step']*1e3)"
timeout 600 python tests/soak_huge_arena.py 300000000 > gpurun_out/soaks/huge_arena_3e8_windowed.txt 2>&1; tail -1 gpurun_out/soaks/huge_arena_3e8_windowed.txt | cut -c1-300
timeout 1200 python tests/soak_huge_arena.py 1000000000 > gpurun_out/soaks/huge_arena_1e9_windowed.txt 2>&1; tail -1 gpurun_out/soaks/huge_arena_1e9_windowed.txt | cut -c1-300
