cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ensemble_pipeline.py tests/test_gpu_bench_contract.py -x -q 2>&1 | tail -3
PB_HOST_THREADS=32 timeout 900 python bench.py --workload ensemble5 --members-per-gpu 256 --steps 20 --warmup 5 > gpurun_out/r3_cfg5_quarter_pipeline_auto.json 2> gpurun_out/r3_cfg5_quarter_pipeline_auto.err
PB_HOST_THREADS=32 timeout 1200 python bench.py --workload ensemble5 --members-per-gpu 1024 --steps 20 --warmup 5 > gpurun_out/r3_cfg5_full_pipeline_auto.json 2> gpurun_out/r3_cfg5_full_pipeline_auto.err
python - <<PY
import json
for f in ('quarter','full'):
    d=json.load(open(f'gpurun_out/r3_cfg5_{f}_pipeline_auto.json'))
    e=d['end_to_end']; print(f, 'wall', e['wall_s'], e['value_end_to_end'], e['pipeline_rank0'], e['last_rows_time_comx_comy_dist'][0][:2])
PY
