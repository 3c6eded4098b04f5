cd $GRAFT_REPO_ROOT
PB_PROFILE_LARGE=0 PB_TRAFFIC_KERNEL="k_" bash tools/profile.sh r5_ens4 --workload ensemble4 --steps 3000 --warmup 100 --no-cpu-baseline --no-end-to-end > gpurun_out/r5_ens4.log 2>&1
find gpurun_out/prof_r5_ens4 -name '*counter_collection.csv' -delete; find gpurun_out/prof_r5_ens4 -name '*kernel_trace.csv' -delete
grep -E "^###|derived: (share|VALU issue|L2)" gpurun_out/prof_r5_ens4/summary.md | head -12
timeout 900 python bench.py > gpurun_out/arena_default.json 2> gpurun_out/arena_default.err; python tools/show_bench.py gpurun_out/arena_default.json | grep -E "value|ensemble_leg|frac"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/arena_default.json").read().strip().splitlines()[-1]); e=d["ensemble_leg"]
print("ensemble_leg: us/step", e["ms_per_step"]*1e3, "e2e weak", e["end_to_end"]["wall_s"], "strong", e["strong_end_to_end"]["wall_s"])
PY
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/arena_steps20.json 2>/dev/null; python tools/show_bench.py gpurun_out/arena_steps20.json | head -2
timeout 600 python bench.py --workload ensemble4 --steps 12000 --warmup 100 > gpurun_out/ensemble4.json 2>/dev/null; python tools/show_bench.py gpurun_out/ensemble4.json | head -3
