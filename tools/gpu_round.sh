# scratch script for one gpurun call (`gpurun -- bash tools/gpu_round.sh`): the round's closing check
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/arena_default.json 2> gpurun_out/arena_default.err; tail -c 300 gpurun_out/arena_default.json
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/arena_steps20.json 2> gpurun_out/arena_steps20.err; python tools/show_bench.py gpurun_out/arena_steps20.json | head -5
# the ensemble layer on a starved host: one producer thread, the reference's placement rule -> bound: host, and the
# same sweep with the O(N) generator next to it
PB_HOST_THREADS=2 timeout 900 python bench.py --workload ensemble5 --members-per-gpu 32 --steps 50 --no-cpu-baseline > gpurun_out/ensemble5_host_bound.json 2> gpurun_out/ensemble5_host_bound.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/ensemble5_host_bound.json").read().strip().splitlines()[-1])
e = d["end_to_end"]
print("ensemble5, PB_HOST_THREADS=2:", d.get("end_to_end_bound"), "wall", round(e["wall_s"], 2), "bound", e["bound_rank0"][0]["bound"],
      "host_s", round(e["bound_rank0"][0]["host_s"], 2), "device_s", round(e["bound_rank0"][0]["device_s"], 2))
f = d.get("end_to_end_fastblob")
if f: print("  fastblob:", "wall", round(f["wall_s"], 2), f["bound_rank0"][0]["bound"])
print("host:", d["host"]["host_threads_rule"])
PY
