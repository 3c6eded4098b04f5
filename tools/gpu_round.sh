cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { python bench.py --workload ensemble4 --members-per-gpu $1 --steps 2400 --warmup 20 --no-end-to-end --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('members/cfg', $1, 'us per batched step', d['ms_per_step']*1e3, 'long', d.get('ms_per_step_long',0)*1e3)"; }
echo "--- default build"; run 256; run 128; run 512
touch particlerobotsimulations_amd/csrc/pb_resident.hip
make -C particlerobotsimulations_amd/csrc EXTRA_DEVFLAGS=-DPB_RESIDENT_MIN_WAVES=8 all 2>&1 | grep -E "error|warning" | head -3
echo "--- 64 VGPRs (two workgroups per CU)"; run 256; run 128; run 512
