#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + three separate PMC passes of the
# arena workload of tools/bench_legs.py, raw output under gpurun_out/prof_<tag>/, then a distilled summary next to it.
# usage: tools/profile.sh <tag> [bench args...]      (PB_PROFILE_VARIANT=3: profile the streamlined kernel)
set -u
TAG=${1:-run}; shift || true
ARGS=${@:---force-variant ${PB_PROFILE_VARIANT:-2} --steps 400 --warmup 100 --no-cpu-baseline --no-survey-literal --no-streamlined --no-large-arena --no-clock --no-blob --no-ensemble-leg --no-both-sums --no-host-round-trip}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export PB_PROFILE_ARGS="$ARGS"   # summarize_profile.py quotes the command in traffic.json
export TMPDIR=/tmp
run() { # name, rocprof flags...
  local name=$1; shift
  timeout 600 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 tools/bench_legs.py $ARGS > "$OUT/$name.log" 2>&1
  echo "$name rc=$?" >> "$OUT/status.txt"
}
run trace --kernel-trace --stats
run pmc_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run pmc_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
run pmc_sq3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
# VALU issue-rate microbenchmark (the yardstick for the force kernel's instruction mix), un-profiled
[ -x tools/valu_rate ] || hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_rate.hip -o tools/valu_rate
tools/valu_rate > "$OUT/valu_rate.txt" 2>&1
# the same command un-profiled: kernel time at un-profiled clocks
python3 tools/bench_legs.py $ARGS 2>/dev/null | tail -1 > "$OUT/bench_unprofiled.json"
python3 tools/summarize_profile.py "$OUT" "$OUT/traffic.json" > "$OUT/summary.md" 2>"$OUT/summarize.err"
cat "$OUT/summary.md"
# SURVEY 8(d) caveat 2: the same kernel at 8 x 10^6 bots (544 MB of state, beyond the 256 MiB Infinity
# Cache): kernel trace + the two HBM-traffic passes, summarised next to the 10^6-bot profile
if [ "${PB_PROFILE_LARGE:-1}" = "1" ]; then
  OUT=gpurun_out/prof_${TAG}_8m
  mkdir -p "$OUT"
  ARGS="--bots 8000000 --steps 60 --warmup 20 --prewarm-ms 0 --no-cpu-baseline --no-survey-literal --no-streamlined --no-large-arena --no-clock --no-blob --no-ensemble-leg --no-both-sums --no-host-round-trip"
  run trace --kernel-trace --stats
  run pmc_fetch --pmc FETCH_SIZE
  run pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  python3 tools/bench_legs.py $ARGS 2>/dev/null | tail -1 > "$OUT/bench_unprofiled.json"
  python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.md" 2>"$OUT/summarize.err"
  cat "$OUT/summary.md"
fi
