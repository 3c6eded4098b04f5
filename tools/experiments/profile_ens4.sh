#!/bin/bash
# rocprofv3 of the BASELINE configs[3] ensemble leg (resident kernel): kernel trace + two PMC passes
OUT=gpurun_out/prof_ens4; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--workload ensemble4 --steps 3000 --warmup 100"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 tools/bench_legs.py $ARGS > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc_sq -- python3 tools/bench_legs.py $ARGS > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -o pmc_sq2 -- python3 tools/bench_legs.py $ARGS > $OUT/pmc2.log 2>&1
python3 tools/summarize_profile.py $OUT > $OUT/summary.md 2>$OUT/summarize.err
head -60 $OUT/summary.md
