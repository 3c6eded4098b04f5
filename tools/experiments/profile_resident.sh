#!/bin/bash
# rocprofv3 of the resident kernel on a full-chip ensemble (256 members of example_obstacle.cfg, 500 bots each)
OUT=gpurun_out/prof_resident; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="-m particlerobotsimulations_amd.ensemble examples/example_obstacle.cfg --members 256 --set max_time 100"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc_sq -- python3 $ARGS > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -o pmc_sq2 -- python3 $ARGS > $OUT/pmc2.log 2>&1
python3 tools/summarize_profile.py $OUT > $OUT/summary.md 2>$OUT/summarize.err
grep -n "k_resident" $OUT/summary.md | head -3
sed -n '/### k_resident/,/^$/p' $OUT/summary.md | head -40
