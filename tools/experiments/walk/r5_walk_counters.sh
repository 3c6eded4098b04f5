#!/bin/bash
# Round 5, VERDICT r4 item 2: counters of every neighbour-walk form (pbSimSetWalk 0/1/2) of the exact kernel and of
# force variant 3 on BASELINE configs[4]'s blobs (16 members of 10^5 bots): launch time, VALU per wave, lane
# utilisation, s_waitcnt share, vector-memory instructions and L1 (TCP) cache-line accesses per wave.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PB_ALLOW_ENV_OVERRIDES=1
OUT=gpurun_out/walk_counters; mkdir -p $OUT
for fv in 2 3; do for w in 0 1 2; do
  export PB_WALK=$w
  ARGS="--workload ensemble5 --members-per-gpu 16 --force-variant $fv --steps 300 --warmup 200 --prewarm-ms 0 --no-cpu-baseline --no-end-to-end"
  tag=v${fv}_w$w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag/trace -o trace -- python3 bench.py $ARGS > $OUT/$tag.trace.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/$tag/pmc1 -o pmc1 -- python3 bench.py $ARGS > $OUT/$tag.pmc1.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $OUT/$tag/pmc2 -o pmc2 -- python3 bench.py $ARGS > $OUT/$tag.pmc2.log 2>&1
  timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_WAVES --output-format csv -d $OUT/$tag/pmc3 -o pmc3 -- python3 bench.py $ARGS > $OUT/$tag.pmc3.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/$tag/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/$tag/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, c in agg.items():
    if "k_force" not in k or len(c["SQ_WAVES"]) < 50: continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    w = m["SQ_WAVES"]
    d = dur.get(k, [0])
    g = lambda n: m.get(n, float("nan"))
    print("variant $fv walk $w: %.1f us/launch | VALU/wave %.0f trans/wave %.0f lane util %.3f | wave cycles: issuing %.2f stalled %.2f waitcnt %.2f | "
          "VMEM_RD/wave %.1f LDS/wave %.1f | TCP line accesses per VMEM_RD %.1f, TCC read req per VMEM_RD %.1f" % (
          sum(d) / len(d) / 1e3, g("SQ_INSTS_VALU") / w, g("SQ_INSTS_VALU_TRANS_F32") / w,
          g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU") / 64, g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
          g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_INSTS_VMEM_RD") / w,
          g("SQ_INSTS_LDS") / w, g("TCP_TOTAL_CACHE_ACCESSES_sum") / g("SQ_INSTS_VMEM_RD"),
          g("TCP_TCC_READ_REQ_sum") / g("SQ_INSTS_VMEM_RD")))
PY
  tail -2 $OUT/$tag.pmc3.log | cut -c1-200 | grep -i "error\|invalid" 
  rm -rf $OUT/$tag
done; done
