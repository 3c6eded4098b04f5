cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/prof_blob
for v in 2 3; do
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d gpurun_out/prof_blob/v$v -o pmc -- python3 tools/bench_legs.py --workload ensemble5 --members-per-gpu 16 --force-variant $v --steps 200 --warmup 20 --no-cpu-baseline --no-end-to-end > gpurun_out/prof_blob/v$v.log 2>&1
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/prof_blob/v$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    if "k_force" in k and len(c["SQ_WAVES"])>50:
        m={n:sum(v)/len(v) for n,v in c.items()}
        print("variant $v", k, "launches", len(c["SQ_WAVES"]), "VALU/wave %.0f"%(m["SQ_INSTS_VALU"]/m["SQ_WAVES"]), "lane util %.3f"%(m["SQ_THREAD_CYCLES_VALU"]/m["SQ_ACTIVE_INST_VALU"]/64), "waitcnt share %.2f"%(m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"]))
PY
done
