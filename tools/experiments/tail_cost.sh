cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/tail; mkdir -p $OUT
ARGS="--bots 200000 --pitch 1.0 --steps 400 --warmup 50 --prewarm-ms 0 --no-cpu-baseline --no-survey-literal --no-streamlined --no-large-arena --no-clock --no-blob --no-ensemble-leg --no-both-sums --no-host-round-trip"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc -o pmc -- python3 tools/bench_legs.py $ARGS > $OUT/log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/tail/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "k_force" in k and len(c["SQ_WAVES"]) > 50:
        m = {n: sum(v) / len(v) for n, v in c.items()}
        print(k, "launches", len(c["SQ_WAVES"]), {n: round(m[n] / m["SQ_WAVES"], 1) for n in m if n != "SQ_WAVES"})
PY
rm -rf $OUT/pmc
