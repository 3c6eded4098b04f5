cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_streamlined.py -m gpu -x -q 2>&1 | grep -v "^[0-9-]*\.[0-9]* [0-9-]*\.[0-9]* [0-9-]*\.[0-9]* $" | tail -3
for rep in 1 2; do
for lib in lib lib_x1 lib_x2; do
  echo "== $lib"
  python tools/ab_bench.py --libdir particlerobotsimulations_amd/$lib --variants 2,3 --bots 1000000 --rounds 4 --steps 300 --skip 300 2>&1 | tail -2 | cut -c1-120
done; done
