cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in lib_base lib; do
  echo "== $lib"
  python tools/ab_bench.py --libdir particlerobotsimulations_amd/$lib --variants 2,3 --bots 1000000 --rounds 4 --steps 300 --skip 300 2>&1 | tail -2 | cut -c1-130
done; done
python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^[0-9-]*\.[0-9]* [0-9-]*\.[0-9]* [0-9-]*\.[0-9]* $" | tail -4
