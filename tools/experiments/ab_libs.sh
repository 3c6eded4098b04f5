#!/bin/bash
# A/B several library builds: each in its own process (libraries cannot coexist), interleaved twice
mkdir -p gpurun_out/r2s
for rep in 1 2; do
for lib in lib lib_w8 lib_p2 lib_p2w8; do
  echo "== $lib (rep $rep)"
  timeout 300 python tools/ab_bench.py --libdir particlerobotsimulations_amd/$lib --variants 2 --bots 1000000 --rounds 4 --steps 300 --skip 300 2>&1 | tail -1 | cut -c1-120
done
done
