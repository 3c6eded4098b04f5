#!/bin/bash
# A/B several library builds (particlerobotsimulations_amd/lib_<tag>, made with
#   make -C particlerobotsimulations_amd/csrc LIBDIR=../lib_<tag> BUILD=build_<tag> EXTRA_DEVFLAGS=... ../lib_<tag>/libparticlebot_hip.so ../lib_<tag>/libparticlebot_host.so):
# each in its own process (two copies of the library cannot coexist), the whole list twice.
#   bash tools/experiments/ab_libs.sh lib lib_w8 lib_p2 ...
for rep in 1 2; do
for lib in "$@"; do
  echo "== $lib (rep $rep)"
  timeout 300 python tools/ab_bench.py --libdir particlerobotsimulations_amd/$lib --variants 2 --bots 1000000 --rounds 4 --steps 300 --skip 300 2>&1 | tail -1 | cut -c1-120
done
done
