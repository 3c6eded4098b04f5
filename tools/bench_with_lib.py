#!/usr/bin/env python3
"""tools/bench_legs.py against an EXPERIMENTAL library build (particlerobotsimulations_amd/lib_<tag>, made with
`make -C particlerobotsimulations_amd/csrc LIBDIR=../lib_<tag> BUILD=build_<tag> EXTRA_DEVFLAGS=... <targets>`), for A/B
runs on the GPU box:   python tools/bench_with_lib.py lib_w0 --workload ensemble5 ...   (never a reported number)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from particlerobotsimulations_amd import _capi  # noqa: E402

_capi.LIB_DIR = os.path.join(ROOT, "particlerobotsimulations_amd", sys.argv[1])
_capi.HIP_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_hip.so")
_capi.HOST_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_host.so")
sys.argv = [os.path.join(ROOT, "tools", "bench_legs.py")] + sys.argv[2:]
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_legs  # noqa: E402

bench_legs.main()
