#!/usr/bin/env python3
"""bench.py against an EXPERIMENTAL library build (particlerobotsimulations_amd/lib_<tag>, made with
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
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
