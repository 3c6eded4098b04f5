#!/usr/bin/env python3
"""Per-workgroup timeline of one k_force launch (diagnostic build -DPB_TIMELINE only):
  make -C particlerobotsimulations_amd/csrc LIBDIR=../lib_timeline BUILD=build_timeline EXTRA_DEVFLAGS=-DPB_TIMELINE ../lib_timeline/libparticlebot_hip.so
  python tools/timeline.py --libdir particlerobotsimulations_amd/lib_timeline [--bots 1000000]
Prints when workgroups start and end (100 MHz real-time counter), how many are resident over time, how
long they live by start time, and how they spread over XCDs / CUs."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libdir", required=True)
    ap.add_argument("--bots", type=int, default=1_000_000)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from particlerobotsimulations_amd import _capi
    _capi.LIB_DIR = os.path.abspath(args.libdir)
    _capi.HIP_SO = os.path.join(_capi.LIB_DIR, "libparticlebot_hip.so")
    import particlerobotsimulations_amd as pb
    pb.legacy.cudaInit(0, None)
    n = args.bots
    sim = bench.make_sim(pb, n, bench.LATTICE_PITCH, seed=1)
    sim.set_lanes_per_bot(1)
    sim.step(200)
    sim.synchronize()  # (the pointer below must not change under running kernels)
    tiles = (n + 255) // 256
    grid = ((tiles + 7) // 8) * 8 if tiles >= 64 else tiles
    buf = pb.DeviceArray((grid + 64, 8), np.uint64, fill=0)  # (+64 rows of slack)
    L = _capi.lib()
    L.pbDebugSetTimeline.argtypes = [C.c_void_p]
    assert L.pbDebugSetTimeline(buf.ptr) == 0
    sim.step(5)
    sim.synchronize()
    L.pbDebugSetTimeline(None)
    t = buf.download()
    print("rows written beyond the grid:", int((t[grid:] != 0).any(axis=1).sum()))
    t = t[:grid]
    t = t[t[:, 1] > 0]
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) * 0.01  # us
    end = (t[:, 1] - t0) * 0.01
    life = end - start
    xcc = t[:, 2].astype(int)
    hw = t[:, 3].astype(np.uint64)
    cu = ((hw >> 8) & 0xF).astype(int)
    sh = ((hw >> 12) & 0x1).astype(int)
    se = ((hw >> 13) & 0x7).astype(int)
    cuid = xcc * 64 + se * 16 + sh * 8 + cu % 16
    print(f"{len(t)} workgroups; launch spans {end.max():.1f} us; starts: min {start.min():.1f} p50 {np.median(start):.1f} max {start.max():.1f}")
    print(f"workgroup lifetime: min {life.min():.1f} p10 {np.percentile(life, 10):.1f} p50 {np.median(life):.1f} "
          f"p90 {np.percentile(life, 90):.1f} max {life.max():.1f} us; sum {life.sum():.0f} us = {life.sum() / end.max():.0f} resident on average")
    print("time(us)  resident  started  finished")
    for a in np.arange(0, end.max() + 5, 5.0):
        print(f"{a:7.0f} {int(((start <= a) & (end > a)).sum()):9d} {int((start <= a).sum()):8d} {int((end <= a).sum()):9d}")
    order = np.argsort(start)
    print("lifetime by start-time decile (us):", [round(float(life[order[i::10]].mean()), 1) for i in range(10)][:1],
          [round(float(np.mean(life[(start >= lo) & (start < lo + 10)])), 1) if ((start >= lo) & (start < lo + 10)).any() else None
           for lo in range(0, int(end.max()), 10)])
    first_pair = (t[:, 4] - t0) * 0.01 - start   # start -> first neighbour pair (the dependent load chain)
    half = (t[:, 5] - t0) * 0.01 - start         # start -> half way through the stencil
    r1 = start < 2.0
    r2 = start > 20.0
    for name, m in (("first round (started < 2 us)", r1), ("later rounds (started > 20 us)", r2)):
        if m.any():
            print(f"{name}: start->first pair p10/p50/p90 {np.percentile(first_pair[m], [10, 50, 90]).round(1).tolist()} us; "
                  f"start->half way {np.percentile(half[m], [10, 50, 90]).round(1).tolist()}; "
                  f"lifetime {np.percentile(life[m], [10, 50, 90]).round(1).tolist()}")
    print("workgroups per XCD:", np.bincount(xcc, minlength=8).tolist())
    per_cu = np.bincount(cuid)
    per_cu = per_cu[per_cu > 0]
    print(f"distinct CUs seen {len(per_cu)}; workgroups per CU min {per_cu.min()} p50 {int(np.median(per_cu))} max {per_cu.max()}")
    first = start < 2.0
    fc = np.bincount(cuid[first])
    fc = fc[fc > 0]
    print(f"workgroups started in the first 2 us: {int(first.sum())} on {len(fc)} CUs, per CU min {fc.min()} max {fc.max()}")
    if args.out:
        np.save(args.out, t)


if __name__ == "__main__":
    main()
