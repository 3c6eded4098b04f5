#!/usr/bin/env python3
"""How often does a 64-lane wave of the throughput force kernel run its contact block, and what would
lane-level scheduling buy?  Rebuilds the kernel's slot order / waves / per-lane segment lists from an
oracle state of the bench workload (CPU only) and counts wave iterations under

  P0  what ships: one candidate per lane per trip, the contact block runs when ANY lane is in contact
  P1  stall-and-batch inside a segment: a lane that meets a contact parks it; the wave runs a contact
      iteration only when every lane is parked or done with the segment

priced with the instruction counts of the shipped kernel (common part A, far tail, contact tail).
Manual analysis script (not collected by pytest); it lives under tests/ because it drives the oracle.
    python tests/model_divergence.py [side] [steps]            bench lattice of side x side bots
    python tests/model_divergence.py blob [bots] [steps]        random blob (the reference's placement rule)
"""
import sys
import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from oracle import orclib

blob = len(sys.argv) > 1 and sys.argv[1] == "blob"
orclib.lib().orc_set_num_threads(orclib.usable_cpus())
if blob:
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    P = orclib.default_params(nCells=n, nDead=0, seed=7, phase_std=0.6, max_time=1e9, light_x=-40.0, light_y=0.0)
    sim = orclib.Sim(P, reset=True)
    pos0 = sim.get("pos").copy()
    ORIGIN, G = 64.0, 512
else:
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    n = side * side
    P = orclib.default_params(nCells=n, nDead=0, seed=1, phase_std=0.0, max_time=1e9, light_x=-230.0, light_y=0.0,
                              grid=2048, arena_half=240.0)
    sim = orclib.Sim(P, reset=True, hex=True)
    pos0 = bench.square_lattice(n, bench.LATTICE_PITCH)
    sim.set("pos", pos0)
    ORIGIN, G = 240.0, 2048
sim.run(steps)
pos, rad = sim.get("pos").astype(np.float64), sim.get("rad").astype(np.float64)

cell = float(np.float32(0.1175) * np.float32(2))
def cells(p):
    return (np.floor((p[:, 0] + ORIGIN) / cell).astype(np.int64) & (G - 1),
            np.floor((p[:, 1] + ORIGIN) / cell).astype(np.int64) & (G - 1))
gx0, gy0 = cells(pos0.astype(np.float64))       # the stale sort of step 0
order = np.argsort(gy0 * G + gx0, kind="stable")
spos, srad = pos[order], rad[order]
keys = (gy0 * G + gx0)[order]
cellS = np.searchsorted(keys, np.arange(G * G + 1))
gx, gy = cells(spos)                              # lists are looked up with the CURRENT cell

A, FAR, CON, OVH = 33, 17, 31, 3
tot = dict(p0_trips=0, p0_contact=0, p1_far=0, p1_contact=0, pairs=0, contacts=0, lanes=0, flat_trips=0, flat_contact=0,
           flat_switch_trips=0)
rng = np.random.default_rng(0)
waves = rng.choice(n // 64, size=min(400, n // 64), replace=False)
for w in waves:
    sl = np.arange(w * 64, w * 64 + 64)
    flat = [[] for _ in range(64)]   # PF: every lane walks its own five ranges back to back
    for r in range(-2, 3):
        lists = []
        for s in sl:
            row = ((gy[s] + r) & (G - 1)) * G
            lo, hi = cellS[row + ((gx[s] - 2) & (G - 1))], cellS[row + ((gx[s] + 2) & (G - 1)) + 1]
            j = np.arange(lo, hi)
            d = np.hypot(spos[j, 0] - spos[s, 0], spos[j, 1] - spos[s, 1])
            c = (d < srad[j] + srad[s]) & (j != s)
            lists.append(c)
        for i, c in enumerate(lists):
            flat[i].append(c)
        L = max(len(c) for c in lists)
        if L == 0:
            continue
        M = np.zeros((64, L), bool)
        V = np.zeros((64, L), bool)
        for i, c in enumerate(lists):
            M[i, :len(c)] = c
            V[i, :len(c)] = True
        tot["p0_trips"] += L
        tot["p0_contact"] += int(M.any(axis=0).sum())
        tot["pairs"] += int(V.sum())
        tot["contacts"] += int(M.sum())
        # P1: simulate
        posn = np.zeros(64, int)
        lens = np.array([len(c) for c in lists])
        parked = np.zeros(64, bool)
        far = con = 0
        while True:
            act = (~parked) & (posn < lens)
            if act.any():
                far += 1                           # every active lane evaluates the common part of its next candidate
                idx = np.flatnonzero(act)
                isc = M[idx, posn[idx]]
                parked[idx[isc]] = True            # contact: park it (the far tail of these lanes is wasted work)
                posn[idx[~isc]] += 1
            elif parked.any():
                con += 1
                posn[parked] += 1
                parked[:] = False
            else:
                break
        tot["p1_far"] += far
        tot["p1_contact"] += con
    # PF: the wave runs until its longest lane is done; the contact block runs when any lane's CURRENT candidate is a
    # contact; a "switch trip" is one in which some lane steps from one range to the next
    tl = [np.concatenate(f) if f else np.zeros(0, bool) for f in flat]
    Lf = max(len(c) for c in tl)
    Mf = np.zeros((64, Lf), bool)
    Sw = np.zeros((64, Lf), bool)
    for i, c in enumerate(tl):
        Mf[i, :len(c)] = c
        ends = np.cumsum([len(x) for x in flat[i]])[:-1]
        Sw[i, ends[ends < Lf]] = True
    tot["flat_trips"] += Lf
    tot["flat_contact"] += int(Mf.any(axis=0).sum())
    tot["flat_switch_trips"] += int(Sw.any(axis=0).sum())
    tot["lanes"] += 64
# PFS: the flattened walk with the 256 bots of a workgroup's tile dealt to its four waves by list length (longest
# first), so that a wave's lanes finish together
pfs = dict(trips=0, contact=0, switch=0, pairs=0)
tiles = rng.choice(n // 256, size=min(100, n // 256), replace=False)
for tI in tiles:
    sl = np.arange(tI * 256, tI * 256 + 256)
    lists, ends_all = [], []
    for s in sl:
        parts = []
        for r in range(-2, 3):
            row = ((gy[s] + r) & (G - 1)) * G
            lo, hi = cellS[row + ((gx[s] - 2) & (G - 1))], cellS[row + ((gx[s] + 2) & (G - 1)) + 1]
            j = np.arange(lo, hi)
            d = np.hypot(spos[j, 0] - spos[s, 0], spos[j, 1] - spos[s, 1])
            parts.append((d < srad[j] + srad[s]) & (j != s))
        lists.append(np.concatenate(parts))
        ends_all.append(np.cumsum([len(x) for x in parts])[:-1])
    order2 = np.argsort([-len(c) for c in lists], kind="stable")
    for w4 in range(4):
        idx = order2[w4 * 64:(w4 + 1) * 64]
        Lf = max(len(lists[i]) for i in idx)
        Mf = np.zeros((64, Lf), bool)
        Sw = np.zeros((64, Lf), bool)
        for a, i in enumerate(idx):
            Mf[a, :len(lists[i])] = lists[i]
            e = ends_all[i]
            Sw[a, e[e < Lf]] = True
            pfs["pairs"] += len(lists[i])
        pfs["trips"] += Lf
        pfs["contact"] += int(Mf.any(axis=0).sum())
        pfs["switch"] += int(Sw.any(axis=0).sum())
W = len(waves)
pairs_per_bot = tot["pairs"] / tot["lanes"]
print(f"{n} bots after {steps} steps; {W} waves sampled; candidate pairs per bot {pairs_per_bot:.1f}, "
      f"contacts per bot {tot['contacts'] / tot['lanes']:.2f}")
p0 = tot["p0_trips"] * (A + FAR + OVH) + tot["p0_contact"] * CON
print(f"P0: trips per wave {tot['p0_trips'] / W:.1f}, of which with the contact block {tot['p0_contact'] / W:.1f} "
      f"({tot['p0_contact'] / tot['p0_trips']:.2f}); modelled VALU per wave {p0 / W:.0f}")
p1 = tot["p1_far"] * (A + FAR + OVH + 4) + tot["p1_contact"] * (CON + 6)
print(f"P1: far iterations per wave {tot['p1_far'] / W:.1f}, contact iterations {tot['p1_contact'] / W:.1f}; "
      f"modelled VALU per wave {p1 / W:.0f}  ({p1 / p0:.3f} of P0)")
SWITCH = 5   # range switch through a per-lane queue in LDS: 2 moves, pointer add, 1 select for the look-ahead load
pf = tot["flat_trips"] * (A + FAR + OVH + 1) + tot["flat_contact"] * CON + tot["flat_switch_trips"] * SWITCH
print(f"PF (per-lane flattened walk): trips per wave {tot['flat_trips'] / W:.1f} (lane utilisation "
      f"{tot['pairs'] / 64 / tot['flat_trips']:.3f} against {tot['pairs'] / 64 / tot['p0_trips']:.3f}), with the contact "
      f"block {tot['flat_contact'] / tot['flat_trips']:.2f}, with a range switch {tot['flat_switch_trips'] / tot['flat_trips']:.2f}; "
      f"modelled VALU per wave {pf / W:.0f}  ({pf / p0:.3f} of P0)")
WS = 4 * len(tiles)
pfs_cost = pfs["trips"] * (A + FAR + OVH + 1) + pfs["contact"] * CON + pfs["switch"] * SWITCH
print(f"PFS (flattened walk, lanes dealt to the tile's four waves by list length): trips per wave {pfs['trips'] / WS:.1f} "
      f"(lane utilisation {pfs['pairs'] / 64 / pfs['trips']:.3f}), with the contact block {pfs['contact'] / pfs['trips']:.2f}, "
      f"with a range switch {pfs['switch'] / pfs['trips']:.2f}; modelled VALU per wave {pfs_cost / WS:.0f}  "
      f"({(pfs_cost / WS) / (p0 / W):.3f} of P0)")
ideal = tot["pairs"] / 64 * (A + OVH) + (tot["pairs"] - tot["contacts"]) / 64 * FAR + tot["contacts"] / 64 * CON
print(f"divergence-free bound: {ideal / W:.0f} ({ideal / p0:.3f} of P0)")
