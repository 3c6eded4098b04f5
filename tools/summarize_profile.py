#!/usr/bin/env python3
"""Distil the rocprofv3 CSVs written by tools/profile.sh into a short markdown summary
(per-kernel time from the kernel trace; per-kernel PMC averages; HBM traffic corrected as
MI355X_MICROARCH.md prescribes: FETCH_SIZE doubled for wide coalesced reads, both in KiB)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def short(name):
    for pre in ("void ", "(anonymous namespace)::"):
        name = name.replace(pre, "")
    name = name.split("(")[0]
    return name.strip()[:60]


def signature(name):
    """A kernel name as pbSimForceKernelName spells it: template arguments and argument types, no `void`, no
    namespace, single spaces."""
    for pre in ("void ", "(anonymous namespace)::"):
        name = name.replace(pre, "")
    return " ".join(name.split())


FULL_NAMES = {}   # short name -> the trace's full Kernel_Name


def build_stamp():
    """lib/build_stamp.json of the libraries this profile ran (csrc/Makefile writes it): content hash of the force
    kernels' sources + commit."""
    import json
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        with open(os.path.join(here, "particlerobotsimulations_amd", "lib", "build_stamp.json")) as fh:
            return json.load(fh)
    except Exception:
        return None


def kernel_trace(root):
    rows = []
    for f in find(os.path.join(root, "trace"), "*kernel_trace.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append(r)
    agg = defaultdict(list)
    meta = {}
    for r in rows:
        k = short(r["Kernel_Name"])
        FULL_NAMES[k] = r["Kernel_Name"]
        agg[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[k] = (r.get("VGPR_Count", "?"), r.get("SGPR_Count", "?"), r.get("LDS_Block_Size", "?"),
                   r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")))
    return agg, meta


def code_object_registers():
    """{short kernel name: (vgpr_count, sgpr_count, LDS bytes, scratch bytes)} from the CODE OBJECTS' own metadata
    (the .hip_fatbin of every csrc/build/*.o): rocprofv3's VGPR_Count column is the allocation as the dispatch
    packet encodes it (granules), not the kernel's register count -- an occupancy argument needs the latter."""
    import subprocess
    import tempfile
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    llvm = "/opt/rocm/lib/llvm/bin"
    out = {}
    for obj in sorted(glob.glob(os.path.join(here, "particlerobotsimulations_amd", "csrc", "build", "*.o"))):
        if obj.endswith(".host.o"):
            continue
        try:
            with tempfile.TemporaryDirectory() as td:
                fat, co = os.path.join(td, "fat"), os.path.join(td, "co")
                subprocess.check_call([f"{llvm}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj],
                                      stderr=subprocess.DEVNULL)
                subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                                       "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                                      stderr=subprocess.DEVNULL)
                notes = subprocess.check_output([f"{llvm}/llvm-readelf", "--notes", co], text=True)
        except Exception:
            continue
        cur = {}
        for line in notes.splitlines():
            line = line.strip()
            for key in (".group_segment_fixed_size", ".name", ".private_segment_fixed_size", ".sgpr_count", ".vgpr_count"):
                if line.startswith(key + ":") or line.startswith("- " + key + ":"):
                    cur[key] = line.split(":", 1)[1].strip()
            if ".vgpr_count" in cur and ".name" in cur and ".sgpr_count" in cur:
                name = cur[".name"]
                for tool in (f"{llvm}/llvm-cxxfilt", "c++filt"):
                    try:
                        name = subprocess.check_output([tool, cur[".name"]], text=True, stderr=subprocess.DEVNULL).strip()
                        break
                    except Exception:
                        pass
                out[short(name)] = (cur[".vgpr_count"], cur[".sgpr_count"], cur.get(".group_segment_fixed_size", "?"),
                                    cur.get(".private_segment_fixed_size", "?"))
                cur = {}
    return out


def pmc(root, sub):
    out = defaultdict(lambda: defaultdict(list))
    for f in find(os.path.join(root, sub), "*counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def valu_roofline(root, d, avg_ns_profiled, simds=1024):
    """Fraction of the launch the SIMDs would need just to ISSUE the kernel's VALU instructions at the
    rates tools/valu_rate measures at 8 waves per SIMD (simple fp32 ops; v_rcp/v_sqrt/v_rsq), against
    the un-profiled launch time of the same command (profiled runs clock lower)."""
    import json
    try:
        rates = json.loads([l for l in open(os.path.join(root, "valu_rate.txt")) if l.startswith("{")][-1])
        bench = json.load(open(os.path.join(root, "bench_unprofiled.json")))
    except Exception:
        return None
    if not (d.get("SQ_INSTS_VALU") and d.get("SQ_WAVES")) or d.get("SQ_INSTS_VALU_TRANS_F32") is None:
        return None
    r = rates["ns_per_wave_instr"]
    ns_simple = (r["fma"] + r["mul"] + r["cmp_sel_mix"]) / 3.0
    ns_trans = (r["rcp"] + r["sqrt"] + r["rsq"]) / 3.0
    valu, trans = d["SQ_INSTS_VALU"], d["SQ_INSTS_VALU_TRANS_F32"]
    valu_ns = ((valu - trans) * ns_simple + trans * ns_trans) / simds
    launch_us = bench.get("roofline", {}).get("avg_launch_us")
    if launch_us is None:
        # ensemble workloads: the line has no per-launch figure; price against the PROFILED launch time
        launch_us = avg_ns_profiled / 1e3
    mhz = rates.get("in_kernel_mhz")
    return {"valu_roofline_frac": valu_ns / 1e3 / launch_us, "valu_time_us": valu_ns / 1e3,
            "valu_rate_mhz": (sum(mhz.values()) / len(mhz)) if mhz else None,
            "valu_rate_cycles": rates.get("cycles_per_wave_instr"),
            "launch_us_unprofiled": launch_us, "ns_simple": ns_simple, "ns_trans": ns_trans,
            "trans_per_wave": trans / d["SQ_WAVES"], "valu_per_wave": valu / d["SQ_WAVES"]}


def main():
    root = sys.argv[1]
    traffic_json = sys.argv[2] if len(sys.argv) > 2 else None
    # the kernel whose counters go into traffic_json: first kernel (by total time) whose name contains this
    traffic_kernel = os.environ.get("PB_TRAFFIC_KERNEL", "k_force")
    agg, meta = kernel_trace(root)
    total = sum(sum(v) for v in agg.values()) or 1
    print(f"# rocprofv3 summary: {os.path.basename(root)}\n")
    print("## kernel trace (`rocprofv3 --kernel-trace --stats`)\n")
    regs = code_object_registers()
    print("(VGPR / SGPR / scratch: `.vgpr_count` / `.sgpr_count` / `.private_segment_fixed_size` of the code object "
          "(csrc/build/*.o); `?` where the object was not found, e.g. runtime copy kernels.  LDS, grid, wg: the trace.)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | VGPR | SGPR | scratch B | LDS | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        m = meta[k]
        r = regs.get(k, ("?", "?", "?", "?"))
        print(f"| {k} | {len(v)} | {sum(v)/1e6:.3f} | {sum(v)/len(v)/1e3:.2f} | {min(v)/1e3:.2f} | {max(v)/1e3:.2f} | "
              f"{100*sum(v)/total:.1f} | {r[0]} | {r[1]} | {r[3]} | {m[2]} | {m[3]} | {m[4]} |")
    counters = defaultdict(dict)
    for sub in ("pmc_sq", "pmc_sq2", "pmc_sq3", "pmc_fetch", "pmc_write"):
        for k, cs in pmc(root, sub).items():
            for c, vals in cs.items():
                counters[k][c] = sum(vals) / len(vals)
    print("\n## PMC averages per launch (separate passes)\n")
    for k, cs in sorted(counters.items(), key=lambda kv: -sum(agg.get(kv[0], [0]))):
        if k not in agg or sum(agg[k]) / total < 0.01:
            continue
        print(f"### {k}\n")
        for c, v in sorted(cs.items()):
            print(f"- {c}: {v:.4g}")
        d = cs
        avg_ns = sum(agg[k]) / len(agg[k])
        split = None
        if d.get("SQ_WAVE_CYCLES"):
            # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are all in quad-cycles and summed over waves:
            # the three disjoint buckets of a wave's life (MI355X_MICROARCH.md, rocprofv3 PMC slots)
            wc = d["SQ_WAVE_CYCLES"]
            split = {"issuing": d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
                     "issue_stalled": d.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                     "parked_on_waitcnt": d.get("SQ_WAIT_ANY", 0.0) / wc}
            if d.get("SQ_ACTIVE_INST_VALU"):
                split["issuing_valu"] = d["SQ_ACTIVE_INST_VALU"] / wc
            print("- derived: share of a wave's cycles: issuing {issuing:.3f} (VALU {v:.3f}), issue-stalled "
                  "{issue_stalled:.3f}, parked on s_waitcnt {parked_on_waitcnt:.3f} "
                  "(ACTIVE_INST_ANY, ACTIVE_INST_VALU, WAIT_INST_ANY, WAIT_ANY over WAVE_CYCLES)".format(
                      v=split.get("issuing_valu", float("nan")), **{k: split[k] for k in
                                                                    ("issuing", "issue_stalled", "parked_on_waitcnt")}))
        if d.get("SQ_INSTS_VALU") and d.get("SQ_WAVES"):
            print(f"- derived: VALU instructions per wave = {d['SQ_INSTS_VALU']/d['SQ_WAVES']:.1f}")
            clk = None
            if d.get("GRBM_GUI_ACTIVE"):
                clk = d["GRBM_GUI_ACTIVE"] / 8.0 / avg_ns  # GHz; reads high on dispatches under ~0.3 ms
            print(f"- derived: VALU instructions per SIMD per launch = {d['SQ_INSTS_VALU']/1024:.0f}; over the "
                  f"profiled launch time {avg_ns/1e3:.1f} us that is one per {avg_ns/(d['SQ_INSTS_VALU']/1024):.3f} ns "
                  f"per SIMD (datasheet floor: 2 cycles = 0.833 ns at 2.4 GHz)"
                  + (f"; GRBM_GUI_ACTIVE/8/time = {clk:.2f} GHz (unreliable under 0.3 ms)" if clk else ""))
        if d.get("SQ_THREAD_CYCLES_VALU") and d.get("SQ_ACTIVE_INST_VALU"):
            print(f"- derived: VALU lane utilisation = "
                  f"{d['SQ_THREAD_CYCLES_VALU']/d['SQ_ACTIVE_INST_VALU']/64:.3f} "
                  f"(THREAD_CYCLES_VALU / (ACTIVE_INST_VALU x 64))")
        vr = valu_roofline(root, d, avg_ns)
        if vr:
            print(f"- derived: VALU issue roofline = {vr['valu_roofline_frac']:.3f} of the un-profiled launch time "
                  f"({vr['valu_time_us']:.1f} us of VALU issue at the microbenchmark's 8-waves/SIMD rates -- "
                  f"{vr['ns_simple']:.3f} ns per simple, {vr['ns_trans']:.3f} ns per transcendental wave-instruction "
                  f"per SIMD"
                  + (f", measured at {vr['valu_rate_mhz']:.0f} MHz in-kernel" if vr.get("valu_rate_mhz") else "")
                  + f" -- over {vr['launch_us_unprofiled']:.1f} us; {vr['trans_per_wave']:.0f} transcendentals "
                  f"of {vr['valu_per_wave']:.0f} VALU instructions per wave)")
        if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
            fetch = 2.0 * d.get("FETCH_SIZE", 0.0) * 1024  # gfx950: reports half of wide coalesced reads
            write = d.get("WRITE_SIZE", 0.0) * 1024
            # (the dominant k_force form: the kernels are visited by descending total time; since round 3 the
            #  fused and the un-fused launch are the same kernel, `fuse` being a runtime flag)
            if traffic_json and traffic_kernel in k and not getattr(main, "_wrote_traffic", False):
                main._wrote_traffic = True
                import json
                valu = {}
                if split:
                    valu["wave_cycle_split"] = split
                if d.get("SQ_INSTS_VALU") and d.get("SQ_WAVES"):
                    valu["valu_insts_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
                if d.get("SQ_THREAD_CYCLES_VALU") and d.get("SQ_ACTIVE_INST_VALU"):
                    valu["valu_lane_utilisation"] = d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"] / 64
                if vr:
                    valu.update({kk: vr[kk] for kk in ("valu_roofline_frac", "valu_time_us", "launch_us_unprofiled",
                                                       "ns_simple", "ns_trans", "trans_per_wave", "valu_rate_mhz",
                                                       "valu_rate_cycles")})
                with open(traffic_json, "w") as fh:
                    json.dump({**valu, "kernel": k, "kernel_signature": signature(FULL_NAMES.get(k, k)),
                               "build": build_stamp(), "calls": len(agg[k]),
                               "profile": os.path.basename(root), "fetch_bytes_per_launch": fetch,
                               "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
                               "avg_launch_us_profiled": avg_ns / 1e3,
                               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of "
                                         "`python3 tools/bench_legs.py " + os.environ.get("PB_PROFILE_ARGS", "--steps 400 --warmup 100 --no-cpu-baseline "
                                         "--no-survey-literal --no-streamlined --no-large-arena --no-clock --no-blob --no-ensemble-leg") + "`; FETCH_SIZE x2 (gfx950 wide-read correction) "
                                         "x1024, WRITE_SIZE x1024 (MI355X_MICROARCH.md, HBM section)"}, fh)
            print(f"- derived: HBM-side traffic per launch = read {fetch/1e6:.1f} MB (FETCH_SIZE x2 x1024) + "
                  f"write {write/1e6:.1f} MB = {(fetch+write)/1e6:.1f} MB; at {avg_ns/1e3:.1f} us/launch = "
                  f"{(fetch+write)/avg_ns:.1f} GB/s")
        if d.get("TCC_HIT_sum") is not None and d.get("TCC_MISS_sum") is not None:
            h, m = d["TCC_HIT_sum"], d["TCC_MISS_sum"]
            if h + m:
                print(f"- derived: L2 hit rate = {h/(h+m):.3f}")
        print()


if __name__ == "__main__":
    main()
