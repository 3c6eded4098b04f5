"""What "within 1e-5 of the reference CUDA path" can mean: the FMA / __powf bracket of the oracle's own
source (tests/fma_bracket.py, oracle/Makefile), measured on every BASELINE.json configuration.

The reference is compiled with nvcc's default -fmad=true (/root/reference/Makefile:79-89) and uses
__powf (particlebot_kernel_impl.cuh:586,589); the oracle is the contraction-free, x*x restatement.
Three builds of oracle/pb_oracle.c -- exact, kernels contracted, contracted + exp2f(2*log2f(x)) -- are
run through teacher-forced 10-step windows; tests/golden/fma_bracket/oracle_builds.json holds the
statistics (generator: tests/golden/make_fma_bracket.py).  This file
  * checks that the bracket builds differ from the oracle ONLY in the kernel functions (FMA
    instructions nowhere else; placement, dead-bot draw and host loop bit-identical),
  * re-measures the small cases and holds them to the fixture,
  * asserts the statements DESIGN.md section 4 makes from the fixture: in a 10-step window a contracted
    build of the same source agrees with the oracle to <= 1e-6 relative at the 99th percentile and
    in the centre of mass, at most a few bots in 10^5 are beyond 1e-5 (threshold flips), and
    without re-synchronisation the 99th percentile leaves 1e-5 after 26 to > 100 steps.
"""
import json
import os
import re
import subprocess

import numpy as np
import pytest

import fma_bracket as fb

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "fma_bracket", "oracle_builds.json")

# functions of pb_oracle.c between its two "device code" markers (+ the OpenMP bodies gcc outlines)
KERNEL_FUNCTIONS = {"orc_integrateSystem", "orc_calcHash", "orc_updateRad_light_wave", "intersects_segment",
                    "intersects_circle", "in_shadow", "orc_updatePhase", "pair_force", "orc_collideSpheres",
                    "obstacle_tail", "orc_collide", "grid_pos", "grid_hash", "dot2", "len2"}


@pytest.fixture(scope="module")
def fixture():
    return json.load(open(FIXTURE))


def fma_functions(so):
    """{function: number of FMA instructions} from the library's disassembly"""
    dis = subprocess.check_output(["objdump", "-d", "--no-show-raw-insn", so], text=True)
    counts, fn = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            fn = m.group(1)
        elif re.search(r"\bvfn?m(add|sub)", line):
            counts[fn] = counts.get(fn, 0) + 1
    return counts


def test_only_the_kernel_functions_are_contracted(orc):
    here = os.path.dirname(orc.__file__)
    orc.lib(), [orc.variant_lib(v) for v in orc.BRACKET_VARIANTS]  # builds all three
    assert orc.lib().orc_build_variant() == b"exact"
    assert fma_functions(os.path.join(here, "libpb_oracle.so")) == {}, "the oracle must not contain one FMA"
    for v in orc.BRACKET_VARIANTS:
        counts = fma_functions(os.path.join(here, f"libpb_oracle_{v}.so"))
        assert counts, v
        for fn in counts:
            base = fn.split(".")[0]  # orc_collide._omp_fn.0, pair_force.constprop.0 ...
            assert base in KERNEL_FUNCTIONS, f"{v}: FMA outside the kernels, in {fn}"
        # the pair force and the force kernel are where it matters
        assert any(f.startswith("pair_force") or f.startswith("orc_collide") for f in counts)


@pytest.mark.parametrize("cfg,over", [("example.cfg", {}), ("example_dead_cells.cfg", {}), ("example_obstacle.cfg", {}),
                                      ("example_object_transport.cfg", {}), ("example_gap.cfg", {}),
                                      ("example_dead_cells.cfg", {"nCells": 5000, "nDead": 1000})])
def test_host_side_is_bit_identical_across_builds(orc, cfg, over):
    """placement (particlebot.cpp:612-748), the dead-bot draw (:178-194) and the min-distance host
    loop (:215-228) are host code: the same bits from all three builds."""
    states = []
    for v in (None,) + tuple(orc.BRACKET_VARIANTS):
        P = orc.load_cfg(fb.EX(cfg), phase_std=0.0, max_time=1e9, **over)
        sim = orc.Sim(P, reset=True, variant=v)
        pos0, rad0 = sim.get("pos"), sim.get("rad")
        mn, mx = np.zeros(1, np.float32), np.zeros(1, np.float32)
        sim._L.orc_minmax_light_distance(P, pos0, sim.n, mn, mx)
        sim.run(1)  # draws the dead bots; hashes and sorts the (still resting) placement
        states.append((pos0, rad0, sim.get("dead"), mn.copy(), mx.copy(), sim.get("hash"), sim.get("index")))
        sim.close()
    for other in states[1:]:
        for a, b in zip(states[0], other):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_one_pair_force_known_difference(orc):
    """The builds are not accidentally the same code: on a contact pair the contracted build's force
    differs from the oracle's in the last bits, and the __powf model changes a far attraction term."""
    import ctypes as C
    P = orc.default_params(nCells=2, nDead=0, seed=1, phase_std=0.0)
    f32 = lambda *v: np.array(v, np.float32)
    rng = np.random.default_rng(7)
    differs = {v: 0 for v in orc.BRACKET_VARIANTS}
    for _ in range(2000):
        a = f32(*rng.uniform(-1, 1, 2))
        ang, d = rng.uniform(0, 2 * np.pi), rng.uniform(0.12, 0.4)
        b = (a + f32(np.cos(ang) * d, np.sin(ang) * d)).astype(np.float32)
        va, vb = f32(*rng.uniform(-.1, .1, 2)), f32(*rng.uniform(-.1, .1, 2))
        out = {}
        for v in (None,) + tuple(orc.BRACKET_VARIANTS):
            force, fa, fr = f32(0, 0), f32(0), f32(0)
            orc.variant_lib(v).orc_collideSpheres(C.byref(P), a, b, va, vb, 0.08, 0.09, P.attraction, force, fa, fr)
            out[v] = np.concatenate([force, fa, fr])
        for v in orc.BRACKET_VARIANTS:
            rel = np.abs(out[v] - out[None]).max() / max(np.abs(out[None]).max(), 1e-30)
            assert rel < 1e-5  # last-bit differences, never more
            differs[v] += int(not np.array_equal(out[v].view(np.uint32), out[None].view(np.uint32)))
    assert differs["fma"] > 100 and differs["fma_powf"] > differs["fma"], differs


@pytest.mark.parametrize("case", fb.CHEAP_CASES)
def test_small_cases_reproduce_the_fixture(orc, fixture, case):
    got = fb.measure_case(orc, case, lambda P: [fb.OracleCandidate(orc, P, v) for v in orc.BRACKET_VARIANTS])
    want = fixture["cases"][case]
    assert got["bots"] == want["bots"]
    for line in fb.format_rows(got):
        print(line)
    for v in orc.BRACKET_VARIANTS:
        for g, w in zip(got["candidates"][v], want["candidates"][v]):
            assert g["epoch"] == w["epoch"]
            if v == "fma":
                # FMA, IEEE division and square root are exact operations: the same gcc gives the same
                # numbers on any x86-64 with FMA
                assert g == w, (case, v, g, w)
            else:
                # exp2f/log2f are libm's (CPU-dispatched variants): same statement, not the same bits
                assert g["window"]["flips"] <= w["window"]["flips"] + 2
                assert g["window"]["p99"] <= max(2 * w["window"]["p99"], 1e-7)
                assert g["window"]["com_rel"] <= max(4 * w["window"]["com_rel"], 1e-8)


def test_the_order_of_additions_alone_flips_nothing(orc):
    """oracle/libpb_oracle_order.so: the oracle's own terms, bit for bit, with a bot's CONTACT terms added after its
    last candidate -- the order of additions of the product's two-pass tolerance kernel (k_force_stream).  Over the same
    teacher-forced windows that order alone moves no bot beyond 1e-5 and leaves the 99th percentile at 1e-8: it is not
    what separates that kernel from the reference (HISTORY.md "Round 4, numerics")."""
    lib = orc.variant_lib("order")
    assert lib.orc_build_variant() == b"order"
    assert fma_functions(os.path.join(os.path.dirname(orc.__file__), "libpb_oracle_order.so")) == {}
    for case in ("cfg1_example_300", "cfg4_obstacle_500", "cfg4_object_transport_201"):
        res = fb.measure_case(orc, case, lambda P: [fb.OracleCandidate(orc, P, "order")])
        row = fb.summarise(res)["order"]
        print(fb.format_rows(res)[0])
        assert row["flips_total"] == 0 and row["p99_max"] <= 1e-7 and row["max_max"] <= 1e-6, (case, row)
        # ... but it is not the identity: some bot differs in the last bits
        assert row["max_max"] > 0.0, case


def test_fixture_covers_every_baseline_config(fixture):
    assert set(fixture["cases"]) == set(fb.CASES)
    for name, (_b, epochs, what) in fb.CASES.items():
        c = fixture["cases"][name]
        assert c["what"] == what and c["window"] == 10 and c["rtol"] == 1e-5
        for v in ("fma", "fma_powf") + (orc_devpowf_variants() if name in fb.DEVPOWF_CASES else ()):
            assert [r["epoch"] for r in c["candidates"][v]] == list(epochs)


def orc_devpowf_variants():
    from oracle import orclib
    return orclib.DEVPOWF_VARIANTS


def test_device_powf_members_touch_nothing_but_the_obstacle_sites(orc):
    """Round 5: oracle/libpb_oracle_devpowf{,_ulp}.so model the reference's DEVICE powf calls
    (particlebot_kernel_impl.cuh:214-229 shadow test, :704-705, :719 circles, :757-779 rectangle corners) as
    exp2f(y*log2f(x)) / as the correctly rounded result moved by -1 / 0 / +1 ulp.  Nothing else may differ: no FMA
    anywhere, the host side bit-identical, a configuration without obstacles bit-identical for 300 steps, and on
    a circle contact the force differs from the oracle's in the last bits only."""
    here = os.path.dirname(orc.__file__)
    for v, tag in (("devpowf", b"devpowf"), ("devpowf_ulp", b"devpowf_ulp"), ("cuda_like", b"fma+powf+devpowf")):
        assert orc.variant_lib(v).orc_build_variant() == tag
    for v in ("devpowf", "devpowf_ulp"):
        assert fma_functions(os.path.join(here, f"libpb_oracle_{v}.so")) == {}
    # no obstacles: the same bits
    ref = None
    for v in (None, "devpowf", "devpowf_ulp"):
        P = orc.load_cfg(fb.EX("example.cfg"), phase_std=0.0, max_time=1e9)
        sim = orc.Sim(P, reset=True, variant=v)
        sim.run(300)
        st = [sim.get(k) for k in ("pos", "vel", "rad", "absForce_a", "absForce_r")]
        sim.close()
        if ref is None:
            ref = st
        else:
            for a, b in zip(ref, st):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), v
    # circles on the rim of the blob: placement identical, the first step's repulsion sums differ in the last bits
    P, _ = fb.CASES["cfg4_obstacle_500_rim"][0](orc)
    sims = {v: orc.Sim(P, reset=True, variant=v) for v in (None, "devpowf", "devpowf_ulp")}
    placed = sims[None].get("pos")
    for v, sim in sims.items():
        assert np.array_equal(sim.get("pos").view(np.uint32), placed.view(np.uint32))
    touched = {v: np.zeros(sims[None].n, bool) for v in ("devpowf", "devpowf_ulp")}
    for _ in range(20):
        for sim in sims.values():
            sim.run(1)
        fr = {v: sim.get("absForce_r") for v, sim in sims.items()}
        for v in touched:
            diff = fr[v].view(np.uint32) != fr[None].view(np.uint32)
            touched[v] |= diff
            if diff.any():
                rel = np.abs(fr[v][diff] - fr[None][diff]) / np.abs(fr[None][diff])
                assert rel.max() < 1e-3, (v, rel.max())   # (a sum of a few small terms: last bits of each)
    for v, t in touched.items():
        assert 1 <= int(t.sum()) <= 60, (v, int(t.sum()))   # the bots on a circle and, later, their neighbours
    for sim in sims.values():
        sim.close()


def test_what_the_device_powf_bracket_says(fixture):
    """The statements DESIGN.md section 4 quotes from the fixture.  (a) The reference's obstacle files verbatim never
    reach their obstacles inside a window of the first 3 100 steps -- nor inside BASELINE configs[3]'s 120 000: the
    device-powf members are the oracle there, bit for bit.  (b) Where bots ARE on the obstacles (the file's own course
    after 230 000 - 450 000 steps; the rim cases) a non-correctly-rounded powf moves no bot beyond 1e-5 in ten steps,
    stays an order of magnitude below what FMA contraction does at the 99th percentile, and every member together
    (`cuda_like`) is the FMA + __powf member within noise.  (c) light_shadow: the shadow test is a yes/no decision
    taken once per 1 200 steps; neither model flips one here."""
    s = fixture["summary"]
    for case in ("cfg4_obstacle_500", "cfg_gap_1000", "cfg4_obstacle_500_shadow"):
        for v in ("devpowf", "devpowf_ulp"):
            assert s[case][v]["max_max"] == 0.0 and s[case][v]["flips_total"] == 0, (case, v)
    touched = 0
    for case in ("cfg4_obstacle_500_rim", "cfg4_obstacle_500_late", "cfg_gap_1000_rim"):
        for v in ("devpowf", "devpowf_ulp"):
            r = s[case][v]
            assert r["flips_total"] == 0 and r["max_max"] <= 1e-6 and r["p99_max"] <= 1e-7, (case, v, r)
            assert r["p99_max"] <= 0.25 * s[case]["fma"]["p99_max"], (case, v)
            touched += r["max_max"] > 0.0
        a, b = s[case]["cuda_like"], s[case]["fma_powf"]
        assert a["flips_total"] <= b["flips_total"] + 1 and a["p99_max"] <= 2 * b["p99_max"] + 1e-8, (case, a, b)
    assert touched >= 4   # both circle cases, both members (the gap's corner contact lasts three steps: last bits of
                          # absForce_r only, no position bit)


def test_what_the_bracket_says(fixture):
    """The statements DESIGN.md section 4 quotes.  `cfg3_arena_crop_10k` is centred on the origin
    (|COM| ~ 1e-3), which makes a RELATIVE centre-of-mass figure meaningless there: its absolute
    deviation is held instead."""
    flips = bot_windows = 0
    for name, c in fixture["cases"].items():
        for v, recs in c["candidates"].items():
            if v == "order":
                # the ORDER of additions alone (exact terms, contacts added last) moves no bot beyond 1e-5, anywhere
                # (p99 <= 1e-7 in the first 3 100 steps; 2.2e-7 once the blob is spread over the obstacle course)
                lim = 1e-6 if name == "cfg4_obstacle_500_late" else 1e-7
                assert all(r["window"]["flips"] == 0 and r["window"]["p99"] <= lim for r in recs), (name, recs)
                continue
            if v in ("devpowf", "devpowf_ulp", "cuda_like"):
                continue   # test_what_the_device_powf_bracket_says
            for r in recs:
                w = r["window"]
                # 10-step teacher-forced window: the bulk agrees far inside 1e-5 ...
                assert w["median"] <= 1e-7 and w["p99"] <= 1e-6, (name, v, r)
                if name == "cfg3_arena_crop_10k":
                    assert w["com_abs"] <= 1e-8, (name, v, r)
                else:
                    assert w["com_rel"] <= 1e-7, (name, v, r)
                # ... a bot beyond 1e-5 is a threshold flip: bounded by a few force jumps (2.5 N * dt^2)
                assert w["max_abs"] <= 10 * 2.5e-4
                flips += w["flips"]
                bot_windows += c["bots"]
                # un-resynchronised, the 99th percentile holds 1e-5 for at least 25 steps (SURVEY.md 0.5
                # measured 10-20 on its probe) ...
                # (round 5's cases -- the 1000-bot gap file, walls pushing into a freshly placed blob, a blob strung out
                #  over the obstacle course -- are looser: 21 at the least; the round-4 cases 26)
                old = name in ("cfg1_example_300", "cfg2a_dead_cells_100", "cfg2b_dead_cells_10k", "cfg3_arena_crop_10k",
                               "cfg4_obstacle_500", "cfg4_object_transport_201", "cfg5_member_1e5_dead20")
                assert r["break_p99"] is None or r["break_p99"] >= (25 if old else 20), (name, v, r)
                # ... and a blob of >= 10^4 bots has its first bot beyond 1e-5 within 20 steps and 1 % of
                # them within 70 (the jammed lattice never leaves 1e-5; small blobs take 26 to > 100 steps)
                if name in ("cfg2b_dead_cells_10k", "cfg5_member_1e5_dead20"):
                    assert r["break_max"] <= 20 and 25 <= r["break_p99"] <= 70, (name, v, r)
    # the reference's own build-to-build spread is NOT "every bot within 1e-5" once there are enough
    # bots: a few per 10^5 bot-windows sit on a force-law discontinuity (contact / static friction)
    rate = flips / bot_windows
    assert 0 < rate < 1e-4, (flips, bot_windows)
    big = fixture["summary"]["cfg5_member_1e5_dead20"]["fma"]
    assert big["flips_total"] >= 1 and big["first_break_max"] <= 10
    small = fixture["summary"]["cfg1_example_300"]["fma"]
    assert small["flips_total"] == 0 and small["first_break_max"] >= 25


def test_gpu_box_record_agrees_with_this_fixture(fixture):
    """tests/golden/fma_bracket/hip_streamlined.json was written on the GPU box (another CPU, the FMA build compiled
    there from the same source): its FMA-build statistics must be THIS container's fixture to the last digit -- the
    bracket is a deterministic function of (source, gcc), not of the machine -- and the record must hold what
    DESIGN.md section 4 quotes for the product's tolerance kernel."""
    rec = json.load(open(os.path.join(HERE, "golden", "fma_bracket", "hip_streamlined.json")))
    assert set(rec["cases"]) == set(fixture["cases"]) - set(fb.FIXTURE_ONLY_CASES)
    flips = {"hip_streamlined": 0, "fma": 0}
    for name, c in rec["cases"].items():
        assert c["candidates"]["fma"] == fixture["cases"][name]["candidates"]["fma"], name
        for cand in flips:
            flips[cand] += sum(r["window"]["flips"] for r in c["candidates"][cand])
            for r in c["candidates"][cand]:
                assert r["window"]["median"] <= 1e-7 and r["window"]["p99"] <= 1e-6
    # the tolerance kernel flips no more bots than the FMA build of the reference's own arithmetic does (+ slack)
    assert flips["fma"] == 20 and flips["hip_streamlined"] <= 2 * flips["fma"], flips   # (round 4: 18; + the gap rim case's 2)
