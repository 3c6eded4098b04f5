"""The streamlined force kernel (force variant 3) against the FMA / __powf bracket.

tests/test_fma_bracket.py measures how far a contracted build of the oracle's own source -- what
nvcc's default -fmad=true and __powf make of the reference (/root/reference/Makefile:79-89,
particlebot_kernel_impl.cuh:586,589) -- drifts from the oracle in teacher-forced 10-step windows.
Here the GPU's streamlined kernel goes through the SAME windows of the SAME BASELINE.json
configurations, next to the FMA build (the + __powf build is in the CPU fixture), and is held to the bracket:

  * bulk: median 0-level, 99th percentile <= 1e-6 relative, centre of mass <= 1e-7 relative
    (absolute on the origin-centred lattice) -- the figures the bracket builds reach;
  * bots beyond 1e-5 ("flips": a bot that lands on the other side of the contact or static-friction
    discontinuity): at most 2 x the bracket's count + a floor of 2 per window set;
  * a flipped bot is at most a few force jumps away (2.5 N * dt^2 = 2.5e-4 per step it persists);
  * un-resynchronised, the 99th percentile holds 1e-5 at least 20 steps.

Every exact kernel (variants 0-2, all lanes-per-bot forms, the resident kernel) is bit-identical
to the oracle and needs none of this.  The statistics are written to
gpurun_out/fma_bracket_gpu.json; a copy of a run on MI355X is committed as
tests/golden/fma_bracket/hip_streamlined.json (DESIGN.md section 4 quotes it)."""
import json
import os

import pytest

import fma_bracket as fb

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "fma_bracket_gpu.json")
_results = {}


@pytest.fixture(scope="module")
def pb():
    import particlerobotsimulations_amd as pb
    pb.legacy.cudaInit(0, None)
    return pb


@pytest.fixture(scope="module", autouse=True)
def _write_results():
    yield
    if _results:
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        with open(OUT, "w") as f:
            json.dump({"window": fb.WINDOW, "horizon": fb.HORIZON, "rtol": fb.RTOL,
                       "candidates": {"hip_streamlined": "libparticlebot_hip.so, pbSimSetForceVariant(sim, 3)",
                                      "fma": "oracle/libpb_oracle_fma.so"},
                       "summary": {n: fb.summarise(r) for n, r in _results.items()}, "cases": _results}, f, indent=1)


@pytest.mark.parametrize("case", [c for c in fb.CASES if c not in fb.FIXTURE_ONLY_CASES])
def test_streamlined_kernel_inside_the_bracket(pb, orc, case):
    orc.lib().orc_set_num_threads(orc.usable_cpus())
    orc.variant_lib("fma").orc_set_num_threads(orc.usable_cpus())

    def factory(P):
        # (the FMA build runs beside the kernel so that both see the same box; the + __powf build adds nothing --
        #  the same within noise, tests/golden/fma_bracket/oracle_builds.json -- and a third of the oracle's CPU time)
        return [fb.HipCandidate(pb, P), fb.OracleCandidate(orc, P, "fma")]

    res = fb.measure_case(orc, case, factory)
    _results[case] = res
    for line in fb.format_rows(res):
        print(line)
    rows = fb.summarise(res)
    hip, bracket = rows["hip_streamlined"], rows["fma"]
    centred = case == "cfg3_arena_crop_10k"
    for r in res["candidates"]["hip_streamlined"]:
        w = r["window"]
        assert w["median"] <= 1e-7 and w["p99"] <= 1e-6, (case, r)
        assert (w["com_abs"] <= 1e-8) if centred else (w["com_rel"] <= 1e-7), (case, r)
        assert w["max_abs"] <= fb.WINDOW * 2.5e-4, (case, r)
        assert r["break_p99"] is None or r["break_p99"] >= 20, (case, r)
    assert hip["flips_total"] <= 2 * bracket["flips_total"] + 2, (case, hip, bracket)
