"""The two neighbour walks of the streamlined kernel (force variant 3; csrc/pb_stream.hip, WALK).
Row by row, the wave runs the longest row of its 64 lanes five times; flattened, every lane walks its own five ranges
back to back and the wave runs its longest list.  Every bot meets the same candidates in the same order (and both forms
list the same 10 contacts per lane before evaluating further ones in place), so the two walks must agree BIT FOR BIT,
pile-ups against a wall included.  The engine chooses per batch at each re-sort: flattened on the reference's kind of random blob, row by row on the bench
lattice.  The kernel's parity with the oracle is tests/test_gpu_streamlined.py / test_gpu_fma_bracket.py, which run
the automatic choice."""
import os

import numpy as np
import pytest

from helpers import assert_bit_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pb():
    import particlerobotsimulations_amd as pb
    pb.legacy.cudaInit(0, None)
    return pb


def run_walks(pb, make, steps):
    """the same state stepped under walk 0, walk 1 and the automatic choice"""
    out = {}
    for mode in (0, 1, -1):
        sim = make()
        sim.set_force_variant(3)
        sim.set_lanes_per_bot(1)
        sim.set_resident(1)
        sim.set_stream_walk(mode)
        assert sim.step(steps) == steps
        cfg = sim.config()
        assert cfg["force_variant"] == 3 and cfg["force_kind"] == 3
        if mode >= 0:
            assert cfg["stream_walk"] == mode
            assert sim.force_kernel_name().startswith(f"k_force_stream<false, false, {'true' if mode else 'false'}>(")
        out[mode] = (sim.get_state(), cfg["stream_walk"], sim.stream_walk_trips())
        sim.close()
    for key in ("pos", "vel", "rad", "absForce_r"):
        assert_bit_equal(out[1][0][key], out[0][0][key], f"flattened vs row-by-row walk: {key}")
        assert_bit_equal(out[-1][0][key], out[0][0][key], f"automatic vs row-by-row walk: {key}")
    return out


def test_blob_takes_the_flattened_walk_and_both_walks_agree_bit_for_bit(pb):
    """A 60 000-bot random blob grown by the reference's placement rule (pb_placement fastblob): cell occupancies vary,
    the flattened walk saves >= 7 % of the trips and is chosen."""
    import bench
    from particlerobotsimulations_amd import host
    n = 60000
    h = host.HostSim(os.path.join(ROOT, "examples", "million_bot_blob.cfg"), engine="host", nCells=str(n))
    pos = h.get("pos")
    h.close()

    def make():
        sp, keep = bench.workload_params(n, seed=1)
        sim = pb.Sim(sp, wall_half=240.0, keepalive=keep)
        sim.set_state(pos=pos, vel=np.zeros((n, 2), np.float32), rad=np.full(n, 0.0775, np.float32),
                      phase=np.zeros(n, np.float32), dead=np.zeros(n, np.int32))
        return sim
    out = run_walks(pb, make, 40)
    rows, flat = out[-1][2]
    print(f"blob: trips row-by-row {rows}, flattened {flat} ({flat / rows:.3f})")
    assert out[-1][1] == 1 and 0 < flat < 0.93 * rows


def test_lattice_keeps_the_row_by_row_walk(pb):
    """bench.py's lattice: every lane of a wave has (nearly) the same list, the flattened walk saves nothing and costs
    cache-line sharing -- the automatic choice stays row by row; pinned, the flattened walk still gives the same bits."""
    import bench
    n = 200000
    out = run_walks(pb, lambda: bench.make_sim(pb, n, bench.LATTICE_PITCH, seed=1), 30)
    rows, flat = out[-1][2]
    print(f"lattice: trips row-by-row {rows}, flattened {flat} ({flat / rows:.3f})")
    assert out[-1][1] == 0 and flat >= 0.93 * rows


def test_waves_at_the_x_wrap_fall_back_and_still_agree(pb, orc):
    """A blob straddling the grid's x-wrap: a wave with a wrapped stencil row runs the row-by-row loop inside the
    flattened kernel (wave-uniform fallback)."""
    from helpers import jittered_blob, simparams_from_orc
    rng = np.random.default_rng(5)
    n = 3000
    P = orc.default_params(nCells=n, nDead=0, seed=9, phase_std=0.0, max_time=1e9, light_x=80.0, light_y=80.0)
    sp, keep = simparams_from_orc(P)
    pos, vel, rad = jittered_blob(n, 0.16, rng, center=(63.0, 61.0))   # walls at +-64: the blob leans on the x-wrap
    zeros = np.zeros(n, np.float32)

    def make():
        sim = pb.Sim(sp, keepalive=keep)
        sim.set_state(pos=pos, vel=vel, rad=rad, phase=zeros, dead=np.zeros(n, np.int32))
        return sim
    run_walks(pb, make, 20)


def test_choice_is_made_when_variant_3_is_selected_later_and_at_each_resort(pb):
    """A batch that already has cell lists when force variant 3 is selected gets its walk chosen at once; the setter
    -1 re-evaluates; 0 / 1 pin."""
    import bench
    n = 150000
    sim = bench.make_sim(pb, n, bench.LATTICE_PITCH, seed=1)
    sim.step(5)                                  # exact kernel: no choice made yet
    assert sim.stream_walk_trips() == (0, 0)
    sim.set_force_variant(3)
    rows, flat = sim.stream_walk_trips()
    assert rows > 0 and flat > 0 and sim.config()["stream_walk"] == 0
    sim.set_stream_walk(1)
    assert sim.config()["stream_walk"] == 1
    sim.set_stream_walk(-1)
    assert sim.config()["stream_walk"] == 0 and sim.stream_walk_trips() == (rows, flat)
    sim.set_force_variant(2)
    assert sim.config()["stream_walk"] == 0      # (the exact kernels have one walk)
    sim.close()
