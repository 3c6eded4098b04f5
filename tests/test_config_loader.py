"""CPU tests of the .cfg loader (main.cpp:594-816, 832-939): the product's C++ loader
(libparticlebot_host.so) against the oracle's C restatement, field by field, on every shipped
example, on the loader's documented quirks, and -- when the reference checkout is present -- on the
reference's own examples/*.cfg."""
import glob
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLES = sorted(glob.glob(os.path.join(ROOT, "examples", "example*.cfg")))
REF_EXAMPLES = sorted(glob.glob("/root/reference/examples/*.cfg"))


@pytest.fixture(scope="module")
def host():
    from particlerobotsimulations_amd import _capi, host
    if not os.path.exists(_capi.HOST_SO):
        import __graft_entry__
        __graft_entry__.build()
    return host


def same(flat, P):
    """every field the two structs share, compared exactly"""
    diffs = []
    for name, _ in P._fields_:
        a, b = getattr(flat, name), getattr(P, name)
        if hasattr(a, "__len__") and not isinstance(a, bytes):
            a, b = list(a), list(b)
        if a != b:
            diffs.append((name, a, b))
    return diffs


@pytest.mark.parametrize("path", EXAMPLES, ids=[os.path.basename(p) for p in EXAMPLES])
def test_examples_resolve_identically(host, orc, path):
    flat = host.load_config(path)
    P = orc.load_cfg(path)
    assert same(flat, P) == []
    assert flat.gridSizeX == 512 and flat.worldOriginX == -64.0 and flat.wallHalf == 64.0
    assert abs(flat.cellSizeX - 0.235) < 1e-6


@pytest.mark.skipif(not REF_EXAMPLES, reason="reference checkout not present")
@pytest.mark.parametrize("path", REF_EXAMPLES, ids=[os.path.basename(p) for p in REF_EXAMPLES])
def test_shipped_examples_equal_reference_examples(host, path):
    """examples/*.cfg in this repo carry the reference's parameter sets (own comments)."""
    mine = os.path.join(ROOT, "examples", os.path.basename(path))
    a, b = host.load_config(mine), host.load_config(path)
    assert bytes(a) == bytes(b)


def test_example_values(host):
    c = host.load_config(os.path.join(ROOT, "examples", "example.cfg"))
    assert (c.nCells, c.nDead, c.seed, c.light_x, c.light_y, c.max_time) == (300, 0, 5555, -2.0, 4.0, 7200.0)
    assert c.csv_filename == b"example_data.csv" and c.camera_y == 9.0
    assert c.phase_std == pytest.approx(0.6) and c.timestep == pytest.approx(0.01)
    c = host.load_config(os.path.join(ROOT, "examples", "example_gap.cfg"))
    assert c.nobstacles == 2 and list(c.x1obs)[:2] == [pytest.approx(-1.2)] * 2 and list(c.y2obs)[:2] == [-1.0, 8.0]
    c = host.load_config(os.path.join(ROOT, "examples", "example_obstacle.cfg"))
    assert c.n_cir_obstacles == 3 and list(c.r_cir_obs)[:3] == [pytest.approx(0.5), pytest.approx(0.3),
                                                                pytest.approx(0.45)]
    c = host.load_config(os.path.join(ROOT, "examples", "example_object_transport.cfg"))
    assert c.nDead == -1 and c.nCells == 201 and c.attractionFactor == 0.0


def test_loader_quirks(host, orc, tmp_path):
    """prefix matching in source order and the parse oddities, reproduced."""
    cfg = tmp_path / "q.cfg"
    cfg.write_text("\n".join([
        "# a comment",
        "Nx", "9",                             # key shorter than 4 chars: skipped together with...
        "config", "CONFIG_HEX",                # ...wait: 'Nx' is skipped, '9' is skipped, then config
        "constraint_contraction", "7.5",       # shadowed by the 10-char 'constraint' test
        "centroid_int", "3.9",                 # strtol into a float field
        "phase_update_interval", "6.7",
        "x_cirXYZ", "1 2",                     # only 5 chars compared, count comes from n_cir_obstacles (0)
        "frictionless", "0.9",                 # prefix 'friction' matches
        "time_to_dead_extra", "5",             # n=14 > strlen: no match, value consumed
        "time_to_dead", "2.5",
        "unknown_key", "1",
        "seed", "42",
        "testing", "1",
        ""]))
    flat = host.load_config(str(cfg))
    P = orc.load_cfg(str(cfg))
    assert same(flat, P) == []
    assert flat.Nx == 5 and flat.config == 0          # CONFIG_RANDOM survives
    assert flat.constraint == 7.5 and flat.constraint_contraction == 10.0
    assert flat.centroid_int == 3.0 and flat.phase_update_interval == 6.0
    assert flat.friction == pytest.approx(0.9)
    assert flat.time_to_dead == 2.5 and flat.seed == 42 and flat.testing == 1


def test_object_transport_cell_size_rule(host, orc):
    """main.cpp:932-935: a big payload widens the grid cells."""
    for rf in ("2.0", "4.0", "6.5"):
        flat = host.load_config(os.path.join(ROOT, "examples", "example_object_transport.cfg"), radFactor=rf)
        P = orc.load_cfg(os.path.join(ROOT, "examples", "example_object_transport.cfg"))
        orc.lib().orc_set_param(__import__("ctypes").byref(P), b"radFactor", rf.encode())
        orc.lib().orc_params_derive(__import__("ctypes").byref(P), 0, 0.0)
        assert flat.cellSizeX == P.cellSizeX
    assert flat.cellSizeX > 0.8


def test_extension_keys(host):
    c = host.load_config(os.path.join(ROOT, "examples", "million_bots.cfg"))
    assert (c.gridSizeX, c.numCells, c.worldOriginX, c.wallHalf, c.config) == (2048, 2048 * 2048, -240.0, 240.0, 0)
    assert c.nCells == 1_000_000


def test_force_variant_key(host):
    """`pb_force_variant` (an extension key, tried after every reference key): 0-3 select the fused engine's force
    kernel, anything else -- and no key at all -- leaves the default (-1: the exact kernels)."""
    ex = os.path.join(ROOT, "examples", "example.cfg")
    assert host.load_config(ex).forceVariant == -1
    for v in (0, 1, 2, 3):
        assert host.load_config(ex, pb_force_variant=str(v)).forceVariant == v
    assert host.load_config(ex, pb_force_variant="7").forceVariant == -1
    assert host.load_config(ex, pb_force_variant="-2").forceVariant == -1


def test_private_generator_equals_glibc_rand(host):
    """PbLibcRand (include/particlebot.h) reproduces srand()/rand() of the glibc the reference runs
    on: the placement and the dead-bot draw depend on it bit for bit."""
    import ctypes
    libc = ctypes.CDLL(None)
    libc.rand.restype = ctypes.c_int
    for seed in (0, 1, 5555, 6666, 7777, 8888, 9999, 2**31 + 5, 2**32 - 1):
        libc.srand(ctypes.c_uint(seed))
        want = [libc.rand() for _ in range(5000)]
        got = host.libc_rand_draws(seed, 5000).tolist()
        assert got == want, seed


KEYS = ["camera_y", "camera_x", "nobstacles", "x1obs", "x2obs", "y1obs", "y2obs", "n_cir_obstacles", "x_cir_obs",
        "y_cir_obs", "r_cir_obs", "min_radius", "max_radius", "centroid_int", "centroid_radius", "centroid_steps",
        "radFactor", "massFactor", "frictionFactor", "attractionFactor", "dump_interval", "sort_interval",
        "testing", "friction", "spring", "damping", "shear", "constraint", "constrained_contraction",
        "constraint_contraction", "attraction", "boundaryDamping", "gravity", "nCells", "nDead", "time_to_dead",
        "max_time", "seed", "light_radius", "light_x", "light_y", "timestep", "light_shadow", "csv_filename",
        "video_filename", "rise_period", "phase_std", "display_shadow", "phase_update_interval", "Nx", "config",
        "DISPLAY_INTERVAL", "VIDEO_INTERVAL"]


def test_loader_fuzz_against_oracle(host, orc, tmp_path):
    """Randomised files: known keys, truncated / extended / re-cased keys (the prefix matching makes
    those interesting), short lines, comments, numeric and junk values, obstacle lists of random
    length.  The product's loader and the oracle's must resolve every file to the same parameters."""
    from hypothesis import HealthCheck, given, settings, strategies as st

    def mutate(key, how, tail):
        if how == 0:
            return key
        if how == 1:
            return key + tail                      # longer: still matches on the prefix
        if how == 2:
            return key[:max(1, len(key) - 1)]      # one character short: usually no match
        if how == 3:
            return key.upper()
        return tail + key

    number = st.one_of(st.integers(-5, 2000).map(str),
                       st.floats(-50, 50, allow_nan=False, width=32).map(lambda f: f"{f:.6g}"),
                       st.sampled_from(["", "abc", "1e3", "0x10", "  7", "3.5.1", "-", "1 2 3", "0.25 0.5 0.75 1 2"]))
    line_pair = st.tuples(st.sampled_from(KEYS), st.integers(0, 4), st.sampled_from(["_x", "2", "zz", "#"]), number)
    filler = st.sampled_from(["# comment", "", "ab", "xyz", "    "])

    @settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
    @given(st.lists(st.one_of(line_pair, filler), min_size=0, max_size=40))
    def run(items):
        lines = []
        for it in items:
            if isinstance(it, tuple):
                key, how, tail, value = it
                lines += [mutate(key, how, tail), value]
            else:
                lines.append(it)
        cfg = tmp_path / "fuzz.cfg"
        cfg.write_text("\n".join(lines) + "\n")
        flat = host.load_config(str(cfg))
        P = orc.load_cfg(str(cfg))
        assert same(flat, P) == [], "\n".join(lines)

    run()
