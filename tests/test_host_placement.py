"""CPU tests of the C++ host class without any device (Engine::HostOnly): the random placement
(Particlebot::reset, particlebot.cpp:612-748) and the dead-bot draw (:178-194) against the oracle,
bit for bit, for every shipped example; the hard-coded presets against their geometry."""
import glob
import os

import numpy as np
import pytest

from helpers import assert_bit_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLES = sorted(glob.glob(os.path.join(ROOT, "examples", "example*.cfg")))


@pytest.fixture(scope="module")
def host():
    from particlerobotsimulations_amd import host
    host.lib()
    return host


@pytest.mark.parametrize("path", EXAMPLES, ids=[os.path.basename(p) for p in EXAMPLES])
def test_random_placement_matches_oracle(host, orc, path):
    h = host.HostSim(path, engine="host")
    o = orc.Sim(orc.load_cfg(path))
    for k in ("pos", "vel", "rad", "phase"):
        assert_bit_equal(h.get(k), o.get(k), k)
    assert_bit_equal(h.get("dead"), o.get("dead"), "dead (payload flag)")
    # every bot but the payload touches the blob: nearest neighbour at (about) 2 r_min.  (Bot 2 is
    # laid r_min off the first pair's midpoint and bot 0 is binned in the wrong cell, so the
    # reference's blob does contain a few overlaps.)
    pos = h.get("pos").astype(np.float64)
    n = len(pos) - (1 if "object_transport" in path else 0)
    d = np.linalg.norm(pos[:n, None] - pos[None, :n], axis=-1) + np.eye(n) * 1e9
    assert d.min() > 0.05 and abs(np.median(d.min(1)) - 2 * 0.0775) < 1e-3


def test_placement_golden_through_the_class(host, golden_dir):
    """the class reproduces the reference-probe placement snapshot directly"""
    h = host.HostSim(os.path.join(ROOT, "examples", "example.cfg"), engine="host")
    g = np.fromfile(os.path.join(golden_dir, "ref_probe", "example_like_pos_step0.bin"), np.float32).reshape(-1, 2)
    assert_bit_equal(h.get("pos"), g, "placement vs reference probe")


def test_dead_draw_matches_oracle(host, orc):
    path = os.path.join(ROOT, "examples", "example_dead_cells.cfg")
    h = host.HostSim(path, engine="host")          # product first: creating it calls srand()
    P = orc.load_cfg(path)
    o = orc.Sim(P)
    dead = h.draw_dead()
    o.update()                                      # the oracle draws inside its first update
    assert dead.sum() == 20
    assert_bit_equal(dead, o.get("dead"), "dead set")


@pytest.mark.parametrize("n,nd", [(1000, 400), (4097, 4096), (6000, 1), (5000, 2500)])
def test_dead_draw_order_statistics_tree_equals_the_erase_loop(host, orc, n, nd):
    """The product draws the dead bots from a Fenwick tree of alive flags (O(log N) per bot); the oracle
    keeps the reference's literal `rand() % size` + vector erase (particlebot.cpp:178-194).  Same bots."""
    path = os.path.join(ROOT, "examples", "example_dead_cells.cfg")
    h = host.HostSim(path, engine="host", nCells=str(n), nDead=str(nd))   # (both sides place the blob first: same stream position)
    P = orc.load_cfg(path, nCells=n, nDead=nd)
    o = orc.Sim(P, reset=True)
    dead = h.draw_dead()
    o.update()
    assert dead.sum() == nd
    assert_bit_equal(dead, o.get("dead"), f"dead set of {nd} among {n}")


@pytest.mark.parametrize("cfg,n,seed", [("example_dead_cells.cfg", 30000, 31), ("example_object_transport.cfg", 12001, 32),
                                        ("example_dead_cells.cfg", 45000, 11), ("example_object_transport.cfg", 40001, 5)])
def test_placement_with_crowded_sector_shortcut_matches_oracle(host, orc, cfg, n, seed):
    """placeRandom skips the trigonometry and the neighbourhood scan for draws into directions that are provably
    crowded (PlacementGrid::RimMask: per placed disc and ring radius, 64 sectors marked as discs are added; rings
    beyond the tracked ones: PlacementGrid::ringCovered) -- 99 % of the draws at 10^5 bots -- while consuming the
    same rand() draws; the oracle runs the reference's literal loop.  Same blob, bit for bit, at sizes where the
    ring has widened many times (the rejection counter passes 200 every few bots) and, from 40 000 bots, with the
    third ring tracked as well."""
    path = os.path.join(ROOT, "examples", cfg)
    h = host.HostSim(path, engine="host", nCells=str(n), seed=str(seed))
    o = orc.Sim(orc.load_cfg(path, nCells=n, seed=seed), reset=True)
    assert_bit_equal(h.get("pos"), o.get("pos"), f"{cfg} at {n} bots: placement")


@pytest.mark.parametrize("min_radius,max_radius,n,seed", [(1.0, 1.5, 1200, 8), (1.0, 1.5, 1500, 77), (0.7, 1.0, 2500, 77),
                                                          (0.2, 0.3, 4000, 77), (0.12, 0.1, 300, 15838),
                                                          (0.12, 0.1, 3000, 15838)])
def test_placement_in_a_rescaled_arena_matches_oracle(host, orc, min_radius, max_radius, n, seed):
    """Big discs: once 2 (ring + limit) exceeds 5, an anchor's ring reaches the seed disc, which sits
    at (5,0) but is FILED under the origin's cell (particlebot.cpp:635-637), where the reference's 3x3 crowded
    test only finds it for candidates near the origin.  ringCovered must not count it as cover elsewhere.
    (The first case is one found by search in which counting it changed the blob from bot 52 on.)
    The last two have discs WIDER than the grid's cells (min_radius > max_radius): the 3x3 scan then misses
    blockers two cells away, so nothing may be inferred from a disc's neighbourhood (round 4 found the earlier
    shortcut wrong there: bot 300-blob of seed 15838)."""
    path = os.path.join(ROOT, "examples", "example.cfg")
    kw = dict(nCells=n, seed=seed, min_radius=min_radius, max_radius=max_radius)
    h = host.HostSim(path, engine="host", **{k: str(v) for k, v in kw.items()})
    o = orc.Sim(orc.load_cfg(path, **kw), reset=True)
    assert_bit_equal(h.get("pos"), o.get("pos"), f"min_radius {min_radius}: placement")


def test_every_mask_decision_agrees_with_the_slow_test(host, capfd, monkeypatch):
    """PB_PLACEMENT_SELFCHECK=1: every draw placeRandom rejects from a crowded-sector mask is also put through the
    reference's own test (cosf, sinf, 3x3 scan, powf length) and the process aborts on the first disagreement:
    millions of direct checks of "sufficient, never necessary", next to the blob-level comparisons above."""
    monkeypatch.setenv("PB_PLACEMENT_SELFCHECK", "1")
    total = 0
    for cfg, kw in (("example_dead_cells.cfg", dict(nCells=45000, nDead=0, seed=3)),
                    ("example_object_transport.cfg", dict(nCells=12001, seed=8)),
                    ("example.cfg", dict(nCells=2500, min_radius=1.0, max_radius=1.5, seed=77)),
                    ("example.cfg", dict(nCells=4000, min_radius=0.2, max_radius=0.3, seed=5))):
        host.HostSim(os.path.join(ROOT, "examples", cfg), engine="host", **{k: str(v) for k, v in kw.items()})
        err = capfd.readouterr().err
        line = [x for x in err.splitlines() if "mask decisions verified" in x]
        assert line and "NOT crowded" not in err, err[-300:]
        total += int(line[-1].split(":")[1].split()[0])
    assert total > 5_000_000


def test_random_parameter_draws_match_oracle(host, orc):
    """Sizes x seeds x radii (incl. discs wider than the grid's cells, where nothing may be inferred from a disc's
    neighbourhood) x payload runs, drawn at random with a fixed seed; tests/diag/placement_fuzz.py is the long form."""
    rng = np.random.default_rng(4)
    ran = 0
    for _ in range(90):
        rmin = float(np.round(10 ** rng.uniform(-1.7, 0.0), 4))
        rmax = float(np.round(rmin * rng.choice([0.8, 1.0, 1.2, 1.5, 2.0, 3.0]), 4))
        n = int(rng.choice([2, 3, 4, 7, 30, 200, 900, 2500]))
        if 2.2 * rmin * np.sqrt(n) + 6 > min(0.45 * 512 * 2 * rmax, 60):
            continue   # (the blob would leave the reference's 512-cell grid, which it indexes out of bounds)
        payload = rng.random() < 0.25 and n > 3
        kw = dict(nCells=n, seed=int(rng.integers(0, 2 ** 31 - 1)), min_radius=rmin, max_radius=rmax,
                  nDead=-1 if payload else 0)
        if payload:
            kw["radFactor"] = float(rng.choice([1.0, 2.0, 5.0]))
        path = os.path.join(ROOT, "examples", "example_object_transport.cfg" if payload else "example.cfg")
        h = host.HostSim(path, engine="host", **{k: str(v) for k, v in kw.items()})
        o = orc.Sim(orc.load_cfg(path, **kw), reset=True)
        assert_bit_equal(h.get("pos"), o.get("pos"), f"{kw}")
        o.close()
        ran += 1
    assert ran >= 50


def test_large_placement_matches_oracle(host, orc):
    """10^4 bots (BASELINE config 2b's scale): the accept/reject loop stays in lock-step."""
    path = os.path.join(ROOT, "examples", "example_dead_cells.cfg")
    h = host.HostSim(path, engine="host", nCells="10000", nDead="2000")
    o = orc.Sim(orc.load_cfg(path, nCells=10000, nDead=2000))
    assert_bit_equal(h.get("pos"), o.get("pos"), "pos")


@pytest.mark.parametrize("name,touching", [("blob", 19), ("blob_upleft", 19), ("lighttest7", 19)])
def test_ten_bot_presets(host, name, touching):
    """particlebot.cpp:492-611: ten discs on a triangular lattice of pitch 2 r_min, all distinct,
    none overlapping, forming one connected cluster."""
    h = host.HostSim(None, engine="host", nCells="10", nDead="0", seed="1", pb_placement=name)
    pos = h.get("pos").astype(np.float64)
    d = np.linalg.norm(pos[:, None] - pos[None], axis=-1)
    iu = np.triu_indices(10, 1)
    assert d[iu].min() > 2 * 0.0775 - 1e-6
    adj = (d < 2 * 0.0775 + 1e-5) & (d > 0)
    seen, todo = {0}, [0]
    while todo:
        for j in np.flatnonzero(adj[todo.pop()]):
            if j not in seen:
                seen.add(int(j)); todo.append(int(j))
    assert len(seen) == 10
    assert np.all(h.get("rad") == np.float32(0.0775))


def test_square_lattice_placement(host):
    h = host.HostSim(None, engine="host", nCells="10000", nDead="0", seed="1", pb_placement="square")
    pos = h.get("pos")
    assert abs(pos.mean(0)).max() < 1e-4
    assert np.allclose(np.diff(pos[:100, 0]), 0.155, atol=1e-6) and np.all(pos[:100, 1] == pos[0, 1])


def test_headless_frame(tmp_path):
    """Particlebot::writeFramePPM (stand-in for the reference's display/video path): a P6 image in
    which every bot is a disc of its radius in updateCol_k's colour, x mirrored as the reference
    draws it, the light a yellow disc, dead bots black."""
    from particlerobotsimulations_amd import host
    sim = host.HostSim(os.path.join(ROOT, "examples", "example_dead_cells.cfg"), engine="host",
                       time_to_dead="0")
    sim.draw_dead()
    path = str(tmp_path / "f.ppm")
    size, half = 600, 4.0
    cx, cy = 5.0, 0.0
    sim.write_frame(path, size=size, center=(cx, cy), half_extent=half)
    raw = open(path, "rb").read()
    header = f"P6\n{size} {size}\n255\n".encode()
    assert raw.startswith(header) and len(raw) == len(header) + size * size * 3
    img = np.frombuffer(raw[len(header):], np.uint8).reshape(size, size, 3)
    pos, rad, dead = sim.get("pos"), sim.get("rad"), sim.get("dead")
    scale = 0.5 * size / half
    hit_live = hit_dead = 0
    for (x, y), r, d in zip(pos, rad, dead):
        px, py = int(0.5 * size - (x - cx) * scale), int(0.5 * size - (y - cy) * scale)
        if not (0 <= px < size and 0 <= py < size):
            continue
        c = img[py, px]
        if d:
            hit_dead += int((c == 0).all())
        else:  # all radii are min_radius after reset: G = 20 + 180, B = 30
            hit_live += int(tuple(c) == (30, 200, 30))
    assert hit_dead == int(dead.sum()) == 20 and hit_live == len(rad) - 20
    # bot area in pixels ~ sum(pi r^2) * scale^2 (bots touch but do not overlap after placement)
    botpix = int(((img != 245).any(axis=2)).sum())
    expect = float((np.pi * rad.astype(np.float64) ** 2).sum()) * scale * scale
    assert 0.9 * expect <= botpix <= 1.15 * expect + np.pi * (0.25 * scale) ** 2


@pytest.mark.parametrize("n", [1, 2, 3, 4, 7])
def test_tiny_placements_match_oracle(host, orc, n):
    """The first bots are special-cased by the reference (seed bot at (5,0), bot 1 beside it, bot 2
    perpendicular to that pair, particlebot.cpp:612-660): the smallest blobs, bit for bit."""
    path = os.path.join(ROOT, "examples", "example.cfg")
    h = host.HostSim(path, engine="host", nCells=str(n))
    P = orc.load_cfg(path)
    P.nCells = n
    o = orc.Sim(P)
    for k in ("pos", "vel", "rad", "phase"):
        assert_bit_equal(h.get(k), o.get(k), f"n={n} {k}")


# ---- pb_placement fastblob: the O(N) random-blob generator ---------------------

def _blob_stats(pos, origin=-64.0, cell=0.235):
    """contacts per bot (pairs at 2 r_min), radius of gyration, mean candidate pairs in the 5x5 stencil"""
    from collections import Counter

    from scipy.spatial import cKDTree
    pos = pos.astype(np.float64)
    n = len(pos)
    tree = cKDTree(pos)
    contacts = 2.0 * len(tree.query_pairs(2 * 0.0775 * 1.001)) / n
    overlaps = len(tree.query_pairs(2 * 0.0775 * 0.999))
    rg = float(np.sqrt(((pos - pos.mean(0)) ** 2).sum(1).mean()))
    cx = np.floor((pos[:, 0] - origin) / cell).astype(int)
    cy = np.floor((pos[:, 1] - origin) / cell).astype(int)
    cnt = Counter(zip(cx.tolist(), cy.tolist()))
    tot = 0
    for (x, y), k in cnt.items():
        tot += k * (sum(cnt.get((x + dx, y + dy), 0) for dx in range(-2, 3) for dy in range(-2, 3)) - 1)
    return contacts, rg, tot / n, overlaps


def test_fastblob_matches_reference_rule_statistics(host):
    """At 10^4 bots the fast generator's blobs are the reference rule's blobs statistically: contacts
    per bot, radius of gyration and the force kernel's candidate pairs per bot (42 +- 2), seed by seed
    within the seed-to-seed spread x a small factor."""
    ex = os.path.join(ROOT, "examples", "example.cfg")
    ref, fast = [], []
    for seed in (11, 12, 13):
        a = host.HostSim(ex, engine="host", nCells="10000", seed=str(seed))
        b = host.HostSim(ex, engine="host", nCells="10000", seed=str(seed), pb_placement="fastblob")
        ref.append(_blob_stats(a.get("pos")))
        fast.append(_blob_stats(b.get("pos")))
        assert not np.array_equal(a.get("pos"), b.get("pos"))  # a different blob of the same kind
    ref, fast = np.array(ref), np.array(fast)
    assert abs(fast[:, 0].mean() - ref[:, 0].mean()) < 0.05 * ref[:, 0].mean(), (ref, fast)   # contacts per bot
    assert abs(fast[:, 1].mean() - ref[:, 1].mean()) < 0.02 * ref[:, 1].mean(), (ref, fast)   # radius of gyration
    assert abs(ref[:, 2].mean() - 42.0) < 2.0 and abs(fast[:, 2].mean() - 42.0) < 2.0, (ref, fast)
    assert abs(fast[:, 2].mean() - ref[:, 2].mean()) < 1.5
    # no real overlaps (the reference's own blob has the bot-2 / bot-0 quirks: a handful)
    assert fast[:, 3].max() <= 3 and ref[:, 3].max() <= 3


def test_fastblob_is_deterministic_linear_and_handles_the_payload(host):
    import time
    ex = os.path.join(ROOT, "examples", "example_object_transport.cfg")
    a = host.HostSim(ex, engine="host", pb_placement="fastblob")
    b = host.HostSim(ex, engine="host", pb_placement="fastblob")
    assert_bit_equal(a.get("pos"), b.get("pos"), "same seed, same blob")
    pos = a.get("pos")
    assert pos[-1, 1] == 0.0 and pos[-1, 0] < pos[:-1, 0].min()  # payload left of the blob (particlebot.cpp:731-735)
    # roughly linear in N: 10x the bots well within 40x the time even on a noisy box (the reference rule: ~32x on a quiet one)
    big = dict(pb_grid_size="2048", pb_arena_half="240")
    t = []
    for n in (20000, 200000):
        t0 = time.perf_counter()
        s = host.HostSim(os.path.join(ROOT, "examples", "example.cfg"), engine="host", nCells=str(n),
                         pb_placement="fastblob", **big)
        t.append(time.perf_counter() - t0)
        p = s.get("pos")
        assert np.isfinite(p).all() and np.abs(p).max() < 240.0
    assert t[1] < 40 * t[0] + 1.0, t   # (generous: the point is "not quadratic", on a possibly noisy host)
