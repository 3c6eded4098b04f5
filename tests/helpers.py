"""Shared test helpers: turn an oracle parameter block into the product's SimParams, build
randomised-but-physical states, compare arrays."""
import numpy as np

LIST_FIELDS = {"x1obs": "nobstacles", "x2obs": "nobstacles", "y1obs": "nobstacles", "y2obs": "nobstacles",
               "x_cir_obs": "n_cir_obstacles", "y_cir_obs": "n_cir_obstacles", "r_cir_obs": "n_cir_obstacles"}

SCALARS = ["nCells", "nDead", "gravity", "spring", "damping", "shear", "attraction", "boundaryDamping",
           "friction", "massFactor", "frictionFactor", "radFactor", "attractionFactor", "constraint",
           "constraint_contraction", "centroid_steps", "centroid_int", "centroid_radius", "light_x", "light_y",
           "phase_update_interval", "control", "config", "min_radius", "max_radius", "rise_period", "freq",
           "nobstacles", "n_cir_obstacles", "Nx", "phase_std", "seed", "light_shadow", "testing",
           "constrained_contraction", "display_shadow", "time_to_dead", "max_time", "numCells"]


def simparams_from_orc(P):
    """(SimParams, keepalive) carrying exactly the values of an oracle OrcParams."""
    from particlerobotsimulations_amd import make_params
    d = {k: getattr(P, k) for k in SCALARS}
    d["gridSize"] = (P.gridSizeX, P.gridSizeY)
    d["worldOrigin"] = (P.worldOriginX, P.worldOriginY)
    d["cellSize"] = (P.cellSizeX, P.cellSizeY)
    for f, cnt in LIST_FIELDS.items():
        d[f] = [getattr(P, f)[i] for i in range(getattr(P, cnt))]
    return make_params(d)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_bit_equal(a, b, what=""):
    if a is None:
        # absForce_a of a batch in which nothing reads it (pbSimSetForceSums mode 0): not maintained.
        # tests/test_gpu_dead_sum.py compares it in mode 1 and everything else in both modes.
        assert "absForce_a" in what, what
        return
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if not np.array_equal(bits(a), bits(b)):
        bad = np.flatnonzero(bits(a).reshape(-1) != bits(b).reshape(-1))
        i = bad[0]
        raise AssertionError(f"{what}: {bad.size}/{a.size} elements differ; first at flat index {i}: "
                             f"{a.reshape(-1)[i]!r} vs {b.reshape(-1)[i]!r}")


def max_rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-30)
    return float(np.max(np.abs(a - b) / scale)) if a.size else 0.0


def jittered_blob(n, spacing, rng, center=(0.0, 0.0), jitter=0.15, rmin=0.0775, rmax=0.1175):
    """n bots on a jittered hexagonal patch with random radii and small random velocities: a
    physical-looking state with contacts, near-contacts (all three attraction regimes) and far pairs."""
    side = int(np.ceil(np.sqrt(n))) + 1
    pts = []
    for j in range(side):
        for i in range(side):
            pts.append((i * spacing + (j % 2) * spacing * 0.5, j * spacing * 0.8660254))
    pts = np.array(pts[:n], dtype=np.float64)
    pts -= pts.mean(0)
    pts += np.asarray(center)
    pts += rng.uniform(-jitter, jitter, size=pts.shape) * spacing
    pos = pts.astype(np.float32)
    rad = rng.uniform(rmin, rmax, size=n).astype(np.float32)
    vel = (rng.standard_normal((n, 2)) * 0.02).astype(np.float32)
    vel[rng.random(n) < 0.2] = 0.0  # some bots at rest: exercises the static-friction hold
    return pos, vel, rad
