"""ctypes binding of the CPU oracle (oracle/libpb_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (particlerobotsimulations_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpb_oracle.so")

ORC_MAX_OBS = 10


class OrcParams(C.Structure):
    _fields_ = [
        ("gridSizeX", C.c_uint32), ("gridSizeY", C.c_uint32), ("numCells", C.c_uint32),
        ("worldOriginX", C.c_float), ("worldOriginY", C.c_float),
        ("cellSizeX", C.c_float), ("cellSizeY", C.c_float),
        ("nCells", C.c_uint32), ("nDead", C.c_int32),
        ("gravity", C.c_float), ("spring", C.c_float), ("damping", C.c_float), ("shear", C.c_float),
        ("attraction", C.c_float), ("boundaryDamping", C.c_float), ("friction", C.c_float),
        ("massFactor", C.c_float), ("frictionFactor", C.c_float), ("radFactor", C.c_float),
        ("attractionFactor", C.c_float),
        ("constraint", C.c_float), ("constraint_contraction", C.c_float),
        ("centroid_steps", C.c_int32), ("centroid_int", C.c_float), ("centroid_radius", C.c_float),
        ("light_x", C.c_float), ("light_y", C.c_float), ("phase_update_interval", C.c_float),
        ("control", C.c_int32), ("config", C.c_int32),
        ("min_radius", C.c_float), ("max_radius", C.c_float), ("rise_period", C.c_float),
        ("freq", C.c_float),
        ("nobstacles", C.c_int32),
        ("x1obs", C.c_float * ORC_MAX_OBS), ("x2obs", C.c_float * ORC_MAX_OBS),
        ("y1obs", C.c_float * ORC_MAX_OBS), ("y2obs", C.c_float * ORC_MAX_OBS),
        ("n_cir_obstacles", C.c_int32),
        ("x_cir_obs", C.c_float * ORC_MAX_OBS), ("y_cir_obs", C.c_float * ORC_MAX_OBS),
        ("r_cir_obs", C.c_float * ORC_MAX_OBS),
        ("Nx", C.c_int32), ("phase_std", C.c_float), ("seed", C.c_uint32),
        ("light_shadow", C.c_uint32), ("testing", C.c_uint32),
        ("constrained_contraction", C.c_uint32), ("display_shadow", C.c_uint32),
        ("time_to_dead", C.c_float), ("max_time", C.c_float),
        ("timestep", C.c_float), ("sort_interval", C.c_float), ("dump_interval", C.c_float),
        ("camera_x", C.c_float), ("camera_y", C.c_float), ("light_radius", C.c_float),
        ("display_interval", C.c_int32), ("video_interval", C.c_int32),
        ("csv_filename", C.c_char * 300), ("video_filename", C.c_char * 300),
        ("wallHalf", C.c_float),
        ("rngKind", C.c_int32),
    ]

    def to_dict(self):
        out = {}
        for name, typ in self._fields_:
            v = getattr(self, name)
            if hasattr(v, "__len__") and not isinstance(v, (bytes, str)):
                v = [float(x) for x in v]
            elif isinstance(v, bytes):
                v = v.decode()
            out[name] = v
        return out


def build(force=False):
    """Compile oracle/libpb_oracle.so with the committed Makefile (gcc, -ffp-contract=off)."""
    src = os.path.join(_HERE, "pb_oracle.c")
    hdr = os.path.join(_HERE, "pb_oracle.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libpb_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _SO


def usable_cpus():
    """CPUs this process may really use: scheduler affinity, capped by the cgroup CPU quota.
    (An OpenMP team larger than that spins at every barrier and runs orders of magnitude slower.)"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except Exception:
            pass
    return max(1, n)


_lib = None
_variants = {}

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_PP = C.POINTER(OrcParams)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    _lib = _bind(_SO)
    return _lib


BRACKET_VARIANTS = ("fma", "fma_powf")
ORDER_VARIANT = "order"  # exact terms, the two-pass kernel's order of additions
# the device powf sites of the obstacle / shadow tests (impl.cuh:214-229, 704-779): a power through base-2
# logarithms; the correctly rounded result nudged by -1 / 0 / +1 ulp; FMA + __powf + device powf all at once
DEVPOWF_VARIANTS = ("devpowf", "devpowf_ulp", "cuda_like")
_VARIANT_TAGS = {"fma": b"fma", "fma_powf": b"fma+powf", "order": b"order", "devpowf": b"devpowf",
                 "devpowf_ulp": b"devpowf_ulp", "cuda_like": b"fma+powf+devpowf"}


def variant_lib(name):
    """A BRACKET build of the same source (oracle/Makefile): "fma" = kernel functions contracted
    into FMAs as nvcc's default -fmad=true does, "fma_powf" = that plus exp2f(2*log2f(x)) at the two
    __powf sites.  These are not the oracle: they measure how far a legitimately different build
    of the reference's arithmetic drifts from it (tests/test_fma_bracket.py)."""
    if name in (None, "exact"):
        return lib()
    if name not in _VARIANT_TAGS:
        raise ValueError(name)
    if name not in _variants:
        so = os.path.join(_HERE, f"libpb_oracle_{name}.so")
        src = os.path.join(_HERE, "pb_oracle.c")
        if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
            subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(so)], stdout=subprocess.DEVNULL)
        L = _bind(so)
        want = _VARIANT_TAGS[name]
        if L.orc_build_variant() != want:
            raise RuntimeError(f"{so} reports build variant {L.orc_build_variant()!r}, expected {want!r}")
        _variants[name] = L
    return _variants[name]


def _bind(so):
    L = C.CDLL(so)
    L.orc_build_variant.restype = C.c_char_p
    L.orc_params_defaults.argtypes = [_PP]
    L.orc_set_param.argtypes = [_PP, C.c_char_p, C.c_char_p]
    L.orc_load_cfg.argtypes = [_PP, C.c_char_p]
    L.orc_load_cfg.restype = C.c_int
    L.orc_params_derive.argtypes = [_PP, C.c_uint32, C.c_float]
    L.orc_integrateSystem.argtypes = [_PP, _f32p, _f32p, _f32p, C.c_float, C.c_uint32]
    L.orc_calcHash.argtypes = [_PP, _u32p, _u32p, _f32p, C.c_uint32]
    L.orc_sortParticlebots.argtypes = [_u32p, _u32p, C.c_uint32]
    L.orc_reorderDataAndFindCellStart.argtypes = [_PP, _u32p, _u32p, _f32p, _f32p, _f32p, _u32p,
                                                  _u32p, _f32p, _f32p, _f32p, C.c_uint32, C.c_uint32]
    L.orc_updateRad_light_wave.argtypes = [_PP, _f32p, _f32p, _f32p, _f32p, C.c_float, C.c_float,
                                           _i32p, C.c_uint32]
    L.orc_updatePhase.argtypes = [_PP, _f32p, _f32p, C.c_float, C.c_float, C.c_float, C.c_uint32]
    L.orc_collide.argtypes = [_PP, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _u32p, _u32p, _u32p,
                              C.c_uint32, C.c_float]
    L.orc_collideSpheres.argtypes = [_PP, _f32p, _f32p, _f32p, _f32p, C.c_float, C.c_float,
                                     C.c_float, _f32p, _f32p, _f32p]
    L.orc_minmax_light_distance.argtypes = [_PP, _f32p, C.c_uint32, _f32p, _f32p]
    L.orc_normal.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_normal.restype = C.c_float
    L.orc_add_normal_noise.argtypes = [C.c_uint32, C.c_uint32, _f32p, C.c_float, C.c_uint32]
    L.orc_xorwow_outputs.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, _u32p]
    L.orc_xorwow_normals.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, _f32p]
    L.orc_xorwow_jump_rows.argtypes = [_u32p]
    L.orc_sim_create.argtypes = [_PP]
    L.orc_sim_create.restype = C.c_void_p
    L.orc_sim_destroy.argtypes = [C.c_void_p]
    L.orc_sim_reset.argtypes = [C.c_void_p, C.c_int]
    L.orc_sim_update.argtypes = [C.c_void_p, C.c_float, C.c_float]
    L.orc_sim_update.restype = C.c_int
    L.orc_sim_force_sort_once.argtypes = [C.c_void_p]
    L.orc_sim_time.argtypes = [C.c_void_p]
    L.orc_sim_time.restype = C.c_float
    L.orc_sim_set_time.argtypes = [C.c_void_p, C.c_float]
    L.orc_sim_phase_draws.argtypes = [C.c_void_p]
    L.orc_sim_phase_draws.restype = C.c_uint32
    L.orc_sim_array.argtypes = [C.c_void_p, C.c_int]
    L.orc_sim_array.restype = C.c_void_p
    L.orc_num_threads.restype = C.c_int
    L.orc_set_num_threads.argtypes = [C.c_int]
    # FILE*-taking entry points are driven through the libc handle below
    L.orc_sim_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_uint32, C.c_int]
    L.orc_sim_dump.restype = C.c_int
    L.orc_sim_load_from_file.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_sim_load_from_file.restype = C.c_int
    L.orc_set_num_threads(usable_cpus())
    return L


_libc = C.CDLL(None)
_libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
_libc.fopen.restype = C.c_void_p
_libc.fclose.argtypes = [C.c_void_p]


def default_params(**over):
    """main.cpp defaults, then overrides (python values), then the derived grid parameters."""
    P = OrcParams()
    lib().orc_params_defaults(C.byref(P))
    grid = over.pop("grid", 0)
    arena_half = over.pop("arena_half", 0.0)
    for k, v in over.items():
        if isinstance(v, (list, tuple, np.ndarray)):
            arr = getattr(P, k)
            for i, x in enumerate(v):
                arr[i] = float(x)
        else:
            setattr(P, k, v)
    lib().orc_params_derive(C.byref(P), int(grid), float(arena_half))
    return P


def load_cfg(path, grid=0, arena_half=0.0, **over):
    P = OrcParams()
    L = lib()
    L.orc_params_defaults(C.byref(P))
    if L.orc_load_cfg(C.byref(P), os.fsencode(path)) != 0:
        raise FileNotFoundError(path)
    for k, v in over.items():
        setattr(P, k, v)
    L.orc_params_derive(C.byref(P), int(grid), float(arena_half))
    return P


class Sim:
    """Whole-simulation oracle object (restates class Particlebot)."""

    _ARR = {"pos": (0, np.float32, 2), "vel": (1, np.float32, 2), "rad": (2, np.float32, 1),
            "phase": (3, np.float32, 1), "absForce_a": (4, np.float32, 1),
            "absForce_r": (5, np.float32, 1), "dead": (6, np.int32, 1),
            "hash": (7, np.uint32, 1), "index": (8, np.uint32, 1)}

    def __init__(self, P, reset=True, hex=False, variant=None):
        self.P = P
        self.n = int(P.nCells)
        self._L = variant_lib(variant)  # None: the oracle; "fma"/"fma_powf": a bracket build
        self._h = self._L.orc_sim_create(C.byref(P))
        if reset:
            self._L.orc_sim_reset(self._h, 1 if hex else 0)

    def close(self):
        if self._h:
            self._L.orc_sim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def view(self, name):
        """numpy VIEW (no copy) of an internal array, original index order."""
        which, dt, w = self._ARR[name]
        ptr = self._L.orc_sim_array(self._h, which)
        buf = (C.c_byte * (self.n * w * 4)).from_address(ptr)
        a = np.frombuffer(buf, dtype=dt)
        return a.reshape(self.n, w) if w > 1 else a

    def get(self, name):
        return self.view(name).copy()

    def set(self, name, value):
        self.view(name)[...] = np.asarray(value).reshape(self.view(name).shape)

    @property
    def time(self):
        return self._L.orc_sim_time(self._h)

    @time.setter
    def time(self, t):
        self._L.orc_sim_set_time(self._h, float(t))

    def force_sort_once(self):
        self._L.orc_sim_force_sort_once(self._h)

    def update(self, dt=None, sort_interval=None):
        dt = self.P.timestep if dt is None else dt
        si = self.P.sort_interval if sort_interval is None else sort_interval
        return self._L.orc_sim_update(self._h, dt, si)

    def run(self, steps, dt=None, sort_interval=None):
        for _ in range(steps):
            if self.update(dt, sort_interval):
                return False
        return True

    def dump(self, path_or_none, dump_interval=None, testing=None, mode="a"):
        di = self.P.dump_interval if dump_interval is None else dump_interval
        tt = self.P.testing if testing is None else testing
        fp = None
        if path_or_none is not None:
            fp = _libc.fopen(os.fsencode(path_or_none), mode.encode())
        try:
            return self._L.orc_sim_dump(self._h, fp, di, tt, 0)
        finally:
            if fp:
                _libc.fclose(fp)

    def load_from_file(self, path):
        fp = _libc.fopen(os.fsencode(path), b"r")
        if not fp:
            raise FileNotFoundError(path)
        try:
            return self._L.orc_sim_load_from_file(self._h, fp)
        finally:
            _libc.fclose(fp)
