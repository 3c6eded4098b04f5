"""ctypes binding of libparticlebot_host.so: the C++ host side (class Particlebot, the .cfg loader)
behind a handful of C wrappers (csrc/pb_capi.cpp)."""
import ctypes as C
import os

import numpy as np

from . import _capi

PB_MAX_OBSTACLES = 10


class FlatConfig(C.Structure):
    """pbFlatConfig in csrc/pb_capi.cpp: a resolved configuration without pointers."""
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
        ("x1obs", C.c_float * PB_MAX_OBSTACLES), ("x2obs", C.c_float * PB_MAX_OBSTACLES),
        ("y1obs", C.c_float * PB_MAX_OBSTACLES), ("y2obs", C.c_float * PB_MAX_OBSTACLES),
        ("n_cir_obstacles", C.c_int32),
        ("x_cir_obs", C.c_float * PB_MAX_OBSTACLES), ("y_cir_obs", C.c_float * PB_MAX_OBSTACLES),
        ("r_cir_obs", C.c_float * PB_MAX_OBSTACLES),
        ("Nx", C.c_int32), ("phase_std", C.c_float), ("seed", C.c_uint32),
        ("light_shadow", C.c_uint32), ("testing", C.c_uint32),
        ("constrained_contraction", C.c_uint32), ("display_shadow", C.c_uint32),
        ("time_to_dead", C.c_float), ("max_time", C.c_float),
        ("timestep", C.c_float), ("sort_interval", C.c_float), ("dump_interval", C.c_float),
        ("camera_x", C.c_float), ("camera_y", C.c_float), ("light_radius", C.c_float),
        ("display_interval", C.c_int32), ("video_interval", C.c_int32),
        ("csv_filename", C.c_char * 300), ("video_filename", C.c_char * 300),
        ("wallHalf", C.c_float), ("rngKind", C.c_int32), ("forceVariant", C.c_int32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    _capi.lib()  # libparticlebot_hip.so first (RTLD_GLOBAL), the host library links against it
    if not os.path.exists(_capi.HOST_SO):
        raise RuntimeError(f"{_capi.HOST_SO} not found: run __graft_entry__.build()")
    L = C.CDLL(_capi.HOST_SO)
    L.pbHostLoadConfig.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(FlatConfig)]
    L.pbHostLoadConfig.restype = C.c_int
    L.pbHostCreate.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.pbHostCreate.restype = C.c_void_p
    L.pbHostDestroy.argtypes = [C.c_void_p]
    L.pbHostReset.argtypes = [C.c_void_p]
    L.pbHostUpdate.argtypes = [C.c_void_p]
    L.pbHostAdvance.argtypes = [C.c_void_p, C.c_int]
    L.pbHostAdvance.restype = C.c_int
    L.pbHostStepsUntilDump.argtypes = [C.c_void_p, C.c_int]
    L.pbHostStepsUntilDump.restype = C.c_int
    L.pbHostTime.argtypes = [C.c_void_p]
    L.pbHostTime.restype = C.c_float
    L.pbHostFinished.argtypes = [C.c_void_p]
    L.pbHostFinished.restype = C.c_int
    L.pbHostDump.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.pbHostDump.restype = C.c_int
    L.pbHostLoadFromFile.argtypes = [C.c_void_p, C.c_char_p]
    L.pbHostLoadFromFile.restype = C.c_int
    L.pbHostDrawDead.argtypes = [C.c_void_p, C.c_void_p]
    L.pbHostSaveCheckpoint.argtypes = [C.c_void_p, C.c_char_p]
    L.pbHostSaveCheckpoint.restype = C.c_int
    L.pbHostLoadCheckpoint.argtypes = [C.c_void_p, C.c_char_p]
    L.pbHostLoadCheckpoint.restype = C.c_int
    L.pbHostGetArray.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.pbHostGetArray.restype = C.c_int
    L.pbHostSetArray.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.pbHostSetArray.restype = C.c_int
    L.pbHostLibcRandDraws.argtypes = [C.c_uint, C.c_int, C.c_void_p]
    L.pbHostNumBots.argtypes = [C.c_void_p]
    L.pbHostNumBots.restype = C.c_uint
    _lib = L
    return L


def libc_rand_draws(seed, n):
    """n outputs of the class's private glibc-compatible generator after seeding with `seed`."""
    out = np.empty(n, np.int32)
    lib().pbHostLibcRandDraws(int(seed), int(n), out.ctypes.data_as(C.c_void_p))
    return out


def _overrides(over):
    if not over:
        return None
    return "\n".join(f"{k}\n{v}" for k, v in over.items()).encode()


def load_config(cfg_path=None, **over):
    """Resolve a .cfg exactly as the runner does; returns a FlatConfig."""
    out = FlatConfig()
    rc = lib().pbHostLoadConfig(os.fsencode(cfg_path) if cfg_path else None, _overrides(over), C.byref(out))
    if rc != 0:
        raise FileNotFoundError(cfg_path)
    return out


class HostSim:
    """class Particlebot driven from Python (engine: 'fused' or 'legacy')."""

    def __init__(self, cfg_path=None, engine="fused", reset=True, **over):
        self._h = lib().pbHostCreate(os.fsencode(cfg_path) if cfg_path else None, _overrides(over),
                                     {"fused": 0, "legacy": 1, "host": 2}[engine])
        if not self._h:
            raise FileNotFoundError(cfg_path)
        self.n = lib().pbHostNumBots(self._h)
        if reset:
            lib().pbHostReset(self._h)

    def close(self):
        if getattr(self, "_h", None):
            lib().pbHostDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self):
        lib().pbHostUpdate(self._h)

    def advance(self, nsteps):
        return lib().pbHostAdvance(self._h, int(nsteps))

    def steps_until_dump(self, max_steps=1 << 20):
        return lib().pbHostStepsUntilDump(self._h, int(max_steps))

    @property
    def time(self):
        return lib().pbHostTime(self._h)

    @property
    def finished(self):
        return bool(lib().pbHostFinished(self._h))

    def dump(self, path, mode="a"):
        if lib().pbHostDump(self._h, os.fsencode(path), mode.encode()) != 0:
            raise OSError(path)

    def load_from_file(self, path):
        if lib().pbHostLoadFromFile(self._h, os.fsencode(path)) != 0:
            raise OSError(path)

    def draw_dead(self):
        out = np.empty(self.n, np.int32)
        lib().pbHostDrawDead(self._h, out.ctypes.data_as(C.c_void_p))
        return out

    def write_frame(self, path, size=800, center=(0.0, 0.0), half_extent=0.0):
        """Binary PPM of the arena seen from above (Particlebot::writeFramePPM).  half_extent <= 0:
        the reference's camera, centred on (camera_x, 0), half extent camera_y * tan(30 deg)."""
        L = lib()
        L.pbHostWriteFrame.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
        if L.pbHostWriteFrame(self._h, os.fsencode(path), int(size), int(size), center[0], center[1],
                              half_extent) != 0:
            raise OSError(f"writeFramePPM({path}) failed")

    def save_checkpoint(self, path):
        rc = lib().pbHostSaveCheckpoint(self._h, os.fsencode(path))
        if rc != 0:
            raise OSError(f"saveCheckpoint({path}) failed ({rc})")

    def load_checkpoint(self, path):
        rc = lib().pbHostLoadCheckpoint(self._h, os.fsencode(path))
        if rc != 0:
            raise OSError(f"loadCheckpoint({path}) failed ({rc})")

    def get(self, name):
        which, dt, w = {"pos": (0, np.float32, 2), "vel": (1, np.float32, 2), "rad": (2, np.float32, 1),
                        "phase": (3, np.float32, 1), "dead": (5, np.int32, 1)}[name]
        out = np.empty((self.n, w) if w > 1 else self.n, dtype=dt)
        assert lib().pbHostGetArray(self._h, which, out.ctypes.data_as(C.c_void_p)) == 0
        return out

    def set(self, name, data, start=0):
        which = {"pos": 0, "vel": 1, "rad": 2, "phase": 3}[name]
        data = np.ascontiguousarray(data, np.float32)
        count = data.shape[0]
        assert lib().pbHostSetArray(self._h, which, data.ctypes.data_as(C.c_void_p), int(start), int(count)) == 0
