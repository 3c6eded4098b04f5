"""ctypes binding of libparticlebot_hip.so (the C-ABI declared in include/particlebot_hip.h).

No fallback: if the HIP library is missing, importing the symbols raises.  Nothing here touches
oracle/.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_PKG, "lib")
HIP_SO = os.path.join(LIB_DIR, "libparticlebot_hip.so")
HOST_SO = os.path.join(LIB_DIR, "libparticlebot_host.so")

PB_MAX_OBSTACLES = 10
PB_OK = 0


class uint2(C.Structure):
    _fields_ = [("x", C.c_uint), ("y", C.c_uint)]


class float2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class SimParams(C.Structure):
    """include/particlebot_kernel.h (layout of the reference's particlebot_kernel.cuh:58-120):
    256 bytes, checked against the C++ struct by a static_assert in csrc/pb_device.hpp."""
    _fields_ = [
        ("gridSize", uint2), ("numCells", C.c_uint),
        ("_pad0", C.c_uint),  # float2 is 8-byte aligned in HIP (and CUDA): hole after numCells
        ("worldOrigin", float2), ("cellSize", float2),
        ("nCells", C.c_uint), ("nDead", C.c_int), ("maxParticlebotsPerCell", C.c_uint),
        ("gravity", C.c_float), ("spring", C.c_float), ("damping", C.c_float), ("shear", C.c_float),
        ("attraction", C.c_float), ("boundaryDamping", C.c_float), ("friction", C.c_float),
        ("massFactor", C.c_float), ("frictionFactor", C.c_float), ("radFactor", C.c_float),
        ("attractionFactor", C.c_float),
        ("constraint", C.c_float), ("constraint_contraction", C.c_float),
        ("centroid_steps", C.c_int), ("centroid_int", C.c_float), ("centroid_radius", C.c_float),
        ("light_x", C.c_float), ("light_y", C.c_float), ("phase_update_interval", C.c_float),
        ("control", C.c_int), ("config", C.c_int),
        ("min_radius", C.c_float), ("max_radius", C.c_float), ("rise_period", C.c_float),
        ("freq", C.c_float),
        ("nobstacles", C.c_int),
        ("x1obs", C.POINTER(C.c_float)), ("x2obs", C.POINTER(C.c_float)),
        ("y1obs", C.POINTER(C.c_float)), ("y2obs", C.POINTER(C.c_float)),
        ("n_cir_obstacles", C.c_int),
        ("x_cir_obs", C.POINTER(C.c_float)), ("y_cir_obs", C.POINTER(C.c_float)),
        ("r_cir_obs", C.POINTER(C.c_float)),
        ("Nx", C.c_int), ("phase_std", C.c_float), ("seed", C.c_uint),
        ("light_shadow", C.c_uint), ("testing", C.c_uint), ("constrained_contraction", C.c_uint),
        ("display_shadow", C.c_uint), ("time_to_dead", C.c_float), ("max_time", C.c_float),
    ]


class pbRngState(C.Structure):
    """48 bytes = sizeof(curandStateXORWOW) (include/particlebot_hip.h)."""
    _fields_ = [("d", C.c_uint), ("v", C.c_uint * 5), ("boxmuller_flag", C.c_int), ("kind", C.c_int),
                ("boxmuller_extra", C.c_float), ("reserved", C.c_float * 3)]


PB_RNG_COUNTER, PB_RNG_XORWOW_CURAND, PB_RNG_XORWOW_ROCRAND = 0, 1, 2


class pbSimStats(C.Structure):
    _fields_ = [("steps", C.c_ulonglong), ("fused_launches", C.c_ulonglong),
                ("plain_launches", C.c_ulonglong), ("state_launches", C.c_ulonglong),
                ("resorts", C.c_ulonglong), ("phase_updates", C.c_ulonglong),
                ("resident_launches", C.c_ulonglong)]


class pbSimConfig(C.Structure):
    _fields_ = [("force_variant", C.c_int), ("force_kind", C.c_int), ("lanes_per_bot", C.c_int),
                ("resident", C.c_int), ("fast_math_ok", C.c_int), ("payload", C.c_int), ("rng", C.c_int), ("offsets64", C.c_int),
                ("attraction_sums", C.c_int), ("dead_sum_form", C.c_int), ("stream_walk", C.c_int)]


class pbForceForm(C.Structure):
    _fields_ = [("flat", C.c_int), ("lanes_per_bot", C.c_int), ("attraction_sums", C.c_int), ("offsets64", C.c_int)]


# every symbol include/particlebot_hip.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
_F = C.c_float
_U = C.c_uint
_I = C.c_int
SYMBOLS = {
    "cudaInit": (None, [_I, _VP]),
    "cudaGLInit": (None, [_I, _VP]),
    "allocateArray": (None, [C.POINTER(_VP), C.c_size_t]),
    "freeArray": (None, [_VP]),
    "threadSync": (None, []),
    "copyArrayFromDevice": (None, [_VP, _VP, _VP, _I]),
    "copyArrayToDevice": (None, [_VP, _VP, _I, _I]),
    "registerGLBufferObject": (None, [_U, C.POINTER(_VP)]),
    "unregisterGLBufferObject": (None, [_VP]),
    "mapGLBufferObject": (_VP, [C.POINTER(_VP)]),
    "unmapGLBufferObject": (None, [_VP]),
    "pbCreateBuffer": (_U, [C.c_size_t]),
    "pbBufferSubData": (None, [_U, C.c_size_t, C.c_size_t, _VP]),
    "pbDeleteBuffer": (None, [_U]),
    "setParameters": (None, [C.POINTER(SimParams)]),
    "pbSetWallHalfExtent": (None, [_F]),
    "integrateSystem": (None, [_VP, _VP, _VP, _F, _U, _F]),
    "calcHash": (None, [_VP, _VP, _VP, _I]),
    "reorderDataAndFindCellStart": (None, [_VP] * 10 + [_U, _U]),
    "updateRad_light_wave": (None, [_VP, _VP, _VP, _VP, _VP, _F, _F, _VP, _I]),
    "curand_setup": (None, [_VP, _I]),
    "add_normal_noise": (None, [_VP, _VP, _F, _I]),
    "updatePhase": (None, [_VP, _VP, _F, _F, _F, _I]),
    "updateCol": (None, [_VP, _VP, _I, _VP, _VP, _VP]),
    "collide": (None, [_VP] * 9 + [_U, _U, _F]),
    "calcCOG": (None, [_VP, _VP, _VP, _I, _F, _I, _F]),
    "sortParticlebots": (None, [_VP, _VP, _U]),
    "pbGetLastErrorString": (C.c_char_p, []),
    "pbGetDevice": (_I, [C.POINTER(_I)]),
    "pbDevicePciBusId": (_I, [_I, C.c_char_p, _I]),
    "pbSetDevice": (_I, [_I]),
    "pbSimCreate": (_I, [C.POINTER(_VP), C.POINTER(SimParams), _F]),
    "pbSimDestroy": (None, [_VP]),
    "pbSimCreateBatch": (_I, [C.POINTER(_VP), C.POINTER(SimParams), _I, _F]),
    "pbSimBatchSize": (_I, [_VP, C.POINTER(_U), C.POINTER(_U)]),
    "pbSimSetStateOf": (_I, [_VP, _U, _VP, _VP, _VP, _VP, _VP]),
    "pbSimSetStateRangeOf": (_I, [_VP, _U, _U, _U, _VP, _VP, _VP, _VP, _VP]),
    "pbSimGetStateOf": (_I, [_VP, _U] + [_VP] * 7),
    "pbSimCentroids": (_I, [_VP, C.POINTER(C.c_double)]),
    "pbSimCentroidSums": (_I, [_VP, C.POINTER(C.c_float)]),
    "pbSimGetLayoutOf": (_I, [_VP, _U, _VP, _VP, C.POINTER(_I)]),
    "pbSimSetLayoutOf": (_I, [_VP, _U, _VP, _VP]),
    "pbSimSetForcesOf": (_I, [_VP, _U, _VP, _VP]),
    "pbSimSetState": (_I, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "pbSimGetState": (_I, [_VP] * 8),
    "pbSimSetTime": (_I, [_VP, _F]),
    "pbSimGetTime": (_I, [_VP, C.POINTER(_F)]),
    "pbSimGetPhaseDraws": (_I, [_VP, C.POINTER(_U)]),
    "pbSimSetPhaseDraws": (_I, [_VP, _U]),
    "pbSimStep": (_I, [_VP, _F, _F, _I, C.POINTER(_I)]),
    "pbSimStepTimed": (_I, [_VP, _F, _F, _I, C.POINTER(_I), C.POINTER(_F)]),
    "pbSimStepTimedWall": (_I, [_VP, _F, _F, _I, C.POINTER(_I), C.POINTER(_F), C.POINTER(C.c_double)]),
    "pbSimSynchronize": (_I, [_VP]),
    "pbSimCentroid": (_I, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pbSimGetStats": (_I, [_VP, C.POINTER(pbSimStats)]),
    "pbSimSetResortEveryStep": (_I, [_VP, _I]),
    "pbSimSetMinDistanceMode": (_I, [_VP, _I]),
    "pbSetMinDistanceMode": (_I, [_I]),
    "pbGetMinDistanceMode": (_I, []),
    "pbHostSqrtThreshold": (C.c_float, [C.c_float]),
    "pbSimSetForceVariant": (_I, [_VP, _I]),
    "pbSimSetLanesPerBot": (_I, [_VP, _I]),
    "pbSimSetResident": (_I, [_VP, _I]),
    "pbSimGetConfig": (_I, [_VP, C.POINTER(pbSimConfig)]),
    "pbSimSetForceSums": (_I, [_VP, _I]),
    "pbSimSetStreamWalk": (_I, [_VP, _I]),
    "pbSimGetStreamWalkTrips": (_I, [_VP, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "pbForceFormCount": (_I, []),
    "pbForceFormGet": (_I, [_I, C.POINTER(pbForceForm)]),
    "pbSimSelectForceForm": (_I, [_VP, _I]),
    "pbSimForceKernelName": (_I, [_VP, C.c_char_p, C.c_size_t]),
    "pbForceFormKernelName": (_I, [_I, _I, C.c_char_p, C.c_size_t]),
    "pbSimSetRng": (_I, [_VP, _I]),
    "pbSimGetRngStatesOf": (_I, [_VP, _U, _VP]),
    "pbSetRngKind": (_I, [_I]),
    "pbGetRngKind": (_I, []),
    "pbClockSampleBegin": (_I, [C.POINTER(_VP), C.c_double]),
    "pbClockSampleEnd": (_I, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pbSelfTest": (_I, [C.c_ulonglong] + [C.POINTER(C.c_ulonglong)] * 4),
    "pbSelfTestHoldThreshold": (_I, [C.c_float, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "pbSelfTestPairGeometry": (_I, [_U, _U, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "pbSelfTestDivision": (_I, [_U, _U, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
}

_lib = None


def lib():
    """Load libparticlebot_hip.so and bind every declared symbol.  Raises if the library or a
    symbol is missing: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(HIP_SO):
        raise RuntimeError(
            f"{HIP_SO} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(HIP_SO, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc, what="call"):
    if rc != PB_OK:
        msg = lib().pbGetLastErrorString()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def np_ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_params(d):
    """Build a SimParams from a dict of scalar fields + obstacle lists.  Returns (params, keepalive)."""
    P = SimParams()
    keep = []
    lists = {"x1obs", "x2obs", "y1obs", "y2obs", "x_cir_obs", "y_cir_obs", "r_cir_obs"}
    for k, v in d.items():
        if k in lists:
            continue
        if k == "gridSize":
            P.gridSize = uint2(int(v[0]), int(v[1]))
        elif k == "worldOrigin":
            P.worldOrigin = float2(float(v[0]), float(v[1]))
        elif k == "cellSize":
            P.cellSize = float2(float(v[0]), float(v[1]))
        else:
            setattr(P, k, v)
    for k in lists:
        vals = list(d.get(k, []))[:PB_MAX_OBSTACLES]
        arr = (C.c_float * max(1, len(vals)))(*[float(x) for x in vals])
        keep.append(arr)
        setattr(P, k, C.cast(arr, C.POINTER(C.c_float)))
    return P, keep
