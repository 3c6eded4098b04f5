"""CPU test: the C-ABI library loads and exports every symbol include/particlebot_hip.h declares
(no compute calls: there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "particlebot_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set()
    for m in re.finditer(r"\b([A-Za-z_]\w*)\s*\(", text):
        name = m.group(1)
        if name in {"defined", "sizeof"}:
            continue
        names.add(name)
    return names


@pytest.fixture(scope="module")
def capi():
    from particlerobotsimulations_amd import _capi
    if not os.path.exists(_capi.HIP_SO):
        import __graft_entry__
        __graft_entry__.build()
    return _capi


def test_header_and_binding_agree(capi):
    declared = _declared_symbols()
    assert declared == set(capi.SYMBOLS), (declared ^ set(capi.SYMBOLS))


def test_library_exports_every_symbol(capi):
    L = capi.lib()
    for name in capi.SYMBOLS:
        assert hasattr(L, name), name


def test_simparams_layout_matches_reference_struct(capi):
    """SimParams keeps the reference's field order/types (particlebot_kernel.cuh:58-120):
    float2/uint2 8-byte aligned as in CUDA/HIP, 8-byte pointers.  The same numbers are
    static_assert'ed against the C++ struct in csrc/pb_device.hpp."""
    import ctypes as C
    P = capi.SimParams
    assert P.gridSize.offset == 0 and P.numCells.offset == 8 and P.worldOrigin.offset == 16
    assert P.nCells.offset == 32 and P.nDead.offset == 36 and P.gravity.offset == 44
    assert P.nobstacles.offset == 144 and P.x1obs.offset == 152 and P.y2obs.offset == 176
    assert P.n_cir_obstacles.offset == 184 and P.x_cir_obs.offset == 192 and P.Nx.offset == 216
    assert P.max_time.offset == 248 and C.sizeof(P) == 256


def test_engine_rejects_bad_arguments_without_a_gpu(capi):
    import ctypes as C
    L = capi.lib()
    h = C.c_void_p()
    assert L.pbSimCreate(C.byref(h), None, 0.0) == 2  # PB_ERR_ARG
    p, keep = capi.make_params({"gridSize": (500, 512), "numCells": 500 * 512, "nCells": 10})
    assert L.pbSimCreate(C.byref(h), C.byref(p), 0.0) == 2
    assert b"power of two" in L.pbGetLastErrorString()


def test_engine_error_paths_without_a_gpu(capi):
    """Every argument check of pbSimCreate[Batch] runs before the device is touched, and a valid
    request on a machine without a GPU fails with PB_ERR_NO_DEVICE and a message (never a crash,
    never a CPU fallback).  Setters and getters reject a null handle."""
    import ctypes as C
    L = capi.lib()
    h = C.c_void_p()
    ok, keep = capi.make_params({"gridSize": (512, 512), "numCells": 512 * 512, "nCells": 10})
    assert L.pbSimCreate(None, C.byref(ok), 0.0) == 2
    zero, k0 = capi.make_params({"gridSize": (512, 512), "numCells": 512 * 512, "nCells": 0})
    assert L.pbSimCreate(C.byref(h), C.byref(zero), 0.0) == 2 and b"nCells" in L.pbGetLastErrorString()
    tiny, k1 = capi.make_params({"gridSize": (4, 4), "numCells": 16, "nCells": 10})
    assert L.pbSimCreate(C.byref(h), C.byref(tiny), 0.0) == 2
    # batch: nsims < 1, mismatched members, too many bots for 32-bit byte offsets
    assert L.pbSimCreateBatch(C.byref(h), C.byref(ok), 0, 0.0) == 2
    other, k2 = capi.make_params({"gridSize": (512, 512), "numCells": 512 * 512, "nCells": 11})
    arr = (capi.SimParams * 2)(ok, other)
    assert L.pbSimCreateBatch(C.byref(h), arr, 2, 0.0) == 2 and b"must share" in L.pbGetLastErrorString()
    big, k3 = capi.make_params({"gridSize": (512, 512), "numCells": 512 * 512, "nCells": 2_000_000_000})
    arr2 = (capi.SimParams * 3)(big, big, big)
    assert L.pbSimCreateBatch(C.byref(h), arr2, 3, 0.0) == 2 and b"2^32" in L.pbGetLastErrorString()
    # a valid request: no device here
    import torch
    if not torch.cuda.is_available():
        rc = L.pbSimCreate(C.byref(h), C.byref(ok), 0.0)
        assert rc == 3 and b"no HIP device" in L.pbGetLastErrorString() and not h.value
    # null handles
    assert L.pbSimStep(None, C.c_float(0.01), C.c_float(180.0), 1, None) == 2
    assert L.pbSimSetForceVariant(None, 2) == 2 and L.pbSimSetLanesPerBot(None, 8) == 2
    assert L.pbSimSetResident(None, 0) == 2 and L.pbSimGetTime(None, None) == 2


def test_ensemble_header_symbols_are_exported_by_the_host_library():
    """include/particlebot_ensemble.h (the multi-GPU ensemble layer's C-ABI) against
    libparticlebot_host.so: every declared function is exported, and the two pure helpers behave."""
    import ctypes as C

    from particlerobotsimulations_amd import host
    text = open(os.path.join(ROOT, "include", "particlebot_ensemble.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = {m.group(1) for m in re.finditer(r"\b(pb(?:Ensemble|Host)\w+)\s*\(", text)}
    assert names == {"pbEnsembleCreate", "pbEnsembleDestroy", "pbEnsembleRun", "pbEnsembleRunSteps",
                     "pbEnsembleSynchronize", "pbEnsembleGetState", "pbEnsembleNumBots", "pbEnsembleShard",
                     "pbEnsembleAssemble", "pbEnsemblePipelineCreate", "pbEnsemblePipelineCreateCheckpointed",
                     "pbEnsemblePipelineRun", "pbEnsemblePipelineDestroy", "pbEnsemblePipelineNumBots",
                     "pbEnsemblePipelineGetState", "pbEnsemblePipelineDryRun", "pbEnsemblePipelineHostThreads", "pbEnsemblePipelineAutoSubBatch", "pbEnsemblePipelineSetLanes", "pbEnsemblePipelineSetCsvDir",
                     "pbEnsemblePipelinePlacementCounts",
                     "pbHostGetResources", "pbHostParseCpuList"}
    L = host.lib()
    for n in names:
        assert hasattr(L, n), n
    L.pbEnsembleShard.argtypes = [C.c_int, C.c_int, C.c_int]
    assert [L.pbEnsembleShard(10, r, 4) for r in range(4)] == [3, 3, 2, 2]
    assert L.pbEnsembleShard(10, 4, 4) == 0 and L.pbEnsembleShard(-1, 0, 4) == 0
    L.pbEnsembleAssemble.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    assert L.pbEnsembleAssemble(4, 2, 1, None, None) != 0   # null buffers are refused
    # without a GPU a valid ensemble request fails cleanly (NULL), never falls back to the CPU
    import torch
    if not torch.cuda.is_available():
        L.pbEnsembleCreate.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int]
        L.pbEnsembleCreate.restype = C.c_void_p
        arr = (C.c_char_p * 1)(b"seed\n1")
        cfg = os.path.join(ROOT, "examples", "example_dead_cells.cfg").encode()
        assert L.pbEnsembleCreate(cfg, None, arr, 1) is None


def test_force_forms_table_is_enumerable_without_a_gpu(capi):
    """The forms table of the exact force kernel (csrc/pb_force.hip): 17 distinct rows -- the reference-shaped
    kernel, the throughput form with 32- and 64-bit offsets, 2..64 lanes per bot, each branch-free shape with
    and without Sum|F_attr| -- and the index checks of its accessors."""
    import ctypes as C
    L = capi.lib()
    n = L.pbForceFormCount()
    assert n == 17
    rows = set()
    for i in range(n):
        f = capi.pbForceForm()
        assert L.pbForceFormGet(i, C.byref(f)) == 0
        rows.add((f.flat, f.lanes_per_bot, f.attraction_sums, f.offsets64))
        assert f.lanes_per_bot in (1, 2, 4, 8, 16, 32, 64)
        assert f.flat or (f.lanes_per_bot == 1 and f.attraction_sums == 1 and f.offsets64 == 0)
        assert not f.offsets64 or f.lanes_per_bot == 1
    assert len(rows) == n
    f = capi.pbForceForm()
    assert L.pbForceFormGet(n, C.byref(f)) == 2 and L.pbForceFormGet(-1, C.byref(f)) == 2
    assert L.pbSimSelectForceForm(None, 0) == 2


def test_min_distance_mode_precedence_api_over_environment(capi):
    """PB_MIN_DISTANCE_MODE used to be applied AFTER the process-wide default in pbSimCreateBatch, so an
    inherited PB_MIN_DISTANCE_MODE=0 silently overrode an explicit pbSetMinDistanceMode(1) (chosen because this
    host's libm failed pbHostLibmCheck).  The environment now only seeds the default; the API call wins.  Each case
    in a child process (the default is decided once per process)."""
    import subprocess
    import sys
    child = ("import sys; sys.path.insert(0, %r)\n"
             "from particlerobotsimulations_amd import _capi\n"
             "L = _capi.lib()\n"
             "a = L.pbGetMinDistanceMode()\n"
             "if len(sys.argv) > 1: assert L.pbSetMinDistanceMode(int(sys.argv[1])) == 0\n"
             "print(a, L.pbGetMinDistanceMode())\n") % ROOT

    def run(env_value, *args):
        env = {k: v for k, v in os.environ.items() if k != "PB_MIN_DISTANCE_MODE"}
        if env_value is not None:
            env["PB_MIN_DISTANCE_MODE"] = env_value
        out = subprocess.check_output([sys.executable, "-c", child, *args], env=env, text=True)
        return [int(x) for x in out.split()[-2:]]

    assert run(None) == [0, 0]
    assert run("1") == [1, 1]
    assert run("0", "1") == [0, 1]      # the case that used to lose
    assert run("1", "0") == [1, 0]
    assert run("7") == [0, 0] and run("11") == [0, 0]
    assert capi.lib().pbSetMinDistanceMode(2) != 0
