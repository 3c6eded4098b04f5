"""particlerobotsimulations_amd -- MI355X-native particle-robot update loop.

Python is only the scripting shell over the C-ABI of libparticlebot_hip.so (hand-written HIP for
gfx950): device memory, the reference's `extern "C"` device boundary (`legacy`) and the resident
fused engine (`Sim`).  The C++ host side (class Particlebot, .cfg loader, headless runner) lives in
csrc/ and libparticlebot_host.so.

There is no CPU fallback anywhere in this package.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import SimParams, make_params, pbSimConfig, pbSimStats  # noqa: F401

__all__ = ["Sim", "Ensemble", "DeviceArray", "legacy", "SimParams", "make_params", "library_paths", "self_test"]


def self_test(div_samples=1 << 32):
    """pbSelfTest: fast exact sqrt/division forms vs the compiler's IEEE forms, on the GPU."""
    vals = [C.c_ulonglong() for _ in range(4)]
    _capi.check(_capi.lib().pbSelfTest(int(div_samples), *[C.byref(v) for v in vals]), "pbSelfTest")
    keys = ("sqrt_checked", "sqrt_mismatches", "div_checked", "div_mismatches")
    return {k: int(v.value) for k, v in zip(keys, vals)}


def self_test_hold_threshold(c):
    """pbSelfTestHoldThreshold: `sqrtf(x) < c` against `x < T(c)` on the GPU for every non-negative float x."""
    checked, bad = C.c_ulonglong(), C.c_ulonglong()
    _capi.check(_capi.lib().pbSelfTestHoldThreshold(float(c), C.byref(checked), C.byref(bad)), "pbSelfTestHoldThreshold")
    return {"checked": checked.value, "mismatches": bad.value}


def self_test_pair_geometry(first_slice=0, slices=64):
    """pbSelfTestPairGeometry: pbDistUnitFast against sqrtf and IEEE division on EVERY (d2, numerator)
    mantissa pair of `slices` of the 64 slices of d2 in [1, 4) (all 64: 2^47 pairs, ~90 s of one MI355X)."""
    checked, bad = C.c_ulonglong(), C.c_ulonglong()
    _capi.check(_capi.lib().pbSelfTestPairGeometry(int(first_slice), int(slices), C.byref(checked), C.byref(bad)),
                "pbSelfTestPairGeometry")
    return {"checked": int(checked.value), "mismatches": int(bad.value)}


def self_test_division(first_slice=0, slices=64):
    """pbSelfTestDivision: pbDiv2Fast against IEEE division on EVERY (denominator, numerator) mantissa pair of
    `slices` of the 64 slices of the denominator range (all 64: 2^46 quotients)."""
    checked, bad = C.c_ulonglong(), C.c_ulonglong()
    _capi.check(_capi.lib().pbSelfTestDivision(int(first_slice), int(slices), C.byref(checked), C.byref(bad)),
                "pbSelfTestDivision")
    return {"checked": int(checked.value), "mismatches": int(bad.value)}


def force_forms():
    """The forms table of the exact per-step force kernel (pbForceFormCount / pbForceFormGet): a list of
    dicts flat / lanes_per_bot / attraction_sums / offsets64, index = row number."""
    L = _capi.lib()
    out = []
    for i in range(L.pbForceFormCount()):
        f = _capi.pbForceForm()
        _capi.check(L.pbForceFormGet(i, C.byref(f)), "pbForceFormGet")
        out.append({k: int(getattr(f, k)) for k, _ in _capi.pbForceForm._fields_})
    return out


def force_form_kernel_name(index, payload=0):
    """Row `index` of the forms table as rocprofv3 names its kernel (pbForceFormKernelName; no device needed)."""
    buf = C.create_string_buffer(2048)
    _capi.check(_capi.lib().pbForceFormKernelName(int(index), int(payload), buf, len(buf)), "pbForceFormKernelName")
    return buf.value.decode()


def library_paths():
    return {"hip": _capi.HIP_SO, "host": _capi.HOST_SO}


class DeviceArray:
    """A caller-owned device buffer allocated through allocateArray() (particlebot.cuh:20)."""

    def __init__(self, shape, dtype=np.float32, fill=None):
        self.shape = tuple(np.atleast_1d(shape).tolist()) if not isinstance(shape, tuple) else shape
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _capi.lib().allocateArray(C.byref(p), max(self.nbytes, 1))
        self.ptr = p
        if fill is not None:
            self.upload(np.full(self.shape, fill, dtype=self.dtype))

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        d.upload(a)
        return d

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes, (a.nbytes, self.nbytes)
        if self.nbytes:
            _capi.lib().copyArrayToDevice(self.ptr, _capi.np_ptr(a), 0, self.nbytes)

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            _capi.lib().copyArrayFromDevice(_capi.np_ptr(out), self.ptr, None, self.nbytes)
        return out

    def free(self):
        if self.ptr is not None and self.ptr.value:
            _capi.lib().freeArray(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _Legacy:
    """The reference's `extern "C"` boundary (particlebot.cuh:15-121), one call per entry point.
    Arguments are DeviceArray objects; semantics are the reference's."""

    def __getattr__(self, name):
        return getattr(_capi.lib(), name)

    def set_parameters(self, params, wall_half=64.0):
        L = _capi.lib()
        L.pbSetWallHalfExtent(float(wall_half))
        L.setParameters(C.byref(params))

    def integrate(self, pos, vel, rad, dt, n, time=0.0):
        _capi.lib().integrateSystem(pos.ptr, vel.ptr, rad.ptr, dt, n, time)

    def calc_hash(self, hash_, index, pos, n):
        _capi.lib().calcHash(hash_.ptr, index.ptr, pos.ptr, n)

    def sort(self, hash_, index, n):
        _capi.lib().sortParticlebots(hash_.ptr, index.ptr, n)

    def reorder(self, cell_start, cell_end, spos, svel, srad, hash_, index, pos, vel, rad, n, num_cells):
        _capi.lib().reorderDataAndFindCellStart(cell_start.ptr, cell_end.ptr, spos.ptr, svel.ptr,
                                               srad.ptr, hash_.ptr, index.ptr, pos.ptr, vel.ptr,
                                               rad.ptr, n, num_cells)

    def update_rad(self, pos, abs_a, abs_r, rad, phase, time, dt, dead, n):
        _capi.lib().updateRad_light_wave(pos.ptr, abs_a.ptr, abs_r.ptr, rad.ptr, phase.ptr, time, dt,
                                        dead.ptr, n)

    def update_phase(self, pos, phase, spacing, max_d, min_d, n):
        _capi.lib().updatePhase(pos.ptr, phase.ptr, spacing, max_d, min_d, n)

    def rng_setup(self, state, n):
        _capi.lib().curand_setup(state.ptr, n)

    def add_noise(self, state, val, std, n):
        _capi.lib().add_normal_noise(state.ptr, val.ptr, std, n)

    def collide(self, new_vel, abs_a, abs_r, spos, svel, srad, index, cell_start, cell_end, n, num_cells, dt):
        _capi.lib().collide(new_vel.ptr, abs_a.ptr, abs_r.ptr, spos.ptr, svel.ptr, srad.ptr, index.ptr,
                           cell_start.ptr, cell_end.ptr, n, num_cells, dt)

    def sync(self):
        _capi.lib().threadSync()


legacy = _Legacy()


class ClockSample:
    """Shader clock held while the enclosed work runs (pbClockSampleBegin/End): a sleeping wave on its
    own stream spans `seconds` of real time; .mhz afterwards.  Diagnostic for the roofline report."""

    def __init__(self, seconds):
        self._h = C.c_void_p()
        _capi.check(_capi.lib().pbClockSampleBegin(C.byref(self._h), float(seconds)), "pbClockSampleBegin")

    def end(self):
        mhz, sec = C.c_double(), C.c_double()
        _capi.check(_capi.lib().pbClockSampleEnd(self._h, C.byref(mhz), C.byref(sec)), "pbClockSampleEnd")
        self._h = None
        self.mhz, self.seconds = mhz.value, sec.value
        return self.mhz


class Sim:
    """One resident simulation on the current GPU (pbSim* in include/particlebot_hip.h)."""

    def __init__(self, params, wall_half=0.0, keepalive=None):
        self._keep = keepalive
        self.params = params
        self.n = int(params.nCells)
        h = C.c_void_p()
        _capi.check(_capi.lib().pbSimCreate(C.byref(h), C.byref(params), float(wall_half)), "pbSimCreate")
        self._h = h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _capi.lib().pbSimDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _in(a, dtype, count):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=dtype).reshape(-1)
        assert a.size == count, (a.size, count)
        return a

    def set_state(self, pos=None, vel=None, rad=None, phase=None, dead=None):
        n = self.n
        pos = self._in(pos, np.float32, 2 * n)
        vel = self._in(vel, np.float32, 2 * n)
        rad = self._in(rad, np.float32, n)
        phase = self._in(phase, np.float32, n)
        dead = self._in(dead, np.int32, n)
        _capi.check(_capi.lib().pbSimSetState(self._h, _capi.np_ptr(pos), _capi.np_ptr(vel), _capi.np_ptr(rad),
                                              _capi.np_ptr(phase), _capi.np_ptr(dead)), "pbSimSetState")

    def set_forces(self, absForce_a, absForce_r):
        """Overwrite absForce_a / absForce_r (original bot order): the two arrays the next step's
        radius actuation reads.  With set_state this restores a complete mid-run state."""
        n = self.n
        a, r = self._in(absForce_a, np.float32, n), self._in(absForce_r, np.float32, n)
        _capi.check(_capi.lib().pbSimSetForcesOf(self._h, 0, _capi.np_ptr(a), _capi.np_ptr(r)), "pbSimSetForcesOf")

    def get_state(self, out=None):
        """The state in original bot order.  `out`: a dict returned by an earlier call, whose arrays are then
        written in place (a caller that reads back often keeps its host buffers: fresh pages cost more than
        the copy)."""
        n = self.n
        if out is None:
            out = {
                "pos": np.empty((n, 2), np.float32), "vel": np.empty((n, 2), np.float32),
                "rad": np.empty(n, np.float32), "phase": np.empty(n, np.float32),
                "dead": np.empty(n, np.int32), "absForce_a": np.empty(n, np.float32),
                "absForce_r": np.empty(n, np.float32),
            }
        elif out.get("absForce_a") is None:
            out["absForce_a"] = np.empty(n, np.float32)
        _capi.check(_capi.lib().pbSimGetState(self._h, *[_capi.np_ptr(out[k]) for k in
                                                         ("pos", "vel", "rad", "phase", "dead",
                                                          "absForce_a", "absForce_r")]), "pbSimGetState")
        if not self.config()["attraction_sums"]:
            out["absForce_a"] = None  # a dead value in this batch: not maintained (set_force_sums)
        return out

    @property
    def time(self):
        t = C.c_float()
        _capi.check(_capi.lib().pbSimGetTime(self._h, C.byref(t)))
        return t.value

    @time.setter
    def time(self, t):
        _capi.check(_capi.lib().pbSimSetTime(self._h, float(t)))

    @property
    def phase_draws(self):
        d = C.c_uint()
        _capi.check(_capi.lib().pbSimGetPhaseDraws(self._h, C.byref(d)))
        return d.value

    @phase_draws.setter
    def phase_draws(self, d):
        _capi.check(_capi.lib().pbSimSetPhaseDraws(self._h, int(d)))

    def step(self, nsteps, dt=0.01, sort_interval=180.0):
        done = C.c_int()
        _capi.check(_capi.lib().pbSimStep(self._h, dt, sort_interval, int(nsteps), C.byref(done)), "pbSimStep")
        return done.value

    def step_timed(self, nsteps, dt=0.01, sort_interval=180.0):
        done = C.c_int()
        ms = C.c_float()
        _capi.check(_capi.lib().pbSimStepTimed(self._h, dt, sort_interval, int(nsteps), C.byref(done),
                                               C.byref(ms)), "pbSimStepTimed")
        return done.value, ms.value

    def step_timed_wall(self, nsteps, dt=0.01, sort_interval=180.0):
        """(steps done, device ms between HIP events, host wall ms from entry to the drained stream) of the same
        launches (pbSimStepTimedWall): synchronise before calling."""
        done, ms, wall = C.c_int(), C.c_float(), C.c_double()
        _capi.check(_capi.lib().pbSimStepTimedWall(self._h, dt, sort_interval, int(nsteps), C.byref(done), C.byref(ms),
                                                   C.byref(wall)), "pbSimStepTimedWall")
        return done.value, ms.value, wall.value

    def synchronize(self):
        _capi.check(_capi.lib().pbSimSynchronize(self._h))

    def centroid(self):
        cx, cy = C.c_double(), C.c_double()
        _capi.check(_capi.lib().pbSimCentroid(self._h, C.byref(cx), C.byref(cy)), "pbSimCentroid")
        return cx.value, cy.value

    def stats(self):
        s = pbSimStats()
        _capi.check(_capi.lib().pbSimGetStats(self._h, C.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in pbSimStats._fields_}

    def config(self):
        """What the next step() launches: force_variant / force_kind / lanes_per_bot / resident /
        fast_math_ok / payload / rng (pbSimGetConfig)."""
        c = pbSimConfig()
        _capi.check(_capi.lib().pbSimGetConfig(self._h, C.byref(c)))
        return {k: int(getattr(c, k)) for k, _ in pbSimConfig._fields_}

    def force_kernel_name(self):
        """The per-step force kernel the next step() launches, as rocprofv3 names it (pbSimForceKernelName)."""
        buf = C.create_string_buffer(2048)
        _capi.check(_capi.lib().pbSimForceKernelName(self._h, buf, len(buf)), "pbSimForceKernelName")
        return buf.value.decode()

    def set_force_variant(self, variant):
        """0/1/2: exact kernels (bit-identical to the oracle; 2 is the default).  3: streamlined
        arithmetic, opt-in, not bit-identical (include/particlebot_hip.h)."""
        _capi.check(_capi.lib().pbSimSetForceVariant(self._h, int(variant)))

    def set_stream_walk(self, mode):
        """-1 (default): the streamlined kernel's neighbour walk is chosen at every re-sort; 0 row by row; 1 flattened
        (pbSimSetStreamWalk; bit-identical either way)."""
        _capi.check(_capi.lib().pbSimSetStreamWalk(self._h, int(mode)), "pbSimSetStreamWalk")

    def stream_walk_trips(self):
        """(row-by-row, flattened) trips per wave summed over the batch, from the last automatic choice."""
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        _capi.check(_capi.lib().pbSimGetStreamWalkTrips(self._h, C.byref(a), C.byref(b)), "pbSimGetStreamWalkTrips")
        return a.value, b.value

    def set_force_sums(self, mode):
        """0 (default): absForce_a only when a member reads it (constrained_contraction); otherwise it
        is a dead value, not computed, and get_state() returns NaN for it.  1: always maintained."""
        _capi.check(_capi.lib().pbSimSetForceSums(self._h, int(mode)))

    def select_force_form(self, index):
        """Pin the batch to row `index` of the force-kernel forms table (force_forms()); -1: automatic."""
        _capi.check(_capi.lib().pbSimSelectForceForm(self._h, int(index)), "pbSimSelectForceForm")

    def set_lanes_per_bot(self, lanes):
        _capi.check(_capi.lib().pbSimSetLanesPerBot(self._h, int(lanes)))

    def set_resident(self, mode):
        """0 automatic, 1 never, 2 whenever the simulation fits one workgroup (<= 1024 bots)."""
        _capi.check(_capi.lib().pbSimSetResident(self._h, int(mode)))

    def set_resort_every_step(self, on):
        _capi.check(_capi.lib().pbSimSetResortEveryStep(self._h, 1 if on else 0))


class Ensemble(Sim):
    """A batch of independent simulations of equal size stepped by the same launches
    (pbSimCreateBatch): seeds of a Monte-Carlo run, points of a parameter sweep."""

    def __init__(self, params_list, wall_half=0.0, keepalive=None):
        self._keep = keepalive
        self.nsims = len(params_list)
        arr = (SimParams * self.nsims)(*params_list)
        self.params = params_list[0]
        self.n = int(params_list[0].nCells)
        h = C.c_void_p()
        _capi.check(_capi.lib().pbSimCreateBatch(C.byref(h), arr, self.nsims, float(wall_half)), "pbSimCreateBatch")
        self._h = h

    def set_state_of(self, member, pos=None, vel=None, rad=None, phase=None, dead=None):
        n = self.n
        pos = self._in(pos, np.float32, 2 * n)
        vel = self._in(vel, np.float32, 2 * n)
        rad = self._in(rad, np.float32, n)
        phase = self._in(phase, np.float32, n)
        dead = self._in(dead, np.int32, n)
        _capi.check(_capi.lib().pbSimSetStateOf(self._h, int(member), _capi.np_ptr(pos), _capi.np_ptr(vel),
                                                _capi.np_ptr(rad), _capi.np_ptr(phase), _capi.np_ptr(dead)),
                    "pbSimSetStateOf")

    def get_state_of(self, member):
        n = self.n
        out = {
            "pos": np.empty((n, 2), np.float32), "vel": np.empty((n, 2), np.float32),
            "rad": np.empty(n, np.float32), "phase": np.empty(n, np.float32),
            "dead": np.empty(n, np.int32), "absForce_a": np.empty(n, np.float32),
            "absForce_r": np.empty(n, np.float32),
        }
        _capi.check(_capi.lib().pbSimGetStateOf(self._h, int(member), *[_capi.np_ptr(out[k]) for k in
                                                                        ("pos", "vel", "rad", "phase", "dead",
                                                                         "absForce_a", "absForce_r")]),
                    "pbSimGetStateOf")
        if not self.config()["attraction_sums"]:
            out["absForce_a"] = None
        return out

    def centroids(self):
        out = np.empty((self.nsims, 2), np.float64)
        _capi.check(_capi.lib().pbSimCentroids(self._h, out.ctypes.data_as(C.POINTER(C.c_double))), "pbSimCentroids")
        return out

    def centroid_sums(self):
        """The reference's own fp32 centroid sums per member (bots added serially in order, particlebot.cpp:335-338):
        what its CSV's centroid columns are divided from, bit for bit."""
        out = np.empty((self.nsims, 2), np.float32)
        _capi.check(_capi.lib().pbSimCentroidSums(self._h, out.ctypes.data_as(C.POINTER(C.c_float))), "pbSimCentroidSums")
        return out
