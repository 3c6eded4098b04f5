"""Ensembles of independent simulations (Monte-Carlo seeds, parameter sweeps) sharded over the GPUs
of a node: SURVEY.md 8(e).

  member k  ->  rank k mod world_size           (equal counts, no data-path collective)
  per rank  ->  ONE batched pbSim holding all of the rank's members (one launch per timestep)
  exchange  ->  per-member summary rows (time, COMx, COMy, distance to light), gathered once at the
                end with all_gather (RCCL over xGMI on GPUs; gloo in the CPU tests)

Run as a script under torch.distributed.run for N > 1:
  python -m particlerobotsimulations_amd.ensemble examples/example_obstacle.cfg --members 256 \
         --seed0 1000 --set max_time 1200
"""
import argparse
import ctypes as C
import json
import os
import time

import numpy as np


def shard(n_members, rank, world):
    """Indices of the members rank `rank` runs (round-robin: equal counts +-1)."""
    return list(range(rank, n_members, world))


def member_overrides(k, seed0=0, sweep=None, cartesian=False):
    """Override text of member k: its seed, plus one sweep value if `sweep` = (key, values).  cartesian (as
    `particlebot_ensemble --cartesian`): the values form a grid and every grid point runs under each seed -- member k is
    grid point k mod G under seed seed0 + k // G -- so the members of one seed share ONE placement."""
    if sweep is None:
        return f"seed\n{seed0 + k}"
    key, values = sweep
    g = len(values)
    seed = seed0 + (k // g if cartesian else k)
    return f"seed\n{seed}\n{key}\n{values[k % g]}"


class LocalEnsemble:
    """The members this rank runs, as ONE batched pbSim on the current GPU (pbEnsemble* in
    csrc/pb_capi.cpp): placement and dead-bot draws on the host from each member's own stream, one
    kernel launch per timestep for the whole batch, summary rows whenever a dump row is due."""

    def __init__(self, cfg_path, overrides_per_member, common=None, max_rows=4096):
        from . import host
        L = host.lib()
        L.pbEnsembleCreate.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int]
        L.pbEnsembleCreate.restype = C.c_void_p
        L.pbEnsembleDestroy.argtypes = [C.c_void_p]
        L.pbEnsembleRun.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.pbEnsembleRun.restype = C.c_long
        L.pbEnsembleRunSteps.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.pbEnsembleRunSteps.restype = C.c_long
        L.pbEnsembleSynchronize.argtypes = [C.c_void_p]
        L.pbEnsembleNumBots.argtypes = [C.c_void_p]
        L.pbEnsembleNumBots.restype = C.c_uint
        L.pbEnsembleGetState.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self._L = L
        self.m = len(overrides_per_member)
        self.max_rows = max_rows
        self._h = None
        self._rows = C.c_int(0)
        self.out = np.zeros((self.m, max_rows, 4), np.float32)
        if self.m == 0:
            return
        arr = (C.c_char_p * self.m)(*[o.encode() for o in overrides_per_member])
        common_b = "\n".join(f"{k}\n{v}" for k, v in (common or {}).items()).encode() or None
        self._h = L.pbEnsembleCreate(os.fsencode(cfg_path), common_b, arr, self.m)
        if not self._h:
            raise RuntimeError("pbEnsembleCreate failed")
        self.n = int(L.pbEnsembleNumBots(self._h))

    def run_steps(self, max_steps):
        """Up to max_steps timesteps of every member (stops at max_time); returns the number run."""
        if self._h is None:
            return 0
        steps = self._L.pbEnsembleRunSteps(self._h, int(max_steps), self.out.ctypes.data_as(C.c_void_p),
                                           self.max_rows, C.byref(self._rows))
        if steps < 0:
            raise RuntimeError("pbEnsembleRunSteps failed")
        return int(steps)

    def run(self):
        """To max_time."""
        return self.run_steps(2 ** 62)

    def synchronize(self):
        if self._h is not None:
            self._L.pbEnsembleSynchronize(self._h)

    @property
    def rows(self):
        """[members, rows written so far, 4] = (time, COMx, COMy, distance of the COM to the light)."""
        return self.out[:, :self._rows.value].copy()

    def final_states(self):
        states = []
        for k in range(self.m):
            st = {"pos": np.empty((self.n, 2), np.float32), "vel": np.empty((self.n, 2), np.float32),
                  "rad": np.empty(self.n, np.float32)}
            if self._L.pbEnsembleGetState(self._h, k, *[st[x].ctypes.data_as(C.c_void_p)
                                                        for x in ("pos", "vel", "rad")]):
                raise RuntimeError("pbEnsembleGetState failed")
            states.append(st)
        return states

    def close(self):
        if self._h is not None:
            self._L.pbEnsembleDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Timings(C.Structure):
    """pbEnsembleTimings (include/particlebot_ensemble.h)."""
    _fields_ = [("wall_s", C.c_double), ("placement_cpu_s", C.c_double), ("placement_wait_s", C.c_double),
                ("upload_s", C.c_double), ("device_s", C.c_double), ("sub_batches", C.c_int), ("sub_batch", C.c_int),
                ("host_threads", C.c_int), ("pinned", C.c_int), ("numa_node", C.c_int),
                ("placement_thread_wall_s", C.c_double), ("lanes", C.c_int), ("placements_run", C.c_int),
                ("placements_shared", C.c_int)]


class HostResources(C.Structure):
    """pbHostResources (include/particlebot_ensemble.h)."""
    _fields_ = [("hardware_threads", C.c_int), ("affinity_cpus", C.c_int), ("cgroup_cpus", C.c_double),
                ("usable_cpus", C.c_int), ("local_world_size", C.c_int), ("host_threads", C.c_int),
                ("device", C.c_int), ("numa_node", C.c_int), ("numa_cpus", C.c_int), ("pin_producers", C.c_int),
                ("pci_bus_id", C.c_char * 32), ("rule", C.c_char * 320)]


def host_resources():
    """What a rank's producer pool is sized from (pbHostGetResources): hardware threads, scheduler affinity, the
    cgroup CPU quota, the ranks sharing the node, the GPU's NUMA node."""
    from . import host
    L = host.lib()
    L.pbHostGetResources.argtypes = [C.POINTER(HostResources)]
    r = HostResources()
    if L.pbHostGetResources(C.byref(r)) != 0:
        raise RuntimeError("pbHostGetResources failed")
    out = {}
    for k, _ in HostResources._fields_:
        v = getattr(r, k)
        out[k] = v.decode() if isinstance(v, bytes) else v
    return out


def _pipeline_lib():
    from . import host
    L = host.lib()
    L.pbEnsemblePipelineCreate.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int,
                                           C.c_int]
    L.pbEnsemblePipelineCreate.restype = C.c_void_p
    L.pbEnsemblePipelineCreateCheckpointed.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int,
                                                       C.c_int, C.c_int, C.c_char_p, C.c_int]
    L.pbEnsemblePipelineCreateCheckpointed.restype = C.c_void_p
    L.pbEnsemblePipelineRun.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                        C.POINTER(Timings)]
    L.pbEnsemblePipelineRun.restype = C.c_long
    L.pbEnsemblePipelineDestroy.argtypes = [C.c_void_p]
    L.pbEnsemblePipelinePlacementCounts.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pbEnsemblePipelinePlacementCounts.restype = None
    L.pbEnsemblePipelineNumBots.argtypes = [C.c_void_p]
    L.pbEnsemblePipelineNumBots.restype = C.c_uint
    L.pbEnsemblePipelineGetState.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pbEnsemblePipelineDryRun.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    return L


class PipelinedEnsemble:
    """The members this rank runs, cut into sub-batches of `sub_batch` members: the host places sub-batch k+1 on
    `host_threads` producer threads while the device steps sub-batch k (pbEnsemblePipeline* in csrc/pb_capi.cpp).
    Placement starts in the constructor.  Rows and final states do not depend on sub_batch or host_threads.
    sub_batch 0: all members in one batch; -1: automatic (pbEnsemblePipelineAutoSubBatch: whole placement
    rounds of the producer pool that bring a sub-batch to ~3e6 bots; for large members)."""

    def __init__(self, cfg_path, overrides_per_member, common=None, sub_batch=0, host_threads=0, max_rows=4096,
                 keep_final_states=False, checkpoint_dir=None, resume=False, lanes=None, csv_dir=None, csv_ids=None):
        """lanes: sub-batches stepped at the same time (None: 2 with sub_batch -1, else 1).
        csv_dir: every member also writes csv_dir/member_<id>.csv, the file the reference writes for that member run
        alone (testing 0), byte for byte; csv_ids: the members' numbers in the whole ensemble (default 0, 1, ...)."""
        self._L = _pipeline_lib()
        self.m = len(overrides_per_member)
        self.max_rows = max_rows
        self.out = np.zeros((self.m, max_rows, 4), np.float32)
        self._rows = C.c_int(0)
        self.timings = None
        self._h = None
        if self.m == 0:
            return
        arr = (C.c_char_p * self.m)(*[o.encode() for o in overrides_per_member])
        common_b = "\n".join(f"{k}\n{v}" for k, v in (common or {}).items()).encode() or None
        # checkpoint_dir: every member saved exactly at each summary row; resume: continue from what is there
        self._h = self._L.pbEnsemblePipelineCreateCheckpointed(
            os.fsencode(cfg_path), common_b, arr, self.m, int(sub_batch), int(host_threads),
            1 if keep_final_states else 0, os.fsencode(checkpoint_dir) if checkpoint_dir else None, 1 if resume else 0)
        if not self._h:
            raise RuntimeError("pbEnsemblePipelineCreate failed")
        if csv_dir is not None:
            ids = (C.c_int * self.m)(*[int(i) for i in csv_ids]) if csv_ids is not None else None
            self._L.pbEnsemblePipelineSetCsvDir.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
            if self._L.pbEnsemblePipelineSetCsvDir(self._h, os.fsencode(csv_dir), ids) != 0:
                self.close()
                raise ValueError("csv_dir cannot be combined with checkpoints")
        if lanes is not None:
            self._L.pbEnsemblePipelineSetLanes.argtypes = [C.c_void_p, C.c_int]
            if self._L.pbEnsemblePipelineSetLanes(self._h, int(lanes)) != 0:
                self.close()
                raise ValueError(f"lanes {lanes}: 1 ... 4")

    def run(self, max_steps=2 ** 62):
        """Every member up to max_steps timesteps (or to max_time); returns the timesteps per member."""
        if self._h is None:
            return 0
        tm = Timings()
        steps = self._L.pbEnsemblePipelineRun(self._h, int(max_steps), self.out.ctypes.data_as(C.c_void_p),
                                              self.max_rows, C.byref(self._rows), C.byref(tm))
        if steps < 0:
            raise RuntimeError("pbEnsemblePipelineRun failed")
        self.timings = {k: getattr(tm, k) for k, _ in Timings._fields_}
        self.n = int(self._L.pbEnsemblePipelineNumBots(self._h))
        return int(steps)

    def run_dry(self, max_steps, nrows=3):
        """CPU-only stand-in for run() (bench.py --dry-run-device, tests): the members are placed for real and taken
        by the pipeline's dry-run consumer; every member then gets `nrows` summary rows made from the checksum of
        its placed state (so a row identifies its member whatever rank placed it), no timestep runs."""
        sums, _ahead = self.dry_run()
        self._rows = C.c_int(nrows)
        for k in range(self.m):
            v = [float((int(sums[k]) >> s) & 0xFFFF) / 65536.0 for s in (0, 16, 32)]
            for r in range(nrows):
                self.out[k, r] = (r * 6.0, v[0], v[1], v[2])
        self.n = int(self._L.pbEnsemblePipelineNumBots(self._h))
        self.timings = {k: 0 for k, _ in Timings._fields_}
        self.timings.update(host_threads=self.host_threads, sub_batch=0, dry_run=True)
        return int(max_steps)

    @property
    def host_threads(self):
        """producer threads this pipeline started"""
        self._L.pbEnsemblePipelineHostThreads.argtypes = [C.c_void_p]
        return int(self._L.pbEnsemblePipelineHostThreads(self._h)) if self._h else 0

    def placement_counts(self):
        """(placements computed, members that took a copy of another member's placement) so far."""
        run, shared = C.c_int(0), C.c_int(0)
        self._L.pbEnsemblePipelinePlacementCounts(self._h, C.byref(run), C.byref(shared))
        return run.value, shared.value

    def dry_run(self, dwell_ms=0):
        """CPU-only: the pipeline's consumer without a device.  Returns (checksums[m] of the placed members,
        the most members ever claimed by the producers beyond the consumed ones)."""
        sums = np.zeros(self.m, np.uint64)
        ahead = C.c_int(0)
        if self._L.pbEnsemblePipelineDryRun(self._h, int(dwell_ms), sums.ctypes.data_as(C.c_void_p), C.byref(ahead)):
            raise RuntimeError("pbEnsemblePipelineDryRun failed")
        return sums, ahead.value

    @property
    def rows(self):
        return self.out[:, :self._rows.value].copy()

    def final_states(self):
        states = []
        for k in range(self.m):
            st = {"pos": np.empty((self.n, 2), np.float32), "vel": np.empty((self.n, 2), np.float32),
                  "rad": np.empty(self.n, np.float32)}
            if self._L.pbEnsemblePipelineGetState(self._h, k, *[st[x].ctypes.data_as(C.c_void_p)
                                                                for x in ("pos", "vel", "rad")]):
                raise RuntimeError("pbEnsemblePipelineGetState failed (keep_final_states not set?)")
            states.append(st)
        return states

    def close(self):
        if self._h is not None:
            self._L.pbEnsemblePipelineDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_local(cfg_path, overrides_per_member, common=None, max_rows=4096, final_state=False):
    """Run the given members (a list of override strings) as one batch on the current GPU.
    Returns (rows[m, r, 4] float32, steps), plus a list of per-member dicts (pos, vel, rad at the
    end of the run) when final_state is set."""
    if len(overrides_per_member) == 0:
        return np.zeros((0, 0, 4), np.float32), 0
    e = LocalEnsemble(cfg_path, overrides_per_member, common, max_rows)
    try:
        steps = e.run()
        if final_state:
            return e.rows, steps, e.final_states()
        return e.rows, steps
    finally:
        e.close()


def gather_summaries(local_rows, n_members, rank, world, dist=None, device="cpu"):
    """All ranks' summary rows assembled in member order: [n_members, rows, 4].  The one collective
    of an ensemble run (tens of KB: latency-bound, nowhere near a link's bandwidth)."""
    if world == 1 or dist is None:
        return local_rows
    import torch
    per = (n_members + world - 1) // world
    rows = local_rows.shape[1] if local_rows.size else 0
    r = torch.tensor([rows], dtype=torch.int64, device=device)
    dist.all_reduce(r, op=dist.ReduceOp.MAX)
    rows = int(r.item())
    pad = np.full((per, rows, 4), np.nan, np.float32)
    pad[:local_rows.shape[0], :local_rows.shape[1]] = local_rows
    mine = torch.from_numpy(pad).to(device)
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    out = np.full((n_members, rows, 4), np.nan, np.float32)
    for rk in range(world):
        ids = shard(n_members, rk, world)
        out[ids] = allv[rk].cpu().numpy()[:len(ids)]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cfg")
    ap.add_argument("--members", type=int, default=32)
    ap.add_argument("--seed0", type=int, default=1000)
    ap.add_argument("--set", nargs=2, action="append", default=[], metavar=("NAME", "VALUE"))
    ap.add_argument("--sweep", nargs="+", default=None, metavar="KEY V1 V2 ...")
    ap.add_argument("--cartesian", action="store_true",
                    help="with --sweep: every swept value under each seed (member k = value k mod G under seed seed0 + k // G)")
    ap.add_argument("--out", default=None, help="write the gathered summaries (.npy) on rank 0")
    ap.add_argument("--sub-batch", type=int, default=0,
                    help="members per sub-batch of the placement/stepping pipeline (0: all at once; -1: automatic, for "
                         "members of ~1e5 bots and more)")
    ap.add_argument("--host-threads", type=int, default=0, help="producer threads (0: this rank's share of the host)")
    ap.add_argument("--csv-dir", default=None,
                    help="member k also writes DIR/member_<k>.csv: the reference's own CSV of that member (testing 0)")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    device = "cpu"
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        device = f"cuda:{local_rank}"
        dist.init_process_group(backend="nccl", device_id=torch.device(device))
        # the engine is pointed at this rank's GPU through its own C-ABI too (torch.cuda.set_device did it for torch's
        # view of the runtime; the batches must not depend on the two sharing one)
        from . import _capi
        _capi.check(_capi.lib().pbSetDevice(local_rank), "pbSetDevice")
    sweep = (args.sweep[0], args.sweep[1:]) if args.sweep else None
    ids = shard(args.members, rank, world)
    t0 = time.perf_counter()
    # (the same pipeline as bin/particlebot_ensemble: placement overlapped with stepping)
    e = PipelinedEnsemble(args.cfg, [member_overrides(k, args.seed0, sweep, args.cartesian) for k in ids], dict(args.set),
                          sub_batch=args.sub_batch, host_threads=args.host_threads, csv_dir=args.csv_dir,
                          csv_ids=ids if args.csv_dir else None)
    steps = e.run()
    rows = e.rows
    placement_s = e.timings["placement_wait_s"] if e.timings else 0.0  # what the device waited for the host
    e.close()
    wall = time.perf_counter() - t0
    allrows = gather_summaries(rows, args.members, rank, world, dist, device)
    if world > 1:
        import torch
        t = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    if rank == 0:
        from . import host
        n = host.load_config(args.cfg, **dict(args.set)).nCells
        last = allrows[:, -1]
        first = allrows[:, 0]
        toward = first[:, 3] - last[:, 3]  # decrease of the COM's distance to the light
        print(json.dumps({
            "cfg": os.path.basename(args.cfg), "members": args.members, "n_gpus": world, "bots_per_member": int(n),
            "steps_per_member": steps, "rows_per_member": int(allrows.shape[1]), "wall_s": wall,
            "placement_s_rank0": placement_s, "placement_share_rank0": placement_s / wall,
            "sims_per_s": args.members / wall, "particle_steps_per_s": args.members * n * steps / wall,
            "progress_toward_light_mean": float(np.nanmean(toward)), "progress_toward_light_std": float(np.nanstd(toward)),
        }))
        if args.out:
            np.save(args.out, allrows)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
