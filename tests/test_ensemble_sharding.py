"""CPU tests of the multi-GPU ensemble layer: member -> rank sharding and the one
collective (all_gather of per-member summary rows), exercised with world_size 2 over gloo."""
import os
import socket

import numpy as np
import pytest


def test_shard_is_a_partition():
    from particlerobotsimulations_amd.ensemble import shard
    for n, w in ((256, 8), (1024, 8), (10, 4), (3, 8), (0, 2)):
        parts = [shard(n, r, w) for r in range(w)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_member_overrides_text():
    from particlerobotsimulations_amd.ensemble import member_overrides
    assert member_overrides(3, seed0=1000) == "seed\n1003"
    assert member_overrides(5, 10, ("nDead", ["0", "10", "20"])) == "seed\n15\nnDead\n20"
    # cartesian (particlebot_ensemble --cartesian): every value under each seed, value fastest
    assert [member_overrides(k, 10, ("nDead", ["0", "10", "20"]), cartesian=True) for k in (0, 2, 3, 7)] == [
        "seed\n10\nnDead\n0", "seed\n10\nnDead\n20", "seed\n11\nnDead\n0", "seed\n12\nnDead\n10"]


def _fake_rows(member, rows):
    """what a rank would have computed for this member"""
    t = np.arange(rows, dtype=np.float32)
    return np.stack([t, member + 0.25 * t, -member - 0.5 * t, 100.0 - member - t], -1)


def _worker(rank, world, port, n_members, rows, q):
    import torch.distributed as dist
    from particlerobotsimulations_amd.ensemble import gather_summaries, shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    ids = shard(n_members, rank, world)
    local = np.stack([_fake_rows(k, rows) for k in ids]) if ids else np.zeros((0, rows, 4), np.float32)
    out = gather_summaries(local.astype(np.float32), n_members, rank, world, dist, "cpu")
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_members", [7, 8])
def test_gather_world_size_2_gloo(n_members):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rows = 5
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_members, rows, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([_fake_rows(k, rows) for k in range(n_members)]).astype(np.float32)
    for r in range(2):
        assert got[r].shape == want.shape
        assert np.array_equal(got[r], want)


def test_gather_single_rank_is_identity():
    from particlerobotsimulations_amd.ensemble import gather_summaries
    x = np.random.default_rng(0).random((3, 4, 4)).astype(np.float32)
    assert gather_summaries(x, 3, 0, 1) is x


# ---- the C++ side of the same layer (include/particlebot_ensemble.h; bin/particlebot_ensemble) ----

@pytest.mark.parametrize("n_members,world", [(7, 2), (8, 2), (256, 8), (1024, 8), (3, 8), (5, 1)])
def test_cxx_shard_and_assemble_agree_with_the_python_layer(n_members, world):
    """pbEnsembleShard / pbEnsembleAssemble (what the RCCL runner uses after its ncclAllGather) against
    ensemble.shard and the layout gather_summaries produces: the gathered buffer is built here exactly
    as an all-gather of equal NaN-padded blocks leaves it."""
    import ctypes as C

    from particlerobotsimulations_amd import host
    from particlerobotsimulations_amd.ensemble import shard
    L = host.lib()
    L.pbEnsembleShard.argtypes = [C.c_int, C.c_int, C.c_int]
    L.pbEnsembleAssemble.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rows = 3
    per = L.pbEnsembleShard(n_members, 0, world)
    assert per == len(shard(n_members, 0, world))
    gathered = np.full((world, per, rows, 4), np.nan, np.float32)
    for r in range(world):
        ids = shard(n_members, r, world)
        assert L.pbEnsembleShard(n_members, r, world) == len(ids)
        for j, k in enumerate(ids):
            gathered[r, j] = _fake_rows(k, rows)
    out = np.full((n_members, rows, 4), -1.0, np.float32)
    assert L.pbEnsembleAssemble(n_members, world, rows, gathered.ctypes.data_as(C.c_void_p),
                                out.ctypes.data_as(C.c_void_p)) == 0
    want = np.stack([_fake_rows(k, rows) for k in range(n_members)]).astype(np.float32)
    assert np.array_equal(out, want)


def test_cxx_runner_fails_loudly_without_a_gpu_or_arguments():
    """bin/particlebot_ensemble: usage error without a configuration; and on a machine without a GPU a
    clear error and a nonzero exit (never a CPU fallback)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "particlerobotsimulations_amd", "bin", "particlebot_ensemble")
    if not os.path.exists(exe):
        pytest.skip("runner not built")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "--members" in r.stderr
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, os.path.join(root, "examples", "example_dead_cells.cfg"), "--members", "2"],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and ("HIP" in r.stderr or "device" in r.stderr), r.stderr


@pytest.mark.parametrize("how", ["tcp", "file"])
def test_cxx_runner_id_exchange_world_size_3(how, tmp_path):
    """bin/particlebot_ensemble's rendezvous (the RCCL unique id from rank 0 to the other ranks) without any GPU:
    `--rendezvous-test` runs just that exchange with a known 128-byte pattern.  Three processes, the readers started
    BEFORE rank 0 (they must retry until it serves), over TCP (the default) and through a file; a stale file left by
    a dead process must not be taken for this launch's."""
    import subprocess
    import sys
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "particlerobotsimulations_amd", "bin",
                       "particlebot_ensemble")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    extra = []
    if how == "file":
        path = tmp_path / "id.bin"
        # a file from an "earlier launch": right magic, the pid of a process that no longer exists, another id
        dead = subprocess.Popen([sys.executable, "-c", "pass"])
        dead.wait()
        path.write_bytes(b"PBIDF2\0\0" + int(dead.pid).to_bytes(4, "little") + bytes(4 + 8 + 128))
        extra = ["--rendezvous", str(path)]
    env = lambda r: dict(os.environ, RANK=str(r), WORLD_SIZE="3", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                         PB_RENDEZVOUS_PORT=str(port))
    procs = [subprocess.Popen([exe, "--rendezvous-test"] + extra, env=env(r), stdout=subprocess.PIPE, text=True)
             for r in (2, 1)]
    import time
    time.sleep(0.3)
    procs.append(subprocess.Popen([exe, "--rendezvous-test"] + extra, env=env(0), stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=60)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert sorted(o.strip() for o in outs) == [f"rendezvous-test rank {r} of 3: ok" for r in range(3)]


def test_cxx_runner_rendezvous_by_host_name_with_strays_and_foreign_files(tmp_path):
    """MASTER_ADDR as a HOST NAME (launchers pass names) resolves; a stray connection that speaks the
    protocol but is not of this launch (another token) neither gets served as a rank nor shortens anyone's wait; a
    rendezvous file written by a LIVE process of another launch (another token) is not taken; a name that does not
    resolve is an error message, not a silent fall-back to the loopback."""
    import subprocess
    import sys
    import time
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "particlerobotsimulations_amd", "bin",
                       "particlebot_ensemble")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = lambda r, **kw: dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="localhost",
                               PB_RENDEZVOUS_PORT=str(port), PB_LAUNCH_TOKEN="this-launch", **kw)
    rank0 = subprocess.Popen([exe, "--rendezvous-test"], env=env(0), stdout=subprocess.PIPE, text=True)
    time.sleep(0.3)
    # a stray: right magic, claims to be rank 1, wrong token  (rank 0 may need longer than 0.3 s to listen on a busy box)
    deadline = time.time() + 10.0
    while True:
        try:
            c = socket.create_connection(("127.0.0.1", port), timeout=5)
            break
        except OSError:
            if time.time() > deadline:
                raise
            time.sleep(0.1)
    c.sendall(b"PBID" + (1).to_bytes(4, "little") + bytes(8))
    time.sleep(0.2)
    c.close()
    rank1 = subprocess.Popen([exe, "--rendezvous-test"], env=env(1), stdout=subprocess.PIPE, text=True)
    outs = [p.communicate(timeout=60)[0] for p in (rank0, rank1)]
    assert rank0.returncode == 0 and rank1.returncode == 0, outs
    assert sorted(o.strip() for o in outs) == [f"rendezvous-test rank {r} of 2: ok" for r in range(2)]
    # file mode: a record of ANOTHER launch whose writer is alive (this very process) is ignored until the deadline
    path = tmp_path / "id.bin"
    path.write_bytes(b"PBIDF2\0\0" + int(os.getpid()).to_bytes(4, "little") + bytes(4) + (12345).to_bytes(8, "little")
                     + bytes(128))
    t0 = time.time()
    p = subprocess.run([exe, "--rendezvous-test", "--rendezvous", str(path)], env=env(1), capture_output=True, text=True,
                       timeout=90)
    assert p.returncode != 0 and time.time() - t0 > 5
    # an unresolvable name
    p = subprocess.run([exe, "--rendezvous-test"], env=dict(env(1), MASTER_ADDR="no-such-host.invalid"),
                       capture_output=True, text=True, timeout=90)
    assert p.returncode != 0 and "cannot resolve MASTER_ADDR" in p.stderr
