"""The ensemble layer's view of the host (pbHostGetResources, csrc/pb_capi.cpp): a rank's producer pool is sized
from what the process may really use -- hardware threads, scheduler affinity, the cgroup CPU quota -- divided by
the ranks of the node, and is pinned to the cores next to its GPU when sysfs names them.  (the
driver's box grants 16 CPUs by quota on a 128-thread host; 127 producer threads were started.)
Each case runs in a child process: the environment is read when the library decides."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys
sys.path.insert(0, %r)
from particlerobotsimulations_amd import ensemble
out = {"res": ensemble.host_resources()}
if len(sys.argv) > 1 and sys.argv[1] == "pipeline":
    import os
    cfg = os.path.join(%r, "examples", "example.cfg")
    p = ensemble.PipelinedEnsemble(cfg, [f"seed\n{k}" for k in range(1, 41)], {"nCells": "30"}, sub_batch=-1)
    sums, ahead = p.dry_run()
    out["producers"] = p.host_threads
    out["checksums"] = [int(x) for x in sums]
    p.close()
print(json.dumps(out))
""" % (ROOT, ROOT)


def run_child(env_over, *args):
    env = {k: v for k, v in os.environ.items() if k not in ("PB_HOST_THREADS", "LOCAL_WORLD_SIZE", "PB_PIN_PRODUCERS",
                                                           "OMPI_COMM_WORLD_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE")}
    env.update(env_over)
    out = subprocess.check_output([sys.executable, "-c", CHILD, *args], env=env, text=True)
    return json.loads(out.strip().splitlines()[-1])


def fake_cgroup(tmp_path, text, nested=None):
    root = tmp_path / "cg"
    root.mkdir()
    (root / "cpu.max").write_text(text)
    proc = tmp_path / "proc_self_cgroup"
    if nested:
        d = root
        for part in nested["path"].strip("/").split("/"):
            d = d / part
            d.mkdir()
        (d / "cpu.max").write_text(nested["text"])
        proc.write_text(f"0::{nested['path']}\n")
    else:
        proc.write_text("0::/\n")
    return {"PB_CGROUP_ROOT": str(root), "PB_PROC_SELF_CGROUP": str(proc)}


def test_quota_caps_the_pool_before_the_ranks_share_it(tmp_path):
    """16 CPUs of quota, 8 ranks on the node: 2 producer threads per rank, whatever the machine has."""
    env = fake_cgroup(tmp_path, "1600000 100000\n")
    env["LOCAL_WORLD_SIZE"] = "8"
    r = run_child(env)["res"]
    hw = r["hardware_threads"]
    assert r["cgroup_cpus"] == 16.0 and r["local_world_size"] == 8
    assert r["usable_cpus"] == min(16, hw, r["affinity_cpus"])
    assert r["host_threads"] == max(1, r["usable_cpus"] // 8)
    assert "cgroup quota 16.00" in r["rule"] and "8 rank(s)" in r["rule"]


def test_unlimited_quota_and_fractional_quota(tmp_path):
    r = run_child(fake_cgroup(tmp_path, "max 100000\n"))["res"]
    assert r["cgroup_cpus"] <= 0 and r["usable_cpus"] == min(r["hardware_threads"], r["affinity_cpus"])
    assert "cgroup quota none" in r["rule"]
    (tmp_path / "b").mkdir()
    r = run_child(fake_cgroup(tmp_path / "b", "250000 100000\n"))["res"]
    assert r["cgroup_cpus"] == 2.5 and r["usable_cpus"] == min(2, r["affinity_cpus"]) and r["host_threads"] == r["usable_cpus"]
    (tmp_path / "c").mkdir()
    r = run_child(fake_cgroup(tmp_path / "c", "50000 100000\n"))["res"]  # half a CPU: still one thread
    assert r["usable_cpus"] == 1 and r["host_threads"] == 1


def test_tightest_level_of_a_nested_cgroup_counts(tmp_path):
    env = fake_cgroup(tmp_path, "800000 100000\n", nested={"path": "/pods/job7", "text": "300000 100000\n"})
    assert run_child(env)["res"]["cgroup_cpus"] == 3.0
    (tmp_path / "b").mkdir()
    env = fake_cgroup(tmp_path / "b", "200000 100000\n", nested={"path": "/pods/job7", "text": "max 100000\n"})
    assert run_child(env)["res"]["cgroup_cpus"] == 2.0


def test_overrides(tmp_path):
    env = fake_cgroup(tmp_path, "400000 100000\n")
    r = run_child(dict(env, PB_HOST_THREADS="3", LOCAL_WORLD_SIZE="8"))["res"]
    assert r["host_threads"] == 3 and "PB_HOST_THREADS" in r["rule"]


def test_pipeline_starts_only_its_share_of_threads(tmp_path):
    """The producer pool of a pipeline created with host_threads <= 0 under a 3-CPU quota: 2 producers (one core is
    left to the thread that drives the device), and the placed members do not depend on it."""
    few = run_child(fake_cgroup(tmp_path, "300000 100000\n"), "pipeline")
    (tmp_path / "b").mkdir()
    many = run_child(fake_cgroup(tmp_path / "b", "max 100000\n"), "pipeline")
    assert few["res"]["host_threads"] == min(3, few["res"]["affinity_cpus"])
    assert few["checksums"] == many["checksums"] and len(set(few["checksums"])) == 40
    assert few["producers"] == max(1, few["res"]["host_threads"] - 1)
    assert many["producers"] == max(1, min(many["res"]["host_threads"], 128) - 1)


def test_numa_node_of_the_gpu_from_sysfs(tmp_path):
    """/sys/bus/pci/devices/<bus id>/numa_node + local_cpulist decide the pinning; numa_node -1 (VMs, single-socket
    hosts) means: do not pin."""
    from particlerobotsimulations_amd import host
    import ctypes as C
    L = host.lib()
    L.pbHostParseCpuList.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_int]
    mine = sorted(os.sched_getaffinity(0))
    buf = (C.c_int * 256)()
    n = L.pbHostParseCpuList(f"{mine[0]}-{mine[-1]},{mine[-1] + 1000}".encode(), buf, 256)
    assert n == len(mine) and list(buf[:n]) == mine           # restricted to the affinity mask
    assert L.pbHostParseCpuList(b"", buf, 256) == 0
    sysroot = tmp_path / "sys"
    dev = sysroot / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    (dev / "local_cpulist").write_text(f"{mine[0]}-{mine[-1]}\n")   # (a node the pool fits in; smaller: next test)
    env = {"PB_SYSFS_ROOT": str(sysroot), "PB_FAKE_PCI_BUS_ID": "0000:C1:00.0"}
    r = run_child(env)["res"]
    assert r["numa_node"] == 1 and r["numa_cpus"] == len(mine) and r["pin_producers"] == 1
    assert r["pci_bus_id"] == "0000:c1:00.0" and "pinned to the GPU's NUMA node" in r["rule"]
    assert run_child(dict(env, PB_PIN_PRODUCERS="0"))["res"]["pin_producers"] == 0
    (dev / "numa_node").write_text("-1\n")
    r = run_child(env)["res"]
    assert r["numa_node"] == -1 and r["pin_producers"] == 0 and "not pinned" in r["rule"]


def test_a_pinned_pool_never_exceeds_the_numa_node(tmp_path):
    """producers were pinned to the GPU's NUMA node but the pool was still sized from the whole machine (a
    lone rank on a two-socket host: 127 threads on 64 cores).  Now a lone rank whose share is larger than the node is
    not pinned; with several ranks per node the automatic share is clamped to the node's cores; an explicit thread
    count is honoured unpinned.  The pipeline's pool follows."""
    mine = sorted(os.sched_getaffinity(0))
    if len(mine) < 6:
        # (with two ranks per node the automatic share must EXCEED the two-core node for a clamp to happen: 6 // 2 = 3)
        pytest.skip("needs six usable CPUs")
    sysroot = tmp_path / "sys"
    dev = sysroot / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("0\n")
    (dev / "local_cpulist").write_text(f"{mine[0]}-{mine[1]}\n")   # a two-core node
    env = {"PB_SYSFS_ROOT": str(sysroot), "PB_FAKE_PCI_BUS_ID": "0000:c1:00.0"}
    env.update(fake_cgroup(tmp_path, "max 100000\n"))
    lone = run_child(env, "pipeline")
    r = lone["res"]
    assert r["numa_cpus"] == 2 and r["host_threads"] == min(len(mine), r["hardware_threads"]) > 2
    assert r["pin_producers"] == 0 and "larger than the GPU's NUMA node" in r["rule"]
    assert lone["producers"] == r["host_threads"] - 1
    shared = run_child(dict(env, LOCAL_WORLD_SIZE="2"), "pipeline")
    r = shared["res"]
    assert r["pin_producers"] == 1 and r["host_threads"] == 2 and "clamped" in r["rule"]
    assert shared["producers"] <= r["numa_cpus"]
    assert shared["checksums"] == lone["checksums"]
    r = run_child(dict(env, LOCAL_WORLD_SIZE="2", PB_HOST_THREADS="3"))["res"]
    assert r["host_threads"] == 3 and r["pin_producers"] == 0
    # a pool that fits is pinned as before
    (dev / "local_cpulist").write_text(f"{mine[0]}-{mine[-1]}\n")
    r = run_child(env)["res"]
    assert r["pin_producers"] == 1 and r["host_threads"] <= r["numa_cpus"]
