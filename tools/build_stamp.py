#!/usr/bin/env python3
"""Written by csrc/Makefile next to the libraries (lib/build_stamp.json): which sources the loaded kernels were built
from.  `kernel_sources_sha16` is a content hash of the files the per-step force kernels are compiled from;
tools/summarize_profile.py copies the stamp into profiles/latest_traffic*.json and bench.py compares it with the stamp
of the library it loaded, so a profile that predates a kernel change is visible in the line.  The commit is
informational (None outside a git checkout, e.g. on the GPU box)."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "particlerobotsimulations_amd", "csrc")
KERNEL_SOURCES = ("pb_force.hip", "pb_stream.hip", "pb_sweep.hpp", "pb_device.hpp", "pb_engine.hpp", "Makefile")


def kernel_sources_sha16():
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(name.encode() + b"\0")
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def git(*args):
    try:
        return subprocess.check_output(["git", "-C", ROOT, *args], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        return None


def stamp():
    head = git("rev-parse", "--short=12", "HEAD")
    dirty = git("status", "--porcelain", "--", "particlerobotsimulations_amd/csrc", "include")
    return {"kernel_sources_sha16": kernel_sources_sha16(), "commit": head,
            "commit_dirty": (bool(dirty) if dirty is not None else None), "kernel_sources": list(KERNEL_SOURCES)}


if __name__ == "__main__":
    out = os.path.join(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "particlerobotsimulations_amd", "lib"),
                       "build_stamp.json")
    with open(out, "w") as fh:
        json.dump(stamp(), fh)
        fh.write("\n")
