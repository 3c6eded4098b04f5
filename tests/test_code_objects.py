"""Occupancy guards read from the CODE OBJECTS themselves (no GPU needed: hipcc cross-compiles gfx950 here):
the register counts and scratch sizes DESIGN.md section 5 argues with, taken from the `.vgpr_count` /
`.private_segment_fixed_size` metadata of csrc/build/*.o exactly as tools/summarize_profile.py prints them
(rocprofv3's own VGPR column is the allocation granule).  A change that pushes the headline kernel
past 64 VGPRs (8 -> 7 waves per SIMD) or makes any force kernel spill fails here, before it is measured."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def regs():
    import summarize_profile
    r = summarize_profile.code_object_registers()
    if not r:
        pytest.skip("no csrc/build/*.o (libraries came prebuilt without their objects)")
    return {k: tuple(int(x) if str(x).isdigit() else x for x in v) for k, v in r.items()}


def test_headline_and_stream_kernels_keep_eight_waves_per_simd(regs):
    vgpr, sgpr, lds, scratch = regs["k_force<false, true, 1, 1, false, false>"]
    assert vgpr <= 64 and scratch == 0 and lds == 9216, (vgpr, sgpr, lds, scratch)
    vgpr, sgpr, lds, scratch = regs["k_force_stream<false, false>"]
    assert vgpr <= 64 and scratch == 0 and lds <= 12288, (vgpr, sgpr, lds, scratch)
    # the form roofline.frac is priced on since round 5 (both magnitude sums; contact magnitudes parked in LDS like the
    # default form's)
    vgpr, sgpr, lds, scratch = regs["k_force<false, true, 1, 1, false, true>"]
    assert vgpr <= 64 and scratch == 0 and lds == 9216, (vgpr, sgpr, lds, scratch)


def test_no_force_kernel_spills_and_all_forms_are_there(regs):
    force = {k: v for k, v in regs.items() if k.startswith(("k_force<", "k_force_stream<", "k_resident<"))}
    # 17 rows of the forms table x payload + the stream kernel's four
    assert sum(k.startswith("k_force<") for k in force) == 34 and sum(k.startswith("k_force_stream<") for k in force) == 4
    for k, (vgpr, sgpr, lds, scratch) in force.items():
        assert scratch == 0, f"{k} spills {scratch} bytes"
        assert vgpr <= 128, (k, vgpr)
