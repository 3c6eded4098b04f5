"""Occupancy guards read from the CODE OBJECTS themselves (no GPU needed: hipcc cross-compiles gfx950 here):
the register counts and scratch sizes DESIGN.md section 3 argues with, taken from the `.vgpr_count` /
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
    vgpr, sgpr, lds, scratch = regs["k_force_stream<false, false, false>"]
    assert vgpr <= 64 and scratch == 0 and lds <= 12288, (vgpr, sgpr, lds, scratch)
    # ... and its flattened-walk form (round 6; 10 KB range queue + 10 KB contact lists: still 8 workgroups per CU)
    vgpr, sgpr, lds, scratch = regs["k_force_stream<false, false, true>"]
    assert vgpr <= 64 and scratch == 0 and lds <= 20480, (vgpr, sgpr, lds, scratch)
    # the form roofline.frac is priced on since round 5 (both magnitude sums; contact magnitudes parked in LDS like the
    # default form's)
    vgpr, sgpr, lds, scratch = regs["k_force<false, true, 1, 1, false, true>"]
    assert vgpr <= 64 and scratch == 0 and lds == 9216, (vgpr, sgpr, lds, scratch)


def test_no_force_kernel_spills_and_all_forms_are_there(regs):
    force = {k: v for k, v in regs.items() if k.startswith(("k_force<", "k_force_stream<", "k_resident<"))}
    # 17 rows of the forms table x payload + the stream kernel's eight (payload x sums kept x walk)
    assert sum(k.startswith("k_force<") for k in force) == 34 and sum(k.startswith("k_force_stream<") for k in force) == 8
    for k, (vgpr, sgpr, lds, scratch) in force.items():
        assert scratch == 0, f"{k} spills {scratch} bytes"
        assert vgpr <= 128, (k, vgpr)


def test_resident_kernels_fit_the_compute_unit(regs):
    """the one-lane-per-bot resident forms (members of 513 ... 1024 bots) park their contact magnitudes in
    LDS since round 5 -- 86 KB, ONE workgroup per CU of 160 KB; the multi-lane forms stay small.  Pinned so that a
    growth past the CU (launch failure) or a silent drop in the multi-lane forms' occupancy is seen."""
    res = {k: v for k, v in regs.items() if k.startswith("k_resident<")}
    assert len(res) == 32    # payload x fast x {8, 4, 2, 1} lanes x both-sums/dead-sum
    for k, (vgpr, sgpr, lds, scratch) in res.items():
        lanes = int(k.split(",")[2])
        assert scratch == 0 and vgpr <= 128, (k, vgpr, scratch)      # 1024 lanes per workgroup: 4 waves per SIMD
        if lanes == 1:
            assert lds == 86064 and lds <= 160 * 1024, (k, lds)
        else:
            assert lds == {2: 24624, 4: 12336, 8: 6192}[lanes], (k, lds)
