"""A committed rocprofv3 profile is quoted by bench.py only for the kernel the LOADED library launches (VERDICT r5
item 3: round 5's headline profiles predated an argument reorder of k_force and nothing noticed).  The library names
its kernels as the profiler does (pbForceFormKernelName: template arguments + argument types, built from the
instantiation's own type); tools/summarize_profile.py records that signature and the build stamp in
profiles/latest_traffic*.json; benchkit.matched_profile drops the counters when the signatures differ."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import benchkit as K  # noqa: E402
import build_stamp  # noqa: E402
import summarize_profile  # noqa: E402

import particlerobotsimulations_amd as pb  # noqa: E402

BOTH = {"flat": 1, "lanes_per_bot": 1, "attraction_sums": 1, "offsets64": 0}


def loaded_signature(form=BOTH):
    return [pb.force_form_kernel_name(i) for i, f in enumerate(pb.force_forms()) if f == form][0]


def test_the_library_names_every_form_as_the_profiler_would():
    names = [pb.force_form_kernel_name(i, payload) for i in range(len(pb.force_forms())) for payload in (0, 1)]
    assert len(set(names)) == len(names)            # every row x payload mode is its own kernel
    sig = loaded_signature()
    assert sig.startswith("k_force<false, true, 1, 1, false, true>(PbDevParams const*, HIP_vector_type<float, 4u> const*")
    assert sig.endswith(")") and "void" not in sig and "anonymous" not in sig
    # the trace's spelling of the same kernel reduces to it
    traced = "void (anonymous namespace)::" + sig
    assert summarize_profile.signature(traced) == sig


def test_a_profile_of_another_signature_is_dropped(tmp_path, monkeypatch):
    sig = loaded_signature()
    stamp = {"kernel_sources_sha16": build_stamp.kernel_sources_sha16(), "commit": "abc"}
    good = {"kernel_signature": sig, "hbm_bytes_per_launch": 7.0e7, "valu_insts_per_wave": 4000.0, "profile": "prof_x",
            "build": stamp}
    # round 5's staleness: the same template arguments, the arguments in another order
    stale = dict(good, kernel_signature=sig.replace("float*, unsigned int,", "unsigned int, float*,", 1))
    assert stale["kernel_signature"] != sig
    old = {k: v for k, v in good.items() if k != "kernel_signature"}      # a round-5 file: no signature at all
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(K, "ROOT", str(tmp_path))
    for name, rec in (("good.json", good), ("stale.json", stale), ("old.json", old)):
        (tmp_path / "profiles" / name).write_text(json.dumps(rec))
    tr, why = K.matched_profile("good.json", sig, stamp)
    assert why is None and tr["hbm_bytes_per_launch"] == 7.0e7 and tr["sources_match"] is True
    tr, why = K.matched_profile("good.json", sig, {"kernel_sources_sha16": "0" * 16})
    assert why is None and tr["sources_match"] is False            # same signature, kernels rebuilt from other sources
    for name in ("stale.json", "old.json", "absent.json"):
        tr, why = K.matched_profile(name, sig, stamp)
        assert tr is None and why, name
    assert "not of the loaded library's kernel" in K.matched_profile("stale.json", sig, stamp)[1]
    # and the roofline figures derived from a dropped profile are None, not another kernel's
    assert K.valu_of_datasheet(None, 10**6, 90.0) is None


def test_the_committed_profiles_name_their_kernel_and_build():
    """profiles/latest_traffic*.json as committed: signature + build stamp present (written by tools/profile.sh on this
    round's build); whether they still match the loaded library is what bench.py reports, not asserted here."""
    for which, form in (("latest_traffic_both_sums.json", BOTH), ("latest_traffic.json", dict(BOTH, attraction_sums=0))):
        rec = K.profiled_traffic(which)
        assert rec is not None, which
        if "kernel_signature" not in rec:       # a pre-round-6 file: bench.py drops it
            assert K.matched_profile(which, loaded_signature(form))[0] is None
            continue
        assert rec["kernel_signature"].startswith("k_force<") and rec["build"]["kernel_sources_sha16"]
        tr, why = K.matched_profile(which, loaded_signature(form))
        assert (tr is None) == (why is not None)


def test_build_stamp_is_written_next_to_the_libraries():
    st = K.loaded_build_stamp()
    assert st is not None and st["kernel_sources_sha16"] == build_stamp.kernel_sources_sha16()
