"""Generates tests/golden/fma_bracket/oracle_builds.json: the FMA / __powf bracket (tests/fma_bracket.py)
of the oracle's own source on every BASELINE.json configuration at an oracle-affordable size.

    python tests/golden/make_fma_bracket.py            # all cases, ~2 min on 8 cores

The numbers are produced by the three builds of oracle/pb_oracle.c that oracle/Makefile defines (gcc
of this image, x86-64 with FMA); nothing of /root/reference is read.  tests/test_fma_bracket.py
re-measures the cheap cases and holds them to this file; tests/test_gpu_fma_bracket.py measures the
GPU's streamlined kernel on the same windows."""
import json
import os
import platform
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import orclib as orc  # noqa: E402
import fma_bracket as fb  # noqa: E402


def main():
    def factory_for(name):
        variants = orc.BRACKET_VARIANTS + (orc.ORDER_VARIANT,)
        if name in fb.DEVPOWF_CASES:  # the device-powf members only differ where there are obstacles
            variants += orc.DEVPOWF_VARIANTS
        return lambda P: [fb.OracleCandidate(orc, P, v) for v in variants]

    names = sys.argv[1:] or list(fb.CASES)
    results = {}
    for name in names:
        t = time.time()
        results[name] = fb.measure_case(orc, name, factory_for(name))
        for line in fb.format_rows(results[name]):
            print(line)
        print(f"  ({time.time() - t:.1f} s)", flush=True)
    out = {
        "generator": "tests/golden/make_fma_bracket.py",
        "compiler": subprocess.check_output(["gcc", "--version"], text=True).splitlines()[0],
        "libc": " ".join(platform.libc_ver()),
        "machine": platform.machine(),
        "teacher": "oracle/libpb_oracle.so (gcc -O2 -ffp-contract=off): the oracle",
        "candidates": {"fma": "oracle/libpb_oracle_fma.so: kernel functions with fp-contract=fast + FMA",
                       "fma_powf": "oracle/libpb_oracle_fma_powf.so: that + exp2f(2*log2f(x)) at impl.cuh:586,589",
                       "order": "oracle/libpb_oracle_order.so: the oracle's own terms, a bot's contact terms added after its "
                                "last candidate (the order of additions of the product's two-pass tolerance kernel)",
                       "devpowf": "oracle/libpb_oracle_devpowf.so (obstacle cases only): the device powf sites of the "
                                  "obstacle and shadow tests (impl.cuh:214-229, 704-705, 719, 757-779) as "
                                  "exp2f(y*log2f(x)); nothing else changed",
                       "devpowf_ulp": "oracle/libpb_oracle_devpowf_ulp.so: the same sites, the correctly rounded result "
                                      "moved by -1 / 0 / +1 ulp by a hash of the argument",
                       "cuda_like": "oracle/libpb_oracle_cuda_like.so: fma + the __powf model + the device-powf model"},
        "summary": {n: fb.summarise(r) for n, r in results.items()},
        "cases": results,
    }
    path = os.path.join(HERE, "fma_bracket", "oracle_builds.json")
    if sys.argv[1:] and os.path.exists(path):  # partial regeneration keeps the other cases
        old = json.load(open(path))
        old["cases"].update(out["cases"])
        old["summary"].update(out["summary"])
        out["cases"], out["summary"] = old["cases"], old["summary"]
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
