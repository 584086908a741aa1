#!/usr/bin/env python3
"""One command to (re-)pin Poseidon2: regenerate include/zkw_poseidon2_params.h (or load an alternative parameter module), rebuild the
oracle and run every reference-held known answer on it (tests/test_reference_fixtures.py, tests/test_oracle_recursion.py). The device
side is pinned by `pytest -m gpu tests/test_gpu_reference_kats.py tests/test_gpu_recursion.py` on a GPU box. No golden under tests/golden
depends on the permutation except the reference's own fixtures, so there is nothing else to regenerate.

    python tools/p2_pin.py                      # the committed parameters (tools/gen_poseidon2_params.py)
    python tools/p2_pin.py --params other.py    # a module defining RC (360 ints), INTERNAL_DIAG_SHIFTS, ... like gen_poseidon2_params.py
"""
import argparse
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--params", default=os.path.join(ROOT, "tools", "gen_poseidon2_params.py"))
    args = ap.parse_args()
    spec = importlib.util.spec_from_file_location("p2_params", args.params)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import gen_poseidon2_params as g  # the renderer

    for name in ("RC", "M4", "INTERNAL_DIAG_SHIFTS", "HALF_FULL_ROUNDS", "PARTIAL_ROUNDS", "INITIAL_EXTERNAL_LAYER", "PARTIAL_CONSTANT_INDEX"):
        if hasattr(mod, name):
            setattr(g, name, getattr(mod, name))
    open(g.HEADER, "w").write(g.render())
    print("wrote", os.path.normpath(g.HEADER))
    subprocess.check_call(["make", "-B", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    rc = subprocess.call([sys.executable, "-m", "pytest", "-q", os.path.join(ROOT, "tests", "test_reference_fixtures.py"),
                          os.path.join(ROOT, "tests", "test_oracle_recursion.py"), os.path.join(ROOT, "tests", "test_oracle_field_hash.py")], cwd=ROOT)
    print("PINNED: every reference-held known answer reproduced" if rc == 0 else "NOT pinned: see the failures above")
    return rc


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.exit(main())
