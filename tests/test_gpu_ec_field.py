"""GPU: the secp256k1 base-field arithmetic of the ECRecover accumulator chain (era_zkevm_test_harness_amd/csrc/ec_field.cuh) — the form with
a value in one lane (ecf) and the form with a limb per lane, DPP row shifts and ballot carry-lookahead (ecl) — against the host arithmetic of
include/zkw_ecrecover.h on edge values (0, 1, p - 1, c, runs of all-ones words) and seeded random ones, 174 x 174 pairs; Jacobian doubling and
mixed addition of the two forms against each other. tests/csrc_gpu/ec_field_test.hip is built with hipcc on the box."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def test_lane_form_and_register_form_equal_the_host_arithmetic(tmp_path):
    exe = str(tmp_path / "ec_field_test")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "csrc_gpu", "ec_field_test.hip"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout[-2000:] + r.stderr[-2000:]
    assert int(r.stdout.split()[1]) >= 174 * 174
