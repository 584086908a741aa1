"""The C ABI from plain C: examples/ram_block.c compiles against include/zkw.h + libzkw.so, fails loudly without a GPU and
runs the RAM path end to end (builder, synthesis, checker, public inputs, recursion queue) on one."""
import os
import subprocess

import pytest

ROOT = os.path.join(os.path.dirname(__file__), "..")
LIBDIR = os.path.abspath(os.path.join(ROOT, "era_zkevm_test_harness_amd"))


@pytest.fixture(scope="module")
def binary(tmp_path_factory):
    from era_zkevm_test_harness_amd import build

    build.build()
    out = str(tmp_path_factory.mktemp("c_example") / "ram_block")
    subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "ram_block.c"),
                    "-o", out, "-L" + LIBDIR, "-lzkw", "-Wl,-rpath," + LIBDIR], check=True)
    return out


def test_compiles_and_fails_loudly_without_a_gpu(binary):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([binary], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_runs_on_the_gpu(binary):
    r = subprocess.run([binary, "5000", "1024", "14"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "5 RAMPermutation instances" in r.stdout and r.stdout.count("satisfied") == 5 and "NOT SATISFIED" not in r.stdout
    assert r.stdout.strip().endswith("ok")
