"""No kernel of libzkw may use scratch (private-segment) memory: every HSA queue that ever ran such a kernel keeps
scratch-per-lane x every wave slot of the chip (1.8 GB for 3400 B per lane), and with the 32 hardware queues that
zkw_blocks_run keeps busy that cost 29 GB of HBM and intermittent HSA_STATUS_ERROR_OUT_OF_RESOURCES aborts (DESIGN.md
3.14). hipcc cross-compiles here; the check reads the compiler's own resource report."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "era_zkevm_test_harness_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src", ["zkw_api.hip", "zkw_sorters.hip", "zkw_precompiles.hip", "zkw_setup.hip", "zkw_block.hip", "zkw_commit.hip", "zkw_batch.hip"])
def test_no_kernel_uses_scratch(src, tmp_path):
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, src), "-o",
                        str(tmp_path / "x.o"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    assert len(names) == len(scratch)
    if src in ("zkw_api.hip", "zkw_sorters.hip", "zkw_precompiles.hip"):
        assert len(names) > 25, names  # (the report is really the kernels': each of these units launches dozens)
    bad = {n: s for n, s in zip(names, scratch) if s}
    assert not bad, f"kernels with scratch memory: {bad}"
