"""The committed layout contracts include/zkw_*_circuit_spec.h are exactly what their generators emit (nobody edits a
generated header by hand, nobody changes a generator without regenerating): every generator is run against a scratch copy
of the tree and its output compared byte for byte. The netlist generators also re-check their netlists against plain
Keccak-f[1600] / hashlib.sha256 on the way."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GENERATORS = {"gen_ram_circuit.py": "zkw_ram_circuit_spec.h", "gen_decommit_sorter_circuit.py": "zkw_decommit_sorter_circuit_spec.h",
              "gen_events_sorter_circuit.py": "zkw_events_sorter_circuit_spec.h", "gen_log_demux_circuit.py": "zkw_log_demux_circuit_spec.h",
              "gen_storage_sorter_circuit.py": "zkw_storage_sorter_circuit_spec.h", "gen_keccak_circuit.py": "zkw_keccak_circuit_spec.h",
              "gen_sha256_circuit.py": "zkw_sha256_circuit_spec.h"}
DERIVED = ("oracle/code_decommitter_circuit.c", "era_zkevm_test_harness_amd/csrc/code_decommitter_circuit_kernels.cuh")


@pytest.mark.parametrize("gen,header", sorted(GENERATORS.items()))
def test_generated_header_is_current(gen, header, tmp_path):
    tools = tmp_path / "tools"
    (tmp_path / "include").mkdir()
    shutil.copytree(os.path.join(ROOT, "tools"), tools, ignore=shutil.ignore_patterns("__pycache__", "probe_*", "p2_*", "ubench_*", "*.hip"))
    r = subprocess.run([sys.executable, str(tools / gen)], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-1500:]
    got = (tmp_path / "include" / header).read_bytes()
    assert got == open(os.path.join(ROOT, "include", header), "rb").read(), f"{header} is stale: run python tools/{gen}"


def test_code_decommitter_headers_and_derived_sources_are_current(tmp_path):
    """gen_sha256_circuit.py also emits the 18-lookups-per-row spec of the CodeDecommitter; its oracle and kernels are
    renamed copies of the SHA-256 ones (tools/gen_code_decommitter_sources.py)"""
    tools = tmp_path / "tools"
    (tmp_path / "include").mkdir()
    shutil.copytree(os.path.join(ROOT, "tools"), tools, ignore=shutil.ignore_patterns("__pycache__", "probe_*", "p2_*", "ubench_*", "*.hip"))
    for rel in ("oracle/sha256_circuit.c", "era_zkevm_test_harness_amd/csrc/sha256_circuit_kernels.cuh"):
        os.makedirs(tmp_path / os.path.dirname(rel), exist_ok=True)
        shutil.copy(os.path.join(ROOT, rel), tmp_path / rel)
    for script in ("gen_sha256_circuit.py", "gen_code_decommitter_sources.py"):
        r = subprocess.run([sys.executable, str(tools / script)], capture_output=True, text=True, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-1500:]
    for rel in ("include/zkw_code_decommitter_circuit_spec.h",) + DERIVED:
        assert (tmp_path / rel).read_bytes() == open(os.path.join(ROOT, rel), "rb").read(), f"{rel} is stale"


@pytest.mark.parametrize("header,prefix", [("zkw_sha256_circuit_spec.h", "SC"), ("zkw_code_decommitter_circuit_spec.h", "DC")])
def test_netlist_lookups_are_grouped_by_table_with_padding_last(header, prefix):
    """k_sc_hist / k_dc_hist count multiplicities by reading a table's rows as one run per cycle and skip the padding by
    position (sha256_circuit_kernels.cuh, sc_hist_plan): the committed specs must keep that shape"""
    import re
    text = open(os.path.join(ROOT, "include", header)).read()
    per_row = int(re.search(rf"#define {prefix}_LOOKUPS_PER_ROW (\d+)", text).group(1))
    const0 = int(re.search(rf"#define {prefix}_REF_CONST (0x[0-9A-Fa-f]+)", text).group(1), 16)
    body = text[text.index(f"#define {prefix}_OPS_INIT"):text.index(f"#define {prefix}_GATES_INIT")]
    ops = [tuple(int(x) for x in m) for m in re.findall(r"\{(\d+), (\d+), (\d+)\}", body)]
    assert len(ops) == int(re.search(rf"#define {prefix}_NUM_OPS (\d+)", text).group(1)) and len(ops) % per_row == 0
    for j in range(1, len(ops)):
        assert ops[j][0] >= ops[j - 1][0]
        if ops[j][0] != ops[j - 1][0]:
            assert j % per_row == 0
        elif ops[j - 1][1:] == (const0, const0):
            assert ops[j][1:] == (const0, const0)
