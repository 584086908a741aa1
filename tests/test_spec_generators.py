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
              "gen_sha256_circuit.py": "zkw_sha256_circuit_spec.h", "gen_poseidon2_params.py": "zkw_poseidon2_params.h",
              "gen_ecrecover_circuit.py": "zkw_ecrecover_circuit_spec.h zkw_ecrecover_ec_spec.h"}
# (gen_sha256_circuit.py / gen_keccak_circuit.py each emit a second header — CodeDecommitter, L1MessagesHasher — from the same netlist;
#  tests/test_oracle_netlist_circuits.py::test_committed_specs_are_current_and_self_checked covers all four)
@pytest.mark.parametrize("gen,header", sorted(GENERATORS.items()))
def test_generated_header_is_current(gen, header, tmp_path):
    tools = tmp_path / "tools"
    (tmp_path / "include").mkdir()
    shutil.copytree(os.path.join(ROOT, "tools"), tools, ignore=shutil.ignore_patterns("__pycache__", "probe_*", "p2_*", "ubench_*", "*.hip"))
    pkg = tmp_path / "era_zkevm_test_harness_amd"  # (the ECRecover generator checks its netlist against the package's plain-integer secp256k1)
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    shutil.copy(os.path.join(ROOT, "era_zkevm_test_harness_amd", "secp256k1.py"), pkg / "secp256k1.py")
    r = subprocess.run([sys.executable, str(tools / gen)], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-1500:]
    for h in header.split():
        got = (tmp_path / "include" / h).read_bytes()
        assert got == open(os.path.join(ROOT, "include", h), "rb").read(), f"{h} is stale: run python tools/{gen}"
