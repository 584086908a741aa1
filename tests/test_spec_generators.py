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


def test_item_lists_of_the_ecrecover_segments():
    """tools/gen_ecrecover_circuit.py split_segment: a segment type's items come as MAIN, MULS, LEAVES... — the sizes add up to the type's
    items, MAIN holds everything the state a segment leaves (PRE: and the globals) descends from, the leaves hold no MUL row and no hint (the
    kernel that walks them has no 256-bit workspace), no two lists of leaves write a common tape value, and PRE's square root is an item of
    its MAIN list (k_ec_chain evaluates it between the items before and after it)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_ecrecover_circuit as g

    spec = g.EcSpec()
    for st in spec.types:
        assert sum(st.parts) == len(st.items) and len(st.parts) >= 3 and all(n > 0 for n in st.parts[2:])
        rw = [g.item_reads_writes(it) for it in st.items]
        main, muls = st.parts[0], st.parts[1]
        written_by_main = set().union(*[rw[i][1] for i in range(main)])
        assert ({r.a for r in st.out} | set(st.globs)) <= written_by_main
        assert all(rw[i][0] <= written_by_main for i in range(main))  # MAIN reads nothing the other lists write
        assert not any(st.items[i]["k"] in (g.I_MUL, g.I_HINT) for i in range(main + muls, len(st.items)))
        lists, at = [], main + muls
        for n in st.parts[2:]:
            lists.append(set().union(*[rw[i][1] for i in range(at, at + n)]))
            at += n
        assert all(not (lists[a] & lists[b]) for a in range(len(lists)) for b in range(a))
        seen = set()
        for i, (rd, wr) in enumerate(rw):  # an item follows what it reads
            assert rd <= seen, (st.name, i)
            seen |= wr
    pre = spec.types[0]
    sq = [i for i, it in enumerate(pre.items) if it["k"] == g.I_HINT and it["hint"] == g.H_SQRT]
    assert len(sq) == 1 and sq[0] < pre.parts[0]
    hdr = open(os.path.join(ROOT, "include", "zkw_ecrecover_ec_spec.h")).read()
    assert f"#define EC_PRE_SQRT_ITEM {sq[0]}\n" in hdr
    assert "#define EC_PART_ITEMS_INIT {" + ", ".join("{" + ", ".join(str(x) for x in st.parts + [0] * (max(len(t.parts) for t in spec.types) - len(st.parts))) + "}" for st in spec.types) + "}" in hdr
