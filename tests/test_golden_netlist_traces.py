"""CPU: the oracle's netlist traces (types 5, 13, 6, 3) on fixed seeds hash to the committed digests
(tests/golden/netlist_trace_digests.json, made by tests/golden/make_netlist_digests.py with the four Poseidon2-dependent
public-input cells zeroed): a silent change of a generator, a layout or a fill shows up here even when GPU and oracle
change together. The GPU is tied to the same traces cell for cell by tests/test_gpu_netlist_circuits.py."""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def test_netlist_traces_match_the_committed_digests(oracle):
    spec = importlib.util.spec_from_file_location("make_netlist_digests", os.path.join(HERE, "golden", "make_netlist_digests.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(HERE, "golden", "netlist_trace_digests.json")))
    got = mod.cases()
    assert got.keys() == want.keys()
    for k in want:
        assert got[k] == want[k], k
