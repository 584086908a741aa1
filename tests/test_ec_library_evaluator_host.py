"""CPU: the library's own evaluator of the ECRecover EC section (include/zkw_ecrecover.h — the code k_ec_chain / k_ec_segments run) compiled
for the host, against the committed tape digests (tests/golden/ecrecover_tape_digests.json: all ~504 000 values of a cycle's tape on
successes, an idle cycle and every failure mode; the oracle and the generator's Python evaluator are pinned to the same digests in
tests/test_oracle_ecrecover_circuit.py). Two walks: program order, and the order of the kernels' fast form — PRE's MAIN items (what the
accumulator chain waits for), the other segments, PRE's remaining parts last and in reverse — which must give the same tape."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from tests.test_oracle_ecrecover_circuit import _cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("ec_host") / "libec_host.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "csrc_host", "ec_library_evaluator.c"), "-o", so])
    lib = C.CDLL(so)
    lib.lib_ec_eval_cycle.restype = C.c_uint32
    lib.lib_ec_tape_per_cycle.restype = C.c_uint32
    lib.lib_ec_ws_bytes.restype = C.c_uint32
    return lib


def _inputs(h, v, r, s):
    return np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in (h, v, r, s)), np.uint8).copy()


@pytest.mark.parametrize("order,ts", [(0, 1), (1, 1), (1, 8)])
def test_library_tape_equals_the_committed_digests(host_lib, order, ts):
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "ecrecover_tape_digests.json")))["sha256_of_tape"]
    n = host_lib.lib_ec_tape_per_cycle()
    for k, (c, want) in enumerate(zip(_cases(), golden)):
        inp = _inputs(*c)
        lane = k % ts  # (the cycle's place among the `ts` interleaved tapes)
        tape = np.full(n * ts, 0xDEAD, np.uint64)
        rc = host_lib.lib_ec_eval_cycle(inp.ctypes.data_as(C.c_void_p), C.c_void_p(tape.ctypes.data + 8 * lane), order, ts)
        assert rc == 0, (c, hex(rc))
        own = tape.reshape(n, ts)
        assert hashlib.sha256(np.ascontiguousarray(own[:, lane]).tobytes()).hexdigest() == want, c
        assert ts == 1 or (np.delete(own, lane, axis=1) == 0xDEAD).all()  # nothing written beside the cycle's own values


def test_workspace_fits_five_workgroups_on_a_cu(host_lib):
    """k_ec_segments holds one ec_ws per lane in LDS: 64 lanes x 5 workgroups within 160 KB, and an odd word stride between the lanes"""
    b = host_lib.lib_ec_ws_bytes()
    assert b % 4 == 0 and (b // 4) % 2 == 1 and 5 * 64 * b <= 160 * 1024
