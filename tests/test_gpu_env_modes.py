"""GPU: the library's environment switches keep working — ZKW_ALLOC_CACHE=0 (every buffer straight from / back to the HIP
runtime), a host-chosen GPU_MAX_HW_QUEUES, zkw_trim_caches between blocks. Each mode runs a small block end to end in a
fresh process (the switches are read when the library / the HIP runtime initialise) and compares its public inputs with
the default mode's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = r"""
import json, sys
sys.path.insert(0, %r)
import numpy as np
from era_zkevm_test_harness_amd import native as nv, synthetic
caps = {2: 5, 3: 7, 4: 64, 5: 3, 6: 4, 7: 2, 8: 1000, 9: 40, 11: 16, 12: 9, 13: 48}
out = {}
for rep in range(2):
    B = nv.Block(0, synthetic.block_after_vm(seed=4), caps)
    bad = []
    n = B.synthesize(1 << 18, ring_slots=2, callback=lambda t, i, tr, s, pi: bad.append(B.check_satisfied(t, tr, s)[0]))
    out = {"n": n, "bad": int(sum(bad)), "pi": {str(t): B.public_inputs(t).tolist() for t in (2, 3, 5, 6, 8, 13)}}
    B.free()
    nv.trim_caches()
print("RESULT " + json.dumps(out))
""" % ROOT


def _run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_modes_agree():
    base = _run({})
    assert base["n"] > 20 and base["bad"] == 0
    for env in ({"ZKW_ALLOC_CACHE": "0"}, {"GPU_MAX_HW_QUEUES": "4"}, {"GPU_MAX_HW_QUEUES": "16", "ZKW_ALLOC_CACHE": "0"}):
        got = _run(env)
        assert got == base, env
