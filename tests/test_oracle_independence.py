"""The oracle is an INDEPENDENT restatement: it may share the generated layout tables and the record / format
headers with the library, but no header that holds semantics — what a lookup table computes, how a cycle's value tape of the ECRecover
circuit is evaluated, what relation an item states. And the product never touches the oracle."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALLOWED = re.compile(r"^(zkw_types\.h|zkw_poseidon2_params\.h|zkw_[a-z0-9_]+_spec\.h|zkw_netlist\.h|zkw_netlist_queue\.h|zkw_netlist_closed_form\.h|zkw_ecrecover_layout\.h)$")


def _includes(path):
    with open(path) as f:
        return re.findall(r'#include\s+"([^"]+)"', f.read())


def test_oracle_includes_only_format_and_table_headers():
    seen = set()
    for name in sorted(os.listdir(os.path.join(ROOT, "oracle"))):
        if not name.endswith((".c", ".h")):
            continue
        for inc in _includes(os.path.join(ROOT, "oracle", name)):
            if inc == "oracle.h":
                continue
            assert inc.startswith("../include/"), (name, inc)  # nothing from the package's csrc/
            base = os.path.basename(inc)
            assert ALLOWED.match(base), f"oracle/{name} includes {inc}: not a format / generated-table header"
            seen.add(base)
    assert "zkw_ecrecover_layout.h" in seen and "zkw_netlist.h" in seen
    # and the shared headers themselves include nothing that carries semantics
    for base in ("zkw_netlist.h", "zkw_ecrecover_layout.h", "zkw_netlist_queue.h", "zkw_netlist_closed_form.h"):
        for inc in _includes(os.path.join(ROOT, "include", base)):
            assert ALLOWED.match(os.path.basename(inc)), (base, inc)


def test_semantics_live_outside_the_shared_headers():
    with open(os.path.join(ROOT, "include", "zkw_netlist.h")) as f:
        nl = f.read()
    assert "nl_table_eval" not in nl and "nl_table_key" not in nl and "static inline" not in nl  # record types and macros only
    with open(os.path.join(ROOT, "include", "zkw_ecrecover_layout.h")) as f:
        lay = f.read()
    for fn in ("ec_eval_cycle", "ec_eval_segment", "ec_check_item", "ec_mulmod", "ec_mul_witness", "ec_build_fixed_tables", "ec_gl_mul"):
        assert fn not in lay, fn
    with open(os.path.join(ROOT, "include", "zkw_ecrecover.h")) as f:
        sem = f.read()
    assert "ec_eval_cycle" in sem and "ec_check_item" in sem  # the library's own: still there, just not in the oracle
    with open(os.path.join(ROOT, "oracle", "ecrecover_eval.c")) as f:
        own = f.read()
    assert "orc_ec_eval_cycle_own" in own and "big_divmod" in own and "big_invmod" in own


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "era_zkevm_test_harness_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        if "_obj" in dirpath or "_alt" in dirpath or "__pycache__" in dirpath:
            continue
        for name in files:
            if not name.endswith((".py", ".hip", ".cuh", ".h")):
                continue
            with open(os.path.join(dirpath, name)) as f:
                txt = f.read()
            code = "\n".join(ln for ln in txt.splitlines() if not ln.lstrip().startswith(("#", "//", "*", "/*")))
            code = re.sub(r'"""(?:.|\n)*?"""', "", code)  # docstrings may NAME the oracle (e.g. "checked against the oracle by ...")
            for needle in ("import pyoracle", "from oracle", "liboracle.so", "import oracle", "dlopen(\"oracle"):
                assert needle not in code, (name, needle)
