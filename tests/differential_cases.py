"""The case generator of tests/test_gpu_differential.py: adversarially SHAPED queues for every witness builder, drawn from a seed.

A case is (builder, seed) -> (inputs, capacity): the shape knobs — queue length around the instance capacity (capacity - 1, capacity,
capacity + 1, one item, many instances), a single cell / page touched by everything or every item its own, rollback-dense or rollback-free log
queues, one hash or all-distinct hashes, timestamps pushed up to 2^32 - 1, empty types — are drawn by the case's own generator, so the corpus
is the list of (builder, seed) pairs and their digests: tests/golden/differential_corpus.json (made by `python tests/differential_cases.py`,
which also prints the shape statistics). No GPU and no oracle are needed to generate a case."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from era_zkevm_test_harness_amd import synthetic  # noqa: E402

BUILDERS = ("ram", "decommit_sorter", "events_sorter", "log_demux", "storage_sorter", "decommitter", "precompile", "linear_hasher")
CASES_PER_BUILDER = 200
U32_MAX = 0xFFFFFFFF


def _pick(rng, options):
    return options[int(rng.integers(0, len(options)))]


def _length_around(rng, capacity):
    """queue lengths that sit on the instance boundaries"""
    return int(_pick(rng, (1, 2, capacity - 1, capacity, capacity + 1, 2 * capacity - 1, 2 * capacity, 2 * capacity + 1, 3 * capacity + int(rng.integers(0, capacity)))))


def _lift_timestamps(rng, ts):
    """with probability 1/3 the (ordered) timestamps are moved so that the largest is 2^32 - 1"""
    if ts.size and rng.integers(0, 3) == 0:
        shift = U32_MAX - int(ts.max())
        return (ts.astype(np.uint64) + np.uint64(shift)).astype(np.uint32)
    return ts


def case(builder, seed):
    rng = np.random.default_rng([seed, BUILDERS.index(builder)])
    if builder == "ram":
        capacity = int(_pick(rng, (4, 7, 16, 33, 64, 128)))
        n = max(1, _length_around(rng, capacity))
        shape = _pick(rng, ("one_cell", "one_page", "every_item_its_own_cell", "few_cells", "ordinary"))
        pages, indices = {"one_cell": (1, 1), "one_page": (1, 64), "every_item_its_own_cell": (1 << 12, 1 << 12), "few_cells": (2, 3), "ordinary": (16, 32)}[shape]
        q = synthetic.ram_trace(n, seed=seed + 11, pages=pages, indices=indices, read_fraction=float(_pick(rng, (0.0, 0.5, 0.95))),
                                ptr_fraction=float(_pick(rng, (0.0, 0.3))))
        if rng.integers(0, 3) == 0:
            q["timestamp"] = q["timestamp"] // 4  # many equal (cell, timestamp) keys: the order must be the stable sort's
        q["timestamp"] = _lift_timestamps(rng, q["timestamp"])
        k = int(rng.integers(0, min(n, 4) + 1))  # bootloader-heap writes at timestamp 0 (non-deterministic queries)
        q["page"][:k] = 10
        q["timestamp"][:k] = 0
        q["rw_flag"][:k] = 1
        return {"q": q, "capacity": capacity, "nondet": k, "shape": shape}
    if builder == "decommit_sorter":
        capacity = int(_pick(rng, (3, 5, 16, 40)))
        n = max(1, _length_around(rng, capacity))
        shape = _pick(rng, ("one_hash", "all_distinct", "few", "ordinary"))
        hashes = {"one_hash": 1, "all_distinct": n, "few": min(n, 3), "ordinary": max(1, n // 3)}[shape]
        q = synthetic.decommit_trace(n, hashes, seed=seed + 5)
        q["timestamp"] = _lift_timestamps(rng, q["timestamp"])
        return {"q": q, "capacity": capacity, "dedup_in": bool(rng.integers(0, 2)), "shape": shape}
    if builder == "events_sorter":
        capacity = int(_pick(rng, (4, 7, 16, 31)))
        nf = max(1, _length_around(rng, capacity) // 2)
        frac = float(_pick(rng, (0.0, 0.3, 0.9, 1.0)))  # 1.0: every event rolled back (rollback-dense), 0.0: none
        q = synthetic.events_trace(nf, frac, seed=seed + 3)
        q["timestamp"] = _lift_timestamps(rng, q["timestamp"])
        return {"q": q, "capacity": capacity, "result_in": bool(rng.integers(0, 2)), "shape": "rollback_%.1f" % frac}
    if builder == "log_demux":
        capacity = int(_pick(rng, (4, 9, 16, 64)))
        n = max(1, _length_around(rng, capacity))
        q = synthetic.mixed_log_queue(n, seed=seed + 7)
        shape = _pick(rng, ("ordinary", "one_output", "no_precompiles"))
        if shape == "one_output":  # every record goes to the same output queue: the other five are empty types
            q[:] = q[int(rng.integers(0, n))]
            q["timestamp"] = np.arange(1, n + 1, dtype=np.uint32)
        elif shape == "no_precompiles":
            keep = q["aux_byte"] != 3
            if keep.any():
                q = q[keep].copy()
        q["timestamp"] = _lift_timestamps(rng, q["timestamp"])
        return {"q": q, "capacity": capacity, "shape": shape}
    if builder == "storage_sorter":
        capacity = int(_pick(rng, (4, 9, 16, 40)))
        n = max(1, _length_around(rng, capacity))
        shape = _pick(rng, ("one_cell", "every_access_its_own_cell", "rollback_dense", "reads_only", "ordinary"))
        cells = {"one_cell": 1, "every_access_its_own_cell": n}.get(shape, max(1, n // 5))
        q = synthetic.storage_trace(n, cells, seed=seed + 13, p_read={"reads_only": 1.0}.get(shape, 0.35), p_rollback={"rollback_dense": 0.9, "reads_only": 0.0}.get(shape, 0.2))
        q["timestamp"] = _lift_timestamps(rng, q["timestamp"])
        return {"q": q, "capacity": capacity, "shape": shape}
    if builder == "decommitter":
        capacity = int(_pick(rng, (1, 4, 7, 16, 2845)))
        n_req = int(_pick(rng, (1, 2, 5, 12, 40)))
        return {"n_req": n_req, "capacity": capacity, "seed": seed + 17, "shape": "requests_%d" % n_req}
    if builder == "precompile":
        kind = int(rng.integers(0, 3))
        capacity = int(_pick(rng, (1, 3, 7, 100000)))
        n_req = int(_pick(rng, (0, 1, 2, 9, 40)))
        req, mq = synthetic.precompile_trace(kind, n_req, seed=seed + 19, max_rounds=int(_pick(rng, (1, 2, 6))))
        return {"kind": kind, "req": req, "mq": mq, "capacity": capacity, "shape": "kind_%d_requests_%d" % (kind, n_req)}
    if builder == "linear_hasher":
        n = int(_pick(rng, (0, 1, 2, 3, 17, 135, 136, 137, 400)))  # 88-byte records against the 136-byte Keccak rate
        q = synthetic.random_log_queries(n, seed=seed + 23)
        return {"q": q, "shape": "messages_%d" % n}
    raise KeyError(builder)


def digest(c):
    h = hashlib.sha256()
    for k in sorted(c):
        v = c[k]
        h.update(k.encode())
        h.update(v.tobytes() if isinstance(v, np.ndarray) else repr(v).encode())
    return h.hexdigest()[:16]


def corpus():
    return [(b, 1000 * BUILDERS.index(b) + k) for b in BUILDERS for k in range(CASES_PER_BUILDER)]


if __name__ == "__main__":
    out, shapes = [], {}
    for b, seed in corpus():
        c = case(b, seed)
        out.append({"builder": b, "seed": seed, "digest": digest(c)})
        shapes.setdefault(b, {}).setdefault(c["shape"], 0)
        shapes[b][c["shape"]] += 1
    path = os.path.join(ROOT, "tests", "golden", "differential_corpus.json")
    with open(path, "w") as f:
        json.dump({"cases_per_builder": CASES_PER_BUILDER, "shapes": shapes, "cases": out}, f, indent=0)
    print(json.dumps(shapes, indent=1))
    print("wrote", path, len(out), "cases")
