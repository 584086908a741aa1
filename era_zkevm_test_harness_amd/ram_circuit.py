"""Host-side helpers for RAMPermutation synthesis (zkw trace v2, include/zkw_ram_circuit_spec.h)."""
import numpy as np

import os
import re

ROWS_PER_CYCLE = 6
N_COLS = 149
N_BOUNDARY_ROWS = 40  # BND_IN, BND_OUT, PI + the closed-form section (sponges of the closed-form input, selections, challenges)


def spec_macros(header="zkw_ram_circuit_spec.h", prefix="RC"):
    """the integer macros of a generated spec header ({name without prefix: value}): row offsets (ROWOFF_*), cells (<ROW>_<var>) ..."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", header)
    out = {}
    with open(path) as f:
        for m in re.finditer(rf"^#define {prefix}_(\w+) (\d+)\b", f.read(), re.M):
            out[m.group(1)] = int(m.group(2))
    return out


REGION_ALIGN = 64


def region_stride(capacity: int) -> int:
    """RC_REGION_STRIDE: rows per region; every region starts on a 64-row boundary."""
    return (capacity + REGION_ALIGN - 1) // REGION_ALIGN * REGION_ALIGN


def boundary_row(capacity: int) -> int:
    """RC_BOUNDARY_ROW: first of the boundary rows (BND_IN, BND_OUT, PI, then the closed-form section)."""
    return ROWS_PER_CYCLE * region_stride(capacity)


def min_rows(capacity: int) -> int:
    """RC_MIN_ROWS: rows a trace needs for a given per-circuit capacity."""
    return boundary_row(capacity) + N_BOUNDARY_ROWS


def smoke(ctx, witness, oracle_out, pyoracle):
    """Used by __graft_entry__.smoke(): synthesise instance 0 on the GPU, compare with the oracle's
    trace cell by cell, and run the GPU satisfiability check."""
    from . import native

    capacity, n_rows = 1024, 1 << 13
    t = native.Trace(ctx, n_rows, 1)
    ctx.synthesize_ram(witness, t, 0, 1, 0)
    got = t.get(0)
    exp = pyoracle.ram_synthesize(oracle_out, 0, capacity, n_rows)
    assert np.array_equal(got, exp), "GPU trace differs from the oracle trace"
    bad, first = ctx.check_if_satisfied_ram(t, 0, capacity)
    assert bad == 0, first
    t.free()
