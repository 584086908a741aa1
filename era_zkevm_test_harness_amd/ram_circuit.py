"""Host-side helpers for RAMPermutation synthesis (zkw trace v1, include/zkw_ram_circuit_spec.h)."""
import numpy as np

ROWS_PER_CYCLE = 6
N_COLS = 149
N_BOUNDARY_ROWS = 3


def min_rows(capacity: int) -> int:
    """RC_MIN_ROWS: rows a trace needs for a given per-circuit capacity."""
    return ROWS_PER_CYCLE * capacity + N_BOUNDARY_ROWS


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
