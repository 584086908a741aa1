"""Cells of the closed-form section of a netlist circuit (include/zkw_netlist_closed_form.h) for the tamper-parity tests: flags, words of
every part, tie cells, permutation inputs / middles / outputs of a commitment sponge and of the compact form, the PI row, unused cells
of the section, its lookup columns, random cells of its rows."""


def _cell_or_none(oracle, ct, cycles, group, tie, k):
    import ctypes as C
    import numpy as np

    out = np.zeros(2, np.uint64)
    f = oracle.lib().orc_nlcf_cell
    f.restype = C.c_int
    rc = f(C.c_int(ct), C.c_uint32(cycles), C.c_int(6), C.c_uint32((group << 26) | (tie << 14) | k), C.c_void_p(out.ctypes.data))
    return (int(out[0]), int(out[1])) if rc == 0 else None


def closed_form_cells(oracle, ct, cycles, rng, n_random=10):
    g = oracle.nlcf_geometry(ct, cycles)
    geo = oracle.nl_geometry(ct)
    G = geo["general"]
    cc = lambda *a, **k: oracle.nlcf_cell(ct, cycles, *a, **k)  # noqa: E731
    perms = g["perms"]
    cells = [cc("pi", 1), cc("flag", 0), cc("flag", 1), cc("oi", 3), cc("oo", g["n_oo"] - 1),
             cc("p2", 3), cc("p2", 9), cc("p2", 60), cc("p2", 127), cc("p2", 130 + 11), cc("p2", 130 * (perms - 3) + 1), cc("p2", 130 * (perms - 2) + 8),
             cc("p2", 130 * (perms - 1) + 119), cc("p2", 130 * (perms - 1) + 90)]
    if g["n_fi"]:
        cells += [cc("fi", 1), cc("fi", g["n_fi"] // 2), cc("fo", g["n_fo"] - 2), cc("fo", 0)]
    gi = 0
    while gi < 64:  # every group: a / b cells of a tie, a register digit of another (group shapes differ: probe the layout)
        got = [_cell_or_none(oracle, ct, cycles, gi, tie, k) for tie, k in ((0, 0), (0, 1), (0, 2), (1, 0), (2, 2), (0, 3))]
        if got[0] is None:
            break
        cells += [x for x in got if x is not None]
        gi += 1
    header_rows = -(-g["header_cells"] // G)
    if g["header_cells"] % G:
        cells.append((G - 1, g["first_row"] + header_rows - 1))  # an unused cell of the header block's last row
    if 130 % G:
        cells.append((G - 1, g["first_row"] + header_rows + -(-130 // G) - 1))  # ... of a permutation's last row
    cells.append((G + 1, g["first_row"] + 1))  # a lookup cell of a section row
    cells.append((int(rng.integers(0, G)), g["rows_used"] + 2))  # below the section
    cells += [(int(rng.integers(0, G)), int(rng.integers(g["first_row"], g["rows_used"]))) for _ in range(n_random)]
    return cells
