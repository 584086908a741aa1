import sys
import numpy as np
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native as nv
ctx = nv.Context(0)
n_rows = 1 << 16
def attempt(name, fn):
    try:
        print(name, fn())
    except Exception as e:
        print(name, "FAILED:", str(e)[:160])
def ram():
    w = ctx.compute_ram_circuit_snapshots(np.zeros(0, nv.MEM_QUERY), 1000, 0)
    t = nv.Trace(ctx, n_rows, 1); ctx.synthesize_ram(w, t, 0, 1, 0); return w.num_instances, ctx.check_if_satisfied_ram(t, 0, 1000)
def dec():
    w = ctx.compute_decommitts_sorter_circuit_snapshots(np.zeros(0, nv.DECOMMIT_QUERY), 50)
    t = nv.Trace(ctx, n_rows, 1); ctx.synthesize_decommit_sorter(w, t, 0, 1, 0); return w.num_instances, ctx.check_if_satisfied_decommit_sorter(t, 0, 50)
def dmx():
    w = ctx.compute_logs_demux(np.zeros(0, nv.LOG_QUERY), 64)
    t = nv.Trace(ctx, n_rows, 1, n_cols=151); ctx.synthesize_log_demux(w, t, 0, 1, 0); return w.num_instances, ctx.check_if_satisfied_log_demux(t, 0, 64)
def sto():
    w = ctx.compute_storage_dedup_and_sort(np.zeros(0, nv.LOG_QUERY), 40)
    t = nv.Trace(ctx, n_rows, 1); ctx.synthesize_storage_sorter(w, t, 0, 1, 0); return w.num_instances, ctx.check_if_satisfied_storage_sorter(t, 0, 40)
def evt():
    w = ctx.compute_events_dedup_and_sort(np.zeros(0, nv.LOG_QUERY), 16)
    t = nv.Trace(ctx, n_rows, 1); ctx.synthesize_events_sorter(w, t, 0, 1, 0); return w.num_instances, ctx.check_if_satisfied_events_sorter(t, 0, 16)
for n, f in (("ram", ram), ("dec", dec), ("dmx", dmx), ("sto", sto), ("evt", evt)):
    attempt(n, f)
