import sys, time, subprocess, numpy as np, torch
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native
ctx = native.Context(0); lib = native.load()
s = torch.cuda.current_stream(); ctx.set_stream(s.cuda_stream); ctx.set_pointer_mode(native.PTR_DEVICE)
def clocks():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True, timeout=20).stdout
        return ' | '.join(l.strip() for l in out.splitlines() if 'sclk' in l or 'mclk' in l)[:300]
    except Exception as e:
        return repr(e)
def run(nc, L, calls, tag):
    enc = torch.randint(0, 2**62, (nc * L, 8), dtype=torch.int64, device='cuda')
    tails = torch.empty((nc * L, 12), dtype=torch.int64, device='cuda')
    offs = np.arange(nc + 1, dtype=np.uint64) * L
    torch.cuda.synchronize()
    ts = []
    for c in range(calls):
        t = time.perf_counter()
        native._check(lib.zkw_queue_push_chain_full_batch(ctx.handle, enc.data_ptr(), offs.ctypes.data, nc, None, tails.data_ptr()))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e6 / L)
    print(tag, nc, 'chains x', L, 'us/step per call:', ' '.join('%.2f' % x for x in ts), flush=True)
print('clocks idle:', clocks())
run(16, 20000, 12, 'short calls')
run(16, 400000, 2, 'long calls ')
print('clocks after:', clocks())
print(subprocess.run(['rocm-smi', '--setperflevel', 'high'], capture_output=True, text=True).stdout[-300:])
print('clocks high:', clocks())
run(16, 20000, 4, 'perf high short')
run(16, 400000, 2, 'perf high long ')
print(subprocess.run(['rocm-smi', '--setperflevel', 'auto'], capture_output=True, text=True).stdout[-200:])
