for nc in 4096 8192 16384; do
  echo "== $nc $(date +%T)"
  timeout -s KILL 60 python - <<PY
import sys, numpy as np, torch, time
sys.path.insert(0,'.')
from era_zkevm_test_harness_amd import native
ctx=native.Context(0); lib=native.load()
s=torch.cuda.current_stream(); ctx.set_stream(s.cuda_stream); ctx.set_pointer_mode(native.PTR_DEVICE)
nc=$nc; L=500
enc=torch.randint(0,2**62,(nc*L,8),dtype=torch.int64,device='cuda'); tails=torch.empty((nc*L,12),dtype=torch.int64,device='cuda')
offs=np.arange(nc+1,dtype=np.uint64)*L
for rep in range(2):
    torch.cuda.synchronize(); t=time.time()
    native._check(lib.zkw_queue_push_chain_full_batch(ctx.handle, enc.data_ptr(), offs.ctypes.data, nc, None, tails.data_ptr()))
    torch.cuda.synchronize(); dt=time.time()-t
    print(nc, L, "%.2f ms  %.2f us/step  %.1f Mperm/s"%(dt*1e3, dt*1e6/L, nc*L/dt/1e6), flush=True)
PY
  echo "rc=$?"
done
