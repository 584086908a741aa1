"""Developer probe (GPU box): wall/latency numbers for the RAM-permutation kernels through the C ABI.
Usage: python tools/gpu_probe.py [--blocks B] [--n N] [--reps R]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from era_zkevm_test_harness_amd import native, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=8)
    ap.add_argument("--n", type=int, default=136714)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--chain-len", type=int, default=20000)
    ap.add_argument("--only", default="all")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ctx = native.Context(0)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_pointer_mode(native.PTR_DEVICE)
    lib = native.load()

    def timed(fn, reps=args.reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    # --- chains: latency per permutation vs number of concurrent chains
    L = args.chain_len
    for n_chains in ((1, 4, 64, 512, 4096, 16384, 65536) if args.only in ('all', 'chain') else ()):
        if n_chains * L > 40_000_000:
            L2 = 40_000_000 // n_chains
        else:
            L2 = L
        enc = torch.randint(0, 2**62, (n_chains * L2, 8), dtype=torch.int64, device=dev)
        tails = torch.empty((n_chains * L2, 12), dtype=torch.int64, device=dev)
        offs = np.arange(n_chains + 1, dtype=np.uint64) * L2
        ms = timed(lambda: native._check(lib.zkw_queue_push_chain_full_batch(
            ctx.handle, enc.data_ptr(), offs.ctypes.data, n_chains, None, tails.data_ptr())))
        print(f"chain: {n_chains:5d} chains x {L2:6d} items: {ms:9.2f} ms  -> {ms * 1e3 / L2:7.2f} us/perm-step, "
              f"{n_chains * L2 / ms / 1e3:9.2f} Mperm/s", flush=True)
        del enc, tails

    # --- grand product, production size
    for B in ((1, 16) if args.only in ('all', 'gp') else ()):
        n = args.n
        lhs = torch.randint(0, 2**62, (2 * B * n, 8), dtype=torch.int64, device=dev)
        ch = torch.randint(0, 2**62, (2, 9), dtype=torch.int64, device=dev)
        lz = torch.empty((2, B * n), dtype=torch.int64, device=dev)
        rz = torch.empty((2, B * n), dtype=torch.int64, device=dev)
        ms = timed(lambda: native._check(lib.zkw_grand_product_chains(
            ctx.handle, lhs.data_ptr(), lhs.data_ptr() + B * n * 64, B * n, 8, ch.data_ptr(), 2, lz.data_ptr(),
            rz.data_ptr())), reps=10)
        alg = 2 * B * n * (64 + 16)
        print(f"grand product: n={B * n} W=8 reps=2: {ms:8.3f} ms, algorithmic {alg / 1e6:.1f} MB -> {alg / ms / 1e6:.1f} GB/s",
              flush=True)
        del lhs, lz, rz

    # --- RAM builder batch
    if args.only not in ('all', 'ram'):
        return
    B, n = args.blocks, args.n
    base = synthetic.ram_trace(n, seed=2)
    q = torch.from_numpy(np.tile(base.view(np.uint8).reshape(n, 48), (B, 1))).to(dev)
    offs = np.arange(B + 1, dtype=np.uint64) * n
    w = native.RamWitness(ctx)
    t0 = time.time()
    ms = timed(lambda: ctx.compute_ram_circuit_snapshots((q.data_ptr(), B * n), n, 0, block_offsets=offs, witness=w), reps=2)
    print(f"ram builder: {B} blocks x {n} queries: {ms:9.2f} ms per batch ({B / ms * 1e3:.2f} instances/s); host wall {time.time() - t0:.2f}s",
          flush=True)


if __name__ == "__main__":
    main()
