"""Builds libzkw.so (hand-written HIP kernels for gfx950 + the extern "C" boundary of include/zkw.h).

`python -m era_zkevm_test_harness_amd.build` or `build()`; hipcc cross-compiles without a GPU. The
library is written next to this file (in-tree: it travels to the GPU box with the repo snapshot and is
git-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libzkw.so")
SOURCES = ["zkw_api.hip", "zkw_sorters.hip", "zkw_precompiles.hip", "zkw_setup.hip", "zkw_block.hip", "zkw_recursion.hip", "zkw_comm.hip", "zkw_vm_trace.hip",
           "zkw_dispatch.hip", "zkw_commit.hip", "zkw_batch.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
if os.environ.get("ZKW_PROBE_BUILD"):  # measurement knobs that produce invalid traces (ZKW_NL_PROBE); never in the default library
    FLAGS.append("-DZKW_PROBE_BUILD")


def _deps():
    out = []
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith((".h", ".cuh")):
                out.append(os.path.join(root, f))
    return out


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdr_mtime = max(os.path.getmtime(p) for p in _deps())
    objs, rebuilt = [], False
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_mtime):
            cmd = [HIPCC, *FLAGS, "-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            rebuilt = True
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
