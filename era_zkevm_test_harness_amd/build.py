"""Builds libzkw.so (hand-written HIP kernels for gfx950 + the extern "C" boundary of include/zkw.h).

`python -m era_zkevm_test_harness_amd.build` or `build()`; hipcc cross-compiles without a GPU. The
library is written next to this file (in-tree: it travels to the GPU box with the repo snapshot and is
git-ignored).
"""
import hashlib
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


def _digest(src):
    """What an object file depends on: the unit, every header it can include, the compiler and its flags. Objects are keyed on this CONTENT
    hash (round 6; rounds 1-5 compared modification times, and an object that travelled in a snapshot with a newer mtime than an edited
    header would have gone unnoticed)."""
    h = hashlib.sha256()
    h.update(" ".join([HIPCC, *FLAGS]).encode())
    for path in [os.path.join(CSRC, src)] + sorted(_deps()):
        h.update(os.path.relpath(path, HERE).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    objs, rebuilt = [], False
    procs = []
    known = {src.replace(".hip", ".o") for src in SOURCES} | {src.replace(".hip", ".sha256") for src in SOURCES}
    for stray in os.listdir(OBJ):  # nothing but this list's objects is ever linked or kept (a foreign or stale object tree is not trusted)
        if stray not in known:
            os.remove(os.path.join(OBJ, stray))
    for src in SOURCES:
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        hp = os.path.join(OBJ, src.replace(".hip", ".sha256"))
        objs.append(op)
        want = _digest(src)
        have = open(hp).read().strip() if os.path.exists(hp) else None
        if force or not os.path.exists(op) or have != want:
            if os.path.exists(hp):
                os.remove(hp)
            cmd = [HIPCC, *FLAGS, "-c", os.path.join(CSRC, src), "-o", op]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, hp, want, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            rebuilt = True
    for src, hp, want, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        with open(hp, "w") as f:
            f.write(want)
        if verbose and out.strip():
            print(out)
    # the library is relinked whenever an object was rebuilt, it is missing, or it is older than an object
    if rebuilt or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
