"""One-off source transformation of round 6 (kept for the record; running it twice is a no-op): builder kernels become BODIES
(`static __device__ void k_x(const VB& vb, ...)`, csrc/zkw_launch.h) and their launch sites ZKW_LAUNCH calls, so that the same body runs
as a launch of its own (zkw_block_run) or as a job of a merged launch (zkw_blocks_run, csrc/zkw_batch.h)."""
import os
import re
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "era_zkevm_test_harness_amd", "csrc")
KERNEL_FILES = ["ram_kernels.cuh", "log_kernels.cuh", "scan_kernels.cuh", "decommit_kernels.cuh", "events_kernels.cuh", "demux_kernels.cuh",
                "storage_kernels.cuh", "decommitter_kernels.cuh", "precompile_kernels.cuh", "storage_application_kernels.cuh",
                "closed_form_kernels.cuh", "public_input_kernels.cuh", "vm_kernels.cuh", "zkw_api.hip"]
LAUNCH_FILES = ["zkw_api.hip", "zkw_sorters.hip", "zkw_precompiles.hip", "scan_kernels.cuh", "closed_forms_host.h"]
KEEP_GLOBAL = re.compile(r"k_chain_|check")
# second pass of round 6: the synthesis kernels too (zkw_blocks_synthesize runs groups of blocks as fibers: one fill launch per type and group)
SYNTH_KERNEL_FILES = ["ram_circuit_kernels.cuh", "decommit_sorter_circuit_kernels.cuh", "events_sorter_circuit_kernels.cuh", "log_demux_circuit_kernels.cuh",
                      "storage_sorter_circuit_kernels.cuh", "netlist_kernels.cuh", "netlist_queue_kernels.cuh", "netlist_closed_form_kernels.cuh",
                      "ecrecover_kernels.cuh", "storage_application_kernels.cuh", "zkw_ctx.h"]

DEF = re.compile(r"(?:static )?__global__ (?:__launch_bounds__\(([^)]*)\) )?void (k_\w+)\(")


def convert_defs(path, bounds):
    s = open(path).read()
    out, pos = [], 0
    for m in DEF.finditer(s):
        name = m.group(2)
        if KEEP_GLOBAL.search(name):
            continue
        bounds[name] = m.group(1)
        out.append(s[pos:m.start()])
        # body extent: from the first '{' after the parameter list to its matching '}'
        depth, i = 0, m.end() - 1
        while True:
            c = s[i]
            if c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
                if depth == 0:
                    break
            i += 1
        params_end = i
        j = s.index("{", params_end)
        depth, k = 0, j
        while True:
            c = s[k]
            if c == "{":
                depth += 1
            elif c == "}":
                depth -= 1
                if depth == 0:
                    break
            k += 1
        body = s[j:k + 1]
        body = body.replace("blockIdx.x", "vb.x").replace("gridDim.x", "vb.nx").replace("blockIdx.y", "vb.y").replace("gridDim.y", "vb.ny").replace("blockIdx.z", "vb.z").replace("gridDim.z", "vb.nz")
        params = s[m.end():params_end]
        head = "static __device__ __forceinline__ void %s(const VB& vb%s" % (name, ", " + params if params.strip() else "")
        out.append(head + s[params_end:j] + body)
        pos = k + 1
    out.append(s[pos:])
    open(path, "w").write("".join(out))


def split_top(s):
    """split 'a, b(c, d), e<f, g>(h)' at top-level commas (parentheses / brackets / braces only; '<' is ambiguous and kernels with
    template commas are wrapped in parentheses at the launch sites)"""
    depth, cur, out = 0, [], []
    for c in s:
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
        if c == "," and depth == 0:
            out.append("".join(cur).strip())
            cur = []
        else:
            cur.append(c)
    out.append("".join(cur).strip())
    return out


def convert_launches(path, bounds, report):
    s = open(path).read()
    out, pos = [], 0
    for m in re.finditer(r"hipLaunchKernelGGL\(", s):
        if m.start() < pos:
            continue
        depth, i = 0, m.end() - 1
        while True:
            c = s[i]
            if c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
                if depth == 0:
                    break
            i += 1
        inner = s[m.end():i]
        a = split_top(inner)
        kern = a[0]
        base = re.sub(r"^HIP_KERNEL_NAME\((.*)\)$", r"\1", kern).strip()
        base = base[1:-1].strip() if base.startswith("(") and base.endswith(")") else base
        name = re.match(r"(\w+)", base).group(1)
        if name not in bounds:
            continue
        grid, block, shm, stream = a[1], a[2], a[3], a[4]
        mctx = re.match(r"^(.*)->stream$", stream)
        g = re.match(r"^dim3\((.*)\)$", grid)
        b = re.match(r"^dim3\((.*)\)$", block)
        if mctx and b and len(split_top(b.group(1))) == 1 and not (g and shm == "0" and len(split_top(g.group(1))) <= 2):
            # the general form: any dim3 grid expression, dynamic LDS
            out.append(s[pos:m.start()])
            out.append("ZKW_LAUNCH_D(%s, (%s), \"%s\", %s, %s, %s, %s)" % (mctx.group(1), base, name, grid, b.group(1), shm, ", ".join(a[5:])))
            pos = i + 1
            continue
        if not (mctx and g and b and shm == "0"):
            report.append("%s: manual: %s" % (os.path.basename(path), inner[:120].replace("\n", " ")))
            continue
        gd, bd = split_top(g.group(1)), split_top(b.group(1))
        if len(bd) != 1 or len(gd) > 2:
            report.append("%s: manual dims: %s" % (os.path.basename(path), inner[:120].replace("\n", " ")))
            continue
        args = ", ".join(a[5:])
        templ = "<" in base
        if templ:
            call = "ZKW_LAUNCH_T(%s, (%s), \"%s\", %s, %s, %s)" % (mctx.group(1), base, name, gd[0], bd[0], args)
            if len(gd) == 2:
                report.append("%s: manual (2-D template): %s" % (os.path.basename(path), inner[:100]))
                continue
        elif len(gd) == 2:
            call = "ZKW_LAUNCH_2D(%s, %s, %s, %s, %s, %s)" % (mctx.group(1), base, gd[0], gd[1], bd[0], args)
        else:
            call = "ZKW_LAUNCH(%s, %s, %s, %s, %s)" % (mctx.group(1), base, gd[0], bd[0], args)
        out.append(s[pos:m.start()])
        out.append(call)
        pos = i + 1
    out.append(s[pos:])
    open(path, "w").write("".join(out))


def main():
    bounds, report = {}, []
    for f in KERNEL_FILES + SYNTH_KERNEL_FILES:
        convert_defs(os.path.join(CSRC, f), bounds)
    # kernels converted in an earlier run of this script (definitions no longer match DEF)
    for f in KERNEL_FILES + SYNTH_KERNEL_FILES:
        for m in re.finditer(r"static __device__ (?:__forceinline__ )?void (k_\w+)\(const VB& vb", open(os.path.join(CSRC, f)).read()):
            bounds.setdefault(m.group(1), None)
    for f in LAUNCH_FILES:
        convert_launches(os.path.join(CSRC, f), bounds, report)
    print("%d kernel bodies" % len(bounds))
    print("\n".join(report))


if __name__ == "__main__":
    sys.exit(main())
