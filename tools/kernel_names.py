"""rocprofv3 prints the two launch forms of a kernel body (csrc/zkw_launch.h) by their mangled names,
`_ZN3zkw8k_singleITnDaXadL_ZNS_L19k_ram_fill_poseidonILi1EE...`: short(name) gives `zkw::k_ram_fill_poseidon<1>` (the name the
__global__ kernel of rounds 1-5 had; ` [merged]` appended for the k_multi form), and leaves every other name as it is."""
import re


def short(name):
    n = name.split("(")[0].replace("void ", "").strip()
    m = re.search(r"k_(single|multi)ITnDaXadL_Z", n)
    if not m:
        return n
    base = tail = None
    for d in re.finditer(r"(\d+)(k_\w+)", n[m.end():]):  # <length><name>: the digits may run into a namespace's closing `_1`
        digits, rest = d.group(1), d.group(2)
        for cut in range(len(digits)):
            length = int(digits[cut:])
            if 2 < length <= len(rest) and (len(rest) == length or rest[length] in "EIR") and re.fullmatch(r"k_[a-z0-9_]+", rest[:length]):
                base, tail = rest[:length], rest[length:]
                break
        if base:
            break
    if not base:
        return n
    t = re.match(r"I((?:Li\d+E)+)E", tail)
    targs = "<" + ", ".join(re.findall(r"Li(\d+)E", t.group(1))) + ">" if t else ""
    return "zkw::" + base + targs + (" [merged]" if m.group(1) == "multi" else "")


if __name__ == "__main__":
    import sys
    for line in sys.stdin:
        print(short(line.strip()))
