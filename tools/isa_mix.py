"""Static instruction mix of kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only).  usage: isa_mix.py file.s substr..."""
import re
import sys
from collections import Counter


def functions(txt):
    for m in re.finditer(r"\n(_Z\S+):\s*;\s*@", txt):
        name = m.group(1)
        body = txt[m.end():txt.index(".Lfunc_end", m.end())]
        yield name, [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".", "_")) and not l.strip().endswith(":")]


def mix(lines):
    c = Counter()
    for l in lines:
        op = l.split()[0]
        if op.startswith("v_"):
            c["valu"] += 1
            if "f64" in op:
                c["f64"] += 1
            if "dpp" in l:
                c["dpp"] += 1
            if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
                c["lane"] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif op.startswith("s_barrier"):
            c["barrier"] += 1
        elif op.startswith(("s_load", "s_buffer")):
            c["smem"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("global_load"):
            c["gload"] += 1
        elif op.startswith("global_store"):
            c["gstore"] += 1
        elif op.startswith("global_atomic"):
            c["gatomic"] += 1
        elif op.startswith("ds_"):
            c["ds"] += 1
        elif op.startswith("scratch"):
            c["scratch"] += 1
        else:
            c["other"] += 1
    return dict(c)


if __name__ == "__main__":
    txt = open(sys.argv[1]).read()
    for name, lines in functions(txt):
        if any(s in name for s in sys.argv[2:]):
            print(name[:90], len(lines), mix(lines))
