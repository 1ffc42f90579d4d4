#!/usr/bin/env python3
"""Are two builds of a HIP source the same machine code?  Compares, kernel by kernel (demangled name), the instruction
streams of two `hipcc --cuda-device-only -S` files after removing comments and renumbering labels.  Used to check that
deleting dead template branches leaves every surviving kernel unchanged.  usage: isa_diff.py before.s after.s"""
import re
import subprocess
import sys


def kernels(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        lines = []
        for ln in body.split("\n"):
            ln = ln.split(";")[0].strip()
            if not ln or ln.startswith("."):
                if re.match(r"\.LBB\d+_\d+:", ln):
                    lines.append(re.sub(r"\.LBB\d+_", ".LBB_", ln))
                continue
            lines.append(re.sub(r"\.LBB\d+_", ".LBB_", ln))
        out[name] = lines
    return out


a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
dem = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
bad = 0
for n in sorted(set(a) | set(b)):
    if n not in b:
        print("removed  ", dem(n)[:110])
    elif n not in a:
        print("NEW      ", dem(n)[:110]); bad += 1
    elif a[n] != b[n]:
        print("DIFFERENT", dem(n)[:110], len(a[n]), len(b[n])); bad += 1
    else:
        print("identical", dem(n)[:110], len(a[n]))
sys.exit(1 if bad else 0)
