#!/usr/bin/env python3
"""Register / spill / LDS summary of the kernels in a gfx950 assembly file (hipcc --cuda-device-only -S): one line per
kernel from its .amdhsa metadata.  usage: kernel_regs.py file.s [substring]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if name.endswith(".kd"):
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0]
    if pat not in dem:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    print(f"{dem:70s} vgpr {g('vgpr_count'):>3s} sgpr {g('sgpr_count'):>3s} spill v{g('vgpr_spill_count')} s{g('sgpr_spill_count')} "
          f"scratch {g('private_segment_fixed_size')} lds {g('group_segment_fixed_size')}")
