#!/usr/bin/env python3
"""chevron.py -- TEST INFRASTRUCTURE ONLY.  stdin: a CUDA source; stdout: the same text with
every launch   kernel[<targs>] <<<grid, block, shmem, stream>>> (   rewritten to
cuda_cpu::launch(grid, block, shmem, stream, [&](auto... a) { kernel[<targs>](a...); }) (
so that g++ can compile it against oracle/cuda_on_cpu/.  Nothing else is touched; the
reference source is streamed from /root/reference into the compiler, never stored."""
import re
import sys

src = sys.stdin.read()
pat = re.compile(r"([A-Za-z_]\w*(?:\s*<[^<>;(){}]*>)?)\s*<<<(.*?)>>>\s*\(", re.S)
out, n = pat.subn(lambda m: "cuda_cpu::launch(%s, [&](auto... a_) { %s(a_...); })(" % (m.group(2), m.group(1)), src)
assert "<<<" not in out, "unhandled launch syntax"
sys.stdout.write(out)
sys.stderr.write("chevron.py: %d launches rewritten\n" % n)
