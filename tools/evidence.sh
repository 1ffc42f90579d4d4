#!/bin/bash
# The end-of-round evidence visit (final code): full GPU suite, smoke, the default bench line, fp16 + INT8 frame kernel
# traces, rocprofv3 kernel stats + FETCH / WRITE PMC passes of the hot-path command, the in-frame SCA call's own passes.
#   gpurun --timeout 3000 -- 'bash tools/evidence.sh r6ev'
exec bash "$(dirname "$0")/visit.sh" "${1:-ev}" tests smoke bench trace trace:int8 prof pmc sca "py:stem_time.py"
