#!/bin/bash
# usage: tools_resusage.sh file.hip  -> one line per kernel: name vgpr sgpr scratch occupancy lds
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -std=c++17 --offload-arch=gfx950 -c "$1" -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for line in sys.stdin:
    m=re.search(r'remark: (.*?)(?: \[-Rpass)',line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'):
        name=t.split(':',1)[1].strip()
        try: name=subprocess.check_output(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt',name]).decode().strip().split('(')[0]
        except Exception: pass
        cur={'name':name}; rows.append(cur)
    elif cur is not None and ':' in t:
        k,v=t.split(':',1); cur[k.strip()]=v.strip()
for r in rows:
    print(r['name'][-70:].ljust(70), 'VGPR',r.get('VGPRs'),'AGPR',r.get('AGPRs'),'SGPR',r.get('SGPRs') or r.get('TotalSGPRs'),'scratch',r.get('ScratchSize [bytes/lane]'),'occ',r.get('Occupancy [waves/SIMD]'),'LDS',r.get('LDS Size [bytes/block]'))
"
