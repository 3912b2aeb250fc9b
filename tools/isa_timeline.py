#!/usr/bin/env python
"""Timeline of the main loop of a gfx950 kernel from the compiler's assembly: runs of VALU instructions, global loads / stores, LDS
accesses, every s_waitcnt with its counters, barriers and branches, in program order -- the listing the layout of the marching
kernels for two resident waves was worked out on.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only adflow_amd/csrc/kernels_roe_march.hip -o /tmp/rm.s
  python tools/isa_timeline.py /tmp/rm.s k_roe_marchILi3ELb0ELb1ELb1ELb0E
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_count import kernel_lines, loops
f,pat=sys.argv[1],sys.argv[2]
txt=open(f).read().split('\n')
s,e=kernel_lines(txt,pat)
ls=loops(txt,s,e)
a,b,c=max(ls,key=lambda t: sum(t[2].values()))
valu=0; out=[]
def flush():
    global valu
    if valu: out.append(f"  valu x{valu}"); valu=0
run=None; cnt=0
for l in txt[a:b+1]:
    m=re.match(r"^\s+([a-z_0-9]+)\s*(.*)",l)
    if not m:
        if re.match(r"^\.LBB",l): flush(); out.append(l.strip())
        continue
    op,args=m.group(1),m.group(2)
    if op.startswith('v_'): valu+=1; continue
    if op.startswith(('global_load','global_store','ds_','scratch_','s_waitcnt','s_barrier','s_cbranch','s_branch')):
        flush()
        key=op if not op.startswith('s_waitcnt') else op+' '+args.split(';')[0].strip()
        if op.startswith(('s_cbranch','s_branch')): key=op+' '+args.split()[0]
        if out and out[-1].startswith(key+' x') :
            n=int(out[-1].split(' x')[-1]); out[-1]=f"{key} x{n+1}"
        else: out.append(f"{key} x1")
print('\n'.join(out))
