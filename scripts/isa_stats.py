#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing: scripts/isa_stats.py file.s kernel-substring"""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
on, body = False, []
for l in open(src):
    if not on and re.match(r"^_Z\S*:", l) and pat in l:
        on = True
        continue
    if on:
        if l.startswith(".Lfunc_end"):
            break
        body.append(l)
ins = [l.split()[0] for l in body if re.match(r"^\s+[a-z]", l)]
def cnt(rx): return sum(1 for i in ins if re.match(rx, i))
print(f"{pat}: total {len(ins)}  valu {cnt(r'v_')}  fp32 {cnt(r'v_(pk_)?(add|sub|mul|fma|fmac|mac)_f32')}  pk {cnt(r'v_pk_')}  "
      f"mov {cnt(r'v_mov|v_accvgpr')}  ds {cnt(r'ds_')}  vmem {cnt(r'global_|buffer_|flat_')}  scratch {cnt(r'scratch_')}  "
      f"salu {cnt(r's_')}  waitcnt {cnt(r's_waitcnt')}")
