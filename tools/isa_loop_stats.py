#!/usr/bin/env python3
"""Instruction mix of the big loops of selected kernels in a hipcc -S dump."""
import re, subprocess, sys
from collections import Counter


def report(loop):
    c = Counter()
    for l in loop:
        op = l.split()[0]
        if op.startswith('v_'):
            c['valu'] += 1
            if 'dpp' in l: c['dpp'] += 1
            if op.startswith(('v_div', 'v_rcp')): c['div'] += 1
            if op.startswith('v_cndmask'): c['cndmask'] += 1
            if op.startswith(('v_fma', 'v_fmac')): c['fma'] += 1
            if op.startswith('v_mov'): c['mov'] += 1
            if op.startswith('v_pk_'): c['pk'] += 1
            if op.startswith('v_cmp'): c['cmp'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
            if op.startswith('s_waitcnt'): c['waitcnt'] += 1
            if op.startswith('s_nop'): c['nop'] += 1
        elif op.startswith('global_load'): c['gload'] += 1
        elif op.startswith('global_store'): c['gstore'] += 1
        elif op.startswith('ds_'): c['ds'] += 1
        elif op.startswith('scratch'): c['scratch'] += 1
    print("   loop instrs", len(loop), dict(c))


txt = open(sys.argv[1]).read()
want = sys.argv[2:]
for f in re.split(r'\n(?=_ZN3dxh\w+:)', txt):
    name = f.split(':', 1)[0].strip()
    if not name.startswith('_ZN3dxh'):
        continue
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if not any(w in dn for w in want):
        continue
    body = f.split('.Lfunc_end')[0]
    lines = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith((';', '//'))]
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    spans = []
    for i, l in enumerate(lines):
        m = re.match(r'^s_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > 150:
            spans.append((labels[m.group(1)], i))
    spans = [sp for sp in spans if not any(o != sp and o[0] <= sp[0] and sp[1] <= o[1] for o in spans)]
    print(dn[:78])
    for sp in spans:
        report([l for l in lines[sp[0]:sp[1] + 1] if not l.startswith('.')])
