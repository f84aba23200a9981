#!/usr/bin/env python3
"""Instruction histogram of the innermost loops of one kernel in a hipcc -save-temps .s file.
usage: tools/isa_loop_hist.py file.s mangled_kernel_name"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(re.escape(name) + r':(.*?)\.Lfunc_end', s, re.S)
lines = m.group(1).split('\n')
labels = {}
for i, l in enumerate(lines):
    mm = re.match(r'(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = i
for i, l in enumerate(lines):
    mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        seg = lines[labels[mm.group(1)]:i]
        c = collections.Counter()
        for x in seg:
            x = x.strip()
            if not x or x.startswith(';') or x.startswith('.'):
                continue
            op = x.split()[0]
            if 'dpp' in x and not op.endswith('_dpp'):
                op += '(dpp)'
            c[op] += 1
        tot = sum(c.values())
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        trans = sum(v for k, v in c.items() if re.match(r'v_(log|exp|rcp|sqrt|rsq|sin|cos)', k))
        print("LOOP %s: %d instrs, %d VALU (%d transcendental), %d DS, %d VMEM, %d SALU" % (
            mm.group(1), tot, valu, trans, sum(v for k, v in c.items() if k.startswith('ds_')),
            sum(v for k, v in c.items() if k.startswith('global_') or k.startswith('buffer_')),
            sum(v for k, v in c.items() if k.startswith('s_'))))
        print("   " + ", ".join("%s %d" % kv for kv in c.most_common(40)))
