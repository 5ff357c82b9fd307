#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc --save-temps listing (development aid).
usage: tools/isa_loop_stats.py file.s <mangled-name substring> [...more substrings]
Prints, per backward branch target (a loop), the number of instructions between the label and the branch, by class."""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith('v_') and ('f64' in op or op.startswith('v_fmac_f64')):
        return 'valu_f64'
    if op.startswith('v_'):
        return 'valu_other'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'):
        return 'wait/nop'
    if op.startswith('s_load') or op.startswith('s_buffer'):
        return 'smem'
    if op.startswith('s_cbranch') or op.startswith('s_branch'):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'):
        return 'vmem'
    if op.startswith('ds_'):
        return 'lds'
    return 'other'


def main():
    path, subs = sys.argv[1], sys.argv[2:]
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        head = l.split(';')[0].strip()
        if head.endswith(':') and not head.startswith('.') and all(s in head for s in subs):
            start = i
            break
    if start is None:
        sys.exit('kernel not found')
    print(lines[start])
    body = []
    for l in lines[start + 1:]:
        if l.startswith('.Lfunc_end'):
            break
        body.append(l)
    labels = {}
    ins = []
    for l in body:
        t = l.strip()
        if not t or t.startswith(';') or t.startswith('.') and not t.endswith(':'):
            continue
        if t.endswith(':'):
            labels[t[:-1]] = len(ins)
            continue
        ins.append(t.split(';')[0].strip())
    print('instructions in kernel:', len(ins))
    for k, t in enumerate(ins):
        m = re.match(r's_c?branch\S*\s+(\S+)', t)
        if m and m.group(1) in labels and labels[m.group(1)] <= k:
            lo = labels[m.group(1)]
            c = Counter(classify(x.split()[0]) for x in ins[lo:k + 1])
            print('loop %-12s %5d instr: %s' % (m.group(1), k + 1 - lo, dict(sorted(c.items()))))


if __name__ == '__main__':
    main()
