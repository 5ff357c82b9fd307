#!/usr/bin/env python3
"""Registers, spills, scratch and LDS per kernel, from the remarks the library build leaves in
ilqr_amd/csrc/build_resources.log (development aid).   usage: tools/kernel_resources.py [substring ...]"""
import os
import re
import sys

LOG = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'ilqr_amd', 'csrc', 'build_resources.log')


def main():
    want = sys.argv[1:]
    cur = None
    rows = []
    for line in open(LOG):
        m = re.search(r'remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|'
                      r'VGPRs Spill|LDS Size \[bytes/block\]): (\S+)', line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == 'Function Name':
            cur = {'name': v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(' [')[0]] = v
    for r in rows:
        name = r['name']
        if want and not all(w in name for w in want):
            continue
        print('%-110s v %3s a %3s spill %4s scratch %5s lds %6s occ %s' % (name[:110], r.get('VGPRs'), r.get('AGPRs'), r.get('VGPRs Spill'),
                                                                         r.get('ScratchSize'), r.get('LDS Size'), r.get('Occupancy')))


if __name__ == '__main__':
    main()
