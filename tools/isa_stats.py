#!/usr/bin/env python3
"""Per-kernel instruction mix of the gfx950 ISA hipcc emits for ilqr_hip.hip (development aid).
usage: python tools/isa_stats.py [kernel-name-substring ...]"""
import os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def main():
    pats = sys.argv[1:] or ['k_linesearchINS_11Sys_acrobotELi32', 'k_backwardINS_11Sys_acrobot', 'k_forwardINS_11Sys_acrobotELb1']
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
                               '-save-temps', '-o', os.path.join(td, 'l.so'), os.path.join(ROOT, 'ilqr_amd/csrc/ilqr_hip.hip')],
                              cwd=td, stderr=subprocess.DEVNULL)
        lines = open(os.path.join(td, 'ilqr_hip-hip-amdgcn-amd-amdhsa-gfx950.s')).read().split('\n')
    starts = {i: l.split(':')[0] for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)}
    for start, name in starts.items():
        if not any(p in name for p in pats):
            continue
        blocks, cur = [], None
        for l in lines[start + 1:]:
            if l.startswith('.Lfunc_end'):
                break
            t = l.strip()
            if re.match(r'^\.LBB\d+_\d+:', t):
                cur = [t.split(':')[0], []]; blocks.append(cur)
            elif cur is not None and t and not t.startswith((';', '.')):
                cur[1].append(t)
        tot = sum(len(b[1]) for b in blocks)
        print('%s: %d blocks, %d instructions' % (name[:90], len(blocks), tot))
        for b, ins in blocks:
            if len(ins) >= 30:
                c = Counter(i.split()[0] for i in ins)
                f64 = sum(v for k, v in c.items() if k.endswith('_f64') or k.endswith('_f64_e32') or k.endswith('_f64_e64'))
                vmem = sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'scratch_', 'flat_')))
                valu = sum(v for k, v in c.items() if k.startswith('v_'))
                print('   %-10s %4d instr: %3d valu (%3d f64, %2d v_mov) %2d vmem %2d salu  div=%d' % (
                    b, len(ins), valu, f64, sum(v for k, v in c.items() if k.startswith('v_mov')), vmem,
                    sum(v for k, v in c.items() if k.startswith('s_')), c.get('v_div_fixup_f64', 0)))
if __name__ == '__main__':
    main()
