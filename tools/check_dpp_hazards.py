#!/usr/bin/env python3
"""Check the built library's gfx950 code for the one hazard the compiler cannot see: a DPP instruction written in inline asm
(v_fmac_f64_dpp / v_mov_b64_dpp ... row_newbcast) that reads a VGPR a VALU instruction wrote less than two wait states
earlier.  The kernels put an s_nop 1 in front of every such instruction or group of them (kernels_backward.h); this tool
proves it on the instructions that were actually emitted, whatever the register allocator did around the asm statements.

usage: tools/check_dpp_hazards.py [path/to/libilqr_hip.so]      exit status 1 and a listing if a violation is found."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    """set of VGPR numbers named by an operand token (v7, v[4:5], -v[2:3], |v[0:1]|), empty for anything else"""
    m = re.search(r'\bv\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r'\bv(\d+)\b', tok)
    return {int(m.group(1))} if m else set()


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix='ilqr_co_')
    try:
        copy = os.path.join(tmp, 'lib.so')
        shutil.copy(lib, copy)
        subprocess.run([OBJDUMP, '--offloading', copy], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        co = [f for f in os.listdir(tmp) if 'amdgcn' in f]
        if not co:
            raise RuntimeError('no device code object found in ' + lib)
        return subprocess.run([OBJDUMP, '-d', os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def check(text):
    """(number of DPALU row-broadcast instructions checked, list of violations)"""
    bad, checked = [], 0
    func, window = '?', []                  # window: the last instructions as (wait states it provides, VGPRs a VALU wrote)
    for line in text.split('\n'):
        m = re.match(r'^[0-9a-f]+ <(.+)>:', line)
        if m:
            func, window = m.group(1), []
            continue
        t = line.split('//')[0].strip()
        if not t or t.endswith(':'):
            continue
        op, _, rest = t.partition(' ')
        ops = [o.strip() for o in rest.split(',')] if rest else []
        if op.endswith('_dpp') and 'row_newbcast' in t and ('_f64' in op or '_b64' in op):
            checked += 1
            src = regs(ops[1]) if len(ops) > 1 else set()
            ws = 0
            for states, written in reversed(window):
                if ws >= 2:
                    break
                if written & src:
                    bad.append((func, t))
                    break
                ws += states
        states = 1
        written = set()
        if op == 's_nop':
            states = int(rest.strip() or 0) + 1
        elif op.startswith('v_') and not op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')) and ops:
            written = regs(ops[0])
        window.append((states, written))
        if len(window) > 8:
            window.pop(0)
    return checked, bad


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'ilqr_amd', 'libilqr_hip.so')
    checked, bad = check(disassemble(lib))
    print('%d DPALU row-broadcast instructions checked, %d hazards' % (checked, len(bad)))
    for func, t in bad[:20]:
        print('  %s: %s' % (func[:80], t))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
