#!/usr/bin/env python3
"""Print the kernel timeline of one solve() from a rocprofv3 --kernel-trace csv (development aid).
usage: python tools/trace_timeline.py <dir with *_kernel_trace.csv> [first_row [rows]]"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 80
t0 = int(rows[first]['Start_Timestamp'])
prev_end = t0
for r in rows[first:first + cnt]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    m = re.search(r'(k_\w+?)(?:INS|E|\b)', r['Kernel_Name'])
    name = m.group(1) if m else r['Kernel_Name'][:30]
    print('%9.3f ms  +%7.1f us  dur %8.1f us  q%-3s grid %-8s %s' % ((s - t0) / 1e6, (s - prev_end) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'),
                                                                 r.get('Grid_Size_X', r.get('Grid_Size', '?')), name))
    prev_end = max(prev_end, e)
