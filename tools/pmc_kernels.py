#!/usr/bin/env python3
"""Average PMC counters per kernel from a rocprofv3 counter_collection.csv (development aid).
usage: python tools/pmc_kernels.py <dir> [kernel substring]"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[-1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    m = re.search(r'k_\w+?INS_\d+Sys_\w+?E', k)
    name = (m.group(0) if m else k)[:48]
    if pat not in name:
        continue
    acc[name][r['Counter_Name']] += float(r['Counter_Value']); cnt[name][r['Counter_Name']] += 1
for n, c in acc.items():
    print(n, {k: '%.4g' % (v / cnt[n][k]) for k, v in sorted(c.items())}, 'launches', max(cnt[n].values()))
