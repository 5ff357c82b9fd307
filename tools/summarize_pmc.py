#!/usr/bin/env python3
"""Condense the rocprofv3 PMC passes collected by tools/collect_profiles.sh into one JSON (per kernel, per launch; the
line search also per round of its search and per width).

usage: python tools/summarize_pmc.py <dir with pmc_*/ and pmc_*.log> [traffic.json]

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a coalesced streaming read, so it is doubled.  The factor was checked on this code's own
access patterns (tools/probe/fetch_calib.hip, profiles/r1_fetch_calibration/)."""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash       # hash of the kernel sources the counters were taken on (bench.py drops stale traffic)

KERNELS = ('k_linesearch', 'k_ls_compact', 'k_accept_riccati', 'k_riccati_grp', 'k_riccati_quad', 'k_riccati', 'k_accept_lin', 'k_save_boundaries',
           'k_linearize', 'k_rollout', 'k_cost', 'k_backward', 'k_forward', 'k_ltvlqr', 'k_mpc_shift')


def short(name):
    for k in KERNELS:
        if k + '<' in name or k + '(' in name or name.startswith(k) or ('_' + k + 'I') in name or name.endswith(k):
            return k
    return None


def ls_width(name):
    m = re.search(r'k_linesearch<[^,]+,\s*(\d+)', name) or re.search(r'k_linesearchINS_\d+Sys_\w+?ELi(\d+)E', name)
    return int(m.group(1)) if m else None


def main():
    d = sys.argv[1]
    out = collections.defaultdict(dict)
    files = []
    for sub in sorted(glob.glob(os.path.join(d, 'pmc_*'))):      # one pass per directory: its NEWEST collection only
        cand = glob.glob(os.path.join(sub, '*', '*_counter_collection.csv'))
        if cand:
            files.append(max(cand, key=os.path.getmtime))
    for f in files:
        rows = list(csv.DictReader(open(f)))
        # rounds of one search: line-search launches in dispatch order since the last backward / accept kernel
        disp = {}
        for r in rows:
            disp.setdefault(int(r['Dispatch_Id']), r['Kernel_Name'])
        round_of, rnd = {}, 0
        for did in sorted(disp):
            k = short(disp[did])
            if k == 'k_linesearch':
                round_of[did] = rnd
                rnd += 1
            elif k != 'k_ls_compact':
                rnd = 0
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            k = short(r['Kernel_Name'])
            if not k:
                continue
            keys = [k]
            if k == 'k_linesearch':
                did = int(r['Dispatch_Id'])
                keys += ['k_linesearch.round%d' % round_of[did], 'k_linesearch.W%s' % ls_width(r['Kernel_Name'])]
            for key in keys:
                agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
                agg[key]['_dur_ns.' + r['Counter_Name']].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
        for k, cs in agg.items():
            for c, v in cs.items():
                if c.startswith('_dur_ns.'):
                    out[k]['dur_us_under_' + c[8:]] = sum(v) / len(v) / 1e3
                else:
                    out[k][c] = sum(v) / len(v)
                    out[k]['launches'] = len(v)
    for k, v in out.items():
        if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            v['hbm_read_bytes_per_launch'] = v['FETCH_SIZE'] * 1024 * 2
            v['hbm_write_bytes_per_launch'] = v['WRITE_SIZE'] * 1024
            v['hbm_bytes_per_launch'] = v['hbm_read_bytes_per_launch'] + v['hbm_write_bytes_per_launch']
        if 'SQ_ACTIVE_INST_VALU' in v and 'GRBM_GUI_ACTIVE' in v:
            # SQ_ACTIVE_INST_* count quad-cycles per wave; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
            v['valu_busy_frac_of_simd_cycles'] = v['SQ_ACTIVE_INST_VALU'] * 4 / (v['GRBM_GUI_ACTIVE'] / 8 * 1024)
        if 'SQ_INSTS_VALU' in v and 'SQ_WAVES' in v and v['SQ_WAVES'] > 0:
            v['valu_insts_per_wave'] = v['SQ_INSTS_VALU'] / v['SQ_WAVES']
        if 'SQ_WAVE_CYCLES' in v and 'SQ_BUSY_CYCLES' in v and v['SQ_BUSY_CYCLES'] > 0:
            # wave-cycles (quad-cycle units, summed over waves) / busy cycles of the SQs: mean resident waves, all SIMDs
            v['mean_resident_waves_x_quad'] = v['SQ_WAVE_CYCLES'] / v['SQ_BUSY_CYCLES']
    print(json.dumps(out, indent=1, sort_keys=True))
    if len(sys.argv) > 2:                       # second argument: write the per-kernel HBM traffic file bench.py reads
        cfg = None
        for f in glob.glob(os.path.join(d, 'pmc_*.log')):
            for line in open(f, errors='replace'):
                if line.startswith('CONFIG '):
                    cfg = json.loads(line[7:])
        keep = ('hbm_bytes_per_launch', 'hbm_read_bytes_per_launch', 'hbm_write_bytes_per_launch', 'launches',
                'valu_busy_frac_of_simd_cycles', 'valu_insts_per_wave', 'SQ_WAVES', 'SQ_INSTS_VALU')
        tr = {'source': 'rocprofv3 --pmc passes of tools/pmc_probe.py (tools/collect_profiles.sh): FETCH_SIZE / WRITE_SIZE (KiB; '
                        'FETCH_SIZE x2 on gfx950), SQ_* and GRBM_GUI_ACTIVE in separate runs; averages per kernel launch; '
                        'a line-search pass is one launch per round',
              'config': cfg, 'source_hash': source_hash(),
              'kernels': {k: {kk: v[kk] for kk in keep if kk in v} for k, v in out.items() if 'hbm_bytes_per_launch' in v}}
        json.dump(tr, open(sys.argv[2], 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
