#!/usr/bin/env python3
"""Condense the rocprofv3 PMC passes collected by tools/collect_profiles.sh into one JSON (per kernel, per launch).

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a coalesced streaming read, so it is doubled.  The factor was checked on this code's own
access pattern (8 B per lane, 512-byte rows): k_cost reads B*N*m*8 bytes of U and FETCH_SIZE reports half of it;
k_rollout's WRITE_SIZE matches the bytes of X and U it writes exactly.  For k_linesearch (W lanes share one 8-byte
address) the factor is an upper bound: narrow requests may be tallied at their true size."""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash       # hash of the kernel sources the counters were taken on (bench.py drops stale traffic)

def short(name):
    for k in ('k_linesearch', 'k_riccati_grp', 'k_riccati_quad', 'k_riccati', 'k_accept_par', 'k_accept_lin', 'k_save_boundaries', 'k_linearize', 'k_rollout', 'k_cost', 'k_backward', 'k_forward', 'k_ltvlqr'):
        if k + '<' in name or k + '(' in name:
            return k
    return None

def main():
    d = sys.argv[1]
    out = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(d, 'pmc_*', '*', '*_counter_collection.csv')):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if k:
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
                agg[k]['_dur_ns'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
        for k, cs in agg.items():
            for c, v in cs.items():
                if c != '_dur_ns':
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
            v['valu_insts_per_launch'] = v.get('SQ_INSTS_VALU')
    print(json.dumps(out, indent=1, sort_keys=True))
    if len(sys.argv) > 2:                       # second argument: write the per-kernel HBM traffic file bench.py reads
        cfg = None
        for f in glob.glob(os.path.join(d, 'pmc_*.log')):
            for line in open(f, errors='replace'):
                if line.startswith('CONFIG '):
                    cfg = json.loads(line[7:])
        tr = {'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_probe.py (tools/collect_profiles.sh); '
                        'bytes per kernel launch, FETCH_SIZE x2 (gfx950), a line-search pass is one launch per round',
              'config': cfg, 'source_hash': source_hash(),
              'kernels': {k: {'hbm_bytes_per_launch': v['hbm_bytes_per_launch'], 'read': v['hbm_read_bytes_per_launch'],
                              'write': v['hbm_write_bytes_per_launch'], 'launches': v['launches']}
                          for k, v in out.items() if 'hbm_bytes_per_launch' in v}}
        json.dump(tr, open(sys.argv[2], 'w'), indent=1, sort_keys=True)

if __name__ == '__main__':
    main()
