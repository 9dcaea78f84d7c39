#!/usr/bin/env python3
"""Summarise the three rocprofv3 --pmc passes over tools/gemm_pmc.py (FETCH_SIZE | WRITE_SIZE | MFMA/LDS counters) into
profiles/<round>/gemm_pmc.{json,txt}.  Usage: pmc_summary.py <fetch_dir> <write_dir> <sq_dir> <out_prefix>

HBM-side bytes follow MI355X_MICROARCH.md's gfx950 recipe: read = 2 x FETCH_SIZE x 1024 (FETCH_SIZE is in KiB and counts
64 B per 128-B request of a 16-B/lane read), write = WRITE_SIZE x 1024."""
import collections, csv, glob, json, sys

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_pmc import SHAPES as _SH, LAUNCHES      # the launches tools/gemm_pmc.py makes, in order

SHAPES = [s[0] for s in _SH]
FLOPS = [2.0 * s[3] * s[4] * s[5] for s in _SH]
CLASSES = [s[1] for s in _SH]      # bench.py's kernel classes (ss_profile)


def load(d):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if 'gemm_' not in r['Kernel_Name'] or 'ss::' not in r['Kernel_Name'] or 'splitk' in r['Kernel_Name']:
            continue
        e = disp.setdefault(int(r['Dispatch_Id']), {'kernel': r['Kernel_Name'].replace('void ss::(anonymous namespace)::', '').split('(')[0],
                                                     'us': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'c': {}})
        e['c'][r['Counter_Name']] = e['c'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    rows = [disp[k] for k in sorted(disp)]
    assert len(rows) == LAUNCHES * len(SHAPES), (d, len(rows))
    return [rows[i * LAUNCHES + LAUNCHES - 1] for i in range(len(SHAPES))]


from kernel_sha import kernel_sources_sha      # noqa: E402


def main():
    fetch, write, sq = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
    out, lines = [], []
    sha = kernel_sources_sha()
    for i, name in enumerate(SHAPES):
        c = sq[i]['c']
        rd, wr = 2.0 * fetch[i]['c']['FETCH_SIZE'] * 1024, write[i]['c']['WRITE_SIZE'] * 1024
        cyc = c['GRBM_GUI_ACTIVE'] / 8.0                                  # summed over the 8 XCDs
        busy = 100.0 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024) if 'SQ_VALU_MFMA_BUSY_CYCLES' in c else None
        rec = {'shape': name, 'class': CLASSES[i], 'kernel': sq[i]['kernel'], 'us': sq[i]['us'], 'tflops': FLOPS[i] / sq[i]['us'] / 1e6,
               'hbm_read_bytes': rd, 'hbm_write_bytes': wr, 'mfma_busy_pct': busy, 'kernel_sources_sha': sha,
               'lds_bank_conflict_cycles': c.get('SQ_LDS_BANK_CONFLICT'), 'lds_idx_active_cycles': c.get('SQ_LDS_IDX_ACTIVE')}
        out.append(rec)
        lines.append(f"{name:72s} {rec['kernel'][:58]:58s} {rec['us']:7.1f} us {rec['tflops']:6.1f} TF  HBM read {rd / 1e6:6.1f} MB  write {wr / 1e6:5.1f} MB"
                     + (f"  MFMA pipe busy {busy:4.1f} %" if busy is not None else ''))
    json.dump(out, open(sys.argv[4] + '.json', 'w'), indent=1)
    with open(sys.argv[4] + '.txt', 'w') as f:
        f.write('# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE\n'
                '#   -- python3 tools/gemm_pmc.py 1024      (three separate passes; 4 launches per shape, last one listed; timings are under counter collection)\n'
                '# HBM read bytes = 2 x FETCH_SIZE x 1024, write bytes = WRITE_SIZE x 1024 (gfx950 corrections of MI355X_MICROARCH.md)\n')
        f.write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
