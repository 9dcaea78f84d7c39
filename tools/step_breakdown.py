#!/usr/bin/env python3
"""Summarise one training step from a rocprofv3 --kernel-trace CSV: per-kernel totals, stream overlap, busy time.
usage: step_breakdown.py <kernel_trace.csv> [rows] [label of the traced configuration]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step ends with its LAST adam_kernel launch (with the early decoder-range update a step has two: the final one is the adam_kernel whose
# next Adam-related launch is the next step's adam_prepare_kernel)
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name'] or 'adam_prepare_kernel' in r['Kernel_Name']]
idx = [i for n, i in enumerate(adam) if 'adam_kernel' in rows[i]['Kernel_Name'] and (n + 1 == len(adam) or 'adam_prepare_kernel' in rows[adam[n + 1]]['Kernel_Name'])]
step = rows[idx[-2] + 1: idx[-1] + 1]
t0 = min(int(r['Start_Timestamp']) for r in step)
t1 = max(int(r['End_Timestamp']) for r in step)
ivals = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
busy, cur_s, cur_e = 0, None, None
for a, b in ivals:
    if cur_e is None or a > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = a, b
    else:
        cur_e = max(cur_e, b)
busy += cur_e - cur_s
tot = sum(b - a for a, b in ivals)
extra = sys.argv[2:]
label = next((a for a in extra if not a.isdigit()), 'Generator_3 training step (B=64, T=128, fp32)')
limit = next((int(a) for a in extra if a.isdigit()), 1000)
print(f'# one {label}: wall {(t1 - t0) / 1e3:.1f} us, GPU busy (union of kernels) {busy / 1e3:.1f} us, '
      f'sum of kernel durations {tot / 1e3:.1f} us, {len(step)} dispatches')
agg = collections.OrderedDict()
for r in step:
    nm = r['Kernel_Name'].replace('void ss::(anonymous namespace)::', '').replace('ss::(anonymous namespace)::', '').split('(')[0]
    key = (nm, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z'])
    agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(f"{'kernel':58s} {'grid(blocks)':>14s} {'calls':>5s} {'total_us':>10s} {'avg_us':>9s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:limit]:
    print(f"{k[0][:58]:58s} {str(k[1]) + 'x' + k[2] + 'x' + k[3]:>14s} {len(v):5d} {sum(v):10.1f} {sum(v) / len(v):9.1f}")
