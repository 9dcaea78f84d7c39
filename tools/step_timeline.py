#!/usr/bin/env python3
"""Print the kernels of the last complete training step in a rocprofv3 --kernel-trace CSV in launch order:
start offset, duration, queue, name, grid.  Shows what is serial and what overlaps."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step ends with its LAST adam_kernel launch (with the early decoder-range update a step has two: the final one is the adam_kernel whose
# next Adam-related launch is the next step's adam_prepare_kernel)
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name'] or 'adam_prepare_kernel' in r['Kernel_Name']]
idx = [i for n, i in enumerate(adam) if 'adam_kernel' in rows[i]['Kernel_Name'] and (n + 1 == len(adam) or 'adam_prepare_kernel' in rows[adam[n + 1]]['Kernel_Name'])]
step = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
qs = {}
for r in step:
    q = qs.setdefault(r['Queue_Id'], len(qs))
    nm = r['Kernel_Name'].replace('void ss::(anonymous namespace)::', '').replace('ss::(anonymous namespace)::', '').split('(')[0]
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    g = f"{int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}"
    print(f"{s / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q} {'    ' * q}{nm[:60]} [{g}]")
