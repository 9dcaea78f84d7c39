#!/usr/bin/env python3
"""Print the kernels of the last complete training step in a rocprofv3 --kernel-trace CSV in launch order:
start offset, duration, queue, name, grid.  Shows what is serial and what overlaps."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
step = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
qs = {}
for r in step:
    q = qs.setdefault(r['Queue_Id'], len(qs))
    nm = r['Kernel_Name'].replace('void ss::(anonymous namespace)::', '').replace('ss::(anonymous namespace)::', '').split('(')[0]
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    g = f"{int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}"
    print(f"{s / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q} {'    ' * q}{nm[:60]} [{g}]")
