#!/usr/bin/env python3
"""Where the data-parallel buckets sit in the step: reads a rocprofv3 --kernel-trace CSV of `bench.py --force-dp --tune dp_model=R`
(every collective replaced by a stand-in kernel of the modelled duration, ss_tune("dp_model")) and prints, for the last complete step,
each bucket's start / end relative to the step and to the end of the backward, and what the collectives add to the step's end.
No multi-GPU node is involved: this is a MODEL of the schedule, not a measurement of RCCL."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
step = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
rel = lambda r, k: (int(r[k]) - t0) / 1e3
coll = [r for r in step if 'dp_model_kernel' in r['Kernel_Name']]
work = [r for r in step if 'dp_model_kernel' not in r['Kernel_Name'] and 'adam' not in r['Kernel_Name']]
bwd_end = max(rel(r, 'End_Timestamp') for r in work)
adam = [r for r in step if 'adam_kernel' in r['Kernel_Name']][0]
print(f'# last step: {len(step)} dispatches, backward (last non-collective kernel) ends at {bwd_end:.1f} us, Adam starts at {rel(adam, "Start_Timestamp"):.1f} us')
print('# bucket   start_us     end_us  duration_us   end - backward_end')
for i, r in enumerate(coll):
    s, e = rel(r, 'Start_Timestamp'), rel(r, 'End_Timestamp')
    print(f'  {i:2d}    {s:9.1f}  {e:9.1f}   {e - s:9.1f}   {e - bwd_end:+9.1f}')
if coll:
    last = max(rel(r, 'End_Timestamp') for r in coll)
    tot = sum(rel(r, 'End_Timestamp') - rel(r, 'Start_Timestamp') for r in coll)
    print(f'# {len(coll)} collectives, {tot:.0f} us of modelled communication; the last one ends {last - bwd_end:+.1f} us after the backward '
          f'(that much is on the critical path in front of Adam)')
