#!/usr/bin/env python
"""Timeline analysis of a rocprofv3 --kernel-trace CSV of `bench.py`: for the steady-state steps (the timed region),
per step: wall time, time with >= 1 kernel running (busy), idle gaps, time with two kernels overlapping, and the summed
kernel time per queue.      python tools/step_timeline.py <..._kernel_trace.csv> [steps_to_analyse]"""
import csv, sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r['Kind'] == 'KERNEL_DISPATCH']
rows.sort(key=lambda r: int(r['Start_Timestamp']))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
# a step starts at the input layout kernel
starts = [i for i, r in enumerate(rows) if 'nchw_to_nhwc4_k' in r['Kernel_Name'] or 'stem_pool_k' in r['Kernel_Name']]
starts = starts[-(nsteps + 4):-3]              # drop the 3 serialised roofline passes at the end, keep the last timed steps
tot = defaultdict(float)
for a, b in zip(starts[:-1], starts[1:]):
    ks = rows[a:b]
    t0 = int(ks[0]['Start_Timestamp']); t1 = int(rows[b]['Start_Timestamp'])
    ev = []
    for r in ks:
        ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
    ev.sort()
    busy = over = 0; depth = 0; last = t0
    for t, d in ev:
        if depth >= 1: busy += t - last
        if depth >= 2: over += t - last
        depth += d; last = t
    tot['wall'] += t1 - t0; tot['busy'] += busy; tot['overlap2'] += over; tot['kernels'] += len(ks)
    tot['sum_kernel'] += sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in ks)
    for r in ks:
        tot['q%s' % r['Queue_Id']] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
n = len(starts) - 1
print('%d steps: wall %.3f ms  busy %.3f ms (idle %.3f)  >=2 kernels in flight %.3f ms  kernels/step %d  summed kernel time %.3f ms'
      % (n, tot['wall'] / n / 1e6, tot['busy'] / n / 1e6, (tot['wall'] - tot['busy']) / n / 1e6, tot['overlap2'] / n / 1e6,
         tot['kernels'] / n, tot['sum_kernel'] / n / 1e6))
print('per queue:', {k: round(v / n / 1e6, 3) for k, v in tot.items() if k.startswith('q')})
