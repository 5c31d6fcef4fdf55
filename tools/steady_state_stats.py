#!/usr/bin/env python
"""Per-kernel statistics of the STEADY-STATE steps of a rocprofv3 --kernel-trace CSV of `bench.py` — the timed region only: no
plan construction, no table replay, no warm-up, none of the three serialised roofline passes at the end.

    python tools/steady_state_stats.py <..._kernel_trace.csv> [steps=8]

A step starts at the first kernel of the plan (the fused stem, or the input layout kernel).  Printed: kernel, calls per step, average
duration, time per step, share of the summed kernel time; then the totals (kernels per step, summed kernel time per step, wall per
step).  What `rocprofv3 --stats` prints for the whole process is next to it in profiles/ (`r04_kernel_stats_*`): its averages agree,
its call counts include the launches that build the plan.
"""
import csv
import sys
from collections import defaultdict


def main():
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if r['Kind'] == 'KERNEL_DISPATCH']
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    starts = [i for i, r in enumerate(rows) if 'nchw_to_nhwc4' in r['Kernel_Name'] or 'stem_pool_k' in r['Kernel_Name']]
    starts = starts[-(nsteps + 4):-3]          # drop the 3 serialised roofline passes at the end, keep the last timed steps
    n = len(starts) - 1
    agg = defaultdict(lambda: [0, 0])
    wall = 0
    for a, b in zip(starts[:-1], starts[1:]):
        wall += int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])
        for r in rows[a:b]:
            e = agg[r['Kernel_Name']]
            e[0] += 1
            e[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    tot = sum(v[1] for v in agg.values())
    print('# steady state: %d consecutive timed steps of %s' % (n, sys.argv[1].split('/')[-1]))
    print('%-112s %9s %10s %12s %6s' % ('kernel', 'calls/step', 'avg_ns', 'ns/step', '%'))
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-112s %9.2f %10.0f %12.0f %6.2f' % (k[:112], c / n, t / c, t / n, 100.0 * t / tot))
    print('%-112s %9.2f %10s %12.0f' % ('TOTAL (summed kernel time per step; kernels per step)', sum(v[0] for v in agg.values()) / n, '', tot / n))
    print('wall per step %.0f ns' % (wall / n))


if __name__ == '__main__':
    main()
