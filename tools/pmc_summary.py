#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv files under a directory): mean counter value per
dispatch, grouped by (kernel, grid size).  Usage: python tools/pmc_summary.py DIR [regex]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    return n[:60]


def main(root, pat='.'):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    seen = set()
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r.get('Kernel_Name', '')
                if not re.search(pat, name):
                    continue
                key = (short(name), r.get('Grid_Size', ''))
                a = acc[key][r['Counter_Name']]
                a[0] += float(r['Counter_Value']); a[1] += 1
                did = (f, r.get('Dispatch_Id'))
                if did not in seen and r.get('Start_Timestamp'):
                    seen.add(did)
                    d = dur[key]
                    d[0] += float(r['End_Timestamp']) - float(r['Start_Timestamp']); d[1] += 1
    ctrs = sorted({c for v in acc.values() for c in v})
    print('\t'.join(['kernel', 'grid', 'n', 'avg_us(profiled)'] + ctrs))
    for key in sorted(acc, key=lambda k: -dur[k][0]):
        v = acc[key]
        n = max(a[1] for a in v.values())
        us = dur[key][0] / max(dur[key][1], 1) / 1e3
        print('\t'.join([key[0], key[1], str(n), '%.1f' % us] + ['%.4g' % (v[c][0] / v[c][1]) if c in v else '-' for c in ctrs]))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '.')
