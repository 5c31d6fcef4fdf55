#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 'rocpd' sqlite) kernel trace: per-kernel calls / total / avg / min / max.

    python tools/rocpd_stats.py gpurun_out/prof/<host>/<pid>_results.db > profiles/rNN_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'void ', '', name)
    return name if len(name) < 110 else name[:107] + '...'


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = cur.execute("select %s, start, end from kernels" % namecol).fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print('# %s: %d kernel dispatches, %.3f ms total GPU kernel time' % (path.split('/')[-1], len(rows), tot / 1e6))
    print('%-112s %7s %12s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-112s %7d %12.1f %10.2f %10.2f %10.2f %6.2f' % (short(n), a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3,
                                                          a[3] / 1e3, 100.0 * a[1] / tot))


if __name__ == '__main__':
    main(sys.argv[1])
