#!/usr/bin/env python
"""Drop the fp16x2 entries (direct tiles, Winograd choices, chain decisions) of the given batch sizes from the shipped tile table, so that
tools/make_tune_table.py measures them again on the current kernels (the table only measures what it does not know).

    python tools/prune_tune_keys.py 1 2 && python tools/make_tune_table.py --only r50_b1 r50_b2 ...
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from yolact_amd import engine
    batches = {int(a) for a in sys.argv[1:]}
    path = os.path.join(engine.TUNE_DIR, 'gfx950.json')
    entries = engine._read_table_file(path)
    keep = {}
    for k, v in entries.items():
        m = re.match(r'^(?:instep\|)?(?:wino|chain1|chain)?\((\d+),', k)
        if m and int(m.group(1)) in batches and k.endswith('|h2'):
            continue
        keep[k] = v
    import torch
    engine._write_table_file(path, keep, torch.device('cuda', 0) if torch.cuda.is_available() else None)
    print('dropped %d of %d entries (batches %s)' % (len(entries) - len(keep), len(entries), sorted(batches)))


if __name__ == '__main__':
    main()
