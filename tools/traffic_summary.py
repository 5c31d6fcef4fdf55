#!/usr/bin/env python
"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of bench.py into HBM bytes per launch per conv
kernel, keyed like bench.py's roofline.kernel.  FETCH_SIZE is doubled (gfx950 counts 128-byte requests as 64 B,
MI355X_MICROARCH.md "HBM").   python tools/traffic_summary.py FETCH_DIR WRITE_DIR > profiles/r02_traffic.json"""
import csv, glob, json, os, re, sys
from collections import defaultdict

TILES = {(2, 2, 1, 2, 2, 2): '128x128', (2, 2, 1, 2, 1, 2): '128x64', (2, 2, 1, 1, 1, 2): '64x64', (4, 1, 1, 1, 1, 2): '128x32',
         (2, 2, 1, 1, 2, 2): '64x128', (1, 1, 4, 1, 1, 2): '32x32k4', (2, 1, 2, 1, 1, 2): '64x32k2', (1, 2, 2, 1, 1, 2): '32x64k2',
         (2, 2, 1, 1, 1, 3): '64x64s3', (2, 2, 1, 1, 1, 4): '64x64s4', (2, 2, 1, 1, 2, 3): '64x128s3', (2, 2, 1, 2, 1, 3): '128x64s3',
         (1, 1, 4, 1, 1, 4): '32x32k4s4', (2, 1, 2, 1, 1, 3): '64x32k2s3', (1, 2, 2, 1, 1, 3): '32x64k2s3',
         (4, 2, 1, 1, 2, 2): '128x128w8', (4, 2, 1, 2, 2, 2): '256x128w8', (4, 2, 1, 1, 4, 2): '128x256w8',
         (2, 2, 1, 2, 2, 3): '128x128s3', (4, 2, 1, 1, 2, 3): '128x128w8s3', (4, 2, 1, 2, 2, 3): '256x128w8s3',
         (4, 2, 1, 1, 2, 4): '128x128w8s4'}


# pipe_h2_k<WM, WN, TM, TN, RING, PLAIN> (csrc/dcn.hip) -> the tile names of yolact_amd/_lib.py DCNP_TILES
PIPE = {(2, 2, 1, 2, 2): 'dcnp64x128', (2, 4, 1, 1, 2): 'dcnp64x128w8', (2, 2, 1, 1, 2): 'dcnp64x64', (4, 2, 1, 2, 2): 'dcnp128x128w8',
        (4, 2, 1, 1, 2): 'dcnp128x64w8', (1, 4, 1, 1, 2): 'dcnp32x128', (3, 2, 1, 2, 1): 'dcnp96x128w6', (4, 2, 1, 2, 1): 'dcnp128x128w8r1',
        (5, 2, 1, 2, 1): 'dcnp160x128w10', (6, 2, 1, 2, 1): 'dcnp192x128w12', (2, 4, 1, 2, 1): 'dcnp64x256w8',
        (3, 4, 1, 2, 1): 'dcnp96x256w12', (4, 4, 1, 2, 1): 'dcnp128x256w16', (2, 4, 2, 2, 1): 'dcnp128x256w8t',
        (2, 2, 2, 2, 1): 'dcnp128x128w4t', (4, 2, 2, 2, 1): 'dcnp256x128w8t'}
CONV = ('conv_igemm', 'pipe_h2_k')


def grouped_dispatches(counter_csv):
    """Dispatch ids of launches with gridDim.y > 1 (the grouped Winograd GEMMs), from the kernel trace of the same run:
    the pointwise loader (template LOADER 3) serves both the 1x1 convolutions and the grouped GEMMs."""
    out = set()
    for f in glob.glob(os.path.join(os.path.dirname(counter_csv), '*kernel_trace.csv')):
        for r in csv.DictReader(open(f)):
            if 'conv_igemm' in r['Kernel_Name'] and int(r.get('Grid_Size_Y', 1) or 1) > 1:
                out.add(r['Dispatch_Id'])
    return out


def steady_rows(f, counter):
    """Rows of one process in dispatch order, cut to the trailing part that repeats with the plan's period (the bench
    passes), so that launches made while autotuning (other tiles, other layers) do not pollute the per-kernel means."""
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter and any(c in r['Kernel_Name'] for c in CONV)]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    names = [r['Kernel_Name'] for r in rows]
    for L in range(20, 400):
        if 3 * L <= len(names) and names[-L:] == names[-2 * L:-L] == names[-3 * L:-2 * L]:
            n = 3
            while (n + 1) * L <= len(names) and names[-(n + 1) * L:-n * L] == names[-L:]:
                n += 1
            return rows[-n * L:]
    return rows


def collect(root, counter):
    acc = defaultdict(lambda: [0.0, 0])
    files = sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:                      # the newest run only (gpurun_out/ accumulates earlier sessions)
        grouped = grouped_dispatches(f)
        for r in steady_rows(f, counter):
            mp = re.search(r'pipe_h2_k<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false|\(bool\)1|\(bool\)0|1|0)>', r['Kernel_Name'])
            if mp:
                vp = tuple(int(x) for x in mp.groups()[:5])
                plain = mp.group(6) in ('true', '(bool)1', '1')
                a = acc['pipe_h2_k<%s,%s>' % (PIPE.get(vp, str(vp)), 'convolution' if plain else 'DCNv2 gather')]
                a[0] += float(r['Counter_Value']); a[1] += 1
                continue
            m = re.search(r'conv_igemm_f32<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>', r['Kernel_Name'])
            if not m:
                continue
            v = tuple(int(x) for x in m.groups())
            tile = TILES.get(v[:6], str(v[:6])) + ('x3' if v[7] in (1, 2) else 'h2' if v[7] >= 3 else '')
            key = ('conv_igemm_f32<%s,winograd grouped GEMM>' % tile) if (v[6] == 3 and r['Dispatch_Id'] in grouped) else \
                'conv_igemm_f32<%s,loader%d>' % (tile, v[6])
            a = acc[key]
            a[0] += float(r['Counter_Value']); a[1] += 1
    return acc


def main(fdir, wdir):
    fe, wr = collect(fdir, 'FETCH_SIZE'), collect(wdir, 'WRITE_SIZE')
    out = {}
    for k in sorted(set(fe) | set(wr)):
        f = fe[k][0] / max(fe[k][1], 1) * 1024 * 2      # KB -> bytes, x2 gfx950 correction
        w = wr[k][0] / max(wr[k][1], 1) * 1024
        out[k] = {'hbm_read_bytes_per_launch': round(f), 'hbm_write_bytes_per_launch': round(w),
                  'bytes_per_launch': round(f + w), 'launches_sampled': fe[k][1],
                  'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2'}
    # bench.py reports these figures only while it runs the tile table they were measured with
    import hashlib
    tp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'yolact_amd', 'tune', 'gfx950.json')
    out['tune_sha'] = hashlib.sha256(open(tp, 'rb').read()).hexdigest()[:16] if os.path.exists(tp) else None
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
