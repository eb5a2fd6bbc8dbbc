"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output).

MI355X_MICROARCH.md "HBM [CDNA4]": on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read
(requests tallied at 64 B instead of 128 B) -> doubled here; both counters are in KiB-like units of
1 KB per rocprofv3's definition (FETCH_SIZE/WRITE_SIZE are reported in kilobytes).  WRITE_SIZE is
uncalibrated (taken as reported).  Output: {kernel: {launches, fetch_bytes, write_bytes, bytes}} with
per-launch averages.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r'dc::(?:\(anonymous namespace\)::)?(\w+)', name)
    return m.group(1) if m else name.split('(')[0][:60]


def collect(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    for f in files:
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                k = short(row['Kernel_Name'])
                acc[k][0] += 1
                acc[k][1] += float(row['Counter_Value'])
    return acc, files


def main():
    dfetch, dwrite, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else None
    fe, f1 = collect(dfetch, 'FETCH_SIZE')
    wr, f2 = collect(dwrite, 'WRITE_SIZE')
    res = {}
    for k in sorted(set(fe) | set(wr)):
        nf, vf = fe.get(k, [0, 0.0])
        nw, vw = wr.get(k, [0, 0.0])
        fetch = 2.0 * vf * 1024.0 / max(nf, 1)      # KB -> B, gfx950 x2 correction
        write = vw * 1024.0 / max(nw, 1)
        res[k] = {'launches': max(nf, nw), 'fetch_bytes': round(fetch), 'write_bytes': round(write),
                  'bytes': round(fetch + write)}
    meta = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per-launch averages',
            'corrections': 'FETCH_SIZE x2 (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; KB->B x1024',
            'workload': workload, 'files': [os.path.basename(x) for x in f1 + f2]}
    with open(out, 'w') as fh:
        json.dump({'meta': meta, 'kernels': res}, fh, indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]['bytes'])[:12]:
        print('%-32s n=%-4d fetch %10.3f MB  write %10.3f MB' % (k, v['launches'], v['fetch_bytes'] / 1e6, v['write_bytes'] / 1e6))


if __name__ == '__main__':
    main()
