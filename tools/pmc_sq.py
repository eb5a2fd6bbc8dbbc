"""Per-kernel averages of the SQ counters of one rocprofv3 --pmc pass (csv output).
Usage: python tools/pmc_sq.py <dir> [out.json]"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name):
    m = re.search(r'dc::(?:\(anonymous namespace\)::)?(\w+)', name)
    return m.group(1) if m else name.split('(')[0][:50]


def main():
    d = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                a = acc[short(row['Kernel_Name'])][row['Counter_Name']]
                a[0] += 1
                a[1] += float(row['Counter_Value'])
    res = {k: {c: v[1] / max(v[0], 1) for c, v in cs.items()} for k, cs in acc.items()}
    for k, cs in res.items():
        cs['launches'] = max(v[0] for v in acc[k].values())
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], 'w'), indent=1)
    keys = sorted(res, key=lambda k: -res[k].get('SQ_WAVE_CYCLES', 0) * res[k]['launches'])[:14]
    for k in keys:
        c = res[k]
        wc = c.get('SQ_WAVE_CYCLES', 0) or 1
        print('%-30s n=%-3d' % (k[:30], c['launches']) + ' '.join('%s=%.3g' % (n.replace('SQ_', ''), v) for n, v in sorted(c.items()) if n != 'launches'))
        print('    of WAVE_CYCLES: wait_any %.2f  wait_inst_any %.2f  active_inst %.2f | mfma_busy/busy_cycles %.3f | lds conflict/idx_active %.3f'
              % (c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_ANY', 0) / wc,
                 c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(c.get('SQ_BUSY_CYCLES', 1), 1),
                 c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1)))


if __name__ == '__main__':
    main()
