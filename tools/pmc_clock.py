"""Effective shader clock per kernel from a rocprofv3 `--pmc GRBM_GUI_ACTIVE --kernel-trace` pass: GRBM_GUI_ACTIVE / kernel wall time
(MI355X_MICROARCH.md, DVFS give-back).  The counter may be summed over the 8 XCDs (then the figure is 8x the clock): the RATIO between
two runs of the same kernel is what the evidence needs.  Usage: python tools/pmc_clock.py <dir> [out.json]"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name):
    m = re.search(r'dc::(?:\(anonymous namespace\)::)?(\w+)', name)
    return m.group(1) if m else name.split('(')[0][:60]


def main():
    d = sys.argv[1]
    dur = {}
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                dur[row.get('Dispatch_Id')] = float(row['End_Timestamp']) - float(row['Start_Timestamp'])
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                if row['Counter_Name'] != 'GRBM_GUI_ACTIVE':
                    continue
                ns = dur.get(row.get('Dispatch_Id'))
                if ns is None and 'Start_Timestamp' in row:
                    ns = float(row['End_Timestamp']) - float(row['Start_Timestamp'])
                if not ns:
                    continue
                a = acc[short(row['Kernel_Name'])]
                a[0] += 1; a[1] += float(row['Counter_Value']); a[2] += ns
    res = {k: {'launches': v[0], 'gui_active_per_launch': v[1] / v[0], 'avg_us': v[2] / v[0] / 1e3,
               'gui_active_per_ns': v[1] / v[2]} for k, v in acc.items() if v[0]}
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], 'w'), indent=1)
    for k in sorted(res, key=lambda k: -res[k]['avg_us'] * res[k]['launches'])[:16]:
        r = res[k]
        print('%-34s n=%-4d avg %9.1f us  GRBM_GUI_ACTIVE/ns %.3f' % (k[:34], r['launches'], r['avg_us'], r['gui_active_per_ns']))


if __name__ == '__main__':
    main()
