"""Per-dispatch durations of the kernels whose name contains PATTERN, in launch order (rocprofv3 rocpd sqlite): the last N dispatches.
usage: python tools/rocpd_dispatches.py <db> <pattern> [N]"""
import sqlite3
import sys

db, pat = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
c = sqlite3.connect(db)
views = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
v = 'kernels' if 'kernels' in views else [x for x in views if 'kernel_dispatch' in x][0]
cols = [r[1] for r in c.execute('pragma table_info(%s)' % v)]
name = [x for x in cols if x in ('name', 'kernel_name')][0]
start = [x for x in cols if x in ('start', 'start_timestamp')][0]
end = [x for x in cols if x in ('end', 'end_timestamp')][0]
grid = [x for x in cols if x.startswith('grid') and x.endswith('x')]
q = 'select %s, %s, %s%s from %s where %s like ? order by %s' % (name, start, end, (', ' + grid[0]) if grid else '', v, name, start)
rows = c.execute(q, ('%' + pat + '%',)).fetchall()
for r in rows[-n:]:
    short = r[0].replace('(anonymous namespace)::', '').split('(')[0][-60:]
    print('%-60s %9.1f us%s' % (short, (r[2] - r[1]) / 1e3, ('  grid %d' % r[3]) if grid else ''))
