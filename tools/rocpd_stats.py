"""Dump the per-kernel summary of a rocprofv3 (rocpd sqlite) result as CSV: name,calls,total_us,avg_us,pct."""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
with open(out, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationUs', 'AverageUs', 'Percentage'])
    for r in rows:
        w.writerow([r[0], r[1], round(r[2], 3), round(r[3], 3), round(r[4], 4)])
print('%d kernels, %.3f ms total' % (len(rows), sum(r[2] for r in rows) / 1e3))
