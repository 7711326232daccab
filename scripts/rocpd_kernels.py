"""per-kernel statistics out of a rocprofv3 rocpd database (the default output format of rocprofv3 7.x when no --output-format
is given): name, calls, average / minimum duration, share.  usage: rocpd_kernels.py results.db [substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ''
rows = db.execute("select name, count(*), avg(end - start), min(end - start), sum(end - start) from kernels group by name order by sum(end - start) desc").fetchall()
tot = sum(r[4] for r in rows)
print('Name,Calls,AverageNs,MinNs,TotalDurationNs,Percentage')
for r in rows:
  if sub in r[0]:
    print('"%s",%d,%.0f,%d,%d,%.2f' % (r[0][:100], r[1], r[2], r[3], r[4], 100.0 * r[4] / tot))
