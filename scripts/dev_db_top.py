"""Dev: top kernels of a rocprofv3 results database.  usage: dev_db_top.py <results.db> [iterations] [rows]"""
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
it = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = ("select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from %s d join %s s on d.kernel_id=s.id "
     "group by s.kernel_name order by 3 desc" % (kd, ks))
tot = 0.0
for i, r in enumerate(c.execute(q)):
    tot += r[2]
    if i < rows:
        print('%7.1f calls %9.1f us/iter %8.1f avg  %s' % (r[1] / it, r[2] / it, r[3], r[0][:120]))
print('total %.1f us/iter' % (tot / it))
