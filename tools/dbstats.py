"""Per-kernel count / average / minimum duration (us) from a rocprofv3 results database (the .db rocprofv3 writes by default)."""
import glob
import sqlite3
import sys

db = sys.argv[1] if sys.argv[1].endswith(".db") else glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
for r in c.execute(q):
    print("%-72s %6d %9.2f %9.2f" % (r[0][:72], r[1], r[2], r[3]))
