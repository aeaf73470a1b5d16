"""Counts / average durations of the NON-moka kernels in a rocprofv3 rocpd result (full demangled names): what else ran.

    python tools/rocpd_names.py <results.db>
"""
import sqlite3,sys,subprocess,collections
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
rows=cur.execute("select name, count(*), avg(end-start) from kernels group by name order by count(*) desc").fetchall()
for n,c,a in rows:
    if 'moka' in n: continue
    d=subprocess.run(["c++filt",n],capture_output=True,text=True).stdout.strip()
    print(c, round(a/1e3,2), d[:200])
