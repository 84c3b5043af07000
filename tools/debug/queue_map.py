"""Which HIP stream ran on which hardware queue: distinct (stream, queue) pairs of a rocprofv3 kernel trace with kernel counts and
the busiest kernel names.   python tools/debug/queue_map.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
sc = [c for c in cols if "stream" in c.lower()]
qc = [c for c in cols if "queue" in c.lower()]
print("columns:", cols)
sel = ", ".join(sc + qc)
rows = db.execute(f"select {sel}, count(*), sum(duration), min(name), max(name) from kernels group by {sel} order by 3 desc").fetchall()
for r in rows:
    print(r)
