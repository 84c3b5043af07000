"""Per-kernel averages of PMC counters from a rocprofv3 rocpd db: python tools/pmc_report.py file.db [name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select kernel_name, grid_size, counter_name, dispatch_id, sum(value), max(duration) from counters_collection "
                  "group by kernel_name, grid_size, counter_name, dispatch_id").fetchall()
agg = {}
for k, g, c, d, v, dur in rows:
    if flt and flt not in k:
        continue
    key = (k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], g)
    a = agg.setdefault(key, {})
    s = a.setdefault(c, [0.0, 0])
    s[0] += v
    s[1] += 1
    t = a.setdefault("_dur_ns", [0.0, 0])
    if c == rows[0][2]:
        t[0] += dur
        t[1] += 1
for (k, g), a in sorted(agg.items()):
    print(f"{k} grid={g}")
    for c, (s, n) in sorted(a.items()):
        print(f"    {c:34s} {s / max(n, 1):16.1f}  (avg of {n})")
