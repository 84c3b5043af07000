"""Timeline of ONE forward at B = 1 from a rocprofv3 kernel trace (the last of 12 forwards): start offset, duration and stream of every
kernel, and the gaps on the critical path.   rocprofv3 --kernel-trace -d D -o t -- python3 tools/debug/latency_trace.py ; then
python tools/debug/latency_timeline.py D/.../t_results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, stream_id, grid_x, workgroup_x from kernels order by start").fetchall()
names = [r[0] for r in rows]
# one forward = from a prep_points kernel to the next one
starts = [i for i, n in enumerate(names) if "prep_points" in n]
i0, i1 = starts[-2], starts[-1]
fw = rows[i0:i1]
t0 = fw[0][1]
end = max(r[2] for r in fw)
print(f"forward: {len(fw)} kernels, {(end - t0) / 1e3:.1f} us from first start to last end; sum of kernel durations {sum(r[2] - r[1] for r in fw) / 1e3:.1f} us")
last_end = {}
prev_end = t0
for name, s, e, st, gx, wx in fw:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:46]
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f}  stream {st}  gap_since_prev_end {(s - prev_end) / 1e3:6.1f}  blocks {gx // max(wx, 1):6d}  {short}")
    prev_end = max(prev_end, e)
