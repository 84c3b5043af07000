"""Summarise a rocprofv3 (ROCm 7.2) rocpd sqlite database: per-kernel calls / total / average / MEDIAN duration,
keyed by kernel name + grid so differently shaped launches of one template stay apart.
usage: python tools/rocpd_summary.py gpurun_out/prof/xyz_results.db [--skip-first K]
--skip-first K (default 1) drops the first K dispatches of every (kernel, grid) key: the very first launch of a kernel pays code
loading / page faults (round 2's table had a 22 ms `ball_query` outlier turning a 4 % kernel into "14.6 %"); the `%` column is
computed from the remaining dispatches, and the median is printed next to the average so that an outlier shows."""
import sqlite3
import statistics
import sys


def main():
    skip = 1
    if "--skip-first" in sys.argv:
        i = sys.argv.index("--skip-first")
        skip = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select name, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, scratch_size, duration, start "
                       "from kernels order by start").fetchall()
    agg = {}
    for name, gx, gy, wx, lds, vg, ag, scr, dur, start in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0]
        key = (short, gx // max(wx, 1), gy, wx)
        a = agg.setdefault(key, {"seen": 0, "durs": [], "lds": lds, "vgpr": vg, "agpr": ag, "scr": scr})
        a["seen"] += 1
        if a["seen"] > skip:
            a["durs"].append(dur)
    agg = {k: a for k, a in agg.items() if a["durs"]}
    for a in agg.values():
        a["calls"], a["total"], a["min"], a["max"] = len(a["durs"]), sum(a["durs"]), min(a["durs"]), max(a["durs"])
        a["med"] = statistics.median(a["durs"])
    tot = sum(a["total"] for a in agg.values())
    print(f"# {sys.argv[1]}: {len(rows)} dispatches ({skip} warm-up dispatch(es) per kernel shape dropped), {tot / 1e6:.3f} ms of kernel time")
    print(f"{'kernel':58s} {'blocks':>9s} {'wg':>5s} {'calls':>6s} {'avg_us':>10s} {'median_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>9s} {'%':>6s} {'vgpr':>5s} {'agpr':>5s} {'lds':>7s} {'scratch':>7s}")
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1]["total"]):
        short, blocks, gy, wx = key
        print(f"{short[:58]:58s} {blocks * max(gy, 1):9d} {wx:5d} {a['calls']:6d} {a['total'] / a['calls'] / 1e3:10.1f} {a['med'] / 1e3:10.1f} {a['min'] / 1e3:10.1f} "
              f"{a['max'] / 1e3:10.1f} {a['total'] / 1e6:9.3f} {100.0 * a['total'] / tot:6.2f} {a['vgpr']:5d} {a['agpr']:5d} {a['lds']:7d} {a['scr']:7d}")


if __name__ == "__main__":
    main()
