"""Summarise a rocprofv3 (ROCm 7.2) rocpd sqlite database: per-kernel calls / total / average / MEDIAN duration,
keyed by kernel name + grid so differently shaped launches of one template stay apart.
usage: python tools/rocpd_summary.py gpurun_out/prof/xyz_results.db [--skip-first K] [--sites SUBSTRING N]
--sites SUBSTRING N: the kernels whose name holds SUBSTRING are launched at N sites per forward in all (the 128-196-256 SA kernels:
enc.sa2 branch 1 -- the table form, one instantiation -- then the two hands' mano.sa1 branch 1 -- the feature-row form, another
instantiation since round 5); each such kernel is also listed per site (its share of the N sites, dispatch order modulo that share),
so that the site bench.py brackets with HIP events (sa2.1 = the table form's only site) can be compared with its own trace average
instead of the average over three different inputs.
--skip-first K (default 1) drops the first K dispatches of every (kernel, grid) key: the very first launch of a kernel pays code
loading / page faults (round 2's table had a 22 ms `ball_query` outlier turning a 4 % kernel into "14.6 %"); the `%` column is
computed from the remaining dispatches, and the median is printed next to the average so that an outlier shows."""
import sqlite3
import statistics
import sys


def main():
    skip = 1
    sites = None
    if "--sites" in sys.argv:
        i = sys.argv.index("--sites")
        sites = (sys.argv[i + 1], int(sys.argv[i + 2]))
        del sys.argv[i:i + 3]
    if "--skip-first" in sys.argv:
        i = sys.argv.index("--skip-first")
        skip = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select name, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, scratch_size, duration, start "
                       "from kernels order by start").fetchall()
    agg = {}
    site_durs = {}
    for name, gx, gy, wx, lds, vg, ag, scr, dur, start in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0]
        key = (short, gx // max(wx, 1), gy, wx)
        if sites and sites[0] in short:
            lst = site_durs.setdefault(key, [])
            lst.append(dur)
        a = agg.setdefault(key, {"seen": 0, "durs": [], "lds": lds, "vgpr": vg, "agpr": ag, "scr": scr})
        a["seen"] += 1
        if a["seen"] > skip:
            a["durs"].append(dur)
    agg = {k: a for k, a in agg.items() if a["durs"]}
    for a in agg.values():
        a["calls"], a["total"], a["min"], a["max"] = len(a["durs"]), sum(a["durs"]), min(a["durs"]), max(a["durs"])
        a["med"] = statistics.median(a["durs"])
    # runtime copy / fill kernels (the ~950 weight tensors uploaded once at start-up) are listed but kept out of the % base
    tot = sum(a["total"] for k, a in agg.items() if not k[0].startswith("__amd_rocclr_"))
    print(f"# {sys.argv[1]}: {len(rows)} dispatches ({skip} warm-up dispatch(es) per kernel shape dropped), {tot / 1e6:.3f} ms of kernel time "
          f"(runtime copy/fill kernels excluded from the total and the % column)")
    print(f"{'kernel':58s} {'blocks':>9s} {'wg':>5s} {'calls':>6s} {'avg_us':>10s} {'median_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>9s} {'%':>6s} {'vgpr':>5s} {'agpr':>5s} {'lds':>7s} {'scratch':>7s}")
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1]["total"]):
        short, blocks, gy, wx = key
        print(f"{short[:58]:58s} {blocks * max(gy, 1):9d} {wx:5d} {a['calls']:6d} {a['total'] / a['calls'] / 1e3:10.1f} {a['med'] / 1e3:10.1f} {a['min'] / 1e3:10.1f} "
              f"{a['max'] / 1e3:10.1f} {a['total'] / 1e6:9.3f} {(100.0 * a['total'] / tot if not short.startswith('__amd_rocclr_') else float('nan')):6.2f} {a['vgpr']:5d} {a['agpr']:5d} {a['lds']:7d} {a['scr']:7d}")


    if sites:
        total = sum(len(d) for d in site_durs.values())
        for key, durs in site_durs.items():
            n = max(1, round(len(durs) * sites[1] / max(total, 1)))      # this kernel's share of the N sites of a forward
            print(f"# launch sites of {key[0]} ({key[1] * max(key[2], 1)} blocks), dispatch order modulo {n}, first forward dropped:")
            for k in range(n):
                d = durs[n + k::n]
                if d:
                    print(f"#   site {k}: {len(d)} launches, avg {sum(d) / len(d) / 1e3:.1f} us, median {statistics.median(d) / 1e3:.1f} us, "
                          f"min {min(d) / 1e3:.1f}, max {max(d) / 1e3:.1f}")


if __name__ == "__main__":
    main()
