"""HBM traffic per launch of the hot kernels from two rocprofv3 PMC passes (one counter per pass, as
MI355X_MICROARCH.md prescribes):   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d A -o p -- python3 bench.py ...
                                   rocprofv3 --pmc WRITE_SIZE --kernel-trace -d B -o p -- python3 bench.py ...
usage: python tools/pmc_traffic.py A/p_results.db B/p_results.db precision out.json [round] [forward steps in the profiled run]
Also the WHOLE-STEP figure north_star asks for: the sum over every kernel dispatch of the run / the number of forward steps.
gfx950 correction: FETCH_SIZE counts 64 B per 128 B request and both counters are in KiB: bytes = (2*FETCH + WRITE) * 1024."""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    rows = sqlite3.connect(db).execute(
        "select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name = ? group by kernel_name, dispatch_id",
        (counter,)).fetchall()
    agg = {}
    for k, _, v in rows:
        k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = agg.setdefault(k, [0.0, 0])
        a[0] += v
        a[1] += 1
    return agg


fdb, wdb, prec, out = sys.argv[1:5]
rnd = int(sys.argv[5]) if len(sys.argv) > 5 else 1
nsteps = int(sys.argv[6]) if len(sys.argv) > 6 else 3          # bench.py --steps 2 --warmup 1
f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
res = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-legs --no-latency --no-cpu-baseline "
                  f"--precision {prec} (second pass: --pmc WRITE_SIZE)",
       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128 B request; KB units)",
       "round": rnd, "precision": prec, "kernels": {}}
step_bytes = 0.0
for k in sorted(f, key=lambda k: -f[k][0]):
    if k not in w:
        continue
    fk, wk = f[k][0] / f[k][1], w[k][0] / w[k][1]
    step_bytes += (2 * f[k][0] + w[k][0]) * 1024 / nsteps
    res["kernels"][k] = {"launches": f[k][1], "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
                         "hbm_bytes_per_launch_corrected": int((2 * fk + wk) * 1024),
                         "hbm_bytes_per_step_corrected": int((2 * f[k][0] + w[k][0]) * 1024 / nsteps)}
res["forward_steps_in_run"] = nsteps
res["step_hbm_bytes_corrected"] = int(step_bytes)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
