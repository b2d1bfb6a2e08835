"""Device timeline of a few steady-state steps: python3 scripts/timeline.py <rocprofv3 dir> [n_dispatches]
(kernel-trace CSV; prints start offset, duration and queue of the dispatches around the middle of the run)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ks = [i for i, r in enumerate(rows) if "k_step_default_b" in r[2]]
i0 = ks[len(ks) * 3 // 4]
t0 = rows[i0][0]
for s, e, name, q in rows[i0:i0 + n]:
    print("%9.1f us  +%7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name[:70]))
