"""Idle time between kernels of the stepping stream: python3 scripts/gaps.py <rocprofv3 output dir>
Reads *kernel_trace.csv, keeps the last 60 % of the dispatches (the steady stepping loop) and prints, per kernel, its mean
duration and the mean idle gap that FOLLOWS it on the device timeline (start of the next dispatch - its end)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
rows = rows[int(len(rows) * 0.4):]
span = rows[-1][1] - rows[0][0]
busy = 0; cur_end = rows[0][0]
per = {}
for i, (s, e, n) in enumerate(rows):
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
    g = rows[i + 1][0] - e if i + 1 < len(rows) else 0
    d = per.setdefault(n[:56], [0, 0, 0]); d[0] += 1; d[1] += e - s; d[2] += g
print("span %.1f ms busy %.1f ms (%.1f %%) dispatches %d" % (span / 1e6, busy / 1e6, 100.0 * busy / span, len(rows)))
for n, (c, d, g) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-56s n %6d dur %8.2f us gap-after %7.2f us" % (n, c, d / c / 1e3, g / c / 1e3))
