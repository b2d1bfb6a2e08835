"""Duration percentiles of the kernels whose name contains <substr>: python3 scripts/khist.py <rocprofv3 dir> <substr> [...]"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
for sub in sys.argv[2:]:
    d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if sub in r["Kernel_Name"]]) / 1e3
    if len(d) == 0: continue
    print("%-28s n %5d  min %7.1f  p10 %7.1f  p50 %7.1f  p90 %7.1f  max %7.1f  mean %7.1f us" % (sub, len(d), d.min(), *np.percentile(d, [10, 50, 90]), d.max(), d.mean()))
