"""Prints the top kernels of a rocprofv3 --kernel-trace --stats output directory: python3 scripts/kstats.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    print("%-64s calls %6s avg %9.2f us  %5.1f%%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
