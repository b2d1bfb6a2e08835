# usage (GPU box): bash scripts/dev_idle_gaps.sh [bench args] -- where the device idles during bench.py's iterations: gaps between consecutive
# kernels (all streams) inside the timed iterations (delimited by the nonbonded launches), by the kernel that follows the gap
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/gaps; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-single --no-kernel-timing "$@" > $out/log 2>&1
tail -1 $out/log | cut -c1-200
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/t/*/*kernel_trace.csv")[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
k1 = [r for r in rows if r[2].startswith("void k_nonbonded_atom_b<false>")]
lo, hi = k1[len(k1) // 3 + 50][0], k1[-130][1]      # from early in the first timed iteration (1 warm-up + 2 timed) to before the stand-alone launches at the end
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
busy_end = rows[0][1]; idle = 0; by = collections.Counter(); cnt = collections.Counter(); big = []
for s, e, n in rows[1:]:
    if s > busy_end:
        g = s - busy_end; idle += g; by[n[:40]] += g; cnt[n[:40]] += 1
        if g > 200e3: big.append((g / 1e3, n[:50]))
    busy_end = max(busy_end, e)
wall = rows[-1][1] - rows[0][0]
print("window %.1f ms, idle %.1f ms (%.1f %%)" % (wall / 1e6, idle / 1e6, 100.0 * idle / wall))
for n, g in by.most_common(14):
    print("   %-42s gaps %5d  total %8.2f ms  mean %7.1f us" % (n, cnt[n], g / 1e6, g / cnt[n] / 1e3))
print("gaps > 200 us:", len(big), "sum %.1f ms" % (sum(b[0] for b in big) / 1e3))
for g, n in sorted(big, reverse=True)[:12]:
    print("   %8.1f us before %s" % (g, n))
PY
rm -rf $out/t
