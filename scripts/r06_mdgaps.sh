# usage (GPU box): bash scripts/r06_mdgaps.sh [R] -- where the device idles in the MD leg of a full iteration (kernel trace of one timed iteration)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-64}
out=gpurun_out/r06/mdgaps_$R; rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 bench.py --md-steps 300 --replicas $R --groups 1 --steps 1 --warmup 1 --no-cpu --no-single > $out/log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/tr/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_step_md_b" in r[2]]
i0, i1 = idx[len(idx) // 2], idx[-1]          # second half of the MD steps: the timed iteration
seg = rows[i0:i1 + 1]
t0, t1 = seg[0][0], seg[-1][1]
nstep = sum(1 for r in seg if "k_step_md_b" in r[2]) - 1
busy = 0; cur_e = t0; gap_after = collections.Counter(); gap_before = collections.Counter(); ngap = collections.Counter()
prev = None
for s, e, n in seg:
    if s > cur_e:
        g = s - cur_e
        gap_before[n[:40]] += g; ngap[n[:40]] += 1
        if prev: gap_after[prev[:40]] += g
    if e > cur_e:
        busy += e - max(s, cur_e); cur_e = e; prev = n
print("MD steps %d: %.1f us per step wall, %.1f us busy, %.1f us idle" % (nstep, (t1 - t0) / 1e3 / nstep, busy / 1e3 / nstep, (t1 - t0 - busy) / 1e3 / nstep))
print("idle before (us per step):"); [print("   %-42s %7.1f  (%d gaps, %.1f us each)" % (k, v / 1e3 / nstep, ngap[k], v / 1e3 / max(1, ngap[k]))) for k, v in gap_before.most_common(8)]
print("idle after (us per step):"); [print("   %-42s %7.1f" % (k, v / 1e3 / nstep)) for k, v in gap_after.most_common(8)]
# one step's timeline
j = [i for i, r in enumerate(seg) if "k_step_md_b" in r[2]][3]
k = [i for i, r in enumerate(seg) if "k_step_md_b" in r[2]][4]
b = seg[j][0]
for s, e, n in seg[j:k + 1]: print("   %9.1f -> %9.1f  %s" % ((s - b) / 1e3, (e - b) / 1e3, n[:60]))
PY
rm -rf $out/tr
