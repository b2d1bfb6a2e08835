#!/bin/bash
# usage (GPU box): bash scripts/sq_counters.sh <tag> [R]   -- SQ counters of the batched kernels (one pass per counter group)
tag=$1; R=${2:-256}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/sq_$tag; rm -rf $out; mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" "SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -- python3 scripts/batch_scaling.py --nsteps 60 $R > $out/g$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
out = "$out"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/*/*counter_collection.csv"):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "_b<" not in n and "_b(" not in n: continue
        per[(n, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (n, d), cs in per.items():
        for c, v in cs.items(): agg[n[:48]][c].append(v)
for n, cs in sorted(agg.items()):
    print(n)
    for c, vals in sorted(cs.items()):
        vals = vals[len(vals) // 2:]
        print("   %-22s %14.0f  (mean of %d launches)" % (c, sum(vals) / len(vals), len(vals)))
PY

rm -rf gpurun_out/sq_$tag/g*/
