cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3f; rm -rf $out; mkdir -p $out
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TA_[A-Z_0-9]*" | sort -u > $out/avail.txt
i=0
for grp in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -- python3 scripts/dev_k1exp.py 256 > $out/g$i.log 2>&1
  tail -2 $out/g$i.log
done
python3 - <<PY
import csv, glob
out = "$out"
res = {}
for g in (1,2,3):
    fs = glob.glob(out + "/g%d/*/*counter_collection.csv" % g)
    if not fs: print("no csv for group", g); continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "k_nonbonded_atom_b<false>" in r["Kernel_Name"] or ("nonbonded_atom_b" in r["Kernel_Name"] and "false" in r["Kernel_Name"])]
    by = {}
    for r in rows: by.setdefault((r["Counter_Name"], int(r["Dispatch_Id"])), 0.0); by[(r["Counter_Name"], int(r["Dispatch_Id"]))] += float(r["Counter_Value"])
    names = sorted({k[0] for k in by})
    for n in names:
        ids = sorted(k[1] for k in by if k[0] == n)
        # launches: 1 (lists) + 3 warm + 30 pruned + 3 warm + 30 full; take pruned = ids[4:34], full = ids[-30:]
        pr = [by[(n, i)] for i in ids[4:34]]; fu = [by[(n, i)] for i in ids[-30:]]
        print("%-28s pruned %.4g   full %.4g   (%d launches)" % (n, sum(pr)/max(1,len(pr)), sum(fu)/max(1,len(fu)), len(ids)))
PY
rm -rf $out/g*/
