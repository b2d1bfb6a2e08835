"""Splits a rocprofv3 kernel trace of `bench.py --md-steps ...` into the two legs: launches of kernels both legs share (the fragment-list
kernels) are told apart by duration (an MD-leg launch covers every atom of every chain, an NCMC-leg launch the few mobile ones).
   python scripts/dev_full_profile_split.py <kernel_trace.csv> <n_chains>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
R = int(sys.argv[2])
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0][:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-50s %8s %10s %10s %10s" % ("kernel", "calls", "median us", "mean us", "total ms"))
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    if sum(v) < 2e3: continue
    if k.startswith(("k_nonbonded_frag_b", "void k_nonbonded_frag_b", "k_frag_lists_b", "k_frag_pre_b", "void k_finalize_b", "k_bonded_entries_b", "k_gather_frag_b")) and len(v) > 10:
        cut = (v[0] * v[-1]) ** 0.5
        lo, hi = [x for x in v if x <= cut], [x for x in v if x > cut]
        for name, part in (("  small launches (NCMC leg)", lo), ("  large launches (MD leg)", hi)):
            if part: print("%-50s %8d %10.1f %10.1f %10.1f" % (k[:22] + name, len(part), part[len(part) // 2], sum(part) / len(part), sum(part) / 1e3))
    else:
        print("%-50s %8d %10.1f %10.1f %10.1f" % (k, len(v), v[len(v) // 2], sum(v) / len(v), sum(v) / 1e3))
