"""Reduces the rocprofv3 --pmc passes of scripts/pmc_nb.sh to one JSON: per-launch means of the batched nonbonded kernel's counters
(the force launches of the stepping loop; bench.py runs with --no-kernel-timing so that the trace holds nothing else), with the gfx950 corrections of
MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read: doubled; SQ_*_CYCLES / SQ_ACTIVE_* are quad-cycles)."""
import csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
sys.path.insert(0, ROOT)
def source_sha():
    from blues_amd import build
    return build.source_sha()   # every source of the library + the compiler flags
res = {}; kernel = None; durations = {}
for f in sorted(glob.glob(out + "/g*/*/*counter_collection.csv")):
    per = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_nonbonded" not in n or "_b<" not in n or "true" in n.split("<")[1][:6]:
            continue
        kernel = n
        per.setdefault(int(r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(per)[3:]   # every force launch of the stepping loop but the first few (bench.py --no-kernel-timing: no stand-alone launches in the trace)
    for c in set(k for d in per.values() for k in d):
        vals = [per[i][c] for i in ids if c in per[i]]
        if vals:
            res[c] = sum(vals) / len(vals)
d = {"_how": "scripts/pmc_nb.sh: rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-single --no-kernel-timing --nsteps-nc 200 " + " ".join(sys.argv[2:]) + "; mean over the force launches of the stepping loop",
     "kernel": kernel, "source_sha": source_sha(), "bench_args": sys.argv[2:], "counters_per_launch": res}
if "FETCH_SIZE" in res or "WRITE_SIZE" in res:
    # FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3's derived metric
    d["traffic_bytes_per_launch"] = 1024.0 * (2.0 * res.get("FETCH_SIZE", 0.0) + res.get("WRITE_SIZE", 0.0))
    d["traffic_note"] = "1024 x (2 x FETCH_SIZE + WRITE_SIZE): FETCH_SIZE doubled per the gfx950 correction for wide coalesced reads (an upper bound for the gathered part)"
args = sys.argv[2:]
wl = args[args.index("--workload") + 1] if "--workload" in args else "rotmove"
R = int(args[args.index("--replicas") + 1]) if "--replicas" in args else 2048   # (bench.py's defaults: 2048 chains ...
G = int(args[args.index("--groups") + 1]) if "--groups" in args else 2            # ... in two batches: 1024 chains per launch)
R = R // max(1, min(G, R))
table = {"%s_R%d" % (wl, R): d}
json.dump(table, open(out + "/pmc_nonbonded.json", "w"), indent=1)
print(json.dumps(table, indent=1))
