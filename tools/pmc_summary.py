"""Per-kernel medians of every counter found in rocprofv3 counter_collection CSVs under the given directories.
usage: python tools/pmc_summary.py <dir> [<dir> ...] [--kernel substr]"""
import csv, glob, os, statistics, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
sub = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else ""
if sub in args: args.remove(sub)
acc = {}
for d in args:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if sub and sub not in k: continue
            per.setdefault((k, r["Counter_Name"], r["Dispatch_Id"]), 0.0)
            per[(k, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (k, c, _), v in per.items(): acc.setdefault((k, c), []).append(v)
for (k, c), v in sorted(acc.items()):
    print(f"{k[:48]:48s} {c:28s} n={len(v):4d} median={statistics.median(v):16.1f}")
