"""Table of tools/f64_phase.py runs: per variant the per-slot medians of the counters of ge_k_features64 and its median duration.
usage: python tools/f64_phase_summary.py <dir holding one sub-directory per variant> [slots per launch]"""
import csv, glob, json, os, statistics, sys
root, B = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2688
rows = {}
for vdir in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(vdir): continue
    acc, dur = {}, []
    for f in glob.glob(os.path.join(vdir, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if "ge_k_features64" not in r["Kernel_Name"]: continue
            key = (r["Counter_Name"], r["Dispatch_Id"]); per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
        for (c, _), v in per.items(): acc.setdefault(c, []).append(v)
    for f in glob.glob(os.path.join(vdir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "ge_k_features64" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows[os.path.basename(vdir)] = dict({c: statistics.median(v) / B for c, v in acc.items()}, median_us=statistics.median(dur) if dur else None)
print(json.dumps(rows, indent=1))
