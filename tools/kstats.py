"""median / p90 duration per kernel from a rocprofv3 kernel_trace.csv (queue-mode launches dominate the median)"""
import csv, sys, glob, statistics, collections
path = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if "ge_k" in r["Kernel_Name"]:
        d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v.sort()
    print(f"{k:40s} n={len(v):4d} median {statistics.median(v):9.1f} us  p90 {v[int(len(v)*0.9)]:9.1f} us  min {v[0]:8.1f}")
