"""idle time between consecutive kernels of the rollout loop, from a rocprofv3 kernel_trace.csv (queue-mode launches only)"""
import csv, glob, statistics, sys
path = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(path)) if "ge_k" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
main = [r for r in rows if "seed" not in name(r)]
gaps = {}
for a, b in zip(main, main[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    gaps.setdefault((name(a), name(b)), []).append(g)
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) > 20:
        print(f"{k[0]:28s} -> {k[1]:28s} n={len(v):4d} median gap {statistics.median(v):7.2f} us  p90 {sorted(v)[int(len(v)*0.9)]:7.2f}")
seed = [r for r in rows if "seed" in name(r)]
print("seed kernels", len(seed))
# where the pre-seeding kernel (side stream) ends relative to the feature kernels of the same vector step
import bisect
f64 = [r for r in main if name(r) == "ge_k_features"]
ends = sorted(int(r["End_Timestamp"]) for r in seed)
starts_seed = sorted(int(r["Start_Timestamp"]) for r in seed)
resets = [r for r in main if name(r).startswith("ge_k_reset")]
late, startlag = [], []
for r in f64:
    fe = int(r["End_Timestamp"])
    i = bisect.bisect_left(ends, fe - 400000)
    cands = [e for e in ends[i:i + 3] if abs(e - fe) < 300000]
    if cands: late.append((min(cands, key=lambda e: abs(e - fe)) - fe) / 1e3)
for r in resets:
    re_ = int(r["End_Timestamp"])
    i = bisect.bisect_left(starts_seed, re_)
    if i < len(starts_seed) and starts_seed[i] - re_ < 300000: startlag.append((starts_seed[i] - re_) / 1e3)
if late: print(f"seed end - feature end: median {statistics.median(late):.1f} us  p90 {sorted(late)[int(len(late)*0.9)]:.1f}")
if startlag: print(f"seed start - reset end: median {statistics.median(startlag):.1f} us")
