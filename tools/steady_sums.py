"""Per-kernel time of the STEADY part of a rocprofv3 kernel trace (from the first step-kernel launch behind the last ge_k_seed,
i.e. behind the full resets): sum, launches, share of the window, and the sum per step-kernel launch.  usage: steady_sums.py <rocprof output dir>"""
import csv, sys, glob, collections
path = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(path))]
rows.sort()
cut = max([i for i, r in enumerate(rows) if r[2].startswith("ge_k_seed")] or [0])
cut = min(i for i, r in enumerate(rows) if i > cut and "ge_k_step" in r[2])
rows = rows[cut:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
d = collections.defaultdict(lambda: [0, 0])
for a, b, k in rows:
    d[k][0] += b - a; d[k][1] += 1
steps = max(1, max(v[1] for k, v in d.items() if "ge_k_step" in k))
busy = sum(v[0] for v in d.values())
print(f"window {(t1 - t0) / 1e6:.2f} ms, {steps} step launches, kernels busy {busy / 1e6:.2f} ms ({busy / (t1 - t0):.2f} of the window)")
for k, v in sorted(d.items(), key=lambda kv: -kv[1][0]):
    if v[0] / busy < 0.003: continue
    print(f"{k[:60]:60s} {v[0] / 1e3:10.1f} us  n={v[1]:5d}  {v[0] / busy:6.3f}  per step {v[0] / 1e3 / steps:8.1f} us")
