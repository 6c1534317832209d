"""GPU busy time (union of kernel intervals) against wall time over the middle half of a rocprofv3 kernel trace, per queue and overall:
python tools/busy_union.py <trace dir>"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(path))]
rows.sort()
steps = [r[0] for r in rows if "ge_k_step" in r[3]]  # the window: between the 40 % and 95 % quantiles of the step launches (steady loop)
lo, hi = steps[int(len(steps) * 0.40)], steps[int(len(steps) * 0.95)]
rows = [r for r in rows if lo <= r[0] < hi and "ge_k" in r[3]]
nsteps = sum(1 for r in rows if "ge_k_step" in r[3])
def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot
wall = hi - lo
print(f"window {wall / 1e6:.2f} ms, {nsteps} step launches, {len(rows)} kernels; busy (union over all queues) {union([(s, e) for s, e, _, _ in rows]) / wall:.3f} of the wall time")
for q in sorted({r[2] for r in rows}):
    iv = [(s, e) for s, e, qq, _ in rows if qq == q]
    print(f"  queue {q}: {len(iv)} kernels, busy {union(iv) / wall:.3f}")
by = {}
for s, e, _, k in rows: by[k] = by.get(k, 0) + (e - s)
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:10]: print(f"  {k:44s} {v / wall:.3f} of the wall time (summed over queues)")
