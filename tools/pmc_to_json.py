"""FETCH_SIZE / WRITE_SIZE per launch (medians) from two rocprofv3 --pmc passes -> profiles/pmc_step_kernel.json
usage: python tools/pmc_to_json.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, os, statistics, sys
def medians(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Dispatch_Id"])
            per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
        for (k, _), v in per.items(): out.setdefault(k, []).append(v)
    return {k: (len(v), statistics.median(v)) for k, v in out.items()}
fetch, write = medians(sys.argv[1], "FETCH_SIZE"), medians(sys.argv[2], "WRITE_SIZE")
step = next(k for k in fetch if "step_path64" in k)
f_kb, w_kb = fetch[step][1], write[step][1]
doc = {
    "kernel": step,
    "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace (MI355X_MICROARCH.md, HBM section); per-launch medians over bench.py --steps 30",
    "fetch_size_kb_raw": f_kb, "write_size_kb": w_kb,
    "fetch_correction": "x2 (gfx950 FETCH_SIZE reports half of a wide coalesced stream; our 4-16 B/lane accesses are outside that calibration, so the corrected figure is an upper bound)",
    "traffic_bytes_per_launch": int((2 * f_kb + w_kb) * 1024),
    "traffic_bytes_per_launch_uncorrected": int((f_kb + w_kb) * 1024),
    "algorithmic_bytes_per_launch": 65536 * 200,
    "all_kernels": {k: {"launches": fetch[k][0], "FETCH_SIZE_KB_median": fetch[k][1], "WRITE_SIZE_KB_median": write.get(k, (0, None))[1]} for k in fetch},
}
json.dump(doc, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("kernel", "fetch_size_kb_raw", "write_size_kb", "traffic_bytes_per_launch")}))
