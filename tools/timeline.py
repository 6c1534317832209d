"""start/end of every kernel of a few consecutive vector steps (relative microseconds), from a rocprofv3 kernel_trace.csv"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(path)) if "ge_k" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mid = len(rows) // 2
t0 = int(rows[mid]["Start_Timestamp"])
for r in rows[mid:mid + 40]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{r['Kernel_Name'].split('(')[0].replace('void ', ''):28s} q{r.get('Queue_Id', '?'):>3s} {s:9.1f} -> {e:9.1f}  ({e - s:7.1f} us)")
