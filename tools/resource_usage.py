"""Per-kernel register / scratch / occupancy table of the HIP library, from hipcc's kernel-resource-usage remarks
(cross-compiles, no GPU needed).  usage: python tools/resource_usage.py [extra hipcc flags ...] [--filter substr]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphenvs_amd import _lib
args = sys.argv[1:]
flt = ""
if "--filter" in args:
    i = args.index("--filter"); flt = args[i + 1]; del args[i:i + 2]
cmd = _lib.compile_command("/tmp/ge_resource_usage.so") + ["-Rpass-analysis=kernel-resource-usage"] + args
err = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: .*?(Function Name|Name): (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(2)], stdout=subprocess.PIPE, text=True).stdout.strip().split("(")[0]}
        rows.append(cur); continue
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).replace(" ", "_").split("_[")[0]] = int(m.group(2))
print(f"{'kernel':44s} {'SGPR':>5s} {'VGPR':>5s} {'AGPR':>5s} {'scratch B/lane':>15s} {'waves/SIMD':>11s} {'VGPR spills':>12s}")
for r in rows:
    if flt and flt not in r["name"]: continue
    print(f"{r['name'][:44]:44s} {r.get('TotalSGPRs', 0):5d} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('ScratchSize', 0):15d} {r.get('Occupancy', 0):11d} {r.get('VGPRs_Spill', 0):12d}")
