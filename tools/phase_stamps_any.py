"""Diagnostic: tools/phase_stamps.py for any geometry: GE_ENV / GE_N / GE_M / GE_B (slot 0's regeneration phases, -DGE_STAMPS build)."""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphenvs_amd import _lib
out = os.path.join(ROOT, "gpurun_out", "libgraphenvs_hip_stamps.so")
subprocess.check_call(_lib.compile_command(out, extra=["-DGE_STAMPS"] + [a for a in sys.argv[1:] if a.startswith("-D")]))
L = _lib.bind(C.CDLL(out)); L.ge_debug_read_stamps.argtypes = [C.c_void_p]
import graphenvs_amd as ge
env_id, n, m = os.environ.get("GE_ENV", "ShortestPath-v0"), int(os.environ.get("GE_N", 512)), int(os.environ.get("GE_M", 1536))
names = {0: "A seed_py", 1: "A topology", 2: "A csr", 3: "A join + weights", 5: "A terminals", 6: "A baseline", 9: "A writeout", 20: "A1 state load", 21: "A1 draws", 22: "A1 terminals"}
for B in [int(b) for b in os.environ.get("GE_BS", "1,512").split(",")]:
    env = ge.VectorGraphEnv(env_id, B, n, m, device="cuda", _library=L, prefetch=0, **json.loads(os.environ.get("GE_KW", "{}")))
    for rep in range(2):
        env.reset(seed=rep); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)(); L.ge_debug_read_stamps(buf); ts = [buf[k] for k in range(32)]
    print(f"{env_id} n={n} m={m} B={B}:")
    for k, nm in names.items():
        nxt = {6: 9, 3: 5}.get(k, k + 1)
        if ts[k] and ts[nxt]: print(f"    {nm:22s} {(ts[nxt]-ts[k])/100:9.1f} us")
    if ts[9] and ts[7] and ts[8] and ts[7] > ts[9]: print(f"    writeout: before the range search {(ts[7]-ts[9])/100:.1f} us, range search {(ts[8]-ts[7])/100:.1f} us, rest {(ts[10]-ts[8])/100:.1f} us")
    elif ts[3] and ts[7] and ts[8]: print(f"    late draws: wanted cells {(ts[7]-ts[3])/100:.1f} us, stream {(ts[8]-ts[7])/100:.1f} us, rest {(ts[5]-ts[8])/100:.1f} us")
    if ts[0] and ts[10]: print(f"    whole slot             {(ts[10]-ts[0])/100:9.1f} us")
    env.close()
