"""Diagnostic: build the library with -DGE_STAMPS (never shipped) and print the time slot 0 spends in
each phase of the reset kernel, alone on the chip and inside a full batch.  Run on the GPU box."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from graphenvs_amd import _lib  # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "libgraphenvs_hip_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
subprocess.check_call(_lib.compile_command(out, extra=["-DGE_STAMPS"] + extra))
L = _lib.bind(C.CDLL(out))
L.ge_debug_read_stamps.argtypes = [C.c_void_p]
import graphenvs_amd as ge  # noqa: E402

print("flags:", extra)
names = {0: "A seed_py", 1: "A topology", 2: "A csr", 3: "A seed_np", 4: "A weights", 5: "A terminals", 6: "A baseline", 9: "A writeout", 11: "B walk item: stage", 12: "B walk item: two rounds (levels, path counts, dependencies, reduction)", 14: "B walk item: write", 20: "A1 state load", 21: "A1 draws", 22: "A1 terminals"}
cfgs = [("ShortestPath-v0", dict(n_nodes=64, n_edges=192), [1, 4096, 65536])]
if "all" in sys.argv[1:]:
    cfgs += [("SteinerTree-v0", dict(n_nodes=256, n_edges=1024, n_dests=8), [1, 2048]),
             ("TSP-v0", dict(n_nodes=128, n_edges=8128, parenting=1), [1, 1024])]
for env_id, kw, Bs in cfgs:
    for B in Bs:
        env = ge.VectorGraphEnv(env_id, B, device="cuda", _library=L, **kw)
        for rep in range(2):
            env.reset(seed=rep)
            torch.cuda.synchronize()
        buf = (C.c_ulonglong * 32)()
        L.ge_debug_read_stamps(buf)
        ts = [buf[k] for k in range(32)]
        print(f"{env_id} {kw} B={B}:")
        for k, nm in names.items():
            nxt = k + 1 if k != 6 else 9
            if k == 12: nxt = 14
            if k == 14: nxt = 17
            if ts[k] and ts[nxt]:
                print(f"    {nm:22s} {(ts[nxt]-ts[k])/100:9.1f} us")
        if ts[29] and ts[26]:  # the LAST attempt of slot 0's G(n, m) rejection loop
            print(f"    A last attempt: rounds {(ts[26]-ts[29])/100:.1f} us, degree pass {(ts[27]-ts[26])/100:.1f} us, connectivity {(ts[28]-ts[27])/100:.1f} us; attempts before it took {(ts[29]-ts[1])/100:.1f} us")
        if ts[11] and ts[17] and ts[30] and ts[31] and ts[17] > ts[11]:  # shader clock during the n <= 64 feature kernel, measured inside it
            print(f"    shader clock while slot 0's feature workgroup ran: {(ts[31]-ts[30]) / ((ts[17]-ts[11]) * 10.0):.3f} GHz  (s_memtime ticks / s_memrealtime 100 MHz ticks)")
