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
subprocess.check_call([_lib.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                       "-DGE_STAMPS", "-I" + _lib.CSRC, os.path.join(_lib.CSRC, "ge_api.hip"), "-o", out])
L = _lib.bind(C.CDLL(out))
L.ge_debug_read_stamps.argtypes = [C.c_void_p]
import graphenvs_amd as ge  # noqa: E402

names = ["seed_py", "topology", "csr", "seed_np", "weights", "terminals", "baseline", "sort+init", "bfs(bc,clos)+clust", "pagerank", "writeout"]
cfgs = [("ShortestPath-v0", dict(n_nodes=64, n_edges=192), [1, 4096, 65536])]
if len(sys.argv) > 1 and sys.argv[1] == "all":
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
        ts = [buf[k] for k in range(12)]
        print(f"{env_id} {kw} B={B}: total {(ts[11]-ts[0])/100:.1f} us")
        for k, nm in enumerate(names):
            print(f"    {nm:22s} {(ts[k+1]-ts[k])/100:9.1f} us")
