"""Diagnostic: time the fused ShortestPath step kernel with parts compiled out (-DGE_ABL=bits; results are
wrong by construction, only the time matters).  bits: 1 no x flag store, 2 no bool-mask bytes, 4 no bit-row /
row_ptr gathers, 8 no weight-byte gather.  Run on the GPU box."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from graphenvs_amd import _lib  # noqa: E402
import graphenvs_amd as ge  # noqa: E402

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for bits in [0, 1, 2, 4, 7]:
    out = os.path.join(ROOT, "gpurun_out", f"libge_abl{bits}.so")
    subprocess.check_call([_lib.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           f"-DGE_ABL={bits}", "-I" + _lib.CSRC, os.path.join(_lib.CSRC, "ge_api.hip"), "-o", out])
    L = _lib.bind(C.CDLL(out))
    env = ge.VectorGraphEnv("ShortestPath-v0", 65536, 64, 192, device="cuda", _library=L, autoreset=False)
    env.reset(seed=0)
    torch.cuda.synchronize()
    best = []
    for rep in range(3):
        env.reset(seed=rep)
        tm = env.timed_rollout(8, policy_seed=1)
        best.append(tm["step_ms"] / 8 * 1e3)
    print(f"GE_ABL={bits:2d}: fused step kernel {min(best):7.2f} us (event-bracketed, first 8 steps after reset)", flush=True)
    os.remove(out)
