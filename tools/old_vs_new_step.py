"""Diagnostic: the headline step kernel of round 2's library (tools/micro/libold_r02.so, built from commit e40820e) against the current
one on the same box: bursts of 5 launches after a reset, HIP events (what round 2's bench line quoted)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphenvs_amd import _lib
import graphenvs_amd as ge
for name, path in (("round 2", os.path.join(ROOT, "tools", "micro", "libold_r02.so")), ("current", _lib.LIB_PATH), ("round 2", os.path.join(ROOT, "tools", "micro", "libold_r02.so")), ("current", _lib.LIB_PATH)):
    L = _lib.bind(C.CDLL(path))
    for B in (65536, 1 << 20):
        env = ge.VectorGraphEnv("ShortestPath-v0", B, 64, 192, device="cuda", _library=L, prefetch=0)
        env.reset(seed=0); env.random_rollout(5, 1); torch.cuda.synchronize()
        empty = sorted(env.timed_step_burst_raw_ms(0) for _ in range(9))[4]
        res = []
        for rep in range(10):
            env.reset(seed=rep); torch.cuda.synchronize()
            res.append((env.timed_step_burst_raw_ms(5, policy_seed=2) - empty) * 1e3 / 5)
        res.sort()
        print(f"{name:8s} B={B}: step kernel median {res[5]:.2f} us  min {res[0]:.2f} us", flush=True)
        env.close()
