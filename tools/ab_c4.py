"""A/B of library builds on BASELINE config 4 (same box, same process): python tools/ab_c4.py name=path.so ... ("cur" = the built library)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphenvs_amd import _lib
import graphenvs_amd as ge
B, steps, settle = int(os.environ.get("GE_B", 16384)), int(os.environ.get("GE_STEPS", 400)), int(os.environ.get("GE_SETTLE", 512))
for rep in range(int(os.environ.get("GE_REPS", 2))):
    for spec in sys.argv[1:]:
        name, _, path = spec.partition("=")
        L = _lib.load() if path == "cur" else _lib.bind(C.CDLL(os.path.join(ROOT, path)))
        env = ge.make_vec("SteinerTree-v0", B, n_nodes=256, n_edges=1024, n_dests=8, device="cuda", _library=L, prefetch=ge.VectorGraphEnv.default_prefetch("SteinerTree-v0", 256, B))
        env.reset(seed=0); env.random_rollout(settle, policy_seed=1); torch.cuda.synchronize()
        t0 = time.perf_counter(); env.random_rollout(steps, policy_seed=1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name:10s} {B * steps / dt / 1e6:8.2f} M env-steps/s  {dt / steps * 1e3:.4f} ms per step", flush=True)
        env.close(); del env
