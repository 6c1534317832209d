"""A/B of library builds on the headline config (same box, same process; GE_SHARDS=3: as three shards): python tools/ab_c2.py name=path.so ... ("cur" = the built library)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphenvs_amd import _lib
import graphenvs_amd as ge
B, steps, S = int(os.environ.get("GE_B", 65536)), int(os.environ.get("GE_STEPS", 300)), int(os.environ.get("GE_SHARDS", 1))
for rep in range(int(os.environ.get("GE_REPS", 2))):
    for spec in sys.argv[1:]:
        name, _, path = spec.partition("=")
        L = _lib.load() if path == "cur" else _lib.bind(C.CDLL(os.path.join(ROOT, path)))
        env = ge.make_vec("ShortestPath-v0", B, shards=S, n_nodes=64, n_edges=192, device="cuda", _library=L, prefetch=0)
        env.reset(seed=0); env.random_rollout(120, policy_seed=1); torch.cuda.synchronize()
        t0 = time.perf_counter(); env.random_rollout(steps, policy_seed=1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name:10s} {B * steps / dt / 1e6:8.2f} M env-steps/s  {dt / steps * 1e6:.1f} us per step", flush=True)
        env.close(); del env
