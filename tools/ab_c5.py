"""A/B of library builds on BASELINE config 5 (same box, same process): python tools/ab_c5.py name=path.so ... ("cur" = the built library)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from graphenvs_amd import _lib
import graphenvs_amd as ge
per_id, steps = int(os.environ.get("GE_PER_ID", 16384)), int(os.environ.get("GE_STEPS", 100))
for rep in range(int(os.environ.get("GE_REPS", 2))):
    for spec in sys.argv[1:]:
        name, _, path = spec.partition("=")
        L = _lib.load() if path == "cur" else _lib.bind(C.CDLL(os.path.join(ROOT, path)))
        rng = np.random.default_rng(0)
        members = []
        for env_id, extra in (("ShortestPath-v0", {}), ("MaxIndependentSet-v0", {}), ("DensestSubgraph-v0", dict(parenting=1))):
            ns = rng.integers(32, 513, per_id)
            sizes = [(int((ns == n).sum()), int(n), 3 * int(n)) for n in np.unique(ns)]
            members.append(ge.RaggedVectorEnv(env_id, sizes, device="cuda", _library=L, prefetch=int(os.environ.get("GE_PREFETCH", 4)), **extra))
        mixed = ge.MixedVectorEnv(members)
        mixed.reset(seed=0); mixed.random_rollout(60, policy_seed=1); torch.cuda.synchronize()
        t0 = time.perf_counter(); mixed.random_rollout(steps, policy_seed=1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name:10s} {mixed.num_envs * steps / dt / 1e6:8.2f} M env-steps/s  {dt / steps * 1e3:.3f} ms per step", flush=True)
        mixed.close(); del mixed, members
