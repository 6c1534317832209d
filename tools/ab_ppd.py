"""A/B of library builds on PerishableProductDelivery n=64 with its default prefetch (same box, same process): python tools/ab_ppd.py name=path.so ..."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphenvs_amd import _lib
import graphenvs_amd as ge
B, steps = 16384, 300
for rep in range(int(os.environ.get("GE_REPS", 2))):
    for spec in sys.argv[1:]:
        name, _, path = spec.partition("=")
        L = _lib.load() if path == "cur" else _lib.bind(C.CDLL(os.path.join(ROOT, path)))
        env = ge.make_vec("PerishableProductDelivery-v0", B, n_nodes=64, n_edges=192, parenting=1, device="cuda", _library=L,
                          prefetch=ge.VectorGraphEnv.default_prefetch("PerishableProductDelivery-v0", 64, B))
        env.reset(seed=0); env.random_rollout(100, policy_seed=1); torch.cuda.synchronize()
        t0 = time.perf_counter(); env.random_rollout(steps, policy_seed=1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name:10s} {B * steps / dt / 1e6:8.2f} M env-steps/s  {dt / steps * 1e6:.1f} us per step (prefetch {env.prefetch})", flush=True)
        env.close(); del env
