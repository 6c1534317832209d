"""make_vec(..., shards=S) on the uniform BASELINE configs (same box, same process): python tools/shards_configs.py c2 c3 c4 [S ...via GE_SHARDS=1,2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphenvs_amd as ge
CFG = {"c2": ("ShortestPath-v0", dict(n_nodes=64, n_edges=192), 65536, 120, 300),
       "c3": ("TSP-v0", dict(n_nodes=128, n_edges=8128, parenting=1), 16384, 256, 256),
       "c4": ("SteinerTree-v0", dict(n_nodes=256, n_edges=1024, n_dests=8), 16384, 512, 400)}
for name in sys.argv[1:] or ["c2"]:
    env_id, kw, B, settle, steps = CFG[name]
    for S in [int(x) for x in os.environ.get("GE_SHARDS", "1,2").split(",")]:
        for rep in range(2):
            env = ge.make_vec(env_id, B, shards=S, **kw)
            env.reset(seed=0); env.random_rollout(settle, policy_seed=1); torch.cuda.synchronize()
            t0 = time.perf_counter(); env.random_rollout(steps, policy_seed=1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"{name} shards={S}: {B * steps / dt / 1e6:8.2f} M env-steps/s  {dt / steps * 1e6:.1f} us per step", flush=True)
            env.close(); del env
