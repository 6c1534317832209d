"""the per-call API (sample_random_actions + step, host in the loop) on the headline batch as 1 / 2 / 3 shards (same box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphenvs_amd as ge
B, K = 65536, 200
for S in (1, 2, 3):
    env = ge.make_vec("ShortestPath-v0", B, shards=S, n_nodes=64, n_edges=192)
    env.reset(seed=0); env.random_rollout(120, policy_seed=1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        a = env.sample_random_actions(policy_seed=1)
        env.step(a)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"shards={S}: sample + step through the Python API: {B * K / dt / 1e6:8.2f} M env-steps/s  {dt / K * 1e6:.1f} us per step", flush=True)
    env.close(); del env
