"""long sharded rollouts (stability): the headline as three shards for 30 000 steps, config 4 as two for 6 000, then the invariants"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphenvs_amd as ge
for env_id, kw, B, S, K in (("ShortestPath-v0", dict(n_nodes=64, n_edges=192), 65536, 3, 30000), ("SteinerTree-v0", dict(n_nodes=256, n_edges=1024, n_dests=8), 16384, 2, 6000)):
    env = ge.make_vec(env_id, B, shards=S, **kw)
    env.reset(seed=0)
    t0 = time.perf_counter()
    for k in range(0, K, 1000):
        env.random_rollout(1000, policy_seed=1)
        torch.cuda.synchronize()
        print(f"{env_id}: {k + 1000} steps, {(k + 1000) * B / (time.perf_counter() - t0) / 1e6:.1f} M env-steps/s so far", flush=True)
    assert int(env.gather("tstep").sum()) == B * K
    env.check_device_errors()
    print(f"{env_id}: ok, {int(env.gather('episode').sum())} episodes", flush=True)
    env.close(); del env
