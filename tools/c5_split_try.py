import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphenvs_amd as ge
per_id, steps = 16384, 100
def build(split):
    rng = np.random.default_rng(0)
    members = []
    for env_id, extra in (("ShortestPath-v0", {}), ("MaxIndependentSet-v0", {}), ("DensestSubgraph-v0", dict(parenting=1))):
        ns = rng.integers(32, 513, per_id)
        parts = split.get(env_id, 1)
        off = 0
        for k in range(parts):
            sub = ns[k::parts]  # every parts-th slot: each shard sees every size
            sizes = [(int((sub == n).sum()), int(n), 3 * int(n)) for n in np.unique(sub)]
            members.append(ge.RaggedVectorEnv(env_id, sizes, device="cuda", env_index_base=off, seed_stride=per_id, **extra))
            off += len(sub)
    return ge.MixedVectorEnv(members)
for name, split in (("as shipped", {}), ("SP x2", {"ShortestPath-v0": 2}), ("SP x2 DS x2", {"ShortestPath-v0": 2, "DensestSubgraph-v0": 2})):
    for rep in range(2):
        mixed = build(split)
        mixed.reset(seed=0); mixed.random_rollout(60, policy_seed=1); torch.cuda.synchronize()
        t0 = time.perf_counter(); mixed.random_rollout(steps, policy_seed=1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"c5 {name:12s} ({len(mixed.members)} engines, {mixed.concurrent_streams} streams): {mixed.num_envs * steps / dt / 1e6:6.2f} M env-steps/s  {dt / steps * 1e3:.3f} ms per step", flush=True)
        mixed.close(); del mixed
