"""env-steps/s of the other BASELINE configs (parity-test cases, not the bench line): device policy, autoreset on.
Usage: python tools/bench_configs.py [c3] [c4] [c2]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphenvs_amd as ge

CONFIGS = {
    "c2": ("ShortestPath-v0", dict(n_nodes=64, n_edges=192), 65536, 200),
    "c3": ("TSP-v0", dict(n_nodes=128, n_edges=8128, parenting=1), 16384, 130),
    "c4": ("SteinerTree-v0", dict(n_nodes=256, n_edges=1024, n_dests=8), 16384, 100),
    "c4long": ("SteinerTree-v0", dict(n_nodes=256, n_edges=1024, n_dests=8), 16384, 700),
    "mis": ("MaxIndependentSet-v0", dict(n_nodes=64, n_edges=192), 65536, 130),
    "ds": ("DensestSubgraph-v0", dict(n_nodes=64, n_edges=192, parenting=1), 65536, 100),
    "mc": ("MulticastRouting-v0", dict(n_nodes=64, n_edges=192, n_dests=5), 65536, 150),
    "dc": ("DistributionCenter-v0", dict(n_nodes=64, n_edges=192), 65536, 100),
    "ppd": ("PerishableProductDelivery-v0", dict(n_nodes=64, n_edges=192, parenting=1), 16384, 300),
}
def bench_c5(slots_per_id=16384, n_sizes=481, K=100, ids=(0, 1, 2)):
    """BASELINE config 5: mixed {ShortestPath, MaxIndependentSet, DensestSubgraph}, n ~ U{32..512} (every size), m = 3n; one
    multi-class engine (one launch sequence) per env id"""
    import numpy as np
    rng = np.random.default_rng(0)
    members = []
    for k, (eid, extra) in enumerate((("ShortestPath-v0", {}), ("MaxIndependentSet-v0", {}), ("DensestSubgraph-v0", dict(parenting=1)))):
        ns = rng.integers(32, 513, slots_per_id)
        if k not in ids: continue  # (c5sp / c5mis / c5ds: one member alone, same size draw)
        sizes = [(int((ns == n).sum()), int(n), 3 * int(n)) for n in np.unique(ns)]
        if os.environ.get("GE_C5_PREFETCH"): extra = dict(extra, prefetch=int(os.environ["GE_C5_PREFETCH"]))  # (refill period of the spare images)
        members.append(ge.RaggedVectorEnv(eid, sizes, **extra))
    mixed = ge.MixedVectorEnv(members)
    t0 = time.perf_counter(); mixed.reset(seed=0); torch.cuda.synchronize(); t_reset = time.perf_counter() - t0
    mixed.random_rollout(10, policy_seed=1); torch.cuda.synchronize()
    ep0 = sum(int(m.g["episode"].sum()) for m in members)
    t0 = time.perf_counter(); mixed.random_rollout(K, policy_seed=1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    B = mixed.num_envs
    print(json.dumps(dict(config="c5", envs=B, size_classes=[len(m.classes) for m in members], steps=K, env_steps_per_s=B * K / dt,
                          ms_per_vector_step=dt * 1e3 / K, episodes=sum(int(m.g["episode"].sum()) for m in members) - ep0,
                          full_reset_ms=t_reset * 1e3, launch_sequences_per_step=len(members))), flush=True)
    mixed.close()


for name in (sys.argv[1:] or ["c3", "c4"]):
    if name in ("c5", "c5sp", "c5mis", "c5ds"):
        bench_c5(ids={"c5": (0, 1, 2), "c5sp": (0,), "c5mis": (1,), "c5ds": (2,)}[name])
        continue
    name, _, pf = name.partition("@")  # name@period: with episode prefetch (spares), refill every `period` steps
    name, _, sh = name.partition("/")  # name/S: as S shards (graphenvs_amd.sharded)
    env_id, kw, B, K = CONFIGS[name]
    env = ge.make_vec(env_id, B, shards=(int(sh) if sh else 1), prefetch=(int(pf) if pf else None), **kw)
    eps = (lambda: int(env.gather("episode").sum())) if hasattr(env, "gather") else (lambda: int(env.t["episode"].sum()))
    # the first launch of a kernel instantiation pays its code-object load (round 1's 106 ms "full reset" of the first config in the
    # list was that): one untimed reset first, the timed one after it
    env.reset(seed=1); torch.cuda.synchronize()
    t0 = time.perf_counter(); env.reset(seed=0); torch.cuda.synchronize(); t_reset = time.perf_counter() - t0
    env.random_rollout(10, policy_seed=1); torch.cuda.synchronize()
    ep0 = eps()
    t0 = time.perf_counter(); env.random_rollout(K, policy_seed=1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps(dict(config=name, shards=getattr(env, "shards", 1), prefetch=env.prefetch, env=env_id, kwargs=kw, envs=B, steps=K, env_steps_per_s=B * K / dt, ms_per_vector_step=dt * 1e3 / K,
                          episodes=eps() - ep0, full_reset_ms=t_reset * 1e3)), flush=True)
    env.close(); del env; torch.cuda.empty_cache()
