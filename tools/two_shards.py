"""Experiment: the headline batch as S independent engines (shards of B / S slots) in ONE process, each rolling out on its own HIP stream --
against one engine of B slots.  python tools/two_shards.py [S ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphenvs_amd as ge
B, K, chunk = int(os.environ.get("GE_B", 65536)), int(os.environ.get("GE_STEPS", 300)), int(os.environ.get("GE_CHUNK", 0))
for S in [int(a) for a in sys.argv[1:]] or [1, 2]:
    envs = [ge.make_vec("ShortestPath-v0", B // S, n_nodes=64, n_edges=192, env_index_base=k * (B // S), seed_stride=B) for k in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    for e in envs: e.reset(seed=0)
    torch.cuda.synchronize()
    def run(steps):
        if chunk:  # alternate the shards every `chunk` steps (the host enqueues far ahead of the GPU either way)
            for s0 in range(0, steps, chunk):
                for e, st in zip(envs, streams):
                    with torch.cuda.stream(st): e.random_rollout(min(chunk, steps - s0), policy_seed=1)
        else:
            for e, st in zip(envs, streams):
                with torch.cuda.stream(st): e.random_rollout(steps, policy_seed=1)
        torch.cuda.synchronize()
    run(120)
    t0 = time.perf_counter(); run(K); dt = time.perf_counter() - t0
    print(f"{S} shard(s) of {B // S}: {B * K / dt / 1e6:8.2f} M env-steps/s  {dt / K * 1e6:.1f} us per step of {B}", flush=True)
    for e in envs: e.close()
    del envs
