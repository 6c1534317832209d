"""Diagnostic: full reset of a uniform engine against a multi-class engine with ONE class of the same geometry (what the class lookup costs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphenvs_amd as ge
for env_id, n, m, B in (("ShortestPath-v0", 256, 768, 2048), ("DensestSubgraph-v0", 256, 768, 2048), ("ShortestPath-v0", 64, 192, 8192)):
    kw = dict(parenting=1) if env_id.startswith("Densest") else {}
    for kind in ("uniform", "ragged"):
        env = ge.make_vec(env_id, B, n_nodes=n, n_edges=m, prefetch=0, **kw) if kind == "uniform" else ge.RaggedVectorEnv(env_id, [(B, n, m)], prefetch=0, **kw)
        env.reset(seed=0); torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); env.reset(seed=1000 * rep); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        print(f"{env_id} n={n} m={m} B={B} {kind:8s}: full reset median {sorted(ts)[2]:.2f} ms", flush=True)
        env.close()
