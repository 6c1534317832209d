"""Diagnostic: how many slots finish per vector step (BASELINE config 4 shape by default), step by step, after bench.py's settle
steps: the queue length of every in-place regeneration launch.  usage: python tools/resets_per_step.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphenvs_amd as ge
K = int(sys.argv[1]) if len(sys.argv) > 1 else 600
env = ge.make_vec("SteinerTree-v0", 16384, n_nodes=256, n_edges=1024, n_dests=8, prefetch=0)
env.reset(seed=0); env.random_rollout(512 + 5, policy_seed=1)
counts = []
for k in range(K):
    env.random_rollout(1, policy_seed=1)
    counts.append(int(env.t["terminated"].sum()))
c = np.array(counts)
print("steps", K, "mean", c.mean(), "median", np.median(c), "p90", np.percentile(c, 90), "max", c.max(), "min", c.min())
print("share of steps with more than 57 finished slots (one round of the 512 resident feature workgroups at 9 per slot):", float((c > 57).mean()))
print("share with more than 114 (two rounds):", float((c > 114).mean()), " more than 171:", float((c > 171).mean()))
print("histogram (bin 20):", np.bincount(c // 20).tolist())
