"""Profiling target: the fused ShortestPath step kernel alone (GE_B env slots, bursts of 5 launches after a reset).
Run under rocprofv3 (--kernel-trace --stats, or one --pmc set per pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphenvs_amd as ge
B = int(os.environ.get("GE_B", 65536))
env = ge.make_vec("ShortestPath-v0", B, n_nodes=64, n_edges=192)
for rep in range(int(os.environ.get("GE_REPS", 6))):
    env.reset(seed=rep); torch.cuda.synchronize()
    env.timed_step_burst_raw_ms(5, policy_seed=2)
torch.cuda.synchronize()
