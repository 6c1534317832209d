"""Diagnostic: duration of the reset kernels against the number of slots reset at once (run under rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphenvs_amd as ge
for B in [64, 256, 1024, 2048, 2560, 2816, 3072, 4096, 8192]:
    env = ge.make_vec("ShortestPath-v0", B, n_nodes=64, n_edges=192)
    for rep in range(3):
        env.reset(seed=rep * 100000)
    torch.cuda.synchronize()
    env.close()
