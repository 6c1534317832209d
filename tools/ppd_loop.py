"""Profiling target: PerishableProductDelivery rollout (16 384 slots), for rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphenvs_amd as ge
env = ge.make_vec(os.environ.get("GE_ENV", "PerishableProductDelivery-v0"), int(os.environ.get("GE_B", 16384)), n_nodes=64, n_edges=192, parenting=1)
env.reset(seed=0); env.random_rollout(20, 1); torch.cuda.synchronize()
env.random_rollout(200, 1); torch.cuda.synchronize()
