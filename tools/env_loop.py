"""Profiling target: a rollout of any env config, for rocprofv3 --kernel-trace.
usage: python tools/env_loop.py <env_id> <num_envs> <steps> '<json kwargs>'"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphenvs_amd as ge
env_id, B, K, kw = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), json.loads(sys.argv[4])
env = ge.make_vec(env_id, B, **kw)
env.reset(seed=0); env.random_rollout(5, 1); torch.cuda.synchronize()
env.random_rollout(K, 1); torch.cuda.synchronize()
