"""Profiling target: full resets of one env config (every slot regenerated: graph kernel + feature kernels over all B slots), for
rocprofv3 --kernel-trace / --pmc.  usage: python tools/reset_loop.py <env_id> <num_envs> <resets> '<json kwargs>'"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphenvs_amd as ge
env_id, B, K, kw = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), json.loads(sys.argv[4])
env = ge.make_vec(env_id, B, prefetch=0, **kw)
for k in range(K):
    env.reset(seed=k * B)
torch.cuda.synchronize()
