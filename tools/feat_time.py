"""Diagnostic: rocprofv3-free timing of the reset path at a given number of slots (HIP events around env.reset)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphenvs_amd as ge
for B in [int(a) for a in sys.argv[1:]] or [768, 2560]:
    env = ge.make_vec("ShortestPath-v0", B, n_nodes=64, n_edges=192)
    env.reset(seed=0); torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.reset(seed=1000 * rep); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print(f"B={B}: full reset (seed + graph + features kernels) median {sorted(ts)[2]:.1f} us", flush=True)
    env.close()
