"""Diagnostic: a full reset (every slot, ITEMS_ALL launches) against the refill of every spare image (queue launches on the image view)
of the same slots: reset() of an engine with spares runs both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphenvs_amd as ge
for B in (1024, 3216, 4096, 16384):
    res = {}
    for pf in (0, 1000):
        env = ge.make_vec("SteinerTree-v0", B, n_nodes=256, n_edges=1024, n_dests=8, prefetch=pf)
        env.reset(seed=0); torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); env.reset(seed=1000 * rep); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res[pf] = sorted(ts)[2]
        env.close()
    print(f"B={B}: full reset {res[0]:.2f} ms = {res[0] * 1e3 / B:.2f} us/slot; reset + refill of every image {res[1000]:.2f} ms -> refill {res[1000] - res[0]:.2f} ms = {(res[1000] - res[0]) * 1e3 / B:.2f} us/slot", flush=True)
