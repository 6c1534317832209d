"""Diagnostic: build the library with -DGE_STAMP_SLOTS (never shipped) and print, for one queue-mode graph launch of the headline config in
steady state, when its slots start and finish relative to the first start.  Run on the GPU box."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from graphenvs_amd import _lib  # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "libgraphenvs_hip_slotstamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(_lib.compile_command(out, extra=["-DGE_STAMP_SLOTS"]))
L = _lib.bind(C.CDLL(out))
import graphenvs_amd as ge  # noqa: E402

env = ge.VectorGraphEnv("ShortestPath-v0", 65536, n_nodes=64, n_edges=192, device="cuda", _library=L)
env.reset(seed=0)
env.random_rollout(160, policy_seed=1)
torch.cuda.synchronize()
for rep in range(3):
    ep0 = env.t["episode"].clone()
    env.random_rollout(1, policy_seed=1)
    torch.cuda.synchronize()
    who = torch.nonzero(env.t["episode"] != ep0).flatten()
    t0 = env.t["final_cost"][who].cpu().numpy(); t1 = env.t["final_heur"][who].cpu().numpy(); acc = env.t["final_len"][who].cpu().numpy() / 100.0
    base = t0.min()
    s, e = (t0 - base) / 100.0, (t1 - base) / 100.0  # us
    d = e - s
    q = lambda a, p: float(np.percentile(a, p))
    print(f"launch {rep}: {len(who)} slots; start p50 {q(s,50):.1f} p90 {q(s,90):.1f} max {s.max():.1f} us; "
          f"duration p10 {q(d,10):.1f} p50 {q(d,50):.1f} p90 {q(d,90):.1f} p99 {q(d,99):.1f} max {d.max():.1f} us; "
          f"graph accepted after p50 {q(acc,50):.1f} p90 {q(acc,90):.1f} max {acc.max():.1f} us; last end {e.max():.1f} us")
    late = np.argsort(e)[-5:]
    print("   latest slots: start", np.round(s[late], 1).tolist(), "accepted after", np.round(acc[late], 1).tolist(), "duration", np.round(d[late], 1).tolist())
