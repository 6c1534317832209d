"""A/B of library builds on full resets of one geometry (same box, same process): GE_ENV / GE_N / GE_M / GE_B / GE_KW as
tools/variant_reset.py; python tools/ab_reset.py name=path.so ... ("cur" = the built library)"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphenvs_amd import _lib
import graphenvs_amd as ge
env_id, n, m, B = os.environ.get("GE_ENV", "SteinerTree-v0"), int(os.environ.get("GE_N", 256)), int(os.environ.get("GE_M", 1024)), int(os.environ.get("GE_B", 4096))
kw = json.loads(os.environ.get("GE_KW", '{"n_dests": 8}' if env_id == "SteinerTree-v0" else "{}"))
for spec in sys.argv[1:]:
    name, _, path = spec.partition("=")
    L = _lib.load() if path == "cur" else _lib.bind(C.CDLL(os.path.join(ROOT, path)))
    env = ge.VectorGraphEnv(env_id, B, n, m, device="cuda", _library=L, prefetch=0, **kw)
    env.reset(seed=0); torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.reset(seed=1000 * rep); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"{name:8s} {env_id} n={n} m={m} B={B}: full reset median {sorted(ts)[2]:.2f} ms = {sorted(ts)[2] * 1e3 / B:.2f} us / slot", flush=True)
    env.close(); del env
