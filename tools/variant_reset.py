"""Diagnostic: build library variants (-D flags) and time a full reset (seed + graph + feature kernels, HIP events) of any env /
geometry.  usage: GE_ENV=SteinerTree-v0 GE_N=256 GE_M=1024 GE_B=4096 python tools/variant_reset.py "NAME=-DFLAG=V ..." ..."""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphenvs_amd import _lib
import graphenvs_amd as ge
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
env_id, n, m, B = os.environ.get("GE_ENV", "SteinerTree-v0"), int(os.environ.get("GE_N", 256)), int(os.environ.get("GE_M", 1024)), int(os.environ.get("GE_B", 4096))
kw = json.loads(os.environ.get("GE_KW", '{"n_dests": 8}' if env_id == "SteinerTree-v0" else "{}"))
for spec in sys.argv[1:] or ["base="]:
    name, _, flags = spec.partition("=")
    out = os.path.join(ROOT, "gpurun_out", f"libge_var_{os.getpid()}_{name}.so")
    subprocess.check_call(_lib.compile_command(out, extra=flags.split()))
    L = _lib.bind(C.CDLL(out))
    env = ge.VectorGraphEnv(env_id, B, n, m, device="cuda", _library=L, prefetch=0, **kw)
    env.reset(seed=0); torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.reset(seed=1000 * rep); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"{name:12s} [{flags}] {env_id} n={n} m={m} B={B}: full reset median {sorted(ts)[2]:.2f} ms = {sorted(ts)[2] * 1e3 / B:.2f} us / slot", flush=True)
    env.close(); del env, L
    os.remove(out)
