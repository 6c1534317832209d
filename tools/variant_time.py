"""Diagnostic: build library variants (-D flags) and time the reset kernels at a few slot counts with rocprofv3-free
HIP events (whole reset = seed + graph + features).  usage: python tools/variant_time.py "NAME=-DFLAG=V ..." ..."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphenvs_amd import _lib
import graphenvs_amd as ge
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for spec in sys.argv[1:]:
    name, _, flags = spec.partition("=")
    out = os.path.join(ROOT, "gpurun_out", f"libge_var_{name}.so")
    subprocess.check_call(_lib.compile_command(out, extra=flags.split()))
    L = _lib.bind(C.CDLL(out))
    res = []
    for B in (768, 2560):
        env = ge.VectorGraphEnv("ShortestPath-v0", B, 64, 192, device="cuda", _library=L)
        env.reset(seed=0); torch.cuda.synchronize()
        ts = []
        for rep in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); env.reset(seed=1000 * rep); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        res.append((B, sorted(ts)[3]))
        env.close()
    print(name, flags, " ".join(f"B={b}: {t:.0f}us" for b, t in res), flush=True)
    os.remove(out)
