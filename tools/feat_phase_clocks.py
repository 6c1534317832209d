"""Diagnostic: where wave 0 of slot 0's first Brandes workgroup spends its shader-clock cycles in the generic feature kernel
(-DGE_STAMPS build; GE_ENV / GE_N / GE_M / GE_BS as tools/phase_stamps_any.py)."""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphenvs_amd import _lib
out = os.path.join(ROOT, "gpurun_out", "libgraphenvs_hip_stamps.so")
subprocess.check_call(_lib.compile_command(out, extra=["-DGE_STAMPS"] + [a for a in sys.argv[1:] if a.startswith("-D")]))
L = _lib.bind(C.CDLL(out)); L.ge_debug_read_stamps.argtypes = [C.c_void_p]
import graphenvs_amd as ge
env_id, n, m = os.environ.get("GE_ENV", "SteinerTree-v0"), int(os.environ.get("GE_N", 256)), int(os.environ.get("GE_M", 1024))
names = ["new level + front", "forward walk", "-", "coefficients", "dependency pull", "end of source", "set-up of source", "outside"]
for B in [int(b) for b in os.environ.get("GE_BS", "1,4096").split(",")]:
    env = ge.VectorGraphEnv(env_id, B, n, m, device="cuda", _library=L, prefetch=0, **json.loads(os.environ.get("GE_KW", "{}")))
    for rep in range(2):
        env.reset(seed=rep); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)(); L.ge_debug_read_stamps(buf); ts = [buf[k] for k in range(32)]
    src, lev = max(1, ts[24]), max(1, ts[25]); tot = sum(ts[16:23])
    print(f"{env_id} n={n} m={m} B={B}: {src} sources, {lev} levels on this wave, {tot / src:.0f} cycles per source")
    print(f"    outside the source loop (staging, hand-over) {ts[23]:9.0f} cycles per workgroup")
    for k, nm in enumerate(names[:7]):
        print(f"    {nm:18s} {ts[16 + k] / src:9.0f} cycles per source  {ts[16 + k] / tot:6.3f}")
    env.close()
