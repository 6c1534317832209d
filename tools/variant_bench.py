"""Diagnostic: build a library variant (-D flags) and run the bench loop with it (for rocprofv3 kernel traces).
usage: python tools/variant_bench.py "-DFLAG=V ..." [steps]"""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphenvs_amd import _lib
import graphenvs_amd as ge
flags = sys.argv[1].split()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
out = os.path.join(ROOT, "gpurun_out", "libge_variant.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
cmd = _lib.compile_command(out, extra=[f for f in flags if not f.startswith("--offload-arch")])
arch = [f for f in flags if f.startswith("--offload-arch")]  # (a target given here replaces the default one, e.g. gfx950:xnack-)
if arch: cmd = [arch[0] if c.startswith("--offload-arch") else c for c in cmd]
subprocess.check_call(cmd)
L = _lib.bind(C.CDLL(out))
env = ge.VectorGraphEnv("ShortestPath-v0", 65536, 64, 192, device="cuda", _library=L)
env.reset(seed=0); env.random_rollout(10, 1); torch.cuda.synchronize()
t0 = time.perf_counter(); env.random_rollout(steps, 1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(" ".join(flags), f"{65536 * steps / dt / 1e6:.1f} M env-steps/s", flush=True)
