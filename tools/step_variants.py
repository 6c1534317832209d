"""Diagnostic: median launch time of the fused ShortestPath step kernel for library variants (-D flags), measured the
way bench.py does (bursts of 5 launches between one HIP event pair, minus an empty pair).
usage: python tools/step_variants.py "" "-DGE_STEP_BLOCK=64" ..."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphenvs_amd import _lib
import graphenvs_amd as ge
for vi, spec in enumerate(sys.argv[1:] or [""]):
    flags = spec.split()
    out = os.path.join(ROOT, "gpurun_out", f"libge_variant_{os.getpid()}_{vi}.so")  # a new path per variant: dlopen caches by path
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(_lib.compile_command(out, extra=flags))
    L = _lib.bind(C.CDLL(out))
    B = int(os.environ.get("GE_B", 65536))
    env = ge.VectorGraphEnv("ShortestPath-v0", B, 64, 192, device="cuda", _library=L)
    env.reset(seed=0); env.random_rollout(5, 1); torch.cuda.synchronize()
    empty = sorted(env.timed_step_burst_raw_ms(0) for _ in range(9))[4]
    res = []
    for rep in range(10):
        env.reset(seed=rep); torch.cuda.synchronize()
        res.append((env.timed_step_burst_raw_ms(5, policy_seed=2) - empty) * 1e3 / 5)
    res.sort()
    print(f"[{spec}] B={B} step kernel median {res[5]:.2f} us  min {res[0]:.2f} us", flush=True)
    env.close(); del env, L
    os.remove(out)
