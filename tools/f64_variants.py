"""Diagnostic: headline loop (ShortestPath n=64 m=192, 65 536 slots, device policy, autoreset) on library variants: ms per vector step.
usage: python tools/f64_variants.py "NAME=-DFLAG ..." ..."""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphenvs_amd import _lib
import graphenvs_amd as ge
for spec in sys.argv[1:] or ["base="]:
    name, _, flags = spec.partition("=")
    out = os.path.join(ROOT, "gpurun_out", f"libge_hv_{os.getpid()}_{name}.so")
    subprocess.check_call(_lib.compile_command(out, extra=flags.split()))
    L = _lib.bind(C.CDLL(out))
    env = ge.VectorGraphEnv("ShortestPath-v0", 65536, 64, 192, device="cuda", _library=L, prefetch=0)
    env.reset(seed=0); env.random_rollout(150, policy_seed=1); torch.cuda.synchronize()
    t0 = time.perf_counter(); env.random_rollout(200, policy_seed=1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tm = env.timed_rollout(100, policy_seed=1)
    print(f"{name:10s} [{flags}] {65536 * 200 / dt / 1e6:7.1f} M env-steps/s, {dt * 1e6 / 200:6.1f} us per step (autoreset part {tm['reset_ms'] * 10:.1f} us)", flush=True)
    env.close(); del env, L
    os.remove(out)
