"""Diagnostic: DistributionCenter throughput for -DGE_DC_LANES variants (sources searched at a time in the reset kernel)."""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphenvs_amd import _lib
import graphenvs_amd as ge
for vi, lanes in enumerate([int(a) for a in sys.argv[1:]] or [64, 32, 16]):
    out = os.path.join(ROOT, "gpurun_out", f"libge_dc_{os.getpid()}_{vi}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([_lib.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", f"-DGE_DC_LANES={lanes}",
                           "-I" + _lib.CSRC, os.path.join(_lib.CSRC, "ge_api.hip"), "-o", out])
    L = _lib.bind(C.CDLL(out))
    env = ge.VectorGraphEnv("DistributionCenter-v0", 65536, 64, 192, device="cuda", _library=L)
    env.reset(seed=0); env.random_rollout(10, 1); torch.cuda.synchronize()
    t0 = time.perf_counter(); env.random_rollout(60, 1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"GE_DC_LANES={lanes}: {65536 * 60 / dt / 1e6:.1f} M env-steps/s, {dt / 60 * 1e3:.2f} ms per vector step", flush=True)
    env.close(); os.remove(out)
