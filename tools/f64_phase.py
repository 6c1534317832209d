"""Diagnostic: per-phase instruction counts of the n <= 64 feature kernel.  Step 1 (no profiler): build ablated variants
    python tools/f64_phase.py --build "base=" "nofwd=-DGE_F64_ABL=1" ...
Step 2 (under rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS ... --kernel-trace): run full resets on one variant
    python3 tools/f64_phase.py --run base
tools/f64_phase_summary.py turns the counter files into a table (differences between variants = the phase's instructions)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphenvs_amd import _lib
def path(name): return os.path.join(ROOT, "gpurun_out", f"libge_f64_{name}.so")
if sys.argv[1] == "--build":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for spec in sys.argv[2:]:
        name, _, flags = spec.partition("=")
        subprocess.check_call(_lib.compile_command(path(name), extra=flags.split()))
        print("built", name, flags)
else:
    import torch
    import graphenvs_amd as ge
    L = _lib.bind(C.CDLL(path(sys.argv[2])))
    B = int(os.environ.get("GE_B", 2688))
    env = ge.VectorGraphEnv("ShortestPath-v0", B, 64, 192, device="cuda", _library=L, prefetch=0)
    for rep in range(6):
        env.reset(seed=1000 * rep)
    torch.cuda.synchronize()
    env.close()
