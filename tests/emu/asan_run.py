"""TEST INFRASTRUCTURE ONLY: a slice of the CPU-harness tests under AddressSanitizer.  Run by tests/test_emu_kernels.py in a child
process with LD_PRELOAD=libasan (the interpreter itself is not instrumented; every malloc of the process -- torch's CPU tensors,
i.e. the slabs, and the per-block LDS -- gets red zones).  Exits non-zero on the first finding."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), HERE):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import build_emu  # noqa: E402
import golden_util as gu  # noqa: E402
import host_checks as hc  # noqa: E402
import oracle  # noqa: E402
import graphenvs_amd as ge  # noqa: E402

emu = build_emu.load(asan=True)
CASES = sys.argv[1:] or ["sp_n10_m20_eval", "st_n10_m20_d3_eval", "tsp_n10_m20_p1_eval", "mis_n12_m20_unweighted_eval", "mc_n10_m20_p4_eval",
                         "ppd_n7_m21_complete"]
for name in CASES:
    st = gu.replay_case(gu.load_case(name), lambda env_id, **kw: ge.GraphEnv(env_id, device="cpu", _library=emu, **kw), policies=("first",))
    assert st["resets"] > 0
    print("asan ok:", name, flush=True)
if len(sys.argv) == 1:
    hc.check_inject_seeds_autoreset(ge, oracle, "cpu", emu, True)
    print("asan ok: host checks", flush=True)
print("ASAN RUN COMPLETE", flush=True)
