"""Diagnostic: resident workgroups per CU (runtime occupancy query) and LDS bytes of the reset-path kernels."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphenvs_amd as ge
from graphenvs_amd import _lib
env = ge.make_vec("ShortestPath-v0", 65536, n_nodes=64, n_edges=192)
L = _lib.load()
L.ge_debug_occupancy.argtypes = [C.c_void_p, C.c_void_p]
out = (C.c_int * 4)()
L.ge_debug_occupancy(env._h, out)
print("blocks/CU: reset", out[0], "features64", out[1], "features(generic)", out[2], "step", out[3], "| reset LDS", env.layout.reset_lds_bytes)
