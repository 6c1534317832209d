"""TEST INFRASTRUCTURE ONLY: build the CPU sanitizer harness of the HIP kernels (see hip_emu.h)."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "graphenvs_amd", "csrc")
OUT = os.path.join(HERE, "libgraphenvs_emu.so")


ASAN_OUT = os.path.join(HERE, "libgraphenvs_emu_asan.so")


def build(force=False, extra=(), out=None, asan=False):
    """asan=True: AddressSanitizer beside UBSan (out-of-bounds on the slabs and behind a block's dynamic LDS, which is then a heap
    block of exactly the requested size); the python that loads it must run under LD_PRELOAD of libasan (asan_preload())."""
    OUT_ = out if out is not None else (ASAN_OUT if asan else OUT)
    return _build(force, list(extra), OUT_, asan)


def asan_preload():
    return subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()


def _build(force, extra, OUT, asan=False):
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "hip_emu.h"),
                                                               os.path.join(ROOT, "include", "graphenvs.h")]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(s) for s in srcs):
        return OUT
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-DGE_EMU", "-x", "c++", "-ffp-contract=off",
           "-I" + HERE, "-I" + CSRC, "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
           "-fsanitize=address,undefined" if asan else "-fsanitize=undefined", "-fno-sanitize-recover=undefined", *extra,
           os.path.join(CSRC, "ge_api.hip"), "-o", OUT]
    subprocess.check_call(cmd)
    return OUT


def load(extra=(), out=None, asan=False):
    sys.path.insert(0, ROOT)
    from graphenvs_amd import _lib
    return _lib.bind(ctypes.CDLL(build(extra=extra, out=out, asan=asan)))


if __name__ == "__main__":
    print(build(force=True))
