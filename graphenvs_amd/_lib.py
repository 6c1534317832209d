"""ctypes binding of libgraphenvs_hip.so (C ABI: include/graphenvs.h).

The HIP library is the product: if it is missing or does not load, importing an engine raises.
There is no CPU fallback.  (tests/emu builds the same sources for a CPU sanitizer run and passes
that handle explicitly through the private ``_library`` argument of VectorGraphEnv.)
"""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.path.join(_PKG, "libgraphenvs_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

GE_OK = 0
ENV_TYPES = {
    "ShortestPath-v0": 0,
    "LongestPath-v0": 1,
    "SteinerTree-v0": 2,
    "TSP-v0": 3,
    "DensestSubgraph-v0": 4,
    "MaxIndependentSet-v0": 5,
    "MulticastRouting-v0": 6,
    "DistributionCenter-v0": 7,
    "PerishableProductDelivery-v0": 8,
}


class GeConfig(C.Structure):
    _fields_ = [
        ("env_type", C.c_int32), ("num_envs", C.c_int32), ("n_nodes", C.c_int32), ("n_edges", C.c_int32),
        ("weighted", C.c_int32), ("parenting", C.c_int32), ("n_dests", C.c_int32), ("spatial", C.c_int32),
        ("is_eval_env", C.c_int32), ("autoreset", C.c_int32), ("n_choices", C.c_double),
        ("env_index_base", C.c_int64), ("seed_stride", C.c_int64), ("node_id_base", C.c_int64),
        ("edge_row_stride", C.c_int64), ("max_distance", C.c_double), ("dt_min", C.c_double), ("dt_max", C.c_double),
    ]


class GeLayout(C.Structure):
    _fields_ = [
        ("F", C.c_int32), ("Fe", C.c_int32), ("A", C.c_int32), ("W", C.c_int32), ("E", C.c_int32),
        ("total_nodes", C.c_int64), ("total_edges", C.c_int64), ("obs_len", C.c_int64),
        ("reset_lds_bytes", C.c_int64), ("feat_parts", C.c_int32), ("eval_scratch_bytes", C.c_int64),
        ("prune_scratch_words", C.c_int64),
    ]


BUFFER_FIELDS = [
    "x", "edge_index", "edge_attr", "row_ptr", "colw", "scode", "sw64", "adj_bits", "node_rec", "rev_edge", "slot_rec", "terminals",
    "node_bits", "target_bits", "counters", "seed", "episode", "heuristic", "mt_state", "aux_bits",
    "mask", "mask_bits", "reward", "terminated", "invalid", "solved", "final_cost", "final_heur",
    "final_len", "reset_list", "reset_count", "work_list", "work_count", "feat_scratch",
    "node_aux", "range_bits", "cover_bits", "actions_out", "stream_state", "eval_scratch", "prune_scratch",
]
SEED_DEPTH = 3  # GE_SEED_DEPTH
STREAM_WORDS = 640  # GE_STREAM_WORDS


class GeBuffers(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in BUFFER_FIELDS]


class GeSpares(C.Structure):
    """ge_spares of include/graphenvs.h: the spare image of every slot and the queues of the episode prefetch"""
    _fields_ = [("image", GeBuffers), ("state", C.c_void_p), ("swap_list", C.c_void_p), ("swap_count", C.c_void_p),
                ("refill_list", C.c_void_p), ("refill_count", C.c_void_p), ("period", C.c_int32)]


# the per-slot slabs a spare image holds (the rest of ge_buffers sequences the engine and is shared with the live slabs)
IMAGE_FIELDS = ["x", "edge_index", "edge_attr", "row_ptr", "colw", "scode", "sw64", "adj_bits", "node_rec", "rev_edge", "slot_rec",
                "terminals", "node_bits", "target_bits", "counters", "heuristic", "aux_bits", "mask", "mask_bits", "node_aux",
                "range_bits", "cover_bits"]

# every symbol include/graphenvs.h declares
SYMBOLS = [
    "ge_abi_version", "ge_get_layout", "ge_create", "ge_destroy", "ge_ragged_table_bytes", "ge_create_ragged", "ge_reset", "ge_step", "ge_step_only",
    "ge_reset_pending", "ge_reset_continue", "ge_inject_state", "ge_mark_restored", "ge_vectorize", "ge_sample_actions", "ge_random_rollout",
    "ge_timed_rollout", "ge_timed_step_burst", "ge_timed_empty_burst", "ge_last_error", "ge_source_hash", "ge_attach_spares",
]


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))] + [
        os.path.join(ROOT, "include", "graphenvs.h")]


def source_hash() -> str:
    """sha256 over the HIP sources and the public header, in file-name order (what the binary was built from)."""
    import hashlib
    h = hashlib.sha256()
    for path in sources():
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()[:32]


def compile_command(out_path: str, extra=()):
    return [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
            # float64 feature/baseline arithmetic must match CPython/numpy bit for bit: no FMA contraction
            "-ffp-contract=off", "-I" + CSRC, '-DGE_SOURCE_HASH="%s"' % source_hash(), *extra,
            os.path.join(CSRC, "ge_api.hip"), "-o", out_path]


def built_hash(path: str = None):
    """source hash embedded in the library file on disk, read from its bytes (no dlopen: a process that already mapped an
    older file of the same name would be handed that one again).  None: no file, or a binary from before the hash existed."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        return None
    data = open(path, "rb").read()
    k = data.find(b"GE_SOURCE_HASH=")
    return data[k + 15:k + 47].decode("ascii", "replace") if k >= 0 else None


last_build = None  # "compiled" or "reused": what the last build() call did


def build(force: bool = False, verbose: bool = False) -> str:
    """Cross-compile the HIP library for gfx950 (works without a GPU).  The binary carries the hash of the sources it
    was built from (ge_source_hash); it is reused only when that hash equals the hash of the sources on disk."""
    global last_build, _lib
    if not force and built_hash() == source_hash():
        last_build = "reused"
        return LIB_PATH
    tmp = "%s.%d.tmp" % (LIB_PATH, os.getpid())  # per process: ranks that all find a stale library must not write one file
    cmd = compile_command(tmp)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL,
                          stderr=None if verbose else subprocess.DEVNULL)
    os.replace(tmp, LIB_PATH)  # atomic, a new inode: a process that already mapped the old file keeps the old one
    last_build = "compiled"
    _lib = None
    return LIB_PATH


def bind(lib):
    """Attach argtypes/restypes for every entry point of include/graphenvs.h."""
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    lib.ge_abi_version.restype = C.c_int
    lib.ge_abi_version.argtypes = []
    lib.ge_get_layout.restype = C.c_int
    lib.ge_get_layout.argtypes = [C.POINTER(GeConfig), C.POINTER(GeLayout)]
    lib.ge_create.restype = C.c_int
    lib.ge_create.argtypes = [C.POINTER(GeConfig), C.POINTER(GeBuffers), C.POINTER(vp)]
    lib.ge_destroy.restype = C.c_int
    lib.ge_destroy.argtypes = [vp]
    lib.ge_ragged_table_bytes.restype = C.c_int64
    lib.ge_ragged_table_bytes.argtypes = [i32]
    lib.ge_create_ragged.restype = C.c_int
    lib.ge_create_ragged.argtypes = [C.POINTER(GeConfig), C.POINTER(GeBuffers), i32, vp, vp, vp, C.POINTER(vp)]
    if hasattr(lib, "ge_attach_spares"):  # (a library of an older ABI bound for a timing comparison lacks it)
        lib.ge_attach_spares.restype = C.c_int
        lib.ge_attach_spares.argtypes = [vp, C.POINTER(GeSpares), vp]
    lib.ge_reset.restype = C.c_int
    lib.ge_reset.argtypes = [vp, vp, vp]
    for name in ("ge_step", "ge_step_only"):
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = [vp, vp, vp]
    lib.ge_reset_pending.restype = C.c_int
    lib.ge_reset_pending.argtypes = [vp, vp]
    lib.ge_inject_state.restype = C.c_int
    lib.ge_inject_state.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.ge_mark_restored.restype = C.c_int
    lib.ge_mark_restored.argtypes = [vp]
    lib.ge_reset_continue.restype = C.c_int
    lib.ge_reset_continue.argtypes = [vp, vp]
    lib.ge_vectorize.restype = C.c_int
    lib.ge_vectorize.argtypes = [vp, vp, vp]
    lib.ge_sample_actions.restype = C.c_int
    lib.ge_sample_actions.argtypes = [vp, u64, vp, vp]
    lib.ge_random_rollout.restype = C.c_int
    lib.ge_random_rollout.argtypes = [vp, u64, i32, vp, vp]
    lib.ge_timed_rollout.restype = C.c_int
    lib.ge_timed_rollout.argtypes = [vp, u64, i32, vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]
    lib.ge_timed_step_burst.restype = C.c_int
    lib.ge_timed_step_burst.argtypes = [vp, u64, i32, vp, vp, C.POINTER(C.c_double)]
    if hasattr(lib, "ge_timed_empty_burst"):
        lib.ge_timed_empty_burst.restype = C.c_int
        lib.ge_timed_empty_burst.argtypes = [vp, i32, vp, C.POINTER(C.c_double)]
    lib.ge_last_error.restype = C.c_char_p
    lib.ge_last_error.argtypes = []
    lib.ge_source_hash.restype = C.c_char_p
    lib.ge_source_hash.argtypes = []
    return lib


_lib = None


ABI_VERSION = 5  # GE_ABI_VERSION of include/graphenvs.h this host was written against


def load():
    """Load the HIP library; raise loudly if it is not there (no fallback).  A binary whose embedded source hash
    differs from the sources on disk is stale: it is rebuilt when hipcc is there, else refused."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"graphenvs_amd: {LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        want, have = source_hash(), built_hash()
        if have != want:
            if not os.path.exists(HIPCC):
                raise RuntimeError(f"graphenvs_amd: {LIB_PATH} was built from other sources (hash {have}, sources {want}) "
                                   "and hipcc is not available to rebuild it")
            import sys
            print(f"graphenvs_amd: {LIB_PATH} is stale (hash {have}, sources {want}): rebuilding", file=sys.stderr)
            build(force=True)
        lib = C.CDLL(LIB_PATH)
        for s in SYMBOLS:
            if not hasattr(lib, s):
                raise RuntimeError(f"graphenvs_amd: {LIB_PATH} does not export {s}")
        bind(lib)
        if lib.ge_abi_version() != ABI_VERSION:
            raise RuntimeError("graphenvs_amd: ABI version mismatch between the python host and the HIP library")
        if lib.ge_source_hash().decode() != want:
            raise RuntimeError("graphenvs_amd: the loaded library does not carry the hash of the sources on disk")
        _lib = lib
    return _lib


def check(lib, rc, what):
    if rc != GE_OK:
        msg = lib.ge_last_error()
        raise RuntimeError(f"graphenvs_amd: {what} failed (code {rc}): {msg.decode() if msg else ''}")
