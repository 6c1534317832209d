"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper over the C oracle (oracle/ge_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (graphenvs_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libge_oracle.so")

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


class OgeCfg(C.Structure):
    _fields_ = [
        ("env_type", C.c_int32), ("n_nodes", C.c_int32), ("n_edges", C.c_int32),
        ("weighted", C.c_int32), ("parenting", C.c_int32), ("n_dests", C.c_int32),
        ("spatial", C.c_int32), ("is_eval_env", C.c_int32), ("n_choices", C.c_double), ("max_distance", C.c_double),
        ("dt_min", C.c_double), ("dt_max", C.c_double),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (no GPU involved)."""
    src = os.path.join(_HERE, "ge_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "ge_oracle.h")),
            os.path.getmtime(os.path.join(_HERE, "..", "graphenvs_amd", "csrc", "ge_christofides.h")),
            os.path.getmtime(os.path.join(_HERE, "..", "graphenvs_amd", "csrc", "ge_clique_removal.h")),
            os.path.getmtime(os.path.join(_HERE, "..", "graphenvs_amd", "csrc", "ge_kou_exact.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libge_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
        L.oge_create.restype = vp
        L.oge_create.argtypes = [C.POINTER(OgeCfg)]
        L.oge_destroy.argtypes = [vp]
        L.oge_reset.restype = C.c_int
        L.oge_reset.argtypes = [vp, i64]
        L.oge_step.restype = C.c_int
        L.oge_step.argtypes = [vp, i64, C.POINTER(dbl), C.POINTER(i32), C.POINTER(i32)]
        for name in ("oge_num_node_features", "oge_num_edge_features", "oge_num_directed_edges",
                     "oge_mask_size", "oge_obs_size", "oge_head", "oge_num_targets"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [vp]
        for name in ("oge_get_nodes", "oge_get_edges", "oge_get_edge_links", "oge_get_obs",
                     "oge_get_mask", "oge_get_features64", "oge_get_terminals"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [vp, vp]
        L.oge_solution_cost.restype = dbl
        L.oge_solution_cost.argtypes = [vp]
        L.oge_heuristic_solution.restype = dbl
        L.oge_heuristic_solution.argtypes = [vp]
        L.oge_inject.restype = C.c_int
        L.oge_inject.argtypes = [vp, vp, vp, vp, vp, C.c_int]
        L.oge_policy_pick.restype = i64
        L.oge_policy_pick.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64]
        L.oge_rollout.restype = i64
        L.oge_rollout.argtypes = [C.POINTER(OgeCfg), i64, i64, i32, i32, C.c_uint64, i32,
                                  C.POINTER(dbl), C.POINTER(i64), C.POINTER(dbl)]
        L.oge_debug_greedy_mis.restype = dbl
        L.oge_debug_greedy_mis.argtypes = [vp, vp]
        L.oge_debug_steiner_tree.restype = dbl
        L.oge_debug_steiner_tree.argtypes = [vp, vp]
        L.oge_debug_christofides.restype = i64
        L.oge_debug_christofides.argtypes = [i32, vp]
        L.oge_debug_min_matching.restype = i64
        L.oge_debug_min_matching.argtypes = [i32, vp, vp]
        L.oge_debug_pyset_int_order.restype = i32
        L.oge_debug_pyset_int_order.argtypes = [vp, i32, vp]
        L.oge_debug_non_neighbors.restype = i32
        L.oge_debug_non_neighbors.argtypes = [vp, i32, vp, i32, vp]
        L.oge_debug_clique_removal.restype = i32
        L.oge_debug_clique_removal.argtypes = [i32, i32, vp, vp]
        L.oge_debug_kou_exact.restype = dbl
        L.oge_debug_kou_exact.argtypes = [i32, i32, i32, vp, vp, vp, vp]
        L.oge_pyset_order.restype = C.c_int
        L.oge_pyset_order.argtypes = [vp, vp, C.c_int, vp, vp]
        L.oge_mt_py_seed.argtypes = [vp, C.c_uint32]
        L.oge_mt_np_seed.argtypes = [vp, C.c_uint32]
        L.oge_mt_next.restype = C.c_uint32
        L.oge_mt_next.argtypes = [vp]
        _lib = L
    return _lib


def make_cfg(env_id, n_nodes, n_edges=-1, weighted=None, parenting=None, n_dests=3, spatial=False,
             is_eval_env=False, n_choices=-1, max_distance=1, target_count=-1, n_products=3, delivery_time=-1, **_ignored) -> OgeCfg:
    t = ENV_TYPES[env_id]
    if parenting is None:
        parenting = 4 if t == 6 else -1  # multicast_routing.py:31
    if n_edges == -1:
        assert t in (6, 8), "n_edges is required"
        n_edges = int((n_nodes * (n_nodes - 1) // 2) * 0.30)  # multicast_routing.py:53-54
    if weighted is None:
        weighted = (t != 4)  # DensestSubgraph defaults to weighted=False (densest_subgraph.py:25)
    dt = (0.0, 0.0)
    if t == 8:  # perishable_product_delivery.py:27,34,53-61
        assert parenting in [1], "Parenting must be 1!"
        assert n_products <= 5, "Max 5 products!"
        assert delivery_time == -1, "the reference only runs with delivery_time=-1"
        n_dests = n_products
        avg_dist = np.log(n_nodes) / np.log(2 * n_edges / n_nodes)
        if weighted:
            avg_dist = avg_dist * (0.3 + 1.0) / 2.0
        dt = (float(avg_dist * 0.6), float(avg_dist * 1.4))
    if t == 7:  # distribution_center.py:29,42-45
        parenting = 2 if parenting == -1 else parenting
        n_dests = n_nodes // 5 if target_count == -1 else target_count
    return OgeCfg(t, n_nodes, n_edges, int(bool(weighted)), int(parenting), int(n_dests),
                  int(bool(spatial)), int(bool(is_eval_env)), float(n_choices), float(max_distance), *dt)


class OracleEnv:
    """Single-env oracle with the reference's reset()/step() shape (numpy in, numpy out)."""

    def __init__(self, env_id, **kwargs):
        self.env_id = env_id
        self.cfg = make_cfg(env_id, **kwargs)
        self._L = lib()
        self._h = self._L.oge_create(C.byref(self.cfg))
        self.n = self.cfg.n_nodes
        self.F = self._L.oge_num_node_features(self._h)
        self.Fe = self._L.oge_num_edge_features(self._h)
        self.E = self._L.oge_num_directed_edges(self._h)
        self.A = self._L.oge_mask_size(self._h)
        self.L = self._L.oge_obs_size(self._h)
        self.attempts = 0

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.oge_destroy(self._h)
            self._h = None

    # --- readers
    def _read(self, fn, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        getattr(self._L, fn)(self._h, out.ctypes.data)
        return out

    def obs(self):
        return self._read("oge_get_obs", (self.L,), np.float32)

    def mask(self):
        return self._read("oge_get_mask", (self.A,), np.uint8).astype(bool)

    def nodes(self):
        return self._read("oge_get_nodes", (self.n, self.F), np.float32)

    def edges(self):
        return self._read("oge_get_edges", (self.E, self.Fe), np.float32)

    def edge_links(self):
        return self._read("oge_get_edge_links", (self.E, 2), np.int64)

    def features64(self):
        return self._read("oge_get_features64", (self.n, 5), np.float64)

    def terminals(self):
        k = self._L.oge_num_targets(self._h) + 1
        return self._read("oge_get_terminals", (k,), np.int32)

    @property
    def head(self):
        return self._L.oge_head(self._h)

    def debug_greedy_mis(self):
        out = np.zeros(self.n, dtype=np.uint8)
        return self._L.oge_debug_greedy_mis(self._h, out.ctypes.data), out.astype(bool)

    def debug_steiner_tree(self):
        out = np.zeros((self.n, self.n), dtype=np.uint8)
        return self._L.oge_debug_steiner_tree(self._h, out.ctypes.data), out.astype(bool)

    @property
    def solution_cost(self):
        return self._L.oge_solution_cost(self._h)

    @property
    def heuristic_solution(self):
        return self._L.oge_heuristic_solution(self._h)

    # --- reference-shaped API
    def reset(self, seed=None):
        self.attempts = self._L.oge_reset(self._h, -1 if seed is None else int(seed))
        return self.obs(), {"mask": self.mask()}

    def step(self, action):
        r, d, s = C.c_double(), C.c_int32(), C.c_int32()
        rc = self._L.oge_step(self._h, int(action), C.byref(r), C.byref(d), C.byref(s))
        if rc != 0:
            raise AssertionError(f"invalid action {action}")
        info = {"mask": self.mask()}
        if s.value >= 0:
            info["solved"] = bool(s.value)
        if d.value:
            info["solution_cost"] = self.solution_cost
            info["heuristic_solution"] = self.heuristic_solution
        return self.obs(), r.value, bool(d.value), False, info

    def inject(self, links, w64, x, terminals):
        links = np.ascontiguousarray(links, dtype=np.int64)
        w64 = np.ascontiguousarray(w64, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float32)
        terminals = np.ascontiguousarray(terminals, dtype=np.int32)
        rc = self._L.oge_inject(self._h, links.ctypes.data, w64.ctypes.data, x.ctypes.data,
                                terminals.ctypes.data, len(terminals))
        assert rc == 0


def policy_pick(mask, policy_seed, env_index, t):
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    return int(lib().oge_policy_pick(m.ctypes.data, len(m), policy_seed, env_index, t))


def rollout(env_id, n_envs, n_steps, first_seed=0, seed_stride=None, policy_seed=0, n_threads=0,
            **kwargs):
    """Random-policy rollout with autoreset over OpenMP threads (cpu_baseline leg)."""
    cfg = make_cfg(env_id, **kwargs)
    sr, ep, rs = C.c_double(), C.c_int64(), C.c_double()
    stride = n_envs if seed_stride is None else seed_stride
    total = lib().oge_rollout(C.byref(cfg), first_seed, stride, n_envs, n_steps, policy_seed,
                              n_threads, C.byref(sr), C.byref(ep), C.byref(rs))
    return {"transitions": int(total), "episodes": int(ep.value), "sum_reward": sr.value,
            "reset_seconds_sum": rs.value}


def mt_stream(kind, seed, count):
    st = np.zeros(625, dtype=np.uint32)
    L = lib()
    (L.oge_mt_py_seed if kind == "py" else L.oge_mt_np_seed)(st.ctypes.data, seed)
    return np.array([L.oge_mt_next(st.ctypes.data) for _ in range(count)], dtype=np.uint32)


def pyset_order(pairs):
    """iteration order of set(pairs) built by adding the int pairs one by one (CPython 3.10 set emulation in the oracle)"""
    u = np.ascontiguousarray([p[0] for p in pairs], dtype=np.int32)
    v = np.ascontiguousarray([p[1] for p in pairs], dtype=np.int32)
    ou, ov = np.zeros(len(pairs) + 1, dtype=np.int32), np.zeros(len(pairs) + 1, dtype=np.int32)
    k = lib().oge_pyset_order(u.ctypes.data, v.ctypes.data, len(pairs), ou.ctypes.data, ov.ctypes.data)
    return [(int(a), int(b)) for a, b in zip(ou[:k], ov[:k])]


def christofides_units(D):
    """tour length (integer units) of the checker's Christofides tour on the closure matrix D [n, n] int32"""
    D = np.ascontiguousarray(D, dtype=np.int32)
    return int(lib().oge_debug_christofides(D.shape[0], D.ctypes.data))


def min_matching(dist):
    """(minimum weight, partner of every vertex) of a perfect matching of the complete graph with lengths dist [k, k] int32"""
    dist = np.ascontiguousarray(dist, dtype=np.int32)
    out = np.zeros(dist.shape[0], dtype=np.int32)
    return int(lib().oge_debug_min_matching(dist.shape[0], dist.ctypes.data, out.ctypes.data)), out


def pyset_int_order(keys):
    """iteration order of set() after adding the ints `keys` in order, by the checker's restatement of setobject.c"""
    k = np.ascontiguousarray(keys, dtype=np.int32); out = np.zeros(max(1, len(k)), dtype=np.int32)
    n = lib().oge_debug_pyset_int_order(k.ctypes.data, len(k), out.ctypes.data)
    assert n >= 0
    return out[:n].tolist()


def non_neighbors_order(nodes, first_adj):
    """iteration order of nx.non_neighbors(G, nodes[0]) for a graph with node dict order `nodes` and G.adj[nodes[0]] order `first_adj`"""
    a = np.ascontiguousarray(nodes, dtype=np.int32); b = np.ascontiguousarray(first_adj, dtype=np.int32)
    out = np.zeros(max(1, len(a)), dtype=np.int32)
    n = lib().oge_debug_non_neighbors(a.ctypes.data, len(a), b.ctypes.data, len(b), out.ctypes.data)
    assert n >= 0
    return out[:n].tolist()


def clique_removal_len(n, row_ptr, col):
    rp = np.ascontiguousarray(row_ptr, dtype=np.int32); c = np.ascontiguousarray(col, dtype=np.int32)
    return int(lib().oge_debug_clique_removal(n, len(c) // 2, rp.ctypes.data, c.ctypes.data))


def kou_exact(n, row_ptr, col, w, terms):
    """sum of delays over networkx's Kou Steiner tree, by the checker's restatement (graph as insertion-order CSR)"""
    rp = np.ascontiguousarray(row_ptr, dtype=np.int32); c = np.ascontiguousarray(col, dtype=np.int32)
    ww = np.ascontiguousarray(w, dtype=np.float64); t = np.ascontiguousarray(terms, dtype=np.int32)
    return float(lib().oge_debug_kou_exact(n, len(c) // 2, len(t), rp.ctypes.data, c.ctypes.data, ww.ctypes.data, t.ctypes.data))
