"""Batched graph environments on one MI355X: the host side of the C ABI.

Mirrors the reference's Gymnasium surface (graph_envs/__init__.py:9-56 and the six env classes):
same ids, same constructor kwargs/defaults/asserts, ``reset(seed=)`` / ``step(actions)`` /
``info['mask']``; the observation is the PyG-shaped view of utils.to_pyg_graph (utils.py:26-29)
kept resident in HBM.  All compute happens in the HIP kernels behind include/graphenvs.h; torch
is used for device memory and streams only.
"""
import ctypes as C
import math
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib

ENV_IDS = tuple(_lib.ENV_TYPES)

# constructor signatures of the reference
_DEFAULTS = {
    "ShortestPath-v0": dict(weighted=True, return_graph_obs=False, parenting=-1, structural_features=True, is_eval_env=False),
    "LongestPath-v0": dict(weighted=True, return_graph_obs=False, is_eval_env=False, parenting=-1),
    "SteinerTree-v0": dict(n_dests=3, weighted=True, parenting=-1, is_eval_env=False),
    "TSP-v0": dict(weighted=True, return_graph_obs=False, parenting=-1, spatial=False, is_eval_env=False),
    "DensestSubgraph-v0": dict(weighted=False, n_choices=-1, return_graph_obs=False, is_eval_env=False, parenting=-1),
    "MaxIndependentSet-v0": dict(weighted=True, return_graph_obs=False, is_eval_env=False),
    "MulticastRouting-v0": dict(n_dests=3, weighted=True, max_distance=-1, parenting=4, is_eval_env=False),
    "DistributionCenter-v0": dict(weighted=True, max_distance=1, target_count=-1, return_graph_obs=False, is_eval_env=False, parenting=2),
    "PerishableProductDelivery-v0": dict(n_products=3, delivery_time=-1, weighted=True, return_graph_obs=False, is_eval_env=False, parenting=-1),
}


def normalize_kwargs(env_id, n_nodes, n_edges=-1, **kwargs):
    """Apply the reference constructors' defaults and asserts (same messages)."""
    if env_id not in _DEFAULTS:
        raise KeyError(f"unknown env id {env_id!r}; hot-path ids are {ENV_IDS}")
    kw = dict(_DEFAULTS[env_id])
    for k, v in kwargs.items():
        if k not in kw:
            raise TypeError(f"{env_id}.__init__() got an unexpected keyword argument {k!r}")
        kw[k] = v
    if env_id == "ShortestPath-v0":
        assert kw["parenting"] == -1, "Parenting is not available for shortest path"  # shortest_path.py:26
    if env_id == "LongestPath-v0":
        assert kw["parenting"] in [0, 1, 2, 3]  # longest_path.py:29
    if env_id == "SteinerTree-v0":
        assert kw["parenting"] == -1, "Parenting not available for this environment"  # steiner_tree.py:29
    if env_id == "TSP-v0":
        assert kw["parenting"] in [1, 2], "Parenting must be either 1 or 2"  # tsp.py:25
        if kw["spatial"]:
            assert kw["weighted"] == True, "Spatial TSP must be weighted"  # noqa: E712  tsp.py:27
    if env_id == "DensestSubgraph-v0":
        assert kw["parenting"] in [0, 1], "Parenting must be 0 or 1"  # densest_subgraph.py:28
        assert kw["weighted"] == False, "Weighted graphs not supported for this env"  # noqa: E712
    if env_id == "DistributionCenter-v0":
        assert kw["parenting"] in [1, 2]  # distribution_center.py:32
        if kw["target_count"] == -1:
            kw["target_count"] = n_nodes // 5  # distribution_center.py:42-45
    if env_id == "MulticastRouting-v0" and kw["parenting"] not in [1, 2, 3, 4]:
        raise ValueError("Invalid parenting type")  # multicast_routing.py:34-35
    if env_id == "PerishableProductDelivery-v0":
        assert kw["parenting"] in [1], "Parenting must be 1!"  # perishable_product_delivery.py:30
        assert kw["n_products"] <= 5, "Max 5 products!"  # :35
        # with any other value the reference constructor leaves dt_mn / dt_mx unset and reset() raises AttributeError (:53,92)
        assert kw["delivery_time"] == -1, "delivery_time must be -1 (the reference only runs with its computed window)"
    if env_id in ("LongestPath-v0", "DensestSubgraph-v0", "MulticastRouting-v0", "PerishableProductDelivery-v0") and n_edges == -1:
        n_edges = int((n_nodes * (n_nodes - 1) // 2) * 0.30)  # longest_path.py:41-42, multicast_routing.py:53-54
    assert n_edges != -1, f"{env_id} needs n_edges"
    if env_id == "DensestSubgraph-v0" and kw["n_choices"] == -1:
        kw["n_choices"] = float(n_nodes // np.exp(1))  # densest_subgraph.py:38-39
    kw["n_nodes"], kw["n_edges"] = int(n_nodes), int(n_edges)
    if env_id == "PerishableProductDelivery-v0":  # perishable_product_delivery.py:53-61, in numpy as the reference computes it
        avg_dist = np.log(n_nodes) / np.log(2 * n_edges / n_nodes)
        if kw["weighted"]:
            avg_dist = avg_dist * (0.3 + 1.0) / 2.0
        kw["_dt_window"] = (float(avg_dist * 0.6), float(avg_dist * 1.4))
    return kw


class GraphBatch(SimpleNamespace):
    """PyG ``Batch``-shaped view of the engine's observation slabs (x, edge_index, edge_attr, batch, ptr).
    Tensors alias engine memory and are valid until the next step()/reset(); ``to_pyg()`` upgrades to a
    real torch_geometric Batch when that package is installed."""

    def to_pyg(self):
        from torch_geometric.data import Batch, Data
        b = Batch(x=self.x, edge_index=self.edge_index, edge_attr=self.edge_attr, batch=self.batch, ptr=self.ptr)
        b._num_graphs = self.num_graphs
        return b

    def clone(self):
        return GraphBatch(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.__dict__.items()})


class _Space(SimpleNamespace):
    pass


def _gymnasium():
    """gymnasium when it is installed (it is not in the build image: everything gymnasium-specific is duck-typed otherwise)"""
    try:
        import gymnasium
        return gymnasium
    except ImportError:
        return None


_GYM = _gymnasium()
# gymnasium.vector.VectorEnv when importable, so that wrappers and trainers that check isinstance accept the engine
_VectorBase = _GYM.vector.VectorEnv if _GYM is not None and hasattr(_GYM, "vector") and hasattr(_GYM.vector, "VectorEnv") else object


class _Slabs(dict):
    """The engine's device slabs by name (what ge_buffers points at).  The scalar state of a slot is packed into
    ``slot_rec`` [B, 2] = {cost as float64 bits, head | status << 16 | aux << 24 | tstep << 32} (include/graphenvs.h
    GE_REC_*); ``t["cost"]`` / ``["head"]`` / ``["status"]`` / ``["tstep"]`` decode it on access (read-only views or copies,
    not slabs: they are not part of state_dict())."""
    DERIVED = ("cost", "head", "status", "tstep")

    def __missing__(self, key):
        rec = dict.__getitem__(self, "slot_rec")
        if key == "cost":
            return rec.view(torch.float64)[:, 0]
        packed = rec[:, 1]
        if key == "head":
            h = packed & 0xFFFF
            return torch.where(h == 0xFFFF, torch.full_like(h, -1), h).to(torch.int32)
        if key == "status":
            return ((packed >> 16) & 0xFF).to(torch.uint8)
        if key == "tstep":
            return (packed >> 32) & 0xFFFFFFFF
        raise KeyError(key)


class VectorGraphEnv(_VectorBase):
    """B independent envs of one id on one GPU.  One instance per process/GPU; no global state.  A subclass of
    gymnasium.vector.VectorEnv when gymnasium is installed (num_envs, single_*_space, *_space, metadata['autoreset_mode'])."""

    def __init__(self, env_id, num_envs, n_nodes, n_edges=-1, device="cuda", autoreset=True, obs_mode="pyg",
                 env_index_base=0, seed_stride=None, strict=False, _library=None, _views=None, node_id_base=0,
                 edge_row_stride=0, record_actions=False, copy_outputs=False, continue_streams=False, _defer_create=False, prefetch=None,
                 **kwargs):
        self.env_id = env_id
        self.kwargs = normalize_kwargs(env_id, n_nodes, n_edges, **kwargs)
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self.obs_mode = obs_mode
        self.strict = strict
        self.copy_outputs = bool(copy_outputs)
        # reset(seed=None) like the reference (shortest_path.py:49-52): every slot draws its next graph from where its previous
        # reset left its `random` / `np.random` streams (5 KiB of saved generator state per slot, written by every reset).
        # Off: reset(seed=None) moves every slot to its next episode seed (seed + seed_stride), like an autoreset does.
        self.continue_streams = bool(continue_streams)
        # return_graph_obs (shortest_path.py:94-95 and siblings): info['graph_obs'] = per-slot (nodes, edges, edge_links) views
        self.return_graph_obs = bool(self.kwargs.get("return_graph_obs", False))
        # True / "same_step": a finished slot is regenerated inside the same step(); "next_step" (gymnasium's default mode): the
        # step that ends an episode returns its final observation and the NEXT step() regenerates the slot, ignoring its action;
        # False: finished slots freeze until reset()
        assert autoreset in (True, False, "same_step", "next_step"), autoreset
        self.autoreset_mode = 2 if autoreset == "next_step" else int(bool(autoreset))
        self.autoreset = bool(autoreset)
        if _library is None:
            if self.device.type != "cuda":
                raise RuntimeError("graphenvs_amd runs on a ROCm GPU only (device='cuda'); there is no CPU path")
            if not torch.cuda.is_available():
                raise RuntimeError("graphenvs_amd: no GPU visible to torch (torch.cuda.is_available() is False)")
            self._L = _lib.load()
        else:
            self._L = _library  # CPU sanitizer harness (tests/emu), host memory
        kw = self.kwargs
        self.n, self.m = kw["n_nodes"], kw["n_edges"]
        self.seed_stride = int(seed_stride) if seed_stride is not None else self.num_envs
        self.env_index_base = int(env_index_base)
        self.cfg = _lib.GeConfig(
            _lib.ENV_TYPES[env_id], self.num_envs, self.n, self.m, int(bool(kw.get("weighted", False))),
            int(kw.get("parenting", -1)), int(kw.get("n_dests", kw.get("target_count", kw.get("n_products", 0)))), int(bool(kw.get("spatial", False))),
            int(bool(kw.get("is_eval_env", False))), self.autoreset_mode, float(kw.get("n_choices", -1)),
            self.env_index_base, self.seed_stride, int(node_id_base), int(edge_row_stride),
            float(kw["max_distance"]) if env_id == "DistributionCenter-v0" else 0.0, *kw.get("_dt_window", (0.0, 0.0)))
        lay = _lib.GeLayout()
        _lib.check(self._L, self._L.ge_get_layout(C.byref(self.cfg), C.byref(lay)), "ge_get_layout")
        self.layout = lay
        self.F, self.Fe, self.A, self.W, self.E, self.obs_len = lay.F, lay.Fe, lay.A, lay.W, lay.E, int(lay.obs_len)
        B, n, E, W, A = self.num_envs, self.n, self.E, self.W, self.A
        AW = (A + 63) // 64
        edge_env = env_id in ("SteinerTree-v0", "MulticastRouting-v0")
        T = max(2, kw.get("n_dests", 0) + 1) if edge_env else max(2, kw.get("target_count", 0), 2 * kw.get("n_products", 0))
        self.T = T
        dev = self.device
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        t = {}
        t["x"] = z((B * n, self.F), torch.float32)
        t["edge_index"] = z((2, B * E), torch.int64)
        t["edge_attr"] = z((B * E, self.Fe), torch.float32)
        t["row_ptr"] = z((B, n + 1), torch.int32)
        t["colw"] = z((B * E,), torch.int16)
        t["scode"] = z((B * E,), torch.uint8)
        t["sw64"] = z((B * E,), torch.float64) if kw.get("spatial", False) else None
        t["adj_bits"] = z((B * n, W), torch.int64)
        t["node_rec"] = z((B * n, 2), torch.int64) if W == 1 else None
        t["rev_edge"] = z((B * E,), torch.int32) if edge_env else None
        t["slot_rec"] = z((B, 2), torch.int64)
        t["terminals"] = z((B, T), torch.int32)
        t["node_bits"] = z((B, W), torch.int64)
        t["target_bits"] = z((B, W), torch.int64)
        t["counters"] = z((B, 2), torch.int32)
        t["seed"] = z((B,), torch.int32)
        t["episode"] = z((B,), torch.int64)
        t["heuristic"] = z((B,), torch.float64)
        t["mt_state"] = z((B, _lib.SEED_DEPTH, 2, 624), torch.int32)
        t["aux_bits"] = z((B,), torch.int64) if (env_id == "DistributionCenter-v0" and W == 1) else None
        t["mask"] = z((B, A), torch.uint8)
        t["mask_bits"] = z((B, AW), torch.int64)
        t["reward"] = z((B,), torch.float64)
        t["terminated"] = z((B,), torch.uint8)
        t["invalid"] = z((B,), torch.uint8)
        t["solved"] = z((B,), torch.int8)
        t["final_cost"] = z((B,), torch.float64)
        t["final_heur"] = z((B,), torch.float64)
        t["final_len"] = z((B,), torch.int32)
        t["reset_list"] = z((B,), torch.int32)
        t["reset_count"] = z(((B + 255) // 256,), torch.int32)
        t["work_list"] = z((B,), torch.int32)
        t["work_count"] = z((4,), torch.int32)
        t["feat_scratch"] = z((B, lay.feat_parts, n), torch.float64) if lay.feat_parts > 1 else None
        t["node_aux"] = z((B, n), torch.int32) if env_id == "MulticastRouting-v0" and kw["parenting"] >= 3 else None
        dc = env_id == "DistributionCenter-v0"
        t["range_bits"] = z((B * n, W), torch.int64) if dc else None
        t["cover_bits"] = z((B, W), torch.int64) if dc else None
        # where the fused policy+step launches record the actions they drew (record_actions=True); off by default: 8 bytes
        # per slot and step less to write
        t["actions_out"] = z((B,), torch.int64) if record_actions else None
        t["stream_state"] = z((B, 2, _lib.STREAM_WORDS), torch.int32) if self.continue_streams else None
        t["eval_scratch"] = z((lay.eval_scratch_bytes,), torch.uint8) if lay.eval_scratch_bytes else None  # TSP is_eval_env: Christofides work space
        t["prune_scratch"] = z((lay.prune_scratch_words,), torch.int64) if lay.prune_scratch_words else None  # node sets of the residual-graph walks (registers up to 512 nodes)
        if _views:  # slabs shared with sibling engines of other geometries (RaggedVectorEnv)
            for k, v in _views.items():
                if t[k] is None:
                    continue
                assert v.dtype == t[k].dtype and v.numel() >= (t[k].numel() if k != "edge_index" else 0), k
                t[k] = v
        self.node_id_base = int(node_id_base)
        self.t = t = _Slabs(t)
        self.bufs = _lib.GeBuffers(**{k: (v.data_ptr() if v is not None else None) for k, v in dict.items(t)})
        h = C.c_void_p()
        if not _defer_create:  # (a size class of a multi-class engine is created by RaggedVectorEnv, all classes at once)
            _lib.check(self._L, self._L.ge_create(C.byref(self.cfg), C.byref(self.bufs), C.byref(h)), "ge_create")
        self._h = h
        # episode prefetch (include/graphenvs.h, ge_attach_spares): every slot's NEXT episode is generated ahead of time into a spare
        # image, `prefetch` steps' worth of finished slots per launch, and moved in by one copy when the slot finishes.  Same outputs
        # with and without; it pays where few slots finish per step (long episodes, large graphs), where a regeneration in place
        # makes the whole step wait for the latency of a few slots.  None = the engine's choice for this env id and size, 0 = off.
        if prefetch is None:  # (the CPU sanitizer harness runs without it unless a test asks: it only doubles the work of a reset there)
            prefetch = self.default_prefetch(env_id, self.n, self.num_envs) if _library is None else 0
        self.prefetch = int(prefetch)
        if not self.autoreset or self.continue_streams:
            self.prefetch = 0
        self.spare = None
        if self.prefetch and not _defer_create:
            self._attach_spares()
        # static parts of the PyG view
        self._batch = torch.arange(B, device=dev, dtype=torch.int64).repeat_interleave(n)
        self._ptr = torch.arange(B + 1, device=dev, dtype=torch.int64) * n + int(node_id_base)
        self._truncated = torch.zeros(B, dtype=torch.bool, device=dev)
        self._actions_scratch = z((B,), torch.int64)
        self._flat = None
        self._was_reset = False
        self._streams = False  # stream_state holds the streams of a reset (continue_streams)
        self.single_action_space = _Space(n=(self.m if edge_env else n), mask_size=A)  # steiner_tree.py:43, multicast_routing.py:67
        self.single_observation_space = _Space(shape=(self.obs_len,), dtype=np.float32)
        if _GYM is not None and hasattr(_GYM, "spaces"):  # the reference's spaces (shortest_path.py:40-42), and their batched forms
            sp = _GYM.spaces
            self.single_action_space = sp.Discrete(self.m if edge_env else n)
            self.single_observation_space = sp.Box(low=-np.inf, high=np.inf, shape=(self.obs_len,), dtype=np.float32)
            self.action_space = sp.MultiDiscrete([self.m if edge_env else n] * B) if hasattr(sp, "MultiDiscrete") else None
            self.observation_space = sp.Box(low=-np.inf, high=np.inf, shape=(B, self.obs_len), dtype=np.float32)
            self.metadata = {"autoreset_mode": {0: "disabled", 1: "same_step", 2: "next_step"}[self.autoreset_mode]}

    @staticmethod
    def default_prefetch(env_id, n, num_envs):
        """refill period chosen when the caller does not say (0 = regenerate in place).  Prefetch pays when FEW slots finish per
        step -- a regeneration in place then costs the whole step the latency of those few -- and loses when thousands do (the
        reset kernels already fill the chip; the copy and the slots that finish twice inside a period are pure overhead).  The
        estimate: slots / a typical episode length under a random policy (measured, profiles/README.md round 3)."""
        length = {"ShortestPath-v0": 0.39 * n, "LongestPath-v0": 0.39 * n, "SteinerTree-v0": 0.8 * n, "MulticastRouting-v0": 0.5 * n,
                  "MaxIndependentSet-v0": 1.0 * n, "DensestSubgraph-v0": 0.36 * n, "DistributionCenter-v0": 6.0,
                  "PerishableProductDelivery-v0": 80.0 * n}.get(env_id)
        if length is None:  # TSP: every slot of a batch finishes in the same step -- one burst per episode, nothing to batch
            return 0
        per_step = num_envs / max(length, 1.0)
        if per_step >= 600:
            return 0
        # ~2 500 regenerations per refill: on BASELINE config 4 (~70 slots finish per step) the steady state gives 52 / 57 / 60 / 62.6 /
        # 58.6 / 56 M env-steps/s at periods 4 / 8 / 16 / 32 / 64 / 128 (bench.py --config c4 --prefetch P, value_200)
        # ... but never longer than about half an episode: a slot that finishes again before the refill takes the (throttled) path
        # in place, and with short episodes every refill would regenerate about every image anyway (DistributionCenter at 1 000
        # slots: 15 -> 3; ShortestPath n = 10 at 256 slots: 39 -> 2)
        period = min(128.0, max(4.0, 2560 / max(per_step, 1.0)), max(2.0, length / 2))
        return int(period)

    def _image_tensors(self, views=None):
        """a second set of the per-slot slabs (ge_spares.image); `views`: slabs given by the caller (multi-class engine)"""
        img = {}
        for k in _lib.IMAGE_FIELDS:
            live = dict.__getitem__(self.t, k)
            img[k] = None if live is None else (views[k] if views and k in views else torch.zeros_like(live))
        return img

    def _attach_spares(self):
        B, dev = self.num_envs, self.device
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        img = self._image_tensors()
        sp = dict(state=z((B,), torch.uint8), swap_list=z((B,), torch.int32), swap_count=z(((B + 255) // 256,), torch.int32),
                  refill_list=z((B,), torch.int32), refill_count=z(((B + 255) // 256,), torch.int32))
        self.spare = dict(image=img, **sp)
        rec = _lib.GeSpares(_lib.GeBuffers(**{k: (v.data_ptr() if v is not None else None) for k, v in img.items()}),
                            *(sp[k].data_ptr() for k in ("state", "swap_list", "swap_count", "refill_list", "refill_count")), self.prefetch)
        _lib.check(self._L, self._L.ge_attach_spares(self._h, C.byref(rec), None), "ge_attach_spares")

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        if self.device.type == "cuda":
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return C.c_void_p(0)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            self._L.ge_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ views
    def graph(self):
        t = self.t
        return GraphBatch(x=t["x"], edge_index=t["edge_index"], edge_attr=t["edge_attr"], batch=self._batch,
                          ptr=self._ptr, num_graphs=self.num_envs)

    def flat_obs(self):
        """utils.vectorize_graph of every slot: float32 [B, obs_len] (utils.py:87-88)."""
        if self._flat is None:
            self._flat = torch.empty((self.num_envs, self.obs_len), dtype=torch.float32, device=self.device)
        _lib.check(self._L, self._L.ge_vectorize(self._h, self._flat.data_ptr(), self._stream()), "ge_vectorize")
        return self._flat

    def _obs(self):
        return self.flat_obs() if self.obs_mode == "flat" else self.graph()

    @property
    def mask(self):
        return self.t["mask"].view(torch.bool)

    def graph_obs(self):
        """info['graph_obs'] of the reference (shortest_path.py:94-95: the live GraphInstance), batched: ``nodes`` [B, n, F],
        ``edges`` [B, E, Fe] as views of the observation slabs and ``edge_links`` [B, E, 2] local node ids."""
        t = self.t
        return SimpleNamespace(nodes=t["x"].view(self.num_envs, self.n, self.F),
                               edges=t["edge_attr"].view(self.num_envs, self.E, self.Fe), edge_links=self.edge_links())

    @staticmethod
    def _copied(out):
        def cp(v):
            if torch.is_tensor(v):
                return v.clone()
            if isinstance(v, dict):
                return {k: cp(x) for k, x in v.items()}
            if isinstance(v, (GraphBatch, SimpleNamespace)):
                return type(v)(**{k: cp(x) for k, x in v.__dict__.items()})
            return v
        return tuple(cp(v) for v in out)

    def _info(self, stepped):
        t = self.t
        info = {"mask": self.mask, "mask_bits": t["mask_bits"]}
        if self.return_graph_obs:
            info["graph_obs"] = self.graph_obs()
        if stepped:
            info.update(solved=t["solved"], solution_cost=t["final_cost"], heuristic_solution=t["final_heur"],
                        invalid_action=t["invalid"].view(torch.bool), episode_length=t["final_len"])
        return info

    # ------------------------------------------------------------------ gym surface
    def _seed_tensor(self, seed):
        B = self.num_envs
        if seed is None:
            if not self._was_reset:
                seed = 0
            else:  # next episode of every slot
                s = (self.t["seed"].cpu().numpy().view(np.uint32).astype(np.int64) + self.seed_stride) % (1 << 32)
                return torch.from_numpy(s.astype(np.uint32).view(np.int32)).to(self.device)
        if isinstance(seed, (int, np.integer)):
            s = (int(seed) + self.env_index_base + np.arange(B, dtype=np.int64)) % (1 << 32)
        else:
            s = np.asarray(seed.cpu() if torch.is_tensor(seed) else seed, dtype=np.int64).reshape(B)
            assert ((s >= 0) & (s < (1 << 32))).all(), "seeds must be in [0, 2**32) (np.random.seed requirement)"
        return torch.from_numpy(s.astype(np.uint32).view(np.int32)).to(self.device)

    def reset(self, seed=None, options=None):
        """reset(seed=s): slot i (global index g) runs the reference's reset(seed=(s+g) mod 2^32); a
        sequence/tensor gives every slot its own seed.  reset() without a seed: see ``continue_streams``."""
        if seed is None and self._streams:
            _lib.check(self._L, self._L.ge_reset_continue(self._h, self._stream()), "ge_reset_continue")
            out = (self._obs(), self._info(False))
            return self._copied(out) if self.copy_outputs else out
        seeds = self._seed_tensor(seed)
        self._seeds_keepalive = seeds
        _lib.check(self._L, self._L.ge_reset(self._h, seeds.data_ptr(), self._stream()), "ge_reset")
        self._was_reset = True
        self._streams = self.continue_streams
        out = (self._obs(), self._info(False))
        return self._copied(out) if self.copy_outputs else out

    def step(self, actions):
        """One transition of every slot.  The returned reward / terminated / info tensors (and the observation) ALIAS engine
        slabs that the next step() overwrites -- unlike a gymnasium VectorEnv, which returns fresh arrays: ``.clone()`` what a
        rollout buffer keeps across steps (or construct the env with ``copy_outputs=True``)."""
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions, dtype=np.int64))
        actions = actions.to(device=self.device, dtype=torch.int64).contiguous()
        assert actions.shape == (self.num_envs,)
        assert self._was_reset, "call reset() (or inject_state()) before step()"
        self._act_keepalive = actions
        _lib.check(self._L, self._L.ge_step(self._h, actions.data_ptr(), self._stream()), "ge_step")
        t = self.t
        if self.strict and bool(t["invalid"].any()):
            bad = torch.nonzero(t["invalid"]).flatten().tolist()
            raise AssertionError(f"invalid action in slots {bad[:8]} (the reference asserts here)")
        out = (self._obs(), t["reward"], t["terminated"].view(torch.bool), self._truncated, self._info(True))
        return self._copied(out) if self.copy_outputs else out

    # ------------------------------------------------------------------ extras
    def sample_random_actions(self, policy_seed=0, out=None):
        out = self._actions_scratch if out is None else out
        _lib.check(self._L, self._L.ge_sample_actions(self._h, int(policy_seed), out.data_ptr(), self._stream()),
                   "ge_sample_actions")
        return out

    def random_rollout(self, n_steps, policy_seed=0):
        _lib.check(self._L, self._L.ge_random_rollout(self._h, int(policy_seed), int(n_steps),
                                                      self._actions_scratch.data_ptr(), self._stream()),
                   "ge_random_rollout")

    def timed_rollout(self, n_steps, policy_seed=0):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        _lib.check(self._L, self._L.ge_timed_rollout(self._h, int(policy_seed), int(n_steps),
                                                     self._actions_scratch.data_ptr(), self._stream(),
                                                     C.byref(a), C.byref(b), C.byref(c)), "ge_timed_rollout")
        return dict(step_ms=a.value, reset_ms=b.value, policy_ms=c.value)

    def timed_step_burst_raw_ms(self, k, policy_seed=0):
        """elapsed ms between one pair of HIP events around k back-to-back (sample+)step launches (k = 0: the bare
        event pair, i.e. the measurement overhead)."""
        ms = C.c_double()
        _lib.check(self._L, self._L.ge_timed_step_burst(self._h, int(policy_seed), int(k), self._actions_scratch.data_ptr(),
                                                        self._stream(), C.byref(ms)), "ge_timed_step_burst")
        return ms.value

    def timed_step_burst(self, k, policy_seed=0):
        """k back-to-back (sample+)step launches between one pair of HIP events, no autoreset in between; returns
        the average launch duration in microseconds."""
        ms = C.c_double()
        _lib.check(self._L, self._L.ge_timed_step_burst(self._h, int(policy_seed), int(k), self._actions_scratch.data_ptr(),
                                                        self._stream(), C.byref(ms)), "ge_timed_step_burst")
        return ms.value * 1e3 / k

    def launch_floor_us(self, k=5, reps=9):
        """what an EMPTY launch of the step kernel's shape takes (median of `reps` bursts of k), microseconds"""
        ms, vals = C.c_double(), []
        empty = sorted(self.timed_step_burst_raw_ms(0) for _ in range(reps))[reps // 2]
        for _ in range(reps):
            _lib.check(self._L, self._L.ge_timed_empty_burst(self._h, int(k), self._stream(), C.byref(ms)), "ge_timed_empty_burst")
            vals.append((ms.value - empty) * 1e3 / k)
        return sorted(vals)[reps // 2]

    def inject_state(self, links, wcode, x, terminals=None, seeds=None):
        """Parity path: load post-reset states produced elsewhere (links [B,E,2] local ids, wcode [B,E] in
        {3..10}, x [B,n,F], terminals [B,T]).  ``seeds`` [B] (optional): the seed each injected episode is taken to have;
        autoreset then continues with reset(seed + seed_stride), ... -- required on an engine with autoreset that was
        never reset()."""
        dev = self.device
        links_np, wcode_np = np.asarray(links), np.asarray(wcode)
        # the kernel indexes LDS rows with these: refuse what the reference could never produce
        assert links_np.min() >= 0 and links_np.max() < self.n, "edge_links must hold local node ids in [0, n_nodes)"
        assert wcode_np.min() >= 3 and wcode_np.max() <= 10, "weight codes are k with weight k/10.0, k in 3..10"
        # the graphs are undirected: the step kernels read the weight of u -> v from either end
        Bn = links_np.shape[0]
        wmat = np.zeros((Bn, self.n, self.n), dtype=np.uint8)
        bi = np.arange(Bn)[:, None]
        wmat[bi, links_np[..., 0], links_np[..., 1]] = wcode_np
        assert np.array_equal(wmat, wmat.transpose(0, 2, 1)), "edge weights must be symmetric (weight(u,v) == weight(v,u))"
        links = torch.as_tensor(links_np, dtype=torch.int64).to(dev).contiguous()
        wcode = torch.as_tensor(np.asarray(wcode), dtype=torch.uint8).to(dev).contiguous()
        x = torch.as_tensor(np.asarray(x), dtype=torch.float32).to(dev).contiguous()
        term = None
        if terminals is not None:
            term_np = np.asarray(terminals)
            assert term_np.min() >= -1 and term_np.max() < self.n, "terminals are local node ids (-1: unused tail)"
            term = torch.as_tensor(term_np, dtype=torch.int32).to(dev).contiguous()
            assert term.shape == (self.num_envs, self.T)
        assert links.shape == (self.num_envs, self.E, 2) and wcode.shape == (self.num_envs, self.E)
        assert x.shape == (self.num_envs, self.n, self.F)
        sd = None
        if seeds is not None:
            sn = np.asarray(seeds.cpu() if torch.is_tensor(seeds) else seeds, dtype=np.int64).reshape(self.num_envs)
            assert ((sn >= 0) & (sn < (1 << 32))).all(), "seeds must be in [0, 2**32)"
            sd = torch.from_numpy(sn.astype(np.uint32).view(np.int32)).to(dev)
        self._inj_keepalive = (links, wcode, x, term, sd)
        _lib.check(self._L, self._L.ge_inject_state(self._h, links.data_ptr(), wcode.data_ptr(), x.data_ptr(),
                                                    term.data_ptr() if term is not None else None,
                                                    sd.data_ptr() if sd is not None else None, self._stream()),
                   "ge_inject_state")
        self._was_reset = True
        return self._obs(), self._info(False)

    def _quiesce(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def state_dict(self):
        """Snapshot of every engine slab: the whole state of the batch, generator states included."""
        self._quiesce()
        sd = {k: v.clone() for k, v in dict.items(self.t) if v is not None and k not in ("eval_scratch", "prune_scratch")}  # work space, not state
        if self.spare is not None:  # the images themselves are not state (they are regenerated); the marker says which invariant the generator ring obeys
            sd["_prefetch"] = torch.ones((), dtype=torch.int32)
        return sd

    def load_state_dict(self, sd):
        self._quiesce()
        # an engine with spares seeds its generator ring one episode later than one without: a snapshot of the former only restores
        # into an engine with spares (the other direction is fine)
        assert "_prefetch" not in sd or self.spare is not None, "snapshot of an engine with prefetch: restore into an engine with prefetch"
        for k, v in sd.items():
            if k != "_prefetch":
                dict.__getitem__(self.t, k).copy_(v)
        if self.spare is not None:
            self.spare["state"].zero_()  # every image is regenerated at the next opportunity
            # next-step autoreset: a slot that finished in the snapshot's last step waits for its regeneration (status 2), queued
            # either for the regeneration in place (reset_list, part of the snapshot) or -- it had a valid image -- for the swap
            # (swap_list, which is not: the images are gone).  Every such slot goes through the regeneration queue, in the layout the
            # step kernels write (a segment per 256 slots, slot order), and this engine's own swap queue is emptied.
            self.spare["swap_count"].zero_()
            B = self.num_envs
            fin = (self.t["status"] == 2).to(torch.int32)
            pad = torch.zeros(((B + 255) // 256) * 256, dtype=torch.int32, device=self.device)
            pad[:B] = fin
            blocks = pad.view(-1, 256)
            rank = torch.cumsum(blocks, dim=1) - blocks
            slot = torch.arange(pad.numel(), device=self.device, dtype=torch.int32).view(-1, 256)
            pos = (slot - slot % 256 + rank)[blocks.bool()].to(torch.int64)
            dict.__getitem__(self.t, "reset_list")[pos] = slot[blocks.bool()]
            dict.__getitem__(self.t, "reset_count").copy_(blocks.sum(dim=1).to(torch.int32))
        _lib.check(self._L, self._L.ge_mark_restored(self._h), "ge_mark_restored")
        self._was_reset = True
        self._streams = self.continue_streams and "stream_state" in sd
        self._quiesce()

    def check_device_errors(self):
        """Raise if a kernel flagged an error it could not report otherwise (synchronises): bit 0 = a G(n, m) rejection loop
        hit its round cap (generator state never seeded or corrupted); the slot's status is 4 and it stays frozen."""
        flags = int(self.t["work_count"][1].item())
        if flags:
            raise RuntimeError(f"graphenvs_amd: device error flags {flags:#x} (bit 0: graph generation gave up -- unseeded generator state?)")

    def edge_links(self):
        """[B, E, 2] local node ids (GraphInstance.edge_links of every slot)."""
        assert self.cfg.edge_row_stride in (0, self.num_envs * self.E), "use RaggedVectorEnv.edge_links for shared slabs"
        ei = self.t["edge_index"].view(2, self.num_envs, self.E)
        off = (torch.arange(self.num_envs, device=self.device, dtype=torch.int64) * self.n + self.node_id_base).view(1, -1, 1)
        return (ei - off).permute(1, 2, 0).contiguous()


def make_vec(env_id, num_envs, shards=1, **kwargs):
    """``shards`` > 1: the batch as that many independent engines on their own HIP streams (graphenvs_amd.sharded.ShardedVectorEnv)"""
    if shards is not None and int(shards) > 1:
        from .sharded import ShardedVectorEnv
        return ShardedVectorEnv(env_id, num_envs, shards=int(shards), **kwargs)
    return VectorGraphEnv(env_id, num_envs, **kwargs)
