"""Single-env facade with the reference's numpy surface: ``make(id, **kwargs)`` returns an object whose
``reset(seed=)`` / ``step(a)`` return what the reference env classes return (flat float32 obs of
utils.vectorize_graph, ``info['mask']`` as a numpy bool array, done-time info keys), computed by the
HIP engine with num_envs=1.  Meant for drop-in checks and small runs; throughput lives in
VectorGraphEnv."""
from collections import namedtuple

import numpy as np
import torch

from .vector_env import VectorGraphEnv

# gym.spaces.GraphInstance has exactly these fields (shortest_path.py:86)
GraphInstance = namedtuple("GraphInstance", ["nodes", "edges", "edge_links"])

# steiner_tree.py:137, max_independent_set.py:109, multicast_routing.py:200, distribution_center.py:150
_F32_REWARD = ("SteinerTree-v0", "MaxIndependentSet-v0", "MulticastRouting-v0", "DistributionCenter-v0")


class GraphEnv:
    def __init__(self, env_id, n_nodes, n_edges=-1, device="cuda", _library=None, **kwargs):
        self.env_id = env_id
        self._v = VectorGraphEnv(env_id, 1, n_nodes, n_edges, device=device, autoreset=False, obs_mode="flat",
                                 continue_streams=True, _library=_library, **kwargs)
        self.n_nodes, self.n_edges = self._v.n, self._v.m
        self.action_space = self._v.single_action_space
        self.observation_space = self._v.single_observation_space
        self._seed = None

    def _np(self, t):
        return t.detach().cpu().numpy()

    def reset(self, seed=None, options=None):
        """reset(seed=s) reproduces the reference's reset(seed=s); reset() after it continues the `random` / `np.random` streams
        where that reset left them, like the reference (shortest_path.py:49-52).  A first reset() without a seed uses seed 0."""
        obs, info = self._v.reset(seed=None if seed is None else [int(seed)])
        self._last_cost = np.float64(0.0)
        self._edges_taken, self._nodes_taken = [], set()  # longest_path.py:115, perishable_product_delivery.py:161, densest_subgraph.py:95
        out = {"mask": self._np(info["mask"])[0].copy()}
        self._graph_obs(out)
        if self.env_id == "PerishableProductDelivery-v0":  # perishable_product_delivery.py:156-158
            term, k = self._np(self._v.t["terminals"])[0], self._v.kwargs["n_products"]
            out["pickups"], out["dropoffs"] = [int(v) for v in term[:k]], [int(v) for v in term[k:2 * k]]
            out["time_left"] = float(self._np(self._v.t["target_bits"])[0].view(np.float64)[0])
        return self._np(obs)[0].copy(), out

    def _graph_obs(self, out):
        """info['graph_obs'] (shortest_path.py:94-95): a copy of the slot's GraphInstance (the reference hands out its live one)"""
        if self._v.return_graph_obs:
            g = self._v.graph_obs()
            out["graph_obs"] = GraphInstance(self._np(g.nodes)[0].copy(), self._np(g.edges)[0].copy(), self._np(g.edge_links)[0].copy())

    def step(self, action):
        head = int(self._np(self._v.t["head"])[0])
        obs, r, term, trunc, info = self._v.step(torch.tensor([int(action)], dtype=torch.int64))
        if bool(self._np(info["invalid_action"])[0]):
            raise AssertionError(f"Mask of {action} is False!")  # shortest_path.py:113
        done = bool(self._np(term)[0])
        rew = self._np(r)[0]
        rew = np.float32(rew) if self.env_id in _F32_REWARD else np.float64(rew)
        out = {"mask": self._np(info["mask"])[0].copy()}
        solved = int(self._np(info["solved"])[0])
        if solved >= 0:
            out["solved"] = bool(solved)
        if self.env_id == "MulticastRouting-v0":  # both keys on every step; the cost is -1 unless solved (multicast_routing.py:202-203,262)
            out["solution_cost"] = np.float32(self._np(self._v.t["final_cost"])[0]) if done else -1
            out["heuristic_solution"] = float(self._np(self._v.t["heuristic"])[0])
        elif self.env_id == "PerishableProductDelivery-v0":  # both keys on every step; the cost is read BEFORE the move (:212-213)
            out["solution_cost"] = self._last_cost
            self._last_cost = np.float64(self._np(self._v.t["cost"])[0])
            out["heuristic_solution"] = float(self._np(self._v.t["heuristic"])[0])
        elif done or self.env_id == "LongestPath-v0":
            cost = self._np(self._v.t["cost"])[0]
            out["solution_cost"] = np.float32(cost) if self.env_id in _F32_REWARD else np.float64(cost)
            out["heuristic_solution"] = float(self._np(self._v.t["heuristic"])[0])
        if self.env_id in ("LongestPath-v0", "PerishableProductDelivery-v0"):  # longest_path.py:160-166, perishable_product_delivery.py:209-224
            self._edges_taken.append((head, int(action)))
            out["edges_taken"] = self._edges_taken
        if self.env_id == "DensestSubgraph-v0":  # densest_subgraph.py:151,172,193
            if int(action) != self.n_nodes - 1:
                self._nodes_taken.add(int(action))
            if done:
                out["nodes_taken"] = self._nodes_taken
        return self._np(obs)[0].copy(), rew, done, False, out

    def close(self):
        self._v.close()

    @property
    def unwrapped(self):
        return self


def make(env_id, **kwargs):
    """gym.make(id, **kwargs) for the hot-path ids (graph_envs/__init__.py:9-56)."""
    return GraphEnv(env_id, **kwargs)
