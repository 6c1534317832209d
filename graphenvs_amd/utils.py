"""Observation codec of the reference (graph_envs/utils.py:14-29,32-73,87-88), same names and
argument meaning.  Pure tensor plumbing: splits/reshapes, no arithmetic."""
import numpy as np
import torch

from .vector_env import GraphBatch


def get_num_features():
    return 5  # feature_extraction.py:40-41


def get_env_info(env_id):
    """(node_f, edge_f, action_type) -- utils.py:32-73 (node_f already includes the 5 structural features)."""
    table = {
        "ShortestPath-v0": (2, 1, "node"), "SteinerTree-v0": (2, 2, "edge"), "MaxIndependentSet-v0": (2, 1, "node"),
        "TSP-v0": (4, 1, "node"), "DistributionCenter-v0": (5, 1, "node"), "MulticastRouting-v0": (4, 2, "edge"),
        "LongestPath-v0": (2, 1, "node"), "DensestSubgraph-v0": (1, 1, "node"),
        "PerishableProductDelivery-v0": (1 + 3 * 5, 1, "node"),
    }
    assert env_id in table, "Unknown env_id"
    node_f, edge_f, action_type = table[env_id]
    return node_f + get_num_features(), edge_f, action_type


def vectorize_graph(graph):
    """utils.py:87-88 for one GraphInstance-like object (nodes, edges, edge_links)."""
    return np.concatenate((np.asarray(graph.nodes).flatten(), np.asarray(graph.edges).flatten(),
                           np.asarray(graph.edge_links).flatten()), dtype=np.float32)


def devectorize_graph(vector, env_id, **kwargs):
    """utils.py:14-23: [bs, L] -> x [bs,n,F], edge_features [bs,2m,Fe], edge_index [bs,2m,2] (long)."""
    bs = vector.shape[0]
    node_f, edge_f, _ = get_env_info(env_id)
    p1 = kwargs["n_nodes"] * node_f
    p2 = p1 + 2 * kwargs["n_edges"] * edge_f
    x = vector[:, :p1].reshape(bs, kwargs["n_nodes"], node_f)
    edge_features = vector[:, p1:p2].reshape(bs, 2 * kwargs["n_edges"], edge_f)
    edge_index = vector[:, p2:].reshape(bs, 2 * kwargs["n_edges"], 2)
    edge_index = edge_index.long() if torch.is_tensor(edge_index) else edge_index.astype(np.int64)
    return x, edge_features, edge_index


def to_pyg_graph(x, edge_features, edge_index):
    """utils.py:26-29 without the per-graph Python loop: one batched view, node ids offset by i*n."""
    x, edge_features, edge_index = (torch.as_tensor(a) for a in (x, edge_features, edge_index))
    bs, n, _ = x.shape
    E = edge_index.shape[1]
    off = (torch.arange(bs, device=x.device, dtype=torch.int64) * n).view(bs, 1, 1)
    ei = (edge_index.long() + off).reshape(bs * E, 2).T.contiguous()
    g = GraphBatch(x=x.reshape(bs * n, -1), edge_attr=edge_features.reshape(bs * E, -1), edge_index=ei,
                   batch=torch.arange(bs, device=x.device).repeat_interleave(n),
                   ptr=torch.arange(bs + 1, device=x.device) * n, num_graphs=bs)
    try:
        return g.to_pyg()
    except ImportError:
        return g
