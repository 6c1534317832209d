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
    """Inverse of vectorize_graph for a batch of flat observations (the reference's utils.devectorize_graph, utils.py:14-23):
    ``vector`` [B, n*F + E*Fe + 2E] -> node features [B, n, F], edge features [B, E, Fe], edge_links [B, E, 2] as int64,
    with E = 2 * n_edges directed edges.  Works on torch tensors and numpy arrays alike; the outputs are views where possible."""
    n, E = int(kwargs["n_nodes"]), 2 * int(kwargs["n_edges"])
    F, Fe, _ = get_env_info(env_id)
    sections = (n * F, E * Fe, 2 * E)              # the three pieces of the flat layout, in order
    assert vector.shape[-1] == sum(sections), "flat observation length does not match n_nodes / n_edges of this env id"
    batch = vector.shape[0]
    if torch.is_tensor(vector):
        nodes, edges, links = torch.split(vector, sections, dim=1)
        links = links.to(torch.int64)
    else:
        nodes, edges, links = np.split(np.asarray(vector), np.cumsum(sections)[:-1], axis=1)
        links = links.astype(np.int64)
    return nodes.reshape(batch, n, F), edges.reshape(batch, E, Fe), links.reshape(batch, E, 2)


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
