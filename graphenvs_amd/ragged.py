"""Ragged and mixed batches (BASELINE config 5): env instances of different (n_nodes, n_edges) — and different
env ids — stepped together.  Every reference env instance has a fixed geometry (constructor kwargs), so a ragged
batch is a set of size classes.  Each class is one uniform engine; the classes of one env id share ONE set of PyG
slabs (x, edge_index, edge_attr: variable-size CSR packing, node ids offset per class through
``ge_config.node_id_base`` / ``edge_row_stride``), so the policy sees a single ragged ``Batch``.  One launch
sequence per class ("one launch per env type")."""
import numpy as np
import torch

from .vector_env import GraphBatch, VectorGraphEnv


class RaggedVectorEnv:
    """One env id, several size classes: ``sizes = [(num_envs, n_nodes, n_edges), ...]``.  Slots are numbered class
    after class; slot g runs seed (seed + g) like a uniform engine."""

    def __init__(self, env_id, sizes, device="cuda", env_index_base=0, seed_stride=None, _library=None, **kwargs):
        self.env_id, self.device = env_id, torch.device(device)
        self.sizes = [(int(b), int(n), int(m)) for b, n, m in sizes]
        self.num_envs = sum(b for b, _, _ in self.sizes)
        stride = int(seed_stride) if seed_stride is not None else self.num_envs
        probe = VectorGraphEnv(env_id, 1, self.sizes[0][1], self.sizes[0][2], device=device, _library=_library, **kwargs)
        F, Fe = probe.F, probe.Fe
        probe.close()
        Nn = sum(b * n for b, n, _ in self.sizes)
        Ne = sum(b * 2 * m for b, _, m in self.sizes)
        dev = self.device
        self.x = torch.zeros((Nn, F), dtype=torch.float32, device=dev)
        self.edge_index = torch.zeros((2, Ne), dtype=torch.int64, device=dev)
        self.edge_attr = torch.zeros((Ne, Fe), dtype=torch.float32, device=dev)
        self.classes, self.slot_ptr = [], [0]
        noff = eoff = slot = 0
        ptr = [0]
        for b, n, m in self.sizes:
            E = 2 * m
            views = dict(x=self.x[noff:noff + b * n], edge_index=self.edge_index[0, eoff:],
                         edge_attr=self.edge_attr[eoff:eoff + b * E])
            env = VectorGraphEnv(env_id, b, n, m, device=device, env_index_base=env_index_base + slot, seed_stride=stride,
                                 _library=_library, _views=views, node_id_base=noff, edge_row_stride=Ne, **kwargs)
            self.classes.append(env)
            ptr += [noff + (i + 1) * n for i in range(b)]
            noff += b * n; eoff += b * E; slot += b
            self.slot_ptr.append(slot)
        self.ptr = torch.tensor(ptr, dtype=torch.int64, device=dev)
        self.batch = torch.repeat_interleave(torch.arange(self.num_envs, device=dev), self.ptr[1:] - self.ptr[:-1])
        self.mask_ptr = np.cumsum([0] + [c.num_envs * c.A for c in self.classes])

    def graph(self):
        return GraphBatch(x=self.x, edge_index=self.edge_index, edge_attr=self.edge_attr, batch=self.batch, ptr=self.ptr,
                          num_graphs=self.num_envs)

    def _split(self, t):
        return [t[self.slot_ptr[i]:self.slot_ptr[i + 1]] for i in range(len(self.classes))]

    def _cat(self, key):
        return torch.cat([c.t[key] for c in self.classes])

    def _info(self, stepped):
        info = {"mask": [c.mask for c in self.classes],  # ragged: one [B_c, A_c] bool view per class
                "mask_flat": torch.cat([c.mask.reshape(-1) for c in self.classes])}
        if stepped:
            info.update(solved=self._cat("solved"), solution_cost=self._cat("final_cost"),
                        heuristic_solution=self._cat("final_heur"), invalid_action=self._cat("invalid").view(torch.bool))
        return info

    def reset(self, seed=0):
        for c, lo in zip(self.classes, self.slot_ptr):
            c.reset(seed=int(seed) + 0)  # env_index_base already carries the class's first slot
        return self.graph(), self._info(False)

    def step(self, actions):
        actions = torch.as_tensor(actions).to(self.device, torch.int64)
        for c, a in zip(self.classes, self._split(actions)):
            c.step(a.contiguous())
        return (self.graph(), self._cat("reward"), self._cat("terminated").view(torch.bool),
                torch.zeros(self.num_envs, dtype=torch.bool, device=self.device), self._info(True))

    def sample_random_actions(self, policy_seed=0):
        return torch.cat([c.sample_random_actions(policy_seed).clone() for c in self.classes])

    def close(self):
        for c in self.classes:
            c.close()


class MixedVectorEnv:
    """Several env ids side by side (each a RaggedVectorEnv or VectorGraphEnv); step takes one action tensor per
    member.  Observation widths differ between ids (utils.get_env_info), so each member keeps its own PyG view."""

    def __init__(self, members):
        self.members = list(members)
        self.num_envs = sum(m.num_envs for m in self.members)

    def reset(self, seed=0):
        outs = [m.reset(seed=seed) for m in self.members]
        return [o for o, _ in outs], [i for _, i in outs]

    def step(self, actions):
        outs = [m.step(a) for m, a in zip(self.members, actions)]
        return tuple(list(col) for col in zip(*outs))

    def sample_random_actions(self, policy_seed=0):
        return [m.sample_random_actions(policy_seed) for m in self.members]

    def close(self):
        for m in self.members:
            m.close()
