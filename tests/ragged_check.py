"""Shared body of the ragged / mixed batch parity test (BASELINE config 5 shape, reduced): every slot of every size
class is replayed on the CPU oracle."""
import numpy as np
import torch


def config5_specs(n_sizes=66, slots_per_class=2, seed=5):
    """BASELINE config 5: {ShortestPath, MaxIndependentSet (= the README's MinVertexCover), DensestSubgraph}, n ~ U{32..512}, m = 3n:
    n_sizes distinct sizes per env id, the extremes and the 64 / 65 boundary of the feature fast path always among them"""
    rng = np.random.default_rng(seed)
    specs = []
    for eid, extra in (("ShortestPath-v0", {}), ("MaxIndependentSet-v0", {}), ("DensestSubgraph-v0", dict(parenting=1))):
        ns = {32, 64, 65, 512}
        while len(ns) < n_sizes:
            ns.add(int(rng.integers(32, 513)))
        order = [int(v) for v in rng.permutation(sorted(ns))]  # classes in no particular order: small and large graphs interleaved
        specs.append((eid, [(slots_per_class, n, 3 * n) for n in order], extra))
    return specs


def check_ragged_mixed(ge, oracle, device, library=None, steps=25, specs=None):
    kw = dict(device=device, _library=library) if library is not None else dict(device=device)
    specs = specs or [("ShortestPath-v0", [(3, 32, 96), (2, 45, 135), (2, 64, 192), (1, 130, 390)], {}),
                      ("MaxIndependentSet-v0", [(2, 32, 96), (2, 70, 210)], {}),
                      ("DensestSubgraph-v0", [(2, 33, 96), (2, 90, 270)], dict(parenting=1))]
    members = [ge.RaggedVectorEnv(eid, sizes, **kw, **extra) for eid, sizes, extra in specs]
    mixed = ge.MixedVectorEnv(members)
    graphs, infos = mixed.reset(seed=11)
    refs = []
    for (eid, sizes, extra), member, g in zip(specs, members, graphs):
        slot, rs = 0, []
        for (b, n, m), cls in zip(sizes, member.classes):
            for i in range(b):
                r = oracle.OracleEnv(eid, n_nodes=n, n_edges=m, **extra)
                r.reset(seed=11 + slot)
                lo, hi = int(member.ptr[slot]), int(member.ptr[slot + 1])
                assert hi - lo == n
                assert np.array_equal(g.x[lo:hi].cpu().numpy(), r.nodes())
                sel = (g.batch[g.edge_index[0]] == slot)
                ei = g.edge_index[:, sel].cpu().numpy() - lo
                assert np.array_equal(ei.T, r.edge_links())
                assert np.array_equal(g.edge_attr[sel].cpu().numpy(), r.edges())
                assert np.array_equal(cls.mask[i].cpu().numpy(), r.mask())
                rs.append((r, n, m, slot))
                slot += 1
        assert g.x.shape[0] == int(member.ptr[-1]) and int(g.batch[-1]) == member.num_envs - 1
        refs.append(rs)
    episodes = 0
    tcount = [[0] * len(rs) for rs in refs]
    for k in range(steps):
        acts = mixed.sample_random_actions(policy_seed=5)
        obs, rew, term, trunc, info = mixed.step(acts)
        for mi, ((eid, sizes, extra), member, rs) in enumerate(zip(specs, members, refs)):
            a = acts[mi].cpu().numpy(); rw = rew[mi].cpu().numpy(); tm = term[mi].cpu().numpy()
            flat = info[mi]["mask_flat"].cpu().numpy()
            off = 0
            xs = None
            for j, (r, n, m, slot) in enumerate(rs):
                want_a = oracle.policy_pick(r.mask(), 5, slot, tcount[mi][j])
                assert int(a[j]) == want_a, (eid, j, k)
                _, rr, dd, _, _ = r.step(int(a[j]))
                tcount[mi][j] += 1
                assert rr == rw[j] and dd == bool(tm[j]), (eid, j, k)
                if dd:
                    episodes += 1
                    ep = tcount[mi][j]  # placeholder to keep the line short
                    r.reset(seed=(11 + slot + member.num_envs * _episode(member, j)) % 2**32)
                assert np.array_equal(flat[off:off + r.A], r.mask()), (eid, j, k)
                off += r.A
                if dd or k == steps - 1:  # the regenerated observation (all five structural columns) of every autoreset, and everything at the end
                    if xs is None:
                        g = obs[mi]
                        xs, ei, ea = g.x.cpu().numpy(), g.edge_index.cpu().numpy(), g.edge_attr.cpu().numpy()
                        ebounds = np.searchsorted(ei[0], member.ptr.cpu().numpy())  # edge_index[0] is non-decreasing: slot after slot
                    lo, hi = int(member.ptr[slot]), int(member.ptr[slot + 1])
                    assert np.array_equal(xs[lo:hi], r.nodes()), (eid, j, k)
                    e0, e1 = int(ebounds[slot]), int(ebounds[slot + 1])
                    assert np.array_equal(ei[:, e0:e1].T - lo, r.edge_links()), (eid, j, k)
                    assert np.array_equal(ea[e0:e1], r.edges()), (eid, j, k)
    assert episodes > 0
    mixed.close()


def _episode(member, j):
    lo = 0
    for c in member.classes:
        if j < lo + c.num_envs:
            return int(c.t["episode"][j - lo])
        lo += c.num_envs
    raise IndexError(j)
