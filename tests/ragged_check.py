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


def check_config5_full_size(ge, oracle, device, per_id=16384, sampled=16, steps=40, seed=11):
    """BASELINE config 5 at its real size -- per env id `per_id` slots whose n is drawn from U{32..512} (every size occurs: 481
    classes), m = 3n, the batch bench.py times -- : invariants over EVERY slot after every step (transition counters, mask bytes ==
    mask bits, episode counters against the terminations seen) and `sampled` slots per id replayed on the oracle step by step,
    policy draws and every regenerated observation included."""
    rng = np.random.default_rng(0)
    pick = np.random.default_rng(1)
    members, plans = [], []
    for eid, extra in (("ShortestPath-v0", {}), ("MaxIndependentSet-v0", {}), ("DensestSubgraph-v0", dict(parenting=1))):
        ns = rng.integers(32, 513, per_id)
        sizes = [(int((ns == n).sum()), int(n), 3 * int(n)) for n in np.unique(ns)]
        m = ge.RaggedVectorEnv(eid, sizes, device=device, **extra)
        assert m.num_envs == per_id and len(m.classes) == len(sizes)
        members.append(m)
        # sampled global slots: the extremes of the size range always among them
        starts = np.cumsum([0] + [b for b, _, _ in sizes])
        slots = sorted(set([0, per_id - 1] + pick.integers(0, per_id, sampled - 2).tolist()))
        plan = []
        for g in slots:
            c = int(np.searchsorted(starts, g, side="right") - 1)
            plan.append(dict(slot=g, cls=c, i=g - int(starts[c]), n=sizes[c][1], m=sizes[c][2],
                             ref=oracle.OracleEnv(eid, n_nodes=sizes[c][1], n_edges=sizes[c][2], **extra), t=0, ep=0))
        plans.append((eid, extra, plan))
    mixed = ge.MixedVectorEnv(members)
    graphs, infos = mixed.reset(seed=seed)

    def check_obs(member, g, p, what):
        lo, hi = int(member.ptr[p["slot"]]), int(member.ptr[p["slot"] + 1])
        assert hi - lo == p["n"], what
        assert np.array_equal(g.x[lo:hi].cpu().numpy(), p["ref"].nodes()), what + " (nodes)"
        cls = member.classes[p["cls"]]
        E = 2 * p["m"]
        # the class's view of the shared [2, Ne] slab starts at its first edge of row 0
        first = (cls.t["edge_index"].data_ptr() - member.edge_index.data_ptr()) // 8 + p["i"] * E
        rows = member.edge_index[:, first:first + E].cpu().numpy() - lo
        assert np.array_equal(rows.T, p["ref"].edge_links()), what + " (edge links)"
        assert np.array_equal(member.edge_attr[first:first + E].cpu().numpy(), p["ref"].edges()), what + " (edge attributes)"

    for (eid, extra, plan), member, g in zip(plans, members, graphs):
        for p in plan:
            p["ref"].reset(seed=seed + p["slot"])
            check_obs(member, g, p, f"{eid} slot {p['slot']} after reset")
            assert np.array_equal(member.classes[p["cls"]].mask[p["i"]].cpu().numpy(), p["ref"].mask())
    seen_terms = [0, 0, 0]
    for k in range(steps):
        acts = mixed.sample_random_actions(policy_seed=5)
        obs, rew, term, trunc, info = mixed.step(acts)
        for mi, ((eid, extra, plan), member) in enumerate(zip(plans, members)):
            a, rw, tm = acts[mi], rew[mi], term[mi]
            seen_terms[mi] += int(tm.sum())
            # ---- every slot
            packed = member.g["slot_rec"][:, 1]
            assert int(((packed >> 32) & 0xFFFFFFFF).sum()) == member.num_envs * (k + 1), (eid, k)  # transitions survive regenerations
            assert int(member.g["episode"].sum()) == seen_terms[mi], (eid, k)                      # same-step autoreset: one new episode per termination
            assert int(((packed >> 16) & 0xFF).max()) == 0, (eid, k)                                # nobody is frozen or pending
            for cls in member.classes[:: max(1, len(member.classes) // 40)]:                        # mask bytes == mask bits (a spread of classes)
                A = cls.A
                bits = cls.t["mask_bits"]
                unpacked = ((bits.unsqueeze(-1) >> torch.arange(64, device=bits.device)) & 1).reshape(cls.num_envs, -1)[:, :A].to(torch.uint8)
                assert torch.equal(unpacked, cls.t["mask"]), (eid, cls.n, k)
            # ---- the sampled slots, on the oracle
            a_np, rw_np, tm_np = a.cpu().numpy(), rw.cpu().numpy(), tm.cpu().numpy()
            for p in plan:
                g = p["slot"]
                want_a = oracle.policy_pick(p["ref"].mask(), 5, g, p["t"])
                assert int(a_np[g]) == want_a, (eid, g, k)
                _, rr, dd, _, _ = p["ref"].step(int(a_np[g]))
                p["t"] += 1
                assert rr == rw_np[g] and dd == bool(tm_np[g]), (eid, g, k)
                if dd:
                    p["ep"] += 1
                    p["ref"].reset(seed=(seed + g + member.num_envs * p["ep"]) % 2**32)
                    check_obs(member, obs[mi], p, f"{eid} slot {g} regenerated at step {k}")
                assert np.array_equal(member.classes[p["cls"]].mask[p["i"]].cpu().numpy(), p["ref"].mask()), (eid, g, k)
    for (eid, extra, plan), member, g in zip(plans, members, obs):
        for p in plan:
            check_obs(member, g, p, f"{eid} slot {p['slot']} at the end")
        assert sum(p["ep"] for p in plan) > 0 or eid == "MaxIndependentSet-v0", eid  # (MaxIndependentSet episodes last n steps)
    for m in members:
        for c in m.classes[:1]:
            c.check_device_errors()
    mixed.close()
