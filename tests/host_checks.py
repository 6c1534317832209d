"""Bodies shared by the CPU-harness tests (tests/test_emu_kernels.py) and the GPU tests (tests/test_gpu_parity.py): host-side
behaviour of the engine that does not depend on where the kernels run."""
import numpy as np
import pytest
import torch


def _extra(device, lib):
    return dict(device=device, _library=lib) if lib is not None else dict(device=device)


def check_call_order(ge, device, lib):
    """stepping an engine whose generator states were never seeded must be refused, not hang the device."""
    env = ge.VectorGraphEnv("TSP-v0", 3, 6, 12, parenting=1, **_extra(device, lib))
    zero = torch.zeros(3, dtype=torch.int64, device=device)
    with pytest.raises(AssertionError):
        env.step(zero)
    env._was_reset = True  # past the host-side assert: the C ABI refuses as well
    with pytest.raises(RuntimeError, match="ge_reset"):
        env.step(zero)
    with pytest.raises(RuntimeError, match="ge_reset"):
        env.random_rollout(1)
    env.close()


def check_inject_seeds_autoreset(ge, oracle, device, lib, mode):
    """inject_state as the first call on an engine with autoreset: refused without seeds; with seeds the episodes that follow
    are reset(seed + k * stride), and seed[] says so."""
    from inject_check import wcodes_from_edges
    B, n, m, stride = 5, 8, 14, 100
    kw = dict(n_nodes=n, n_edges=m)
    refs = [oracle.OracleEnv("ShortestPath-v0", **kw) for _ in range(B)]
    seeds = [40 + 3 * i for i in range(B)]
    for r, sd in zip(refs, seeds):
        r.reset(seed=sd)
    env = ge.VectorGraphEnv("ShortestPath-v0", B, obs_mode="flat", autoreset=mode, seed_stride=stride, **_extra(device, lib), **kw)
    links = np.stack([r.edge_links() for r in refs]); wcode = np.stack([wcodes_from_edges(r.edges()[:, 0]) for r in refs])
    x = np.stack([r.nodes() for r in refs]); terms = np.stack([r.terminals() for r in refs]).astype(np.int32)
    with pytest.raises(RuntimeError, match="seed"):
        env.inject_state(links, wcode, x, terms)
    obs, info = env.inject_state(links, wcode, x, terms, seeds=seeds)
    assert np.array_equal(obs.cpu().numpy(), np.stack([r.obs() for r in refs]))
    pending, resets = [False] * B, 0
    for k in range(30):
        a = env.sample_random_actions(policy_seed=2).clone()
        obs, rew, term, _, info = env.step(a)
        a, rew, term = a.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
        for i, r in enumerate(refs):
            if pending[i]:  # next-step mode: regenerated now, action ignored
                seeds[i] += stride; r.reset(seed=seeds[i]); pending[i] = False; resets += 1
                assert float(rew[i]) == 0 and not bool(term[i])
                continue
            _, rr, dd, _, _ = r.step(int(a[i]))
            assert rr == float(rew[i]) and dd == bool(term[i]), (k, i)
            if dd and mode is True:
                seeds[i] += stride; r.reset(seed=seeds[i]); resets += 1
            elif dd:
                pending[i] = True
        assert np.array_equal(env.flat_obs().cpu().numpy(), np.stack([r.obs() for r in refs])), k
        assert np.array_equal(info["mask"].cpu().numpy(), np.stack([r.mask() for r in refs])), k
    assert resets > B
    assert env.t["seed"].cpu().numpy().view(np.uint32).tolist() == seeds
    env.check_device_errors()
    env.close()


def check_state_dict_move(ge, device, lib):
    kw = dict(n_nodes=8, n_edges=14, obs_mode="flat", **_extra(device, lib))
    a = ge.VectorGraphEnv("ShortestPath-v0", 7, **kw)
    a.reset(seed=3); a.random_rollout(5, policy_seed=1)
    sd = a.state_dict()
    b = ge.VectorGraphEnv("ShortestPath-v0", 7, **kw)
    b.load_state_dict(sd)
    for k in range(12):
        a.random_rollout(1, policy_seed=1); b.random_rollout(1, policy_seed=1)
        a._quiesce(); b._quiesce()
        for key in ("reward", "terminated", "mask", "slot_rec", "node_bits", "episode", "seed", "x"):
            assert torch.equal(a.t[key], b.t[key]), (k, key)
    a.close(); b.close()


def check_graph_obs(ge, device, lib):
    env = ge.VectorGraphEnv("ShortestPath-v0", 4, 8, 14, obs_mode="flat", return_graph_obs=True, copy_outputs=True, **_extra(device, lib))
    obs, info = env.reset(seed=1)
    g = info["graph_obs"]
    assert g.nodes.shape == (4, 8, env.F) and g.edges.shape == (4, 28, 1) and g.edge_links.shape == (4, 28, 2)
    flat = torch.cat([g.nodes.reshape(4, -1), g.edges.reshape(4, -1), g.edge_links.reshape(4, -1).float()], dim=1)
    assert torch.equal(flat, obs)  # utils.vectorize_graph of info['graph_obs'] is the observation (reference tests/test_shortest_path.py:20-24)
    a = env.sample_random_actions(policy_seed=1).clone()
    _, rew1, _, _, info1 = env.step(a)
    keep = rew1.clone()
    env.step(env.sample_random_actions(policy_seed=1).clone())
    assert (rew1 == keep).all()  # copy_outputs: what step() returned is not overwritten by the next step
    single = ge.GraphEnv("ShortestPath-v0", n_nodes=8, n_edges=14, return_graph_obs=True, **_extra(device, lib))
    o, inf = single.reset(seed=2)  # slot 1 of the batch above ran seed 1 + 1
    assert np.array_equal(inf["graph_obs"].nodes, g.nodes[1].cpu().numpy()) and np.array_equal(inf["graph_obs"].edge_links, g.edge_links[1].cpu().numpy())
    from graphenvs_amd import utils
    assert np.array_equal(utils.vectorize_graph(inf["graph_obs"]), o)
    env.close(); single.close()


def check_continue_streams(ge, oracle, device, lib):
    """reset() without a seed in a batch with autoreset (continue_streams=True): every slot draws its next graph from where its LAST
    regeneration -- whichever seeded episode that was -- left the two streams (shortest_path.py:49-52), and the autoreset episodes
    after it are seeded as if that reset had been one (the ring entry it skipped is refilled)."""
    B, n, m, stride = 6, 6, 9, 50
    kw = dict(n_nodes=n, n_edges=m)
    for env_id, extra in (("ShortestPath-v0", {}), ("SteinerTree-v0", dict(n_dests=2))):
        refs = [oracle.OracleEnv(env_id, **kw, **extra) for _ in range(B)]
        seeds = [7 + i for i in range(B)]
        env = ge.VectorGraphEnv(env_id, B, obs_mode="flat", autoreset=True, seed_stride=stride, continue_streams=True,
                                **_extra(device, lib), **kw, **extra)
        with pytest.raises(RuntimeError, match="ge_reset"):  # nothing to continue yet: the C ABI refuses
            env._streams = True; env.reset()
        env._streams = False
        env.reset(seed=seeds)
        for r, sd in zip(refs, seeds):
            r.reset(seed=sd)
        resets = 0

        def roll(K):
            nonlocal resets
            for k in range(K):
                a = env.sample_random_actions(policy_seed=3).clone()
                obs, rew, term, _, info = env.step(a)
                a, rew, term = a.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
                for i, r in enumerate(refs):
                    _, rr, dd, _, _ = r.step(int(a[i]))
                    assert rr == float(rew[i]) and dd == bool(term[i]), (env_id, k, i)
                    if dd:
                        seeds[i] += stride; r.reset(seed=seeds[i]); resets += 1
                assert np.array_equal(env.flat_obs().cpu().numpy(), np.stack([r.obs() for r in refs])), (env_id, k)
                assert np.array_equal(info["mask"].cpu().numpy(), np.stack([r.mask() for r in refs])), (env_id, k)

        roll(12)
        for rnd in range(2):
            obs, info = env.reset()  # continues every slot's streams
            for i, r in enumerate(refs):
                r.reset(); seeds[i] += stride  # the bookkeeping moves on one episode
            assert np.array_equal(obs.cpu().numpy(), np.stack([r.obs() for r in refs])), (env_id, rnd)
            assert np.array_equal(info["mask"].cpu().numpy(), np.stack([r.mask() for r in refs])), (env_id, rnd)
            roll(25)  # more than GE_SEED_DEPTH autoresets per slot: round the ring, past the refilled entry
        assert resets > 4 * B
        assert env.t["seed"].cpu().numpy().view(np.uint32).tolist() == seeds
        env.check_device_errors()
        env.close()


_LIVE = ("reward", "terminated", "mask", "mask_bits", "slot_rec", "node_bits", "episode", "seed", "x", "edge_index", "edge_attr", "heuristic",
         "final_cost", "final_heur", "final_len", "solved")


def _same_live(a, b, where):
    a._quiesce(); b._quiesce()
    for key in _LIVE:
        assert torch.equal(a.t[key], b.t[key]), (where, key)


def check_prefetch_reset_mid_rollout(ge, device, lib, mode):
    """reset() / inject-free restart in the middle of a rollout of an engine with spares.  With next-step autoreset the
    swap queue of the last step is consumed at the START of the next ge_step: a reset in between must empty it, or the image of
    episode 1 is copied over the slot that was just reset.  Compared with the engine without spares, slab by slab."""
    kw = dict(n_nodes=6, n_edges=9, obs_mode="flat", autoreset=mode, seed_stride=100, **_extra(device, lib))
    a = ge.VectorGraphEnv("ShortestPath-v0", 8, prefetch=3, **kw)
    b = ge.VectorGraphEnv("ShortestPath-v0", 8, prefetch=0, **kw)
    a.reset(seed=5); b.reset(seed=5)
    hit = False
    for k in range(12):  # step until some slot has just terminated with a valid image, then reset
        valid = a.spare["state"].clone()
        a.random_rollout(1, policy_seed=2); b.random_rollout(1, policy_seed=2)
        _same_live(a, b, ("before", k))
        if bool(((a.t["terminated"] != 0) & (valid.to(a.t["terminated"].device) != 0)).any()) and k >= 2:
            hit = True
            break
    assert hit
    a.reset(seed=77); b.reset(seed=77)
    _same_live(a, b, "reset")
    for k in range(10):
        a.random_rollout(1, policy_seed=3); b.random_rollout(1, policy_seed=3)
        _same_live(a, b, ("after", k))
    assert int(a.t["episode"].sum()) > 0
    a.close(); b.close()


def check_prefetch_state_dict_next_step(ge, device, lib):
    """a snapshot of an engine with spares in next-step mode, taken right after a step in which slots finished with a valid
    image (they wait in the swap queue, which is not state): restored into a fresh engine with spares -- whose own swap queue holds
    entries of an earlier rollout -- every such slot is regenerated, and nothing else is."""
    kw = dict(n_nodes=6, n_edges=9, obs_mode="flat", autoreset="next_step", seed_stride=100, **_extra(device, lib))
    a = ge.VectorGraphEnv("ShortestPath-v0", 8, prefetch=3, **kw)
    ref = ge.VectorGraphEnv("ShortestPath-v0", 8, prefetch=0, **kw)
    a.reset(seed=5); ref.reset(seed=5)
    hit = False
    for k in range(12):
        valid = a.spare["state"].clone()
        a.random_rollout(1, policy_seed=2); ref.random_rollout(1, policy_seed=2)
        if bool(((a.t["terminated"] != 0) & (valid.to(a.t["terminated"].device) != 0)).any()) and k >= 2:
            hit = True
            break
    assert hit
    sd = a.state_dict()
    b = ge.VectorGraphEnv("ShortestPath-v0", 8, prefetch=3, **kw)
    b.reset(seed=900); b.random_rollout(7, policy_seed=9)  # leaves stale entries in its own queues
    b.load_state_dict(sd)
    for k in range(12):
        b.random_rollout(1, policy_seed=2); ref.random_rollout(1, policy_seed=2)
        _same_live(b, ref, k)
    assert int(b.t["episode"].min()) > 0
    a.close(); b.close(); ref.close()


def check_shards_equal_one_engine(ge, device, env_id="ShortestPath-v0", kw=None, B=10, K=25, library=None):
    """the batch as 1 and as 3 independent engines (make_vec(shards=)): the same slots, bit for bit, after a reset and a rollout of the
    device policy with autoreset -- per-slot tensors gathered in slot order, the observations shard after shard"""
    import torch
    kw = dict(kw or dict(n_nodes=10, n_edges=20))
    extra = dict(device=device, _library=library) if library is not None else dict(device=device)
    one = ge.make_vec(env_id, B, prefetch=0, **extra, **kw)
    many = ge.make_vec(env_id, B, shards=3, prefetch=0, **extra, **kw)
    assert [m.num_envs for m in many.members] == [B // 3 + (1 if k < B % 3 else 0) for k in range(3)] and many.num_envs == B
    one.reset(seed=7); many.reset(seed=7)
    for round_ in range(2):
        for key in ("episode", "tstep", "seed", "reward", "terminated", "cost", "solved"):
            assert torch.equal(one.t[key], many.gather(key)), (round_, key)
        assert torch.equal(one.t["x"], torch.cat([m.t["x"] for m in many.members])), round_
        assert torch.equal(one.t["mask"], torch.cat([m.t["mask"] for m in many.members])), round_
        one.random_rollout(K, policy_seed=5); many.random_rollout(K, policy_seed=5)
    assert int(one.t["episode"].sum()) > 0
    # step(): one action tensor per shard in, one entry per shard out (streams forked and joined per call)
    a1 = one.sample_random_actions(policy_seed=9).clone()
    am = many.sample_random_actions(policy_seed=9)
    assert torch.equal(a1, torch.cat(am))
    o1 = one.step(a1)
    om = many.step([a.clone() for a in am])
    assert torch.equal(o1[1], torch.cat(om[1])) and torch.equal(o1[2], torch.cat(om[2]))  # reward, terminated
    assert torch.equal(one.t["x"], torch.cat([m.t["x"] for m in many.members]))
    many.check_device_errors()
    one.close(); many.close()
