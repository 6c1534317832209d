"""GPU (MI355X): libgraphenvs_hip.so through the C ABI against (1) the golden vectors captured from the
reference and (2) the CPU oracle on seeded batches, plus size-independent properties at the full
BASELINE batch sizes.  Bit-exact for masks/done/indices/obs; rewards compared exactly (f64)."""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def _ge():
    import graphenvs_amd as ge
    return ge


UNBUILT = ("is not built yet",)


@pytest.mark.parametrize("name", gu.case_names())
def test_hip_engine_replays_reference_golden(name):
    ge = _ge()
    case = gu.load_case(name)
    try:
        st = gu.replay_case(case, lambda env_id, **kw: ge.make(env_id, **kw))
    except RuntimeError as e:
        if any(u in str(e) for u in UNBUILT):
            pytest.skip(str(e))
        raise
    assert st["resets"] > 0


CASES = [
    ("ShortestPath-v0", dict(n_nodes=64, n_edges=192, is_eval_env=True), 256, 60),
    ("ShortestPath-v0", dict(n_nodes=10, n_edges=20, weighted=False), 64, 30),
    ("ShortestPath-v0", dict(n_nodes=28, n_edges=32, is_eval_env=True), 48, 30),  # deep sparse graphs: feature fallback list, many rejections
    ("LongestPath-v0", dict(n_nodes=20, n_edges=50, parenting=1, is_eval_env=True), 64, 40),
    ("SteinerTree-v0", dict(n_nodes=40, n_edges=100, n_dests=5), 64, 80),
    ("SteinerTree-v0", dict(n_nodes=30, n_edges=80, n_dests=29, is_eval_env=True), 32, 60),
    ("TSP-v0", dict(n_nodes=16, n_edges=120, parenting=1), 32, 40),
    ("TSP-v0", dict(n_nodes=20, n_edges=60, parenting=1), 32, 50),
    ("TSP-v0", dict(n_nodes=14, n_edges=40, parenting=2), 32, 40),
    ("TSP-v0", dict(n_nodes=70, n_edges=300, parenting=1, spatial=True), 16, 80),
    ("TSP-v0", dict(n_nodes=20, n_edges=190, parenting=1, spatial=True), 16, 30),
    ("LongestPath-v0", dict(n_nodes=24, n_edges=50, parenting=2), 48, 40),
    ("LongestPath-v0", dict(n_nodes=100, n_edges=300, parenting=2), 16, 60),
    ("ShortestPath-v0", dict(n_nodes=130, n_edges=400, is_eval_env=True), 16, 60),
    ("DensestSubgraph-v0", dict(n_nodes=64, n_edges=192, parenting=1), 64, 60),
    ("DensestSubgraph-v0", dict(n_nodes=30, n_edges=60, parenting=0, is_eval_env=True), 64, 40),
    ("MaxIndependentSet-v0", dict(n_nodes=70, n_edges=200), 64, 150),
    # own baselines: engine == oracle exactly; oracle vs reference by bounds (test_oracle_golden.py)
    ("SteinerTree-v0", dict(n_nodes=40, n_edges=100, n_dests=5, is_eval_env=True), 32, 60),
    ("SteinerTree-v0", dict(n_nodes=70, n_edges=200, n_dests=30, is_eval_env=True), 16, 80),
    ("TSP-v0", dict(n_nodes=16, n_edges=60, parenting=1, is_eval_env=True), 32, 40),
    ("TSP-v0", dict(n_nodes=70, n_edges=400, parenting=1, spatial=True, is_eval_env=True), 16, 80),
    ("MaxIndependentSet-v0", dict(n_nodes=130, n_edges=500, weighted=False, is_eval_env=True), 16, 140),
    ("MulticastRouting-v0", dict(n_nodes=64, n_edges=192, n_dests=5), 64, 80),
    ("MulticastRouting-v0", dict(n_nodes=40, n_edges=100, n_dests=4, parenting=2), 48, 60),
    ("MulticastRouting-v0", dict(n_nodes=30, n_edges=70, n_dests=6, parenting=3, is_eval_env=True), 48, 60),
    ("MulticastRouting-v0", dict(n_nodes=20, n_edges=50, parenting=1), 32, 30),
    ("MulticastRouting-v0", dict(n_nodes=150, n_edges=500, n_dests=8, is_eval_env=True), 16, 80),
    ("DistributionCenter-v0", dict(n_nodes=64, n_edges=192), 64, 40),
    ("DistributionCenter-v0", dict(n_nodes=30, n_edges=70, parenting=1, max_distance=0.8, target_count=10), 48, 40),
    ("DistributionCenter-v0", dict(n_nodes=100, n_edges=260, weighted=False, is_eval_env=True), 16, 60),
    ("PerishableProductDelivery-v0", dict(n_nodes=20, n_edges=50, parenting=1), 32, 150),
    ("PerishableProductDelivery-v0", dict(n_nodes=8, n_edges=9, parenting=1), 64, 200),  # sparse: placement retries
    ("PerishableProductDelivery-v0", dict(n_nodes=30, n_edges=90, n_products=5, parenting=1, is_eval_env=True), 16, 600),
    ("PerishableProductDelivery-v0", dict(n_nodes=100, n_edges=300, n_products=1, weighted=False, parenting=1), 16, 500),
    # above 512 nodes: the residual-graph walks of parenting >= 2 keep their node sets in prune_scratch; spatial TSP
    ("LongestPath-v0", dict(n_nodes=560, n_edges=1500, parenting=2), 6, 60),
    ("TSP-v0", dict(n_nodes=530, n_edges=1800, parenting=2), 4, 40),
    ("TSP-v0", dict(n_nodes=540, n_edges=1700, parenting=1, spatial=True), 4, 40),
    # the generic feature kernel at its wave counts above 256 nodes (one workgroup of 16 / 12 / 9 waves per CU), the late
    # numpy draws of weighted graphs above 256 nodes, and a dense graph whose rows take many quads
    ("ShortestPath-v0", dict(n_nodes=300, n_edges=900), 6, 12),
    ("ShortestPath-v0", dict(n_nodes=400, n_edges=1200, is_eval_env=True), 4, 10),
    ("DensestSubgraph-v0", dict(n_nodes=512, n_edges=1536, parenting=1), 4, 8),
    ("MaxIndependentSet-v0", dict(n_nodes=257, n_edges=4000), 4, 8),
]


@pytest.mark.parametrize("env_id,kw,B,K", CASES)
def test_batched_autoreset_rollout_matches_oracle(env_id, kw, B, K):
    """B slots, K vector steps with the device policy and same-step autoreset; every slot is replayed on
    the CPU oracle with reset(seed=(s0 + k*stride) mod 2^32) per episode."""
    import oracle
    ge = _ge()
    stride, base, s0 = 7919, 11, 2**32 - 40  # seeds wrap around 2^32 inside the batch
    env = ge.make_vec(env_id, B, obs_mode="flat", seed_stride=stride, env_index_base=base, **kw)
    obs, info = env.reset(seed=s0)
    refs = [oracle.OracleEnv(env_id, **kw) for _ in range(B)]
    seeds = [(s0 + base + i) % 2**32 for i in range(B)]
    want = np.stack([r.reset(seed=s)[0] for r, s in zip(refs, seeds)])
    assert np.array_equal(obs.cpu().numpy(), want)
    assert np.array_equal(info["mask"].cpu().numpy(), np.stack([r.mask() for r in refs]))
    tcount = [0] * B
    episodes = 0
    for k in range(K):
        a = env.sample_random_actions(policy_seed=77).clone().cpu().numpy()
        assert a.tolist() == [oracle.policy_pick(r.mask(), 77, base + i, tcount[i]) for i, r in enumerate(refs)]
        obs, rew, term, trunc, info = env.step(torch.from_numpy(a).cuda())
        rew, term = rew.cpu().numpy(), term.cpu().numpy()
        solved, fc, fh = info["solved"].cpu().numpy(), info["solution_cost"].cpu().numpy(), info["heuristic_solution"].cpu().numpy()
        assert not trunc.any() and not info["invalid_action"].any()
        for i, r in enumerate(refs):
            _, rr, dd, _, inf = r.step(int(a[i]))
            tcount[i] += 1
            assert rr == rew[i], (k, i, rr, rew[i])
            assert dd == bool(term[i]), (k, i)
            assert int(solved[i]) == (int(inf["solved"]) if "solved" in inf else -1), (k, i)
            if dd:
                assert fc[i] == inf["solution_cost"], (k, i)
                if not np.isnan(inf["heuristic_solution"]):
                    assert fh[i] == inf["heuristic_solution"], (k, i)
                episodes += 1
                seeds[i] = (seeds[i] + stride) % 2**32
                r.reset(seed=seeds[i])
        assert np.array_equal(info["mask"].cpu().numpy(), np.stack([r.mask() for r in refs])), k
        if k % 10 == 9 or k == K - 1:
            assert np.array_equal(env.flat_obs().cpu().numpy(), np.stack([r.obs() for r in refs])), k
    assert episodes > 0 or kw["n_nodes"] > 256  # (a tour of 530 nodes, or a walk over 300, does not end within the steps replayed here)


def test_invalid_actions_are_flagged_and_leave_state_untouched():
    ge = _ge()
    env = ge.make_vec("ShortestPath-v0", 8, n_nodes=10, n_edges=20, autoreset=False, obs_mode="flat")
    obs, info = env.reset(seed=3)
    before = {k: v.clone() for k, v in env.state_dict().items() if k in ("x", "slot_rec", "node_bits", "mask", "mask_bits")}
    mask = info["mask"].cpu().numpy()
    bad = torch.tensor([int(np.nonzero(~mask[i])[0][0]) for i in range(8)], device="cuda")
    bad[1] = 10; bad[2] = -5; bad[3] = -1  # out of range, negative, explicit no-op
    obs, rew, term, trunc, info = env.step(bad)
    inv = info["invalid_action"].cpu().numpy()
    assert inv.tolist() == [True, True, True, False, True, True, True, True]
    assert not term.any() and (rew == 0).all()
    after = env.state_dict()
    for k, v in before.items():
        assert torch.equal(v, after[k]), k
    env2 = ge.make_vec("ShortestPath-v0", 8, n_nodes=10, n_edges=20, autoreset=False, strict=True)
    env2.reset(seed=3)
    with pytest.raises(AssertionError):
        env2.step(bad)


def test_autoreset_off_freezes_finished_slots():
    ge = _ge()
    env = ge.make_vec("MaxIndependentSet-v0", 4, n_nodes=6, n_edges=8, autoreset=False)
    env.reset(seed=0)
    for k in range(6):
        obs, rew, term, trunc, info = env.step(torch.full((4,), k, device="cuda"))
    assert term.all() and (env.t["status"] == 1).all()
    x = env.t["x"].clone()
    obs, rew, term, trunc, info = env.step(torch.zeros(4, dtype=torch.int64, device="cuda"))
    assert not term.any() and (rew == 0).all() and torch.equal(x, env.t["x"])


@pytest.mark.parametrize("env_id,kw", [("ShortestPath-v0", dict(n_nodes=24, n_edges=60)),
                                       ("SteinerTree-v0", dict(n_nodes=130, n_edges=420, n_dests=6)),   # generic feature path: partial sums
                                       ("DistributionCenter-v0", dict(n_nodes=40, n_edges=110))])
def test_shards_are_invariant_to_the_partition(env_id, kw):
    """the RNG is keyed by the global slot index, so a 2-way shard equals the unsharded batch (bit for bit, the
    float64 feature sums included: their order depends on the geometry, never on the batch size)."""
    ge = _ge()
    kw = dict(obs_mode="flat", **kw)
    whole = ge.make_vec(env_id, 64, seed_stride=64, **kw)
    lo = ge.make_vec(env_id, 32, seed_stride=64, env_index_base=0, **kw)
    hi = ge.make_vec(env_id, 32, seed_stride=64, env_index_base=32, **kw)
    for e in (whole, lo, hi):
        e.reset(seed=123)
    for k in range(40):
        for e in (whole, lo, hi):
            e.step(e.sample_random_actions(policy_seed=5))
        assert torch.equal(whole.t["reward"], torch.cat([lo.t["reward"], hi.t["reward"]]))
        assert torch.equal(whole.t["terminated"], torch.cat([lo.t["terminated"], hi.t["terminated"]]))
    assert torch.equal(whole.flat_obs(), torch.cat([lo.flat_obs(), hi.flat_obs()]))
    assert torch.equal(whole.t["episode"], torch.cat([lo.t["episode"], hi.t["episode"]]))


def test_pyg_view_matches_flat_codec():
    ge = _ge()
    from graphenvs_amd import utils
    B, n, m = 16, 12, 30
    env = ge.make_vec("SteinerTree-v0", B, n_nodes=n, n_edges=m, n_dests=3)
    g, info = env.reset(seed=1)
    flat = env.flat_obs()
    x, ef, ei = utils.devectorize_graph(flat, "SteinerTree-v0", n_nodes=n, n_edges=m)
    pg = utils.to_pyg_graph(x, ef, ei)
    assert torch.equal(pg.x, g.x) and torch.equal(pg.edge_attr, g.edge_attr) and torch.equal(pg.edge_index, g.edge_index)
    assert torch.equal(pg.batch, g.batch) and torch.equal(pg.ptr, g.ptr)
    assert torch.equal(env.edge_links(), ei)


@pytest.mark.parametrize("env_id,kw,B,K", [
    ("ShortestPath-v0", dict(n_nodes=64, n_edges=192), 65536, 60),     # BASELINE config 2
    ("TSP-v0", dict(n_nodes=128, n_edges=8128, parenting=1), 16384, 130),  # BASELINE config 3, full batch, one whole episode
    ("SteinerTree-v0", dict(n_nodes=256, n_edges=1024, n_dests=8), 16384, 40),  # BASELINE config 4: one GPU's shard of the 131 072
])
def test_full_size_invariants_and_sampled_oracle_parity(env_id, kw, B, K):
    """At BASELINE sizes: size-independent invariants on every slot + exact oracle replay of sampled slots."""
    import oracle
    ge = _ge()
    env = ge.make_vec(env_id, B, **kw)
    env.reset(seed=0)
    sample = sorted(set(np.random.default_rng(0).integers(0, B, 6).tolist() + [0, B - 1]))
    refs = {i: oracle.OracleEnv(env_id, **kw) for i in sample}
    seeds = {i: i for i in sample}
    for i in sample:
        refs[i].reset(seed=i)
    total_done = 0
    for k in range(K):
        a = env.sample_random_actions(policy_seed=3)
        a_cpu = a[sample].cpu().numpy() if len(sample) else None
        obs, rew, term, trunc, info = env.step(a)
        total_done += int(term.sum())
        rs, ts = rew[sample].cpu().numpy(), term[sample].cpu().numpy()
        ms = info["mask"][sample].cpu().numpy()
        for j, i in enumerate(sample):
            _, rr, dd, _, _ = refs[i].step(int(a_cpu[j]))
            assert rr == rs[j] and dd == bool(ts[j]), (k, i)
            if dd:
                seeds[i] = (seeds[i] + B) % 2**32
                refs[i].reset(seed=seeds[i])
            assert np.array_equal(ms[j], refs[i].mask()), (k, i)
        assert not info["invalid_action"].any()
    t = env.t
    assert int(t["tstep"].sum()) == B * K and int(t["episode"].sum()) == total_done
    # mask bytes == packed bits on every slot
    bits = t["mask_bits"].cpu().numpy().view(np.uint64)
    unpacked = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :env.A].astype(bool)
    assert np.array_equal(unpacked, t["mask"].cpu().numpy().astype(bool))
    if env_id == "ShortestPath-v0":
        # mask == N(head) & ~visited; visited flags in x mirror node_bits; exactly one target per slot
        n = env.n
        head = t["head"].long()
        adj = t["adj_bits"].view(B, n)[torch.arange(B, device="cuda"), head]
        assert torch.equal(t["mask_bits"][:, 0], adj & ~t["node_bits"][:, 0])
        x = t["x"].view(B, n, env.F)
        vis = torch.zeros(B, dtype=torch.int64, device="cuda")
        for v in range(n):
            vis |= (x[:, v, 0] == 1).long() << v
        assert torch.equal(vis, t["node_bits"][:, 0])
        assert (x[:, :, 1].sum(1) == 1).all()
        # edge_index is symmetric and row-major by source inside every slot
        ei = t["edge_index"].view(2, B, env.E)
        assert (ei[0, :, 1:] >= ei[0, :, :-1]).all()
    g = env.graph()
    assert g.x.shape == (B * env.n, env.F) and g.edge_index.shape == (2, B * env.E)
    for i in sample:
        flat_i = torch.cat([t["x"].view(B, -1)[i], t["edge_attr"].view(B, -1)[i],
                            (t["edge_index"].view(2, B, env.E)[:, i].T - i * env.n).reshape(-1).float()]).cpu().numpy()
        assert np.array_equal(flat_i, refs[i].obs()), i


def test_ragged_mixed_batch_matches_oracle():
    """BASELINE config 5 shape (reduced): {ShortestPath, MaxIndependentSet (= "MinVertexCover"), DensestSubgraph},
    ragged n, shared PyG slabs per env id."""
    import oracle
    from ragged_check import check_ragged_mixed
    check_ragged_mixed(_ge(), oracle, "cuda", steps=40)


def test_soak_many_episodes_sampled_slots_match_oracle():
    """600 vector steps at B = 8192 (about 200 000 regenerated episodes): 64 sampled slots are replayed on the oracle
    step by step, and their full observation (all five structural columns included) is compared after every reset."""
    import oracle
    ge = _ge()
    B, K, n, m = 8192, 600, 64, 192
    env = ge.make_vec("ShortestPath-v0", B, n_nodes=n, n_edges=m, is_eval_env=True)
    env.reset(seed=77)
    sample = sorted(set(np.random.default_rng(1).integers(0, B, 64).tolist()))
    idx = torch.tensor(sample, device="cuda")
    refs = {i: oracle.OracleEnv("ShortestPath-v0", n_nodes=n, n_edges=m, is_eval_env=True) for i in sample}
    seeds = {i: 77 + i for i in sample}
    for i in sample:
        refs[i].reset(seed=seeds[i])
    resets = 0
    E, F = env.E, env.F
    for k in range(K):
        a = env.sample_random_actions(policy_seed=9)
        a_s = a[idx].cpu().numpy()
        obs, rew, term, trunc, info = env.step(a)
        rs, ts, ms = rew[idx].cpu().numpy(), term[idx].cpu().numpy(), info["mask"][idx].cpu().numpy()
        fh = info["heuristic_solution"][idx].cpu().numpy()
        changed = []
        for j, i in enumerate(sample):
            _, rr, dd, _, inf = refs[i].step(int(a_s[j]))
            assert rr == rs[j] and dd == bool(ts[j]), (k, i)
            if dd:
                assert fh[j] == inf["heuristic_solution"], (k, i)
                seeds[i] = (seeds[i] + B) % 2**32
                refs[i].reset(seed=seeds[i])
                changed.append((j, i))
                resets += 1
            assert np.array_equal(ms[j], refs[i].mask()), (k, i)
        if changed:
            x = env.t["x"].view(B, n * F)[idx].cpu().numpy()
            ea = env.t["edge_attr"].view(B, E)[idx].cpu().numpy()
            ei = env.t["edge_index"].view(2, B, E)[:, idx].cpu().numpy()
            for j, i in changed:
                want = refs[i].obs()
                assert np.array_equal(x[j], want[: n * F]), (k, i, "x incl. structural features")
                assert np.array_equal(ea[j], want[n * F: n * F + E]), (k, i)
                links = (ei[:, j].T - i * n).reshape(-1).astype(np.float32)
                assert np.array_equal(links, want[n * F + E:]), (k, i)
    assert resets > 1000 and int(env.t["work_count"][0]) >= 0


@pytest.mark.parametrize("env_id,kw", [("ShortestPath-v0", dict(n_nodes=40, n_edges=100)),
                                       ("LongestPath-v0", dict(n_nodes=20, n_edges=50, parenting=1)),
                                       ("SteinerTree-v0", dict(n_nodes=70, n_edges=200, n_dests=6)),
                                       ("TSP-v0", dict(n_nodes=12, n_edges=40, parenting=1)),
                                       ("MaxIndependentSet-v0", dict(n_nodes=20, n_edges=40)),
                                       ("DensestSubgraph-v0", dict(n_nodes=30, n_edges=80, parenting=1)),
                                       ("MulticastRouting-v0", dict(n_nodes=40, n_edges=110, n_dests=5)),
                                       ("MulticastRouting-v0", dict(n_nodes=80, n_edges=260, n_dests=7, parenting=2)),
                                       ("DistributionCenter-v0", dict(n_nodes=50, n_edges=140)),
                                       ("DistributionCenter-v0", dict(n_nodes=90, n_edges=300, parenting=1)),
                                       ("PerishableProductDelivery-v0", dict(n_nodes=30, n_edges=90, n_products=3, parenting=1))])
def test_inject_state_then_step_matches_oracle(env_id, kw):
    """ge_inject_state: the oracle's post-reset states are loaded instead of sampled (SURVEY 7 parity path)."""
    import oracle
    from inject_check import check_inject
    check_inject(_ge(), oracle, "cuda", env_id, kw, B=16, steps=80)


def test_tsp_degenerate_start_action_and_dead_end():
    """tsp.py:203-211: choosing the start node while standing on it ends the episode with reward -n, cost -1."""
    import oracle
    ge = _ge()
    kw = dict(n_nodes=10, n_edges=20, parenting=1)
    env = ge.make_vec("TSP-v0", 4, autoreset=False, **kw)
    env.reset(seed=[3, 4, 5, 6])
    refs = [oracle.OracleEnv("TSP-v0", **kw) for _ in range(4)]
    for r, s in zip(refs, (3, 4, 5, 6)):
        r.reset(seed=s)
    obs, rew, term, trunc, info = env.step(torch.zeros(4, dtype=torch.int64, device="cuda"))
    for i, r in enumerate(refs):
        _, rr, dd, _, inf = r.step(0)
        assert rr == float(rew[i]) == -10.0 and dd and bool(term[i])
        assert float(info["solution_cost"][i]) == inf["solution_cost"] == -1.0 and int(info["solved"][i]) == 0


def test_state_dict_roundtrip_and_unseeded_reset():
    ge = _ge()
    env = ge.make_vec("ShortestPath-v0", 32, n_nodes=16, n_edges=40, obs_mode="flat")
    env.reset(seed=9)
    env.random_rollout(7, policy_seed=2)
    sd = env.state_dict()
    a = env.sample_random_actions(policy_seed=2).clone()
    env.step(a)
    after = env.flat_obs().clone(); rew = env.t["reward"].clone()
    env.load_state_dict(sd)
    env.step(a)
    assert torch.equal(env.flat_obs(), after) and torch.equal(env.t["reward"], rew)
    # reset(seed=None) moves every slot to its next episode seed
    seeds0 = env.t["seed"].clone()
    env.reset()
    assert torch.equal(env.t["seed"], seeds0 + 32) and int(env.t["episode"].sum()) == 0


def _random_config(rng):
    """a random valid constructor call of one of the nine envs (sizes the oracle replays in a moment)"""
    env_id = str(rng.choice(["ShortestPath-v0", "LongestPath-v0", "SteinerTree-v0", "TSP-v0", "DensestSubgraph-v0", "MaxIndependentSet-v0",
                         "MulticastRouting-v0", "DistributionCenter-v0", "PerishableProductDelivery-v0"]))
    n = int(rng.choice([5, 7, 9, 16, 31, 33, 64, 65, 90, 129, 140]))
    if env_id == "PerishableProductDelivery-v0":
        n = min(n, 128)
    ng = n - 1 if env_id == "DensestSubgraph-v0" else n
    # G(n, m) is sampled by rejection until connected (TSP: also no degree-1 node, no cut through node 0): keep m where
    # that succeeds within a few attempts -- the reference itself (and the device loop) would spin for ages below it
    floor_m = int(np.ceil((0.75 if env_id != "TSP-v0" else 1.1) * ng * np.log(ng))) if ng > 9 else ng + 1
    m = int(min(max(floor_m, round(rng.choice([1.0, 1.3, 2.0, 4.0]) * floor_m)), ng * (ng - 1) // 2))
    kw = dict(n_nodes=n, n_edges=m)
    kw["is_eval_env"] = bool(rng.integers(2))
    if env_id not in ("DensestSubgraph-v0",):
        kw["weighted"] = bool(rng.integers(4) > 0)
    if env_id == "LongestPath-v0":
        kw["parenting"] = int(rng.integers(0, 3))
    if env_id == "TSP-v0":
        kw["parenting"] = int(rng.integers(1, 3)); kw["spatial"] = bool(kw["weighted"] and rng.integers(3) == 0)
    if env_id == "DensestSubgraph-v0":
        kw["parenting"] = int(rng.integers(0, 2))
    if env_id == "SteinerTree-v0":
        kw["n_dests"] = int(rng.integers(1, n))
    if env_id == "MulticastRouting-v0":
        kw["n_dests"] = int(rng.integers(1, n)); kw["parenting"] = int(rng.integers(1, 5))
    if env_id == "DistributionCenter-v0":
        kw["parenting"] = int(rng.integers(1, 3)); kw["max_distance"] = float(rng.choice([0.5, 1, 1.0, 1.7, 3]))
        kw["target_count"] = int(rng.integers(0, max(1, n // 2)))
    if env_id == "PerishableProductDelivery-v0":
        kw["parenting"] = 1; kw["n_products"] = int(rng.integers(1, min(5, n // 2) + 1))
    return env_id, kw


@pytest.mark.parametrize("chunk", range(12))
def test_randomized_configs_match_oracle(chunk):
    """differential test: random constructor arguments of all nine envs, device policy, autoreset, every output compared
    with the CPU oracle (which is pinned by the reference fixtures)"""
    import oracle
    ge = _ge()
    import os
    rng = np.random.default_rng(int(os.environ.get("GE_FUZZ_SEED", 20260)) + chunk)  # (GE_FUZZ_SEED / GE_FUZZ_CONFIGS: a longer soak with other seeds)
    done_cfgs = 0
    for _ in range(int(os.environ.get("GE_FUZZ_CONFIGS", 12))):
        env_id, kw = _random_config(rng)
        B, K, stride, base, s0 = 6, 30, 101, 5, int(rng.integers(0, 2**32))
        try:
            env = ge.make_vec(env_id, B, obs_mode="flat", seed_stride=stride, env_index_base=base, **kw)
        except RuntimeError as e:  # geometry outside what is built (LDS), reported loudly
            assert "GE_E_TOOBIG" in str(e) or "GE_E_UNSUPPORTED" in str(e) or "fit" in str(e) or "built for" in str(e), (env_id, kw, str(e))
            continue
        tag = (env_id, kw, s0)
        obs, info = env.reset(seed=s0)
        refs = [oracle.OracleEnv(env_id, **kw) for _ in range(B)]
        seeds = [(s0 + base + i) % 2**32 for i in range(B)]
        want = np.stack([r.reset(seed=s)[0] for r, s in zip(refs, seeds)])
        assert np.array_equal(obs.cpu().numpy(), want), tag
        assert np.array_equal(info["mask"].cpu().numpy(), np.stack([r.mask() for r in refs])), tag
        tcount = [0] * B
        for k in range(K):
            a = env.sample_random_actions(policy_seed=3).clone()
            obs, rew, term, trunc, info = env.step(a)
            a, rew, term = a.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
            fc, fh, solved = info["solution_cost"].cpu().numpy(), info["heuristic_solution"].cpu().numpy(), info["solved"].cpu().numpy()
            for i, r in enumerate(refs):
                if a[i] < 0:  # empty mask (e.g. no target to cover): the slot idles
                    assert not r.mask().any(), tag
                    continue
                assert a[i] == oracle.policy_pick(r.mask(), 3, base + i, tcount[i]), tag
                _, rr, dd, _, inf = r.step(int(a[i]))
                tcount[i] += 1
                assert rr == rew[i] and dd == bool(term[i]), (tag, k, i, rr, rew[i])
                assert int(solved[i]) == (int(inf["solved"]) if "solved" in inf else -1), (tag, k, i)
                if dd:
                    assert fc[i] == inf["solution_cost"], (tag, k, i, fc[i], inf["solution_cost"])
                    if not np.isnan(inf["heuristic_solution"]):
                        assert fh[i] == inf["heuristic_solution"], (tag, k, i)
                    seeds[i] = (seeds[i] + stride) % 2**32
                    r.reset(seed=seeds[i])
            assert np.array_equal(info["mask"].cpu().numpy(), np.stack([r.mask() for r in refs])), (tag, k)
        assert np.array_equal(env.flat_obs().cpu().numpy(), np.stack([r.obs() for r in refs])), tag
        env.close()
        done_cfgs += 1
    assert done_cfgs >= 6


@pytest.mark.parametrize("env_id,kw", [("ShortestPath-v0", dict(n_nodes=64, n_edges=192)),
                                       ("LongestPath-v0", dict(n_nodes=24, n_edges=50, parenting=2)),
                                       ("TSP-v0", dict(n_nodes=14, n_edges=40, parenting=2)),
                                       ("MulticastRouting-v0", dict(n_nodes=30, n_edges=70, n_dests=4)),
                                       ("DensestSubgraph-v0", dict(n_nodes=30, n_edges=60, parenting=1)),
                                       ("DistributionCenter-v0", dict(n_nodes=70, n_edges=200))])
def test_next_step_autoreset_matches_oracle(env_id, kw):
    import oracle
    gu.check_next_step_autoreset(_ge(), oracle, env_id, kw, 48, 80, "cuda")


def test_cpp_host_on_the_c_abi_alone_matches_the_python_host():
    """examples/abi_demo.cpp allocates every buffer with hipMalloc and calls only include/graphenvs.h (no Python, no torch in
    that process); its counters must equal what the Python host gets for the same seeds and policy."""
    import json
    import os
    import subprocess
    ge = _ge()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "abi_demo")
    if not os.path.exists(exe):
        import __graft_entry__
        exe = __graft_entry__.build_abi_demo()
    B, K = 2048, 60
    out = subprocess.run([exe, str(B), str(K)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout.strip().splitlines()[-1])
    env = ge.make_vec("ShortestPath-v0", B, n_nodes=64, n_edges=192)
    env.reset(seed=0)
    env.random_rollout(K, policy_seed=1)
    torch.cuda.synchronize()
    assert got["episodes"] == int(env.t["episode"].sum()) and got["transitions"] == int(env.t["tstep"].sum())
    # the demo adds left to right; numpy's sum() is pairwise, cumsum() is sequential
    assert got["cost_sum"] == float(np.cumsum(env.t["cost"].cpu().numpy())[-1])
    assert got["x_sum"] == float(np.cumsum(env.t["x"].cpu().numpy().astype(np.float64).ravel())[-1])


@pytest.mark.parametrize("chunk", range(3))
def test_randomized_invalid_and_noop_actions(chunk):
    """random constructor calls of all nine envs, a third of the actions replaced by arbitrary integers in [-3, A + 3): the engine
    flags exactly what the oracle (= the reference's asserts) refuses, leaves those slots untouched, treats -1 as a no-op"""
    import oracle
    ge = _ge()
    rng = np.random.default_rng(41000 + chunk)
    done_cfgs = 0
    for _ in range(10):
        env_id, kw = _random_config(rng)
        B, K, stride = 8, 40, 97
        try:
            env = ge.make_vec(env_id, B, obs_mode="flat", seed_stride=stride, **kw)
        except RuntimeError:
            continue
        s0 = int(rng.integers(0, 2**31))
        env.reset(seed=s0)
        refs = [oracle.OracleEnv(env_id, **kw) for _ in range(B)]
        seeds = [s0 + i for i in range(B)]
        for r, sd in zip(refs, seeds):
            r.reset(seed=sd)
        for k in range(K):
            a = env.sample_random_actions(policy_seed=11).clone().cpu().numpy()
            wild = rng.random(B) < 0.35
            a[wild] = rng.integers(-3, env.A + 3, size=int(wild.sum()))
            before = env.flat_obs().clone()
            obs, rew, term, trunc, info = env.step(torch.from_numpy(a).cuda())
            inv, rew, term = info["invalid_action"].cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
            after = env.flat_obs()
            for i, r in enumerate(refs):
                tag = (env_id, kw, k, i, int(a[i]))
                if a[i] == -1:
                    assert not inv[i] and rew[i] == 0 and not term[i], tag
                    continue
                try:
                    _, rr, dd, _, inf = r.step(int(a[i]))
                except AssertionError:
                    assert inv[i] and rew[i] == 0 and not term[i], tag
                    assert torch.equal(before[i], after[i]), tag
                    continue
                assert not inv[i] and rr == rew[i] and dd == bool(term[i]), tag
                if dd:
                    seeds[i] = (seeds[i] + stride) % 2**32
                    r.reset(seed=seeds[i])
            assert np.array_equal(info["mask"].cpu().numpy(), np.stack([r.mask() for r in refs])), (env_id, kw, k)
        assert np.array_equal(env.flat_obs().cpu().numpy(), np.stack([r.obs() for r in refs])), (env_id, kw)
        env.close()
        done_cfgs += 1
    assert done_cfgs >= 5


def test_call_order_is_guarded():
    import host_checks as hc
    hc.check_call_order(_ge(), "cuda", None)


@pytest.mark.parametrize("mode", [True, "next_step"])
def test_inject_with_seeds_then_autoreset(mode):
    import host_checks as hc
    import oracle
    hc.check_inject_seeds_autoreset(_ge(), oracle, "cuda", None, mode)


def test_state_dict_moves_between_engines():
    import host_checks as hc
    hc.check_state_dict_move(_ge(), "cuda", None)


def test_unseeded_reset_continues_the_streams_of_every_slot():
    import host_checks as hc
    import oracle
    hc.check_continue_streams(_ge(), oracle, "cuda", None)


def test_return_graph_obs_and_copy_outputs():
    import host_checks as hc
    hc.check_graph_obs(_ge(), "cuda", None)


def test_config5_ragged_batch_one_engine_per_env_id():
    """BASELINE config 5 for real: three env ids, 66 distinct sizes each from U{32..512} (n = 512 inside the slab), every slot
    replayed on the oracle; ONE multi-class engine -- one launch sequence -- per env id."""
    import oracle
    from ragged_check import check_ragged_mixed, config5_specs
    specs = config5_specs()
    assert all(len({n for _, n, _ in sizes}) >= 64 and max(n for _, n, _ in sizes) == 512 for _, sizes, _ in specs)
    check_ragged_mixed(_ge(), oracle, "cuda", steps=70, specs=specs)


# what an engine with episode prefetch must reproduce of the same engine without: every live slab except the generator ring (seeded
# one episode later by design), the queues and the work space
_PREFETCH_SKIP = ("mt_state", "reset_list", "reset_count", "work_list", "work_count", "feat_scratch", "eval_scratch")
PREFETCH_CASES = [
    ("ShortestPath-v0", dict(n_nodes=64, n_edges=192, is_eval_env=True), 4096, 3, True),      # a tenth of the finished slots finish again before the refill
    ("ShortestPath-v0", dict(n_nodes=64, n_edges=192), 4096, 1, "next_step"),
    ("LongestPath-v0", dict(n_nodes=24, n_edges=50, parenting=2), 300, 5, True),
    ("SteinerTree-v0", dict(n_nodes=256, n_edges=1024, n_dests=8), 512, 16, True),             # BASELINE config 4's geometry
    ("SteinerTree-v0", dict(n_nodes=40, n_edges=100, n_dests=5, is_eval_env=True), 200, 7, True),
    ("TSP-v0", dict(n_nodes=20, n_edges=60, parenting=1), 300, 4, True),
    ("TSP-v0", dict(n_nodes=70, n_edges=300, parenting=1, spatial=True), 64, 9, True),
    ("DensestSubgraph-v0", dict(n_nodes=64, n_edges=192, parenting=1), 1000, 2, True),
    ("MaxIndependentSet-v0", dict(n_nodes=130, n_edges=500, weighted=False, is_eval_env=True), 40, 6, True),
    ("MulticastRouting-v0", dict(n_nodes=64, n_edges=192, n_dests=5), 1000, 5, True),
    ("MulticastRouting-v0", dict(n_nodes=40, n_edges=100, n_dests=4, parenting=2), 300, 5, "next_step"),
    ("DistributionCenter-v0", dict(n_nodes=64, n_edges=192), 1000, 2, True),
    ("DistributionCenter-v0", dict(n_nodes=100, n_edges=260, weighted=False), 64, 3, True),
    ("PerishableProductDelivery-v0", dict(n_nodes=20, n_edges=50, parenting=1), 300, 11, True),
]


@pytest.mark.parametrize("env_id,kw,B,period,mode", PREFETCH_CASES)
def test_prefetch_reproduces_the_engine_without_it(env_id, kw, B, period, mode):
    """ge_attach_spares changes when a regeneration is paid for, never what it produces: after every stretch of steps the two
    engines hold the same slabs (observation, CSR, masks, scalar state, outputs of the last step, seed / episode of every slot)."""
    ge = _ge()
    a = ge.make_vec(env_id, B, prefetch=0, autoreset=mode, **kw)
    b = ge.make_vec(env_id, B, prefetch=period, autoreset=mode, **kw)
    assert a.spare is None and b.spare is not None
    a.reset(seed=3); b.reset(seed=3)
    served = 0
    for stretch in (1, 2, 5, 13, 40) + ((120, 160) if kw["n_nodes"] >= 100 else ()):  # (episodes of config 4 last ~200 steps)
        for _ in range(stretch):
            valid = b.spare["state"].clone()
            a.random_rollout(1, policy_seed=4); b.random_rollout(1, policy_seed=4)
            served += int((valid.bool() & (b.t["terminated"] != 0)).sum())
        for key, va in dict.items(a.t):
            if va is None or key in _PREFETCH_SKIP:
                continue
            vb = dict.__getitem__(b.t, key)
            if key == "range_bits" and a.t["aux_bits"] is not None:
                # DistributionCenter, n <= 64: coverage rows are computed when first needed and aux_bits says which exist; the
                # others hold whatever an earlier episode left, which differs between a regeneration in place and a copied image
                n = kw["n_nodes"]
                have = ((a.t["aux_bits"].view(-1, 1) >> torch.arange(n, device=va.device).view(1, -1)) & 1).bool().view(-1, 1)
                va, vb = torch.where(have, va, torch.zeros_like(va)), torch.where(have, vb, torch.zeros_like(vb))
            assert torch.equal(va, vb), (stretch, key)
    assert int(a.t["episode"].sum()) > 0 and served > 0
    a.check_device_errors(); b.check_device_errors()
    a.close(); b.close()


@pytest.mark.parametrize("mode", [True, "next_step"])
def test_prefetch_survives_a_reset_in_the_middle_of_a_rollout(mode):
    import host_checks as hc
    hc.check_prefetch_reset_mid_rollout(_ge(), "cuda", None, mode)


def test_prefetch_state_dict_in_next_step_mode():
    import host_checks as hc
    hc.check_prefetch_state_dict_next_step(_ge(), "cuda", None)


def test_config5_at_full_size_invariants_and_sampled_slots():
    """BASELINE config 5 as bench.py times it: 3 x 16 384 slots, n ~ U{32..512} (481 size classes per env id), one multi-class
    engine per id -- invariants over every slot, 16 sampled slots per id against the oracle (policy, rewards, masks, regenerated
    observations)"""
    import oracle
    from ragged_check import check_config5_full_size
    check_config5_full_size(_ge(), oracle, "cuda")


def test_spatial_tsp_fixture_sits_on_the_pow_boundary():
    """tsp_n12_m30_p1_spatial_pow2: seeds where the reference's float ** 2 (libm pow) and a multiplication differ in the last bit of an
    edge weight.  The engine multiplies: the replay finds rewards that agree to 1e-12 relative but not exactly -- the documented
    tolerance of spatial TSP is exercised, not vacuous"""
    case = gu.load_case("tsp_n12_m30_p1_spatial_pow2")
    st = gu.replay_case(case, lambda env_id, **kw: _ge().GraphEnv(env_id, **kw))
    assert st["inexact_rewards"] > 0 and st["steps"] > 30


@pytest.mark.gpu
def test_shards_equal_one_engine():
    """make_vec(shards=3): three engines on their own HIP streams hold the same slots as one engine, bit for bit (graphenvs_amd.sharded)"""
    import host_checks as hc
    ge = _ge()
    hc.check_shards_equal_one_engine(ge, "cuda", kw=dict(n_nodes=64, n_edges=192), B=100, K=40)
    hc.check_shards_equal_one_engine(ge, "cuda", env_id="SteinerTree-v0", kw=dict(n_nodes=70, n_edges=200, n_dests=4), B=10, K=60)
