"""CPU: the HIP kernel sources compiled for the sanitizer harness (tests/emu, UBSan + divergent-barrier
detector) replay the reference's golden vectors.  This is a debugging aid for a GPU-less container;
the parity claims rest on tests/test_gpu_parity.py, which runs libgraphenvs_hip.so on the MI355X."""
import os
import sys

import numpy as np
import pytest

import golden_util as gu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import build_emu  # noqa: E402
import host_checks as hc  # noqa: E402

import graphenvs_amd as ge  # noqa: E402

FAST = ["sp_n10_m20_eval", "sp_n5_m7", "sp_n33_m70", "sp_n64_m192_eval", "lp_n10_m20_p0", "lp_n10_m20_p1",
        "st_n10_m20_d1_eval", "st_n10_m20_d3_eval", "st_n10_m20_d9_eval", "st_n5_m10_d4", "tsp_n8_m28_p1",
        "tsp_n10_m20_p1", "tsp_n10_m20_p1_eval", "tsp_n12_m30_p2_spatial_eval", "tsp_n12_m30_p1_unweighted", "mis_n6_m8", "mis_n5_m7_unweighted", "mis_n12_m20_unweighted_eval", "ds_n10_m20_p1",
        "ds_n10_m20_p0_eval", "ds_n100_m300_p1", "sp_n10_m20_unweighted", "st_n10_m20_d3_unweighted",
        "lp_n10_m20_p2", "lp_n64_m192_p2", "lp_n12_m24_p3", "lp_n30_m60_p3", "tsp_n10_m20_p2", "tsp_n12_m30_p2_spatial",
        "mc_n10_m20_p4", "mc_n10_m20_p3_d2", "mc_n10_m20_p2", "mc_n10_m20_p1", "mc_n10_m20_p4_eval",
        "mc_n12_auto_edges_p4_unweighted", "mc_n64_m192_p4_d40_eval",
        "dc_n10_m20_p2", "dc_n10_m20_p1", "dc_n12_m25_p2_unweighted_t4", "dc_n20_m40_p2_dist1p5_eval", "dc_n64_m192_p2_dist0p7",
        "ppd_n8_m9_p3", "ppd_n12_auto_unweighted", "ppd_n7_m21_complete", "mc_n8_m28_complete_p4_eval",
        "dc_n8_m28_complete_dist0p6"]


@pytest.fixture(scope="module")
def emu():
    return build_emu.load()


@pytest.fixture(scope="module")
def stress():
    """ONE more build of the harness with every path that only large or deep graphs take forced onto small ones: the generic feature
    kernel for slots deeper than 3 levels (-DGE_F64_LV=3), the late numpy draws of graphs above 256 nodes (-DGE_NP_EARLY_MAX=8), the
    residual-graph walks of parenting >= 2 above 512 nodes (-DGE_MAXW=1: above 64), the placement of PerishableProductDelivery above
    128 nodes (-DGE_PPD_WIDE_ABOVE=6), the generic feature kernel's betweenness partial sums in LDS instead of registers, as above 512
    nodes (-DGE_BCW_REG_W=1: above 64), and its two-rows-per-lane walks (-DGE_BW_U=2 -DGE_BW_TWO_ABOVE=64: not shipped, kept as a measured variant)"""
    return build_emu.load(extra=["-DGE_F64_LV=3", "-DGE_NP_EARLY_MAX=8", "-DGE_MAXW=1", "-DGE_PPD_WIDE_ABOVE=6", "-DGE_BCW_REG_W=1", "-DGE_BW_TWO_ABOVE=64", "-DGE_BW_U=2"],
                          out=os.path.join(os.path.dirname(build_emu.OUT), "libgraphenvs_emu_stress.so"))


@pytest.mark.parametrize("name", FAST)
def test_emulated_kernels_replay_golden(emu, name):
    case = gu.load_case(name)
    st = gu.replay_case(case, lambda env_id, **kw: ge.GraphEnv(env_id, device="cpu", _library=emu, **kw))
    assert st["resets"] > 0


@pytest.mark.parametrize("n,m", [(10, 20), (4, 5)])  # n = 4: episodes of one or two steps -- a slot finishes in consecutive steps and
def test_emulated_batch_autoreset_matches_oracle(emu, n, m):  # walks round the ring of pre-seeded generator states
    import oracle
    B, K, stride = 6, 40, 1000
    env = ge.VectorGraphEnv("ShortestPath-v0", B, n, m, device="cpu", _library=emu, is_eval_env=True,
                            seed_stride=stride, env_index_base=3)
    env.reset(seed=5)
    refs = [oracle.OracleEnv("ShortestPath-v0", n_nodes=n, n_edges=m, is_eval_env=True) for _ in range(B)]
    eps = [0] * B
    for i, r in enumerate(refs):
        r.reset(seed=5 + 3 + i)
    for k in range(K):
        a = env.sample_random_actions(policy_seed=9).clone().numpy()
        want_a = [oracle.policy_pick(r.mask(), 9, 3 + i, k) for i, r in enumerate(refs)]
        assert a.tolist() == want_a
        _, rew, term, _, info = env.step(a)
        for i, r in enumerate(refs):
            _, rr, dd, _, inf = r.step(int(a[i]))
            assert float(rew[i]) == rr and bool(term[i]) == dd
            if dd:
                assert float(info["solution_cost"][i]) == inf["solution_cost"]
                assert float(info["heuristic_solution"][i]) == inf["heuristic_solution"]
                assert int(info["solved"][i]) == int(inf["solved"])
                eps[i] += 1
                r.reset(seed=(5 + 3 + i + eps[i] * stride) % 2**32)
            assert np.array_equal(info["mask"][i].numpy(), r.mask())
        flat = env.flat_obs().numpy()
        for i, r in enumerate(refs):
            assert np.array_equal(flat[i], r.obs())
    assert sum(eps) > B


def _prefetch_rollout(emu, env_id, kw, B, K, period, autoreset=True, stride=1000):
    """an engine with spares (episode prefetch) against the oracle: rewards, done, info, masks and the whole observation after
    every step; returns (episodes, how many of them were served by a spare image)"""
    import oracle
    env = ge.VectorGraphEnv(env_id, B, device="cpu", _library=emu, seed_stride=stride, env_index_base=3, prefetch=period,
                            autoreset=autoreset, obs_mode="flat", **kw)
    env.reset(seed=5)
    refs = [oracle.OracleEnv(env_id, **kw) for _ in range(B)]
    eps, pend, swaps = [0] * B, [False] * B, 0
    for i, r in enumerate(refs):
        r.reset(seed=5 + 3 + i)
    for k in range(K):
        a = env.sample_random_actions(policy_seed=9).clone().numpy()
        valid = env.spare["state"].clone().numpy()
        _, rew, term, _, info = env.step(a)
        for i, r in enumerate(refs):
            if pend[i]:  # next-step autoreset: this step regenerated the slot and ignored its action
                eps[i] += 1
                r.reset(seed=(5 + 3 + i + eps[i] * stride) % 2**32)
                pend[i] = False
                assert float(rew[i]) == 0 and not bool(term[i])
            else:
                _, rr, dd, _, inf = r.step(int(a[i]))
                assert float(rew[i]) == rr and bool(term[i]) == dd, (k, i)
                if dd:
                    assert float(info["solution_cost"][i]) == inf["solution_cost"]
                    assert float(info["heuristic_solution"][i]) == inf["heuristic_solution"]
                    swaps += int(valid[i])
                    if autoreset == "next_step":
                        pend[i] = True
                    else:
                        eps[i] += 1
                        r.reset(seed=(5 + 3 + i + eps[i] * stride) % 2**32)
            assert np.array_equal(info["mask"][i].numpy(), r.mask()), (k, i)
        flat = env.flat_obs().numpy()
        for i, r in enumerate(refs):
            assert np.array_equal(flat[i], r.obs()), (k, i)
    env.close()
    return sum(eps), swaps


@pytest.mark.parametrize("env_id,kw,B,K,period,autoreset", [
    ("ShortestPath-v0", dict(n_nodes=10, n_edges=20, is_eval_env=True), 6, 40, 1, True),
    ("ShortestPath-v0", dict(n_nodes=10, n_edges=20, is_eval_env=True), 6, 40, 7, True),
    ("ShortestPath-v0", dict(n_nodes=4, n_edges=5, is_eval_env=True), 6, 40, 4, True),   # episodes of one or two steps: most finish again before the refill
    ("ShortestPath-v0", dict(n_nodes=10, n_edges=20, is_eval_env=True), 6, 40, 3, "next_step"),
    ("SteinerTree-v0", dict(n_nodes=12, n_edges=30, n_dests=3, is_eval_env=True), 5, 40, 5, True),
    ("DistributionCenter-v0", dict(n_nodes=12, n_edges=25), 5, 30, 2, True),
    ("ShortestPath-v0", dict(n_nodes=70, n_edges=160), 3, 36, 4, True),                    # generic feature kernel, feat_parts workgroups
])
def test_emulated_prefetch_matches_oracle(emu, env_id, kw, B, K, period, autoreset):
    episodes, swaps = _prefetch_rollout(emu, env_id, kw, B, K, period, autoreset)
    assert episodes >= B and swaps > 0
    if kw["n_nodes"] == 4:
        assert swaps < episodes  # both ways of getting the next episode ran: the image, and the regeneration in place


@pytest.mark.parametrize("env_id,kw", [("ShortestPath-v0", dict(n_nodes=12, n_edges=30)),
                                       ("LongestPath-v0", dict(n_nodes=12, n_edges=30, parenting=1)),
                                       ("SteinerTree-v0", dict(n_nodes=12, n_edges=30, n_dests=3))])
def test_emulated_fused_rollout_equals_sample_then_step(emu, env_id, kw):
    import torch
    a = ge.VectorGraphEnv(env_id, 70, device="cpu", _library=emu, obs_mode="flat", **kw)
    b = ge.VectorGraphEnv(env_id, 70, device="cpu", _library=emu, obs_mode="flat", **kw)
    a.reset(seed=1); b.reset(seed=1)
    for k in range(15):
        a.step(a.sample_random_actions(policy_seed=4).clone())
        b.random_rollout(1, policy_seed=4)
        for key in ("reward", "terminated", "mask", "mask_bits", "head", "node_bits", "cost", "episode", "tstep", "x", "solved"):
            assert torch.equal(a.t[key], b.t[key]), (k, key)
    assert int(a.t["episode"].sum()) > 0


def test_emulated_feature_fast_path_falls_back_to_generic_when_too_deep(stress):
    """-DGE_F64_LV=3 forces the lane-per-source path to hand deep slots to the generic kernel."""
    lib = stress
    for name in ["sp_n33_m70", "ds_n10_m20_p1", "tsp_n10_m20_p1", "tsp_n8_m28_p1", "dc_n8_m28_complete_dist0p6",  # (the last two: complete graphs, whose rows the generic kernel does not stage)
                 "ds_n100_m300_p1"]:  # (above 64 nodes: partial sums in LDS and two rows per lane in this build)
        case = gu.load_case(name)
        gu.replay_case(case, lambda env_id, **kw: ge.GraphEnv(env_id, device="cpu", _library=lib, **kw), policies=("first",))


def test_emulated_feature_fast_path_falls_back_when_a_path_count_has_no_reciprocal():
    """-DGE_F64_INV=4: the n <= 64 feature kernel keeps reciprocals of the path counts 1 .. 3 only, so nearly every slot meets a count
    without one in its forward pull (not in the level search, as with -DGE_F64_LV=3) and is recomputed by the generic kernel"""
    lib = build_emu.load(extra=["-DGE_F64_INV=4"], out=os.path.join(os.path.dirname(build_emu.OUT), "libgraphenvs_emu_inv4.so"))
    for name in ["sp_n33_m70", "sp_n64_m192_eval", "ds_n10_m20_p1", "mis_n6_m8"]:
        gu.replay_case(gu.load_case(name), lambda env_id, **kw: ge.GraphEnv(env_id, device="cpu", _library=lib, **kw), policies=("first",))


def test_emulated_late_numpy_draws_of_large_graphs(stress):
    """-DGE_NP_EARLY_MAX=8: graphs above 8 nodes take the path of n > 256 -- the delay matrix is not materialised, the draw scan
    only counts and picks out the cells of the edges (ge_np_draws_edges)."""
    lib = stress
    for name in ["sp_n10_m20_eval", "sp_n33_m70", "st_n10_m20_d3_eval", "mc_n10_m20_p4_eval", "dc_n10_m20_p2", "lp_n10_m20_p1",
                 "ppd_n10_m20", "ppd_n12_m30_p5_eval"]:  # (PerishableProductDelivery draws its matrix INSIDE the rejection loop: the codes of an attempt's edges)
        gu.replay_case(gu.load_case(name), lambda env_id, **kw: ge.GraphEnv(env_id, device="cpu", _library=lib, **kw), policies=("first",))


@pytest.mark.parametrize("env_id,kw", [("LongestPath-v0", dict(n_nodes=70, n_edges=170, parenting=2)),
                                       ("LongestPath-v0", dict(n_nodes=66, n_edges=150, parenting=3)),
                                       ("TSP-v0", dict(n_nodes=66, n_edges=200, parenting=2))])
def test_emulated_residual_walks_in_memory(stress, env_id, kw):
    """-DGE_MAXW=1: graphs above 64 nodes take the path of graphs above 512 -- the node sets of the parenting >= 2 walks of step()
    live in prune_scratch instead of registers (longest_path.py:134-143, tsp.py:181-194)."""
    import oracle
    lib = stress
    B, K = 2, 30
    env = ge.VectorGraphEnv(env_id, B, device="cpu", _library=lib, obs_mode="flat", autoreset=False, **kw)
    assert env.t["prune_scratch"] is not None
    env.reset(seed=4)
    refs = [oracle.OracleEnv(env_id, **kw) for _ in range(B)]
    for i, r in enumerate(refs):
        r.reset(seed=4 + i)
    alive = [True] * B
    for k in range(K):
        a = env.sample_random_actions(policy_seed=6).clone().numpy()
        _, rew, term, _, info = env.step(a)
        for i, r in enumerate(refs):
            if not alive[i]:
                continue
            _, rr, dd, _, _ = r.step(int(a[i]))
            assert float(rew[i]) == rr and bool(term[i]) == dd, (k, i)
            assert np.array_equal(info["mask"][i].numpy(), r.mask()), (k, i)
            alive[i] = not dd
    env.close()


def test_emulated_perishable_delivery_placement_of_large_graphs(stress):
    """-DGE_PPD_WIDE_ABOVE=6: graphs above 6 nodes take the placement of graphs above 128 -- no Floyd-Warshall matrix in LDS,
    distances per pickup, node sets of W words (ge_ppd_place_wide); the fixtures at n = 200 / 300 run on the GPU."""
    lib = stress
    for name in ["ppd_n8_m9_p3", "ppd_n7_m21_complete"]:  # (sparse: placement retries; complete: the closed-form candidate order)
        gu.replay_case(gu.load_case(name), lambda env_id, **kw: ge.GraphEnv(env_id, device="cpu", _library=lib, **kw), policies=("first",))


def test_emulated_dense_rows_use_the_scode_fallback(emu):
    """degree > 16: the nibble-packed node record cannot hold the row, the step falls back to row_ptr + scode."""
    import oracle
    B, n, m = 5, 24, 230
    env = ge.VectorGraphEnv("ShortestPath-v0", B, n, m, device="cpu", _library=emu, obs_mode="flat", autoreset=False)
    env.reset(seed=2)
    refs = [oracle.OracleEnv("ShortestPath-v0", n_nodes=n, n_edges=m) for _ in range(B)]
    for i, r in enumerate(refs):
        r.reset(seed=2 + i)
    assert max(int(bin(int(v) & (2**64 - 1)).count("1")) for v in env.t["adj_bits"].flatten().tolist()) > 16
    alive = [True] * B
    for k in range(12):
        a = env.sample_random_actions(policy_seed=1).clone().numpy()
        _, rew, term, _, info = env.step(a)
        for i, r in enumerate(refs):
            if not alive[i]:
                continue
            _, rr, dd, _, _ = r.step(int(a[i]))
            assert float(rew[i]) == rr and bool(term[i]) == dd
            assert np.array_equal(info["mask"][i].numpy(), r.mask())
            alive[i] = not dd


def test_emulated_ragged_mixed_batch_matches_oracle(emu):
    import oracle
    from ragged_check import check_ragged_mixed
    check_ragged_mixed(ge, oracle, "cpu", library=emu, steps=12)


@pytest.mark.parametrize("env_id,kw", [("ShortestPath-v0", dict(n_nodes=12, n_edges=30)),
                                       ("SteinerTree-v0", dict(n_nodes=12, n_edges=30, n_dests=4)),
                                       ("TSP-v0", dict(n_nodes=9, n_edges=20, parenting=1)),
                                       ("MaxIndependentSet-v0", dict(n_nodes=9, n_edges=14)),
                                       ("DensestSubgraph-v0", dict(n_nodes=12, n_edges=24, parenting=1)),
                                       ("MulticastRouting-v0", dict(n_nodes=12, n_edges=30, n_dests=3)),
                                       ("MulticastRouting-v0", dict(n_nodes=12, n_edges=30, n_dests=3, parenting=2)),
                                       ("DistributionCenter-v0", dict(n_nodes=15, n_edges=40)),
                                       ("PerishableProductDelivery-v0", dict(n_nodes=12, n_edges=30, parenting=1)),
                                       ("TSP-v0", dict(n_nodes=8, n_edges=28, parenting=1))])  # complete graph: the engine's reset kernel keeps no edge list in LDS, the inject launch does
def test_emulated_inject_state_then_step(emu, env_id, kw):
    import oracle
    from inject_check import check_inject
    check_inject(ge, oracle, "cpu", env_id, kw, library=emu)


def test_emulated_complete_graph_tsp_above_64_nodes_keeps_no_edge_lists_in_lds(emu):
    """TSP on the complete graph, n > 64 (BASELINE config 3's shape): the reset kernel holds neither the {neighbour, code} list nor
    the per-entry code list in LDS (GeParams.nocolw / nowsort), the generic feature kernel stages code bytes only"""
    import oracle
    kw, B = dict(n_nodes=66, n_edges=66 * 65 // 2, parenting=1), 2
    env = ge.VectorGraphEnv("TSP-v0", B, device="cpu", _library=emu, seed_stride=1000, env_index_base=3, prefetch=0, obs_mode="flat", **kw)
    env.reset(seed=5)
    refs = [oracle.OracleEnv("TSP-v0", **kw) for _ in range(B)]
    for i, r in enumerate(refs):
        r.reset(seed=5 + 3 + i)
    flat = env.flat_obs().numpy()
    for i, r in enumerate(refs):
        assert np.array_equal(flat[i], r.obs()), i
    for k in range(3):
        a = env.sample_random_actions(policy_seed=9).clone().numpy()
        _, rew, term, _, info = env.step(a)
        for i, r in enumerate(refs):
            _, rr, dd, _, _ = r.step(int(a[i]))
            assert float(rew[i]) == rr and bool(term[i]) == dd and np.array_equal(info["mask"][i].numpy(), r.mask()), (k, i)
    env.close()


@pytest.mark.parametrize("env_id,kw", [("MaxIndependentSet-v0", dict(n_nodes=66, n_edges=66 * 65 // 2, weighted=False)),
                                       ("ShortestPath-v0", dict(n_nodes=66, n_edges=66 * 65 // 2))])
def test_emulated_complete_graph_above_64_nodes_of_an_env_without_edge_weights_in_its_pagerank(emu, env_id, kw):
    """a complete graph on all n > 64 nodes of an env other than TSP: the generic feature kernel stages no neighbour list (the
    neighbour of a row entry is a closed form) and the unweighted pagerank reads sinv * x of that neighbour"""
    import oracle
    B = 2
    env = ge.VectorGraphEnv(env_id, B, device="cpu", _library=emu, seed_stride=1000, env_index_base=3, prefetch=0, obs_mode="flat", **kw)
    env.reset(seed=5)
    refs = [oracle.OracleEnv(env_id, **kw) for _ in range(B)]
    for i, r in enumerate(refs):
        r.reset(seed=5 + 3 + i)
    flat = env.flat_obs().numpy()
    for i, r in enumerate(refs):
        assert np.array_equal(flat[i], r.obs()), i
    env.close()


def test_emulated_shards_equal_one_engine(emu):
    hc.check_shards_equal_one_engine(ge, "cpu", library=emu)


def test_emulated_next_step_autoreset(emu):
    import oracle
    gu.check_next_step_autoreset(ge, oracle, "ShortestPath-v0", dict(n_nodes=10, n_edges=20), 5, 40, "cpu", lib=emu)
    gu.check_next_step_autoreset(ge, oracle, "SteinerTree-v0", dict(n_nodes=12, n_edges=26, n_dests=3), 4, 40, "cpu", lib=emu)


def test_emulated_call_order_is_guarded(emu):
    hc.check_call_order(ge, "cpu", emu)


@pytest.mark.parametrize("mode", [True, "next_step"])
def test_emulated_inject_with_seeds_then_autoreset(emu, mode):
    import oracle
    hc.check_inject_seeds_autoreset(ge, oracle, "cpu", emu, mode)


def test_emulated_state_dict_moves_between_engines(emu):
    hc.check_state_dict_move(ge, "cpu", emu)


@pytest.mark.parametrize("mode", [True, "next_step"])
def test_emulated_prefetch_survives_a_reset_in_the_middle_of_a_rollout(emu, mode):
    hc.check_prefetch_reset_mid_rollout(ge, "cpu", emu, mode)


def test_emulated_prefetch_state_dict_in_next_step_mode(emu):
    hc.check_prefetch_state_dict_next_step(ge, "cpu", emu)


@pytest.mark.parametrize("env_id,kw", [("TSP-v0", dict(n_nodes=14, n_edges=50, parenting=1)), ("TSP-v0", dict(n_nodes=12, n_edges=30, parenting=2, spatial=True)),
                                       ("MaxIndependentSet-v0", dict(n_nodes=20, n_edges=45, weighted=False)),
                                       ("MaxIndependentSet-v0", dict(n_nodes=36, n_edges=80, weighted=False)),
                                       ("SteinerTree-v0", dict(n_nodes=16, n_edges=36, n_dests=4)),
                                       ("SteinerTree-v0", dict(n_nodes=30, n_edges=70, n_dests=20, weighted=False))])
def test_emulated_sequential_baseline_kernels_match_the_checker_through_autoresets(emu, env_id, kw):
    """is_eval_env of TSP (closure + Christofides kernels), of unweighted MaxIndependentSet (clique removal kernel) and of SteinerTree
    (Kou kernel), in full-reset and queue mode: heuristic[] of the running episode and info['heuristic_solution'] of the finished one equal the checker's, slot by slot"""
    import oracle
    B, stride = 5, 100
    env = ge.VectorGraphEnv(env_id, B, device="cpu", _library=emu, obs_mode="flat", seed_stride=stride, is_eval_env=True, **kw)
    seeds = list(range(3, 3 + B))
    env.reset(seed=seeds)
    refs = [oracle.OracleEnv(env_id, is_eval_env=True, **kw) for _ in range(B)]
    for r, sd in zip(refs, seeds):
        r.reset(seed=sd)
    ends = 0
    for k in range(min(2 * kw["n_nodes"] + 2, 60)):
        assert env.t["heuristic"].tolist() == [r.heuristic_solution for r in refs], k
        a = env.sample_random_actions(policy_seed=4).clone()
        _, rew, term, _, info = env.step(a)
        for i, r in enumerate(refs):
            _, rr, dd, _, inf = r.step(int(a[i]))
            assert rr == float(rew[i]) and dd == bool(term[i])
            if dd:
                assert float(info["heuristic_solution"][i]) == inf["heuristic_solution"], (k, i)
                seeds[i] += stride; r.reset(seed=seeds[i]); ends += 1
    assert ends >= (B if kw["n_nodes"] <= 20 else 1) and all(r.heuristic_solution > 0 for r in refs)
    env.check_device_errors()
    env.close()


def test_emulated_unseeded_reset_continues_the_streams_of_every_slot(emu):
    import oracle
    hc.check_continue_streams(ge, oracle, "cpu", emu)


def test_emulated_return_graph_obs_and_copy_outputs(emu):
    hc.check_graph_obs(ge, "cpu", emu)


def _asan_child(args, env_extra):
    import subprocess
    env = dict(os.environ, LD_PRELOAD=build_emu.asan_preload(), ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", **env_extra)
    build_emu.build(asan=True)
    return subprocess.run([sys.executable, os.path.join(os.path.dirname(build_emu.OUT), "asan_run.py"), *args], env=env, capture_output=True,
                          text=True, timeout=1500)


def test_emulated_kernels_under_address_sanitizer():
    """the kernels under AddressSanitizer (slabs and per-block LDS are heap blocks with red zones): golden replays of all
    nine envs, unseeded reset, inject + autoreset"""
    r = _asan_child([], {})
    assert r.returncode == 0 and "ASAN RUN COMPLETE" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr, r.stderr[-3000:]


def test_address_sanitizer_harness_sees_an_lds_overrun():
    """the detector detects: with 64 bytes less LDS than the launch asked for, the same run must die in AddressSanitizer"""
    r = _asan_child(["sp_n10_m20_eval"], {"GE_EMU_LDS_SHORTFALL": "64"})
    assert r.returncode != 0 and "AddressSanitizer" in r.stderr, (r.returncode, r.stderr[-2000:])
