"""CPU: the C oracle (oracle/ge_oracle.c) against the golden vectors captured from the real
reference (oracle/gen_golden.py), plus the known answers of SURVEY.md section 10 and the two
MT19937 streams against CPython / numpy themselves."""
import random

import numpy as np
import pytest

import golden_util as gu
import oracle


@pytest.mark.parametrize("name", gu.case_names())
def test_oracle_matches_reference_golden(name):
    case = gu.load_case(name)
    stats = gu.replay_case(case, lambda env_id, **kw: oracle.OracleEnv(env_id, **kw))
    assert stats["resets"] > 0


def test_spatial_tsp_fixture_sits_on_the_pow_boundary():
    """tsp.py:85 squares with float ** 2 (libm pow); the checker multiplies.  On the seeds of this fixture the two differ in the last
    bit of an edge weight the policies traverse: rewards are equal within 1e-12 relative, not exactly (golden_util.replay_case)"""
    st = gu.replay_case(gu.load_case("tsp_n12_m30_p1_spatial_pow2"), lambda env_id, **kw: oracle.OracleEnv(env_id, **kw))
    assert st["inexact_rewards"] > 0
    for other in ("tsp_n12_m30_p2_spatial", "tsp_n12_m30_p2_spatial_eval", "tsp_n600_m2000_p1_spatial"):  # (the other spatial fixtures are exact)
        assert gu.replay_case(gu.load_case(other), lambda env_id, **kw: oracle.OracleEnv(env_id, **kw))["inexact_rewards"] == 0


@pytest.mark.parametrize("name", ["sp_n10_m20_eval", "st_n10_m20_d3_eval", "ds_n10_m20_p1", "mis_n6_m8", "tsp_n10_m20_p2"])
def test_oracle_continues_streams_on_unseeded_reset(name):
    """reset(seed=None) continues the MT19937 streams (shortest_path.py:49-52)."""
    case = gu.load_case(name)
    meta = case["meta"]
    for si, seed in enumerate(case["seeds"]):
        env = oracle.OracleEnv(meta["env_id"], **meta["kwargs"])
        env.reset(seed=int(seed))
        T = int(case["first_length"][si])
        for t in range(T):
            env.step(int(case["first_actions"][si, t]))
        obs2, info2 = env.reset()
        assert gu.sha64(obs2) == case["first_reset2_obs_sha"][si]
        assert np.array_equal(info2["mask"], case["first_reset2_mask"][si])


def test_survey_known_answers():
    """SURVEY.md section 10, seed 0, lowest-valid-index policy."""
    import hashlib
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
    e = oracle.OracleEnv("ShortestPath-v0", n_nodes=10, n_edges=20, is_eval_env=True)
    obs, info = e.reset(seed=0)
    assert len(obs) == 190 and sha(obs) == "c2040d7d83b56476" and sha(e.edge_links()) == "560e8ee18327c253"
    assert list(e.terminals()) == [2, 4]
    assert "".join(str(int(b)) for b in info["mask"]) == "0001100011"
    rs = []
    for a in (3, 1, 4):
        obs, r, d, _, info = e.step(a)
        rs.append(r)
    assert rs == [-0.8, -0.3, -0.7] and d and info["solved"] is True
    assert info["heuristic_solution"] == 0.3 and info["solution_cost"] == 1.8 and sha(obs) == "39737d0aab663239"
    assert "".join(str(int(b)) for b in info["mask"]) == "1000001010"

    e = oracle.OracleEnv("SteinerTree-v0", n_nodes=10, n_edges=20, n_dests=9, is_eval_env=True)
    obs, info = e.reset(seed=0)
    assert sha(obs) == "85f8267b9b5dd68f" and list(e.terminals()) == [2, 4, 6, 3, 9, 1, 0, 8, 5, 7]
    for a in (8, 9, 10, 11, 13, 14, 6, 16, 21):
        obs, r, d, _, info = e.step(a)
    assert d and info["heuristic_solution"] == 3.2 and abs(info["solution_cost"] - 4.3) < 1e-6 and not info["mask"].any()

    e = oracle.OracleEnv("TSP-v0", n_nodes=10, n_edges=20, parenting=2)
    obs, info = e.reset(seed=0)
    assert len(obs) == 210 and sha(obs) == "33286effa4e58caf"
    assert "".join(str(int(b)) for b in info["mask"]) == "0000100011"
    for a in (4, 1, 3, 9, 2, 8, 7, 5, 6):
        obs, r, d, _, info = e.step(a)
    assert d and info["solved"] is False and abs(r + 20.9) < 1e-12 and abs(info["solution_cost"] - 5.3) < 1e-12

    e = oracle.OracleEnv("DensestSubgraph-v0", n_nodes=10, n_edges=20, parenting=1)
    obs, info = e.reset(seed=0)
    assert len(obs) == 180 and sha(obs) == "c93b059b99574f1a" and sha(e.edge_links()) == "f328476caf411c60"
    rs = [e.step(a)[1] for a in (0, 4, 1)]
    assert rs[0] == 0 and rs[1] == 0.5 and abs(rs[2] - 1 / 6) < 1e-15
    assert "".join(str(int(b)) for b in e.mask()) == "0011011110"


@pytest.mark.parametrize("seed", [0, 1, 7, 12345, 2**32 - 1])
def test_mt19937_streams_match_cpython_and_numpy(seed):
    """python random.seed(int) = init_by_array([s]); np.random.seed(int) = init_genrand(s)."""
    py = oracle.mt_stream("py", seed, 1500)
    random.seed(seed)
    assert [random.getrandbits(32) for _ in range(1500)] == py.tolist()
    npy = oracle.mt_stream("np", seed, 1500)
    rs = np.random.RandomState(seed)
    want = rs.randint(0, 2**32, size=1500, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(npy, want)


def test_policy_pick_is_uniform_and_deterministic():
    mask = np.zeros(64, dtype=np.uint8)
    mask[[3, 17, 40, 63]] = 1
    picks = [oracle.policy_pick(mask, 42, 5, t) for t in range(4000)]
    assert picks == [oracle.policy_pick(mask, 42, 5, t) for t in range(4000)]
    counts = np.bincount(picks, minlength=64)
    assert set(np.nonzero(counts)[0]) == {3, 17, 40, 63} and counts[counts > 0].min() > 850
    assert oracle.policy_pick(np.zeros(8, dtype=np.uint8), 1, 2, 3) == -1


def test_rollout_counts_transitions():
    out = oracle.rollout("ShortestPath-v0", n_envs=8, n_steps=50, n_nodes=10, n_edges=20, n_threads=2)
    assert out["transitions"] == 400 and out["episodes"] > 8


def test_pyset_order_matches_cpython():
    """the multicast baseline sums a python set of (u, v) tuples: the oracle's CPython set emulation against set() itself"""
    rng = random.Random(5)
    for _ in range(200):
        n = rng.choice([5, 10, 64, 200, 512])
        pairs = [(rng.randrange(n), rng.randrange(n)) for _ in range(rng.randint(0, min(600, n * 3)))]
        s = set()
        for p in pairs:
            s.add(p)
        assert list(s) == oracle.pyset_order(pairs)


# ---- own baselines (validity and bounds instead of bit parity with networkx's Kou / Christofides / clique removal)
def _nx_graph(env):
    import networkx as nx
    G = nx.Graph()
    G.add_nodes_from(range(env.n))
    links, w = env.edge_links(), env.edges()[:, 0]
    for (u, v), d in zip(links, w):
        G.add_edge(int(u), int(v), delay=float(np.float32(d)))
    return G


def test_int_set_restatement_matches_the_interpreter():
    """graphenvs_amd/csrc/ge_clique_removal.h restates CPython's setobject.c for int keys: iteration order of set(list), and of
    `d.keys() - a.keys() - {x}` as nx.non_neighbors builds it (PySet_New(dict), difference_update, set.__sub__: linear probes,
    dummies, the resize rules, set_merge's copy paths), against the running interpreter's own sets"""
    import random
    rng = random.Random(5)
    for t in range(1500):
        n = rng.choice([3, 8, 20, 64, 130, 512, 1000, 4000])
        keys = rng.sample(range(n), rng.randint(0, min(n, rng.choice([5, 10, 40, 200, 2000]))))
        assert list(set(keys)) == oracle.pyset_int_order(keys), (t, len(keys))
    for t in range(2500):
        n = rng.choice([3, 6, 9, 17, 33, 64, 100, 257, 600])
        k = rng.randint(1, n)
        nodes = rng.sample(range(n), k)
        if rng.random() < 0.5:
            nodes.sort()
        deg = rng.randint(0, k - 1) if rng.random() < 0.7 else rng.randint(max(0, k - 3), k - 1)
        adj = rng.sample(nodes[1:], deg)
        want = list(dict.fromkeys(nodes).keys() - dict.fromkeys(adj).keys() - {nodes[0]})
        assert want == oracle.non_neighbors_order(nodes, adj), (t, k, deg)


def test_mis_baseline_is_networkx_clique_removal_exactly():
    """len(nx.approximation.maximum_independent_set(G)) (max_independent_set.py:63-67) on random graphs, sparse to dense: the value
    hangs on dict orders of graph copies and on set iteration orders, all restated"""
    nx = pytest.importorskip("networkx")
    import random
    rng = random.Random(7)
    for t in range(120):
        n = rng.choice([5, 8, 12, 20, 33, 64, 90, 130])
        m = rng.randint(n - 1, min(n * (n - 1) // 2, rng.choice([2, 3, 4, 8]) * n))
        G = nx.gnm_random_graph(n, m, seed=t)
        rp, col = [0], []
        for v in range(n):
            col += list(G.adj[v]); rp.append(len(col))
        assert oracle.clique_removal_len(n, rp, col) == len(nx.approximation.maximum_independent_set(G)), (t, n, m)


@pytest.mark.parametrize("seed", range(4))
def test_mis_env_baseline_equals_networkx_on_the_env_graph(seed):
    """the env's heuristic_solution against networkx run on a graph with the env graph's node and adjacency dict orders"""
    nx = pytest.importorskip("networkx")
    n = 40
    env = oracle.OracleEnv("MaxIndependentSet-v0", n_nodes=n, n_edges=100, weighted=False, is_eval_env=True)
    env.reset(seed=seed)
    rows = {v: [] for v in range(n)}
    for a, b in env.edge_links():  # row-major by source, insertion-order columns
        rows[int(a)].append(int(b))
    H = nx.Graph()
    H.add_nodes_from(range(n))
    H._adj = {v: {w: {} for w in rows[v]} for v in range(n)}  # the adjacency dicts in exactly these orders
    rp, col = [0], []
    for v in range(n):
        col += rows[v]; rp.append(len(col))
    assert env.heuristic_solution == len(nx.approximation.maximum_independent_set(H)) == oracle.clique_removal_len(n, rp, col)


def test_steiner_baseline_is_networkx_kou_exactly():
    """sum of delays over nx...steiner_tree(G, dests, weight='delay', method='kou') (steiner_tree.py:84-87), float64 sum order
    included, on random connected graphs (weighted in tenths, and unweighted: every tie there is)"""
    nx = pytest.importorskip("networkx")
    import math
    import random
    rng = random.Random(3); nrng = np.random.RandomState(3)
    for t in range(150):
        n = rng.choice([6, 10, 14, 20, 30, 40, 64, 100])
        m = min(rng.randint(int(0.8 * n * math.log(n)) + 2, int(1.6 * n * math.log(n)) + 4), n * (n - 1) // 2)
        while True:
            G = nx.gnm_random_graph(n, m, seed=rng.randint(0, 10 ** 9))
            if nx.is_connected(G):
                break
        delay = nrng.randint(3, 10, size=(n, n)) / 10.0 if t % 5 else np.ones((n, n))
        for u, v, d in G.edges(data=True):
            d["delay"] = delay[u, v]
        dests = nrng.choice(n, rng.randint(3, max(3, n - 2)), replace=False)
        T = nx.algorithms.approximation.steinertree.steiner_tree(G, dests, weight="delay", method="kou")
        ref = sum([G[u][v]["delay"] for u, v in T.edges()])
        rp, col, w = [0], [], []
        for v in range(n):
            for u in G.adj[v]:
                col.append(u); w.append(G[v][u]["delay"])
            rp.append(len(col))
        assert oracle.kou_exact(n, rp, col, w, [int(x) for x in dests]) == ref, (t, n, m)


@pytest.mark.parametrize("seed", range(4))
def test_steiner_env_baseline_equals_networkx_on_the_env_graph(seed):
    """the env's heuristic_solution against networkx run on a graph with the env graph's node / adjacency dict orders and weights"""
    nx = pytest.importorskip("networkx")
    n = 24
    env = oracle.OracleEnv("SteinerTree-v0", n_nodes=n, n_edges=60, n_dests=6, is_eval_env=True)
    env.reset(seed=seed)
    H = nx.Graph()
    H.add_nodes_from(range(n))
    for (a, b), d in zip(env.edge_links(), env.edges()[:, 0]):  # row-major by source, insertion-order columns
        H._adj[int(a)][int(b)] = H._adj[int(b)].get(int(a), {"delay": round(float(d), 1)})
    T = nx.algorithms.approximation.steinertree.steiner_tree(H, np.array([int(t) for t in env.terminals()]), weight="delay", method="kou")
    assert env.heuristic_solution == sum([H[u][v]["delay"] for u, v in T.edges()])


@pytest.mark.parametrize("seed", range(3))
def test_own_greedy_mis_fallback_is_a_maximal_independent_set(seed):
    """the min-degree greedy set of round 1 stays as the value reported if the clique-removal restatement's work space were too small"""
    import itertools
    env = oracle.OracleEnv("MaxIndependentSet-v0", n_nodes=12, n_edges=20, weighted=False, is_eval_env=True)
    env.reset(seed=seed)
    size, member = env.debug_greedy_mis()
    G = _nx_graph(env)
    S = set(np.nonzero(member)[0].tolist())
    assert size == len(S)
    assert not any(G.has_edge(u, v) for u, v in itertools.combinations(S, 2))            # independent
    assert all(any(G.has_edge(v, u) for u in S) for v in G if v not in S)                 # maximal


@pytest.mark.parametrize("seed", range(3))
def test_own_steiner_fallback_is_a_tree_over_the_terminals_within_twice_the_optimum(seed):
    """the Kou-style tree of round 1 stays as the value reported if the exact restatement's work space were ever too small"""
    import itertools
    import networkx as nx
    env = oracle.OracleEnv("SteinerTree-v0", n_nodes=10, n_edges=18, n_dests=3, is_eval_env=True)
    env.reset(seed=seed)
    cost, adj = env.debug_steiner_tree()
    G = _nx_graph(env)
    terms = [int(t) for t in env.terminals()]
    T = nx.Graph([(u, v) for u in range(10) for v in range(u + 1, 10) if adj[u, v]])
    assert all(G.has_edge(u, v) for u, v in T.edges) and nx.is_tree(T) and all(t in T for t in terms)
    assert all(T.degree(v) > 1 for v in T if v not in terms)                                # no non-terminal leaf
    assert abs(cost - sum(G[u][v]["delay"] for u, v in T.edges)) < 1e-6
    others = [v for v in range(10) if v not in terms]
    opt = min(nx.minimum_spanning_tree(G.subgraph(terms + list(extra)), weight="delay").size(weight="delay")
              for k in range(len(others) + 1) for extra in itertools.combinations(others, k)
              if nx.is_connected(G.subgraph(terms + list(extra))))
    assert opt - 1e-6 <= cost <= 2 * opt + 1e-6


def _subset_dp_matching(d):
    """minimum weight of a perfect matching by dynamic programming over subsets (the lowest free vertex is matched next)"""
    k = d.shape[0]
    f = {0: 0}
    for mask in range(1 << k):
        if mask not in f:
            continue
        i = next((b for b in range(k) if not (mask >> b) & 1), None)
        if i is None:
            continue
        for j in range(i + 1, k):
            if not (mask >> j) & 1:
                m2, v = mask | (1 << i) | (1 << j), f[mask] + int(d[i, j])
                if v < f.get(m2, 1 << 62):
                    f[m2] = v
    return f[(1 << k) - 1]


def test_tsp_baseline_matching_is_a_minimum_weight_perfect_matching():
    """the blossom matching of the TSP baseline (graphenvs_amd/csrc/ge_christofides.h, compiled into the checker) against a
    subset DP: tie-heavy weights (many blossoms), weight codes, wide ranges, Manhattan metrics"""
    rng = np.random.default_rng(1)
    for trial in range(600):
        k = int(rng.choice([2, 4, 6, 8, 10, 12]))
        kind = trial % 4
        if kind == 0:
            d = rng.integers(1, 4, size=(k, k))
        elif kind == 1:
            d = rng.integers(3, 10, size=(k, k))
        elif kind == 2:
            d = rng.integers(1, 100000, size=(k, k))
        else:
            pts = rng.integers(0, 6, size=(k, 2)); d = np.abs(pts[:, None, :] - pts[None, :, :]).sum(-1) + 1
        d = np.triu(d, 1); d = d + d.T
        w, m = oracle.min_matching(d)
        assert all(m[m[i]] == i and m[i] != i for i in range(k)), (trial, m)        # perfect
        assert w == sum(int(d[i, m[i]]) for i in range(k)) // 2 == _subset_dp_matching(d), (trial, k)


def test_tsp_baseline_matching_agrees_with_networkx_on_larger_graphs():
    nx = pytest.importorskip("networkx")
    rng = np.random.default_rng(3)
    for k in (20, 40, 64, 100):
        d = rng.integers(3, 60, size=(k, k)); d = np.triu(d, 1); d = d + d.T
        w, m = oracle.min_matching(d)
        G = nx.Graph()
        G.add_weighted_edges_from((i, j, int(d[i, j])) for i in range(k) for j in range(i + 1, k))
        assert w == sum(int(d[u, v]) for u, v in nx.min_weight_matching(G)), k


def test_tsp_baseline_tour_is_within_three_halves_of_the_optimum():
    """Christofides' guarantee against exhaustive search on the metric closure of random connected graphs, n <= 9"""
    import itertools
    rng = np.random.default_rng(2)
    worst = 0.0
    for trial in range(150):
        n = int(rng.integers(3, 10))
        D = np.full((n, n), 10 ** 8, dtype=np.int64); np.fill_diagonal(D, 0)
        edges = [(i, int(rng.integers(0, i))) for i in range(1, n)] + [tuple(int(x) for x in rng.integers(0, n, 2)) for _ in range(int(rng.integers(0, 2 * n)))]
        for u, v in edges:
            if u != v:
                D[u, v] = D[v, u] = min(D[u, v], int(rng.integers(3, 10)))
        for k in range(n):
            D = np.minimum(D, D[:, k:k + 1] + D[k:k + 1, :])
        tour = oracle.christofides_units(D.astype(np.int32))
        opt = min(sum(int(D[p[i], p[(i + 1) % n]]) for i in range(n)) for p in ([0] + list(q) for q in itertools.permutations(range(1, n))))
        assert opt <= tour and 2 * tour <= 3 * opt, (trial, n, tour, opt)
        worst = max(worst, tour / opt)
    assert worst > 1.0  # the instances are not all trivial


@pytest.mark.parametrize("kw", [dict(n_nodes=10, n_edges=20, parenting=1), dict(n_nodes=12, n_edges=30, parenting=2, spatial=True),
                                dict(n_nodes=9, n_edges=36, parenting=1, weighted=False)])
def test_tsp_baseline_against_networkx_christofides_on_env_graphs(kw):
    """the env's heuristic_solution next to the reference's call (tsp.py:114-117) on the same graph: two Christofides tours of one
    metric closure -- OPT <= both <= 1.5 OPT -- and never worse than the double-tree walk 2 x MST"""
    nx = pytest.importorskip("networkx")
    for seed in range(4):
        env = oracle.OracleEnv("TSP-v0", is_eval_env=True, **kw)
        env.reset(seed=seed)
        G = _nx_graph(env)
        for u, v, d in G.edges(data=True):
            d["weight"] = d["delay"]  # networkx builds the closure with the attribute name 'weight' whatever it is asked for
        cyc = nx.approximation.traveling_salesman_problem(G, weight="weight", cycle=True)
        ref = sum(G[u][v]["weight"] for u, v in zip(cyc, cyc[1:]))
        ours = env.heuristic_solution
        mst = nx.minimum_spanning_tree(G, weight="delay").size(weight="delay")
        assert ref / 1.5 - 1e-4 <= ours <= 1.5 * ref + 1e-4, (seed, ours, ref)
        assert mst - 1e-4 <= ours <= 2 * mst + 1e-4, (seed, ours, mst)
