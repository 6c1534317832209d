"""Shared body: ge_inject_state (parity path of SURVEY 7) -- post-reset states produced by the oracle are loaded into
the engine, then both are stepped with the same actions."""
import numpy as np
import torch


def wcodes_from_edges(edges_f32):
    """weight code k with float32(k/10.0) == edge weight (k = 10 for 1.0)"""
    lut = {np.float32(k / 10.0): k for k in range(3, 11)}
    return np.array([lut[np.float32(w)] for w in edges_f32], dtype=np.uint8)


def check_inject(ge, oracle, device, env_id, kw, B=6, steps=30, library=None):
    extra = dict(device=device, _library=library) if library is not None else dict(device=device)
    refs = [oracle.OracleEnv(env_id, **kw) for _ in range(B)]
    for i, r in enumerate(refs):
        r.reset(seed=500 + i)
    env = ge.VectorGraphEnv(env_id, B, autoreset=False, obs_mode="flat", **extra, **kw)
    links = np.stack([r.edge_links() for r in refs])
    wcode = np.stack([wcodes_from_edges(r.edges()[:, 0]) for r in refs])
    x = np.stack([r.nodes() for r in refs])
    T = env.T
    terms = None
    if env_id in ("ShortestPath-v0", "LongestPath-v0", "SteinerTree-v0", "MulticastRouting-v0", "PerishableProductDelivery-v0"):
        terms = np.stack([np.pad(r.terminals(), (0, T - len(r.terminals()))) for r in refs]).astype(np.int32)
    if env_id == "DistributionCenter-v0":  # the targets, in the order they were drawn; unused tail = -1
        tc = refs[0].cfg.n_dests
        terms = np.stack([np.pad(r.terminals()[:tc], (0, T - tc), constant_values=-1) for r in refs]).astype(np.int32)
    obs, info = env.inject_state(links, wcode, x, terms)
    assert np.array_equal(obs.cpu().numpy(), np.stack([r.obs() for r in refs]))
    assert np.array_equal(info["mask"].cpu().numpy(), np.stack([r.mask() for r in refs]))
    alive = [True] * B
    for k in range(steps):
        a = env.sample_random_actions(policy_seed=3).clone().cpu().numpy()
        obs, rew, term, trunc, info = env.step(torch.from_numpy(a).to(device))
        rew, term, mask = rew.cpu().numpy(), term.cpu().numpy(), info["mask"].cpu().numpy()
        for i, r in enumerate(refs):
            if not alive[i]:
                assert a[i] == -1
                continue
            _, rr, dd, _, _ = r.step(int(a[i]))
            assert rr == rew[i] and dd == bool(term[i]), (k, i)
            assert np.array_equal(mask[i], r.mask()), (k, i)
            alive[i] = not dd
        if not any(alive):
            break
    assert np.array_equal(env.flat_obs().cpu().numpy(), np.stack([r.obs() for r in refs]))
    env.close()
