#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): env-steps/sec, ShortestPath-v0 n=64 m=192, 65 536 envs per GPU,
random-valid-action policy generated on device, same-step autoreset ON (every reset regenerates the graph,
features and masks on the GPU, seed-exact with the reference).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus N ...        # no WORLD_SIZE in the environment: spawns the N ranks itself (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --config c3        # BASELINE config 3 alone (TSP n=128 complete, 16 384 slots); c4 = SteinerTree n=256 m=1024; c5 = mixed ragged

One JSON line on rank 0.  `value` = (envs x steps over all ranks) / max-over-ranks wall time, inputs resident
in HBM.  Whatever --warmup says, the engine first runs SETTLE untimed steps (more than the longest episode), so the
timed window is the steady state of the autoreset loop and not the transient that follows a full reset; the line
reports the resets per step inside the timed window beside a reference window and flags a non-stationary one.
`roofline` is the step kernel (HBM-bound; algorithmic bytes per env-step from SURVEY 8d) timed with HIP events on its
stream; `roofline_1m` the same kernel at 1 M slots (where the launch floor no longer hides the memory system);
`roofline_reset` the reset path's issue-rate evidence (rocprofv3 SQ counters, profiles/); `cpu_baseline` is the CPU
oracle (a port of the reference semantics, not the reference) on the host cores, rank 0, N=1 only.
`configs` (N=1, config c2 only, measured AFTER the contract's timed region): BASELINE configs 3, 4 and 5 on the same GPU, each
with its own value, step-kernel roofline and bounded cpu_baseline.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
SETTLE = 128           # untimed steps in front of the warm-up: > 2 x the longest ShortestPath episode at n = 64
PROFILE_TAGS = ("r04", "r03", "r02")

# BASELINE.json configs that fit one GPU.  algo_bytes: SURVEY.md 8(d), canonical 32-bit CSR + byte mask, per env-step.
CONFIGS = {
    # shards: the batch as that many independent engines on their own HIP streams (graphenvs_amd.sharded: c2 305 -> 353 M with two, 391 M with
    # three; c4 104 -> 107 M with two; c3 313 -> 247 M, its slots all finish in the same step -- measured, profiles/r04_shards.txt)
    "c2": dict(env_id="ShortestPath-v0", kw=dict(n_nodes=64, n_edges=192), envs=65536, algo_bytes=200, steps=200, shards=3,
               kernel="ge_k_step_path64<true, false> (fused device policy + step)",
               metric="env-steps/sec (whole node), ShortestPath-v0 n=64 m=192 batch=65536, 1/2/4/8 GPU",  # BASELINE.json's metric, verbatim
               workload="ShortestPath-v0 n_nodes=64 n_edges=192 weighted"),
    "c3": dict(env_id="TSP-v0", kw=dict(n_nodes=128, n_edges=8128, parenting=1), envs=16384, algo_bytes=1700, steps=256,
               kernel="ge_k_step<3, true> (fused device policy + step)",
               metric="env-steps/sec (whole node), TSP-v0 n=128 complete graph batch=16384 per GPU",
               workload="TSP-v0 n_nodes=128 n_edges=8128 (complete) parenting=1 weighted"),
    "c4": dict(env_id="SteinerTree-v0", kw=dict(n_nodes=256, n_edges=1024, n_dests=8), envs=16384, algo_bytes=150, steps=400, shards=2,
               kernel="ge_k_step_edge<2, true> (fused device policy + step, a quad of lanes per slot, incremental [B, 2m] mask)",
               metric="env-steps/sec (whole node), SteinerTree-v0 n=256 m=1024 n_dests=8 batch=16384 per GPU",
               workload="SteinerTree-v0 n_nodes=256 n_edges=1024 n_dests=8 weighted"),
    # mixed ragged batch: three multi-class engines (one per env id) side by side; algo_bytes per member at the mean size n = 272, m = 3n
    "c5": dict(members=(("ShortestPath-v0", {}, 200), ("MaxIndependentSet-v0", {}, 4 + 4 + 272), ("DensestSubgraph-v0", dict(parenting=1), 24 + 272 + 40)),
               envs=3 * 16384, steps=100, split={"ShortestPath-v0": 2},
               kernel="ge_k_step<ENV, true, RAGGED> (thread per slot, class looked up per slot), one launch per env id",
               metric="env-steps/sec (whole node), mixed {ShortestPath, MaxIndependentSet, DensestSubgraph} ragged n in [32,512], m = 3n, 3 x 16384 slots per GPU",
               workload="mixed ragged: 16 384 slots per env id, n ~ U{32..512} (481 size classes per id), m = 3n, one multi-class engine per id (ShortestPath: two, the slots of every size dealt alternately)"),
}


# ---------------------------------------------------------------------------------------------------------------- launcher
def launch_ranks(args, argv):
    """--gpus N > 1 without a launcher around us: start the N ranks as child processes BEFORE anything here touches the GPU (this
    process never does), one per GPU, rendezvous on 127.0.0.1; rank 0's stdout (the JSON line) is ours."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = None if rank == 0 else subprocess.DEVNULL  # one JSON line, from rank 0
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    rc = 0
    for p in procs:
        r = p.wait()
        rc = rc or r
    return rc


# ---------------------------------------------------------------------------------------------------------------- pieces
def step_kernel_us(env, reps=9, burst=5, seed0=2000):
    """average launch duration of the (fused policy +) step kernel: bursts of `burst` back-to-back launches between ONE
    pair of HIP events on the launch stream, right after a full reset, minus the bare event-pair overhead"""
    empty = sorted(env.timed_step_burst_raw_ms(0) for _ in range(9))[4]
    vals = []
    for rep in range(reps):
        env.reset(seed=seed0 + rep)
        vals.append((env.timed_step_burst_raw_ms(burst, policy_seed=2) - empty) * 1e3 / burst)
    return sorted(vals)[len(vals) // 2]


def load_profile():
    """counter evidence collected in its own rocprofv3 --pmc passes (tools/collect_profiles.sh): only quoted when it was collected
    from THESE sources (the file carries the source hash of the library it measured)"""
    from graphenvs_amd import _lib
    prof, prof_src = {}, None
    for tag in PROFILE_TAGS:
        ppath = os.path.join(ROOT, "profiles", tag + "_pmc_summary.json")
        if os.path.exists(ppath):
            cand = json.load(open(ppath))
            if cand.get("source_hash") == _lib.source_hash():
                return cand, "profiles/%s_pmc_summary.json (source hash %s, collected %s)" % (tag, cand["source_hash"], cand.get("collected"))
            if prof_src is None:
                prof_src = "none: profiles/%s_pmc_summary.json was collected from other sources (hash %s, library %s)" % (tag, cand.get("source_hash"), _lib.source_hash())
    return prof, prof_src


def cpu_baseline(name, cfg, args, budget_s):
    """the C oracle (oracle/, test infrastructure: here only as the reported CPU baseline, never in the measured path) with OpenMP on the
    host cores, on a bounded sample of the same workload"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    cores = os.cpu_count() or 1
    if name == "c5":  # the same three env ids over a spread of sizes from the same range, until the time budget is spent
        import numpy as np
        sizes = [int(n) for n in np.random.default_rng(0).integers(32, 513, 64)]
        t1 = time.perf_counter(); trans = eps = used = 0
        for n in sizes:
            for env_id, kw, _ in cfg["members"]:
                r = oracle.rollout(env_id, n_envs=cores, n_steps=20, n_threads=cores, policy_seed=1, n_nodes=n, n_edges=3 * n, **kw)
                trans += r["transitions"]; eps += r["episodes"]
            used += 1
            if time.perf_counter() - t1 > budget_s:
                break
        cdt = time.perf_counter() - t1
        return {"value": trans / cdt, "unit": "env-steps/s", "cores": cores, "kind": "port",
                "sample": "%d sizes from U{32..512} x 3 env ids x %d envs x 20 steps (autoreset on, %d episodes), C oracle with OpenMP, %.1f s" % (used, cores, eps, cdt)}
    okw = cfg["kw"]
    cpu_envs = args.cpu_envs or ((64 if name == "c2" else 2) * cores)
    cpu_steps = 1000 if name == "c2" else 200
    oracle.rollout(cfg["env_id"], n_envs=cores, n_steps=5, n_threads=cores, **okw)
    t1 = time.perf_counter(); trans = eps = rounds = 0
    while True:  # whole rollouts until the budget is spent (at least one)
        r = oracle.rollout(cfg["env_id"], n_envs=cpu_envs, n_steps=cpu_steps, n_threads=cores, policy_seed=1 + rounds, **okw)
        trans += r["transitions"]; eps += r["episodes"]; rounds += 1
        if time.perf_counter() - t1 > budget_s or name == "c2":
            break
    cdt = time.perf_counter() - t1
    return {"value": trans / cdt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d x (%d envs x %d steps) of the same workload (autoreset on, %d episodes), C oracle with OpenMP, %.1f s" % (rounds, cpu_envs, cpu_steps, eps, cdt)}


def make_c5(ge, cfg, dev, slots_per_id, rank, world):
    """BASELINE config 5: n ~ U{32..512} for every slot (every size occurs), m = 3n; one multi-class engine (one launch sequence) per id.
    Rank r of a shard owns the global slots [r * B, (r + 1) * B) of every member."""
    import numpy as np
    rng = np.random.default_rng(0)
    members = []
    # An env id whose regenerations are the longest (ShortestPath: the n x n numpy draws of its weighted graphs) runs as two engines --
    # the slots of every size dealt alternately -- so that one half's latency-bound graph kernel overlaps the other's feature kernel:
    # four engines on the device's four hardware queues (profiles/r04_shards.txt: 25.4 -> 28.5 M; five engines 22.3 M)
    labels = []
    for env_id, extra, algo in cfg["members"]:
        ns = rng.integers(32, 513, slots_per_id)
        parts = cfg.get("split", {}).get(env_id, 1)
        off = 0
        for k in range(parts):
            sub = ns[k::parts]
            sizes = [(int((sub == n).sum()), int(n), 3 * int(n)) for n in np.unique(sub)]
            members.append(ge.RaggedVectorEnv(env_id, sizes, device=dev, env_index_base=rank * slots_per_id + off, seed_stride=world * slots_per_id, **extra))
            labels.append((env_id + (" [%d/%d]" % (k + 1, parts) if parts > 1 else ""), extra, algo))
            off += len(sub)
    return ge.MixedVectorEnv(members), members, labels


def measure_c5(ge, torch, cfg, args, dev, rank, world, barrier, prof):
    per_id = (args.envs // 3) if args.envs else cfg["envs"] // 3
    mixed, members, labels = make_c5(ge, cfg, dev, per_id, rank, world)
    B = mixed.num_envs
    steps = args.steps if args.steps > 0 else cfg["steps"]
    eps = lambda: sum(int(m.g["episode"].sum()) for m in members)
    mixed.reset(seed=0)
    settle = 40  # the short episodes (MaxIndependentSet aside) have turned over several times; the refill cadence is in its cycle
    mixed.random_rollout(settle, policy_seed=1)
    mixed.random_rollout(args.warmup, policy_seed=1)
    barrier()
    ep0 = eps(); t0 = time.perf_counter()
    mixed.random_rollout(steps, policy_seed=1)
    barrier()
    dt = time.perf_counter() - t0
    episodes = eps() - ep0
    # step kernel of every member inside its own loop (HIP events on the launch stream)
    rl = []
    for m, (env_id, _, algo) in zip(members, labels):
        tm = m.timed_rollout(30, policy_seed=1)
        us = tm["step_ms"] * 1e3 / 30
        ach = algo * m.num_envs / (us * 1e-6) / 1e9
        rl.append({"member": env_id, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                   "traffic": None, "algorithmic_bytes_per_launch": algo * m.num_envs, "avg_launch_us": us,
                   "avg_launch_us_method": "HIP events around every step launch of 30 loop steps (includes ~2 us of event-pair overhead)",
                   "autoreset_ms_per_step": tm["reset_ms"] / 30, "size_classes": len(m.classes)})
    for m in members:
        m.g["work_count"]  # (slabs stay referenced until close)
    mixed.close()
    return dt, steps, B, episodes, settle, rl, [m.prefetch for m in members]


def measure_uniform(ge, torch, name, cfg, args, dev, rank, world, barrier, reduce_max, prof, prof_src, emu=None):
    """the contract's loop on one uniform engine: settle, W warm-up steps, K timed steps between barriers, then the per-kernel figures"""
    B = args.envs or cfg["envs"]
    steps = args.steps if args.steps > 0 else cfg["steps"]
    extra = dict(device="cpu", _library=emu) if emu is not None else dict(device=dev)
    shards = 1 if emu is not None else (args.shards if args.shards > 0 else cfg.get("shards", 1))
    extra0 = dict(extra)
    if shards > 1 and args.serial_shards:
        extra = dict(extra, concurrent=False)
    env = ge.make_vec(cfg["env_id"], B, shards=shards, env_index_base=rank * B, seed_stride=world * B,
                      prefetch=(None if args.prefetch < 0 else args.prefetch), **extra, **cfg["kw"])
    shards = getattr(env, "shards", 1)  # (as many as the device has streams that run beside one another)
    sharded = hasattr(env, "gather")
    total = (lambda key: int(env.gather(key).sum())) if sharded else (lambda key: int(env.t[key].sum()))  # a per-slot counter over the whole batch
    env.reset(seed=0)
    settle = (SETTLE if name == "c2" else 2 * cfg["kw"]["n_nodes"]) if emu is None else 2
    env.random_rollout(settle, policy_seed=1)          # past the transient of the synchronised start
    env.random_rollout(args.warmup, policy_seed=1)     # the contract's W untimed steps
    barrier()
    ep0 = total("episode")
    t0 = time.perf_counter()
    env.random_rollout(steps, policy_seed=1)
    barrier()
    dt = reduce_max(time.perf_counter() - t0)
    episodes = total("episode") - ep0
    assert total("tstep") == B * (steps + args.warmup + settle)
    value = world * B * steps / dt
    # reference window (outside the contract's timed region): 200 more steps, timed the same way -- the steady-state figure beside
    # a short driver window -- and their reset rate, to judge the timed window against
    REF = (200 if name == "c2" else steps) if emu is None else steps
    barrier()
    t1 = time.perf_counter()
    env.random_rollout(REF, policy_seed=1)
    barrier()
    value_ref = B * REF / (time.perf_counter() - t1)  # this rank's slots only
    ref_rate = (total("episode") - ep0 - episodes) / REF
    rate = episodes / steps
    env.check_device_errors()
    out = {
        "metric": cfg["metric"],
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%s, %d env slots per GPU, random valid actions on device, same-step autoreset "
                               "(seed-exact G(n,m)+features on device)" % (cfg["workload"], B),
                   "envs_per_gpu": B, "shards": shards, "prefetch_period": env.prefetch, "episodes_finished_per_gpu": episodes, "settle_steps": settle,
                   "parallelism": "batch shard x%d, no collective" % world},
        "resets_per_step": rate, "resets_per_step_reference_window": ref_rate,
        "window_stationary": bool(ref_rate > 0 and abs(rate / ref_rate - 1.0) <= 0.10),
        "value_200": value_ref * world, "value_200_steps": REF,
    }
    if emu is not None:  # rehearsal of the launcher on the CPU harness: the loop above is all there is to rehearse
        env.close()
        return out
    # the step kernel inside the real loop (one HIP-event pair per launch on the launch stream, `steps` launches): this is the figure
    # rocprofv3's per-kernel average of the same loop agrees with, and the one the roofline uses.  What one event pair adds when a
    # kernel sits between the two events is NOT the time of an empty pair (part of the events' own latency then overlaps the
    # kernel): it is calibrated on the kernel itself -- five launches in five pairs against five launches in one pair, same state,
    # right after a reset: (sum of singles - burst) / 4.  The back-to-back burst right after a full reset (step_kernel_us) is a few
    # percent faster -- warm caches, nothing between the launches -- and is reported beside it.
    # With shards a launch of the step kernel covers ONE shard's slots: the calibration runs on shard 0 alone, and the loop's figure
    # comes from shard 0's events while the other shards roll out beside it on their streams, as in the timed region.
    pe = env.members[0] if sharded else env
    Bk = pe.num_envs
    ovh = []
    for rep_ in range(7):
        pe.reset(seed=3000 + rep_)
        singles = sum(pe.timed_step_burst_raw_ms(1, policy_seed=2) for _ in range(5))
        pe.reset(seed=3000 + rep_)
        ovh.append((singles - pe.timed_step_burst_raw_ms(5, policy_seed=2)) * 1e3 / 4)
    pair_us = sorted(ovh)[len(ovh) // 2]
    env.reset(seed=0); env.random_rollout(settle, policy_seed=1)  # back in the steady state of the loop
    tm = env.timed_rollout(steps, policy_seed=1)
    step_us = tm["step_ms"] * 1e3 / steps - pair_us
    burst_us = step_kernel_us(pe)
    algo = cfg["algo_bytes"]
    achieved = algo * Bk / (step_us * 1e-6) / 1e9
    traffic = prof.get("step_kernel_traffic", {}).get(name)
    out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": prof_src,
                       "algorithmic_bytes_per_launch": algo * Bk, "slots_per_launch": Bk, "avg_launch_us": step_us,
                       "avg_launch_us_method": "HIP events around every launch of the timed loop, minus what an event pair adds around this kernel (%.2f us: five launches in five pairs against five in one pair)" % pair_us,
                       "avg_launch_us_burst": burst_us, "launch_floor_us": pe.launch_floor_us(),  # an EMPTY kernel of the same grid, block and LDS
                       "kernel": cfg["kernel"]}
    out["kernel_ms_per_vector_step"] = {"step": tm["step_ms"] / steps, "autoreset": tm["reset_ms"] / steps, "policy": tm["policy_ms"] / steps}
    env.close()
    if sharded:  # beside the shard's launch in the loop: the same kernel over the WHOLE batch in one launch, alone (a burst right after a reset)
        whole = ge.make_vec(cfg["env_id"], B, env_index_base=rank * B, seed_stride=world * B, prefetch=0, **extra0, **cfg["kw"])
        whole.reset(seed=0)
        w_us = step_kernel_us(whole)
        out["roofline"]["one_engine_launch"] = {"slots_per_launch": B, "algorithmic_bytes_per_launch": algo * B, "avg_launch_us_burst": w_us,
                                                "achieved": algo * B / (w_us * 1e-6) / 1e9, "frac": algo * B / (w_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                "note": "one engine's launch over the whole batch, nothing beside it; the loop's launches cover one shard and share the chip with the other shards' kernels"}
        whole.close()
    return out


# ---------------------------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 200; c3: 256 = two whole TSP episodes, every slot resets exactly "
                    "twice whatever the phase of the window; c4: 400; c5: 100)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--envs", type=int, default=0, help="env slots per GPU (default: the config's)")
    ap.add_argument("--shards", type=int, default=0, help="independent engines the batch of a GPU is split into, each on its own HIP stream (default: the config's; 1 = one engine)")
    ap.add_argument("--serial-shards", action="store_true", help="diagnostic (counter collection): the shards one after the other on one stream, so that a launch's counters are its own")
    ap.add_argument("--prefetch", type=int, default=-1, help="episode prefetch: refill period in steps, 0 = regenerate in place (default: the engine's choice for the config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-1m", action="store_true", help="skip the 1 M-slot step-kernel roofline leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (BASELINE configs 3-5 behind the headline line)")
    ap.add_argument("--cpu-envs", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL); gloo only for rehearsals")
    ap.add_argument("--share-device", action="store_true", help="rehearsal on a one-GPU box: every rank uses cuda:0 (rendezvous over gloo: RCCL refuses two ranks on one device)")
    ap.add_argument("--emu", action="store_true", help="rehearsal without a GPU (tests only): the ranks drive the CPU sanitizer harness of the kernels; not a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d: the line would report the wrong n_gpus" % (args.gpus, world)

    import torch
    import torch.distributed as dist

    distributed = world > 1
    emu = None
    if args.emu:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import build_emu
        emu = build_emu.load()
        args.backend = "gloo"
    if args.share_device:
        local_rank = 0
        if args.backend == "nccl":
            args.backend = "gloo"
    if emu is None:
        torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    import graphenvs_amd as ge

    dev = f"cuda:{local_rank}"

    def barrier():
        if emu is None:
            torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        if emu is None:
            torch.cuda.synchronize()

    def reduce_max(dt):
        if not distributed:
            return dt
        tt = torch.tensor([dt], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    prof, prof_src = load_profile() if emu is None else ({}, None)
    cfg = CONFIGS[args.config]
    single = rank == 0 and world == 1 and emu is None

    def c5_line(sub_args):
        dt, steps, B, episodes, settle, rl, pf = measure_c5(ge, torch, CONFIGS["c5"], sub_args, dev, rank, world, barrier, prof)
        dt = reduce_max(dt)
        c = CONFIGS["c5"]
        return {"metric": c["metric"], "value": world * B * steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": steps, "warmup": sub_args.warmup,
                "ms_per_step": dt * 1e3 / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": c["workload"], "envs_per_gpu": B, "prefetch_period": pf, "episodes_finished_per_gpu": episodes, "settle_steps": settle,
                           "parallelism": "batch shard x%d, no collective" % world, "launch_sequences_per_step": len(rl), "engines": [r["member"] for r in rl]},
                "resets_per_step": episodes / steps,
                "roofline": max(rl, key=lambda r: r["avg_launch_us"]), "roofline_members": rl}

    if args.config == "c5":
        out = c5_line(args)
    else:
        out = measure_uniform(ge, torch, args.config, cfg, args, dev, rank, world, barrier, reduce_max, prof, prof_src, emu)
    if "reset_path" in prof and args.config == "c2":
        out["roofline_reset"] = prof["reset_path"]  # SQ-counter issue rates of the feature / graph kernels (tools/pmc_sq_passes.sh)
        out["roofline_reset_source"] = prof_src
    if single and args.config == "c2" and not args.no_1m:
        B1 = 1 << 20
        big = ge.make_vec(cfg["env_id"], B1, device=dev, **cfg["kw"])
        big.reset(seed=0)
        us1 = step_kernel_us(big, reps=5)
        ach1 = cfg["algo_bytes"] * B1 / (us1 * 1e-6) / 1e9
        out["roofline_1m"] = {"bound": "hbm", "achieved": ach1, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach1 / HBM_PEAK_GBS,
                              "traffic": prof.get("step_kernel_traffic", {}).get("c2_1m"), "algorithmic_bytes_per_launch": cfg["algo_bytes"] * B1,
                              "avg_launch_us": us1, "launch_floor_us": big.launch_floor_us(), "slots": B1, "kernel": cfg["kernel"]}
        big.close()
        del big
    # (every CPU leg comes after every GPU measurement: the OpenMP team of the C oracle keeps its threads spinning for a while after a
    # parallel region, and a GPU loop timed right behind it -- a thousand launches from one host thread -- ran five times slower once)
    if single and args.config == "c2" and not args.no_configs:
        # BASELINE configs 3, 4, 5 on the same GPU, after (and outside) the contract's timed region: each the same loop at its own
        # default sizes, with the step-kernel roofline and a bounded cpu_baseline of its own
        block = {}
        for name in ("c3", "c4", "c5"):
            sub = argparse.Namespace(**vars(args)); sub.steps = 0; sub.envs = 0; sub.prefetch = -1; sub.shards = 0; sub.config = name
            try:
                torch.cuda.empty_cache()
                block[name] = c5_line(sub) if name == "c5" else measure_uniform(ge, torch, name, CONFIGS[name], sub, dev, rank, world, barrier, reduce_max, prof, prof_src)
            except Exception as ex:  # the headline line is not hostage to a side measurement: say what failed
                block[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        out["configs"] = block
    if single and not args.no_cpu_baseline:
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        out["cpu_baseline"] = cpu_baseline(args.config, cfg, args, budget_s=10.0)
        for name, line in out.get("configs", {}).items():
            if "error" not in line:
                sub = argparse.Namespace(**vars(args)); sub.config = name
                line["cpu_baseline"] = cpu_baseline(name, CONFIGS[name], sub, budget_s=6.0)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
