#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): env-steps/sec, ShortestPath-v0 n=64 m=192, 65 536 envs per GPU,
random-valid-action policy generated on device, same-step autoreset ON (every reset regenerates the graph,
features and masks on the GPU, seed-exact with the reference).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --config c3        # BASELINE config 3 (TSP n=128 complete, 16 384 slots); c4 = SteinerTree n=256 m=1024

One JSON line on rank 0.  `value` = (envs x steps over all ranks) / max-over-ranks wall time, inputs resident
in HBM.  Whatever --warmup says, the engine first runs SETTLE untimed steps (more than the longest episode), so the
timed window is the steady state of the autoreset loop and not the transient that follows a full reset; the line
reports the resets per step inside the timed window beside a reference window and flags a non-stationary one.
`roofline` is the step kernel (HBM-bound; algorithmic bytes per env-step from SURVEY 8d) timed with HIP events on its
stream; `roofline_1m` the same kernel at 1 M slots (where the launch floor no longer hides the memory system);
`roofline_reset` the reset path's issue-rate evidence (rocprofv3 SQ counters, profiles/); `cpu_baseline` is the CPU
oracle (a port of the reference semantics, not the reference) on the host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
SETTLE = 128           # untimed steps in front of the warm-up: > 2 x the longest ShortestPath episode at n = 64

# BASELINE.json configs that fit one GPU.  algo_bytes: SURVEY.md 8(d), canonical 32-bit CSR + byte mask, per env-step.
CONFIGS = {
    "c2": dict(env_id="ShortestPath-v0", kw=dict(n_nodes=64, n_edges=192), envs=65536, algo_bytes=200,
               kernel="ge_k_step_path64<true, false> (fused device policy + step)",
               metric="env-steps/sec (whole node), ShortestPath-v0 n=64 m=192 batch=65536, 1/2/4/8 GPU",  # BASELINE.json's metric, verbatim
               workload="ShortestPath-v0 n_nodes=64 n_edges=192 weighted"),
    "c3": dict(env_id="TSP-v0", kw=dict(n_nodes=128, n_edges=8128, parenting=1), envs=16384, algo_bytes=1700,
               kernel="ge_k_step<3, true> (fused device policy + step)",
               metric="env-steps/sec (whole node), TSP-v0 n=128 complete graph batch=16384 per GPU",
               workload="TSP-v0 n_nodes=128 n_edges=8128 (complete) parenting=1 weighted"),
    "c4": dict(env_id="SteinerTree-v0", kw=dict(n_nodes=256, n_edges=1024, n_dests=8), envs=16384, algo_bytes=150,
               kernel="ge_k_step_edge<2, true> (fused device policy + step, a quad of lanes per slot, incremental [B, 2m] mask)",
               metric="env-steps/sec (whole node), SteinerTree-v0 n=256 m=1024 n_dests=8 batch=16384 per GPU",
               workload="SteinerTree-v0 n_nodes=256 n_edges=1024 n_dests=8 weighted"),
}


def step_kernel_us(env, reps=9, burst=5, seed0=2000):
    """average launch duration of the (fused policy +) step kernel: bursts of `burst` back-to-back launches between ONE
    pair of HIP events on the launch stream, right after a full reset, minus the bare event-pair overhead"""
    empty = sorted(env.timed_step_burst_raw_ms(0) for _ in range(9))[4]
    vals = []
    for rep in range(reps):
        env.reset(seed=seed0 + rep)
        vals.append((env.timed_step_burst_raw_ms(burst, policy_seed=2) - empty) * 1e3 / burst)
    return sorted(vals)[len(vals) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 200; c3: 256 = two whole TSP episodes, every slot resets exactly "
                    "twice whatever the phase of the window; c4: 400)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--envs", type=int, default=0, help="env slots per GPU (default: the config's)")
    ap.add_argument("--prefetch", type=int, default=-1, help="episode prefetch: refill period in steps, 0 = regenerate in place (default: the engine's choice for the config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-1m", action="store_true", help="skip the 1 M-slot step-kernel roofline leg")
    ap.add_argument("--cpu-envs", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL); gloo only for rehearsals")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    import graphenvs_amd as ge

    cfg = CONFIGS[args.config]
    if args.steps <= 0:
        args.steps = {"c2": 200, "c3": 256, "c4": 400}[args.config]
    B = args.envs or cfg["envs"]
    dev = f"cuda:{local_rank}"
    env = ge.make_vec(cfg["env_id"], B, device=dev, env_index_base=rank * B, seed_stride=world * B,
                      prefetch=(None if args.prefetch < 0 else args.prefetch), **cfg["kw"])
    env.reset(seed=0)
    settle = SETTLE if args.config == "c2" else 2 * cfg["kw"]["n_nodes"]
    env.random_rollout(settle, policy_seed=1)          # past the transient of the synchronised start
    env.random_rollout(args.warmup, policy_seed=1)     # the contract's W untimed steps

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    ep0 = int(env.t["episode"].sum())
    t0 = time.perf_counter()
    env.random_rollout(args.steps, policy_seed=1)
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([dt], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    episodes = int(env.t["episode"].sum()) - ep0
    assert int(env.t["tstep"].sum()) == B * (args.steps + args.warmup + settle)
    value = world * B * args.steps / dt
    # reference window (outside the contract's timed region): 200 more steps, timed the same way -- the steady-state figure beside
    # a short driver window -- and their reset rate, to judge the timed window against
    REF = 200 if args.config == "c2" else args.steps
    barrier()
    t1 = time.perf_counter()
    env.random_rollout(REF, policy_seed=1)
    torch.cuda.synchronize()
    value_ref = B * REF / (time.perf_counter() - t1)  # this rank's slots only
    ref_rate = (int(env.t["episode"].sum()) - ep0 - episodes) / REF
    rate = episodes / args.steps
    env.check_device_errors()

    # per-kernel time, HIP events on the launch stream.  (1) every launch of the real loop bracketed by its own event
    # pair (carries ~3 us of event overhead per bracket); (2) the step kernel alone (step_kernel_us), the figure that
    # agrees with rocprofv3's kernel trace (profiles/) and is used for the roofline.
    # the step kernel inside the real loop (one HIP-event pair per launch on the launch stream, `args.steps` launches, minus the bare
    # event-pair overhead): this is the figure rocprofv3's per-kernel average of the same loop agrees with, and the one the roofline
    # uses.  The back-to-back burst right after a full reset (step_kernel_us) is a few percent faster -- warm caches, nothing between
    # the launches -- and is reported beside it.
    # What one event pair adds when a kernel sits between the two events is NOT the time of an empty pair (part of the events'
    # own latency then overlaps the kernel): it is calibrated on the kernel itself -- five launches in five pairs against five
    # launches in one pair, same state, right after a reset: (sum of singles - burst) / 4.
    ovh = []
    for rep_ in range(7):
        env.reset(seed=3000 + rep_)
        singles = sum(env.timed_step_burst_raw_ms(1, policy_seed=2) for _ in range(5))
        env.reset(seed=3000 + rep_)
        ovh.append((singles - env.timed_step_burst_raw_ms(5, policy_seed=2)) * 1e3 / 4)
    pair_us = sorted(ovh)[len(ovh) // 2]
    env.reset(seed=0); env.random_rollout(settle, policy_seed=1)  # back in the steady state of the loop
    tm = env.timed_rollout(args.steps, policy_seed=1)
    loop_us = tm["step_ms"] * 1e3 / args.steps - pair_us
    burst_us = step_kernel_us(env)
    step_us = loop_us
    algo = cfg["algo_bytes"]
    achieved = algo * B / (step_us * 1e-6) / 1e9
    # counter evidence collected in its own rocprofv3 --pmc passes (tools/collect_profiles.sh): only quoted when it was collected
    # from THESE sources (the file carries the source hash of the library it measured)
    from graphenvs_amd import _lib
    prof, prof_src = {}, None
    for tag in ("r03", "r02"):
        ppath = os.path.join(ROOT, "profiles", tag + "_pmc_summary.json")
        if os.path.exists(ppath):
            cand = json.load(open(ppath))
            if cand.get("source_hash") == _lib.source_hash():
                prof, prof_src = cand, "profiles/%s_pmc_summary.json (source hash %s, collected %s)" % (tag, cand["source_hash"], cand.get("collected"))
                break
            if prof_src is None:
                prof_src = "none: profiles/%s_pmc_summary.json was collected from other sources (hash %s, library %s)" % (tag, cand.get("source_hash"), _lib.source_hash())
    traffic = prof.get("step_kernel_traffic", {}).get(args.config)

    out = {
        "metric": cfg["metric"],
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%s, %d env slots per GPU, random valid actions on device, same-step autoreset "
                               "(seed-exact G(n,m)+features on device)" % (cfg["workload"], B),
                   "envs_per_gpu": B, "prefetch_period": env.prefetch, "episodes_finished_per_gpu": episodes, "settle_steps": settle,
                   "parallelism": "batch shard x%d, no collective" % world},
        "resets_per_step": rate, "resets_per_step_reference_window": ref_rate,
        "window_stationary": bool(ref_rate > 0 and abs(rate / ref_rate - 1.0) <= 0.10),
        "value_200": value_ref * world, "value_200_steps": REF,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": prof_src,
                     "algorithmic_bytes_per_launch": algo * B, "avg_launch_us": step_us,
                     "avg_launch_us_method": "HIP events around every launch of the timed loop, minus what an event pair adds around this kernel (%.2f us: five launches in five pairs against five in one pair)" % pair_us,
                     "avg_launch_us_burst": burst_us, "kernel": cfg["kernel"]},
        "kernel_ms_per_vector_step": {"step": tm["step_ms"] / args.steps, "autoreset": tm["reset_ms"] / args.steps,
                                      "policy": tm["policy_ms"] / args.steps},
    }
    if "reset_path" in prof and args.config == "c2":
        out["roofline_reset"] = prof["reset_path"]  # SQ-counter issue rates of ge_k_features64 / ge_k_reset<0> (tools/pmc_sq_passes.sh)
        out["roofline_reset_source"] = prof_src
    env.close()
    del env
    if rank == 0 and world == 1 and args.config == "c2" and not args.no_1m:
        B1 = 1 << 20
        big = ge.make_vec(cfg["env_id"], B1, device=dev, **cfg["kw"])
        big.reset(seed=0)
        us1 = step_kernel_us(big, reps=5)
        ach1 = algo * B1 / (us1 * 1e-6) / 1e9
        out["roofline_1m"] = {"bound": "hbm", "achieved": ach1, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach1 / HBM_PEAK_GBS,
                              "traffic": prof.get("step_kernel_traffic", {}).get("c2_1m"), "algorithmic_bytes_per_launch": algo * B1,
                              "avg_launch_us": us1, "slots": B1, "kernel": cfg["kernel"]}
        big.close()
        del big
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle  # test infrastructure, used here only as the reported CPU baseline
        cores = os.cpu_count() or 1
        cpu_envs = args.cpu_envs or 64 * cores
        cpu_steps = 1000 if args.config == "c2" else 200
        if args.config != "c2":
            cpu_envs = args.cpu_envs or 2 * cores
        okw = cfg["kw"]
        oracle.rollout(cfg["env_id"], n_envs=cores, n_steps=5, n_threads=cores, **okw)
        t1 = time.perf_counter()
        r = oracle.rollout(cfg["env_id"], n_envs=cpu_envs, n_steps=cpu_steps, n_threads=cores, policy_seed=1, **okw)
        cdt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": r["transitions"] / cdt, "unit": "env-steps/s", "cores": cores, "kind": "port",
                               "sample": "%d envs x %d steps of the same workload (autoreset on, %d episodes), C oracle "
                                         "with OpenMP, %.1f s" % (cpu_envs, cpu_steps, r["episodes"], cdt)}
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
