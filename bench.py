#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): env-steps/sec, ShortestPath-v0 n=64 m=192, 65 536 envs per GPU,
random-valid-action policy generated on device, same-step autoreset ON (every reset regenerates the graph,
features and masks on the GPU, seed-exact with the reference).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One JSON line on rank 0.  `value` = (envs x steps over all ranks) / max-over-ranks wall time, inputs resident
in HBM.  `roofline` is the step kernel (HBM-bound; 200 algorithmic bytes per env-step) timed with
HIP events on its stream; `cpu_baseline` is the CPU oracle (a port of the reference semantics, not the
reference) on the host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 200  # ShortestPath C2, canonical 32-bit CSR + byte mask
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=65536, help="env slots per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    B, n, m = args.envs, 64, 192
    env = ge.make_vec("ShortestPath-v0", B, n_nodes=n, n_edges=m, device=f"cuda:{local_rank}",
                      env_index_base=rank * B, seed_stride=world * B)
    env.reset(seed=0)
    env.random_rollout(args.warmup, policy_seed=1)

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
    assert int(env.t["tstep"].sum()) == B * (args.steps + args.warmup)
    value = world * B * args.steps / dt

    # per-kernel time, HIP events on the launch stream.  (1) every launch of the real loop bracketed by its own event
    # pair (carries ~3 us of event overhead per bracket); (2) the step kernel alone: bursts of 5 back-to-back launches
    # between ONE event pair right after a full reset (>= 80 % of the slots still running), which is the figure that
    # agrees with rocprofv3's kernel trace (profiles/) and is used for the roofline.
    tm = env.timed_rollout(args.steps, policy_seed=1)
    KB = 5
    empty = sorted(env.timed_step_burst_raw_ms(0) for _ in range(9))[4]  # an event pair with nothing in between
    bursts = []
    for rep in range(9):
        env.reset(seed=2000 + rep)
        bursts.append((env.timed_step_burst_raw_ms(KB, policy_seed=2) - empty) * 1e3 / KB)
    step_us = sorted(bursts)[len(bursts) // 2]
    achieved = ALGO_BYTES_PER_ENV_STEP * B / (step_us * 1e-6) / 1e9
    pmc_path = os.path.join(ROOT, "profiles", "pmc_step_kernel.json")
    traffic = json.load(open(pmc_path)).get("traffic_bytes_per_launch") if os.path.exists(pmc_path) else None

    out = {
        "metric": "env-steps/sec (whole node), ShortestPath-v0 n=64 m=192 batch=65536, 1/2/4/8 GPU",  # BASELINE.json's metric, verbatim
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": "ShortestPath-v0 n_nodes=64 n_edges=192 weighted, %d env slots per GPU, random valid "
                               "actions on device, same-step autoreset (seed-exact G(n,m)+features on device)" % B,
                   "envs_per_gpu": B, "episodes_finished_per_gpu": episodes, "parallelism": "batch shard x%d, no collective" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * B, "avg_launch_us": step_us,
                     "avg_launch_us_single_bracket": tm["step_ms"] * 1e3 / args.steps,
                     "kernel": "ge_k_step_path64<true> (fused device policy + step)"},
        "kernel_ms_per_vector_step": {"step": tm["step_ms"] / args.steps, "autoreset": tm["reset_ms"] / args.steps,
                                      "policy": tm["policy_ms"] / args.steps},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle  # test infrastructure, used here only as the reported CPU baseline
        cores = os.cpu_count() or 1
        cpu_envs = args.cpu_envs or 64 * cores
        cpu_steps = 1000
        oracle.rollout("ShortestPath-v0", n_envs=cores, n_steps=10, n_nodes=n, n_edges=m, n_threads=cores)
        t1 = time.perf_counter()
        r = oracle.rollout("ShortestPath-v0", n_envs=cpu_envs, n_steps=cpu_steps, n_nodes=n, n_edges=m,
                           n_threads=cores, policy_seed=1)
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
