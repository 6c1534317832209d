"""CPU, world_size 2 over gloo: the N>1 path of bench.py / multi-GPU use is a plain shard of the batch
dimension.  Each rank owns slots [rank*B, (rank+1)*B) through env_index_base; no data-path
collective exists, only the max-over-ranks reduction of the wall time.  The ranks drive the emulated kernels
(tests/emu) because this container has no GPU; the sharding logic under test is host code."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, HERE, os.path.join(HERE, "emu")):
        sys.path.insert(0, p)
    import build_emu
    import graphenvs_amd as ge
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = 24
    env = ge.VectorGraphEnv("ShortestPath-v0", B, 12, 30, device="cpu", _library=build_emu.load(), obs_mode="flat",
                            env_index_base=rank * B, seed_stride=world * B)
    env.reset(seed=7)
    dist.barrier()
    env.random_rollout(30, policy_seed=3)
    dist.barrier()
    steps = torch.tensor([float(env.t["tstep"].sum())])
    dist.all_reduce(steps, op=dist.ReduceOp.SUM)  # whole-job count, as bench.py reports it
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=env.flat_obs().numpy(), reward=env.t["reward"].numpy(),
             episode=env.t["episode"].numpy(), total=steps.numpy(), tmax=t.numpy())
    dist.destroy_process_group()


def test_two_rank_shard_equals_single_engine(tmp_path):
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu
    import graphenvs_amd as ge
    build_emu.build()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert r0["total"][0] == 2 * 24 * 30 and r0["tmax"][0] == 2.0
    whole = ge.VectorGraphEnv("ShortestPath-v0", 48, 12, 30, device="cpu", _library=build_emu.load(), obs_mode="flat",
                              seed_stride=48)
    whole.reset(seed=7)
    whole.random_rollout(30, policy_seed=3)
    assert np.array_equal(whole.flat_obs().numpy(), np.concatenate([r0["flat"], r1["flat"]]))
    assert np.array_equal(whole.t["reward"].numpy(), np.concatenate([r0["reward"], r1["reward"]]))
    assert np.array_equal(whole.t["episode"].numpy(), np.concatenate([r0["episode"], r1["episode"]]))
