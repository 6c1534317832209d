"""A uniform batch as several independent engines ("shards") in flight together on one GPU.

One vector step of one engine is a chain of launches that use the chip in different ways: the graph kernel of a regeneration is a
latency chain (every queued slot resident at once, the launch as long as its unluckiest G(n, m) sample), the feature kernel behind
it is bound by instruction issue.  Two engines over the two halves of the batch, each on its own HIP stream, run those phases against
each other: measured on the headline (ShortestPath n=64 m=192, 65 536 slots, MI355X) 305 -> 353 M env-steps/s with two shards,
229 M with three (profiles/r04_shards.txt).  Slots keep their numbers -- shard k owns the consecutive slots [k B/S, (k+1) B/S) and
slot g runs seed (seed + g) exactly as in one engine (``env_index_base`` / ``seed_stride``), so results do not depend on S.
"""
import torch

from .ragged import MixedVectorEnv
from .vector_env import VectorGraphEnv


class ShardedVectorEnv(MixedVectorEnv):
    """``ShardedVectorEnv(env_id, num_envs, shards=2, **kwargs)``: ``members[k]`` is the VectorGraphEnv of shard k.  reset / step /
    sample_random_actions / random_rollout are MixedVectorEnv's (one entry per shard, every shard on its own stream); ``gather(key)``
    concatenates a per-slot tensor of the shards in slot order."""

    def __init__(self, env_id, num_envs, shards=2, device="cuda", env_index_base=0, seed_stride=None, concurrent=True, **kwargs):
        num_envs, shards = int(num_envs), int(shards)
        if shards < 1 or shards > num_envs:
            raise ValueError("shards must be between 1 and num_envs")
        stride = int(seed_stride) if seed_stride is not None else num_envs
        if torch.device(device).type == "cuda" and shards > 1 and concurrent:  # no more shards than streams that run beside one another on this device
            from .ragged import _member_streams
            shards = max(1, min(shards, len(_member_streams(device, shards))))
        sizes = [num_envs // shards + (1 if k < num_envs % shards else 0) for k in range(shards)]
        members, off = [], 0
        for b in sizes:
            members.append(VectorGraphEnv(env_id, b, device=device, env_index_base=int(env_index_base) + off, seed_stride=stride, **kwargs))
            off += b
        super().__init__(members, concurrent=concurrent)  # (concurrent=False: the shards one after the other on the caller's stream -- counter collection)
        self.env_id, self.shards, self.seed_stride, self.env_index_base = env_id, shards, stride, int(env_index_base)
        self.device = members[0].device

    @property
    def prefetch(self):
        return self.members[0].prefetch

    def gather(self, key):
        """the per-slot tensor ``key`` (a key of VectorGraphEnv.t: "episode", "tstep", "reward", ...) of every slot, in slot order"""
        return torch.cat([m.t[key] for m in self.members])

    def check_device_errors(self):
        for m in self.members:
            m.check_device_errors()

    def timed_rollout(self, n_steps, policy_seed=0):
        """profiling: the rollout of every shard, shard 0's with a HIP-event pair around its policy, step kernel and autoreset launches
        (VectorGraphEnv.timed_rollout) while the other shards roll out beside it on their streams.  Returns shard 0's figures."""
        if not self._cuda:
            for m in self.members[1:]:
                m.random_rollout(n_steps, policy_seed)
            return self.members[0].timed_rollout(n_steps, policy_seed)
        cur = torch.cuda.current_stream(self.device)
        fork = cur.record_event()
        for st in self._streams:
            st.wait_event(fork)
        for m, st in zip(self.members[1:], self._streams[1:]):
            with torch.cuda.stream(st):
                m.random_rollout(n_steps, policy_seed)
        with torch.cuda.stream(self._streams[0]):
            out = self.members[0].timed_rollout(n_steps, policy_seed)
        for st in self._streams:
            cur.wait_stream(st)
        return out
