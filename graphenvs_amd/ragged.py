"""Ragged and mixed batches (BASELINE config 5): env instances of different (n_nodes, n_edges) -- and different env ids --
stepped together.  Every reference env instance has a fixed geometry (constructor kwargs), so a ragged batch is a list of size
classes.  ``RaggedVectorEnv`` is ONE multi-class engine per env id (``ge_create_ragged``): every kernel launch covers all classes
(a workgroup looks up the class of its slot), the classes share one set of PyG slabs (x, edge_index, edge_attr: variable-size CSR
packing, node ids offset per class through ``ge_config.node_id_base`` / ``edge_row_stride``), and the per-slot outputs are single
tensors over all slots.  ``MixedVectorEnv`` puts env ids side by side: one launch sequence per env id."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .vector_env import GraphBatch, VectorGraphEnv

# per-slot arrays kept engine-wide (class c owns rows [start_c, start_c + B_c))
_GLOBAL = dict(seed=((), torch.int32), episode=((), torch.int64), mt_state=((_lib.SEED_DEPTH, 2, 624), torch.int32),
               slot_rec=((2,), torch.int64), heuristic=((), torch.float64), reward=((), torch.float64),
               terminated=((), torch.uint8), invalid=((), torch.uint8), solved=((), torch.int8),
               final_cost=((), torch.float64), final_heur=((), torch.float64), final_len=((), torch.int32),
               counters=((2,), torch.int32))


class RaggedVectorEnv:
    """One env id, several size classes: ``sizes = [(num_envs, n_nodes, n_edges), ...]``.  Slots are numbered class after class;
    slot g runs seed (seed + g) like a uniform engine.  ``classes[c]`` are views of class c (``.mask`` [B_c, A_c], ``.t[...]``)."""

    def __init__(self, env_id, sizes, device="cuda", env_index_base=0, seed_stride=None, autoreset=True, _library=None, prefetch=None,
                 **kwargs):
        self.env_id, self.device = env_id, torch.device(device)
        self.sizes = [(int(b), int(n), int(m)) for b, n, m in sizes]
        self.num_envs = B = sum(b for b, _, _ in self.sizes)
        stride = int(seed_stride) if seed_stride is not None else B
        self.seed_stride, self.env_index_base = stride, int(env_index_base)
        extra = dict(device=device, _library=_library) if _library is not None else dict(device=device)
        probe = VectorGraphEnv(env_id, 1, self.sizes[0][1], self.sizes[0][2], _defer_create=True, **extra, **kwargs)
        F, Fe, edge_env = probe.F, probe.Fe, env_id in ("SteinerTree-v0", "MulticastRouting-v0")
        self._L = probe._L
        dev = self.device
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        Nn = sum(b * n for b, n, _ in self.sizes)
        Ne = sum(b * 2 * m for b, _, m in self.sizes)
        self.x, self.edge_index, self.edge_attr = z((Nn, F), torch.float32), z((2, Ne), torch.int64), z((Ne, Fe), torch.float32)
        A_of = lambda n, m: 2 * m if edge_env else n
        self.mask_flat = z((sum(b * A_of(n, m) for b, n, m in self.sizes),), torch.uint8)
        self.g = {k: z((B,) + shape, dt) for k, (shape, dt) in _GLOBAL.items()}
        self.g["reset_list"], self.g["reset_count"] = z((B,), torch.int32), z(((B + 255) // 256,), torch.int32)
        self.g["work_list"], self.g["work_count"] = z((B,), torch.int32), z((4,), torch.int32)
        self.classes, self.slot_ptr, self._offsets = [], [0], []
        noff = eoff = slot = moff = 0
        ptr = [0]
        for b, n, m in self.sizes:
            E, A = 2 * m, A_of(n, m)
            views = dict(x=self.x[noff:noff + b * n], edge_index=self.edge_index[0, eoff:], edge_attr=self.edge_attr[eoff:eoff + b * E],
                         mask=self.mask_flat[moff:moff + b * A].view(b, A))
            views.update({k: self.g[k][slot:slot + b] for k in _GLOBAL})
            views.update({k: self.g[k] for k in ("reset_list", "reset_count", "work_list", "work_count")})
            env = VectorGraphEnv(env_id, b, n, m, env_index_base=self.env_index_base + slot, seed_stride=stride, autoreset=autoreset,
                                 _views=views, node_id_base=noff, edge_row_stride=Ne, _defer_create=True, prefetch=0, **extra, **kwargs)
            self.classes.append(env)
            self._offsets.append((noff, eoff, slot, moff, b, n, E, A))
            ptr += [noff + (i + 1) * n for i in range(b)]
            noff += b * n; eoff += b * E; slot += b; moff += b * A
            self.slot_ptr.append(slot)
        nc = len(self.classes)
        self._table = torch.zeros(int(self._L.ge_ragged_table_bytes(nc)), dtype=torch.uint8, device=dev)
        self._slot_class, self._class_start = z((B,), torch.int32), z((nc + 1,), torch.int32)
        cfgs = (_lib.GeConfig * nc)(*[c.cfg for c in self.classes])
        bufs = (_lib.GeBuffers * nc)(*[c.bufs for c in self.classes])
        h = C.c_void_p()
        _lib.check(self._L, self._L.ge_create_ragged(cfgs, bufs, nc, self._table.data_ptr(), self._slot_class.data_ptr(),
                                                     self._class_start.data_ptr(), C.byref(h)), "ge_create_ragged")
        self._h = h
        # episode prefetch (include/graphenvs.h, ge_attach_spares): a ragged batch is where it pays most -- every step a few slots of
        # many different sizes finish, and regenerated in place they cost the step the latency of the largest of them.  Refill every
        # 4 steps by default: on BASELINE config 5 (profiles/r03_c5_refill_period.txt) 0 / 2 / 4 / 6 / 8 / 16 / 32 steps give
        # 15.3 / 19.6 / 20.8 / 19.1 / 19.0 / 17.2 / 16.0 M env-steps/s -- the small classes' episodes last a handful of steps, and a
        # slot that finishes again before its image is refilled takes the in-place path
        self.prefetch = (4 if autoreset and _library is None else 0) if prefetch is None else int(prefetch)
        self.spare = None
        if self.prefetch and autoreset:
            self._attach_spares()
        self.ptr = torch.tensor(ptr, dtype=torch.int64, device=dev)
        self.batch = torch.repeat_interleave(torch.arange(B, device=dev), self.ptr[1:] - self.ptr[:-1])
        self._truncated = torch.zeros(B, dtype=torch.bool, device=dev)
        self._actions = z((B,), torch.int64)
        self._flat = None
        self._was_reset = False

    def _attach_spares(self):
        """a spare image of every class, packed like the live slabs (one x / edge_index / edge_attr / mask for all classes)"""
        B, dev, nc = self.num_envs, self.device, len(self.classes)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        shared = dict(x=torch.zeros_like(self.x), edge_index=torch.zeros_like(self.edge_index), edge_attr=torch.zeros_like(self.edge_attr),
                      mask=torch.zeros_like(self.mask_flat))
        shared.update({k: torch.zeros_like(self.g[k]) for k in _GLOBAL if k in _lib.IMAGE_FIELDS})
        sp = dict(state=z((B,), torch.uint8), swap_list=z((B,), torch.int32), swap_count=z(((B + 255) // 256,), torch.int32),
                  refill_list=z((B,), torch.int32), refill_count=z(((B + 255) // 256,), torch.int32))
        images, recs = [], []
        for env, (noff, eoff, slot, moff, b, n, E, A) in zip(self.classes, self._offsets):
            views = dict(x=shared["x"][noff:noff + b * n], edge_index=shared["edge_index"][0, eoff:], edge_attr=shared["edge_attr"][eoff:eoff + b * E],
                         mask=shared["mask"][moff:moff + b * A].view(b, A))
            views.update({k: shared[k][slot:slot + b] for k in _GLOBAL if k in _lib.IMAGE_FIELDS})
            img = env._image_tensors(views)
            images.append(img)
            recs.append(_lib.GeSpares(_lib.GeBuffers(**{k: (v.data_ptr() if v is not None else None) for k, v in img.items()}),
                                      *(sp[k].data_ptr() for k in ("state", "swap_list", "swap_count", "refill_list", "refill_count")), self.prefetch))
        self._table_spare = torch.zeros_like(self._table)
        self.spare = dict(images=images, shared=shared, **sp)
        arr = (_lib.GeSpares * nc)(*recs)
        _lib.check(self._L, self._L.ge_attach_spares(self._h, arr, self._table_spare.data_ptr()), "ge_attach_spares")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream) if self.device.type == "cuda" else C.c_void_p(0)

    def graph(self):
        return GraphBatch(x=self.x, edge_index=self.edge_index, edge_attr=self.edge_attr, batch=self.batch, ptr=self.ptr,
                          num_graphs=self.num_envs)

    def _info(self, stepped):
        info = {"mask": [c.mask for c in self.classes],  # ragged: one [B_c, A_c] bool view per class
                "mask_flat": self.mask_flat.view(torch.bool)}
        if stepped:
            g = self.g
            info.update(solved=g["solved"], solution_cost=g["final_cost"], heuristic_solution=g["final_heur"],
                        invalid_action=g["invalid"].view(torch.bool), episode_length=g["final_len"])
        return info

    def reset(self, seed=0):
        s = (int(seed) + self.env_index_base + np.arange(self.num_envs, dtype=np.int64)) % (1 << 32)
        self._seeds = torch.from_numpy(s.astype(np.uint32).view(np.int32)).to(self.device)
        _lib.check(self._L, self._L.ge_reset(self._h, self._seeds.data_ptr(), self._stream()), "ge_reset")
        self._was_reset = True
        return self.graph(), self._info(False)

    def step(self, actions):
        actions = torch.as_tensor(actions).to(self.device, torch.int64).contiguous()
        assert actions.shape == (self.num_envs,) and self._was_reset
        self._act_keepalive = actions
        _lib.check(self._L, self._L.ge_step(self._h, actions.data_ptr(), self._stream()), "ge_step")
        g = self.g
        return self.graph(), g["reward"], g["terminated"].view(torch.bool), self._truncated, self._info(True)

    def sample_random_actions(self, policy_seed=0):
        _lib.check(self._L, self._L.ge_sample_actions(self._h, int(policy_seed), self._actions.data_ptr(), self._stream()), "ge_sample_actions")
        return self._actions

    def random_rollout(self, n_steps, policy_seed=0):
        _lib.check(self._L, self._L.ge_random_rollout(self._h, int(policy_seed), int(n_steps), self._actions.data_ptr(), self._stream()),
                   "ge_random_rollout")

    def timed_rollout(self, n_steps, policy_seed=0):
        """the rollout with a HIP-event pair around the policy, the step kernel and the autoreset launches of every step (profiling)"""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        _lib.check(self._L, self._L.ge_timed_rollout(self._h, int(policy_seed), int(n_steps), self._actions.data_ptr(), self._stream(),
                                                     C.byref(a), C.byref(b), C.byref(c)), "ge_timed_rollout")
        return dict(step_ms=a.value, reset_ms=b.value, policy_ms=c.value)

    def flat_obs(self):
        """utils.vectorize_graph of every slot, one [B_c, obs_len_c] tensor per class (views of one buffer)."""
        lens = [c.num_envs * c.obs_len for c in self.classes]
        if self._flat is None:
            self._flat = torch.empty(sum(lens), dtype=torch.float32, device=self.device)
        _lib.check(self._L, self._L.ge_vectorize(self._h, self._flat.data_ptr(), self._stream()), "ge_vectorize")
        out, off = [], 0
        for c, ln in zip(self.classes, lens):
            out.append(self._flat[off:off + ln].view(c.num_envs, c.obs_len)); off += ln
        return out

    def close(self):
        if getattr(self, "_h", None):
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            self._L.ge_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_STREAMS = {}


def _runs_beside(a, b, device):
    """do kernels on streams a and b overlap?  The runtime maps streams onto a handful of hardware queues, and two streams on one
    queue run one behind the other.  Probe: a spin kernel on each, timed together against one alone."""
    spin = getattr(torch.cuda, "_sleep", None)
    if spin is None:
        return True
    cycles = 400000  # ~0.2 ms
    def timed(streams):
        torch.cuda.synchronize(device)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream(device)
        t0.record(cur)
        for st in streams:
            st.wait_event(t0)
            with torch.cuda.stream(st):
                spin(cycles)
        for st in streams:
            cur.wait_stream(st)
        t1.record(cur)
        torch.cuda.synchronize(device)
        return t0.elapsed_time(t1)
    timed([a]); one = min(timed([a]) for _ in range(3)); both = min(timed([a, b]) for _ in range(3))
    return both < 1.5 * one


def _member_streams(device, want):
    """up to `want` streams of the process (cached: every MixedVectorEnv uses the same ones) that run BESIDE one another -- a fresh
    stream that shares a hardware queue with one already chosen is set aside and the next is tried; when eight in a row fail the
    device has no queue left and the list ends there (profiles/r04_shards.txt: two shards on one queue 212 M env-steps/s instead of
    353 M; five shards on a device with four queues 166 M)"""
    dev = torch.device(device)
    have = _STREAMS.setdefault(str(dev), [])
    aside = _STREAMS.setdefault(str(dev) + " aside", [])
    full = _STREAMS.setdefault(str(dev) + " full", [False])
    while len(have) < want and not full[0]:
        for _ in range(8):
            cand = torch.cuda.Stream(device=dev)
            if all(_runs_beside(e, cand, dev) for e in have):
                have.append(cand)
                break
            aside.append(cand)  # (kept alive: a freed stream's queue slot would be handed out again)
        else:
            full[0] = True
    return have[:want]


class MixedVectorEnv:
    """Several env ids side by side (each a RaggedVectorEnv or VectorGraphEnv); step takes one action tensor per
    member.  Observation widths differ between ids (utils.get_env_info), so each member keeps its own PyG view.

    The members are independent engines, so on the GPU every call fans out over one HIP stream per member and joins on the caller's
    stream before it returns (``concurrent=False``: one after the other on the caller's stream): the regeneration kernels of a
    member with large graphs hold a workgroup per CU for hundreds of microseconds, and the other members' launches fill the rest
    of the chip meanwhile.  Results do not depend on it -- nothing is shared between members."""

    def __init__(self, members, concurrent=True):
        self.members = list(members)
        self.num_envs = sum(m.num_envs for m in self.members)
        dev = getattr(self.members[0], "device", None)
        self._cuda = concurrent and dev is not None and torch.device(dev).type == "cuda" and len(self.members) > 1
        # member k on a stream of its own -- the same streams for every MixedVectorEnv of the process, chosen so that they run beside one
        # another (_member_streams); with fewer such streams than members, members share them round-robin
        if self._cuda:
            pool = _member_streams(self.members[0].device, len(self.members))
            self._streams = [pool[k % len(pool)] for k in range(len(self.members))]
            self.concurrent_streams = len(pool)
        else:
            self._streams, self.concurrent_streams = None, 1

    def _each(self, fn, args=None):
        """fn(member[, arg]) for every member: on the member's own stream between a fork from and a join on the current stream"""
        args = [None] * len(self.members) if args is None else list(args)
        call = lambda m, a: fn(m) if a is None else fn(m, a)
        if not self._cuda:
            return [call(m, a) for m, a in zip(self.members, args)]
        cur = torch.cuda.current_stream(self.members[0].device)
        fork = cur.record_event()
        outs = []
        for m, a, st in zip(self.members, args, self._streams):
            st.wait_event(fork)
            with torch.cuda.stream(st):
                outs.append(call(m, a))
        for st in self._streams:
            cur.wait_stream(st)
        # tensors a member allocated inside its call (sampled actions, copy_outputs clones) belong to the side stream's pool: tell the
        # caching allocator that the caller's stream uses them too, or a free followed by a direct call on a member could reuse the
        # memory while the caller's stream still reads it
        def mark(v):
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)
            elif isinstance(v, dict):
                for x in v.values():
                    mark(x)
            elif isinstance(v, (tuple, list)):
                for x in v:
                    mark(x)
            elif hasattr(v, "__dict__") and not callable(v):
                for x in vars(v).values():
                    mark(x)
        mark(outs)
        return outs

    def reset(self, seed=0):
        outs = self._each(lambda m: m.reset(seed=seed))
        return [o for o, _ in outs], [i for _, i in outs]

    def step(self, actions):
        outs = self._each(lambda m, a: m.step(a), actions)
        return tuple(list(col) for col in zip(*outs))

    def sample_random_actions(self, policy_seed=0):
        return self._each(lambda m: m.sample_random_actions(policy_seed))

    def random_rollout(self, n_steps, policy_seed=0):
        """n_steps fused (device policy + step + autoreset) vector steps of every member.  The members are independent engines and
        nothing is read in between, so the streams are forked ONCE, the launches of the members alternate step by step (the host
        enqueues far ahead of the GPU: a member enqueued whole would run alone until the next one's launches arrive) and the caller's
        stream joins ONCE at the end -- no event between streams per step, and no member waits for the regeneration round of another."""
        n_steps = int(n_steps)
        if not self._cuda:
            for m in self.members:
                m.random_rollout(n_steps, policy_seed)
            return
        cur = torch.cuda.current_stream(self.members[0].device)
        fork = cur.record_event()
        for st in self._streams:
            st.wait_event(fork)
        chunk = max(1, int(os.environ.get("GE_ROLLOUT_CHUNK", "4")))  # steps a member enqueues before the next member's turn (1 .. 8 measured alike; fewer host calls)
        for s0 in range(0, n_steps, chunk):
            for m, st in zip(self.members, self._streams):
                with torch.cuda.stream(st):
                    m.random_rollout(min(chunk, n_steps - s0), policy_seed)
        for st in self._streams:
            cur.wait_stream(st)

    def close(self):
        for m in self.members:
            m.close()
