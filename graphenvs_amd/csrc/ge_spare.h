// Episode prefetch ("spares", ge_attach_spares).  Episode k+1 of a slot is a pure function of (seed, k+1), so it can be produced
// before the slot needs it: every slot owns a SPARE IMAGE -- a second copy of every per-slot slab (observation, CSR, masks, scalar
// state) -- that the reset path fills, many slots per launch, every `period` steps (GeRun.refill: the same graph / feature kernels
// run on a view of the engine whose ge_buffers is the image and whose queue is the refill list).  A slot that finishes with a
// valid image is not regenerated: the step kernel queues it in swap_list and ge_k_swap streams the image over the live slabs.
// A slot that finishes again before its image was refilled takes the ordinary synchronous regeneration.  Results are those of
// the synchronous engine, bit for bit (reset() of the reference: shortest_path.py:47-98 and siblings) -- what changes is when the
// latency of a regeneration is paid, and how many slots share it.
#pragma once
#include "ge_params.h"
#include "ge_platform.h"
#include "ge_reset.h"
#include "ge_step.h"

#define GE_SWAP_GROUPS 10
// part `p` of `parts` of a byte range, copied by the whole workgroup in the widest unit the three alignments allow
GE_DEV void ge_copy_part(void *dst, const void *src, int64_t bytes, int p, int parts, int tid, int nt) {
  if (!dst || !src || bytes <= 0) return;
  const uintptr_t al = (uintptr_t)dst | (uintptr_t)src | (uintptr_t)bytes;
  if ((al & 15) == 0) {
    const int64_t u = bytes >> 4, lo = u * p / parts, hi = u * (p + 1) / parts;
    const ulonglong2 *s = (const ulonglong2 *)src; ulonglong2 *d = (ulonglong2 *)dst;
    int64_t k = lo + tid;
    for (; k + 3 * (int64_t)nt < hi; k += 4 * (int64_t)nt) {  // four loads in flight before the first store
      const ulonglong2 a = s[k], b = s[k + nt], c = s[k + 2 * (int64_t)nt], e = s[k + 3 * (int64_t)nt];
      d[k] = a; d[k + nt] = b; d[k + 2 * (int64_t)nt] = c; d[k + 3 * (int64_t)nt] = e;
    }
    for (; k < hi; k += nt) d[k] = s[k];
  } else if ((al & 3) == 0) {
    const int64_t u = bytes >> 2, lo = u * p / parts, hi = u * (p + 1) / parts;
    const uint32_t *s = (const uint32_t *)src; uint32_t *d = (uint32_t *)dst;
    for (int64_t k = lo + tid; k < hi; k += nt) d[k] = s[k];
  } else {
    const int64_t lo = bytes * p / parts, hi = bytes * (p + 1) / parts;
    const uint8_t *s = (const uint8_t *)src; uint8_t *d = (uint8_t *)dst;
    for (int64_t k = lo + tid; k < hi; k += nt) d[k] = s[k];
  }
}

// the spare image S of slot `env` (class-local index; C = the slot's uniform (sub-)engine) over its live slabs, part p of parts;
// part 0 also moves the slot's bookkeeping to the new episode
GE_DEV void ge_swap_slot(const GeParams &C, const ge_buffers &S, int env, int p, int parts, int tid, int nt) {
  const ge_buffers &G = C.buf;
  const int n = C.n, W = C.W, E = C.E;
  const int64_t nbase = (int64_t)env * n, ebase = (int64_t)env * E, Ne = C.edge_row_stride;
  // parts == GE_SWAP_GROUPS: every part copies WHOLE arrays (the large ones alone: a contiguous 16-byte stream with four loads in
  // flight); any other count: every part copies its share of every array
#define GE_CP(group, field, off, count) do { if (parts != GE_SWAP_GROUPS) ge_copy_part(G.field ? (void *)(G.field + (off)) : nullptr, S.field ? (const void *)(S.field + (off)) : nullptr, (int64_t)(count) * (int64_t)sizeof(*G.field), p, parts, tid, nt); \
    else if (p == (group)) ge_copy_part(G.field ? (void *)(G.field + (off)) : nullptr, S.field ? (const void *)(S.field + (off)) : nullptr, (int64_t)(count) * (int64_t)sizeof(*G.field), 0, 1, tid, nt); } while (0)
  // (a workgroup's copies are a chain of load -> store round trips to HBM: the small arrays are dealt over several groups)
  GE_CP(0, edge_index, ebase, E);
  GE_CP(1, edge_index, Ne + ebase, E);
  GE_CP(2, edge_attr, ebase * C.Fe, (int64_t)E * C.Fe);
  GE_CP(3, x, nbase * C.F, (int64_t)n * C.F);
  GE_CP(4, adj_bits, nbase * W, (int64_t)n * W);
  GE_CP(4, node_rec, nbase * 2, (int64_t)n * 2);
  GE_CP(5, rev_edge, ebase, E);
  GE_CP(5, sw64, ebase, E);
  GE_CP(6, range_bits, nbase * W, (int64_t)n * W);
  GE_CP(6, colw, ebase, E);
  GE_CP(7, scode, ebase, E);
  GE_CP(7, mask, (int64_t)env * C.A, C.A);
  GE_CP(7, row_ptr, (int64_t)env * (n + 1), n + 1);
  GE_CP(8, node_aux, nbase, n);
  GE_CP(8, terminals, (int64_t)env * C.T, C.T);
  GE_CP(8, node_bits, (int64_t)env * W, W);
  GE_CP(8, target_bits, (int64_t)env * W, W);
  GE_CP(9, counters, (int64_t)env * 2, 2);
  GE_CP(9, mask_bits, (int64_t)env * C.AW, C.AW);
  GE_CP(9, aux_bits, (int64_t)env, 1);
  GE_CP(9, cover_bits, (int64_t)env * W, W);
#undef GE_CP
  if (p == 0 && tid == 0) {
    // what ge_reset_env's last lines do for a slot regenerated in place: the image carries cost 0, head, destination and (a failed
    // generation) status 4; the slot keeps its transition count; next-step autoreset marks it "regenerated in this ge_step"
    const uint64_t img = S.slot_rec[2 * (int64_t)env + 1], live = G.slot_rec[2 * (int64_t)env + 1];
    uint64_t status = (img >> GE_REC_STATUS_SHIFT) & 0xffull;
    if (status != 4ull) status = (C.autoreset == 2) ? 3ull : 0ull;
    const uint64_t keep = GE_REC_HEAD_MASK | (0xffull << GE_REC_AUX_SHIFT);
    G.slot_rec[2 * (int64_t)env] = S.slot_rec[2 * (int64_t)env];
    G.slot_rec[2 * (int64_t)env + 1] = (img & keep) | (status << GE_REC_STATUS_SHIFT) | ((live >> GE_REC_TSTEP_SHIFT) << GE_REC_TSTEP_SHIFT);
    G.final_heur[env] = G.heuristic[env];  // of the episode that just ended
    G.heuristic[env] = S.heuristic[env];
    G.seed[env] = G.seed[env] + (uint32_t)C.seed_stride;
    G.episode[env] = G.episode[env] + 1;
  }
}

// grid-stride over (queued slot, part); P = the engine-wide block, S = the image of a uniform engine; RAGGED: RS.classes[c].buf is
// the image of class c
template <bool RAGGED>
GE_KERNEL ge_k_swap(GeParams P, GeRagged R, GeRagged RS, ge_buffers S, int parts) {
  int *pre = (int *)ge_dyn_smem();
  const int nblk = (P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK;
  if (ge_tid() < GE_WAVE) ge_queue_prefix_of(P.swap_count, nblk, pre, ge_tid());
  ge_sync();
  const int count = pre[nblk];
  for (int q = ge_bid(); q < count * parts; q += ge_gdim()) {
    const int item = q / parts, part = q % parts;
    const int env = ge_queue_slot_of(P.swap_list, nblk, pre, item);
    if (part == 0 && ge_tid() == 0) P.spare_state[env] = 0;  // the image is consumed (the step kernel cleared it already; a restored snapshot may not have)
    if constexpr (RAGGED) {
      const int cls = (int)ge_uniform_u32((uint32_t)R.slot_class[env]);
      ge_swap_slot(R.classes[cls], RS.classes[cls].buf, env - R.class_start[cls], part, parts, ge_tid(), ge_bdim());
    } else {
      ge_swap_slot(P, S, env, part, parts, ge_tid(), ge_bdim());
    }
  }
}

// the slots whose image is empty, as a queue in the layout of reset_list / reset_count (block g lists its slots at [256 g, ..)):
// the refill launches read it through the image view's reset_list / reset_count
GE_KERNEL ge_k_refill_list(GeParams P, int32_t *list, int32_t *count) {
  const int tid = ge_tid(), i0 = ge_bid() * ge_bdim(), i = i0 + tid;
  int *wcnt = (int *)ge_dyn_smem();
  const bool want = i < P.B && P.spare_state[i] == 0;
  const uint64_t b = ge_ballot(want);
  const int lane = tid & 63, wave = tid >> 6, nw = ge_bdim() >> 6;
  if (lane == 0) wcnt[wave] = ge_popc64(b);
  ge_sync();
  int off = 0;
  for (int w = 0; w < wave; w++) off += wcnt[w];
  if (want) list[i0 + off + ge_popc64(b & ((1ull << lane) - 1ull))] = i;
  if (tid == 0) { int tot = 0; for (int w = 0; w < nw; w++) tot += wcnt[w]; count[ge_bid()] = tot; }
}
