// Step kernel: one thread advances one env slot; a workgroup of 256 slots stages its new action
// masks in LDS and writes the bool mask slab with coalesced 8-byte stores.  Replaces step() and
// _get_mask() of the six reference envs (shortest_path.py:101-141, longest_path.py:121-196,
// steiner_tree.py:116-157, tsp.py:170-258, densest_subgraph.py:101-196,
// max_independent_set.py:92-124).
#pragma once
#include "ge_params.h"
#include "ge_platform.h"
#include "ge_reset.h"

GE_DEV uint64_t ge_full_word(int A, int w) {
  int lo = w * 64, hi = lo + 64; if (hi > A) hi = A;
  if (hi <= lo) return 0ull;
  return (hi - lo == 64) ? ~0ull : ((1ull << (hi - lo)) - 1ull);
}

// weight code of the directed edge u -> a inside slot `env`: position by rank in the bit row, no scan
GE_DEV int ge_edge_code(const GeParams &P, int env, int u, int a) {
  const uint64_t *row = P.buf.adj_bits + ((int64_t)env * P.n + u) * P.W;
  int r0 = P.buf.row_ptr[(int64_t)env * (P.n + 1) + u];
  return P.buf.scode[(int64_t)env * P.E + r0 + ge_rank_below(row, a)];
}

// slot_rec: {cost as float64 bits, packed word} -- include/graphenvs.h GE_REC_*
GE_DEV int ge_rec_head(uint64_t p) { const int h = (int)(p & GE_REC_HEAD_MASK); return h == (int)GE_REC_HEAD_MASK ? -1 : h; }
GE_DEV int ge_rec_status(uint64_t p) { return (int)((p >> GE_REC_STATUS_SHIFT) & 0xffull); }
GE_DEV int ge_rec_aux(uint64_t p) { return (int)((p >> GE_REC_AUX_SHIFT) & 0xffull); }
GE_DEV uint64_t ge_rec_tstep(uint64_t p) { return p >> GE_REC_TSTEP_SHIFT; }
GE_DEV uint64_t ge_rec_make(int head, int status, int aux, uint64_t tstep) {
  return ((uint64_t)head & GE_REC_HEAD_MASK) | ((uint64_t)status << GE_REC_STATUS_SHIFT) | ((uint64_t)aux << GE_REC_AUX_SHIFT) | (tstep << GE_REC_TSTEP_SHIFT);
}

// Finished slots of this workgroup go to the workgroup's own segment of reset_list, in slot order, and the segment length to
// reset_count[workgroup]: no device-scope atomics, deterministic order.  Collective.
// With spares attached (ge_attach_spares) a finished slot whose spare image is valid goes to the swap queue (swap_list / swap_count,
// same layout) instead: `swap` says which.  wcnt: 2 ints per wave of the workgroup.
// SPARES = false: the engine has none (a kernel instantiated so carries none of the second queue's work).
template <bool SPARES = true>
GE_DEV void ge_enqueue_reset(const GeParams &P, int *wcnt, int i0, int i, int tid, bool want, bool swap = false) {
  if constexpr (!SPARES) {
    const uint64_t b = ge_ballot(want);
    const int lane = tid & 63, wave = tid >> 6, nw = ge_bdim() >> 6;
    const int rank = ge_popc64(b & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[wave] = ge_popc64(b);
    ge_sync();
    int off = 0;
    for (int w = 0; w < wave; w++) off += wcnt[w];
    if (want) P.buf.reset_list[i0 + off + rank] = i;
    if (tid == 0) { int tot = 0; for (int w = 0; w < nw; w++) tot += wcnt[w]; P.buf.reset_count[ge_bid()] = tot; }
    return;
  }
  const bool wr = want && !swap, ws = want && swap;
  const uint64_t b = ge_ballot(wr), b2 = ge_ballot(ws);
  const int lane = tid & 63, wave = tid >> 6, nw = ge_bdim() >> 6;
  const uint64_t below = (1ull << lane) - 1ull;
  if (lane == 0) { wcnt[wave] = ge_popc64(b); wcnt[nw + wave] = ge_popc64(b2); }
  ge_sync();
  int off = 0, off2 = 0;
  for (int w = 0; w < wave; w++) { off += wcnt[w]; off2 += wcnt[nw + w]; }
  if (wr) P.buf.reset_list[i0 + off + ge_popc64(b & below)] = i;
  if (ws) P.swap_list[i0 + off2 + ge_popc64(b2 & below)] = i;
  if (tid == 0) {
    int tot = 0, tot2 = 0;
    for (int w = 0; w < nw; w++) { tot += wcnt[w]; tot2 += wcnt[nw + w]; }
    P.buf.reset_count[ge_bid()] = tot;
    if (P.swap_count) P.swap_count[ge_bid()] = tot2;
  }
}

#ifndef GE_MAXW
#define GE_MAXW 8  // parenting >= 2 walks the residual graph per thread: node sets of up to 8 words (n <= 512) live in registers
#endif

// nodes reachable from `start` inside `alive` minus `skip` (skip < 0: none); rows = the slot's adjacency bit rows
// FR / NX: the frontier and the next level -- registers (arrays of GE_MAXW words) or, above 64 * GE_MAXW nodes, two more sets of the
// slot's prune_scratch
template <class FR, class NX>
GE_DEV void ge_reach_sets(const uint64_t *rows, int W, const uint64_t *alive, int start, int skip, uint64_t *R, FR &fr, NX &nx) {
  for (int w = 0; w < W; w++) { R[w] = 0; fr[w] = 0; }
  R[start >> 6] = fr[start >> 6] = 1ull << (start & 63);
  for (;;) {
    for (int w = 0; w < W; w++) nx[w] = 0;
    for (int w = 0; w < W; w++)
      for (uint64_t f = fr[w]; f; f &= f - 1) {
        const uint64_t *row = rows + (int64_t)(w * 64 + ge_ctz64(f)) * W;
        for (int w2 = 0; w2 < W; w2++) nx[w2] |= row[w2];
      }
    uint64_t any = 0;
    for (int w = 0; w < W; w++) {
      uint64_t m = nx[w] & alive[w] & ~R[w];
      if (skip >= 0 && (skip >> 6) == w) m &= ~(1ull << (skip & 63));
      R[w] |= m; fr[w] = m; any |= m;
    }
    if (!any) break;
  }
}
GE_DEV void ge_reach_thread(const uint64_t *rows, int W, const uint64_t *alive, int start, int skip, uint64_t *R) {
  uint64_t fr[GE_MAXW], nx[GE_MAXW];
  ge_reach_sets(rows, W, alive, start, skip, R, fr, nx);
}
GE_DEV void ge_reach_thread_mem(const uint64_t *rows, int W, const uint64_t *alive, int start, int skip, uint64_t *R, uint64_t *fr, uint64_t *nx) {
  ge_reach_sets(rows, W, alive, start, skip, R, fr, nx);
}

// ---------------------------------------------------------------------------------------------
// bench / test policy: uniform choice among valid actions, keyed by (policy_seed, global slot, tstep)
GE_DEV uint64_t ge_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

GE_DEV int ge_nth_set_bit(uint64_t word, uint32_t r) {
  for (uint32_t k = 0; k < r; k++) word &= word - 1;
  return ge_ctz64(word);
}

// the action of slot i under that policy (-1: empty mask, or a frozen slot)
// up to eight words of a slot's row in one round trip: every load unconditional (a word past the end re-reads word 0) and the
// words that do not exist zeroed afterwards.  A `for (w < W)` over global memory with a run-time bound is one load, one wait per
// trip -- ~2 us each at one wave per CU, and the thread-per-slot step kernel of a 512-node class had forty of them in a row.
GE_DEV void ge_words8(const uint64_t *p, int W, uint64_t (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 8; j++) o[j] = p[j < W ? j : 0];
#pragma unroll
  for (int j = 0; j < 8; j++) if (j >= W) o[j] = 0ull;
}

GE_DEV int64_t ge_policy_pick(const GeParams &P, int i, uint64_t policy_seed) {
  const uint64_t *mb = P.buf.mask_bits + (int64_t)i * P.AW;
  const uint64_t packed = P.buf.slot_rec[2 * (int64_t)i + 1];
  if (P.AW <= 8) {  // the whole row in registers
    uint64_t m[8]; ge_words8(mb, P.AW, m);
    uint32_t cnt = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) cnt += (uint32_t)ge_popc64(m[w]);
    if (!cnt || ge_rec_status(packed) == 1 || ge_rec_status(packed) == 4) return -1;
    const uint64_t gi = (uint64_t)(P.env_index_base + i), ts = ge_rec_tstep(packed);
    const uint64_t z = ge_mix64(policy_seed + gi * 0x9E3779B97F4A7C15ull + ts * 0xD1B54A32D192ED03ull);
    uint32_t r = (uint32_t)(((z >> 32) * (uint64_t)cnt) >> 32);
    int64_t pick = -1;
#pragma unroll
    for (int w = 0; w < 8; w++) {
      const uint32_t pc = (uint32_t)ge_popc64(m[w]);
      if (pick < 0 && r < pc) pick = (int64_t)w * 64 + ge_nth_set_bit(m[w], r);
      if (pick < 0) r -= pc;
    }
    return pick;
  }
  uint32_t cnt = 0;
  for (int w = 0; w < P.AW; w++) cnt += (uint32_t)ge_popc64(mb[w]);
  if (!cnt || ge_rec_status(packed) == 1 || ge_rec_status(packed) == 4) return -1;
  uint64_t gi = (uint64_t)(P.env_index_base + i), ts = ge_rec_tstep(packed);
  uint64_t z = ge_mix64(policy_seed + gi * 0x9E3779B97F4A7C15ull + ts * 0xD1B54A32D192ED03ull);
  uint32_t r = (uint32_t)(((z >> 32) * (uint64_t)cnt) >> 32);
  for (int w = 0; w < P.AW; w++) {
    uint64_t word = mb[w]; uint32_t pc = (uint32_t)ge_popc64(word);
    if (r < pc) return (int64_t)w * 64 + ge_nth_set_bit(word, r);
    r -= pc;
  }
  return -1;
}

// SAMPLE: the device policy is evaluated here (one launch per rollout step) and the action is also written to actions_out.
// RAGGED (multi-class engine): PG is the engine-wide block (B = all slots, the reset queue); every thread looks up its slot's class
// and runs that class's transition on the class-local slot index -- P then lives in memory (per-lane loads), which the reset-
// dominated ragged workloads can afford.  Mask bytes are written by the slot's own thread (the classes' mask widths differ).
// PRUNE: parenting >= 2 of LongestPath / TSP (per-thread walks of the residual graph over node sets of up to GE_MAXW words, ~64
// registers of scratch sets): its own instantiation, so that the plain transitions (BASELINE config 3 runs TSP with parenting 1)
// do not carry those registers.
// PRUNE = 2: the same walks for graphs above 64 * GE_MAXW nodes -- the node sets live in the slot's prune_scratch (four sets of W
// words in global memory) instead of registers: slow, but these are graphs on which the reference itself takes seconds per step.
template <int ENV, bool SAMPLE, bool RAGGED, int PRUNE>
GE_KERNEL ge_k_step(GeParams PG, GeRagged R, const int64_t *actions, uint64_t policy_seed) {
  const int tid = ge_tid();
  const int i0 = ge_bid() * ge_bdim();
  const int ig = i0 + tid;  // global slot
  int cls = 0, lo = 0;
  if constexpr (RAGGED) { if (ig < PG.B) { cls = R.slot_class[ig]; lo = R.class_start[cls]; } }
  const GeParams &P = RAGGED ? R.classes[cls] : PG;
  const ge_buffers &G = P.buf;
  const int i = ig - lo;    // slot inside its class (== ig in a uniform engine)
  const int n = P.n, W = P.W, F = P.F, A = P.A, AW = P.AW;
  const int WS = RAGGED ? PG.W : W;  // words per thread in the LDS stage (the widest class)
  constexpr int t = ENV;  // one instantiation per env type: the simple envs do not carry the others' registers
  uint64_t *stage = (uint64_t *)ge_dyn_smem();  // [blockDim][WS] new node masks (node-action envs)
  const bool edge_mask = (t == GE_STEINER_TREE || t == GE_MULTICAST_ROUTING);
  bool wrote_mask = false;  // this slot's node mask changed and sits in `stage`

  bool want_reset = false, want_swap = false;

  if (ig < PG.B) {
    const int64_t nbase = (int64_t)i * n;
    int64_t a64;
    if (SAMPLE) { a64 = ge_policy_pick(P, i, policy_seed); if (PG.buf.actions_out) PG.buf.actions_out[ig] = a64; } else a64 = actions[ig];
    const ulonglong2 rec = ((const ulonglong2 *)G.slot_rec)[i];
    double cost = ge_u64_as_f64(rec.x);
    const int st = ge_rec_status(rec.y);
    const int head = ge_rec_head(rec.y);
    int head_out = head;
    double reward = 0.0; int done = 0, solved = -1, invalid = 0; bool acted = false;
    bool cost_hidden = false;  // multicast: info['solution_cost'] stays -1 unless the episode is solved
    bool cost_lagged = false; double cost_before = 0.0;  // perishable delivery: info['solution_cost'] is read before the move
    if (st != 0 || a64 == -1) {
      // frozen slot (finished, autoreset off), a slot regenerated at the start of this step (next-step autoreset: its
      // action is ignored) or an explicit no-op: nothing moves
    } else {
      bool in_range = a64 >= 0 && a64 < (int64_t)A;
      int a = in_range ? (int)a64 : 0;
      bool mbit = in_range && ((G.mask_bits[(int64_t)i * AW + (a >> 6)] >> (a & 63)) & 1ull);
      switch (t) {
        case GE_SHORTEST_PATH:
        case GE_LONGEST_PATH: {
          const bool lp = (t == GE_LONGEST_PATH);
          if (!mbit) { invalid = 1; break; }
          bool nbr = (G.adj_bits[(nbase + head) * W + (a >> 6)] >> (a & 63)) & 1ull;
          bool vis_a = (G.node_bits[(int64_t)i * W + (a >> 6)] >> (a & 63)) & 1ull;
          if (lp && P.parenting >= 1 && (!nbr || vis_a)) { invalid = 1; break; }
          acted = true;
          int code = nbr ? ge_edge_code(P, i, head, a) : -1;
          double wgt = (code >= 0) ? ge_wlut(code) : 0.0;  // adj[head, a] (0 when not adjacent)
          if (lp) { reward = wgt; cost -= wgt; } else { reward = -wgt; cost -= reward; }
          if (lp && (!nbr || vis_a)) {  // longest_path.py:169-173 (parenting 0 only): no state change
            done = 1; solved = 0; reward = -2.0 * n; break;
          }
          int dest = G.terminals[(int64_t)i * P.T + 1];
          if (a == dest) { done = 1; solved = 1; }
          head_out = a;
          G.x[(nbase + a) * F + 0] = 1.f;
          uint64_t any = 0;
          // node sets of the walk: registers (accessed as arrays, so that they stay there), or the slot's prune_scratch (PRUNE == 2)
          uint64_t la_[PRUNE == 1 ? GE_MAXW : 1], lr_[PRUNE == 1 ? GE_MAXW : 1];
          uint64_t *const sc_ = (PRUNE == 2) ? G.prune_scratch + (int64_t)i * 4 * W : nullptr;
#define alive(w) (*(PRUNE == 2 ? &sc_[w] : &la_[w]))
#define R(w) (*(PRUNE == 2 ? &sc_[W + (w)] : &lr_[w]))
          if (PRUNE == 0 && W <= 8) {  // the visited set and the chosen node's row in registers, one round trip (ge_words8)
            uint64_t vb8[8], row8[8];
            ge_words8(G.node_bits + (int64_t)i * W, W, vb8); ge_words8(G.adj_bits + (nbase + a) * W, W, row8);
#pragma unroll
            for (int w = 0; w < 8; w++) if (w < W) {
              if ((a >> 6) == w) { vb8[w] |= 1ull << (a & 63); G.node_bits[(int64_t)i * W + w] = vb8[w]; }
              const uint64_t nm = (lp && P.parenting == 0) ? ge_full_word(A, w) : (row8[w] & ~vb8[w]);
              stage[tid * WS + w] = nm; any |= nm;
            }
            wrote_mask = !(lp && P.parenting == 0);
            if (!done && !any) { done = 1; solved = 0; reward = lp ? -2.0 * n : -(double)n; }
            break;
          }
          for (int w = 0; w < W; w++) {
            uint64_t vb = G.node_bits[(int64_t)i * W + w];
            if ((a >> 6) == w) { vb |= 1ull << (a & 63); G.node_bits[(int64_t)i * W + w] = vb; }
            if (PRUNE == 2 || (PRUNE == 1 && w < GE_MAXW)) alive(w) = ge_full_word(n, w) & ~vb;
          }
          const bool prune = PRUNE && lp && P.parenting >= 2 && a != dest;  // longest_path.py:134-143 (dest still in alt_G)
          int n_alive = 0;
          if constexpr (PRUNE) {
            if (prune) {
              if constexpr (PRUNE == 2) ge_reach_thread_mem(G.adj_bits + nbase * W, W, sc_, dest, -1, sc_ + W, sc_ + 2 * W, sc_ + 3 * W);
              else ge_reach_thread(G.adj_bits + nbase * W, W, la_, dest, -1, lr_);
            }
            if (lp && P.parenting == 3) for (int w = 0; w < W; w++) n_alive += ge_popc64(alive(w));
          }
          for (int w = 0; w < W; w++) {
            uint64_t vb = G.node_bits[(int64_t)i * W + w];
            uint64_t nm = (lp && P.parenting == 0) ? ge_full_word(A, w) : (G.adj_bits[(nbase + a) * W + w] & ~vb);
            if constexpr (PRUNE) { if (prune) { nm &= R(w); if (P.parenting == 3 && n_alive <= n / 3) nm |= alive(w); } }
            stage[tid * WS + w] = nm; any |= nm;
          }
          wrote_mask = !(lp && P.parenting == 0);
          if (!done && !any) { done = 1; solved = 0; reward = lp ? -2.0 * n : -(double)n; }
          break;
        }
        case GE_TSP: {
          const int start = 0;
          if (a64 == start && head == start) {  // tsp.py:203-211
            acted = true; done = 1; solved = 0; reward = -(double)n; cost = -1.0; break;
          }
          if (!mbit) { invalid = 1; break; }
          acted = true;
          int code = ge_edge_code(P, i, head, a);
          double wgt = ge_wlut(code);
          if (P.spatial) wgt = G.sw64[(int64_t)i * P.E + G.row_ptr[(int64_t)i * (n + 1) + head] + ge_rank_below(G.adj_bits + (nbase + head) * W, a)];
          reward = 0.0 - wgt;
          cost = cost + wgt;
          G.x[(nbase + a) * F + 0] = 1.f;
          head_out = a;
          int taken = G.counters[i * 2] + 1;
          G.counters[i * 2] = taken;
          if (taken == n && a == start) { done = 1; solved = 1; }
          uint64_t any = 0;
          for (int w = 0; w < W; w++) {
            uint64_t vb = G.node_bits[(int64_t)i * W + w];
            if ((a >> 6) == w) { vb |= 1ull << (a & 63); G.node_bits[(int64_t)i * W + w] = vb; }
            uint64_t nm = G.adj_bits[(nbase + a) * W + w] & ~vb;
            if (taken < n - 1 && w == 0) nm &= ~1ull;  // start only once everything else is taken (tsp.py:178-179)
            stage[tid * WS + w] = nm; any |= nm;
          }
          if constexpr (PRUNE) if (P.parenting >= 2 && any) {  // tsp.py:181-194: a move must leave the untaken nodes (start excluded) connected
            uint64_t la_[PRUNE == 1 ? GE_MAXW : 1], lr_[PRUNE == 1 ? GE_MAXW : 1];
            uint64_t *const sc_ = (PRUNE == 2) ? G.prune_scratch + (int64_t)i * 4 * W : nullptr;
            int n_alive = 0;
            for (int w = 0; w < W; w++) {
              alive(w) = ge_full_word(n, w) & ~G.node_bits[(int64_t)i * W + w];
              if (w == 0) alive(w) &= ~1ull;
              n_alive += ge_popc64(alive(w));
            }
            any = 0;
            bool stop = false;
            for (int w = 0; w < W; w++) {
              uint64_t nm = stage[tid * WS + w];
              for (uint64_t cnd = nm; cnd && !stop; cnd &= cnd - 1) {
                int v = w * 64 + ge_ctz64(cnd);
                if (v == start) continue;
                if (n_alive - 1 == 0) { stop = true; break; }  // G_copy has no node left
                int from = -1;
                for (int w2 = 0; w2 < W && from < 0; w2++) { uint64_t r = alive(w2); if ((v >> 6) == w2) r &= ~(1ull << (v & 63)); if (r) from = w2 * 64 + ge_ctz64(r); }
                if constexpr (PRUNE == 2) ge_reach_thread_mem(G.adj_bits + nbase * W, W, sc_, from, v, sc_ + W, sc_ + 2 * W, sc_ + 3 * W);
                else ge_reach_thread(G.adj_bits + nbase * W, W, la_, from, v, lr_);
                int reached = 0;
                for (int w2 = 0; w2 < W; w2++) reached += ge_popc64(R(w2));
                if (reached != n_alive - 1) nm &= ~(1ull << (v & 63));
              }
              stage[tid * WS + w] = nm; any |= nm;
            }
          }
          wrote_mask = true;
          if (!done && !any) { done = 1; reward -= (double)(n * 2); solved = 0; }
          break;
        }
#undef alive
#undef R
        case GE_STEINER_TREE: {
          if (!mbit) { invalid = 1; break; }
          acted = true;
          const int64_t ebase = (int64_t)i * P.E;
          const int32_t *rp = G.row_ptr + (int64_t)i * (n + 1);
          uint16_t e = G.colw[ebase + a];
          int v = e >> 4;
          float r = -(float)ge_wlut(e & 15);
          float c32 = (float)cost; c32 -= r; cost = (double)c32;  // numpy float32 accumulator
          reward = (double)r;
          G.x[(nbase + v) * F + 0] = 1.f;
          uint64_t missing = 0;
          for (int w = 0; w < W; w++) {
            uint64_t vb = G.node_bits[(int64_t)i * W + w];
            if ((v >> 6) == w) { vb |= 1ull << (v & 63); G.node_bits[(int64_t)i * W + w] = vb; }
            missing |= G.target_bits[(int64_t)i * W + w] & ~vb;
          }
          if (!missing) { done = 1; solved = 1; }
          // incremental mask: edges leaving v open towards nodes outside the tree; edges entering v close
          uint64_t *mb = G.mask_bits + (int64_t)i * AW;
          uint8_t *mby = G.mask + (int64_t)i * A;
          for (int k = rp[v]; k < rp[v + 1]; k++) {
            int u = G.colw[ebase + k] >> 4;
            bool intree = (G.node_bits[(int64_t)i * W + (u >> 6)] >> (u & 63)) & 1ull;
            if (!intree) { mb[k >> 6] |= 1ull << (k & 63); mby[k] = 1; }
            int rk = G.rev_edge[ebase + k];
            mb[rk >> 6] &= ~(1ull << (rk & 63)); mby[rk] = 0;
          }
          break;
        }
        case GE_MULTICAST_ROUTING: {  // multicast_routing.py:191-266
          if (!mbit) { invalid = 1; break; }
          acted = true; cost_hidden = true;
          const int64_t ebase = (int64_t)i * P.E;
          const int32_t *rp = G.row_ptr + (int64_t)i * (n + 1);
          const uint16_t e = G.colw[ebase + a];
          const int v = e >> 4;
          const int u = (int)(G.edge_index[ebase + a] - P.node_id_base - nbase);
          const float delay = (float)ge_wlut(e & 15);
          float r = -delay;
          float c32 = (float)cost; c32 -= r; cost = (double)c32;  // numpy float32 accumulator
          const double fail = -2.0 * n * P.n_dests;
          uint64_t *nbits = G.node_bits + (int64_t)i * W;
          const bool has_u = (nbits[u >> 6] >> (u & 63)) & 1ull, has_v = (nbits[v >> 6] >> (v & 63)) & 1ull;
          if (!has_u || has_v) { done = 1; solved = 0; reward = fail; break; }  // :211-217 (parenting 1 only): nothing changes
          nbits[v >> 6] |= 1ull << (v & 63);
          G.x[(nbase + v) * F + 0] = 1.f;
          G.edge_attr[(ebase + a) * 2 + 1] = 1.f;
          const float dv = G.x[(nbase + u) * F + 3] + delay;  // float32 + float32
          G.x[(nbase + v) * F + 3] = dv;
          uint64_t *mb = G.mask_bits + (int64_t)i * AW;
          uint8_t *mby = G.mask + (int64_t)i * A;
          // the mask after the move (also returned by the failure exits below)
          if (P.parenting == 1) { mb[a >> 6] &= ~(1ull << (a & 63)); mby[a] = 0; }  // not taken
          else if (P.parenting == 2) {  // tree -> outside edges
            for (int k = rp[v]; k < rp[v + 1]; k++) {
              const int w = G.colw[ebase + k] >> 4;
              if (!((nbits[w >> 6] >> (w & 63)) & 1ull)) { mb[k >> 6] |= 1ull << (k & 63); mby[k] = 1; }
              const int rk = G.rev_edge[ebase + k];
              mb[rk >> 6] &= ~(1ull << (rk & 63)); mby[rk] = 0;
            }
          } else {  // per outside node the tree edge of smallest float32 distance, first index on ties (np.argmin)
            int32_t *best = G.node_aux + nbase;
            mb[a >> 6] &= ~(1ull << (a & 63)); mby[a] = 0;  // a == best[v]
            best[v] = -1;
            for (int k = rp[v]; k < rp[v + 1]; k++) {
              const uint16_t ek = G.colw[ebase + k];
              const int w = ek >> 4;
              if ((nbits[w >> 6] >> (w & 63)) & 1ull) continue;
              const float dn = dv + (float)ge_wlut(ek & 15);
              const int cur = best[w];
              bool take = cur < 0;
              if (!take) {
                const int uc = (int)(G.edge_index[ebase + cur] - P.node_id_base - nbase);
                const float dc = G.x[(nbase + uc) * F + 3] + (float)ge_wlut(G.colw[ebase + cur] & 15);
                take = dn < dc || (dn == dc && k < cur);
                if (take) { mb[cur >> 6] &= ~(1ull << (cur & 63)); mby[cur] = 0; }
              }
              if (take) { best[w] = k; mb[k >> 6] |= 1ull << (k & 63); mby[k] = 1; }
            }
          }
          const bool is_t = (G.target_bits[(int64_t)i * W + (v >> 6)] >> (v & 63)) & 1ull;
          if (is_t) {
            if (dv > G.x[(nbase + v) * F + 2] + 1e-4f) { done = 1; solved = 0; reward = fail; break; }  // :230, float32
            r += 1.f;
          }
          reward = (double)r;
          uint64_t missing = 0, any = 0;
          for (int w = 0; w < W; w++) missing |= G.target_bits[(int64_t)i * W + w] & ~nbits[w];
          for (int w = 0; w < AW; w++) any |= mb[w];
          if (!missing) { done = 1; solved = 1; cost_hidden = false; }
          else if (!any) { done = 1; solved = 0; reward = fail; }
          break;
        }
        case GE_PERISHABLE_DELIVERY: {  // perishable_product_delivery.py:198-271
          if (!mbit) { invalid = 1; break; }
          acted = true; cost_lagged = true; cost_before = cost;
          const int np_ = P.n_dests;
          const int32_t *term = G.terminals + (int64_t)i * P.T;  // pickups, then drop-offs
          int pst = G.counters[i * 2];                            // 2 bits per product: 0 waiting, 1 carried, 2 delivered
          int now = head;
          if (a == head) {  // pick up the product waiting here (pickups are distinct nodes, so np.random.choice has one candidate and draws nothing)
            int prod = -1;
            for (int p = 0; p < np_; p++) if (prod < 0 && term[p] == head && ((pst >> (2 * p)) & 3) == 0) prod = p;
            pst |= 1 << (2 * prod);
            for (int v = 0; v < n; v++) G.x[(nbase + v) * F + 1 + prod] = -1.f;  // the whole HAS_P column (:224)
            reward = 2.0;
          } else {
            const double wgt = ge_wlut(ge_edge_code(P, i, head, a));
            reward = -wgt;
            cost = cost_before + wgt;
            G.x[(nbase + head) * F + 0] = 0.f; G.x[(nbase + a) * F + 0] = 1.f;
            head_out = a; now = a;
            // :241 subtracts adj[head, action] after head became action -- the zero diagonal -- so TIME_LEFT never runs down
            for (int p = 0; p < np_; p++)
              if (((pst >> (2 * p)) & 3) == 1 && term[np_ + p] == a) {  // delivered
                reward += 2.0;
                pst = (pst & ~(3 << (2 * p))) | (2 << (2 * p));
                for (int v = 0; v < n; v++) { G.x[(nbase + v) * F + 1 + p] = 0.f; G.x[(nbase + v) * F + 6 + p] = 0.f; G.x[(nbase + v) * F + 11 + p] = 0.f; }
              }
          }
          G.counters[i * 2] = pst;
          int hs = 0;  // x[:, HAS_P].sum(): 1 per waiting product, -n per carried one
          bool waits_here = false;
          for (int p = 0; p < np_; p++) {
            const int s2 = (pst >> (2 * p)) & 3;
            hs += (s2 == 0) ? 1 : (s2 == 1 ? -n : 0);
            if (s2 == 0 && term[p] == now) waits_here = true;
          }
          if (hs == 0) { done = 1; solved = 1; reward += 2.0 * n; }
          else if (G.counters[i * 2 + 1] + 1 >= n * np_ * 50) { done = 1; solved = 0; reward = -2.0 * n * np_; }  // max_steps
          for (int w = 0; w < W; w++) {
            uint64_t nm = G.adj_bits[(nbase + now) * W + w];
            if (waits_here && (now >> 6) == w) nm |= 1ull << (now & 63);
            stage[tid * WS + w] = nm;
          }
          wrote_mask = true;
          break;
        }
        case GE_DISTRIBUTION_CENTER: {  // distribution_center.py:144-178
          if (!mbit) { invalid = 1; break; }
          acted = true;
          float r = -G.x[(nbase + a) * F + 0];
          float c32 = (float)cost; c32 -= r; cost = (double)c32;  // numpy float32 accumulator
          G.x[(nbase + a) * F + 1] = 1.f;
          uint64_t *taken = G.node_bits + (int64_t)i * W, *cov = G.cover_bits + (int64_t)i * W;
          const uint64_t *tg = G.target_bits + (int64_t)i * W, *Ra = G.range_bits + (nbase + a) * W;
          taken[a >> 6] |= 1ull << (a & 63);
          uint64_t left = 0;
          for (int w = 0; w < W; w++) {  // cover what is in range of the new centre; +1 per target covered for the first time
            const uint64_t old = cov[w], fresh = Ra[w] & ~old;
            cov[w] = old | Ra[w];
            for (uint64_t f = fresh; f; f &= f - 1) G.x[(nbase + w * 64 + ge_ctz64(f)) * F + 3] = 1.f;
            r += (float)ge_popc64(fresh & tg[w]);
            left |= tg[w] & ~(old | Ra[w]);
            stage[tid * WS + w] = (P.parenting == 2) ? 0ull : ge_full_word(A, w);
          }
          if (P.parenting == 2)  // union of the ranges of the targets still uncovered (distribution_center.py:133-135)
            for (int w0 = 0; w0 < W; w0++)
              for (uint64_t tl = tg[w0] & ~cov[w0]; tl; tl &= tl - 1) {
                const uint64_t *Rt = G.range_bits + (nbase + w0 * 64 + ge_ctz64(tl)) * W;
                for (int w = 0; w < W; w++) stage[tid * WS + w] |= Rt[w];
              }
          for (int w = 0; w < W; w++) stage[tid * WS + w] &= ~taken[w];
          wrote_mask = true;
          reward = (double)r;
          if (!left) { done = 1; solved = 1; }
          break;
        }
        case GE_DENSEST_SUBGRAPH: {
          bool taken_a = in_range && ((G.node_bits[(int64_t)i * W + (a >> 6)] >> (a & 63)) & 1ull);
          if (!mbit || taken_a) { invalid = 1; break; }
          acted = true; solved = 1;
          if (a == n - 1) { reward = 0.0; done = 1; break; }  // stop action: densest_subgraph.py:148-154
          int k = G.counters[i * 2], ecnt = G.counters[i * 2 + 1], new_edges = 0;
          uint64_t any = 0;
          uint64_t vb8[8], row8[8], un8[8];
          const bool regs = W <= 8;  // the slot's rows in registers, one round trip (ge_words8)
          if (regs) {
            ge_words8(G.node_bits + (int64_t)i * W, W, vb8); ge_words8(G.adj_bits + (nbase + a) * W, W, row8);
            if (P.parenting != 0) ge_words8(G.target_bits + (int64_t)i * W, W, un8);
#pragma unroll
            for (int w = 0; w < 8; w++) new_edges += ge_popc64(row8[w] & vb8[w]);
          } else
          for (int w = 0; w < W; w++) new_edges += ge_popc64(G.adj_bits[(nbase + a) * W + w] & G.node_bits[(int64_t)i * W + w]);
          reward = (k == 0) ? 0.0 : ((double)(ecnt + new_edges) / (double)(k + 1)) - ((double)ecnt / (double)k);
          ecnt += new_edges; k += 1;
          G.counters[i * 2] = k; G.counters[i * 2 + 1] = ecnt;
          G.x[(nbase + a) * F + 0] = 1.f;
          cost = (double)ecnt / (double)k;
          if (regs) {
#pragma unroll
            for (int w = 0; w < 8; w++) if (w < W) {
              if ((a >> 6) == w) { vb8[w] |= 1ull << (a & 63); G.node_bits[(int64_t)i * W + w] = vb8[w]; }
              uint64_t nm;
              if (P.parenting == 0) nm = ge_full_word(A, w) & ~vb8[w];
              else { const uint64_t un = un8[w] | row8[w]; G.target_bits[(int64_t)i * W + w] = un; nm = un & ~vb8[w]; }
              stage[tid * WS + w] = nm; any |= nm;
            }
          } else
          for (int w = 0; w < W; w++) {
            uint64_t vb = G.node_bits[(int64_t)i * W + w];
            if ((a >> 6) == w) { vb |= 1ull << (a & 63); G.node_bits[(int64_t)i * W + w] = vb; }
            uint64_t nm;
            if (P.parenting == 0) nm = ge_full_word(A, w) & ~vb;
            else {  // neighbours of the taken set: running union kept in target_bits
              uint64_t un = G.target_bits[(int64_t)i * W + w] | G.adj_bits[(nbase + a) * W + w];
              G.target_bits[(int64_t)i * W + w] = un;
              nm = un & ~vb;
            }
            stage[tid * WS + w] = nm; any |= nm;
          }
          wrote_mask = true;
          if ((double)k == P.n_choices) done = 1;
          (void)any;
          break;
        }
        case GE_MAX_INDEPENDENT_SET: {
          if (!mbit) { invalid = 1; break; }
          acted = true;
          float r = -G.x[(nbase + a) * F + 0];
          float c32 = (float)cost; c32 -= r; cost = (double)c32;
          reward = (double)r;
          G.x[(nbase + a) * F + 1] = 1.f;
          uint64_t any = 0;
          if (W <= 8) {
            uint64_t vb8[8]; ge_words8(G.node_bits + (int64_t)i * W, W, vb8);
#pragma unroll
            for (int w = 0; w < 8; w++) if (w < W) {
              if ((a >> 6) == w) { vb8[w] |= 1ull << (a & 63); G.node_bits[(int64_t)i * W + w] = vb8[w]; }
              const uint64_t nm = ge_full_word(A, w) & ~vb8[w];
              stage[tid * WS + w] = nm; any |= nm;
            }
          } else
          for (int w = 0; w < W; w++) {
            uint64_t vb = G.node_bits[(int64_t)i * W + w];
            if ((a >> 6) == w) { vb |= 1ull << (a & 63); G.node_bits[(int64_t)i * W + w] = vb; }
            uint64_t nm = ge_full_word(A, w) & ~vb;
            stage[tid * WS + w] = nm; any |= nm;
          }
          wrote_mask = true;
          if (!any) { done = 1; solved = 1; }
          break;
        }
      }
    }
    G.reward[i] = reward;
    G.terminated[i] = (uint8_t)done;
    G.invalid[i] = (uint8_t)invalid;
    G.solved[i] = (int8_t)solved;
    int st_out = (st == 3) ? 0 : st;  // a slot regenerated at the start of this step (next-step autoreset) runs from the next step on
    uint64_t ts = ge_rec_tstep(rec.y);
    if (acted) {
      int len = G.counters[i * 2 + (t == GE_DENSEST_SUBGRAPH || t == GE_TSP ? 0 : 1)];
      if (!(t == GE_DENSEST_SUBGRAPH || t == GE_TSP)) { len += 1; G.counters[i * 2 + 1] = len; }
      ts = (ts + 1) & 0xffffffffull;
      if (done) {
        G.final_cost[i] = cost_hidden ? -1.0 : (cost_lagged ? cost_before : cost);
        G.final_len[i] = len;
        if (P.autoreset != 1) G.final_heur[i] = G.heuristic[i];  // same-step autoreset: the reset kernel copies it before it overwrites `heuristic`
        if (P.autoreset) {
          want_reset = true;
          want_swap = PG.spare_state && PG.spare_state[ig];  // the slot's next episode waits in its spare image
          if (want_swap) PG.spare_state[ig] = 0;
          st_out = 2;
          if (P.autoreset == 1) wrote_mask = false;  // same-step: the reset kernel rewrites the whole slot right away
        } else {
          st_out = 1;
        }
      }
    }
    if (acted || st == 3) ((ulonglong2 *)G.slot_rec)[i] = make_ulonglong2(ge_f64_as_u64(cost), ge_rec_make(head_out, st_out, ge_rec_aux(rec.y), ts));
    if (wrote_mask && !edge_mask) for (int w = 0; w < W; w++) G.mask_bits[(int64_t)i * AW + w] = stage[tid * WS + w];
    if constexpr (RAGGED) {  // the slot's own thread expands its mask words to bool bytes
      if (wrote_mask && !edge_mask) {
        uint8_t *out = G.mask + (int64_t)i * A;
        // eight bool bytes per store where the row's bytes are 8-aligned (bit k of a byte of the set -> byte k: the multiply spreads
        // the byte, the mask picks bit k of copy k, the add-and-shift turns "non-zero" into 1), single bytes at the ragged ends
        int v = 0;
        const int head_bytes = (int)((8 - ((uintptr_t)out & 7)) & 7);
        for (; v < A && v < head_bytes; v++) out[v] = (uint8_t)((stage[tid * WS + (v >> 6)] >> (v & 63)) & 1ull);
        for (; v + 8 <= A; v += 8) {
          const int w = v >> 6, sh = v & 63;
          uint64_t b8 = stage[tid * WS + w] >> sh;
          if (sh > 56 && w + 1 < (A + 63) / 64) b8 |= stage[tid * WS + w + 1] << (64 - sh);
          b8 &= 0xffull;
          uint64_t y = (b8 * 0x0101010101010101ull) & 0x8040201008040201ull;
          y = ((y + 0x7f7f7f7f7f7f7f7full) >> 7) & 0x0101010101010101ull;
          *(uint64_t *)(out + v) = y;
        }
        for (; v < A; v++) out[v] = (uint8_t)((stage[tid * WS + (v >> 6)] >> (v & 63)) & 1ull);
      }
    }
  }

  uint8_t *flag = (uint8_t *)(stage + (size_t)ge_bdim() * WS);
  flag[tid] = wrote_mask ? 1 : 0;
  ge_enqueue_reset(PG, (int *)(flag + ge_bdim()), i0, ig, tid, want_reset, want_swap);  // contains the barrier; global slot ids
  if (edge_mask || RAGGED) return;  // SteinerTree updates its [B, 2m] mask incrementally above
  // ---- bool mask slab: [B, n] bytes, this workgroup owns the contiguous range of its slots
  int nb = P.B - i0; if (nb > ge_bdim()) nb = ge_bdim();
  if (nb <= 0) return;
  uint8_t *out = G.mask + (int64_t)i0 * A;
  if ((A & 7) == 0) {
    int groups = nb * (A >> 3);
    for (int g = tid; g < groups; g += ge_bdim()) {
      int e = (g << 3) / A, v0 = (g << 3) % A;
      if (!flag[e]) continue;
      uint64_t b8 = (stage[e * W + (v0 >> 6)] >> (v0 & 63)) & 0xffull;
      uint64_t y = (b8 * 0x0101010101010101ull) & 0x8040201008040201ull;
      y = ((y + 0x7f7f7f7f7f7f7f7full) >> 7) & 0x0101010101010101ull;
      *(uint64_t *)(out + ((int64_t)g << 3)) = y;
    }
  } else {
    int total = nb * A;
    for (int idx = tid; idx < total; idx += ge_bdim()) {
      int e = idx / A, v = idx % A;
      if (!flag[e]) continue;
      out[idx] = (uint8_t)((stage[e * W + (v >> 6)] >> (v & 63)) & 1ull);
    }
  }
}

// Edge-action envs (SteinerTree, MulticastRouting: one action per directed edge, [B, 2m] masks), a QUAD of lanes per slot.
// A workgroup is 1 024 threads for the same 256 slots a thread-per-slot workgroup would own (the reset queue keeps its layout) --
// four times the waves to hide the five dependent memory round trips of a transition (slot state -> chosen edge -> row extents ->
// row -> reverse edges), and every per-slot access a quad makes is one 32-byte sector instead of four:
//  * the workgroup's 256 mask rows (contiguous in mask_bits) and node sets are staged in LDS with coalesced loads;
//  * SAMPLE: the device policy's "r-th set bit of the row" is a popcount per lane over the words q, q + 4, ... and a prefix over
//    the quad (DPP), not a scan of 32 words per thread;
//  * the row of the node that joins the tree is walked four edges at a time; mask words are updated with LDS atomics and only the
//    words that changed are written back, together with the bool bytes of the edges that changed (an edge v -> u opens iff u is
//    outside the tree; the reverse edge u -> v was open iff u is inside: only those are touched).
// Replaces step() / _get_mask() of steiner_tree.py:116-157 and multicast_routing.py:155-266 for graphs whose rows fit the LDS stage
// (ge_edge_fits); larger ones keep the thread-per-slot kernel above.
#define GE_EDGE_LPS 4
#define GE_EDGE_THREADS (GE_STEP_BLOCK * GE_EDGE_LPS)
GE_HOSTDEV int ge_edge_row_stride(int AW) { return AW + 4; }  // + 32 bytes: the quads of a wave start their rows in different banks
GE_HOSTDEV size_t ge_edge_lds_bytes(int AW, int W) { return (size_t)GE_STEP_BLOCK * (size_t)(ge_edge_row_stride(AW) + W) * 8 + 2 * (GE_EDGE_THREADS / 64) * 4 + 64; }
GE_HOSTDEV bool ge_edge_fits(int AW, int W) { return AW <= 32 && ge_edge_lds_bytes(AW, W) <= 100 * 1024; }

GE_DEV uint32_t ge_quad_or32(uint32_t v) { v |= ge_quad_xor1(v); v |= ge_quad_xor2(v); return v; }
GE_DEV uint64_t ge_quad_or64(uint64_t v) { return (uint64_t)ge_quad_or32((uint32_t)v) | ((uint64_t)ge_quad_or32((uint32_t)(v >> 32)) << 32); }

// the words of a slot's mask row that changed (bit w of `dirty`), dealt over the lanes of the quad
GE_DEV void ge_edge_writeback(uint64_t *dst, const uint64_t *row, uint32_t dirty, int q) {
  int idx = 0;
  for (uint32_t m = dirty; m; m &= m - 1, idx++) if ((idx & (GE_EDGE_LPS - 1)) == q) { const int w = (int)__builtin_ctz(m); dst[w] = row[w]; }
}

template <int ENV, bool SAMPLE>
GE_KERNEL_LB(GE_EDGE_THREADS, 1) ge_k_step_edge(GeParams P, const int64_t *actions, uint64_t policy_seed) {
  static_assert(ENV == GE_STEINER_TREE || ENV == GE_MULTICAST_ROUTING, "edge-action envs");
  const ge_buffers &G = P.buf;
  const int tid = ge_tid(), q = tid & (GE_EDGE_LPS - 1);
  const int i0 = ge_bid() * GE_STEP_BLOCK, sl = tid >> 2, i = i0 + sl;  // slot of this quad
  const int n = P.n, W = P.W, F = P.F, A = P.A, AW = P.AW, RS = ge_edge_row_stride(AW);
  uint64_t *rows = (uint64_t *)ge_dyn_smem();           // [256][RS] mask rows
  uint64_t *nbs = rows + (size_t)GE_STEP_BLOCK * RS;    // [256][W] node sets (in tree / has the message)
  int *wcnt = (int *)(nbs + (size_t)GE_STEP_BLOCK * W);
  int nb = P.B - i0; if (nb > GE_STEP_BLOCK) nb = GE_STEP_BLOCK;
  // ---- stage: the workgroup's mask rows and node sets are contiguous in HBM
  for (int idx = tid; idx < nb * AW; idx += GE_EDGE_THREADS) { const int e = idx / AW, w = idx - e * AW; rows[e * RS + w] = G.mask_bits[(int64_t)i0 * AW + idx]; }
  for (int idx = tid; idx < nb * W; idx += GE_EDGE_THREADS) nbs[idx] = G.node_bits[(int64_t)i0 * W + idx];
  ge_sync();
  uint64_t *row = rows + sl * RS, *nbits = nbs + sl * W;
  bool want_reset = false, want_swap = false;
  if (i < P.B) {
    const int64_t nbase = (int64_t)i * n, ebase = (int64_t)i * P.E;
    const ulonglong2 rec = ((const ulonglong2 *)G.slot_rec)[i];
    ge_quad_sync();  // the four lanes have read the record before lane 0 may rewrite it (lockstep on the GPU; the CPU harness runs lanes one after another)
    const int st = ge_rec_status(rec.y);
    uint64_t ts = ge_rec_tstep(rec.y);
    // ---- action
    int64_t a64;
    if (SAMPLE) {
      // words q, q + 4, ...: chunk j of the row is the words 4 j .. 4 j + 3, one per lane, so word order is (chunk, lane) order
      uint64_t wd[8]; uint32_t cnt = 0, ctot[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        wd[j] = (4 * j + q < AW) ? row[4 * j + q] : 0ull;
        const uint64_t g = ge_quad_gather16((uint32_t)ge_popc64(wd[j]));
        ctot[j] = (uint32_t)((g & 0xffffu) + ((g >> 16) & 0xffffu) + ((g >> 32) & 0xffffu) + (g >> 48));
        cnt += ctot[j];
      }
      a64 = -1;
      if (cnt && st != 1 && st != 4) {
        const uint64_t z = ge_mix64(policy_seed + (uint64_t)(P.env_index_base + i) * 0x9E3779B97F4A7C15ull + ts * 0xD1B54A32D192ED03ull);
        uint32_t r = (uint32_t)(((z >> 32) * (uint64_t)cnt) >> 32);
        int jsel = 0;
#pragma unroll
        for (int j = 0; j < 7; j++) if (jsel == j && r >= ctot[j]) { r -= ctot[j]; jsel = j + 1; }
        uint64_t wsel = wd[0];
#pragma unroll
        for (int j = 1; j < 8; j++) if (jsel == j) wsel = wd[j];
        const uint32_t mine = (uint32_t)ge_popc64(wsel);
        const uint64_t g = ge_quad_gather16(mine);
        const uint32_t c0 = (uint32_t)(g & 0xffffu), c1 = (uint32_t)((g >> 16) & 0xffffu), c2 = (uint32_t)((g >> 32) & 0xffffu);
        const int qsel = r < c0 ? 0 : (r < c0 + c1 ? 1 : (r < c0 + c1 + c2 ? 2 : 3));
        const uint32_t before = (q > 0 ? c0 : 0u) + (q > 1 ? c1 : 0u) + (q > 2 ? c2 : 0u);
        const uint32_t rr = (q == qsel) ? r - before : 0u;
        const uint32_t bit = (q == qsel) ? (uint32_t)ge_nth_set_bit(wsel, rr) : 0u;
        const uint64_t gb = ge_quad_gather16(bit);
        a64 = (int64_t)(4 * jsel + qsel) * 64 + (int64_t)((gb >> (16 * qsel)) & 0xffffu);
      }
      if (q == 0 && G.actions_out) G.actions_out[i] = a64;
    } else {
      a64 = actions[i];
    }
    double cost = ge_u64_as_f64(rec.x);
    double reward = 0.0; int done = 0, solved = -1, invalid = 0; bool acted = false, cost_hidden = false;
    if (st == 0 && a64 != -1) {
      const bool in_range = a64 >= 0 && a64 < (int64_t)A;
      const int a = in_range ? (int)a64 : 0;
      const bool mbit = in_range && ((row[a >> 6] >> (a & 63)) & 1ull);
      if (!mbit) invalid = 1;
      else {
        acted = true;
        const uint16_t ea = G.colw[ebase + a];
        const int v = ea >> 4;
        const float delay = (float)ge_wlut(ea & 15);
        float r = -delay;
        float c32 = (float)cost; c32 -= r; cost = (double)c32;  // numpy float32 accumulator
        uint8_t *mby = G.mask + (int64_t)i * A;
        const int32_t *rp = G.row_ptr + (int64_t)i * (n + 1);
        if constexpr (ENV == GE_STEINER_TREE) {  // steiner_tree.py:123-157
          const int r0 = rp[v], r1 = rp[v + 1];
          reward = (double)r;
          if (q == 0) { G.x[(nbase + v) * F + 0] = 1.f; nbits[v >> 6] |= 1ull << (v & 63); }
          ge_quad_sync();
          uint32_t dirty = 0;  // words of the row that changed
          for (int k = r0 + q; k < r1; k += GE_EDGE_LPS) {
            const int u = G.colw[ebase + k] >> 4;
            if (!((nbits[u >> 6] >> (u & 63)) & 1ull)) { atomicOr((unsigned long long *)&row[k >> 6], 1ull << (k & 63)); mby[k] = 1; dirty |= 1u << (k >> 6); }  // v -> u opens
            else {  // u -> v closes (it was open: u in the tree, v outside until now)
              const int rk = G.rev_edge[ebase + k];
              atomicAnd((unsigned long long *)&row[rk >> 6], ~(1ull << (rk & 63))); mby[rk] = 0; dirty |= 1u << (rk >> 6);
            }
          }
          ge_quad_sync();
          ge_edge_writeback(G.mask_bits + (int64_t)i * AW, row, ge_quad_or32(dirty), q);
          uint64_t missing = 0;
          for (int w = q; w < W; w += GE_EDGE_LPS) { missing |= G.target_bits[(int64_t)i * W + w] & ~nbits[w]; G.node_bits[(int64_t)i * W + w] = nbits[w]; }
          if (!ge_quad_or64(missing)) { done = 1; solved = 1; }
        } else {  // multicast_routing.py:191-266
          cost_hidden = true;
          const int u = (int)(G.edge_index[ebase + a] - P.node_id_base - nbase);
          const double fail = -2.0 * n * P.n_dests;
          const bool has_u = (nbits[u >> 6] >> (u & 63)) & 1ull, has_v = (nbits[v >> 6] >> (v & 63)) & 1ull;
          if (!has_u || has_v) { done = 1; solved = 0; reward = fail; }  // :211-217 (parenting 1 only): nothing changes
          else {
            const float dv = G.x[(nbase + u) * F + 3] + delay;  // float32 + float32
            ge_quad_sync();  // (every lane has read the sets before lane 0 changes them)
            if (q == 0) {
              nbits[v >> 6] |= 1ull << (v & 63);
              G.x[(nbase + v) * F + 0] = 1.f; G.x[(nbase + v) * F + 3] = dv;
              G.edge_attr[(ebase + a) * 2 + 1] = 1.f;
            }
            ge_quad_sync();
            const int r0 = rp[v], r1 = rp[v + 1];
            // the mask after the move (also returned by the failure exits below)
            if (P.parenting == 1) {  // not taken
              if (q == 0) { row[a >> 6] &= ~(1ull << (a & 63)); mby[a] = 0; G.mask_bits[(int64_t)i * AW + (a >> 6)] = row[a >> 6]; }
            } else if (P.parenting == 2) {  // tree -> outside edges
              uint32_t dirty = 0;
              for (int k = r0 + q; k < r1; k += GE_EDGE_LPS) {
                const int w = G.colw[ebase + k] >> 4;
                if (!((nbits[w >> 6] >> (w & 63)) & 1ull)) { atomicOr((unsigned long long *)&row[k >> 6], 1ull << (k & 63)); mby[k] = 1; dirty |= 1u << (k >> 6); }
                else {
                  const int rk = G.rev_edge[ebase + k];
                  atomicAnd((unsigned long long *)&row[rk >> 6], ~(1ull << (rk & 63))); mby[rk] = 0; dirty |= 1u << (rk >> 6);
                }
              }
              ge_quad_sync();
              ge_edge_writeback(G.mask_bits + (int64_t)i * AW, row, ge_quad_or32(dirty), q);
            } else {  // per outside node the tree edge of smallest float32 distance, first index on ties (np.argmin)
              int32_t *best = G.node_aux + nbase;
              uint32_t dirty = 1u << (a >> 6);
              if (q == 0) { row[a >> 6] &= ~(1ull << (a & 63)); mby[a] = 0; best[v] = -1; }  // a == best[v]
              ge_quad_sync();
              for (int k = r0 + q; k < r1; k += GE_EDGE_LPS) {  // the neighbours of a row are distinct nodes: the lanes update different entries of best[]
                const uint16_t ek = G.colw[ebase + k];
                const int w = ek >> 4;
                if ((nbits[w >> 6] >> (w & 63)) & 1ull) continue;
                const float dn = dv + (float)ge_wlut(ek & 15);
                const int cur = best[w];
                bool take = cur < 0;
                if (!take) {
                  const int uc = (int)(G.edge_index[ebase + cur] - P.node_id_base - nbase);
                  const float dc = G.x[(nbase + uc) * F + 3] + (float)ge_wlut(G.colw[ebase + cur] & 15);
                  take = dn < dc || (dn == dc && k < cur);
                  if (take) { atomicAnd((unsigned long long *)&row[cur >> 6], ~(1ull << (cur & 63))); mby[cur] = 0; dirty |= 1u << (cur >> 6); }
                }
                if (take) { best[w] = k; atomicOr((unsigned long long *)&row[k >> 6], 1ull << (k & 63)); mby[k] = 1; dirty |= 1u << (k >> 6); }
              }
              ge_quad_sync();
              ge_edge_writeback(G.mask_bits + (int64_t)i * AW, row, ge_quad_or32(dirty), q);
            }
            ge_quad_sync();
            const bool is_t = (G.target_bits[(int64_t)i * W + (v >> 6)] >> (v & 63)) & 1ull;
            bool late = false;
            if (is_t) {
              if (dv > G.x[(nbase + v) * F + 2] + 1e-4f) late = true;  // :230, float32
              else r += 1.f;
            }
            uint64_t missing = 0, any = 0;
            for (int w = q; w < W; w += GE_EDGE_LPS) { missing |= G.target_bits[(int64_t)i * W + w] & ~nbits[w]; G.node_bits[(int64_t)i * W + w] = nbits[w]; }
            for (int w = q; w < AW; w += GE_EDGE_LPS) any |= row[w];
            missing = ge_quad_or64(missing); any = ge_quad_or64(any);
            reward = (double)r;
            if (late) { done = 1; solved = 0; reward = fail; }
            else if (!missing) { done = 1; solved = 1; cost_hidden = false; }
            else if (!any) { done = 1; solved = 0; reward = fail; }
          }
        }
      }
    }
    if (q == 0) {
      G.reward[i] = reward;
      G.terminated[i] = (uint8_t)done;
      G.invalid[i] = (uint8_t)invalid;
      G.solved[i] = (int8_t)solved;
      int st_out = (st == 3) ? 0 : st;  // a slot regenerated at the start of this step (next-step autoreset) runs from the next step on
      if (acted) {
        const int len = G.counters[i * 2 + 1] + 1;
        G.counters[i * 2 + 1] = len;
        ts = (ts + 1) & 0xffffffffull;
        if (done) {
          G.final_cost[i] = cost_hidden ? -1.0 : cost;
          G.final_len[i] = len;
          if (P.autoreset != 1) G.final_heur[i] = G.heuristic[i];
          if (P.autoreset) {
            want_reset = true;
            want_swap = P.spare_state && P.spare_state[i];
            if (want_swap) P.spare_state[i] = 0;
            st_out = 2;
          } else st_out = 1;
        }
      }
      if (acted || st == 3) ((ulonglong2 *)G.slot_rec)[i] = make_ulonglong2(ge_f64_as_u64(cost), ge_rec_make(ge_rec_head(rec.y), st_out, ge_rec_aux(rec.y), ts));
    }
  }
  ge_enqueue_reset(P, wcnt, i0, i, tid, want_reset, want_swap);  // contains the barrier; only lane 0 of a quad ever wants
}

// Headline fast path: ShortestPath / LongestPath(parenting 0,1) with n <= 64.  One u64 per node set.  The kernel is
// three phases per slot: (A) every load -- the coalesced slot state (one 16-byte record {cost, packed head / status /
// destination / step count}, the visited set, the mask), then ONE 16-byte gather for the record of the chosen node;
// (B) the transition in registers; (C) every store.  No load is issued after the first store: vmcnt counts loads and
// stores in order, so a late load (or a register reused at a control-flow join) would make the wave wait for its stores
// to be acknowledged in the middle of the kernel.  The graphs are undirected, so the weight of the move head -> a is read
// from a's record (rank of the head among a's neighbours): the head's own record is never needed, and nothing about the
// head is carried in the slot state beyond its index.  Optional fused sampling of the random policy (SAMPLE) so a rollout
// step is a single launch.
#ifndef GE_ABL
#define GE_ABL 0  // diagnostic ablation bits (tools/step_variants.py, results are wrong by construction); 0 when shipped
#endif
#define GE_ON(bit) (!(GE_ABL & (bit)))  // 1 x flag, 2 bool-mask bytes, 4 gather, 8 policy, 16 state stores, 64 output stores
template <bool SAMPLE, bool SPARES>
GE_KERNEL ge_k_step_path64(GeParams P, const int64_t *actions_in, uint64_t policy_seed) {
  const ge_buffers &G = P.buf;
  const int tid = ge_tid();
  const int i0 = ge_bid() * ge_bdim();
  const int i = i0 + tid;
  const int n = P.n, F = P.F;
  const bool lp = (P.env_type == GE_LONGEST_PATH), open_mask = lp && P.parenting == 0;
  uint64_t *stage = (uint64_t *)ge_dyn_smem();
  uint8_t *flag = (uint8_t *)(stage + ge_bdim());
  bool wrote_mask = false;
  bool want_reset = false, want_swap = false;
  if (i < P.B) {
    // ---- phase A, round 1: slot state (coalesced)
    const ulonglong2 rec = ((const ulonglong2 *)G.slot_rec)[i];
    const uint64_t mb = G.mask_bits[i];
    const uint64_t vis0 = G.node_bits[i];
    uint8_t spare = 0;  // 1: the slot's next episode waits in its spare image
    if constexpr (SPARES) spare = P.spare_state[i];
    const int head = (int)(rec.y & 63ull), st = ge_rec_status(rec.y), dest = ge_rec_aux(rec.y);
    const uint64_t ts = ge_rec_tstep(rec.y);
    const double cost0 = ge_u64_as_f64(rec.x);
    int64_t a64;
    if (SAMPLE) {
      const uint32_t cnt = (uint32_t)ge_popc64(mb);
      if (!cnt || st == 1 || st == 4) a64 = -1;
      else if (!GE_ON(8)) a64 = ge_ctz64(mb);
      else {
        uint64_t z = ge_mix64(policy_seed + (uint64_t)(P.env_index_base + i) * 0x9E3779B97F4A7C15ull + ts * 0xD1B54A32D192ED03ull);
        a64 = ge_nth_set_bit(mb, (uint32_t)(((z >> 32) * (uint64_t)cnt) >> 32));
      }
    } else {
      a64 = actions_in[i];
    }
    const bool in_range = a64 >= 0 && a64 < (int64_t)n;
    const int a = in_range ? (int)a64 : 0;
    // ---- phase A, round 2: the record of the chosen node (its bit row and the weight codes of its 16 smallest neighbours)
    const int64_t nbase = (int64_t)i * n;
    const ulonglong2 arec = GE_ON(4) ? ((const ulonglong2 *)G.node_rec)[nbase + a] : make_ulonglong2(mb * 3, 0x3333333333333333ull);
    const bool nbr = (arec.x >> head) & 1ull;                         // adjacency is symmetric
    const int rank = ge_popc64(arec.x & ((1ull << head) - 1ull));      // position of the head among a's neighbours
    int code = (int)((arec.y >> (4 * (rank & 15))) & 15ull);
    if (rank > 15) {  // rare: a has more than 16 neighbours and the head is not among the 16 smallest -> rank-indexed byte in the scode slab
      int pos = G.row_ptr[(int64_t)i * (n + 1) + a] + rank;
      if (pos >= P.E) pos = P.E - 1;
      code = G.scode[(int64_t)i * P.E + pos];
    }

    // ---- phase B: the transition, in registers (shortest_path.py:101-141, longest_path.py:121-196)
    double reward = 0.0, cost = cost0;
    int done = 0, solved = -1, invalid = 0;
    bool acted = false, moved = false;
    uint64_t vis = vis0, nm = mb;
    if (st == 0 && a64 != -1) {
      const bool mbit = in_range && ((mb >> a) & 1ull);
      const bool vis_a = (vis0 >> a) & 1ull;
      if (!mbit || (lp && P.parenting >= 1 && (!nbr || vis_a))) invalid = 1;
      else {
        acted = true;
        const double wgt = nbr ? ge_wlut(code) : 0.0;
        if (lp) { reward = wgt; cost -= wgt; } else { reward = -wgt; cost -= reward; }
        if (lp && (!nbr || vis_a)) { done = 1; solved = 0; reward = -2.0 * n; }  // longest_path.py:169-173
        else {
          moved = true;
          if (a == dest) { done = 1; solved = 1; }
          vis = vis0 | (1ull << a);
          nm = open_mask ? mb : (arec.x & ~vis);
          if (!done && !nm) { done = 1; solved = 0; reward = lp ? -2.0 * n : -(double)n; }
        }
      }
    }
    const bool fin = acted && done;
    want_reset = fin && P.autoreset;
    want_swap = want_reset && spare;
    wrote_mask = moved && !open_mask && !(want_reset && P.autoreset == 1);  // same-step: the reset kernel rewrites the slot right away
    stage[tid] = nm;
    // a slot regenerated at the start of this step (status 3, next-step autoreset) ignored its action and runs from the next step on
    const int st_out = fin ? (P.autoreset ? 2 : 1) : (st == 3 ? 0 : st);
    // next-step autoreset / no autoreset: the baseline of the episode that ends here (same-step: the reset kernel copies it)
    double heur = 0.0;
    if (fin && P.autoreset != 1) heur = G.heuristic[i];

    // ---- phase C: stores only
    ge_wait_loads();
    if (GE_ON(64)) {
      if (SAMPLE && G.actions_out) G.actions_out[i] = a64;
      G.reward[i] = reward;
      G.terminated[i] = (uint8_t)done;
      G.invalid[i] = (uint8_t)invalid;
      G.solved[i] = (int8_t)solved;
    }
    if ((acted || st == 3) && GE_ON(16))
      ((ulonglong2 *)G.slot_rec)[i] = make_ulonglong2(ge_f64_as_u64(cost), ge_rec_make(moved ? a : head, st_out, dest, acted ? ((ts + 1) & 0xffffffffull) : ts));
    if (moved) {
      if (GE_ON(1)) G.x[(nbase + a) * F + 0] = 1.f;
      if (GE_ON(16)) G.node_bits[i] = vis;
    }
    if (wrote_mask && GE_ON(16)) G.mask_bits[i] = nm;
    if constexpr (SPARES) { if (want_swap) P.spare_state[i] = 0; }
    if (fin) {
      G.final_cost[i] = cost;
      G.final_len[i] = ge_popc64(vis0);  // every move adds one node to the visited set, which starts as {source}
      if (P.autoreset != 1) G.final_heur[i] = heur;
    }
  }
  flag[tid] = wrote_mask ? 1 : 0;
  ge_enqueue_reset<SPARES>(P, (int *)(flag + ge_bdim()), i0, i, tid, want_reset, want_swap);  // contains the barrier
  int nb = P.B - i0; if (nb > ge_bdim()) nb = ge_bdim();
  if (nb <= 0 || !GE_ON(2)) return;
  uint8_t *out = G.mask + (int64_t)i0 * n;
  if ((n & 7) == 0) {
    const int groups = nb * (n >> 3);
    for (int g = tid; g < groups; g += ge_bdim()) {
      const int e = (g << 3) / n, v0 = (g << 3) % n;
      if (!flag[e]) continue;
      const uint64_t b8 = (stage[e] >> v0) & 0xffull;
      uint64_t y = (b8 * 0x0101010101010101ull) & 0x8040201008040201ull;
      y = ((y + 0x7f7f7f7f7f7f7f7full) >> 7) & 0x0101010101010101ull;
      *(uint64_t *)(out + ((int64_t)g << 3)) = y;
    }
  } else {
    const int total = nb * n;
    for (int idx = tid; idx < total; idx += ge_bdim()) {
      const int e = idx / n, v = idx % n;
      if (!flag[e]) continue;
      out[idx] = (uint8_t)((stage[e] >> v) & 1ull);
    }
  }
}

// DistributionCenter, n <= 64: the coverage range of the node each slot is about to choose, unless the row already exists
// (a target's, or a node chosen earlier in the episode).  64-thread workgroups, one lane per slot, columns in LDS.
GE_KERNEL ge_k_dc_range(GeParams P, const int64_t *actions) {
  const int lane = ge_tid(), i = ge_bid() * GE_WAVE + lane, n = P.n;
  double *S = (double *)ge_dyn_smem();
  uint8_t *stk = (uint8_t *)(S + n * GE_WAVE);
  if (i >= P.B) return;
  const int64_t a64 = actions[i];
  if (ge_rec_status(P.buf.slot_rec[2 * (int64_t)i + 1]) != 0 || a64 < 0 || a64 >= (int64_t)n) return;
  const int a = (int)a64;
  if (!((P.buf.mask_bits[i] >> a) & 1ull)) return;  // the step kernel will flag it invalid
  const uint64_t have = P.buf.aux_bits[i];
  if ((have >> a) & 1ull) return;
  const int64_t nbase = (int64_t)i * n;
  // the same label-correcting search as ge_dc_search, over node_rec: one 16-byte gather per relaxed node gives its neighbours
  // (bit row) and their weight codes (nibbles in ascending-neighbour order); rows with more than 16 neighbours read scode
  double *Sc = S + lane; uint8_t *st = stk + lane;
  const double cutoff = P.max_distance;
  for (int v = 0; v < n; v++) Sc[v * GE_WAVE] = __builtin_inf();
  Sc[a * GE_WAVE] = 0.0;
  // by rounds over a 64-bit frontier that only holds labels which can still be extended by the smallest delay (ge_dc_search,
  // ge_reset.h, has the reasons): ~7 relaxed nodes per search where the LIFO stack of rounds 2-3 relaxed 40-100
  (void)st;
  uint64_t reached = 1ull << a, cur = 1ull << a;
  while (cur) {
    uint64_t nxt = 0;
    for (; cur; cur &= cur - 1) {
      const int u = ge_ctz64(cur);
      const double du = Sc[u * GE_WAVE];
      const ulonglong2 rec = ((const ulonglong2 *)P.buf.node_rec)[nbase + u];
      const bool wide = ge_popc64(rec.x) > 16;
      const uint8_t *codes = wide ? P.buf.scode + (int64_t)i * P.E + P.buf.row_ptr[(int64_t)i * (n + 1) + u] : nullptr;
      int k = 0;
      for (uint64_t r = rec.x; r; r &= r - 1, k++) {
        const int v = ge_ctz64(r);
        const int code = wide ? (int)codes[k] : (int)((rec.y >> (4 * k)) & 15ull);
        const double d = du + ge_wlut(code);
        if (d <= cutoff && d < Sc[v * GE_WAVE]) {
          Sc[v * GE_WAVE] = d; reached |= 1ull << v;
          if (d + 0.3 <= cutoff) nxt |= 1ull << v;
        }
      }
    }
    cur = nxt;
  }
  P.buf.range_bits[nbase + a] = reached;
  P.buf.aux_bits[i] = have | (1ull << a);
}

template <bool RAGGED>
GE_KERNEL ge_k_sample(GeParams P, GeRagged R, uint64_t policy_seed, int64_t *actions) {
  int i = ge_bid() * ge_bdim() + ge_tid();
  if (i >= P.B) return;
  if constexpr (RAGGED) { const int cls = R.slot_class[i]; actions[i] = ge_policy_pick(R.classes[cls], i - R.class_start[cls], policy_seed); }
  else actions[i] = ge_policy_pick(P, i, policy_seed);
}

// utils.vectorize_graph for every slot (utils.py:87-88): [x.ravel | edge_attr.ravel | links.ravel] as f32
GE_KERNEL ge_k_vectorize(GeParams P, float *out) {
  const int64_t L = (int64_t)P.n * P.F + (int64_t)P.E * P.Fe + 2 * (int64_t)P.E;
  const int64_t p1 = (int64_t)P.n * P.F, p2 = p1 + (int64_t)P.E * P.Fe;
  const int64_t total = (int64_t)P.B * L, Ne = P.edge_row_stride;
  for (int64_t idx = (int64_t)ge_bid() * ge_bdim() + ge_tid(); idx < total; idx += (int64_t)ge_gdim() * ge_bdim()) {
    int64_t b = idx / L, j = idx % L; float v;
    if (j < p1) v = P.buf.x[b * p1 + j];
    else if (j < p2) v = P.buf.edge_attr[b * (int64_t)P.E * P.Fe + (j - p1)];
    else { int64_t q = j - p2, e = q >> 1; v = (float)(P.buf.edge_index[(q & 1) * Ne + b * P.E + e] - b * P.n - P.node_id_base); }
    out[idx] = v;
  }
}
