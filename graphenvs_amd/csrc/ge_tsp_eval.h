// Baselines that run as sequential programs, one lane per slot (is_eval_env only).
// TSP-v0: info['heuristic_solution'] = length of a Christofides tour (tsp.py:114-117 calls
// nx.approximation.traveling_salesman_problem: all-pairs Dijkstra, Christofides on the metric closure).  Two launches behind the
// graph kernel, on the slabs it wrote, outside the reset kernels so that the training path (no baseline) carries none of this:
//   ge_k_tsp_closure  one wave per regenerated slot, one LANE per Dijkstra source: integer distances (weight codes = tenths;
//                     spatial: Euclidean lengths in 1/65536) into the slot's scratch block, D[n][n];
//   ge_k_tsp_tour     one LANE per regenerated slot runs ge_christofides.h on that block (spanning tree, exact minimum-weight
//                     perfect matching of the odd nodes by the blossom algorithm, Eulerian circuit, shortcutting) and replaces the
//                     double-tree value 2 x MST the graph kernel left in heuristic[].
// This is an evaluation-time path (the reference spends seconds per reset in networkx here): lanes run unrelated sequential
// programs, nothing is tuned beyond keeping every slot of a launch busy side by side.
#pragma once
#include "ge_params.h"
#include "ge_reset.h"

#ifdef GE_EMU
#define GE_CH_FN static inline
#define GE_CR_FN static inline
#else
#define GE_CH_FN static __device__
#define GE_CH_HD static __host__ __device__ inline
#define GE_CR_FN static __device__
#define GE_CR_HD static __host__ __device__ inline
#endif
#include "ge_christofides.h"
#include "ge_clique_removal.h"
#include "ge_kou_exact.h"

#define GE_TSP_EVAL_THREADS 64

GE_DEV int32_t ge_tsp_units(const GeParams &P, double w) { return (int32_t)llrint(w * 65536.0); }

// items of the launch: every slot (full reset) or the queued ones; `pre` (LDS) holds the queue prefix in queue mode
GE_DEV int ge_tsp_count(const GeParams &P, int *pre, int queue) {
  if (!queue) return P.B;
  if (ge_tid() < GE_WAVE) ge_queue_prefix_wave(P, pre, ge_tid());
  ge_sync();
  return pre[(P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK];
}

GE_KERNEL ge_k_tsp_closure(GeParams P, int queue, uint8_t *scratch, uint64_t slot_bytes, int pre_off) {
  uint64_t *done_all = (uint64_t *)ge_dyn_smem();  // [64 lanes][W] settled sets
  int *pre = (int *)(ge_dyn_smem() + pre_off);
  const int count = ge_tsp_count(P, pre, queue);
  const int n = P.n, W = P.W, lane = ge_tid();
  uint64_t *done = done_all + lane * W;
  const ge_buffers &G = P.buf;
  for (int q = ge_bid(); q < count; q += ge_gdim()) {
    const int env = queue ? ge_queue_slot(P, pre, q) : q;
    int32_t *D = (int32_t *)(scratch + (uint64_t)env * slot_bytes);
    const int32_t *rp = G.row_ptr + (int64_t)env * (n + 1);
    const int64_t ebase = (int64_t)env * P.E;
    for (int s = lane; s < n; s += GE_WAVE) {  // array Dijkstra from s: the row D[s][*] is the tentative-distance array
      int32_t *row = D + (int64_t)s * n;
      for (int v = 0; v < n; v++) row[v] = INT32_MAX;
      for (int w = 0; w < W; w++) done[w] = 0ull;
      row[s] = 0;
      for (int it = 0; it < n; it++) {
        int v = -1; int32_t best = INT32_MAX;
        for (int u = 0; u < n; u++) { const int32_t d = row[u]; if (d < best && !((done[u >> 6] >> (u & 63)) & 1ull)) { best = d; v = u; } }
        if (v < 0) break;
        done[v >> 6] |= 1ull << (v & 63);
        if (P.spatial) {  // sw64: float64 lengths in ascending-neighbour order
          int j = rp[v];
          for (int w = 0; w < W; w++)
            for (uint64_t bits = G.adj_bits[((int64_t)env * n + v) * W + w]; bits; bits &= bits - 1, j++) {
              const int u = w * 64 + ge_ctz64(bits); const int32_t d = best + ge_tsp_units(P, G.sw64[ebase + j]);
              if (d < row[u]) row[u] = d;
            }
        } else {
          for (int k = rp[v]; k < rp[v + 1]; k++) {
            const uint16_t cw = G.colw[ebase + k]; const int u = cw >> 4; const int32_t d = best + (int32_t)(cw & 15);
            if (d < row[u]) row[u] = d;
          }
        }
      }
    }
  }
}

GE_KERNEL ge_k_tsp_tour(GeParams P, int queue, uint8_t *scratch, uint64_t slot_bytes) {
  int *pre = (int *)ge_dyn_smem();
  const int count = ge_tsp_count(P, pre, queue);
  const int n = P.n;
  for (int q = ge_bid() * GE_TSP_EVAL_THREADS + ge_tid(); q < count; q += ge_gdim() * GE_TSP_EVAL_THREADS) {
    const int env = queue ? ge_queue_slot(P, pre, q) : q;
    uint8_t *blk = scratch + (uint64_t)env * slot_bytes;
    ge_ch c;
    c.err = 0;
    ge_ch_carve(&c, blk + ge_ch_align((uint64_t)n * (uint64_t)n * 4u), n);
    c.D = (const int32_t *)blk;
    const int64_t tot = ge_christofides_tour(&c);
    if (tot >= 0) P.buf.heuristic[env] = (double)tot / (P.spatial ? 65536.0 : 10.0);  // else: the double-tree walk stays
  }
}

// MaxIndependentSet-v0 baseline (is_eval_env, unweighted): len(nx.approximation.maximum_independent_set(G))
// (max_independent_set.py:63-67), networkx's clique removal reproduced exactly (ge_clique_removal.h: dict orders of the graph
// copies, CPython's set tables).  One LANE per regenerated slot on the slot's scratch block; replaces the min-degree greedy value
// the graph kernel left in heuristic[] (which stays if the work space were ever too small).  Evaluation-time path, like the above.
GE_KERNEL ge_k_mis_baseline(GeParams P, int queue, uint8_t *scratch, uint64_t slot_bytes) {
  int *pre = (int *)ge_dyn_smem();
  const int count = ge_tsp_count(P, pre, queue);
  const int n = P.n;
  const ge_buffers &G = P.buf;
  for (int q = ge_bid() * GE_TSP_EVAL_THREADS + ge_tid(); q < count; q += ge_gdim() * GE_TSP_EVAL_THREADS) {
    const int env = queue ? ge_queue_slot(P, pre, q) : q;
    ge_cr_work w;
    ge_cr_carve(&w, scratch + (uint64_t)env * slot_bytes, n, P.m);
    const int32_t *rp = G.row_ptr + (int64_t)env * (n + 1);
    const int64_t ebase = (int64_t)env * P.E;
    for (int v = 0; v <= n; v++) w.ga.off[v] = rp[v];
    for (int k = 0; k < P.E; k++) w.ga.adj[k] = (uint16_t)(G.colw[ebase + k] >> 4);  // insertion-order columns
    const int32_t r = ge_cr_solve(&w);
    if (r >= 0) G.heuristic[env] = (double)r;
  }
}

// SteinerTree-v0 baseline (is_eval_env, 1 < n_dests < n - 1): the sum of delays over networkx's Kou Steiner tree
// (steiner_tree.py:84-87), reproduced exactly (ge_kou_exact.h: heap-ordered Dijkstra paths, stable Kruskal order over subgraph
// views, CPython's set tables for tuples and ints, float64 sum order).  One LANE per regenerated slot; replaces the own Kou-style
// value the graph kernel left in heuristic[] (which stays if the work space were ever too small).
GE_HOSTDEV uint64_t ge_steiner_slot_bytes(int n, int m, int T) {
  return ((((uint64_t)(2 * m + 1) * 2) + 15) & ~15ull) + ((((uint64_t)(2 * m + 1) * 8) + 15) & ~15ull) + ge_kou_arena_bytes(n, m, T);
}

GE_KERNEL ge_k_steiner_baseline(GeParams P, int queue, uint8_t *scratch, uint64_t slot_bytes) {
  int *pre = (int *)ge_dyn_smem();
  const int count = ge_tsp_count(P, pre, queue);
  const int n = P.n;
  const ge_buffers &G = P.buf;
  for (int q = ge_bid() * GE_TSP_EVAL_THREADS + ge_tid(); q < count; q += ge_gdim() * GE_TSP_EVAL_THREADS) {
    const int env = queue ? ge_queue_slot(P, pre, q) : q;
    uint8_t *blk = scratch + (uint64_t)env * slot_bytes;
    uint16_t *adj = (uint16_t *)blk;
    double *w = (double *)(blk + ((((uint64_t)(2 * P.m + 1) * 2) + 15) & ~15ull));
    const int64_t ebase = (int64_t)env * P.E;
    for (int k = 0; k < P.E; k++) { const uint16_t cw = G.colw[ebase + k]; adj[k] = (uint16_t)(cw >> 4); w[k] = ge_wlut(cw & 15); }
    ge_cr_arena a;
    a.base = (uint8_t *)w + ((((uint64_t)(2 * P.m + 1) * 8) + 15) & ~15ull); a.top = 0; a.peak = 0; a.cap = ge_kou_arena_bytes(n, P.m, P.T); a.err = 0;
    ge_kou_in g;
    g.n = n; g.m = P.m; g.T = P.T; g.off = G.row_ptr + (int64_t)env * (n + 1); g.adj = adj; g.w = w; g.terms = G.terminals + (int64_t)env * P.T;
    int err = 0;
    const double v = ge_kou_exact(&g, &a, &err);
    if (!err) G.heuristic[env] = v;
  }
}
