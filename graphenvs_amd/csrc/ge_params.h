// Engine parameters passed by value to every kernel, and the LDS carve of the reset kernel.
#pragma once
#include <stdint.h>

#include "../../include/graphenvs.h"

#define GE_WAVE 64
#define GE_MT_N 624
#define GE_MT_M 397
#ifndef GE_DC_LANES
#define GE_DC_LANES 16  // DistributionCenter, n <= 64: target ranges searched at a time in the reset kernel (one lane each); sizes the LDS columns
#endif
#ifndef GE_STEP_BLOCK
#define GE_STEP_BLOCK 256
#endif
#define GE_RESET_THREADS 128
#define GE_SEED_TILE_STRIDE 68  // u32 per row of a seeding tile [16 words][64 slots]: 68 mod 64 = 4, the flush reads 64 different banks
#define GE_SEED_LDS_BYTES (2 * 16 * GE_SEED_TILE_STRIDE * 4 + 64 * 4 + 64)
#ifndef GE_RESET_WAVES_PER_SIMD
#define GE_RESET_WAVES_PER_SIMD 6  // 24 waves / CU = 12 two-wave workgroups: one round for the ~2 700 resets of a headline step
#endif

// byte offsets into the dynamic LDS of one reset workgroup (one env resident per workgroup)
struct GeLds {
  int mt;       // u32[624]  MT19937 state of the python stream (wave 0)
  int mt2;      // u32[624]  MT19937 state of the numpy stream (wave 1)
  int wm;       // numpy wave output: nibble matrix u32[n*n/8] of delay codes, or a byte list of m / n codes
  int abits;    // u64[n*W]  adjacency bit rows
  int elist;    // u32[m]    accepted edges in insertion order (u | v << 16); later G.edges order map
  int fill;     // i32[n]    per-row fill counters / degrees
  int rowptr;   // i32[n+1]
  int colw;     // u16[E]    (col << 4) | weight code, insertion order
  int wsort;    // u8[E]     weight codes in ascending-neighbour order (the scode slab of this slot)
  int tmp;      // u32[E]    (edge id << 16 | col) scatter buffer
  int dist;     // i32[n]
  int perm;     // i32[n]
  int f64a;     // f64[2][n] Dijkstra distances (sigma, delta); also stages the first mask words
  int bits;     // u64[6][W] frontier / visited / next / removed-or-targets / prune / removed (first mask)
  int misc;     // i32[16]
  int fw;       // f64[n*n]  PerishableProductDelivery: Floyd-Warshall matrix (else absent)
  int dcs;      // DistributionCenter, n <= 64: f64[n][GE_DC_LANES] distances (one column per source lane) + u8[64][GE_DC_LANES] work stacks
  int kou;      // own Steiner baseline (is_eval_env, 1 < n_dests < n-1): u64[2][n*W] path union / tree, f64[T] keys, i32[2][T]
  int pre;      // i32[nblk+1] exclusive prefix of the per-workgroup reset counts
  int total;
};

// LDS carve of the generic structural-feature kernel
// Two roles share one allocation.  Every workgroup stages rowptr and owns bc / clos.  A Brandes workgroup (a share of the slot's BFS
// sources) stages the rows as QUADS -- four neighbour ids (u16) per 8-byte word, every row padded to whole quads with the id n, the
// "zero node" whose coefficient is always 0.0 -- and a descriptor {first quad, quads} per node, then uses `waves` per-wave areas
// {sigma, delta, coeff (float64, coeff with the zero node's entry), bcw when the partial sums do not fit registers, mark bytes, order
// list, level starts}; the node-level workgroup (clustering, pagerank) uses the adjacency bit rows, the sorted edge copy and its
// float64 arrays INSTEAD -- in the same bytes (`node` == `wave0`) when the two roles are different workgroups (feat_parts > 1),
// behind the per-wave areas when one workgroup does both.
struct GeLdsF {
  int rowptr, colw, bc, clos;          // common (colw: the quads, or -- complete graph on all n nodes -- the weight-code bytes)
  int rq;                              // u32[n] row descriptor: first quad | quads << 16
  int zq;                              // index of the all-padding quad (a lane whose row has ended reads it)
  int wave0, wave_stride, waves;       // per-wave areas of the Brandes role
  int w_sigma, w_mark, w_ord, w_lvl;   // offsets inside a per-wave area (delta, coeff, bcw follow sigma)
  int node, abits, scw, prx, clus;     // node role: abits, scw, then prx, prn, sinv, diff (float64[n] each), clus
  int wl;                              // node role: the 16 weights k / 10.0 by code (weighted pagerank reads one per row entry and iteration)
  int pre, total;
};

struct GeParams {
  int32_t env_type, B, n, m, E, W, F, Fe, A, AW, T, ng, nflag;
  int32_t weighted, parenting, n_dests, is_eval, autoreset, complete;
  int32_t spatial;   // TSP with coordinates and float64 Euclidean weights (sw64 slab)
  int32_t feat_parts;   // n > 64: workgroups sharing one slot's BFS sources in the feature kernel
  int32_t np_early;  // the numpy wave can produce every weight code without the topology (dense delay matrix fits LDS)
  int32_t nocolw;    // TSP on the complete graph (BASELINE config 3), no is_eval baseline: the reset kernel keeps no {neighbour, code} list
                     // in LDS -- the neighbour of directed edge idx is a closed form and its code sits in wsort (ascending order IS
                     // insertion order there): 32 of the 68 KB of a 128-node slot, i.e. four workgroups per CU instead of two
  int32_t nowsort;   // ... and above 64 nodes (no node_rec to fill) no per-entry code list either: the code of edge (a, b), a < b, is
                     // draw number a (n - 1) - a (a - 1) / 2 + (b - a - 1) of the numpy wave's byte list -- 20 KB per slot, eight workgroups
  int32_t cost_off;  // DistributionCenter: byte offset of the node-cost list inside the wm scratch
  uint64_t div_m;    // complete graphs: floor(2^40 / (ng - 1)) + 1, so that idx / (ng - 1) == (idx * div_m) >> 40 for every directed-edge index
  double n_choices;
  double max_distance;  // DistributionCenter coverage radius
  double dt_min, dt_max;  // PerishableProductDelivery delivery-time window
  int64_t env_index_base, seed_stride, node_id_base, edge_row_stride;
  ge_buffers buf;
  GeLds lds;
  GeLdsF ldsf;
  // episode prefetch (ge_attach_spares; all NULL without it).  spare_state[slot] = 1: the slot's spare image holds its NEXT
  // episode; the step kernels then queue a finished slot in swap_list / swap_count (same layout as reset_list / reset_count)
  // instead of the regeneration queue, and a copy kernel moves the image in
  uint8_t *spare_state;
  int32_t *swap_list, *swap_count;
  int32_t bucket;  // multi-class engine: the LDS bucket of this size class (ge_api.hip, GeBucket); 0 in a uniform engine
};

// What a launch of the reset path does.  Decoded ONCE, on the host, from a validated request (ge_api.hip: run_*), so that the
// kernels test named flags instead of comparing a `mode` integer in two dozen places (a value outside the enum used to fall into
// the full-reset branches of a queue launch -- profiles/README.md, "the aperture violation of round 2").
enum { GE_ITEMS_ALL = 0, GE_ITEMS_QUEUE = 1, GE_ITEMS_LIST = 2 };
struct GeRun {
  int32_t items;       // GE_ITEMS_ALL: every slot; GE_ITEMS_QUEUE: P.buf.reset_list / reset_count; GE_ITEMS_LIST: work_list (feature kernels only)
  int32_t restart;     // episode 0 of every item: 1 = seeded seeds[slot] (ge_reset), 2 = seeded inj.seeds[slot] (ge_inject_state with seeds)
  int32_t next;        // the item moves to its next episode: seed + seed_stride, ring entry (episode + 1) % GE_SEED_DEPTH
  int32_t cont;        // the two streams continue from stream_state instead of a pre-seeded ring entry (reset(seed=None))
  int32_t inject;      // topology / weights / terminals / x are the caller's (GeInject), nothing is drawn
  int32_t refill;      // P.buf is the SPARE image of the engine: the slot itself keeps running its episode -- no episode advance,
                       // no final_heur, the slot's step count is not read; the feature kernel marks the image valid instead
  int32_t seed_ahead;  // queue launches: seeding workgroups write the states of episode (e + seed_ahead), seeded seed[] + seed_ahead *
                       // seed_stride, into ring entry (e + seed_ahead) % GE_SEED_DEPTH, e = episode[] of the item; 0 = none
};

// Multi-class ("ragged") engine, BASELINE config 5: slots of different (n, m) stepped by ONE launch sequence.  A size class is a
// uniform sub-engine with its own GeParams (geometry, LDS carves, slab segments); every kernel maps a global slot to (class, local
// slot) and runs the class's code path.  Per-slot scalars, generator states and the reset queue are global arrays in slot order.
struct GeRagged {
  const GeParams *classes;     // [n_classes] device copy of the class table
  const int32_t *slot_class;   // [B_total]
  const int32_t *class_start;  // [n_classes + 1] first global slot of every class
  int32_t n_classes;
};

static inline int ge_align16(int v) { return (v + 15) & ~15; }

// queue_B: slots of the whole engine (the reset queue's prefix array is sized by it; == P.B for a uniform engine)
static inline void ge_make_lds(GeParams &P, int queue_B) {
  GeLds &L = P.lds;
  int o = 0;
  auto take = [&](int bytes) { int r = o; o = ge_align16(o + bytes); return r; };
  L.mt = take(GE_MT_N * 4);
  L.mt2 = take(GE_MT_N * 4);
  { // the dense n x n nibble matrix of delay codes is only drawn by the envs whose weights are delay[u, v] (ge_numpy_wave); TSP,
    // MaxIndependentSet and DensestSubgraph draw a byte list of m / n codes (or nothing)
    const bool matrix_env = P.env_type == GE_SHORTEST_PATH || P.env_type == GE_LONGEST_PATH || P.env_type == GE_STEINER_TREE ||
                            P.env_type == GE_MULTICAST_ROUTING || P.env_type == GE_DISTRIBUTION_CENTER || P.env_type == GE_PERISHABLE_DELIVERY;
    const int matrix = (P.np_early && matrix_env && P.weighted) ? ((P.n * P.n + 7) / 8) * 4 : 0;
    int nb = matrix > 16 ? matrix : 16; if (P.m > nb) nb = P.m; if (P.n > nb) nb = P.n;
    if (P.spatial && 32 * P.n > nb) nb = 32 * P.n;  // raw draws u32[4n] + coordinates f64[2n]
    P.cost_off = 0;
    if (P.env_type == GE_DISTRIBUTION_CENTER) { P.cost_off = matrix; if (P.cost_off + P.n > nb) nb = P.cost_off + P.n; }
    L.wm = take(nb); }
  L.abits = take(P.n * P.W * 8);
  L.elist = take((P.complete ? P.n : (P.m > 0 ? P.m : 1)) * 4);  // complete graphs have no sampled edge list (only the n-entry path stack of the multicast baseline lives here)
  L.fill = take(P.n * 4);
  L.rowptr = take((P.n + 1) * 4);
  L.colw = take((P.nocolw ? 8 : (P.E > 0 ? P.E : 1)) * 2);
  L.wsort = take(P.nowsort ? 16 : (P.E > 0 ? P.E : 1));
  L.tmp = take(P.complete ? 16 : (P.E > 0 ? P.E : 1) * 4);
  L.dist = take(P.n * 4);
  L.perm = take(P.n * 4);
  { // the two float64 node arrays are the Dijkstra / Prim / placement scratch of the is_eval baselines and of the envs that measure
    // distances at reset (MulticastRouting's delay bound, DistributionCenter's ranges, PerishableProductDelivery's windows); everybody
    // else only stages the first mask words there -- 8 KB of a 512-node ShortestPath slot, which is what kept it above half a CU's LDS
    const bool dist_env = P.env_type == GE_MULTICAST_ROUTING || P.env_type == GE_DISTRIBUTION_CENTER || P.env_type == GE_PERISHABLE_DELIVERY;
    int need = (P.is_eval || dist_env) ? 2 * P.n * 8 : 0, mw = ((P.E > P.n ? P.E : P.n) / 64 + 2) * 8; L.f64a = take(need > mw ? need : mw); }
  L.bits = take(6 * P.W * 8);
  L.misc = take(16 * 4);
  L.fw = (P.env_type == GE_PERISHABLE_DELIVERY && P.n <= 128) ? take(P.n * P.n * 8) : 0;  // (larger graphs: distances per pickup, ge_ppd_place_wide)
  L.dcs = (P.env_type == GE_DISTRIBUTION_CENTER && P.n <= 64) ? take(P.n * GE_DC_LANES * 8 + 64 * GE_DC_LANES) : 0;
  L.kou = (P.env_type == GE_STEINER_TREE && P.is_eval && P.n_dests > 1 && P.n_dests < P.n - 1) ? take(2 * P.n * P.W * 8 + P.T * 16) : 0;
  // the queue prefix is only needed while a workgroup looks up its slot: it overlays the scratch that follows
  { int pb = ((queue_B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK + 1) * 4; L.pre = L.mt; if (pb > GE_MT_N * 8) { L.pre = take(pb); } }
  if (o < GE_SEED_LDS_BYTES) o = GE_SEED_LDS_BYTES;  // the seeding workgroups of the queue-mode launch use the same allocation
  L.total = o;
}

// Workgroups that share one slot's BFS sources, given the waves a workgroup runs (GeLdsF.waves): every wave should get four
// sources or more -- a workgroup stages the slot's graph before its first search -- and never more parts than the layout query
// promised (feat_scratch is sized by them); a split geometry stays split (the LDS carve depends on it)
#ifndef GE_FEAT_SRC_PER_WAVE
#define GE_FEAT_SRC_PER_WAVE 4
#endif
static inline void ge_tune_feat_parts(GeParams &P) {
  if (P.feat_parts <= 1) return;
  int want = P.n / (P.ldsf.waves * GE_FEAT_SRC_PER_WAVE);
  if (want < 2) want = 2;
  if (want < P.feat_parts) P.feat_parts = want;
}

// force_waves > 0: waves per workgroup of the generic feature kernel are given (multi-class engine: one launch geometry per bucket)
// budget: LDS bytes the wave count may be sized for (a uniform engine aims at two workgroups per CU; a multi-class engine passes the
// whole CU for the classes of a bucket whose widest class leaves room for one workgroup per CU only)
#ifndef GE_BCW_REG_W
#define GE_BCW_REG_W 8
#endif
// ^ graphs of up to 512 nodes: a wave of the generic feature kernel keeps its betweenness partial sums in registers
static inline void ge_make_ldsf(GeParams &P, int queue_B, int force_waves = 0, int budget = 160 * 1024 / 2) {
  GeLdsF &L = P.ldsf;
  int o = 0;
  auto take = [&](int bytes) { int r = o; o = ge_align16(o + bytes); return r; };
  const int n = P.n, E = P.E > 0 ? P.E : 1;
  const bool split = P.feat_parts > 1;  // the Brandes role and the node role are different workgroups
  L.rowptr = take((n + 1) * 4);
  // (a complete graph on all n nodes keeps one weight-code BYTE per row entry, ascending-neighbour order, instead of a neighbour list:
  // the neighbour is a closed form and there is no search -- half the LDS of BASELINE config 3's slot, twice the workgroups per CU)
  const bool closed = P.complete && P.ng == n;
  const int nquads = (E + 3 * n) / 4 + 1;  // every row padded to whole quads, + the all-padding quad
  L.zq = nquads - 1;
  L.colw = take(closed ? E : nquads * 8);
  L.rq = closed ? L.colw : take(n * 4);
  L.bc = take(n * 8);
  L.clos = take(n * 8);
  const int common = o;
  // per-wave area: sigma, delta, coeff[n + 1] (+ bcw when the partial sums do not fit registers), mark[n + 1], ord, lvl
  L.w_sigma = 0;
  L.w_mark = (P.W <= GE_BCW_REG_W ? 3 : 4) * n * 8 + 8;
  L.w_ord = ge_align16(L.w_mark + n + 1);
  L.w_lvl = L.w_ord + ge_align16(2 * n);
  L.wave_stride = ge_align16(L.w_lvl + 2 * (n + 2));
  // node role
  const int node_bytes = ge_align16(n * P.W * 8) + (closed ? 0 : ge_align16(E * 2)) + 6 * ge_align16(n * 8);
  // the wave count fixes the order of the float64 betweenness partial sums: it is decided from the graph geometry alone (with a
  // nominal 1 KB for the queue prefix), never from the batch size, so that any shard reproduces the unsharded run bit for bit
  const int fixed = common + 1024 + (split ? 0 : node_bytes);
  // the kernel holds 128 vector registers: 16 waves per CU, as two workgroups of at most 8 or one of at most 16
  const int whole = 160 * 1024 - 2048;
  int waves = (budget - fixed) / L.wave_stride;
  if (budget >= whole) { if (waves > 16) waves = 16; }
  else {
    if (waves > 8) waves = 8;
    // where two workgroups of half a CU's LDS each hold fewer waves than ONE workgroup with all of it would (n > 256: a wave's
    // area is 29 n bytes), one workgroup of up to 16 waves runs per CU -- a wave is a BFS source in flight, and at these sizes the
    // pass is a chain of LDS round trips that only more sources in flight hide
    int one = (whole - fixed) / L.wave_stride;
    if (one > 16) one = 16;
    if (one > 2 * waves) waves = one;
  }
  if (waves < 1) waves = 1;
  if (n <= 64) waves = 1;  // only the rare fallback of the n <= 64 fast path lands here
  // complete graph on all nodes (TSP config 3): Brandes is skipped, the workgroup is the pagerank over n rows of n-1 entries --
  // one thread per pagerank row: more waves would only hold LDS
  if (P.complete && P.ng == n && n > 64) { waves = (n + 63) / 64; if (waves > 8) waves = 8; }
  if (force_waves > 0) waves = force_waves;
  const int o_common = o;
  for (;;) {
    o = o_common;
    L.waves = waves;
    L.wave0 = o;
    const int waves_end = o + waves * L.wave_stride;
    L.node = split ? L.wave0 : waves_end;
    o = L.node;
    L.abits = take(n * P.W * 8);
    L.scw = closed ? L.colw : take(E * 2);  // rows in ascending-neighbour order ({neighbour, code}; a closed graph keeps the code bytes only)
    L.prx = take(5 * ge_align16(n * 8));          // prx, prn, sinv, diff, diff2
    L.clus = take(n * 8);
    L.wl = take(16 * 8);
    if (o < waves_end) o = waves_end;
    const int pre_bytes = ((queue_B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK + 1) * 4;
    L.pre = take(pre_bytes);
    L.total = o;
    // (an engine of so many slots that its queue prefix outgrows the nominal 1 KB: give up waves rather than fail; a multi-class
    // launch keeps a second prefix behind the widest class's carve)
    if (L.total + pre_bytes + 64 <= 160 * 1024 || waves == 1 || force_waves > 0) break;
    waves--;
  }
}
