// Reset kernel: one wave64 workgroup regenerates one env slot entirely on the device, with the
// slot's graph resident in LDS.  Replaces, per slot, the body of reset() in the six reference envs
// (shortest_path.py:47-98, longest_path.py:53-118, steiner_tree.py:50-113, tsp.py:50-164,
// densest_subgraph.py:52-98, max_independent_set.py:41-89) including the third-party arithmetic
// they call: CPython random / numpy legacy MT19937 streams, networkx gnm_random_graph,
// is_connected, the five structural features of feature_extraction.py:6-37, Dijkstra / MST totals.
#pragma once
#include "ge_params.h"
#include "ge_platform.h"

// weight code k -> k/10.0 (codes 3..9 = randint(3,10)/10.0, 10 = 1.0 for unweighted graphs; any other code reads as 1.0)
// One correctly rounded float64 division -- 3 / 10.0 IS the double the literal 0.3 names, and so on -- instead of an eight-way
// switch: in a loop whose lanes hold different codes the switch is eight masked branches per call.
GE_DEV double ge_wlut(int code) {
  const int k = (code >= 3 && code <= 9) ? code : 10;
  return (double)k / 10.0;
}

GE_DEV uint32_t ge_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

// Regenerate all 624 words, 64 lanes wide.  Word i needs the OLD words i, i+1 and, for i < 227, the old word i+397;
// for i >= 227 the NEW word i-227.  So the state falls into three spans [0,227) [227,454) [454,624) whose words are
// independent of each other: per span every lane reads its (up to four) operands, the wave synchronises once, and
// writes -- six ordering points per twist instead of twenty.
GE_DEV void ge_mt_twist(uint32_t *mt, int lane) {
  const int lo[3] = {0, GE_MT_N - GE_MT_M, 2 * (GE_MT_N - GE_MT_M)}, hi[3] = {GE_MT_N - GE_MT_M, 2 * (GE_MT_N - GE_MT_M), GE_MT_N};
#pragma unroll
  for (int ph = 0; ph < 3; ph++) {
    uint32_t a[4], b[4], c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = lo[ph] + lane + 64 * k;
      a[k] = b[k] = c[k] = 0u;
      if (i < hi[ph]) {
        int i1 = i + 1; if (i1 == GE_MT_N) i1 = 0;
        int im = i + GE_MT_M; if (im >= GE_MT_N) im -= GE_MT_N;
        a[k] = mt[i]; b[k] = mt[i1]; c[k] = mt[im];
      }
    }
    ge_wave_sync();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = lo[ph] + lane + 64 * k;
      if (i < hi[ph]) {
        const uint32_t y = (a[k] & 0x80000000u) | (b[k] & 0x7fffffffu);
        mt[i] = c[k] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
    }
    ge_wave_sync();
  }
}

// The same regeneration with the state in REGISTERS: word 64 r + lane in R[r] (row 9: lanes 0-47).  The long draws (an n x n delay
// matrix is 1.14 n^2 raw words: 480 blocks at n = 512) spent most of a block in the six LDS round trips of ge_mt_twist and the reads
// of the scan behind it; here a block is ~12 register operations per row and the neighbours come through the lane crossbar
// (ds_bpermute: no LDS memory, no ordering points).  Rows are regenerated in order, so a row reads the rows behind it already NEW
// and the rows ahead still OLD -- exactly the in-place algorithm:
//   word i+1   : lane + 1 of the row (lane 63: lane 0 of the next row; word 623: the NEW word 0);
//   word i+397 : for i < 227 the old rows r + 6 / r + 7 rotated by 13 lanes; for i >= 227 the new word i - 227 = rows r - 4 / r - 3
//                rotated by 29 lanes (row 3 straddles word 227 and takes both).
// Each source lane picks the row its reader wants, so a row costs two crossbar reads.
GE_DEV uint32_t ge_mt_mix(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
#define GE_MT_ROWS 10
GE_DEV void ge_mt_twist_regs(uint32_t (&R)[GE_MT_ROWS], int lane) {
  const int l1 = (lane + 1) & 63, l13 = (lane + 13) & 63, l29 = (lane + 29) & 63;
  // Four batches of rows -- {0, 1, 2} {3, 4, 5} {6, 7, 8} {9} -- whose members do not depend on one another (a row needs the rows
  // four and three behind it NEW): the crossbar reads of a batch are issued together, one round trip per batch instead of one per row
  uint32_t B[3], C[3];
#pragma unroll
  for (int r = 0; r < 3; r++) { B[r] = ge_shfl_u32(lane == 0 ? R[r + 1] : R[r], l1); C[r] = ge_shfl_u32(lane >= 13 ? R[r + 6] : R[r + 7], l13); }
#pragma unroll
  for (int r = 0; r < 3; r++) R[r] = ge_mt_mix(R[r], B[r], C[r]);
  {
    const uint32_t c1 = ge_shfl_u32(R[9], l13), c2 = ge_shfl_u32(R[0], l29);
#pragma unroll
    for (int r = 3; r < 6; r++) B[r - 3] = ge_shfl_u32(lane == 0 ? R[r + 1] : R[r], l1);
#pragma unroll
    for (int r = 4; r < 6; r++) C[r - 3] = ge_shfl_u32(lane >= 29 ? R[r - 4] : R[r - 3], l29);
    C[0] = lane < 35 ? c1 : c2;
#pragma unroll
    for (int r = 3; r < 6; r++) R[r] = ge_mt_mix(R[r], B[r - 3], C[r - 3]);
  }
#pragma unroll
  for (int r = 6; r < 9; r++) { B[r - 6] = ge_shfl_u32(lane == 0 ? R[r + 1] : R[r], l1); C[r - 6] = ge_shfl_u32(lane >= 29 ? R[r - 4] : R[r - 3], l29); }
#pragma unroll
  for (int r = 6; r < 9; r++) R[r] = ge_mt_mix(R[r], B[r - 6], C[r - 6]);
  {
    const uint32_t b9 = ge_shfl_u32(lane == 0 ? R[0] : R[9], lane == 47 ? 0 : l1), c9 = ge_shfl_u32(lane >= 29 ? R[5] : R[6], l29);
    R[9] = ge_mt_mix(R[9], b9, c9);
  }
}
GE_DEV void ge_mt_to_regs(uint32_t (&R)[GE_MT_ROWS], const uint32_t *mt, int lane) {
#pragma unroll
  for (int r = 0; r < GE_MT_ROWS; r++) R[r] = mt[(GE_WAVE * r + lane < GE_MT_N) ? GE_WAVE * r + lane : 0];
}
GE_DEV void ge_mt_from_regs(uint32_t *mt, const uint32_t (&R)[GE_MT_ROWS], int lane) {
#pragma unroll
  for (int r = 0; r < GE_MT_ROWS; r++) if (GE_WAVE * r + lane < GE_MT_N) mt[GE_WAVE * r + lane] = R[r];
}

// random.seed(int) for 0 <= s < 2^32: init_by_array([s]) ([py] _randommodule.c).  1 246 dependent steps, run by
// lane 0 on the vector ALU (measured: the scalar-ALU form of the same chain is 1.6x slower on gfx950).  Collective.
GE_DEV void ge_mt_seed_python(uint32_t *mt, uint32_t seed, int lane) {
  if (lane == 0) {
    uint32_t b = 19650218u, prev = b;
    mt[0] = b;
    for (int i = 1; i < GE_MT_N; i++) {
      b = 1812433253u * (b ^ (b >> 30)) + (uint32_t)i;            // init_genrand(19650218)[i]
      prev = (b ^ ((prev ^ (prev >> 30)) * 1664525u)) + seed;      // + key[0] + j, j == 0
      mt[i] = prev;
    }
    mt[0] = prev;  // i wrapped: mt[0] = mt[N-1]
    prev = (mt[1] ^ ((prev ^ (prev >> 30)) * 1664525u)) + seed;    // 624th iteration at i = 1
    mt[1] = prev;
    for (int i = 2; i < GE_MT_N; i++) {
      prev = (mt[i] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
      mt[i] = prev;
    }
    mt[0] = prev;
    mt[1] = (mt[1] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - 1u;
    mt[0] = 0x80000000u;
  }
  ge_wave_sync();
}

// np.random.seed(int): init_genrand(s) ([np] mt19937.c).  Collective.
GE_DEV void ge_mt_seed_numpy(uint32_t *mt, uint32_t seed, int lane) {
  if (lane == 0) {
    uint32_t prev = seed;
    mt[0] = prev;
    for (int i = 1; i < GE_MT_N; i++) { prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i; mt[i] = prev; }
  }
  ge_wave_sync();
}


// copy one pre-seeded MT19937 state (624 words, contiguous in HBM) into LDS: all ten loads of a lane are issued before
// the first store, so the wave pays one memory round trip instead of ten
GE_DEV void ge_mt_load(uint32_t *mt, const uint32_t *src, int lane) {
  uint32_t r[10];
#pragma unroll
  for (int k = 0; k < 10; k++) { int i = lane + 64 * k; r[k] = i < GE_MT_N ? src[i] : 0u; }
#pragma unroll
  for (int k = 0; k < 10; k++) { int i = lane + 64 * k; if (i < GE_MT_N) mt[i] = r[k]; }
  ge_wave_sync();
}

// the reverse: a stream as a regeneration left it (state words, then the read position) to ge_buffers.stream_state
GE_DEV void ge_mt_save(uint32_t *dst, const uint32_t *mt, int pos, int lane) {
  ge_wave_sync();
#pragma unroll
  for (int k = 0; k < 10; k++) { int i = lane + 64 * k; if (i < GE_MT_N) dst[i] = mt[i]; }
  if (lane == 0) dst[GE_MT_N] = (uint32_t)pos;
}

GE_DEV int ge_wave_incl_scan(int x, int lane) {
  for (int off = 1; off < GE_WAVE; off <<= 1) {
    int t = ge_shfl_i32(x, lane >= off ? lane - off : 0);
    if (lane >= off) x += t;
  }
  return x;
}

GE_DEV uint32_t ge_mask_below(uint32_t v) {  // smallest 2^k - 1 >= v
  return v ? (0xffffffffu >> ge_clz32(v)) : 0u;
}

struct GeRctx {
  uint32_t *mt, *mt2, *wm; uint64_t *abits; uint32_t *elist; int *fill; int *rowptr; uint16_t *colw; uint8_t *wsort;
  uint32_t *tmp; int *dist; int *perm;
  double *sigma, *delta;
  uint64_t *bits; int *misc;
};

GE_DEV GeRctx ge_carve(const GeParams &P) {
  unsigned char *s = ge_dyn_smem();
  const GeLds &L = P.lds;
  GeRctx c;
  c.mt = (uint32_t *)(s + L.mt); c.mt2 = (uint32_t *)(s + L.mt2); c.wm = (uint32_t *)(s + L.wm); c.abits = (uint64_t *)(s + L.abits); c.elist = (uint32_t *)(s + L.elist);
  c.fill = (int *)(s + L.fill); c.rowptr = (int *)(s + L.rowptr); c.colw = (uint16_t *)(s + L.colw);
  c.wsort = (uint8_t *)(s + L.wsort); c.tmp = (uint32_t *)(s + L.tmp); c.dist = (int *)(s + L.dist);
  c.perm = (int *)(s + L.perm);
  double *f = (double *)(s + L.f64a);
  c.sigma = f; c.delta = f + P.n;
  c.bits = (uint64_t *)(s + L.bits); c.misc = (int *)(s + L.misc);
  return c;
}

// Level-synchronous BFS on the LDS bit rows from `start` over the nodes [0,ng) that are not in `removed`
// (W words in LDS, may be null).  Leaves the visited set (removed nodes included) in c.bits[W..2W) and returns
// the number of nodes reached.  Collective over the wave.
GE_DEV int ge_reach_wave(const GeRctx &c, int ng, int W, int start, const uint64_t *removed, int lane) {
  uint64_t *fr = c.bits, *vis = c.bits + W, *nx = c.bits + 2 * W;
  if (lane < W) {
    uint64_t s = ((start >> 6) == lane) ? (1ull << (start & 63)) : 0ull;
    fr[lane] = s; vis[lane] = s | (removed ? removed[lane] : 0ull);
  }
  ge_wave_sync();
  for (;;) {
    uint64_t any = 0;
    for (int k = 0; k < W; k++) {
      int v = k * GE_WAVE + lane;
      bool hit = false;
      if (v < ng && !((vis[k] >> lane) & 1ull)) {
        for (int w = 0; w < W; w++) if (c.abits[v * W + w] & fr[w]) { hit = true; break; }
      }
      uint64_t b = ge_ballot(hit);
      if (lane == 0) nx[k] = b;
      any |= b;
    }
    ge_wave_sync();
    if (!any) break;
    if (lane < W) { vis[lane] |= nx[lane]; fr[lane] = nx[lane]; }
    ge_wave_sync();
  }
  int cnt = 0;
  for (int w = 0; w < W; w++) cnt += ge_popc64(vis[w] & ~(removed ? removed[w] : 0ull));
  ge_wave_sync();
  return cnt;
}

// [nx] is_connected over nodes [0,ng) minus `skip`
GE_DEV bool ge_connected(const GeRctx &c, int ng, int W, int skip, int lane) {
  uint64_t *rem = c.bits + 3 * W;
  if (lane < W) rem[lane] = (skip >= 0 && (skip >> 6) == lane) ? (1ull << (skip & 63)) : 0ull;
  ge_wave_sync();
  int start = (skip == 0) ? 1 : 0;
  return ge_reach_wave(c, ng, W, start, rem, lane) == ng - (skip >= 0 ? 1 : 0);
}

GE_DEV double ge_pw_leaf(const double *a, int n, int lane) {  // numpy pairwise sum, n <= 128
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; i++) res += a[i];
    return res;
  }
  int lim = n - (n % 8);
  double r = 0.0;
  if (lane < 8) { r = a[lane]; for (int i = 8; i < lim; i += 8) r += a[i + lane]; }
  double r0 = ge_shfl_f64(r, 0), r1 = ge_shfl_f64(r, 1), r2 = ge_shfl_f64(r, 2), r3 = ge_shfl_f64(r, 3);
  double r4 = ge_shfl_f64(r, 4), r5 = ge_shfl_f64(r, 5), r6 = ge_shfl_f64(r, 6), r7 = ge_shfl_f64(r, 7);
  double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (int i = lim; i < n; i++) res += a[i];
  return res;
}
template <int D>
GE_DEV double ge_pw(const double *a, int n, int lane) {
  if (n <= 128) return ge_pw_leaf(a, n, lane);
  if constexpr (D > 0) {
    int n2 = n / 2; n2 -= n2 % 8;
    double l = ge_pw<D - 1>(a, n2, lane);
    double r = ge_pw<D - 1>(a + n2, n - n2, lane);
    return l + r;
  } else {
    return 0.0;
  }
}

// number of neighbours of a bit row that are smaller than v
GE_DEV int ge_rank_below(const uint64_t *row, int v) {
  int r = 0;
  for (int w = 0; w < (v >> 6); w++) r += ge_popc64(row[w]);
  return r + ge_popc64(row[v >> 6] & ((1ull << (v & 63)) - 1ull));
}

// neighbour / weight code of directed edge idx (insertion order) -- from the LDS list, or (GeParams.nocolw: complete graph, every row
// holds all other nodes in ascending order) the closed form and the code list in ascending-neighbour order
GE_DEV int ge_list_nbr(const GeParams &P, const GeRctx &c, int idx) {
  if (P.nocolw) { const int u = (int)(((uint64_t)(uint32_t)idx * P.div_m) >> 40), k = idx - u * (P.ng - 1); return k < u ? k : k + 1; }
  return (int)(c.colw[idx] >> 4);
}
GE_DEV int ge_list_code(const GeParams &P, const GeRctx &c, int idx) {
  if (P.nowsort) {  // straight from the numpy wave's byte list: draw number of the undirected edge (complete_graph = combinations(nodes, 2))
    if (!P.weighted) return 10;
    const int u = (int)(((uint64_t)(uint32_t)idx * P.div_m) >> 40), k = idx - u * (P.ng - 1), v = k < u ? k : k + 1;
    const int a = u < v ? u : v, b = u < v ? v : u;
    return (int)((const uint8_t *)c.wm)[a * (P.ng - 1) - ((a * (a - 1)) >> 1) + (b - a - 1)];
  }
  return P.nocolw ? (int)c.wsort[idx] : (int)(c.colw[idx] & 15);
}

// source node of directed edge idx
GE_DEV int ge_row_of(const GeParams &P, const GeRctx &c, int idx) {
  return P.complete ? (int)(((uint64_t)(uint32_t)idx * P.div_m) >> 40) : (int)c.tmp[idx];  // complete: every row has ng - 1 entries
}

// slot of the directed entry u->v in ascending-neighbour order (no row scan: rank inside the bit row)
GE_DEV int ge_sorted_pos(const GeRctx &c, int W, int u, int v) { return c.rowptr[u] + ge_rank_below(c.abits + u * W, v); }
// the same, with the closed form of a complete graph (every node below v except u itself is a neighbour)
GE_DEV int ge_sorted_pos_p(const GeParams &P, const GeRctx &c, int u, int v) {
  return P.complete ? c.rowptr[u] + v - (v > u ? 1 : 0) : ge_sorted_pos(c, P.W, u, v);
}


struct GeInject { const int64_t *links; const uint8_t *wcode; const float *x; const int32_t *terminals; const uint32_t *seeds; };
#ifndef GE_GNM_ROUND_CAP
#define GE_GNM_ROUND_CAP (1 << 22)
#endif


// Diagnostic build only (-DGE_STAMPS, never shipped): 100 MHz timestamps of slot 0's reset phases.
#if defined(GE_STAMPS) && !defined(GE_EMU)
__device__ unsigned long long ge_stamp_buf[32];
#define GE_STAMP(k) do { if (lane == 0 && env == 0) { ge_stamp_buf[k] = wall_clock64(); if ((k) == 11) ge_stamp_buf[30] = clock64(); if ((k) == 17) ge_stamp_buf[31] = clock64(); } } while (0)  // 30 / 31: the shader clock (s_memtime) at the ends of the n <= 64 feature kernel's slot 0
#define GE_STAMP_T0(k) do { if (tid == 0 && env == 0) ge_stamp_buf[k] = wall_clock64(); } while (0)
#define GE_STAMP_B0(k) do { if (lane == 0 && ge_bid() == 0) ge_stamp_buf[k] = wall_clock64(); } while (0)  // (workgroup 0 = slot 0 of a full reset)
#elif defined(GE_STAMP_SLOTS) && !defined(GE_EMU)
// Diagnostic build only (-DGE_STAMP_SLOTS, never shipped; tools/slot_times.py): when every slot's regeneration starts (stamp 0), when
// its graph is accepted (stamp 2) and when it is written (stamp 10), 100 MHz, into final_cost / final_len / final_heur of the slot --
// the outputs of the step are garbage in such a build
#define GE_STAMP(k) do { if (lane == 0) { if ((k) == 0) P.buf.final_cost[env] = (double)wall_clock64(); else if ((k) == 2) P.buf.final_len[env] = (int32_t)(wall_clock64() - (unsigned long long)P.buf.final_cost[env]); else if ((k) == 10) P.buf.final_heur[env] = (double)wall_clock64(); } } while (0)
#define GE_STAMP_T0(k) do { } while (0)
#define GE_STAMP_B0(k) do { } while (0)
#else
#define GE_STAMP(k) do { } while (0)
#define GE_STAMP_T0(k) do { } while (0)
#define GE_STAMP_B0(k) do { } while (0)
#endif


// masked-rejection draws of randint(3, 10) ([np] buffered_bounded_masked_uint32), 64 per round: draw k of the
// accepted sequence gets code 3 + value.  sink 0: nibble matrix wm[k] (k = i*n + j of the delay matrix);
// sink 1: byte list wm[k]; sink 2: delay[i, j] lands by rank in wsort (needs the topology); sink 3: randint(1, 4) node costs
// (mask 3, reject > 2; distribution_center.py:82) as a byte list at wm + cost_off.  One wave.
// where accepted draw number idx (value val) of a ge_np_draws sequence lands
GE_DEV void ge_np_sink(const GeParams &P, const GeRctx &c, int sink, int idx, uint32_t val) {
  const int n = P.n, W = P.W;
  const uint32_t code = (sink == 3 ? 1u : 3u) + val;
  if (sink == 3) ((uint8_t *)c.wm)[P.cost_off + idx] = (uint8_t)code;
  else if (sink == 0) atomicOr(&c.wm[idx >> 3], code << (4 * (idx & 7)));
  else if (sink == 1) ((uint8_t *)c.wm)[idx] = (uint8_t)code;
  else {
    const int i = (int)((unsigned)idx / (unsigned)n), j = idx - i * n;
    if (i < j && ((c.abits[i * W + (j >> 6)] >> (j & 63)) & 1ull)) {
      c.wsort[ge_sorted_pos(c, W, i, j)] = (uint8_t)code; c.wsort[ge_sorted_pos(c, W, j, i)] = (uint8_t)code;
    }
  }
}

GE_DEV void ge_np_draws(const GeParams &P, const GeRctx &c, uint32_t *mt, int &nppos, int total, int lane, int sink) {
  int base = 0;
  const uint32_t bm = (sink == 3) ? 3u : 7u;
  const uint64_t below = (1ull << lane) - 1ull;
  while (base < total) {
    if (nppos >= GE_MT_N && total - base > GE_MT_N) {
      // whole blocks of 624 raw words that cannot end the sequence (a block accepts at most 624): generated and scanned in registers
      uint32_t R[GE_MT_ROWS];
      ge_mt_to_regs(R, mt, lane);
      do {
        ge_mt_twist_regs(R, lane);
#pragma unroll
        for (int k = 0; k < GE_MT_ROWS; k++) {  // (a row at a time: nothing but the state stays live across the rows)
          const uint32_t val = (GE_WAVE * k + lane < GE_MT_N) ? (ge_temper(R[k]) & bm) : bm;
          const uint64_t bal = ge_ballot(val < bm);
          if (val < bm) ge_np_sink(P, c, sink, base + ge_popc64(bal & below), val);
          base += ge_popc64(bal);
        }
      } while (total - base > GE_MT_N);
      ge_mt_from_regs(mt, R, lane);
      ge_wave_sync();
      nppos = GE_MT_N;  // (the block in LDS is used up)
      continue;
    }
    if (nppos >= GE_MT_N) { ge_mt_twist(mt, lane); nppos = 0; }
    // four chunks of 64 raw words at a time while they lie inside the current block and cannot finish the sequence: the four
    // state reads, tempering chains and ballots are independent, which is what a single wave needs to stay off the LDS latency
    // (the n x n delay matrix of a 256-node graph is 65 536 accepted draws: 290 us one chunk at a time)
    if (GE_MT_N - nppos >= 4 * GE_WAVE) {
      uint32_t val[4]; uint64_t bal[4];
#pragma unroll
      for (int k = 0; k < 4; k++) val[k] = ge_temper(mt[nppos + GE_WAVE * k + lane]) & bm;
      int tot = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) { bal[k] = ge_ballot(val[k] < bm); tot += ge_popc64(bal[k]); }
      if (base + tot < total) {  // wave-uniform
        int off = base;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (val[k] < bm) ge_np_sink(P, c, sink, off + ge_popc64(bal[k] & below), val[k]);
          off += ge_popc64(bal[k]);
        }
        base = off; nppos += 4 * GE_WAVE;
        continue;
      }
    }
    {
      const int p = nppos + lane; const bool valid = p < GE_MT_N;
      const uint32_t val = valid ? (ge_temper(mt[p]) & bm) : 8u;
      const bool acc = valid && val < bm;
      const uint64_t bal = ge_ballot(acc);
      const int rank = ge_popc64(bal & below);
      const int idx = base + rank;
      if (acc && idx < total) ge_np_sink(P, c, sink, idx, val);
      const int nacc = ge_popc64(bal);
      if (base + nacc >= total) {  // the stream stops right after the last needed accepted draw
        const int need = total - base - 1;
        const uint64_t lastb = ge_ballot(acc && rank == need);
        nppos += ge_ctz64(lastb) + 1;
        base = total;
      } else {
        base += nacc;
        nppos += (GE_MT_N - nppos < GE_WAVE) ? (GE_MT_N - nppos) : GE_WAVE;
      }
    }
  }
  ge_wave_sync();  // rounds only read the state and add into their sink: one ordering point at the end
}

// randint(3, 10, size=(n, n)) when the n x n matrix does not fit LDS (n > 256): only the cells delay[u, v], u < v, of the m edges are
// ever read (shortest_path.py:60,66-67), but the stream has to be consumed to its end.  So the scan only COUNTS: per 64 raw words
// one tempering, one ballot, one popcount -- and the accepted draw whose number is the next wanted cell (cells in ascending order
// in `tcell`, built from the adjacency bit rows) is fished out of its chunk when the running count passes it: about one chunk in
// three at n = 512.  (The draw-by-draw form tested every accepted draw against the adjacency matrix: a division and an LDS read per
// draw, 262 144 draws per slot at n = 512.)  Needs the topology: c.abits, c.rowptr; uses c.elist (free once the CSR is built) for the
// wanted cells and c.dist for the per-row offsets.  One wave.
// wanted cells of the n x n draw, ascending, into tcell[m]: edge (u, v), u < v, is number up[u] + |{w in N(u): u < w < v}| where
// up[u] (left in c.dist[u]) counts the edges of smaller rows.  Needs the adjacency bit rows only.
GE_DEV void ge_np_edge_cells(const GeParams &P, const GeRctx &c, uint32_t *tcell, int lane) {
  const int n = P.n, W = P.W;
  {
    int carry = 0;
    for (int k0 = 0; k0 < n; k0 += GE_WAVE) {
      const int u = k0 + lane; int d = 0;
      if (u < n) for (int w = u >> 6; w < W; w++) { uint64_t b = c.abits[u * W + w]; if (w == (u >> 6)) b &= ~((2ull << (u & 63)) - 1ull); d += ge_popc64(b); }
      const int incl = ge_wave_incl_scan(d, lane);
      if (u < n) c.dist[u] = carry + incl - d;
      carry += ge_shfl_i32(incl, GE_WAVE - 1);
    }
    ge_wave_sync();
    for (int u = lane; u < n; u += GE_WAVE) {
      int t = c.dist[u];
      for (int w = u >> 6; w < W; w++) {
        uint64_t b = c.abits[u * W + w]; if (w == (u >> 6)) b &= ~((2ull << (u & 63)) - 1ull);
        for (; b; b &= b - 1) tcell[t++] = (uint32_t)(u * n + w * 64 + ge_ctz64(b));
      }
    }
    ge_wave_sync();
  }
}
// number of the cell of edge {u, v} in that order (ge_np_edge_cells has run: c.dist = up[])
GE_DEV int ge_np_edge_number(const GeRctx &c, int W, int u, int v) {
  const int a = u < v ? u : v, b = u < v ? v : u;
  int t = c.dist[a];
  for (int w = a >> 6; w <= (b >> 6); w++) {
    uint64_t bits = c.abits[a * W + w];
    if (w == (a >> 6)) bits &= ~((2ull << (a & 63)) - 1ull);
    if (w == (b >> 6)) bits &= (1ull << (b & 63)) - 1ull;
    t += ge_popc64(bits);
  }
  return t;
}
// the scan: n * n masked draws of randint(3, 10), of which only the ones that land on a wanted cell are kept -- tcode[t] = the code
// of wanted cell number t
GE_DEV void ge_np_draws_cells(const GeParams &P, const GeRctx &c, uint32_t *mt, int &nppos, int lane, const uint32_t *tcell, uint8_t *tcode) {
  const int n = P.n, m = P.m, total = n * n;
  const uint64_t below = (1ull << lane) - 1ull;
  // The hot loop touches no graph structure: the wanted cells are held 64 at a time in a register (lane j: cell number t0 + j, read
  // with v_readlane), and the accepted draw that lands on one is stored by its lane as tcode[t]; the codes go to their two slots
  // of the ascending-neighbour order afterwards, one lane per edge.
  GE_STAMP_B0(7);
  int base = 0, t = 0, t0 = 0;
  uint32_t treg = lane < m ? tcell[lane] : 0xffffffffu;
  uint32_t next = ge_readlane_u32(treg, 0);
  auto take = [&](uint32_t val, uint64_t bal, int cnt) {
    while (next - (uint32_t)base < (uint32_t)cnt) {  // (wave-uniform)
      const int r = (int)(next - (uint32_t)base);
      if (((bal >> lane) & 1ull) && ge_popc64(bal & below) == r) tcode[t] = (uint8_t)(3u + val);
      t++;
      if (t - t0 == GE_WAVE) { t0 = t; treg = (t0 + lane < m) ? tcell[t0 + lane] : 0xffffffffu; }
      next = ge_readlane_u32(treg, t - t0);
    }
  };
  while (base < total) {
    if (nppos >= GE_MT_N && total - base > GE_MT_N) {
      // whole blocks of 624 raw words that cannot end the sequence (a block accepts at most 624): generated and scanned in registers,
      // ten chunks per trip
      uint32_t R[GE_MT_ROWS];
      ge_mt_to_regs(R, mt, lane);
      do {
        ge_mt_twist_regs(R, lane);
        // count first (ballots and popcounts: scalar registers); the block is looked at again only when a wanted cell lies in it
        // (about one block in three at n = 512)
        int tot = 0;
#pragma unroll
        for (int k = 0; k < GE_MT_ROWS; k++) tot += ge_popc64(ge_ballot(GE_WAVE * k + lane < GE_MT_N && (ge_temper(R[k]) & 7u) < 7u));
        if (next - (uint32_t)base < (uint32_t)tot) {
#pragma unroll
          for (int k = 0; k < GE_MT_ROWS; k++) {
            const uint32_t val = (GE_WAVE * k + lane < GE_MT_N) ? (ge_temper(R[k]) & 7u) : 7u;
            const uint64_t bal = ge_ballot(val < 7u); const int cnt = ge_popc64(bal);
            take(val, bal, cnt); base += cnt;
          }
        } else base += tot;
      } while (total - base > GE_MT_N);
      ge_mt_from_regs(mt, R, lane);
      ge_wave_sync();
      nppos = GE_MT_N;  // (the block in LDS is used up)
      continue;
    }
    if (nppos >= GE_MT_N) { ge_mt_twist(mt, lane); nppos = 0; }
    if (GE_MT_N - nppos >= 4 * GE_WAVE) {
      uint32_t val[4]; uint64_t bal[4]; int cnt[4];
#pragma unroll
      for (int k = 0; k < 4; k++) val[k] = ge_temper(mt[nppos + GE_WAVE * k + lane]) & 7u;
      int tot = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) { bal[k] = ge_ballot(val[k] < 7u); cnt[k] = ge_popc64(bal[k]); tot += cnt[k]; }
      if (base + tot < total) {  // wave-uniform: the four chunks cannot end the sequence
        if (next - (uint32_t)base < (uint32_t)tot) {
#pragma unroll
          for (int k = 0; k < 4; k++) { take(val[k], bal[k], cnt[k]); base += cnt[k]; }
        } else base += tot;
        nppos += 4 * GE_WAVE;
        continue;
      }
    }
    {
      const int p = nppos + lane; const bool valid = p < GE_MT_N;
      const uint32_t val = valid ? (ge_temper(mt[p]) & 7u) : 8u;
      const bool acc = valid && val < 7u;
      uint64_t bal = ge_ballot(acc);
      int nacc = ge_popc64(bal);
      if (base + nacc >= total) {  // the stream stops right after the last needed accepted draw
        const int need = total - base - 1;
        const int last = ge_ctz64(ge_ballot(acc && ge_popc64(bal & below) == need));
        bal &= (last >= 63) ? ~0ull : ((2ull << last) - 1ull); nacc = need + 1;
        take(val, bal, nacc);
        nppos += last + 1; base = total;
      } else {
        take(val, bal, nacc);
        base += nacc;
        nppos += (GE_MT_N - nppos < GE_WAVE) ? (GE_MT_N - nppos) : GE_WAVE;
      }
    }
  }
  ge_wave_sync();
  GE_STAMP_B0(8);
}
// delay[u, v] of edge number e to both directions of the ascending-neighbour order (needs the CSR row starts)
GE_DEV void ge_np_codes_to_rows(const GeParams &P, const GeRctx &c, const uint32_t *tcell, const uint8_t *tcode, int lane) {
  const int n = P.n, W = P.W, m = P.m;
  for (int e = lane; e < m; e += GE_WAVE) {
    const uint32_t cell = tcell[e];
    const int u = (int)(cell / (uint32_t)n), v = (int)(cell - (uint32_t)u * (uint32_t)n);
    const uint8_t code = tcode[e];
    c.wsort[ge_sorted_pos(c, W, u, v)] = code; c.wsort[ge_sorted_pos(c, W, v, u)] = code;
  }
  ge_wave_sync();
}
// graphs too large for the dense matrix, after the topology and its CSR exist: cells, scan, codes to their rows
GE_DEV void ge_np_draws_edges(const GeParams &P, const GeRctx &c, uint32_t *mt, int &nppos, int lane) {
  uint32_t *tcell = c.elist;
  uint8_t *tcode = (uint8_t *)c.wm;  // the code of wanted cell number t (the numpy wave drew nothing into wm: no matrix at this size)
  ge_np_edge_cells(P, c, tcell, lane);
  ge_np_draws_cells(P, c, mt, nppos, lane, tcell, tcode);
  ge_np_codes_to_rows(P, c, tcell, tcode, lane);
}

// first `need` entries of legacy numpy permutation(pn) into c.perm (np.random.choice(pn, k, replace=False) is its first k entries)
GE_DEV void ge_np_terminals(const GeParams &P, const GeRctx &c, uint32_t *mt, int &nppos, int lane, int pn, int need) {
  // Fisher-Yates from the top: for i = pn-1 .. 1: j_i = interval(i) (masked rejection), swap(a[i], a[j_i]).  Only the first
  // `need` entries of the result are used, and the content of a final position can be traced BACKWARDS through the swaps
  // (last swap first): pos = p; for i = 1 .. pn-1: pos == i ? j_i : (pos == j_i ? i : pos) ends at the element that lands
  // in p.  So the only serial part is the acceptance scan that finds the j_i (a few scalar operations per raw draw); the
  // trace runs one lane per wanted position.
  const int n = pn;
  int i = n - 1;
  if (n <= 4 * GE_WAVE && need <= GE_WAVE) {
    // the j's stay in (up to four) registers, j_i in lane i % 64 of register i / 64: v_writelane in the scan, v_readlane in the
    // trace, no LDS anywhere in the two serial loops
    uint32_t j0 = 0, j1 = 0, j2 = 0, j3 = 0;
    while (i >= 1) {
      if (nppos >= GE_MT_N) { ge_mt_twist(mt, lane); nppos = 0; }
      const int avail = (GE_MT_N - nppos < GE_WAVE) ? (GE_MT_N - nppos) : GE_WAVE;
      const uint32_t dr = lane < avail ? ge_temper(mt[nppos + lane]) : 0u;
      int k = 0;
      while (i >= 1 && k < avail) {  // wave-uniform
        const uint32_t j = ge_readlane_u32(dr, k) & ge_mask_below((uint32_t)i);
        k++;
        if (j > (uint32_t)i) continue;
        const int r = i >> 6;
        if (r == 0) j0 = ge_writelane_u32(j0, j, i & 63); else if (r == 1) j1 = ge_writelane_u32(j1, j, i & 63);
        else if (r == 2) j2 = ge_writelane_u32(j2, j, i & 63); else j3 = ge_writelane_u32(j3, j, i & 63);
        i--;
      }
      nppos += k;
    }
    int pos = lane;
    for (int q = 1; q < n; q++) {
      const int r = q >> 6;
      const int j = (int)ge_readlane_u32(r == 0 ? j0 : r == 1 ? j1 : r == 2 ? j2 : j3, q & 63);
      pos = (pos == q) ? j : ((pos == j) ? q : pos);
    }
    if (lane < need && lane < n) c.perm[lane] = pos;
    ge_wave_sync();
    return;
  }
  while (i >= 1) {
    if (nppos >= GE_MT_N) { ge_mt_twist(mt, lane); nppos = 0; }
    const int avail = (GE_MT_N - nppos < GE_WAVE) ? (GE_MT_N - nppos) : GE_WAVE;
    const uint32_t dr = lane < avail ? ge_temper(mt[nppos + lane]) : 0u;
    int k = 0;
    while (i >= 1 && k < avail) {  // wave-uniform
      const uint32_t j = ge_readlane_u32(dr, k) & ge_mask_below((uint32_t)i);
      k++;
      if (j > (uint32_t)i) continue;
      if (lane == 0) c.perm[i] = (int)j;
      i--;
    }
    nppos += k;
  }
  ge_wave_sync();
  for (int p0 = 0; p0 < need; p0 += GE_WAVE) {
    int pos = p0 + lane;
    for (int q = 1; q < n; q++) { const int j = c.perm[q]; pos = (pos == q) ? j : ((pos == j) ? q : pos); }
    if (p0 + lane < need && p0 + lane < n) c.dist[p0 + lane] = pos;  // perm[] still holds the j's the next chunk reads
  }
  ge_wave_sync();
  for (int p = lane; p < need && p < n; p += GE_WAVE) c.perm[p] = c.dist[p];
  ge_wave_sync();
}

// [nx] dijkstra from `src` to every node: least fixpoint of d[u] = min_v fl(d[v] + w(v,u)) (float addition is monotone, so
// the distances do not depend on the relaxation order); Jacobi sweeps in LDS.  Distances are left in c.sigma.
GE_DEV void ge_dijkstra_wave(const GeRctx &c, int n, int src, int lane, double cutoff = __builtin_inf()) {
  for (int v = lane; v < n; v += GE_WAVE) c.sigma[v] = (v == src) ? 0.0 : __builtin_inf();
  ge_wave_sync();
  for (int it = 0; it < n; it++) {
    uint64_t any = 0;
    for (int k0 = 0; k0 < n; k0 += GE_WAVE) {
      int v = k0 + lane; bool ch = false;
      if (v < n) {
        double best = c.sigma[v];
        for (int k = c.rowptr[v]; k < c.rowptr[v + 1]; k++) {
          double d = c.sigma[c.colw[k] >> 4] + ge_wlut(c.colw[k] & 15);
          if (d < best && d <= cutoff) { best = d; ch = true; }  // beyond the cutoff nothing is needed: a node inside it has its whole shortest-path prefix inside
        }
        c.delta[v] = best;
      }
      any |= ge_ballot(ch);
    }
    ge_wave_sync();
    for (int v = lane; v < n; v += GE_WAVE) c.sigma[v] = c.delta[v];
    ge_wave_sync();
    if (!any) break;
  }
}

// CPython tuple hash of (a, b), small non-negative ints (Objects/tupleobject.c, xxHash-style; hash(int) = int)
GE_DEV uint64_t ge_pytuple2_hash(uint64_t a, uint64_t b) {
  const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P5 = 2870177450012600261ull;
  uint64_t acc = P5;
  acc += a * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
  acc += b * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
  acc += 2ull ^ (P5 ^ 3527539ull);
  return acc == ~0ull ? 1546275796ull : acc;
}

// multicast_routing.py:108-118 (is_eval_env): total delay of the union of the shortest paths source -> destinations.
// [nx] _dijkstra_multisource keeps for every node the path of its LAST strict improvement and pops a heap of
// (distance, push counter, node): the pop order is reproduced without a heap (a node's live entry is its latest push),
// and the python set of (u, v) tuples the edges are collected in is summed in CPython's set iteration order
// (Objects/setobject.c: 8 slots growing to the power of two above 4*used at fill*5 >= mask*3; slot, 9 linear probes,
// then i*5 + 1 + (perturb >>= 5)).  The set tables live in the slot's x rows, which are rewritten afterwards.
GE_DEV double ge_multicast_baseline(const GeParams &P, const GeRctx &c, int env, int lane) {
  const int n = P.n, W = P.W;
  double *seen = c.sigma; int *cnt = c.dist, *pred = c.fill; uint64_t *fin = c.bits;
  for (int v = lane; v < n; v += GE_WAVE) { seen[v] = __builtin_inf(); cnt[v] = -1; pred[v] = -1; }
  if (lane < W) fin[lane] = 0ull;
  ge_wave_sync();
  if (lane == 0) { seen[0] = 0.0; cnt[0] = 0; }
  int counter = 1;
  ge_wave_sync();
  for (;;) {
    double bs = __builtin_inf(); int bc = 0x7fffffff, bv = -1;
    for (int v = lane; v < n; v += GE_WAVE)
      if (cnt[v] >= 0 && !((fin[v >> 6] >> (v & 63)) & 1ull) && (bv < 0 || seen[v] < bs || (seen[v] == bs && cnt[v] < bc))) { bs = seen[v]; bc = cnt[v]; bv = v; }
    for (int off = 32; off >= 1; off >>= 1) {
      const double os = ge_shfl_f64(bs, lane ^ off); const int oc = ge_shfl_i32(bc, lane ^ off), ov = ge_shfl_i32(bv, lane ^ off);
      if (ov >= 0 && (bv < 0 || os < bs || (os == bs && oc < bc))) { bs = os; bc = oc; bv = ov; }
    }
    if (bv < 0) break;
    const int v = bv;
    ge_wave_sync();
    if (lane == 0) fin[v >> 6] |= 1ull << (v & 63);
    ge_wave_sync();
    for (int k0 = c.rowptr[v]; k0 < c.rowptr[v + 1]; k0 += GE_WAVE) {  // G._adj[v] in insertion order
      const int k = k0 + lane; const bool valid = k < c.rowptr[v + 1];
      const int u = valid ? (c.colw[k] >> 4) : 0;
      const double d = valid ? bs + ge_wlut(c.colw[k] & 15) : 0.0;
      const bool imp = valid && !((fin[u >> 6] >> (u & 63)) & 1ull) && (cnt[u] < 0 || d < seen[u]);
      const uint64_t Bm = ge_ballot(imp);
      if (imp) { seen[u] = d; cnt[u] = counter + ge_popc64(Bm & ((1ull << lane) - 1ull)); pred[u] = v; }
      counter += ge_popc64(Bm);
      ge_wave_sync();
    }
  }
  double total = 0.0;
  if (lane == 0) {
    uint32_t *slab = (uint32_t *)(P.buf.x + (int64_t)env * n * P.F);
    const uint32_t cap = (uint32_t)(n * P.F), EMPTY = 0xffffffffu;
    uint32_t *tab = slab; uint32_t mask = 7, used = 0; bool low = true;
    for (uint32_t i = 0; i < 8; i++) tab[i] = EMPTY;
    int *stack = (int *)c.elist;  // u32[m], m >= n - 1; free once the CSR is built
    for (int di = 1; di <= P.n_dests; di++) {  // for d in dests: for u, v in zip(path[:-1], path[1:]): add((u, v))
      int len = 0;
      for (int v = c.perm[di]; v != 0; v = pred[v]) stack[len++] = v;
      int u = 0;
      for (int q = len - 1; q >= 0; q--) {
        const int v = stack[q];
        const uint32_t key = ((uint32_t)u << 16) | (uint32_t)v;
        const uint64_t h = ge_pytuple2_hash((uint64_t)u, (uint64_t)v);
        u = v;
        uint64_t perturb = h; uint32_t i = (uint32_t)h & mask; int slot = -1; bool present = false;
        for (;;) {
          const uint32_t probes = (i + 9 <= mask) ? 9u : 0u;
          for (uint32_t j = 0; j <= probes && slot < 0 && !present; j++) {
            if (tab[i + j] == EMPTY) slot = (int)(i + j);
            else if (tab[i + j] == key) present = true;
          }
          if (slot >= 0 || present) break;
          perturb >>= 5; i = (uint32_t)((uint64_t)i * 5 + 1 + perturb) & mask;
        }
        if (present) continue;
        tab[slot] = key; used++;
        if ((uint64_t)used * 5 < (uint64_t)mask * 3) continue;
        uint32_t newsize = 8; while (newsize <= used * 4) newsize <<= 1;
        uint32_t *nt = low ? slab + (cap - newsize) : slab;  // old + new tables fit the slab (used <= n - 1)
        for (uint32_t q2 = 0; q2 < newsize; q2++) nt[q2] = EMPTY;
        for (uint32_t q2 = 0; q2 <= mask; q2++) {
          const uint32_t kk = tab[q2];
          if (kk == EMPTY) continue;
          const uint64_t hh = ge_pytuple2_hash((uint64_t)(kk >> 16), (uint64_t)(kk & 0xffffu));
          uint64_t pt = hh; uint32_t ii = (uint32_t)hh & (newsize - 1);
          for (;;) {
            int sl = -1;
            if (nt[ii] == EMPTY) sl = (int)ii;
            else if (ii + 9 <= newsize - 1) for (uint32_t j = 1; j <= 9 && sl < 0; j++) if (nt[ii + j] == EMPTY) sl = (int)(ii + j);
            if (sl >= 0) { nt[sl] = kk; break; }
            pt >>= 5; ii = (uint32_t)((uint64_t)ii * 5 + 1 + pt) & (newsize - 1);
          }
        }
        tab = nt; mask = newsize - 1; low = !low;
      }
    }
    for (uint32_t q2 = 0; q2 <= mask; q2++) {  // sum([G[u][v]['delay'] for u, v in all_path_edges])
      const uint32_t kk = tab[q2];
      if (kk == EMPTY) continue;
      const int u = (int)(kk >> 16), v = (int)(kk & 0xffffu);
      for (int k = c.rowptr[u]; k < c.rowptr[u + 1]; k++) if ((c.colw[k] >> 4) == v) { total += ge_wlut(c.colw[k] & 15); break; }
    }
  }
  total = ge_shfl_f64(total, 0);
  ge_wave_sync();
  return total;
}

// ---------------------------------------------------------------------------------------------------------------------
// Own baselines.  networkx's Kou Steiner tree, Christofides tour and clique-removal independent set depend on
// dict / set iteration orders deep inside the library; the survey asks validity and bound checks of them, not bit parity.
// These are this project's deterministic heuristics of the same kind, step for step what oracle/ge_oracle.c does
// (greedy_mis_size, kou_style_steiner, mst_total), and are compared with the oracle bit for bit.

GE_DEV uint64_t ge_full_word_n(int n, int w) {  // word w of the node set {0..n-1}
  const int lo = w * 64, hi = lo + 64 < n ? lo + 64 : n;
  return hi <= lo ? 0ull : (hi - lo == 64 ? ~0ull : ((1ull << (hi - lo)) - 1ull));
}

// total weight of a minimum spanning tree, the picked weights added in ascending order (python sum over Kruskal's edges,
// steiner_tree.py:80-81).  Weight codes: Prim on the integer codes and a counting sort; spatial TSP: float64 weights.
GE_DEV double ge_mst_total_wave(const GeParams &P, const GeRctx &c, int env, int lane) {
  const int n = P.n, W = P.W;
  uint64_t *intree = c.bits;
  if (lane < W) intree[lane] = 0ull;
  double s = 0.0;
  if (!P.spatial) {
    for (int v = lane; v < n; v += GE_WAVE) c.dist[v] = (v == 0) ? 0 : 255;
    if (lane < 16) c.misc[lane] = 0;
    ge_wave_sync();
    for (int it = 0; it < n; it++) {
      uint32_t best = 0xffffffffu;
      for (int v = lane; v < n; v += GE_WAVE)
        if (!((intree[v >> 6] >> (v & 63)) & 1ull)) { uint32_t key = ((uint32_t)c.dist[v] << 16) | (uint32_t)v; if (key < best) best = key; }
      for (int off = 32; off >= 1; off >>= 1) { uint32_t o = ge_shfl_u32(best, lane ^ off); if (o < best) best = o; }
      int pick = (int)(best & 0xffffu), code = (int)(best >> 16);
      ge_wave_sync();
      if (lane == 0) { intree[pick >> 6] |= 1ull << (pick & 63); if (it) c.misc[code & 15]++; }
      ge_wave_sync();
      for (int k = c.rowptr[pick] + lane; k < c.rowptr[pick + 1]; k += GE_WAVE) {
        int u = c.colw[k] >> 4, cd = c.colw[k] & 15;
        if (!((intree[u >> 6] >> (u & 63)) & 1ull) && cd < c.dist[u]) c.dist[u] = cd;
      }
      ge_wave_sync();
    }
    for (int code = 3; code <= 10; code++) for (int r = 0; r < c.misc[code]; r++) s += ge_wlut(code);
    ge_wave_sync();
    return s;
  }
  const double *sw = P.buf.sw64 + (int64_t)env * P.E;  // ascending-neighbour order
  double *key = c.sigma, *picked = c.delta;
  for (int v = lane; v < n; v += GE_WAVE) key[v] = (v == 0) ? 0.0 : __builtin_inf();
  ge_wave_sync();
  for (int it = 0; it < n; it++) {
    double bk = __builtin_inf(); int bv = 0x7fffffff;
    for (int v = lane; v < n; v += GE_WAVE)
      if (!((intree[v >> 6] >> (v & 63)) & 1ull) && (key[v] < bk || (key[v] == bk && v < bv))) { bk = key[v]; bv = v; }
    for (int off = 32; off >= 1; off >>= 1) {
      const double ok = ge_shfl_f64(bk, lane ^ off); const int ov = ge_shfl_i32(bv, lane ^ off);
      if (ok < bk || (ok == bk && ov < bv)) { bk = ok; bv = ov; }
    }
    const int pick = bv;
    ge_wave_sync();
    if (lane == 0) { intree[pick >> 6] |= 1ull << (pick & 63); if (it) picked[it - 1] = bk; }
    ge_wave_sync();
    for (int k = c.rowptr[pick] + lane; k < c.rowptr[pick + 1]; k += GE_WAVE) {
      const int u = c.colw[k] >> 4;
      const double w = sw[ge_sorted_pos(c, W, pick, u)];
      if (!((intree[u >> 6] >> (u & 63)) & 1ull) && w < key[u]) key[u] = w;
    }
    ge_wave_sync();
  }
  if (lane == 0) {  // ascending, then 0 + w0 + w1 + ...
    for (int a = 1; a < n - 1; a++) { const double w = picked[a]; int b = a - 1; while (b >= 0 && picked[b] > w) { picked[b + 1] = picked[b]; b--; } picked[b + 1] = w; }
    for (int a = 0; a < n - 1; a++) s += picked[a];
  }
  s = ge_shfl_f64(s, 0);
  ge_wave_sync();
  return s;
}

// maximal independent set, min-degree greedy (lowest index on ties); returns its size
GE_DEV double ge_greedy_mis_wave(const GeParams &P, const GeRctx &c, int lane) {
  const int n = P.n, W = P.W;
  uint64_t *alive = c.bits;
  if (lane < W) alive[lane] = ge_full_word_n(n, lane);
  ge_wave_sync();
  int size = 0;
  for (;;) {
    uint32_t best = 0xffffffffu;
    for (int v = lane; v < n; v += GE_WAVE) {
      if (!((alive[v >> 6] >> (v & 63)) & 1ull)) continue;
      int d = 0;
      for (int w = 0; w < W; w++) d += ge_popc64(c.abits[v * W + w] & alive[w]);
      const uint32_t key = ((uint32_t)d << 16) | (uint32_t)v;
      if (key < best) best = key;
    }
    for (int off = 32; off >= 1; off >>= 1) { uint32_t o = ge_shfl_u32(best, lane ^ off); if (o < best) best = o; }
    if (best == 0xffffffffu) break;
    const int pick = (int)(best & 0xffffu);
    size++;
    ge_wave_sync();
    if (lane < W) alive[lane] &= ~(c.abits[pick * W + lane] | (((pick >> 6) == lane) ? (1ull << (pick & 63)) : 0ull));
    ge_wave_sync();
  }
  return (double)size;
}

// 2-approximate Steiner tree in the manner of Kou, Markowsky and Berman (see oracle/ge_oracle.c kou_style_steiner for the
// statement of every tie-break): Prim over the terminals in the metric closure with one Dijkstra per joining terminal,
// closure edges expanded along that Dijkstra tree, Prim over the union of the paths, non-terminal leaves pruned, weights
// added in ascending (u, v).  Bit matrices S (union of paths) and T2 (tree) and the terminal keys live in the `kou` carve.
GE_DEV double ge_kou_steiner_wave(const GeParams &P, const GeRctx &c, int lane) {
  const int n = P.n, W = P.W, T = P.n_dests + 1;
  uint64_t *S = (uint64_t *)(ge_dyn_smem() + P.lds.kou), *T2 = S + n * W;
  double *key = (double *)(T2 + n * W);
  int *par = (int *)(key + T), *in = par + T;
  for (int i = lane; i < 2 * n * W; i += GE_WAVE) S[i] = 0ull;
  ge_dijkstra_wave(c, n, c.perm[0], lane);
  for (int j = lane; j < T; j += GE_WAVE) { key[j] = c.sigma[c.perm[j]]; par[j] = 0; in[j] = (j == 0); }
  ge_wave_sync();
  for (int it = 1; it < T; it++) {
    double bk = __builtin_inf(); int bj = 0x7fffffff;
    for (int k = lane; k < T; k += GE_WAVE) if (!in[k] && (key[k] < bk || (key[k] == bk && k < bj))) { bk = key[k]; bj = k; }
    for (int off = 32; off >= 1; off >>= 1) {
      const double ok = ge_shfl_f64(bk, lane ^ off); const int oj = ge_shfl_i32(bj, lane ^ off);
      if (ok < bk || (ok == bk && oj < bj)) { bk = ok; bj = oj; }
    }
    const int j = bj, tj = c.perm[j];
    ge_wave_sync();
    ge_dijkstra_wave(c, n, tj, lane);
    for (int v = c.perm[par[j]]; v != tj;) {  // walk the Dijkstra tree of t_j from the parent terminal back to t_j
      int u = 0x7fffffff;
      for (int k = c.rowptr[v] + lane; k < c.rowptr[v + 1]; k += GE_WAVE) {
        const int cn = c.colw[k] >> 4;
        if (c.sigma[cn] + ge_wlut(c.colw[k] & 15) == c.sigma[v] && cn < u) u = cn;
      }
      for (int off = 32; off >= 1; off >>= 1) { const int o = ge_shfl_i32(u, lane ^ off); if (o < u) u = o; }
      if (lane == 0) { S[v * W + (u >> 6)] |= 1ull << (u & 63); S[u * W + (v >> 6)] |= 1ull << (v & 63); }
      v = u;
    }
    ge_wave_sync();
    if (lane == 0) in[j] = 1;
    ge_wave_sync();
    for (int k = lane; k < T; k += GE_WAVE) if (!in[k] && c.sigma[c.perm[k]] < key[k]) { key[k] = c.sigma[c.perm[k]]; par[k] = j; }
    ge_wave_sync();
  }
  // Prim over the union S from the first terminal: keys (w, node), lowest parent on ties
  double *d2 = c.sigma; int *p2 = c.dist; uint64_t *done = c.bits;
  for (int v = lane; v < n; v += GE_WAVE) { d2[v] = __builtin_inf(); p2[v] = -1; }
  if (lane < W) done[lane] = 0ull;
  ge_wave_sync();
  if (lane == 0) d2[c.perm[0]] = 0.0;
  ge_wave_sync();
  for (;;) {
    double bk = __builtin_inf(); int bv = 0x7fffffff;
    for (int v = lane; v < n; v += GE_WAVE)
      if (!((done[v >> 6] >> (v & 63)) & 1ull) && d2[v] < __builtin_inf() && (d2[v] < bk || (d2[v] == bk && v < bv))) { bk = d2[v]; bv = v; }
    for (int off = 32; off >= 1; off >>= 1) {
      const double ok = ge_shfl_f64(bk, lane ^ off); const int ov = ge_shfl_i32(bv, lane ^ off);
      if (ov != 0x7fffffff && (bv == 0x7fffffff || ok < bk || (ok == bk && ov < bv))) { bk = ok; bv = ov; }
    }
    if (bv == 0x7fffffff) break;
    const int v = bv;
    ge_wave_sync();
    if (lane == 0) {
      done[v >> 6] |= 1ull << (v & 63);
      if (p2[v] >= 0) { const int q = p2[v]; T2[v * W + (q >> 6)] |= 1ull << (q & 63); T2[q * W + (v >> 6)] |= 1ull << (v & 63); }
    }
    ge_wave_sync();
    for (int k = c.rowptr[v] + lane; k < c.rowptr[v + 1]; k += GE_WAVE) {
      const int u = c.colw[k] >> 4;
      if (!((S[v * W + (u >> 6)] >> (u & 63)) & 1ull) || ((done[u >> 6] >> (u & 63)) & 1ull)) continue;
      const double w = ge_wlut(c.colw[k] & 15);
      if (w < d2[u] || (w == d2[u] && v < p2[u])) { d2[u] = w; p2[u] = v; }
    }
    ge_wave_sync();
  }
  // prune non-terminal leaves (confluent: any order ends in the same tree); terminals as a bit set in c.bits[W..2W)
  uint64_t *tb = c.bits + W;
  if (lane < W) { uint64_t b = 0; for (int k = 0; k < T; k++) if ((c.perm[k] >> 6) == lane) b |= 1ull << (c.perm[k] & 63); tb[lane] = b; }
  ge_wave_sync();
  for (;;) {
    uint64_t any = 0;
    for (int v0 = 0; v0 < n; v0 += GE_WAVE) {
      const int v = v0 + lane;
      bool leaf = false; int nb = 0;
      if (v < n && !((tb[v >> 6] >> (v & 63)) & 1ull)) {
        int deg = 0;
        for (int w = 0; w < W; w++) { const uint64_t r = T2[v * W + w]; deg += ge_popc64(r); if (r) nb = w * 64 + ge_ctz64(r); }
        leaf = deg == 1;
      }
      any |= ge_ballot(leaf);
      ge_wave_sync();
      if (leaf) { T2[v * W + (nb >> 6)] = 0ull; atomicAnd((unsigned long long *)&T2[nb * W + (v >> 6)], ~(1ull << (v & 63))); }
      ge_wave_sync();
    }
    if (!any) break;
  }
  double cost = 0.0;
  if (lane == 0)
    for (int u = 0; u < n; u++)
      for (int w = 0; w < W; w++)
        for (uint64_t r = T2[u * W + w]; r; r &= r - 1) {
          const int v = w * 64 + ge_ctz64(r);
          if (v > u) cost += ge_wlut(c.wsort[ge_sorted_pos(c, W, u, v)]);
        }
  cost = ge_shfl_f64(cost, 0);
  ge_wave_sync();
  return cost;
}

// one 32-bit numpy draw, wave-uniform (every lane reads the same word)
GE_DEV uint32_t ge_np_next(uint32_t *mt, int &nppos, int lane) {
  if (nppos >= GE_MT_N) { ge_mt_twist(mt, lane); nppos = 0; }
  const uint32_t r = ge_temper(mt[nppos]); nppos++;
  return r;
}
// [np] legacy randint(0, cnt) for cnt >= 1 (what np.random.choice(list) draws its index with): masked rejection,
// nothing is drawn when cnt == 1
GE_DEV int ge_np_index(uint32_t *mt, int &nppos, int cnt, int lane) {
  if (cnt <= 1) return 0;
  const uint32_t rng = (uint32_t)(cnt - 1), mask = ge_mask_below(rng);
  for (;;) { const uint32_t v = ge_np_next(mt, nppos, lane) & mask; if (v <= rng) return (int)v; }
}
// index of the k-th set bit (k from 0) of a two-word set; the caller guarantees it exists
GE_DEV int ge_select2(uint64_t w0, uint64_t w1, int k) {
  const int c0 = ge_popc64(w0);
  uint64_t w = w0; int base = 0;
  if (k >= c0) { w = w1; k -= c0; base = 64; }
  for (int j = 0; j < k; j++) w &= w - 1;
  return base + ge_ctz64(w);
}

#ifndef GE_PPD_WIDE_ABOVE
#define GE_PPD_WIDE_ABOVE 128  // (a test build lowers it to run the placement of large graphs on small ones)
#endif
// ---- PerishableProductDelivery above 128 nodes: no n x n matrix in LDS and node sets of W words.
// Distances from `p` to every node into dist[] (LDS), straight from the adjacency bit rows and the nibble matrix of delays (the CSR
// does not exist yet inside the rejection loop): in-place relaxation sweeps to the least fixpoint of d[v] = min_u d[u] + delay(u, v)
// -- float addition is monotone, so the fixpoint does not depend on the order of the relaxations.
GE_DEV void ge_ppd_dist(const GeParams &P, const GeRctx &c, int p, double *dist, int lane) {
  const int n = P.n, W = P.W;
  for (int v = lane; v < n; v += GE_WAVE) dist[v] = (v == p) ? 0.0 : __builtin_inf();
  ge_wave_sync();
  for (;;) {
    bool improved = false;
    for (int v0 = 0; v0 < n; v0 += GE_WAVE) {
      const int v = v0 + lane;
      if (v < n) {
        double best = dist[v];
        for (int w = 0; w < W; w++)
          for (uint64_t r = c.abits[v * W + w]; r; r &= r - 1) {
            const int u = w * 64 + ge_ctz64(r);
            double wuv = 1.0;
            if (P.weighted && P.np_early) { const int cell = (u < v ? u : v) * n + (u < v ? v : u); wuv = ge_wlut((int)((c.wm[cell >> 3] >> (4 * (cell & 7))) & 15u)); }
            else if (P.weighted) wuv = ge_wlut((int)((const uint8_t *)c.wm)[ge_np_edge_number(c, W, u, v)]);  // above 256 nodes: the codes of the edges only, by edge number
            const double d = dist[u] + wuv;
            if (d < best) best = d;
          }
        if (best < dist[v]) { dist[v] = best; improved = true; }
      }
    }
    ge_wave_sync();
    if (!ge_ballot(improved)) break;
  }
}
// perishable_product_delivery.py:92-108, one attempt on a connected graph whose delay matrix is in the nibble matrix:
// delivery_time = rand() * (dt_max - dt_min) + dt_min; apsp = floyd_warshall; per product a pickup among the unused nodes
// and a drop-off among the unused nodes closer than delivery_time + 1e-6, listed in the key order of the apsp[pickup]
// dict ([nx] floyd_warshall builds defaultdicts: the node itself, the neighbours below it ascending -- G.edges reports
// an edge from its lower end --, the neighbours above it in adjacency insertion order, then every other node ascending,
// inserted by the reads of the first sweep).  pk / dp persist over failed attempts, as self.pickups / self.dropoffs do.
// Returns false when a pickup has no drop-off in range.  n <= 128 (two-word node sets); wave-uniform control flow.
GE_DEV bool ge_ppd_place(const GeParams &P, const GeRctx &c, double *D, double rnd, int &nppos, int *pk, int *dp, double &dt_out, int lane) {
  const int n = P.n, m = P.m, np_ = P.n_dests;
  const double dt = rnd * (P.dt_max - P.dt_min) + P.dt_min;
  dt_out = dt;
  for (int idx = lane; idx < n * n; idx += GE_WAVE) {
    const int u = idx / n, v = idx - u * n;
    double w = __builtin_inf();
    if (u == v) w = 0.0;
    else if ((c.abits[u * P.W + (v >> 6)] >> (v & 63)) & 1ull) {
      const int cell = (u < v ? u : v) * n + (u < v ? v : u);
      w = P.weighted ? ge_wlut((int)((c.wm[cell >> 3] >> (4 * (cell & 7))) & 15u)) : 1.0;
    }
    D[idx] = w;
  }
  ge_wave_sync();
  for (int w = 0; w < n; w++) {  // row w and column w do not change during sweep w (x + 0 == x, strict >): the cells are independent
    for (int v0 = 0; v0 < n; v0 += GE_WAVE) {
      const int v = v0 + lane;
      const double dwv = v < n ? D[w * n + v] : 0.0;
      if (v < n && dwv < __builtin_inf())  // inf + x never improves anything
        for (int u = 0; u < n; u++) {
          const double d = D[u * n + w] + dwv;  // D[u][w]: the same word for every lane (broadcast read)
          if (D[u * n + v] > d) D[u * n + v] = d;
        }
    }
    ge_wave_sync();
  }
  for (int i = 0; i < np_; i++) {
    uint64_t used0 = 0, used1 = 0;
    for (int j = 0; j < np_; j++) {
      if (pk[j] >= 0) { if (pk[j] < 64) used0 |= 1ull << pk[j]; else used1 |= 1ull << (pk[j] - 64); }
      if (dp[j] >= 0) { if (dp[j] < 64) used0 |= 1ull << dp[j]; else used1 |= 1ull << (dp[j] - 64); }
    }
    const uint64_t all0 = n >= 64 ? ~0ull : ((1ull << n) - 1ull), all1 = n <= 64 ? 0ull : (n >= 128 ? ~0ull : ((1ull << (n - 64)) - 1ull));
    const uint64_t f0 = all0 & ~used0, f1 = all1 & ~used1;
    const int p = ge_select2(f0, f1, ge_np_index(c.mt2, nppos, ge_popc64(f0) + ge_popc64(f1), lane));
    pk[i] = p;  // replaces what a failed attempt may have left in this entry: the used set is rebuilt
    used0 = 0; used1 = 0;
    for (int j = 0; j < np_; j++) {
      if (pk[j] >= 0) { if (pk[j] < 64) used0 |= 1ull << pk[j]; else used1 |= 1ull << (pk[j] - 64); }
      if (dp[j] >= 0) { if (dp[j] < 64) used0 |= 1ull << dp[j]; else used1 |= 1ull << (dp[j] - 64); }
    }
    const uint64_t pass0 = ge_ballot(lane < n && D[p * n + lane] < dt + 1e-6) & ~used0;
    const uint64_t pass1 = (n > 64 ? ge_ballot(lane + 64 < n && D[p * n + lane + 64] < dt + 1e-6) : 0ull) & ~used1;
    const uint64_t nb0 = c.abits[p * P.W], nb1 = P.W > 1 ? c.abits[p * P.W + 1] : 0ull;
    const uint64_t lo0 = p < 64 ? ((1ull << p) - 1ull) : ~0ull, lo1 = p < 64 ? 0ull : ((1ull << (p - 64)) - 1ull);  // nodes below p
    const uint64_t B0 = pass0 & nb0 & lo0, B1 = pass1 & nb1 & lo1;          // neighbours below p, ascending
    const uint64_t C0 = pass0 & nb0 & ~lo0, C1 = pass1 & nb1 & ~lo1;        // neighbours above p, adjacency insertion order
    const uint64_t R0 = pass0 & ~nb0, R1 = pass1 & ~nb1;                    // the rest, ascending (p itself is used)
    const int nB = ge_popc64(B0) + ge_popc64(B1), nC = ge_popc64(C0) + ge_popc64(C1), nR = ge_popc64(R0) + ge_popc64(R1);
    if (nB + nC + nR == 0) return false;
    int k = ge_np_index(c.mt2, nppos, nB + nC + nR, lane);
    int d;
    if (k < nB) d = ge_select2(B0, B1, k);
    else if (k < nB + nC && P.complete) d = ge_select2(C0, C1, k - nB);  // [nx] complete_graph adds the edges in ascending order (no edge list here)
    else if (k < nB + nC) {  // the (k - nB)-th edge of p, in insertion order, whose other end is in C
      k -= nB; d = -1;
      for (int e0 = 0; e0 < m && d < 0; e0 += GE_WAVE) {
        const int e = e0 + lane;
        int other = -1;
        if (e < m) { const uint32_t uv = c.elist[e]; const int u = (int)(uv & 0xffffu), v = (int)(uv >> 16); other = (u == p) ? v : ((v == p) ? u : -1); }
        const bool hit = other >= 0 && (((other < 64 ? C0 : C1) >> (other & 63)) & 1ull);
        const uint64_t H = ge_ballot(hit);
        const int cnt = ge_popc64(H);
        if (k < cnt) { uint64_t hh = H; for (int j = 0; j < k; j++) hh &= hh - 1; d = ge_shfl_i32(other, ge_ctz64(hh)); }
        else k -= cnt;
      }
    } else d = ge_select2(R0, R1, k - nB - nC);
    dp[i] = d;
  }
  for (int j = 0; j < np_; j++) if (pk[j] < 0 || dp[j] < 0) return false;
  return true;
}

// the k-th set bit (k from 0) of a set of W words given word by word; the caller guarantees it exists
template <class F>
GE_DEV int ge_select_words(int W, int k, F word) {
  for (int w = 0; w < W; w++) {
    uint64_t x = word(w);
    const int cnt = ge_popc64(x);
    if (k < cnt) { for (int j = 0; j < k; j++) x &= x - 1; return w * 64 + ge_ctz64(x); }
    k -= cnt;
  }
  return -1;
}
// ge_ppd_place for n > 128 (same draws, same candidate orders).  The drop-off filter `apsp[pickup][v] < delivery_time + 1e-6`
// (perishable_product_delivery.py:100-104) is evaluated on distances summed from the pickup outwards, where nx.floyd_warshall
// associates the same path's delays in the order of its sweeps: the two can differ in the last bit of a float64, which only matters
// for a node whose distance is within 1e-15 of the threshold (a uniformly drawn real): not observed, not excluded.
GE_DEV bool ge_ppd_place_wide(const GeParams &P, const GeRctx &c, double rnd, int &nppos, int *pk, int *dp, double &dt_out, int lane) {
  const int n = P.n, m = P.m, np_ = P.n_dests, W = P.W;
  const double dt = rnd * (P.dt_max - P.dt_min) + P.dt_min;
  dt_out = dt;
  uint64_t *used = c.bits, *pass = c.bits + W;  // c.bits: six sets of W words, free inside the rejection loop
  double *dist = c.sigma;
  auto rebuild_used = [&]() {
    ge_wave_sync();  // every lane has read the set it chose from before it is rewritten
    if (lane < W) {
      uint64_t u = 0;
      for (int j = 0; j < np_; j++) {
        if (pk[j] >= 0 && (pk[j] >> 6) == lane) u |= 1ull << (pk[j] & 63);
        if (dp[j] >= 0 && (dp[j] >> 6) == lane) u |= 1ull << (dp[j] & 63);
      }
      used[lane] = u;
    }
    ge_wave_sync();
  };
  for (int i = 0; i < np_; i++) {
    rebuild_used();
    int nfree = 0;
    for (int w = 0; w < W; w++) nfree += ge_popc64(ge_full_word_n(n, w) & ~used[w]);
    const int p = ge_select_words(W, ge_np_index(c.mt2, nppos, nfree, lane), [&](int w) { return ge_full_word_n(n, w) & ~used[w]; });
    pk[i] = p;  // replaces what a failed attempt may have left in this entry: the used set is rebuilt
    rebuild_used();
    ge_ppd_dist(P, c, p, dist, lane);
    for (int w = 0; w < W; w++) {
      const int v = w * GE_WAVE + lane;
      const uint64_t b = ge_ballot(v < n && dist[v] < dt + 1e-6);
      if (lane == 0) pass[w] = b & ~used[w];
    }
    ge_wave_sync();
    auto nbw = [&](int w) { return c.abits[p * W + w]; };
    auto low = [&](int w) { return w < (p >> 6) ? ~0ull : (w == (p >> 6) ? ((1ull << (p & 63)) - 1ull) : 0ull); };  // nodes below p
    auto Bw = [&](int w) { return pass[w] & nbw(w) & low(w); };    // neighbours below p, ascending
    auto Cw = [&](int w) { return pass[w] & nbw(w) & ~low(w); };   // neighbours above p, adjacency insertion order
    auto Rw = [&](int w) { return pass[w] & ~nbw(w); };            // the rest, ascending (p itself is used)
    int nB = 0, nC = 0, nR = 0;
    for (int w = 0; w < W; w++) { nB += ge_popc64(Bw(w)); nC += ge_popc64(Cw(w)); nR += ge_popc64(Rw(w)); }
    if (nB + nC + nR == 0) return false;
    int k = ge_np_index(c.mt2, nppos, nB + nC + nR, lane);
    int d;
    if (k < nB) d = ge_select_words(W, k, Bw);
    else if (k < nB + nC && P.complete) d = ge_select_words(W, k - nB, Cw);
    else if (k < nB + nC) {  // the (k - nB)-th edge of p, in insertion order, whose other end is in C
      k -= nB; d = -1;
      for (int e0 = 0; e0 < m && d < 0; e0 += GE_WAVE) {
        const int e = e0 + lane;
        int other = -1;
        if (e < m) { const uint32_t uv = c.elist[e]; const int u = (int)(uv & 0xffffu), v = (int)(uv >> 16); other = (u == p) ? v : ((v == p) ? u : -1); }
        const bool hit = other >= 0 && ((Cw(other >> 6) >> (other & 63)) & 1ull);
        const uint64_t H = ge_ballot(hit);
        const int cnt = ge_popc64(H);
        if (k < cnt) { uint64_t hh = H; for (int j = 0; j < k; j++) hh &= hh - 1; d = ge_shfl_i32(other, ge_ctz64(hh)); }
        else k -= cnt;
      }
    } else d = ge_select_words(W, k - nB - nC, Rw);
    dp[i] = d;
  }
  for (int j = 0; j < np_; j++) if (pk[j] < 0 || dp[j] < 0) return false;
  return true;
}

// MulticastRouting tail of the numpy stream (multicast_routing.py:98,106): dests = arange(1, n)[permutation(n - 1)[:k]],
// then ONE rand() for max_distance.  Leaves perm[0] = 0 (the source), perm[1..k] = dests and the rand in misc[0..1].
GE_DEV void ge_np_multicast_tail(const GeParams &P, const GeRctx &c, uint32_t *mt, int &nppos, int lane) {
  ge_np_terminals(P, c, mt, nppos, lane, P.n - 1, P.n_dests);
  const int k = P.n_dests;
  for (int base = ((k - 1) / GE_WAVE) * GE_WAVE; base >= 0; base -= GE_WAVE) {  // shift up by one slot, +1 on the values
    const int idx = base + lane;
    const int val = idx < k ? c.perm[idx] : 0;
    ge_wave_sync();
    if (idx < k) c.perm[idx + 1] = val + 1;
    ge_wave_sync();
  }
  if (lane == 0) c.perm[0] = 0;
  uint32_t r2[2];
  for (int q = 0; q < 2; q++) {
    if (nppos >= GE_MT_N) { ge_mt_twist(mt, lane); nppos = 0; }
    r2[q] = ge_temper(mt[nppos]); nppos++;
  }
  // [np] mt19937_next_double: (a >> 5, b >> 6) -> 53 bits
  if (lane == 0) *(double *)c.misc = ((double)(int32_t)(r2[0] >> 5) * 67108864.0 + (double)(int32_t)(r2[1] >> 6)) / 9007199254740992.0;
  ge_wave_sync();
}

// The numpy wave (second wave of the reset workgroup): everything the numpy stream produces that does not
// depend on the topology runs beside the python-stream graph sampling of the first wave.
template <int ENV>
GE_DEV void ge_numpy_wave(const GeParams &P, const GeRctx &c, const uint32_t *mt_src, int lane, int env, int np0, uint32_t *save) {
  constexpr int t = ENV;  // compile-time: every env type gets its own reset kernel, so none pays for the others' registers
  const int n = P.n;
  (void)env;  // only the diagnostic stamps name the slot
  if (t == GE_DENSEST_SUBGRAPH) {  // seeds numpy but never draws (densest_subgraph.py:52-98): the saved stream is the one it was given
    if (save && save != mt_src) { for (int i = lane; i < GE_MT_N; i += GE_WAVE) save[i] = mt_src[i]; if (lane == 0) save[GE_MT_N] = (uint32_t)np0; }
    return;
  }
  GE_STAMP(20);
  ge_mt_load(c.mt2, mt_src, lane);  // pre-seeded (or, continuing, as the previous regeneration left it)
  GE_STAMP(21);
  if (!P.np_early) return;               // big delay matrix: the first wave draws after the topology is known
  int nppos = np0;
  const bool path_like = (t == GE_SHORTEST_PATH || t == GE_LONGEST_PATH || t == GE_STEINER_TREE || t == GE_MULTICAST_ROUTING || t == GE_DISTRIBUTION_CENTER ||
                          t == GE_PERISHABLE_DELIVERY);
  if (P.spatial) {  // tsp.py:81-83: x, y = np.random.rand() * 10 per node; rand() = two 32-bit draws, no rejection
    uint32_t *raw = c.wm; double *xy = (double *)(c.wm + 4 * n);
    for (int p0 = 0; p0 < 4 * n; p0 += GE_WAVE) {
      if (nppos >= GE_MT_N) { ge_mt_twist(c.mt2, lane); nppos = 0; }
      int p = nppos + lane; bool valid = p < GE_MT_N && p0 + lane < 4 * n;
      if (valid) raw[p0 + lane] = ge_temper(c.mt2[p]);
      int adv = (GE_MT_N - nppos < GE_WAVE) ? (GE_MT_N - nppos) : GE_WAVE;
      if (p0 + adv > 4 * n) adv = 4 * n - p0;
      nppos += adv; p0 += adv - GE_WAVE;  // a short block (state boundary) advances by what it delivered
      ge_wave_sync();
    }
    for (int k = lane; k < 2 * n; k += GE_WAVE) {  // [np] mt19937_next_double: (a >> 5, b >> 6) -> 53 bits
      int32_t a = (int32_t)(raw[2 * k] >> 5), b = (int32_t)(raw[2 * k + 1] >> 6);
      xy[k] = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0 * 10;
    }
    ge_wave_sync();
  } else if (P.weighted) {
    if (path_like) {
      for (int i = lane; i < (n * n + 7) / 8; i += GE_WAVE) c.wm[i] = 0u;
      ge_wave_sync();
      ge_np_draws(P, c, c.mt2, nppos, n * n, lane, 0);
    } else ge_np_draws(P, c, c.mt2, nppos, t == GE_TSP ? P.m : n, lane, 1);
  }
  GE_STAMP(22);
  if (t == GE_DISTRIBUTION_CENTER) ge_np_draws(P, c, c.mt2, nppos, n, lane, 3);  // node costs, weighted or not
  if (t == GE_PERISHABLE_DELIVERY) {  // first attempt: delay matrix, then the rand() of delivery_time; the rest needs the topology
    const uint32_t ra = ge_np_next(c.mt2, nppos, lane), rb = ge_np_next(c.mt2, nppos, lane);
    if (lane == 0) { *(double *)c.misc = ((double)(int32_t)(ra >> 5) * 67108864.0 + (double)(int32_t)(rb >> 6)) / 9007199254740992.0; c.misc[2] = nppos; }
    ge_wave_sync();
    return;  // the first wave goes on with this stream (and saves it)
  }
  if (t == GE_MULTICAST_ROUTING) ge_np_multicast_tail(P, c, c.mt2, nppos, lane);
  else if (path_like) ge_np_terminals(P, c, c.mt2, nppos, lane, n, P.T);
  GE_STAMP(23);
  if (save) ge_mt_save(save, c.mt2, nppos, lane);
}

// DistributionCenter, n <= 64: nodes within `cutoff` of `s` (float64 sums taken from s outwards), by a label-correcting search
// of ONE lane over distances S[node * ss] -- the least fixpoint does not depend on the order the nodes are relaxed in.  The order is
// by ROUNDS: the nodes whose label improved in one round (a 64-bit set) are relaxed in the next, so a label is extended about once
// per hop count that reaches it (cutoff 1.0 over delays 0.3-0.9: four rounds, ~300 row entries per source; the LIFO stack of
// rounds 2-3 re-relaxed nodes as often as a deeper detour improved them: 106 us of one wave per slot for its 12 targets, the
// whole "writeout" phase of tools/phase_stamps_any.py).  rowptr / colw may be the LDS copies of the reset kernel or the global
// slabs; S is the lane's own column (stk / ks: unused scratch of the earlier form, kept in the signature for the callers).
template <class RP, class CW>
GE_DEV uint64_t ge_dc_search(double cutoff, int n, int s, const RP *rowptr, const CW *colw, double *S, int ss, uint8_t *stk, int ks) {
  (void)stk; (void)ks;
  for (int v = 0; v < n; v++) S[v * ss] = __builtin_inf();
  S[s * ss] = 0.0;
  uint64_t reached = 1ull << s, cur = 1ull << s;
  while (cur) {
    uint64_t nxt = 0;
    for (; cur; cur &= cur - 1) {
      const int u = ge_ctz64(cur);
      const double du = S[u * ss];
      const int r1 = rowptr[u + 1];
      // four row entries per trip: the entries, then the labels they point at, are loaded UNCONDITIONALLY (an entry past the end of
      // the row re-reads the row's first) and used afterwards (the relaxations of a trip touch different nodes: no parallel edges)
      for (int k0 = rowptr[u]; k0 < r1; k0 += 4) {
        uint32_t e[4]; double sv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) e[j] = (uint32_t)colw[k0 + j < r1 ? k0 + j : k0];
#pragma unroll
        for (int j = 0; j < 4; j++) sv[j] = S[(int)(e[j] >> 4) * ss];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int v = (int)(e[j] >> 4);
          const double d = du + ge_wlut((int)(e[j] & 15u));
          if (k0 + j < r1 && d <= cutoff && d < sv[j]) {
            S[v * ss] = d; reached |= 1ull << v;
            // every delay is at least 0.3 (codes 3 .. 10 of ge_wlut) and float64 addition is monotone: a label that cannot be extended
            // by 0.3 cannot be extended by any entry of its row, so its node is not relaxed again (cutoff 1.0: every label above 0.7,
            // i.e. most of what lies two or three hops out).  Kept out of the set rather than skipped when it comes up: the lanes of
            // a wave walk their sets in step, and a trip costs every lane what it costs the one with a row to scan
            if (d + 0.3 <= cutoff) nxt |= 1ull << v;
          }
        }
      }
    }
    cur = nxt;
  }
  return reached;
}

// Write-out of the slabs that are indexed by directed edge or copied from LDS as they stand -- edge_index, edge_attr, colw, scode,
// rev_edge, row_ptr, adj_bits, node_rec --, by BOTH waves of the workgroup (`tid` over `nthreads`): it needs nothing the first wave
// keeps in registers, and it was 45 of the 62 us a 256-node SteinerTree slot spent writing (one wave: 32 trips of a dozen dependent
// LDS reads each, the reverse-edge search the longest).  The reverse edge of u -> v is looked for in v's row four entries per trip,
// every read unconditional (a search that stops at the hit is a chain of deg / 2 dependent reads).
template <int ENV>
GE_DEV void ge_write_edge_slabs(const GeParams &P, const GeRctx &c, int env, int tid, int nthreads) {
  constexpr int t = ENV;
  const ge_buffers &G = P.buf;
  const int n = P.n, W = P.W, E = P.E;
  const int64_t nbase = (int64_t)env * n, ebase = (int64_t)env * E, Ne = P.edge_row_stride;
  for (int idx = tid; idx < E; idx += nthreads) {
    int u = ge_row_of(P, c, idx), v = ge_list_nbr(P, c, idx), code = ge_list_code(P, c, idx);
    G.edge_index[ebase + idx] = P.node_id_base + nbase + u;
    G.edge_index[Ne + ebase + idx] = P.node_id_base + nbase + v;
    float wv = (t == GE_DENSEST_SUBGRAPH || t == GE_MAX_INDEPENDENT_SET) ? 1.f : (float)ge_wlut(code);
    if (P.spatial) wv = (float)G.sw64[ebase + ge_sorted_pos_p(P, c, u, v)];
    if (P.Fe == 2) { G.edge_attr[(ebase + idx) * 2] = wv; G.edge_attr[(ebase + idx) * 2 + 1] = 0.f; }
    else G.edge_attr[ebase + idx] = wv;
    G.colw[ebase + idx] = (uint16_t)((v << 4) | code);
    G.scode[ebase + idx] = P.nowsort ? (uint8_t)code : c.wsort[idx];  // (complete graph: ascending-neighbour order is insertion order)
    if (G.rev_edge) {
      const int r0 = c.rowptr[v], r1 = c.rowptr[v + 1];
      int r = -1;
      for (int k0 = r0; k0 < r1 && r < 0; k0 += 4) {
        int w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = (int)(c.colw[k0 + j < r1 ? k0 + j : r0] >> 4);
#pragma unroll
        for (int j = 3; j >= 0; j--) if (k0 + j < r1 && w[j] == u) r = k0 + j;  // (a simple graph: one hit)
      }
      G.rev_edge[ebase + idx] = r;
    }
  }
  for (int v = tid; v <= n; v += nthreads) G.row_ptr[(int64_t)env * (n + 1) + v] = c.rowptr[v];
  for (int i = tid; i < n * W; i += nthreads) G.adj_bits[nbase * W + i] = c.abits[i];
  if (G.node_rec) {  // W == 1: {bit row, nibble-packed codes of the 16 smallest neighbours}
    for (int v = tid; v < n; v += nthreads) {
      uint64_t codes = 0; int d = c.rowptr[v + 1] - c.rowptr[v]; if (d > 16) d = 16;
      for (int k = 0; k < d; k++) codes |= (uint64_t)(c.wsort[c.rowptr[v] + k] & 15) << (4 * k);
      G.node_rec[(nbase + v) * 2] = c.abits[v]; G.node_rec[(nbase + v) * 2 + 1] = codes;
    }
  }
}

template <int ENV>
GE_DEV void ge_reset_env(const GeParams &P, int env, const uint32_t *seeds, const GeRun &run, const GeInject &inj) {
  // two waves: wave 0 = python stream + everything that needs the topology; wave 1 = numpy stream (ge_numpy_wave).
  // Inside a wave only wave-level hand-offs are used; the two block barriers are the join and the end of the slot.
  const int tid = ge_tid_fresh();
  const int lane = tid & (GE_WAVE - 1);
  const int wv = tid >> 6;
  const int n = P.n, ng = P.ng, W = P.W, m = P.m, E = P.E, F = P.F, T = P.T;
  constexpr int t = ENV;
  const bool path_like_t = (t == GE_SHORTEST_PATH || t == GE_LONGEST_PATH || t == GE_STEINER_TREE || t == GE_MULTICAST_ROUTING);
  GeRctx c = ge_carve(P);
  const int64_t nbase = (int64_t)env * n, ebase = (int64_t)env * E;
  // episode bookkeeping.  A full reset or an injection with seeds starts episode 0 with the given seed; a queued slot moves to
  // its next episode, whose generator states sit in ring entry (episode mod GE_SEED_DEPTH); an injection without seeds leaves
  // seed / episode alone (the episodes that follow continue the earlier sequence).  In queue mode seed[] / episode[] still hold
  // the OLD episode while this kernel runs -- its seeding workgroups read them too -- and the feature kernel, the last kernel
  // of a regeneration, advances them (ge_advance_episode).
  // run.cont (reset(seed=None), shortest_path.py:49-52): like a queued regeneration the slot moves to its next episode, but the
  // two streams are the ones its previous regeneration left in stream_state, not freshly seeded ones.
  // run.refill: P.buf is the engine's spare image; seed[] / episode[] (shared with the live slabs) name the episode the slot is
  // still running, the image receives the one after it.
  const bool restart = run.restart != 0;
  const bool next = run.next != 0;
  const bool queued = run.items == GE_ITEMS_QUEUE && !run.refill;  // a finished slot regenerated in place
  const int64_t episode = restart ? 0 : P.buf.episode[env] + (next ? 1 : 0);
  const uint32_t seed = restart ? (run.restart == 1 ? seeds[env] : inj.seeds[env]) : P.buf.seed[env] + (next ? (uint32_t)P.seed_stride : 0u);
  const int ring = (int)(episode % GE_SEED_DEPTH);
  // where the streams are left (recomputed at each use: nothing about it stays live through the kernel)
  auto keep_at = [&](int which) -> uint32_t * { return P.buf.stream_state ? P.buf.stream_state + ((int64_t)env * 2 + which) * GE_STREAM_WORDS : nullptr; };
  // a stream's read position when this regeneration starts: a freshly seeded state is twisted before its first draw
  auto pos0 = [&](int which) -> int { return run.cont ? (int)keep_at(which)[GE_MT_N] : GE_MT_N; };
  const uint32_t *mt_src = P.buf.mt_state + ((int64_t)env * GE_SEED_DEPTH + ring) * 2 * GE_MT_N;
  if (run.cont) mt_src = keep_at(0);
  int src = 0, dest = -1;
  int ppd_pk[5] = {-1, -1, -1, -1, -1}, ppd_dp[5] = {-1, -1, -1, -1, -1};  // perishable_product_delivery.py:72-73
  bool gen_failed = false;
  double ppd_dt = 0.0;
  if (wv == 1) {
    if (!run.inject) { ge_numpy_wave<ENV>(P, c, run.cont ? keep_at(1) : mt_src + GE_MT_N, lane, env, pos0(1), keep_at(1)); ge_sync(); }
    ge_sync();  // the first wave has the slot complete in LDS: both waves write the edge slabs
    ge_write_edge_slabs<ENV>(P, c, env, tid, 2 * GE_WAVE);
    ge_sync();
    return;
  }

  GE_STAMP(0);
  if (!run.inject) {
    // ---------------------------------------------------------------- topology (python stream)
    ge_mt_load(c.mt, mt_src, lane);  // pre-seeded (ge_k_seed)
    GE_STAMP(1);
    int pypos = pos0(0);
    int ppd_attempt = 0, ppd_pos = 0;  // PerishableProductDelivery: the numpy stream is continued by this wave after the join
    bool failed = false;               // the G(n, m) loop hit its round cap (wave-uniform)
    const int shift = 32 - (32 - ge_clz32((uint32_t)ng));  // getrandbits(ng.bit_length())
    for (;;) {
      GE_STAMP(29);  // (diagnostic build: the last attempt starts here)
      for (int i = lane; i < n * W; i += GE_WAVE) c.abits[i] = 0ull;
      ge_wave_sync();
      if (P.complete) {  // [nx] complete_graph: sorted rows
        for (int v = lane; v < ng; v += GE_WAVE)
          for (int w = 0; w < W; w++) {
            int lo = w * 64, hi = lo + 64; if (hi > ng) hi = ng;
            uint64_t bitsw = (hi <= lo) ? 0ull : ((hi - lo == 64) ? ~0ull : ((1ull << (hi - lo)) - 1ull));
            if ((v >> 6) == w) bitsw &= ~(1ull << (v & 63));
            c.abits[v * W + w] = bitsw;
          }
      } else {
        // [nx] gnm_random_graph.  A draw below ng is a node pick; picks pair up as (u, v) in
        // stream order; a pair is added unless u == v or the edge exists (in the matrix, or earlier in this round).
        int cnt = 0, have_u = 0, carry_u = 0;
        const uint64_t below = (1ull << lane) - 1ull;
        int rounds = 0;
        // TWO draws per lane and round (128 raw words): a round is a chain of ~5 dependent LDS / ballot hops whatever its width, so half the
        // rounds per attempt.  Draw j of a lane is stream position pypos + 64 j + lane; the picks of the second half follow those of the
        // first in the u, v, u, v, ... sequence.  Every round inserts first -- the returning ds_or on the canonical (min, max) bit tells a
        // lane that the same edge was proposed by another lane of this round, and only the keys a lane LOST on are looked at again -- and the
        // round that reaches edge m takes the bits of the pairs behind the completing draw out again (they were absent before the round,
        // so clearing them restores it).  (64 draws per round with an exact walk over every eligible key in the rounds that could reach
        // edge m, rounds 1-3: 216 us of autoreset per headline step; insert-first everywhere 208.5; 128 draws 206.5.)
        for (;;) {
          // an all-zero (never seeded) or corrupted generator state draws the same pair for ever: give up loudly instead of
          // hanging the GPU.  A healthy stream needs ~n^2 ln(n) / 128 rounds in the worst (nearly complete) case.
          if (++rounds > GE_GNM_ROUND_CAP) { failed = true; break; }
          if (pypos >= GE_MT_N) { ge_mt_twist(c.mt, lane); pypos = 0; }
          const int p0 = pypos + lane, p1 = p0 + GE_WAVE;
          const bool valid0 = p0 < GE_MT_N, valid1 = p1 < GE_MT_N;
          const uint32_t r0 = valid0 ? (ge_temper(c.mt[p0]) >> shift) : (uint32_t)ng, r1 = valid1 ? (ge_temper(c.mt[p1]) >> shift) : (uint32_t)ng;
          const bool pick0 = valid0 && r0 < (uint32_t)ng, pick1 = valid1 && r1 < (uint32_t)ng;
          const uint64_t V0 = ge_ballot(pick0), V1 = ge_ballot(pick1);
          const int n0 = ge_popc64(V0);
          const bool is_v0 = pick0 && ((ge_popc64(V0 & below) + have_u) & 1), is_v1 = pick1 && ((n0 + ge_popc64(V1 & below) + have_u) & 1);
          // the u of a v-pick is the pick just before it in the stream
          const uint64_t b0 = V0 & below, b1 = V1 & below;
          const uint32_t ur0 = ge_shfl_u32(r0, b0 ? 63 - (int)__builtin_clzll(b0) : 0), ur1 = ge_shfl_u32(r1, b1 ? 63 - (int)__builtin_clzll(b1) : 0);
          const int last0 = V0 ? (int)ge_readlane_u32(r0, 63 - (int)__builtin_clzll(V0)) : carry_u;  // last pick before the second half
          const int u0 = b0 ? (int)ur0 : carry_u, v0 = (int)r0, u1 = b1 ? (int)ur1 : last0, v1 = (int)r1;
          const bool elig0 = is_v0 && u0 != v0 && !((c.abits[u0 * W + (v0 >> 6)] >> (v0 & 63)) & 1ull);
          const bool elig1 = is_v1 && u1 != v1 && !((c.abits[u1 * W + (v1 >> 6)] >> (v1 & 63)) & 1ull);
          const uint32_t key0 = elig0 ? (uint32_t)((u0 < v0 ? u0 : v0) << 12 | (u0 < v0 ? v0 : u0)) : 0xffffffffu;
          const uint32_t key1 = elig1 ? (uint32_t)((u1 < v1 ? u1 : v1) << 12 | (u1 < v1 ? v1 : u1)) : 0xfffffffeu;
          (void)ge_ballot(elig0 || elig1);  // the point after which every lane has read the pre-round matrix (lockstep on the GPU; the CPU harness runs lanes one after another)
          bool lost0 = false, lost1 = false;
          if (elig0) {
            const int a = u0 < v0 ? u0 : v0, b = u0 < v0 ? v0 : u0; const unsigned long long bit = 1ull << (b & 63);
            lost0 = (atomicOr((unsigned long long *)&c.abits[a * W + (b >> 6)], bit) & bit) != 0;
          }
          if (elig1) {  // (behind the first half's: it sees their bits)
            const int a = u1 < v1 ? u1 : v1, b = u1 < v1 ? v1 : u1; const unsigned long long bit = 1ull << (b & 63);
            lost1 = (atomicOr((unsigned long long *)&c.abits[a * W + (b >> 6)], bit) & bit) != 0;
          }
          bool dup0 = false, dup1 = false;  // the same edge proposed earlier in this round
          uint64_t rem0 = ge_ballot(lost0), rem1 = ge_ballot(lost1);
          while (rem0 | rem1) {  // (wave-uniform) one trip per key that may have an earlier occurrence
            const uint32_t k0 = rem0 ? ge_readlane_u32(key0, ge_ctz64(rem0)) : ge_readlane_u32(key1, ge_ctz64(rem1));
            const uint64_t same0 = ge_ballot(elig0 && key0 == k0), same1 = ge_ballot(elig1 && key1 == k0);
            // every proposer but the first in stream order
            if (elig0 && key0 == k0 && lane != ge_ctz64(same0)) dup0 = true;
            if (elig1 && key1 == k0 && (same0 != 0ull || lane != ge_ctz64(same1))) dup1 = true;
            rem0 &= ~same0; rem1 &= ~same1;
          }
          bool acc0 = elig0 && !dup0, acc1 = elig1 && !dup1;
          const uint64_t A0 = ge_ballot(acc0), A1 = ge_ballot(acc1);
          const int na0 = ge_popc64(A0), na1 = ge_popc64(A1);
          const int arank0 = ge_popc64(A0 & below), arank1 = na0 + ge_popc64(A1 & below);
          const bool last_round = cnt + na0 + na1 >= m;
          int consumed, nacc = na0 + na1;
          if (last_round) {  // the stream stops right after the draw that completed edge m
            const int need = m - cnt - 1;
            bool undo0 = false, undo1 = false;
            if (need < na0) { const int fl = ge_ctz64(ge_ballot(acc0 && arank0 == need)); undo0 = acc0 && lane > fl; undo1 = acc1; acc0 = acc0 && lane <= fl; acc1 = false; consumed = fl + 1; }
            else { const int fl = ge_ctz64(ge_ballot(acc1 && arank1 == need)); undo1 = acc1 && lane > fl; acc1 = acc1 && lane <= fl; consumed = GE_WAVE + fl + 1; }
            if (undo0) { const int a = u0 < v0 ? u0 : v0, b = u0 < v0 ? v0 : u0; atomicAnd((unsigned long long *)&c.abits[a * W + (b >> 6)], ~(1ull << (b & 63))); }
            if (undo1) { const int a = u1 < v1 ? u1 : v1, b = u1 < v1 ? v1 : u1; atomicAnd((unsigned long long *)&c.abits[a * W + (b >> 6)], ~(1ull << (b & 63))); }
            nacc = need + 1;
          } else {
            consumed = (GE_MT_N - pypos < 2 * GE_WAVE) ? (GE_MT_N - pypos) : 2 * GE_WAVE;
            const int npick = n0 + ge_popc64(V1) + have_u;
            have_u = npick & 1;
            if (have_u && (V0 | V1)) carry_u = V1 ? (int)ge_readlane_u32(r1, 63 - (int)__builtin_clzll(V1)) : last0;  // with no pick in this round the pending u is carried unchanged
          }
          if (acc0) {  // (the (min, max) bit is already there; setting it again is harmless)
            atomicOr((unsigned long long *)&c.abits[u0 * W + (v0 >> 6)], (unsigned long long)(1ull << (v0 & 63)));
            atomicOr((unsigned long long *)&c.abits[v0 * W + (u0 >> 6)], (unsigned long long)(1ull << (u0 & 63)));
            c.elist[cnt + arank0] = (uint32_t)u0 | ((uint32_t)v0 << 16);
          }
          if (acc1) {
            atomicOr((unsigned long long *)&c.abits[u1 * W + (v1 >> 6)], (unsigned long long)(1ull << (v1 & 63)));
            atomicOr((unsigned long long *)&c.abits[v1 * W + (u1 >> 6)], (unsigned long long)(1ull << (u1 & 63)));
            c.elist[cnt + arank1] = (uint32_t)u1 | ((uint32_t)v1 << 16);
          }
          cnt += nacc; pypos += consumed;
          ge_wave_sync();
          if (last_round) break;
        }
      }
      ge_wave_sync();
      GE_STAMP(26);  // (diagnostic build: the last attempt's rounds end here)
      if (failed) break;
      // most disconnected G(n, m) samples have an isolated node (TSP also rejects a node of degree 1, tsp.py:65-68): one pass over
      // the degrees settles those attempts without the BFS; the launch lasts as long as its unluckiest slot's attempts
      uint64_t low = 0;
      for (int k = 0; k < W; k++) {
        const int v = k * GE_WAVE + lane; int d = 0;
        if (v < ng) for (int w = 0; w < W; w++) d += ge_popc64(c.abits[v * W + w]);
        low |= ge_ballot(v < ng && (d == 0 || (t == GE_TSP && d == 1)));
      }
      GE_STAMP(27);
      bool ok = !low && ge_connected(c, ng, W, -1, lane);
      GE_STAMP(28);
      if (ok && t == GE_TSP) ok = ge_connected(c, ng, W, 0, lane);  // tsp.py:69-71
      if (ok && t == GE_PERISHABLE_DELIVERY) {  // perishable_product_delivery.py:75-111: weights and placement belong to the attempt
        double rnd;
        const bool late = P.weighted && !P.np_early;  // above 256 nodes the n x n delay matrix does not fit LDS: this wave draws it for every
                                                      // attempt and keeps the codes of the attempt's edges only (c.tmp is free inside the loop)
        if (ppd_attempt == 0) { ge_sync(); if (late) ppd_pos = pos0(1); else { ppd_pos = c.misc[2]; rnd = *(const double *)c.misc; } }  // join: the numpy wave drew the first matrix and rand()
        if (ppd_attempt > 0 || late) {
          if (late) { ge_np_edge_cells(P, c, c.tmp, lane); ge_np_draws_cells(P, c, c.mt2, ppd_pos, lane, c.tmp, (uint8_t *)c.wm); }
          else if (P.weighted) { for (int i = lane; i < (n * n + 7) / 8; i += GE_WAVE) c.wm[i] = 0u; ge_wave_sync(); ge_np_draws(P, c, c.mt2, ppd_pos, n * n, lane, 0); }
          const uint32_t ra = ge_np_next(c.mt2, ppd_pos, lane), rb = ge_np_next(c.mt2, ppd_pos, lane);
          rnd = ((double)(int32_t)(ra >> 5) * 67108864.0 + (double)(int32_t)(rb >> 6)) / 9007199254740992.0;
        }
        ppd_attempt++;
        ok = (n > GE_PPD_WIDE_ABOVE) ? ge_ppd_place_wide(P, c, rnd, ppd_pos, ppd_pk, ppd_dp, ppd_dt, lane)
                       : ge_ppd_place(P, c, (double *)(ge_dyn_smem() + P.lds.fw), rnd, ppd_pos, ppd_pk, ppd_dp, ppd_dt, lane);
      }
      if (ok) break;
      // a rejected attempt: from here on this wave is on the launch's critical path (tools/slot_times.py: the median slot is written
      // 42 us after the launch starts, the slots that need three or four attempts after 70 us) -- let it win the issue arbitration
      // against its eleven neighbours' waves (320.7 -> 316.7 us per headline step; raising the numpy wave as well gives it back)
      ge_wave_priority(1);
    }
    ge_wave_priority(0);
    if (failed) {  // flag it, and carry on with a valid graph (a path plus chords up to m edges) so that nothing downstream runs out of bounds
      if (lane == 0) atomicOr((unsigned int *)&P.buf.work_count[1], 1u);
      gen_failed = true;
      for (int i = lane; i < n * W; i += GE_WAVE) c.abits[i] = 0ull;
      ge_wave_sync();
      if (lane == 0) {
        int cnt = 0;
        for (int d = 1; d < ng && cnt < m; d++)
          for (int u = 0; u + d < ng && cnt < m; u++) {
            const int v = u + d;
            c.abits[u * W + (v >> 6)] |= 1ull << (v & 63); c.abits[v * W + (u >> 6)] |= 1ull << (u & 63);
            c.elist[cnt++] = (uint32_t)u | ((uint32_t)v << 16);
          }
      }
      ge_wave_sync();
      if (t == GE_PERISHABLE_DELIVERY) { if (ppd_attempt == 0) { ge_sync(); ppd_pos = c.misc[2]; } for (int i = 0; i < P.n_dests; i++) { ppd_pk[i] = i; ppd_dp[i] = P.n_dests + i; } }
    }
    if (P.buf.stream_state) {
      ge_mt_save(keep_at(0), c.mt, pypos, lane);
      if (t == GE_PERISHABLE_DELIVERY) ge_mt_save(keep_at(1), c.mt2, ppd_pos, lane);  // this wave took the numpy stream over
    }
  } else {
    // ---------------------------------------------------------------- injected topology
    for (int i = lane; i < n * W; i += GE_WAVE) c.abits[i] = 0ull;
    for (int v = lane; v < n; v += GE_WAVE) c.fill[v] = 0;
    ge_wave_sync();
    for (int idx = lane; idx < E; idx += GE_WAVE) {
      int u = (int)inj.links[(ebase + idx) * 2], v = (int)inj.links[(ebase + idx) * 2 + 1];
      atomicAdd(&c.fill[u], 1);
      atomicOr((unsigned long long *)&c.abits[u * W + (v >> 6)], (unsigned long long)(1ull << (v & 63)));
    }
    ge_wave_sync();
  }

  GE_STAMP(2);
  // ------------------------------------------------------------------ CSR in insertion order (needs the topology only: it runs
  // while the numpy wave, the longer of the two, is still drawing)
  if (!run.inject) {
    for (int v = lane; v < n; v += GE_WAVE) { int d = 0; for (int w = 0; w < W; w++) d += ge_popc64(c.abits[v * W + w]); c.fill[v] = d; }
    ge_wave_sync();
  }
  {
    int carry = 0;
    for (int k0 = 0; k0 < n; k0 += GE_WAVE) {
      int v = k0 + lane; int d = v < n ? c.fill[v] : 0;
      int incl = ge_wave_incl_scan(d, lane);
      if (v < n) c.rowptr[v] = carry + incl - d;
      carry += ge_shfl_i32(incl, GE_WAVE - 1);
    }
    if (lane == 0) c.rowptr[n] = carry;
    ge_wave_sync();
  }
  if (run.inject) {
    for (int idx = lane; idx < E; idx += GE_WAVE) {
      int u = (int)inj.links[(ebase + idx) * 2], v = (int)inj.links[(ebase + idx) * 2 + 1];
      c.colw[idx] = (uint16_t)((v << 4) | (inj.wcode[ebase + idx] & 15));
      if (!P.complete) c.tmp[idx] = (uint32_t)u;
    }
    ge_wave_sync();
  } else if (P.complete) {
    if (!P.nocolw) for (int idx = lane; idx < E; idx += GE_WAVE) {
      const int u = ge_row_of(P, c, idx), k = idx - u * (ng - 1);
      int v = k < u ? k : k + 1;
      c.colw[idx] = (uint16_t)((v << 4) | 10);
    }
    ge_wave_sync();
  } else {
    for (int v = lane; v < n; v += GE_WAVE) c.fill[v] = 0;
    ge_wave_sync();
    for (int id = lane; id < m; id += GE_WAVE) {
      uint32_t e = c.elist[id]; int u = (int)(e & 0xffffu), v = (int)(e >> 16);
      int s = c.rowptr[u] + atomicAdd(&c.fill[u], 1); c.tmp[s] = ((uint32_t)id << 16) | (uint32_t)v;
      int s2 = c.rowptr[v] + atomicAdd(&c.fill[v], 1); c.tmp[s2] = ((uint32_t)id << 16) | (uint32_t)u;
    }
    ge_wave_sync();
    for (int v = lane; v < n; v += GE_WAVE) {  // order each row by edge id == insertion order
      int lo = c.rowptr[v], hi = c.rowptr[v + 1];
      for (int a = lo + 1; a < hi; a++) {
        uint32_t key = c.tmp[a]; int b = a - 1;
        while (b >= lo && c.tmp[b] > key) { c.tmp[b + 1] = c.tmp[b]; b--; }
        c.tmp[b + 1] = key;
      }
    }
    ge_wave_sync();
    for (int idx = lane; idx < E; idx += GE_WAVE) c.colw[idx] = (uint16_t)(((c.tmp[idx] & 0xffffu) << 4) | 10u);
    ge_wave_sync();
    for (int v = lane; v < n; v += GE_WAVE) for (int k = c.rowptr[v]; k < c.rowptr[v + 1]; k++) c.tmp[k] = (uint32_t)v;
    ge_wave_sync();
  }

  GE_STAMP(3);
  if (!run.inject && t != GE_PERISHABLE_DELIVERY) ge_sync();  // join: the numpy wave's codes and terminals are in LDS
  // ------------------------------------------------------------------ weight codes + terminals
  if (!run.inject) {
    if (!P.nowsort) for (int idx = lane; idx < E; idx += GE_WAVE) c.wsort[idx] = 10;
    ge_wave_sync();
    const bool path_like = path_like_t;
    const bool matrix_w = path_like_t || t == GE_DISTRIBUTION_CENTER || t == GE_PERISHABLE_DELIVERY;  // delay[u, v] of an n x n randint matrix
    if (P.weighted && matrix_w && P.np_early) {  // delay[u, v], u < v, from the nibble matrix the numpy wave filled
      for (int idx = lane; idx < E; idx += GE_WAVE) {
        int u = ge_row_of(P, c, idx), v = c.colw[idx] >> 4;
        int a = u < v ? u : v, b = u < v ? v : u, cell = a * n + b;
        c.wsort[ge_sorted_pos_p(P, c, u, v)] = (uint8_t)((c.wm[cell >> 3] >> (4 * (cell & 7))) & 15u);
      }
    } else if (P.spatial) {  // tsp.py:85-86: Euclidean distance, float64, stored in ascending-neighbour order
      const double *xy = (const double *)(c.wm + 4 * n);
      for (int idx = lane; idx < E; idx += GE_WAVE) {
        int u = ge_row_of(P, c, idx), v = c.colw[idx] >> 4;
        int a = u < v ? u : v, b = u < v ? v : u;  // G.edges yields (min, max)
        double dx = xy[2 * a] - xy[2 * b], dy = xy[2 * a + 1] - xy[2 * b + 1];
        P.buf.sw64[ebase + ge_sorted_pos_p(P, c, u, v)] = __builtin_sqrt(dx * dx + dy * dy);
      }
    } else if (P.weighted && t == GE_TSP && P.complete) {
      // complete graph ([nx] complete_graph = combinations(nodes, 2)): G.edges is (0,1) (0,2) .. (0,n-1) (1,2) ..., so the edge
      // (a, b), a < b, is draw number a (n - 1) - a (a - 1) / 2 + (b - a - 1): no scan over the edge list
      if (!P.nowsort) for (int idx = lane; idx < E; idx += GE_WAVE) {
        const int u = ge_row_of(P, c, idx), v = ge_list_nbr(P, c, idx);
        const int a = u < v ? u : v, b = u < v ? v : u;
        c.wsort[ge_sorted_pos_p(P, c, u, v)] = ((const uint8_t *)c.wm)[a * (ng - 1) - ((a * (a - 1)) >> 1) + (b - a - 1)];
      }
    } else if (P.weighted && t == GE_TSP) {  // k-th edge of G.edges (u ascending, insertion order, v > u) gets draw k
      int carry = 0;
      for (int k0 = 0; k0 < E; k0 += GE_WAVE) {
        int idx = k0 + lane; int fl = 0;
        if (idx < E) fl = ((int)(c.colw[idx] >> 4) > ge_row_of(P, c, idx)) ? 1 : 0;
        int incl = ge_wave_incl_scan(fl, lane);
        if (fl) {
          int u = ge_row_of(P, c, idx), v = c.colw[idx] >> 4;
          uint8_t code = ((const uint8_t *)c.wm)[carry + incl - 1];
          c.wsort[ge_sorted_pos(c, W, u, v)] = code; c.wsort[ge_sorted_pos(c, W, v, u)] = code;
        }
        carry += ge_shfl_i32(incl, GE_WAVE - 1);
      }
    } else if (t == GE_MAX_INDEPENDENT_SET) {
      for (int v = lane; v < n; v += GE_WAVE) c.fill[v] = P.weighted ? (int)((const uint8_t *)c.wm)[v] : 10;
    } else if (P.weighted && t == GE_PERISHABLE_DELIVERY) {  // drawn inside the rejection loop (codes by edge number in wm): to their rows
      ge_np_edge_cells(P, c, c.elist, lane);
      ge_np_codes_to_rows(P, c, c.elist, (const uint8_t *)c.wm, lane);
    } else if (P.weighted && matrix_w) {  // n too large for the dense matrix: draw now, codes land by rank
      int nppos = pos0(1);
      ge_np_draws_edges(P, c, c.mt2, nppos, lane);
      if (t == GE_DISTRIBUTION_CENTER) ge_np_draws(P, c, c.mt2, nppos, n, lane, 3);
      if (t == GE_MULTICAST_ROUTING) ge_np_multicast_tail(P, c, c.mt2, nppos, lane);
      else ge_np_terminals(P, c, c.mt2, nppos, lane, n, P.T);
      if (P.buf.stream_state) ge_mt_save(keep_at(1), c.mt2, nppos, lane);
    }
    ge_wave_sync();
    if (!P.nocolw) for (int idx = lane; idx < E; idx += GE_WAVE) {  // codes from ascending order back to insertion order
      int u = ge_row_of(P, c, idx), v = c.colw[idx] >> 4;
      c.colw[idx] = (uint16_t)((v << 4) | c.wsort[ge_sorted_pos_p(P, c, u, v)]);
    }
    ge_wave_sync();
    GE_STAMP(5);
    if (path_like) { src = c.perm[0]; dest = c.perm[1]; }  // multicast: perm[0] = 0 (multicast_routing.py:97)
    if (t == GE_PERISHABLE_DELIVERY) {
      if (lane < P.n_dests) { c.perm[lane] = ppd_pk[lane]; c.perm[P.n_dests + lane] = ppd_dp[lane]; }
      ge_wave_sync();
    }
  } else {
    for (int idx = lane; idx < E; idx += GE_WAVE) {
      int u = ge_row_of(P, c, idx), v = c.colw[idx] >> 4;
      c.wsort[ge_sorted_pos_p(P, c, u, v)] = (uint8_t)(c.colw[idx] & 15);
    }
    ge_wave_sync();
    if (path_like_t || t == GE_DISTRIBUTION_CENTER || t == GE_PERISHABLE_DELIVERY) {
      for (int k = lane; k < T; k += GE_WAVE) c.perm[k] = inj.terminals[(int64_t)env * T + k];
      ge_wave_sync();
      if (path_like_t) { src = c.perm[0]; dest = c.perm[1]; }
    }
  }
  if (t == GE_TSP) { src = 0; dest = -1; }

  GE_STAMP(6);
  // ------------------------------------------------------------------ baselines (is_eval_env)
  double heuristic = 0.0;
  if (!run.inject && P.is_eval) {
    const double kNaN = __builtin_nan("");
    if (t == GE_SHORTEST_PATH || t == GE_LONGEST_PATH || (t == GE_STEINER_TREE && P.n_dests == 1)) {
      ge_dijkstra_wave(c, n, src, lane);
      double d = c.sigma[dest];
      heuristic = (t == GE_LONGEST_PATH) ? -d : d;
      ge_wave_sync();
    } else if (t == GE_STEINER_TREE && P.n_dests == n - 1) {
      heuristic = ge_mst_total_wave(P, c, env, lane);  // steiner_tree.py:80-81
    } else if (t == GE_DENSEST_SUBGRAPH) heuristic = -1.0;             // densest_subgraph.py:85-88
    else if (t == GE_MAX_INDEPENDENT_SET) heuristic = P.weighted ? -1.0 : ge_greedy_mis_wave(P, c, lane);  // replaced by ge_k_mis_baseline (clique removal, exactly); this greedy value only stays if its work space were too small
    else if (t == GE_TSP) { const double mst = ge_mst_total_wave(P, c, env, lane); heuristic = mst + mst; }  // own double-tree walk in place of Christofides
    else if (t == GE_STEINER_TREE) heuristic = ge_kou_steiner_wave(P, c, lane);  // own 2-approximation in place of networkx's Kou
    else if (t == GE_MULTICAST_ROUTING) heuristic = 0.0;                 // computed below, after the delay bound
    else if (t == GE_DISTRIBUTION_CENTER) heuristic = -1.0;              // distribution_center.py:91
    else if (t == GE_PERISHABLE_DELIVERY) heuristic = 0.0;               // computed below
    else heuristic = kNaN;
  }

  // ------------------------------------------------------------------ multicast: delay bound (multicast_routing.py:101-106)
  double max_distance = 0.0;
  if (!run.inject && t == GE_MULTICAST_ROUTING) {
    ge_dijkstra_wave(c, n, 0, lane);
    double fn = -__builtin_inf(), ft = -__builtin_inf();  // farthest node, farthest destination
    for (int v = lane; v < n; v += GE_WAVE) { const double d = c.sigma[v]; if (d > fn) fn = d; }
    for (int k = 1 + lane; k <= P.n_dests; k += GE_WAVE) { const double d = c.sigma[c.perm[k]]; if (d > ft) ft = d; }
    for (int off = 32; off >= 1; off >>= 1) {
      const double a = ge_shfl_f64(fn, lane ^ off), b = ge_shfl_f64(ft, lane ^ off);
      if (a > fn) fn = a;
      if (b > ft) ft = b;
    }
    max_distance = *(const double *)c.misc * (fn - ft) + ft;  // np.random.rand() * (farthest_node - farthest_target) + farthest_target
    ge_wave_sync();
    if (P.is_eval) heuristic = ge_multicast_baseline(P, c, env, lane);
  }

  if (t == GE_DISTRIBUTION_CENTER) heuristic = -1.0;  // distribution_center.py:91, eval or not
  if (!run.inject && t == GE_PERISHABLE_DELIVERY) {
    heuristic = 0.0;
    if (P.is_eval) {  // perishable_product_delivery.py:147-154: curr_node stays the head; total += a; total += b
      double d0[5];
      ge_dijkstra_wave(c, n, 0, lane);
      for (int i = 0; i < P.n_dests; i++) d0[i] = c.sigma[c.perm[i]];
      ge_wave_sync();
      for (int i = 0; i < P.n_dests; i++) {
        ge_dijkstra_wave(c, n, c.perm[i], lane);
        heuristic += d0[i];
        heuristic += c.sigma[c.perm[P.n_dests + i]];
        ge_wave_sync();
      }
    }
  }
  GE_STAMP(9);
  // ------------------------------------------------------------------ write the slot to HBM
  const ge_buffers &G = P.buf;
  uint64_t *tbits = c.bits + 3 * W;  // target set
  if (lane < W) {
    uint64_t tb = 0;
    if (t == GE_SHORTEST_PATH || t == GE_LONGEST_PATH) { if ((dest >> 6) == lane) tb = 1ull << (dest & 63); }
    else if (t == GE_STEINER_TREE || t == GE_MULTICAST_ROUTING) for (int k = 1; k <= P.n_dests; k++) { int dk = c.perm[k]; if ((dk >> 6) == lane) tb |= 1ull << (dk & 63); }
    else if (t == GE_DISTRIBUTION_CENTER) for (int k = 0; k < P.n_dests; k++) { int dk = c.perm[k]; if ((dk >> 6) == lane) tb |= 1ull << (dk & 63); }
    tbits[lane] = tb;
  }
  ge_sync();  // the slot is complete in LDS: the second wave starts on its half of the edge slabs (ge_write_edge_slabs)
  if (run.inject) {
    for (int idx = lane; idx < n * F; idx += GE_WAVE) G.x[nbase * F + idx] = inj.x[nbase * F + idx];
  } else {  // flag columns only; the five structural columns are written by the features kernel
    const int nf = P.nflag;
    for (int idx = lane; idx < n * nf; idx += GE_WAVE) {
      int v = idx / nf, col = idx % nf; float val = 0.f;
      bool is_t = (tbits[v >> 6] >> (v & 63)) & 1ull;
      if (t == GE_SHORTEST_PATH || t == GE_STEINER_TREE) val = (col == 0) ? (v == src ? 1.f : 0.f) : (is_t ? 1.f : 0.f);
      else if (t == GE_LONGEST_PATH) val = (col == 0) ? (v == src ? 1.f : 0.f) : (is_t ? 1.f : ((P.parenting == 0 && v == src) ? 2.f : 0.f));
      else if (t == GE_TSP) val = (col == 1 && v == 0) ? 1.f : ((P.spatial && col >= 2) ? (float)((const double *)(c.wm + 4 * n))[2 * v + (col - 2)] : 0.f);
      else if (t == GE_MAX_INDEPENDENT_SET) val = (col == 0) ? (float)ge_wlut(c.fill[v]) : 0.f;
      else if (t == GE_MULTICAST_ROUTING)  // HAS_MSG, IS_TARGET, MAX_DISTANCE, DISTANCE_FROM_SOURCE (multicast_routing.py:124-129)
        val = (col == 0) ? (v == src ? 1.f : 0.f) : (col == 1) ? (is_t ? 1.f : 0.f) : (col == 2) ? (float)max_distance : (v == src ? 0.f : -1.f);
      else if (t == GE_PERISHABLE_DELIVERY) {  // IS_HEAD, HAS_P[5], NEEDS_P[5], TIME_LEFT_P[5] (perishable_product_delivery.py:118-131)
        const int pi = (col - 1) % 5;
        if (col == 0) val = (v == 0) ? 1.f : 0.f;
        else if (pi >= P.n_dests) val = 0.f;
        else if (col <= 5) val = (v == c.perm[pi]) ? 1.f : 0.f;
        else if (col <= 10) val = (v == c.perm[P.n_dests + pi]) ? 1.f : 0.f;
        else val = (float)ppd_dt;
      }
      else if (t == GE_DISTRIBUTION_CENTER)  // WEIGHT, IS_TAKEN, IS_TARGET, IS_COVERED, MAX_DISTANCE (distribution_center.py:99-102)
        val = (col == 0) ? (float)((const uint8_t *)c.wm)[P.cost_off + v] : (col == 2) ? (is_t ? 1.f : 0.f) : (col == 4) ? (float)P.max_distance : 0.f;
      G.x[(nbase + v) * F + col] = val;
    }
  }
  ge_write_edge_slabs<ENV>(P, c, env, tid, 2 * GE_WAVE);  // (this wave's half)
  // first mask (reset() -> info['mask'])
  const int A = P.A, AW = P.AW;
  const bool node_started = path_like_t;
  uint64_t *prune = c.bits + 4 * W;  // parenting >= 2: nodes that stay selectable
  if (lane < W) prune[lane] = ~0ull;
  ge_wave_sync();
  if (t == GE_LONGEST_PATH && P.parenting >= 2) {  // longest_path.py:134-143: has_path(alt_G, k, dest), alt_G = G - src
    uint64_t *rem = c.bits + 5 * W;
    if (lane < W) rem[lane] = ((src >> 6) == lane) ? (1ull << (src & 63)) : 0ull;
    ge_wave_sync();
    ge_reach_wave(c, n, W, dest, rem, lane);
    if (lane < W) {
      uint64_t keep = c.bits[W + lane] & ~rem[lane];
      if (P.parenting == 3 && n - 1 <= n / 3) keep = ~rem[lane];  // never true for n >= 2; kept for fidelity
      prune[lane] = keep;
    }
    ge_wave_sync();
  } else if (t == GE_TSP && P.parenting >= 2) {  // tsp.py:181-194: drop v if alt_G - v is disconnected, alt_G = G - start
    uint64_t *rem = c.bits + 5 * W;
    for (int v = 1; v < n; v++) {
      if (!((c.abits[0 * W + (v >> 6)] >> (v & 63)) & 1ull)) continue;  // uniform: candidates are N(start)
      if (n - 2 == 0) break;
      if (lane < W) rem[lane] = (lane == 0 ? 1ull : 0ull) | (((v >> 6) == lane) ? (1ull << (v & 63)) : 0ull);
      ge_wave_sync();
      int from = (v == 1) ? 2 : 1;
      int reached = ge_reach_wave(c, n, W, from, rem, lane);
      if (reached != n - 2 && lane == (v >> 6)) prune[lane] &= ~(1ull << (v & 63));
      ge_wave_sync();
    }
  }
  GE_STAMP_B0(7);
  if (t == GE_DISTRIBUTION_CENTER) {
    // distribution_center.py:25-26,117-118: nodes within max_distance of every node, each from that node as the source
    // (float sums depend on the direction); the first mask is the union over the targets
    uint64_t *acc = c.bits + 5 * W;
    if (lane < W) acc[lane] = 0ull;
    ge_wave_sync();
    if (n <= GE_WAVE) {
      // one LANE per source: every lane runs its own label-correcting search (a stack of nodes to relax, LIFO; the least
      // fixpoint does not depend on the order) over distances S[node][lane] in LDS -- a lane only ever touches its own
      // column, i.e. its own bank pair, so the searches never conflict.  Several times fewer instructions than 64 wave-wide
      // Bellman-Ford runs; GE_DC_LANES sources at a time keep the columns at 16 KB of LDS.
      // Only the rows the mask needs now -- the targets' (parenting 2) -- are computed here; the row of a chosen centre is
      // computed when it is chosen (ge_k_dc_range, in front of the step kernel), as the reference does (distribution_center.py:
      // 117-118, 155).  cur_rec[2 * env] records which rows of range_bits exist.
      double *S = (double *)(ge_dyn_smem() + P.lds.dcs);
      uint8_t *stk = (uint8_t *)(S + n * GE_DC_LANES);
      const uint64_t want = (P.parenting == 2) ? tbits[0] : 0ull;
      uint64_t mine = 0;
      for (uint64_t left = want; left;) {  // GE_DC_LANES sources at a time, lane k takes the k-th wanted node
        int s = -1;
        { uint64_t l2 = left; for (int k = 0; k < GE_DC_LANES && l2; k++) { const int b = ge_ctz64(l2); l2 &= l2 - 1; if (k == lane) s = b; } }
        uint64_t taken_now = 0; { uint64_t l2 = left; for (int k = 0; k < GE_DC_LANES && l2; k++) { taken_now |= l2 & (~l2 + 1); l2 &= l2 - 1; } }
        left &= ~taken_now;
        if (s >= 0) {
          const uint64_t reached = ge_dc_search(P.max_distance, n, s, c.rowptr, c.colw, S + lane, GE_DC_LANES, stk + lane, GE_DC_LANES);
          G.range_bits[nbase + s] = reached;
          mine |= reached;
        }
        ge_wave_sync();
      }
      for (int off = 32; off >= 1; off >>= 1) mine |= ge_shfl_u64(mine, lane ^ off);
      if (lane == 0) G.aux_bits[env] = want;
      if (lane == 0) acc[0] = mine;
      ge_wave_sync();
    } else
    for (int s = 0; s < n; s++) {
      ge_dijkstra_wave(c, n, s, lane, P.max_distance);
      const bool s_is_target = (tbits[s >> 6] >> (s & 63)) & 1ull;
      for (int w0 = 0; w0 < W; w0++) {
        const int v = w0 * GE_WAVE + lane;
        const uint64_t word = ge_ballot(v < n && c.sigma[v] <= P.max_distance);
        if (lane == 0) { G.range_bits[(nbase + s) * W + w0] = word; if (s_is_target) acc[w0] |= word; }
      }
      ge_wave_sync();
    }
  }
  GE_STAMP_B0(8);
  for (int w = lane; w < AW; w += GE_WAVE) {
    uint64_t mb;
    int lo = w * 64, hi = lo + 64; if (hi > A) hi = A;
    uint64_t full = (hi - lo == 64) ? ~0ull : ((1ull << (hi - lo)) - 1ull);
    if (t == GE_SHORTEST_PATH || (t == GE_LONGEST_PATH && P.parenting != 0) || t == GE_TSP) mb = c.abits[src * W + w] & prune[w];
    else if (t == GE_PERISHABLE_DELIVERY) {  // neighbours of the head, and the head itself if a product waits there (:176-181)
      mb = c.abits[0 * W + w];
      if (w == 0) for (int i = 0; i < P.n_dests; i++) if (c.perm[i] == 0) mb |= 1ull;
    }
    else if (t == GE_STEINER_TREE || (t == GE_MULTICAST_ROUTING && P.parenting >= 2)) {  // edges leaving src: steiner_tree.py:116-120;
      // multicast_routing.py:155-186: parenting 2 the same; parenting >= 3 keeps, per outside node, its best tree edge, and
      // with only the source in the tree that is the one edge from the source
      int a = c.rowptr[src], b = c.rowptr[src + 1]; mb = 0;
      int l2 = a > lo ? a : lo, h2 = b < hi ? b : hi;
      if (h2 > l2) mb = ((h2 - l2 == 64) ? ~0ull : ((1ull << (h2 - l2)) - 1ull)) << (l2 - lo);
    } else if (t == GE_DISTRIBUTION_CENTER && P.parenting == 2) mb = (c.bits + 5 * W)[w];
    else mb = full;  // LP parenting 0, Densest first step, MIS, DistributionCenter parenting 1
    G.mask_bits[(int64_t)env * AW + w] = mb;
    ((uint64_t *)c.sigma)[w] = mb;  // staged for the byte expansion below (sigma is free now)
  }
  ge_wave_sync();
  for (int idx = lane; idx < A; idx += GE_WAVE) G.mask[(int64_t)env * A + idx] = (uint8_t)((((uint64_t *)c.sigma)[idx >> 6] >> (idx & 63)) & 1ull);
  for (int w = lane; w < W; w += GE_WAVE) {
    uint64_t nb = 0;
    if (node_started && (src >> 6) == w) nb = 1ull << (src & 63);
    G.node_bits[(int64_t)env * W + w] = nb;
    G.target_bits[(int64_t)env * W + w] = tbits[w];
    if (G.cover_bits) G.cover_bits[(int64_t)env * W + w] = 0ull;
  }
  if (G.node_aux) {  // multicast parenting >= 3: the selectable edge into every node
    for (int v = lane; v < n; v += GE_WAVE) G.node_aux[nbase + v] = -1;
    ge_wave_sync();
    for (int k = c.rowptr[src] + lane; k < c.rowptr[src + 1]; k += GE_WAVE) G.node_aux[nbase + (c.colw[k] >> 4)] = k;
  }
  for (int k = lane; k < T; k += GE_WAVE) G.terminals[(int64_t)env * T + k] = (t == GE_TSP) ? 0 : ((t == GE_DENSEST_SUBGRAPH || t == GE_MAX_INDEPENDENT_SET) ? -1 : ((t == GE_DISTRIBUTION_CENTER && k >= P.n_dests) ? -1 : c.perm[k]));
  if (lane == 0) {
    const uint64_t head16 = (t == GE_DENSEST_SUBGRAPH || t == GE_MAX_INDEPENDENT_SET || t == GE_DISTRIBUTION_CENTER) ? GE_REC_HEAD_MASK : (uint64_t)src;
    uint64_t status = (queued && P.autoreset == 2) ? 3ull : 0ull;  // 3: regenerated at the start of this ge_step (next-step mode)
    if (gen_failed) status = 4ull;
    const uint64_t aux = ((t == GE_SHORTEST_PATH || t == GE_LONGEST_PATH) && W == 1) ? (uint64_t)dest : 0ull;
    // transitions since the last full reset survive a regeneration (the device policy is keyed by them)
    const uint64_t tstep = queued ? (G.slot_rec[2 * (int64_t)env + 1] >> GE_REC_TSTEP_SHIFT) : 0ull;
    G.slot_rec[2 * (int64_t)env] = 0ull;  // cost = +0.0
    G.slot_rec[2 * (int64_t)env + 1] = head16 | (status << GE_REC_STATUS_SHIFT) | (aux << GE_REC_AUX_SHIFT) | (tstep << GE_REC_TSTEP_SHIFT);
    G.counters[env * 2] = 0; G.counters[env * 2 + 1] = 0;
    if (queued) G.final_heur[env] = G.heuristic[env];  // of the episode that just ended (same-step autoreset reads it after this kernel)
    G.heuristic[env] = heuristic;
    if (run.items != GE_ITEMS_QUEUE) { G.seed[env] = seed; G.episode[env] = episode; }  // (a queue launch: the feature kernel advances them, or -- refill -- the swap)
    if (t == GE_PERISHABLE_DELIVERY && !run.inject) G.target_bits[(int64_t)env * W] = ge_f64_as_u64(ppd_dt);  // info['time_left'] of reset() (perishable_product_delivery.py:158) as float64 bits: this env has no target set
  }
  ge_sync();
  GE_STAMP(10);
}

// Queue mode: every step workgroup left (count, segment) in reset_count / reset_list; each reset workgroup
// rebuilds the exclusive prefix of the counts in LDS and finds its slot by binary search.
GE_DEV int ge_queue_prefix_of(const int32_t *rc, int nblk, int *pre, int lane) {  // one wave; no barrier
  int carry = 0;
  for (int k0 = 0; k0 < nblk; k0 += GE_WAVE) {
    int k = k0 + lane; int cnt = k < nblk ? rc[k] : 0;
    int incl = ge_wave_incl_scan(cnt, lane);
    if (k < nblk) pre[k] = carry + incl - cnt;
    carry += ge_shfl_i32(incl, GE_WAVE - 1);
  }
  if (lane == 0) pre[nblk] = carry;
  return carry;
}
GE_DEV int ge_queue_prefix_wave(const GeParams &P, int *pre, int lane) { return ge_queue_prefix_of(P.buf.reset_count, (P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK, pre, lane); }
GE_DEV int ge_queue_slot_of(const int32_t *list, int nblk, const int *pre, int q) {
  int lo = 0, hi = nblk;
  while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (pre[mid] <= q) lo = mid; else hi = mid; }
  return list[lo * GE_STEP_BLOCK + (q - pre[lo])];
}
GE_DEV int ge_queue_slot(const GeParams &P, const int *pre, int q) { return ge_queue_slot_of(P.buf.reset_list, (P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK, pre, q); }


// ---------------------------------------------------------------------------------------------------------------
// MT19937 pre-seeding, GE_SEED_DEPTH - 1 episodes ahead.  Seeding is a chain of 1 246 + 623 dependent steps per slot; one LANE
// runs the chain of one slot entirely in registers (the pass-1 words that init_by_array's second pass consumes are recomputed
// in step with it instead of being stored and read back), so a wave seeds 64 slots side by side.  Every 16 words the wave
// transposes its [16 words][64 slots] tile through LDS and writes it as 64-byte pieces, four slots per store instruction:
// the states leave as full sectors instead of 1 869 scattered 4-byte stores per slot (which kept a whole launch busy for
// ~190 us and slowed whatever ran beside it).  A seeding workgroup is two waves: wave 0 the python stream (random.seed(int) =
// init_by_array([s])), wave 1 the numpy stream (np.random.seed(int) = init_genrand(s)) of the same 64 slots.
// The ring: the generator states of episode e sit in entry e mod GE_SEED_DEPTH.  While a slot runs episode e, entries (e+1) and
// (e+2) are valid and entry e is free; the launch that moves it to e+1 reads entry (e+1) and refills entry e with episode e+3.

// init_genrand(19650218): the constant sequence init_by_array starts from, evaluated at compile time (the chain that produces
// it is as long as the one it feeds: as a table it costs one scalar load per step instead of three vector operations)
struct GeMtInit { uint32_t v[GE_MT_N]; };
constexpr GeMtInit ge_make_mt_init() {
  GeMtInit t{};
  uint32_t b = 19650218u;
  t.v[0] = b;
  for (int i = 1; i < GE_MT_N; i++) { b = 1812433253u * (b ^ (b >> 30)) + (uint32_t)i; t.v[i] = b; }
  return t;
}
GE_CONSTANT GeMtInit ge_mt_init = ge_make_mt_init();

// per-lane destinations of a tile flush: lane (sub, w) = (lane >> 4, lane & 15) writes word w of the slots 4 r + sub, r = 0..15.
// cb[r] = index of the slot's first 16-word piece in mt_state (one register per slot instead of a 64-bit pointer), 0xffffffff = none.
struct GeSeedDst { uint32_t cb[16]; };
GE_DEV GeSeedDst ge_seed_dst(const uint32_t *base, int stream, int lane) {
  GeSeedDst d;
  const int sub = lane >> 4;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const uint32_t sb = base[4 * r + sub];
    d.cb[r] = (sb != 0xffffffffu) ? (sb * 2u + (uint32_t)stream) * (uint32_t)(GE_MT_N / 16) : 0xffffffffu;
  }
  return d;
}
// write the tile (words [16 c, 16 c + 16) of 64 slots) to the slots' state arrays as 64-byte pieces.  Collective over the wave.
GE_DEV void ge_seed_flush(const GeParams &P, const uint32_t *tile, const GeSeedDst &d, int c, int lane) {
  ge_wave_sync();
  const uint32_t *src = tile + (lane & 15) * GE_SEED_TILE_STRIDE + (lane >> 4);
  uint32_t *out = P.buf.mt_state + (lane & 15);
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const uint32_t v = src[4 * r];
    if (d.cb[r] != 0xffffffffu) out[(int64_t)(d.cb[r] + (uint32_t)c) * 16] = v;
  }
  ge_wave_sync();
}

// one seeding workgroup (two waves) over 64 (slot, ring entry, seed) items: `sb` = slot * GE_SEED_DEPTH + entry of this lane's
// item or 0xffffffff.  `lds` = GE_SEED_LDS_BYTES of scratch.  Collective over the workgroup (wave-level syncs only).
GE_DEV void ge_seed_group(const GeParams &P, unsigned char *lds, uint32_t sb, uint32_t seed, int tid) {
  const int lane = tid & (GE_WAVE - 1), wv = tid >> 6;
  uint32_t *tile = (uint32_t *)lds + wv * 16 * GE_SEED_TILE_STRIDE;
  uint32_t *base = (uint32_t *)lds + 2 * 16 * GE_SEED_TILE_STRIDE;  // shared by both waves: same items
  if (wv == 0) base[lane] = sb;
  ge_sync();
  const GeSeedDst dst = ge_seed_dst(base, wv, lane);
  uint32_t *mine = tile + lane;
  if (wv == 0) {  // python: init_by_array([seed]) ([py] _randommodule.c)
    uint32_t prev = 19650218u, first = 0u;
    for (int i = 1; i < GE_MT_N; i++) {  // pass 1, to its end: only its last word and mt[1] are needed to start pass 2
      prev = (ge_mt_init.v[i] ^ ((prev ^ (prev >> 30)) * 1664525u)) + seed;  // + key[0] + j, j == 0
      if (i == 1) first = prev;
    }
    const uint32_t m1 = (first ^ ((prev ^ (prev >> 30)) * 1664525u)) + seed;  // i wrapped: the 624th iteration lands on mt[1]
    prev = m1;
    uint32_t p1 = first;  // pass 1 again, in step with pass 2 (its words are consumed once, in order: nothing is stored)
    for (int c = 0; c < GE_MT_N / 16; c++) {
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int i = 16 * c + k;
        uint32_t out;
        if (i == 0) out = 0x80000000u;   // mt[0]
        else if (i == 1) out = 0u;       // mt[1] is the last word to be known: patched below
        else {
          p1 = (ge_mt_init.v[i] ^ ((p1 ^ (p1 >> 30)) * 1664525u)) + seed;          // pass-1 value of mt[i]
          prev = (p1 ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;       // final mt[i]
          out = prev;
        }
        mine[k * GE_SEED_TILE_STRIDE] = out;
      }
      ge_seed_flush(P, tile, dst, c, lane);
    }
    if (sb != 0xffffffffu) P.buf.mt_state[((int64_t)sb * 2 + 0) * GE_MT_N + 1] = (m1 ^ ((prev ^ (prev >> 30)) * 1566083941u)) - 1u;
  } else {        // numpy: init_genrand(seed) ([np] mt19937.c)
    uint32_t prev = seed;
    for (int c = 0; c < GE_MT_N / 16; c++) {
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int i = 16 * c + k;
        if (i > 0) prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i;
        mine[k * GE_SEED_TILE_STRIDE] = prev;
      }
      ge_seed_flush(P, tile, dst, c, lane);
    }
  }
  ge_sync();  // the scratch may be reused
}

// Full reset / seeded injection: ring entries jlo .. GE_SEED_DEPTH - 1 of every slot, entry j from seeds[slot] + j * seed_stride
// (episode j of a slot whose first episode is seeded seeds[slot]).  128-thread workgroups, 64 items each.
GE_KERNEL ge_k_seed(GeParams P, const uint32_t *seeds, int jlo) {
  const int tid = ge_tid();
  const int64_t items = (int64_t)(GE_SEED_DEPTH - jlo) * P.B;
  for (int64_t g0 = (int64_t)ge_bid() * GE_WAVE; g0 < items; g0 += (int64_t)ge_gdim() * GE_WAVE) {
    const int64_t g = g0 + (ge_tid_fresh() & (GE_WAVE - 1));
    uint32_t sb = 0xffffffffu, seed = 0u;
    if (g < items) {
      const int j = jlo + (int)(g / P.B), env = (int)(g % P.B);
      sb = (uint32_t)env * GE_SEED_DEPTH + (uint32_t)j;
      seed = seeds[env] + (uint32_t)j * (uint32_t)P.seed_stride;
    }
    ge_seed_group(P, ge_dyn_smem(), sb, seed, tid);
  }
}

// ge_reset_continue: every slot leaves its episode e without the queued regeneration whose seeding workgroups would have
// refilled ring entry e mod GE_SEED_DEPTH with the states of episode e + GE_SEED_DEPTH; this launch (in front of the graph kernel,
// while seed[] / episode[] still name episode e) does it for the whole batch.
GE_KERNEL ge_k_seed_next(GeParams P) {
  const int tid = ge_tid();
  for (int64_t g0 = (int64_t)ge_bid() * GE_WAVE; g0 < P.B; g0 += (int64_t)ge_gdim() * GE_WAVE) {
    const int64_t g = g0 + (ge_tid_fresh() & (GE_WAVE - 1));
    uint32_t sb = 0xffffffffu, seed = 0u;
    if (g < P.B) {
      sb = (uint32_t)g * GE_SEED_DEPTH + (uint32_t)(P.buf.episode[g] % GE_SEED_DEPTH);
      seed = P.buf.seed[g] + (uint32_t)GE_SEED_DEPTH * (uint32_t)P.seed_stride;
    }
    ge_seed_group(P, ge_dyn_smem(), sb, seed, tid);
  }
}

// the feature kernel (the last kernel of a queued regeneration) moves the slot's bookkeeping to the new episode
GE_DEV void ge_advance_episode(const GeParams &P, int env) {
  P.buf.seed[env] = P.buf.seed[env] + (uint32_t)P.seed_stride;
  P.buf.episode[env] = P.buf.episode[env] + 1;
}
// ... or, when the launch refilled the slot's spare image, marks the image valid (the slot keeps running its episode; the swap
// that moves the image in advances seed[] / episode[])
GE_DEV void ge_finish_item(const GeParams &P, const GeRun &run, int env) {
  if (run.refill) P.spare_state[env] = 1; else ge_advance_episode(P, env);
}

// Queue launches: the first `nseed` workgroups are seeding workgroups (64 queued slots each): they write the generator states
// of a later episode of the slot (run.seed_ahead episodes past the one seed[] / episode[] name) into the ring entry that episode
// will be read from, while the other workgroups regenerate the queued slots from the entries seeded long ago -- no second
// stream, no event.  Without spares seed_ahead = GE_SEED_DEPTH: the entry of the episode that just ended is refilled.  With
// spares every regeneration runs one episode earlier relative to its use, and seed_ahead = 2 ("Episode prefetch").
// RAGGED (multi-class engine): P is the engine-wide parameter block (B = all slots; queue, seed[], episode[], mt_state are global
// arrays in slot order) and every workgroup looks up the class of its slot: R.classes[class] is that class's uniform sub-engine.
template <int ENV, bool RAGGED>
GE_KERNEL_LB(GE_RESET_THREADS, GE_RESET_WAVES_PER_SIMD) ge_k_reset(GeParams P, GeRagged R, const uint32_t *seeds, GeRun run, GeInject inj, int nseed, int bucket) {
  int *pre = (int *)(ge_dyn_smem() + P.lds.pre);  // overlays the MT19937 scratch: rebuilt before every lookup
  if (ge_bid() == 0 && ge_tid() == 0) P.buf.work_count[0] = 0;  // fallback list of the feature fast path
  const int nblk = (P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK;
  const bool queue = run.items == GE_ITEMS_QUEUE;
  int count = P.B;
  if (queue) {
    if (ge_tid() < GE_WAVE) ge_queue_prefix_wave(P, pre, ge_tid());
    ge_sync();
    count = pre[nblk];
  }
  if (queue && ge_bid() < nseed) {
    for (int g0 = ge_bid() * GE_WAVE; g0 < count; g0 += nseed * GE_WAVE) {
      const int tid = ge_tid_fresh();
      if (g0 != ge_bid() * GE_WAVE) { if (tid < GE_WAVE) ge_queue_prefix_wave(P, pre, tid); ge_sync(); }
      const int q = g0 + (tid & (GE_WAVE - 1));
      uint32_t sb = 0xffffffffu, seed = 0u;
      if (q < count) {
        const int env = ge_queue_slot(P, pre, q);
        const int64_t ep = P.buf.episode[env] + run.seed_ahead;  // the episode whose states are written now
        sb = (uint32_t)env * GE_SEED_DEPTH + (uint32_t)(ep % GE_SEED_DEPTH);
        seed = P.buf.seed[env] + (uint32_t)run.seed_ahead * (uint32_t)P.seed_stride;
      }
      ge_sync();  // every lane has its item before the scratch (which holds the queue prefix) is reused
      ge_seed_group(P, ge_dyn_smem(), sb, seed, tid);
    }
    return;
  }
  const int first = queue ? nseed : 0, stride = ge_gdim() - first;
  for (int q = ge_bid() - first; q < count; q += stride) {
    int env = q;
    if (queue) {
      if (q != ge_bid() - first) { const int t2 = ge_tid_fresh(); if (t2 < GE_WAVE) ge_queue_prefix_wave(P, pre, t2); ge_sync(); }
      env = ge_queue_slot(P, pre, q);
      ge_sync();  // every thread has its slot before the scratch is reused
    }
    if constexpr (RAGGED) {
      const int cls = (int)ge_uniform_u32((uint32_t)R.slot_class[env]);
      // one launch per LDS bucket of size classes (bucket >= 0): the dynamic LDS of this launch -- hence the workgroups a CU holds --
      // is sized for the bucket's largest class, not the engine's; the slots of other buckets are left to their launch
      if (bucket >= 0 && R.classes[cls].bucket != bucket) continue;  // (uniform over the workgroup)
      const int lo = R.class_start[cls];
      ge_reset_env<ENV>(R.classes[cls], env - lo, seeds ? seeds + lo : seeds, run, inj);
    } else {
      ge_reset_env<ENV>(P, env, seeds, run, inj);
    }
  }
}
