// Structural-feature kernels: the five columns of feature_extraction.py:6-37 for freshly generated slots.
//   ge_features_generic_env : any n; graph staged in LDS, level-synchronous Brandes, one source at a time.
//   ge_features64_env       : n <= 64; one LANE per BFS source -- every lane walks its own shortest-path DAG
//                             with 64-bit set arithmetic (ctz over frontier words), sigma/delta live in LDS
//                             columns [node][lane], so the 64 sources advance together without barriers.
// Both produce float64 results in the reference's operation order where the order is observable, then round to
// float32 exactly once (sf = torch.tensor(sf)).
#pragma once
#include "ge_params.h"
#include "ge_platform.h"
#include "ge_reset.h"

struct GeFctx {
  uint64_t *abits; int *rowptr; uint16_t *colw; uint16_t *scw;
  uint64_t *sets;                               // per wave: visited set / level being discovered (W words each)
  double *sigma, *delta, *coeff, *bcw;          // per-wave scratch of the Brandes pass (this wave's area)
  double *bcw0; int wave_f64;                   // bcw of wave 0 and the float64 stride between waves (partial sums are combined in wave order)
  double *bc, *clos;                            // common
  double *prx, *prn, *sinv, *diff, *clus;       // node role
  uint16_t *ord, *lvl;                          // per wave: BFS order of the current source; lvl[d] = where level d starts in it
};

GE_DEV GeFctx ge_carve_f(const GeParams &P, int tid) {
  unsigned char *s = ge_dyn_smem();
  const GeLdsF &L = P.ldsf;
  const int n = P.n, an = ((n * 8 + 15) & ~15) / 8;
  GeFctx c;
  c.rowptr = (int *)(s + L.rowptr); c.colw = (uint16_t *)(s + L.colw);
  c.bc = (double *)(s + L.bc); c.clos = (double *)(s + L.clos);
  unsigned char *w = s + L.wave0 + (tid >> 6) * L.wave_stride;
  c.sets = (uint64_t *)w;
  // (P.W <= GE_BCW_REG_W: the wave's betweenness partial sums live in registers while it runs and are handed over in delta[])
  const int bcw_at = P.W <= GE_BCW_REG_W ? n : 3 * n;
  c.sigma = (double *)(w + L.w_sigma); c.delta = c.sigma + n; c.coeff = c.sigma + 2 * n; c.bcw = c.sigma + bcw_at;
  c.ord = (uint16_t *)(w + L.w_ord); c.lvl = (uint16_t *)(w + L.w_lvl);
  c.bcw0 = (double *)(s + L.wave0 + L.w_sigma) + bcw_at; c.wave_f64 = L.wave_stride / 8;
  c.abits = (uint64_t *)(s + L.abits); c.scw = (uint16_t *)(s + L.scw);
  c.prx = (double *)(s + L.prx); c.prn = c.prx + an; c.sinv = c.prx + 2 * an; c.diff = c.prx + 3 * an; c.clus = (double *)(s + L.clus);
  return c;
}

// Any n.  The workgroup has 1..8 waves (whatever fits LDS): the graph is staged once, then every wave runs the
// level-synchronous Brandes pass for its own sources (s = wave, wave + waves, ...) on private scratch; per-wave
// betweenness partial sums are combined in wave order, then clustering and pagerank.
// nparts > 1: the slot's BFS sources are dealt over nparts workgroups (part 0 .. nparts-1), and ONE MORE workgroup
// (part == nparts) does the node-level work -- clustering, pagerank, degrees -- beside them instead of behind one of them.
#ifndef GE_PR_CH
#define GE_PR_CH 4  // row entries of a pagerank trip (their loads in flight together)
#endif
#ifndef GE_FABL
#define GE_FABL 0  // diagnostic ablation bits of the generic feature kernel (tools/variant_reset.py; the results are wrong by construction): 1 forward
                   // push, 2 backward coefficient pass, 4 backward pull, 8 pagerank iterations; 0 when shipped
#endif
// Diagnostic build only (-DGE_STAMPS, never shipped; tools/feat_phase_clocks.py): shader-clock cycles wave 0 of slot 0's first Brandes
// workgroup spends in the phases of the generic kernel -- [k] = cycles from stamp k to the next stamp, summed over sources and levels:
// 0 level discovery, 1 path-count pull, 2 front update, 3 coefficients, 4 dependency pull, 5 end of a source, 6 set-up of a source;
// ge_stamp_buf[16 + k]; [24] = sources, [25] = levels
#if defined(GE_STAMPS) && !defined(GE_EMU)
#define GE_FSTAMP_DECL unsigned long long fs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fs_t = clock64(), fs_src = 0, fs_lev = 0; int fs_k = 7
#define GE_FSTAMP(k) do { const unsigned long long now_ = clock64(); fs_acc[fs_k] += now_ - fs_t; fs_t = now_; fs_k = (k); if ((k) == 6) fs_src++; if ((k) == 1) fs_lev++; } while (0)
#define GE_FSTAMP_OUT do { if (env == 0 && part == 0 && tid == 0) { for (int k_ = 0; k_ < 8; k_++) ge_stamp_buf[16 + k_] = fs_acc[k_]; ge_stamp_buf[24] = fs_src; ge_stamp_buf[25] = fs_lev; } } while (0)
#else
#define GE_FSTAMP_DECL do { } while (0)
#define GE_FSTAMP(k) do { } while (0)
#define GE_FSTAMP_OUT do { } while (0)
#endif
GE_HOSTDEV int ge_feat_workgroups(int feat_parts) { return feat_parts > 1 ? feat_parts + 1 : 1; }
// One pull of the Brandes pass over the nodes ord[k0 .. k1): a lane takes TWO nodes (k, k + 64) and walks both rows together, eight
// entries of each per trip -- the pass is a chain of LDS round trips (order list -> row bounds -> neighbour ids -> their values), and
// on one node with four entries per trip a level of 150 nodes cost three such chains one after the other (4 400 cycles per level
// at n = 256, tools/feat_phase_clocks.py).  DELTA false: sigma[v] = sum of front[u] (path counts: integers, exact in any order).
// DELTA true: delta[v] = sum of sigma[v] * coeff[w] IN ROW ORDER; an entry past the end of the row adds sigma[v] * 0.0 = +0.0.
template <bool DELTA, int U>
GE_DEV void ge_brandes_pull_u(const GeFctx &c, int k0, int k1, int lane) {
  constexpr int CH = 8;
  for (int kb = k0; kb < k1; kb += U * GE_WAVE) {
    int v[U], e[U], r1[U]; bool on[U]; double acc[U], sv[U];
#pragma unroll
    // every load below is UNCONDITIONAL, from an address that is valid whatever the lane holds (k0, node 0 / row entry 0 stand in),
    // and the value is selected afterwards: a load under a condition compiles to a branch around it with a wait behind it, one LDS
    // round trip per entry
    for (int u = 0; u < U; u++) { const int k = kb + u * GE_WAVE + lane; on[u] = k < k1; v[u] = (int)c.ord[on[u] ? k : k0]; }
#pragma unroll
    for (int u = 0; u < U; u++) { const int a = c.rowptr[v[u]], b = c.rowptr[v[u] + 1]; e[u] = on[u] ? a : 0; r1[u] = on[u] ? b : 0; sv[u] = DELTA ? c.sigma[v[u]] : 1.0; acc[u] = 0.0; }
    while (e[0] < r1[0] || (U > 1 && e[U - 1] < r1[U - 1])) {
      int w[U][CH]; double f[U][CH]; bool ok[U][CH];
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int j = 0; j < CH; j++) { ok[u][j] = e[u] + j < r1[u]; w[u][j] = (int)(c.colw[ok[u][j] ? e[u] + j : 0] >> 4); }
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int j = 0; j < CH; j++) { const double x = c.coeff[w[u][j]]; f[u][j] = ok[u][j] ? x : 0.0; }
#pragma unroll
      for (int u = 0; u < U; u++) {
#pragma unroll
        for (int j = 0; j < CH; j++) acc[u] += DELTA ? sv[u] * f[u][j] : f[u][j];
        e[u] += CH;
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) if (on[u]) (DELTA ? c.delta : c.sigma)[v[u]] = acc[u];
  }
}

// OR of the adjacency bit rows of the nodes ord[lo .. hi): lane = group * Wp + word, a group takes every NG-th node, eight rows per
// trip -- unconditional loads from addresses that are always valid, selected afterwards (see ge_brandes_pull_u) -- so the loads of
// a trip are in flight together.  The caller combines the groups.
GE_DEV uint64_t ge_level_rows_or(const uint16_t *ord, const uint64_t *rows, int lo, int hi, int gg, int gw, int W, int NG) {
  uint64_t un = 0;
  for (int k = lo + gg; k < hi; k += 8 * NG) {
    int u[8]; uint64_t r[8]; bool ok[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { ok[j] = k + j * NG < hi && gw < W; u[j] = (int)ord[ok[j] ? k + j * NG : lo]; }
#pragma unroll
    for (int j = 0; j < 8; j++) { const uint64_t x = rows[u[j] * W + (gw < W ? gw : 0)]; r[j] = ok[j] ? x : 0ull; }
    un |= ((r[0] | r[1]) | (r[2] | r[3])) | ((r[4] | r[5]) | (r[6] | r[7]));
  }
  return un;
}

// (a level of at most 64 nodes -- the first and the last levels of every search -- takes one row per lane: the kernel is bound by
// vector-instruction issue, and the second row of a lane costs its instructions whether or not it exists)
template <bool DELTA>
GE_DEV void ge_brandes_pull(const GeFctx &c, int k0, int k1, int lane) {
  if (k1 - k0 <= GE_WAVE) ge_brandes_pull_u<DELTA, 1>(c, k0, k1, lane); else ge_brandes_pull_u<DELTA, 2>(c, k0, k1, lane);
}

GE_DEV void ge_features_generic_env(const GeParams &P, int env, int part, int nparts) {
  const int tid = ge_tid_fresh(), nthreads = ge_bdim();
  const int lane = tid & (GE_WAVE - 1), wv = tid >> 6, nwaves = nthreads >> 6;
  const int n = P.n, W = P.W, E = P.E, F = P.F, t = P.env_type;
  const ge_buffers &G = P.buf;
  const int64_t nbase = (int64_t)env * n, ebase = (int64_t)env * E;
  GeFctx c = ge_carve_f(P, tid);
  // Two roles (GeLdsF): the Brandes workgroups (a share of the BFS sources each) and the node-level workgroup (clustering, pagerank,
  // degrees); with nparts == 1 one workgroup is both.  The node role's arrays -- adjacency bit rows, sorted edge copy, pagerank
  // vectors -- overlay the Brandes role's per-wave areas when the roles are different workgroups, so a Brandes workgroup holds only
  // what it uses (n = 256: 7 waves where 5 fitted, n = 512: 7 where 4 did) and reads the bit rows of its search from global memory
  // (8-32 KB per slot, shared by the slot's nine workgroups: cache hits).
  const bool node_part = nparts > 1 && part == nparts;  // this workgroup only does the node-level work
  const bool node_role = node_part || nparts == 1, brandes_role = !node_part;
  // the slot's adjacency bit rows in HBM.  (A copy in LDS for the searches, where it costs neither a wave nor a workgroup per CU, was
  // measured: the same cycles per source at n = 200 / 400 / 512, 26.6 -> 17 K cycles of discovery at n = 256, and config 5 lost a
  // fifth -- dropped.)
  const uint64_t *arows = G.adj_bits + nbase * W;
  // stage the slot's graph in LDS
  if (node_role) for (int i = tid; i < n * W; i += nthreads) c.abits[i] = arows[i];
  // complete graph on all n nodes: row i holds every other node in ascending order -- entry q of the row is node q (q < i) or q + 1 --
  // and only the weight codes are staged, a byte each (scode: ascending-neighbour order)
  const bool closed = P.complete && P.ng == n;
  uint8_t *c8 = (uint8_t *)c.colw;
  for (int v = tid; v <= n; v += nthreads) c.rowptr[v] = G.row_ptr[(int64_t)env * (n + 1) + v];
  if (closed) { for (int idx = tid; idx < E; idx += nthreads) c8[idx] = G.scode[ebase + idx]; }
  else for (int idx = tid; idx < E; idx += nthreads) c.colw[idx] = G.colw[ebase + idx];
  ge_sync();
  // rows in ascending-column order (scipy canonical CSR): position by rank in the bit row (complete graphs: scw IS colw)
  if (node_role && !P.complete) for (int v = tid; v < n; v += nthreads)
    for (int k = c.rowptr[v]; k < c.rowptr[v + 1]; k++) {
      uint16_t e = c.colw[k];
      c.scw[c.rowptr[v] + ge_rank_below(c.abits + v * W, e >> 4)] = e;
    }
  const bool bcw_reg = W <= GE_BCW_REG_W;  // the wave's betweenness partial sums in registers: node lane + 64 j in bcr[j]
  double bcr[GE_BCW_REG_W];
#pragma unroll
  for (int j = 0; j < GE_BCW_REG_W; j++) bcr[j] = 0.0;
  if (brandes_role && !bcw_reg) for (int v = lane; v < n; v += GE_WAVE) c.bcw[v] = 0.0;
  ge_sync();
  // Brandes betweenness + closeness: one level-synchronous BFS per source, sources dealt round-robin to the waves
  // complete graph on all n nodes (TSP config 3): every pair is adjacent, so no shortest path has an interior node
  // (betweenness is a sum of zeros) and every BFS has one level of n-1 nodes (closeness (n-1)/(n-1) * (n-1)/(n-1))
  const bool trivial = (P.complete && P.ng == n) || node_part;
  if (trivial && !node_part) for (int v = tid; v < n; v += nthreads) c.clos[v] = (((double)n - 1.0) / (double)(n - 1)) * (((double)n - 1.0) / (double)(n - 1));
  // One BFS per source, level by level, the nodes kept in discovery order (ord) with the start of every level (lvl): each pass
  // touches the nodes of ONE level and their rows -- O(n + E) per source.  No atomics and no per-node level array:
  //  * the next level is found with the adjacency BIT rows: the lanes OR the rows of the current level's nodes (groups of Wp
  //    lanes, one word each, combined with shuffles), minus the visited set;
  //  * path counts by pull: a node of the new level adds front[u] over its row, where front[] (kept in the coefficient array,
  //    which the forward pass does not need) holds sigma(u) for the nodes of the current level and 0 for every other node -- the
  //    counts are integers, exact in any order;
  //  * dependencies by pull, as always: a node of level lev - 1 adds sigma(v) * coeff(w) over its row IN ROW ORDER; coeff[] is 0
  //    outside level lev, so the test "is w one level deeper" is gone -- a term sigma(v) * 0.0 = +0.0 leaves the sum as it is, and
  //    every float64 sum keeps the order (and the value) it always had.
  const uint64_t below = (1ull << lane) - 1ull;
  uint64_t *vis = c.sets, *nxt = vis + W;  // visited set / level being discovered (this wave's words)
  int Wp = 1; while (Wp < W) Wp <<= 1;                  // lanes per group: lane = group * Wp + word
  const int NG = GE_WAVE / Wp, gw = lane & (Wp - 1), gg = lane / Wp;
  GE_FSTAMP_DECL;
  for (int s = part * nwaves + wv; s < n && !trivial; s += nwaves * nparts) {
    GE_FSTAMP(6);
    for (int v = lane; v < n; v += GE_WAVE) { c.sigma[v] = (v == s) ? 1.0 : 0.0; c.delta[v] = 0.0; c.coeff[v] = (v == s) ? 1.0 : 0.0; }
    if (lane < W) vis[lane] = ((s >> 6) == lane) ? (1ull << (s & 63)) : 0ull;
    if (lane == 0) { c.ord[0] = (uint16_t)s; c.lvl[0] = 0; c.lvl[1] = 1; }
    ge_wave_sync();
    int d = 0, reach = 1, lo = 0, hi = 1; int64_t tot = 0;
    for (;;) {
      GE_FSTAMP(0);
      uint64_t un = ge_level_rows_or(c.ord, arows, lo, hi, gg, gw, W, NG);
      for (int off = Wp; off < GE_WAVE; off <<= 1) un |= ge_shfl_u64(un, lane ^ off);
      if (lane < W) { const uint64_t nw = un & ~vis[lane]; vis[lane] |= nw; nxt[lane] = nw; }
      ge_wave_sync();
      int found = 0;
      for (int w0 = 0; w0 < W; w0 += 8) {  // eight words of the new level per trip (their reads in flight together)
        uint64_t b[8];
#pragma unroll
        for (int j = 0; j < 8; j++) b[j] = nxt[w0 + j < W ? w0 + j : w0];
#pragma unroll
        for (int j = 0; j < 8; j++) if (w0 + j < W) {  // (wave-uniform)
          if ((b[j] >> lane) & 1ull) c.ord[hi + found + ge_popc64(b[j] & below)] = (uint16_t)((w0 + j) * GE_WAVE + lane);
          found += ge_popc64(b[j]);
        }
      }
      if (!found) break;
      ge_wave_sync();
      GE_FSTAMP(1);
      if (!(GE_FABL & 1)) ge_brandes_pull<false>(c, hi, hi + found, lane);
      ge_wave_sync();
      GE_FSTAMP(2);
      for (int k = lo + lane; k < hi; k += GE_WAVE) c.coeff[c.ord[k]] = 0.0;                                     // the front moves on
      for (int k = hi + lane; k < hi + found; k += GE_WAVE) { const int v = c.ord[k]; c.coeff[v] = c.sigma[v]; }
      d++; lo = hi; hi += found; reach += found; tot += (int64_t)d * found;
      if (lane == 0) c.lvl[d + 1] = (uint16_t)hi;
      ge_wave_sync();
    }
    GE_FSTAMP(0);
    for (int k = lo + lane; k < hi; k += GE_WAVE) c.coeff[c.ord[k]] = 0.0;  // coeff[] is all zero again
    ge_wave_sync();
    for (int lev = d; lev >= 1; lev--) {  // backward accumulation, level by level
      GE_FSTAMP(3);
      const int l0 = c.lvl[lev], l1 = c.lvl[lev + 1], p0 = c.lvl[lev - 1];
      for (int k = l0 + lane; k < l1 && !(GE_FABL & 2); k += GE_WAVE) { const int v = c.ord[k]; c.coeff[v] = (1.0 + c.delta[v]) / c.sigma[v]; if (!bcw_reg) c.bcw[v] += c.delta[v]; }
      ge_wave_sync();
      GE_FSTAMP(4);
      if (!(GE_FABL & 4)) ge_brandes_pull<true>(c, p0, l0, lane);  // (+0.0 for a neighbour that is not one level deeper)
      ge_wave_sync();
      for (int k = l0 + lane; k < l1; k += GE_WAVE) c.coeff[c.ord[k]] = 0.0;
    }
    GE_FSTAMP(5);
    ge_wave_sync();
    if (bcw_reg) {  // betweenness[v] += delta_s(v), v != s, in source order (a node the search did not reach, or of the last level, adds +0.0)
#pragma unroll
      for (int j = 0; j < GE_BCW_REG_W; j++) { const int v = lane + GE_WAVE * j; if (j < W) bcr[j] += (v < n && v != s) ? c.delta[v] : 0.0; }
    }
    if (lane == 0) {  // closeness_centrality, wf_improved
      double cc = 0.0;
      if (tot > 0 && n > 1) { cc = ((double)reach - 1.0) / (double)tot; double sc = ((double)reach - 1.0) / (double)(n - 1); cc *= sc; }
      c.clos[s] = cc;
    }
    ge_wave_sync();
  }
  GE_FSTAMP(7);
  GE_FSTAMP_OUT;
  if (bcw_reg && brandes_role) {  // (this wave's delta[] is free now: every one of its searches is over)
#pragma unroll
    for (int j = 0; j < GE_BCW_REG_W; j++) { const int v = lane + GE_WAVE * j; if (j < W && v < n) c.bcw[v] = bcr[j]; }
  }
  ge_sync();
  // betweenness: per-wave partial sums (each in source order) added in wave order
  if (brandes_role) for (int v = tid; v < n; v += nthreads) { double acc = 0.0; for (int w = 0; w < nwaves; w++) acc += c.bcw0[w * c.wave_f64 + v]; c.bc[v] = acc; }
  ge_sync();
  if (nparts > 1) {
    // several workgroups share this slot's sources (few slots, many CUs): closeness of the own sources is final, the
    // betweenness partial goes to scratch and ge_k_feat_combine adds the parts in part order
    if (!node_part) {
      for (int v = tid; v < n; v += nthreads) G.feat_scratch[((int64_t)env * nparts + part) * n + v] = c.bc[v];
      for (int v = tid; v < n; v += nthreads) if ((v % (nwaves * nparts)) / nwaves == part) G.x[(nbase + v) * F + P.nflag + 2] = (float)c.clos[v];
      ge_sync();
      return;
    }
  }
  // node-level work, rows dealt to every thread of the workgroup (each node's sums keep their order; the iteration count and
  // the pairwise error sum are the same in every wave)
  if (n > 2 && nparts == 1) { double scale = 1.0 / (double)((int64_t)(n - 1) * (int64_t)(n - 2)); for (int v = tid; v < n; v += nthreads) c.bc[v] *= scale; }  // (several parts: ge_k_feat_combine rescales)
  // clustering (directed formula on the symmetric graph)
  double *clus = c.clus;
  for (int i = tid; i < n; i += nthreads) {
    int64_t common = 0, dg = c.rowptr[i + 1] - c.rowptr[i];
    for (int k = c.rowptr[i]; k < c.rowptr[i + 1]; k++) {
      int j;
      if (closed) { const int q = k - c.rowptr[i]; j = q < i ? q : q + 1; } else j = c.colw[k] >> 4;
      for (int w = 0; w < W; w++) common += ge_popc64(c.abits[i * W + w] & c.abits[j * W + w]);
    }
    int64_t t8 = 8 * common, dt = 2 * dg, db = dg;
    clus[i] = (t8 == 0) ? 0.0 : (double)t8 / (double)((dt * (dt - 1) - 2 * db) * 2);
  }
  // pagerank ([nx] _pagerank_scipy): pull over in-neighbours in ascending order
  const bool prw = (t == GE_TSP);
  const double pinit = 1.0 / (double)n;
  int ndang = 0;
  for (int k0 = 0; k0 < n; k0 += GE_WAVE) {  // every wave counts the dangling nodes itself
    int i = k0 + lane;
    ndang += ge_popc64(ge_ballot(i < n && c.rowptr[i + 1] == c.rowptr[i]));
  }
  double *wl = (double *)(ge_dyn_smem() + P.ldsf.wl);  // weight of a code: a table read per row entry instead of a division
  if (tid < 16) wl[tid] = ge_wlut(tid);
  ge_sync();
  for (int i = tid; i < n; i += nthreads) {
    double S = 0.0;
    for (int k = c.rowptr[i]; k < c.rowptr[i + 1]; k++) S += (prw ? (P.spatial ? G.sw64[ebase + k] : wl[closed ? (int)c8[k] : (int)(c.scw[k] & 15)]) : 1.0) * 1.0;
    c.sinv[i] = (S != 0.0) ? 1.0 / S : 0.0;
    c.prx[i] = pinit;
  }
  ge_sync();
  const double alpha = 0.85, oma = 1 - alpha, tol = 1.0e-6;
  bool conv = false;
  for (int it = 0; it < ((GE_FABL & 8) ? 1 : 100) && !conv; it++) {
    double dsum = 0.0;
    if (ndang) { bool first = true; for (int i = 0; i < n; i++) if (c.rowptr[i + 1] == c.rowptr[i]) { dsum = first ? c.prx[i] : dsum + c.prx[i]; first = false; } }
    for (int i = tid; i < n; i += nthreads) {
      double acc = 0.0;
      // GE_PR_CH row entries per trip, every load unconditional (an entry past the end of the row re-reads its first) and the terms
      // added in row order afterwards: the entry, then sinv / x / the weight it points at, are two LDS round trips per TRIP where
      // the entry-by-entry loop paid two per entry (a complete 128-node graph: 127 entries per row, ~25 iterations)
      const int r0 = c.rowptr[i], r1 = c.rowptr[i + 1];
      for (int k0 = r0; k0 < r1; k0 += GE_PR_CH) {
        int jn[GE_PR_CH], cd[GE_PR_CH]; double sv[GE_PR_CH], xv[GE_PR_CH], wv[GE_PR_CH];
#pragma unroll
        for (int q = 0; q < GE_PR_CH; q++) {
          const int kk = k0 + q < r1 ? k0 + q : r0;
          if (closed) { const int qq = kk - r0; jn[q] = qq < i ? qq : qq + 1; cd[q] = (int)c8[kk]; }
          else { const uint32_t e = (uint32_t)c.scw[kk]; jn[q] = (int)(e >> 4); cd[q] = (int)(e & 15u); }
        }
#pragma unroll
        for (int q = 0; q < GE_PR_CH; q++) {
          sv[q] = c.sinv[jn[q]]; xv[q] = c.prx[jn[q]];
          wv[q] = prw ? (P.spatial ? G.sw64[ebase + (k0 + q < r1 ? k0 + q : r0)] : wl[cd[q]]) : 1.0;
        }
#pragma unroll
        for (int q = 0; q < GE_PR_CH; q++) if (k0 + q < r1) acc += (sv[q] * wv[q]) * xv[q];
      }
      double xn = alpha * (acc + dsum * pinit) + oma * pinit;
      c.prn[i] = xn;
      c.diff[i] = __builtin_fabs(xn - c.prx[i]);
    }
    ge_sync();
    double err = ge_pw<5>(c.diff, n, lane);
    ge_sync();  // every wave has read diff[] and prx[] before they are rewritten
    for (int i = tid; i < n; i += nthreads) c.prx[i] = c.prn[i];
    ge_sync();
    if (err < (double)n * tol) conv = true;
  }

  // sf = torch.tensor(sf) -> float32; x[:, -5:] = sf
  for (int v = tid; v < n; v += nthreads) {
    float *xr = G.x + (nbase + v) * F + P.nflag;
    xr[0] = (float)(2.0 * (double)(c.rowptr[v + 1] - c.rowptr[v]));
    if (nparts == 1) { xr[1] = (float)c.bc[v]; xr[2] = (float)c.clos[v]; }
    xr[3] = (float)c.prx[v]; xr[4] = (float)clus[v];
  }
  ge_sync();
}

// ------------------------------------------------------------------------------------------------
// n <= 64 fast path.  LDS carve (bytes): abits 512 | lvl LV*512 | sig 64*66*2 | del 64*65*8 |
// x,y,sinv,diff,clos 5*512 | scode E (TSP only) | pre.
#ifndef GE_F64_LV
#define GE_F64_LV 12   // BFS levels kept per source; deeper graphs (or sigma > 65535) take the generic path
#endif
#define GE_F64_SS 66   // u16 stride of a sigma row  (33 dwords: odd, spreads banks)
#define GE_F64_SD 65   // f64 stride of a delta row

#ifndef GE_F64_INV
#define GE_F64_INV 128  // reciprocals 1/k kept in LDS for the path counts k < GE_F64_INV (larger counts divide)
#endif
struct GeF64 { uint64_t *abits, *lvl; uint16_t *sig; double *del, *x, *y, *sinv, *diff, *clos, *inv; uint8_t *scode; };

GE_HOSTDEV int ge_f64_bytes(int E, int tsp, int nblk) {
  int o = 512 + GE_F64_LV * 512 + 64 * GE_F64_SS * 2 + 64 * GE_F64_SD * 8 + 5 * 512 + GE_F64_INV * 8;
  o = (o + 15) & ~15;
  if (tsp) o += (E + 15) & ~15;
  return o + (nblk + 2) * 4 + 16;
}

GE_DEV GeF64 ge_carve_f64(int E, int tsp) {
  unsigned char *s = ge_dyn_smem();
  GeF64 c;
  c.abits = (uint64_t *)s; s += 512;
  c.lvl = (uint64_t *)s; s += GE_F64_LV * 512;
  c.del = (double *)s; s += 64 * GE_F64_SD * 8;
  c.x = (double *)s; s += 512; c.y = (double *)s; s += 512; c.sinv = (double *)s; s += 512; c.diff = (double *)s; s += 512; c.clos = (double *)s; s += 512;
  c.inv = (double *)s; s += GE_F64_INV * 8;
  c.sig = (uint16_t *)s; s += 64 * GE_F64_SS * 2;
  c.scode = (uint8_t *)(((uintptr_t)s + 15) & ~(uintptr_t)15);
  return c;
}
// byte offset of the queue prefix (nblk + 1 ints) and the overflow flag inside the dynamic LDS of the n <= 64 feature kernel
GE_HOSTDEV int ge_f64_pre_off(int E, int tsp, int nblk) { return ge_f64_bytes(E, tsp, nblk) - (nblk + 2) * 4 - 8; }

// 320-thread workgroup.  Waves 0-3 are the walkers: the Brandes walks are chains of LDS round trips, so the 64
// sources are spread over four waves (16 quads each) on the CU's four SIMDs.  Wave 4 is the node wave (one lane
// per node): it computes clustering and pagerank WHILE the walkers run their forward pass, then reduces
// betweenness / closeness and writes the five columns.  Slots that are too deep or whose path counts exceed the
// 16-bit counters are appended to work_list for the generic kernel.
#ifndef GE_F64_QL
#define GE_F64_QL 4  // lanes per BFS source (4 = quad: 16 sources per wave, 4 walker waves; 2 = pair: 32 per wave, 2 waves)
#endif
#ifndef GE_F64_K
#define GE_F64_K 3   // nodes of a level a quad handles per walk iteration (measured on the headline config: 2: 248 us, 3: 242 us, 4: 244 us)
#endif
#ifndef GE_F64_ABL
#define GE_F64_ABL 0  // diagnostic ablation bits (tools/f64_phase.py; the results are wrong by construction): 1 forward walk, 2 backward walk, 4 pagerank
                      // iterations, 8 clustering, 16 betweenness reduction; 0 when shipped
#endif
#define GE_F64_WALKERS (64 * GE_F64_QL)
#define GE_F64_THREADS (GE_F64_WALKERS + 64)
#define GE_F64_SLICE (64 / GE_F64_QL)  // nodes per lane slice
#if GE_F64_SLICE > 32
typedef uint64_t ge_slice_t;  // one lane per source: the lane serves every target itself
#define GE_SLICE_CTZ(x) ge_ctz64(x)
#else
typedef uint32_t ge_slice_t;
#define GE_SLICE_CTZ(x) ((int)__builtin_ctz(x))
#endif
GE_DEV void ge_features64_env(const GeParams &P, int env, int *ovf_flag, int32_t *work_count, int32_t *work_list, int env_global) {
  const int tid = ge_tid_fresh();
  const bool node_wave = tid >= GE_F64_WALKERS;
  const int lane = tid - GE_F64_WALKERS;  // node index inside the node wave
  const int n = P.n, E = P.E, F = P.F, t = P.env_type;
  const ge_buffers &G = P.buf;
  const int64_t nbase = (int64_t)env * n, ebase = (int64_t)env * E;
  const bool prw = (t == GE_TSP);
  GeF64 c = ge_carve_f64(E, prw);
  const bool live = node_wave && lane < n;
  GE_STAMP(11);
  const uint64_t adj = live ? G.adj_bits[nbase + lane] : 0ull;
  const int deg = ge_popc64(adj);
  if (node_wave) c.abits[lane] = adj;
  if (tid == 0) *ovf_flag = 0;
  for (int i = tid; i < 64 * GE_F64_SS / 2; i += GE_F64_THREADS) ((uint32_t *)c.sig)[i] = 0u;
  for (int i = tid; i < 64 * GE_F64_SD; i += GE_F64_THREADS) c.del[i] = 0.0;
  if (prw) for (int i = tid; i < E; i += GE_F64_THREADS) c.scode[i] = G.scode[ebase + i];
  for (int i = tid; i < GE_F64_INV; i += GE_F64_THREADS) c.inv[i] = 1.0 / (double)(i > 0 ? i : 1);  // correctly rounded reciprocals
  ge_sync();

  GE_STAMP(12);
  // ---- Brandes, lane = source.  Forward: BFS + path counts, pushing sigma along DAG edges.
  // Four lanes (a quad) per source, 16 sources per wave: the quad keeps identical copies of the walk state and
  // splits the pushes of a node -- lane q takes the q-th, (q+4)-th, ... target -- so a node's DAG edges are
  // served in parallel while the per-source order of the float64 delta sums stays fixed.
  const int s = (tid >> 6) * (64 / GE_F64_QL) + ((tid & 63) / GE_F64_QL);
  const int q = tid & (GE_F64_QL - 1);
  const bool walker = !node_wave && s < n;
  // push targets of this lane: node (SLICE q + b), b = bit inside the slice.  The sigma counters are half-words, two to a dword;
  // a row is GE_F64_SS (even) half-words, so the dword of (node, s) is row * SS/2 + (s >> 1) and the half inside it depends on s
  // alone: one multiply-add per push instead of rebuilding the index
  static_assert(GE_F64_SS % 2 == 0, "sigma rows must be whole dwords");
  uint32_t *const sig_mine = (uint32_t *)c.sig + (GE_F64_SLICE * q) * (GE_F64_SS / 2) + (s >> 1);
  const int sig_sh = 16 * (s & 1);
  double *const del_mine = c.del + (GE_F64_SLICE * q) * GE_F64_SD + s;
  bool ovf = false;
  int D = 0, reach = 1; int64_t tot = 0;
  GE_STAMP_T0(24);
  static_assert(GE_F64_QL == 4, "the walk keeps the visited / next-level sets as 16-bit slices, one per lane of a quad");
  const uint16_t *const adj_mine = (const uint16_t *)c.abits + q;  // this lane's 16 columns of an adjacency row: adj_mine[4 u]
  if (walker && !(GE_F64_ABL & 1)) {
    // the lane only ever pushes to the nodes of its own slice, so it keeps just that slice of the visited set and of the level being
    // discovered (one 32-bit operation where the whole sets took two); the quad assembles the whole next level once per level
    uint64_t cur = 1ull << s;
    uint32_t vis16 = (uint32_t)((1ull << s) >> (16 * q)) & 0xffffu, nxt16 = 0u;
    if (q == 0) c.sig[s * GE_F64_SS + s] = 1;
    for (;;) {
      if (cur == 0) {
        const uint64_t nxt = ge_quad_gather16(nxt16);
        if (!nxt) break;
        vis16 |= nxt16; D++;
        if (D < GE_F64_LV) { if (q == 0) c.lvl[D * 64 + s] = nxt; } else ovf = true;
        const int cnt = ge_popc64(nxt);
        reach += cnt; tot += (int64_t)D * cnt;
        cur = nxt; nxt16 = 0u;
      }
      ge_quad_sync();  // the quad's pushes of the previous nodes are in LDS before these nodes are read
      // GE_F64_K nodes of the current level per iteration: their counts are final, their pushes are commutative adds, and the K
      // reads / push streams are independent of each other (the fixed cost of an iteration is shared, the LDS round trips overlap)
      int u[GE_F64_K]; bool has[GE_F64_K];
#pragma unroll
      for (int k = 0; k < GE_F64_K; k++) { has[k] = cur != 0; u[k] = has[k] ? ge_ctz64(cur) : u[0]; cur &= cur - 1; }
      uint32_t su[GE_F64_K], ab[GE_F64_K];
#pragma unroll
      for (int k = 0; k < GE_F64_K; k++) { su[k] = c.sig[u[k] * GE_F64_SS + s]; ab[k] = adj_mine[4 * u[k]]; }
      uint32_t any_su = 0; ge_slice_t mine[GE_F64_K], any_mine = 0;
      const uint32_t open16 = ~vis16;
#pragma unroll
      for (int k = 0; k < GE_F64_K; k++) {
        any_su |= su[k];
        // lane q of the quad serves the targets in nodes [16q, 16q+16): a 16-bit slice per node, 32-bit bit tricks
        mine[k] = has[k] ? (ab[k] & open16) : 0u; any_mine |= mine[k];
        su[k] <<= sig_sh;
      }
      nxt16 |= any_mine;
      if (any_su > 1023u) ovf = true;            // 64 parents x 1023 still fit the 16-bit counters
      while (any_mine) {
        any_mine = 0;
#pragma unroll
        for (int k = 0; k < GE_F64_K; k++) {
          if (mine[k]) {  // sigma[v] += sigma[u]: one ds_add_u32 on the half-word's dword, nothing to wait for
            ge_lds_add_u32(sig_mine + GE_SLICE_CTZ(mine[k]) * (GE_F64_SS / 2), su[k]); mine[k] &= mine[k] - 1;
          }
          any_mine |= mine[k];
        }
      }
    }
  }
  double clus = 0.0, x = 0.0;
  if (node_wave) {  // concurrent with the walkers' forward pass
  // clustering (directed formula on the symmetric graph)
  if (live && !(GE_F64_ABL & 8)) {
    int64_t common = 0;
    for (uint64_t r = adj; r; r &= r - 1) common += ge_popc64(adj & c.abits[ge_ctz64(r)]);
    const int64_t t8 = 8 * common, dt = 2 * (int64_t)deg, db = deg;
    clus = (t8 == 0) ? 0.0 : (double)t8 / (double)((dt * (dt - 1) - 2 * db) * 2);
  }
  // pagerank ([nx] _pagerank_scipy): x_new[i] = sum over in-neighbours j ascending of (sinv[j]*w_ji) * x[j]
  int rp = 0;  // row start in ascending-neighbour order = exclusive scan of degrees
  if (prw) { int incl = ge_wave_incl_scan(deg, lane); rp = incl - deg; }
  double S = 0.0;
  { int k = 0; for (uint64_t r = adj; r; r &= r - 1, k++) S += (prw ? ge_wlut(c.scode[rp + k]) : 1.0) * 1.0; }
  const double sinv = (S != 0.0) ? 1.0 / S : 0.0;
  const double pinit = 1.0 / (double)n;
  const double alpha = 0.85, oma = 1 - alpha, tol = 1.0e-6;
  const uint64_t dangling = ge_ballot(live && deg == 0);
  x = pinit;
  c.sinv[lane] = sinv;
  // the 16 smallest neighbours as byte indices in four registers: the pull of an iteration becomes 16 unrolled (predicated) LDS
  // reads and adds in ascending-neighbour order instead of a ctz / clear-lowest-bit loop of max-degree trips; `rest` = what is left
  // of a row with more than 16 neighbours
  uint32_t nb4[4] = {0u, 0u, 0u, 0u}; uint64_t rest = adj;
  { int k = 0; for (; rest && k < 16; rest &= rest - 1, k++) nb4[k >> 2] |= (uint32_t)ge_ctz64(rest) << (8 * (k & 3)); }
  bool conv = false;
  for (int it = 0; it < ((GE_F64_ABL & 4) ? 1 : 100) && !conv; it++) {
    c.x[lane] = x; c.y[lane] = sinv * x;  // unweighted: data'[j->i] * x[j] is the same product for every i
    ge_wave_sync();
    double dsum = 0.0;
    { bool first = true; for (uint64_t r = dangling; r; r &= r - 1) { double xv = c.x[ge_ctz64(r)]; dsum = first ? xv : dsum + xv; first = false; } }
    double xn = 0.0;
    if (live) {
      double acc = 0.0;
      if (prw) { int k = 0; for (uint64_t r = adj; r; r &= r - 1, k++) { const int j = ge_ctz64(r); acc += (c.sinv[j] * ge_wlut(c.scode[rp + k])) * c.x[j]; } }
      else {
#pragma unroll
        for (int k = 0; k < 16; k++) { const double yv = c.y[(nb4[k >> 2] >> (8 * (k & 3))) & 63u]; acc += (k < deg) ? yv : 0.0; }  // x + 0.0 == x
        for (uint64_t r = rest; r; r &= r - 1) acc += c.y[ge_ctz64(r)];
      }
      xn = alpha * (acc + dsum * pinit) + oma * pinit;
    }
    c.diff[lane] = live ? __builtin_fabs(xn - x) : 0.0;
    ge_wave_sync();
    const double err = ge_pw<0>(c.diff, n, lane);
    x = xn;
    ge_wave_sync();
    if (err < (double)n * tol) conv = true;
  }
  }
  // No barrier between the passes: the backward pass only touches the lane's own source columns.  A slot that
  // overflowed is discarded as a whole after the last barrier (the generic kernel recomputes it).
  if (ovf) *ovf_flag = 1;
  GE_STAMP(13);
  GE_STAMP_T0(25);
  // Backward: dependencies, deepest level first; delta[v] += sigma[v] * (1 + delta[w]) / sigma[w]
  if (walker && !ovf && !(GE_F64_ABL & 2)) {
    int d = D;
    uint64_t cur = d >= 1 ? c.lvl[d * 64 + s] : 0ull;
    uint64_t prev = d >= 2 ? c.lvl[(d - 1) * 64 + s] : (1ull << s);
    uint32_t prev16 = (uint32_t)(prev >> (16 * q)) & 0xffffu;  // this lane's slice of the level above
    while (d >= 1) {
      if (cur == 0) {
        d--;
        if (d >= 1) { cur = c.lvl[d * 64 + s]; prev = d >= 2 ? c.lvl[(d - 1) * 64 + s] : (1ull << s); prev16 = (uint32_t)(prev >> (16 * q)) & 0xffffu; }
        continue;
      }
      ge_quad_sync();
      // GE_F64_K nodes of the level per iteration (independent division chains overlap).  del[w] holds S(w) = sum of
      // coeff over w's DAG successors (deeper level, finished); delta(w) = sigma(w) * S(w) and coeff(w) =
      // (1 + delta(w)) / sigma(w) is pushed to w's predecessors.  S stays in del[]: the betweenness reduction
      // multiplies by sigma again, so nothing is rewritten here.
      int w[GE_F64_K]; bool has[GE_F64_K];
#pragma unroll
      for (int k = 0; k < GE_F64_K; k++) { has[k] = cur != 0; w[k] = has[k] ? ge_ctz64(cur) : w[0]; cur &= cur - 1; }
      uint32_t sg[GE_F64_K], ab[GE_F64_K]; double S[GE_F64_K], rs[GE_F64_K];
#pragma unroll
      for (int k = 0; k < GE_F64_K; k++) { sg[k] = c.sig[w[k] * GE_F64_SS + s]; S[k] = c.del[w[k] * GE_F64_SD + s]; ab[k] = adj_mine[4 * w[k]]; }
#pragma unroll
      for (int k = 0; k < GE_F64_K; k++) rs[k] = c.inv[sg[k] < GE_F64_INV ? sg[k] : 0u];
      double coeff[GE_F64_K]; ge_slice_t mine[GE_F64_K], any_mine = 0;
#pragma unroll
      for (int k = 0; k < GE_F64_K; k++) {
        // coeff(w) = (1 + delta(w)) / sigma(w) with delta(w) = sigma(w) S(w), i.e. 1 / sigma(w) + S(w): the reciprocal of the
        // (small, integer) path count comes from the LDS table, one float64 add instead of a multiply, an add and a division
        coeff[k] = (sg[k] < GE_F64_INV ? rs[k] : 1.0 / (double)sg[k]) + S[k];
        // lane q serves the predecessors in nodes [16q, 16q+16); a node's accumulator is always updated by the same lane, in the
        // same interleaving of the K push streams, iteration after iteration: the float64 sum order is fixed
        mine[k] = has[k] ? (ab[k] & prev16) : 0u; any_mine |= mine[k];
      }
      while (any_mine) {
        any_mine = 0;
#pragma unroll
        for (int k = 0; k < GE_F64_K; k++) {
          if (mine[k]) { ge_lds_add_f64(del_mine + GE_SLICE_CTZ(mine[k]) * GE_F64_SD, coeff[k]); mine[k] &= mine[k] - 1; }  // ds_add_f64
          any_mine |= mine[k];
        }
      }
    }
  }
  GE_STAMP_T0(26);
  // closeness (wf_improved) of source s, handed to the node-per-lane phase through LDS
  if (walker && q == 0) {
    double cl = 0.0;
    if (tot > 0 && n > 1) { cl = ((double)reach - 1.0) / (double)tot; cl *= ((double)reach - 1.0) / (double)(n - 1); }
    c.clos[s] = cl;
  }
  ge_sync();
  GE_STAMP(14);
  if (*ovf_flag) {  // uniform: hand the slot to the generic kernel
    if (tid == 0) { int k = atomicAdd(&work_count[0], 1); work_list[k] = env_global; }
    ge_sync();
    return;
  }
  if (node_wave) {  // betweenness / closeness reduction and the write, one lane per node
  // betweenness[w] = sum over sources in node order, w itself excluded; then the 1/((n-1)(n-2)) rescale
  double bc = 0.0;
  if (live && !(GE_F64_ABL & 16)) {
    // delta = sigma * S, added in source order.  Eight sources per trip with their sixteen LDS reads in flight (the additions stay a
    // chain, in order); the node's own column and the columns of sources that do not exist (zero, never written) add +0.0.
#pragma unroll 1
    for (int s0 = 0; s0 < 64; s0 += 8) {
      uint32_t sg[8]; double sv[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { sg[k] = c.sig[lane * GE_F64_SS + s0 + k]; sv[k] = c.del[lane * GE_F64_SD + s0 + k]; }
#pragma unroll
      for (int k = 0; k < 8; k++) bc += (s0 + k != lane) ? (double)sg[k] * sv[k] : 0.0;
    }
    if (n > 2) bc *= 1.0 / (double)((int64_t)(n - 1) * (int64_t)(n - 2));
  }
  const double clos = live ? c.clos[lane] : 0.0;
  GE_STAMP(16);
  if (live) {
    float *xr = G.x + (nbase + lane) * F + P.nflag;
    xr[0] = (float)(2.0 * (double)deg); xr[1] = (float)bc; xr[2] = (float)clos; xr[3] = (float)x; xr[4] = (float)clus;
  }
  }
  ge_sync();
  GE_STAMP(17);
}


// the class of a global slot (multi-class engine), as a wave-uniform value
GE_DEV int ge_slot_class(const GeRagged &R, int env) { return (int)ge_uniform_u32((uint32_t)R.slot_class[env]); }

// run.items GE_ITEMS_ALL: every slot; GE_ITEMS_QUEUE: the slots of P.buf.reset_list; GE_ITEMS_LIST: work_list (fallback of the
// fast path).  pre_off: byte offset of the queue prefix inside the dynamic LDS (behind the largest class's scratch in a multi-class
// engine).
template <bool RAGGED>
GE_KERNEL ge_k_features(GeParams P, GeRagged R, GeRun run, int pre_off, int bucket) {
  int *pre = (int *)(ge_dyn_smem() + pre_off);
  const bool queue = run.items == GE_ITEMS_QUEUE, list = run.items == GE_ITEMS_LIST;
  int count = list ? P.buf.work_count[0] : P.B;
  if (queue) {
    if (ge_tid() < GE_WAVE) ge_queue_prefix_wave(P, pre, ge_tid());
    ge_sync();
    count = pre[(P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK];
  }
  // workgroups per item: the uniform engine's fallback list is rare and small (one workgroup each); the multi-class engine sends
  // every slot of a class with n > 64 through the list, feat_parts workgroups each (P.feat_parts = the largest class's)
  const int fparts = (list && !RAGGED) ? 1 : P.feat_parts;
  const int nparts = ge_feat_workgroups(fparts);  // workgroups per item (engine-wide: the largest class's)
  for (int q = ge_bid(); q < count * nparts; q += ge_gdim()) {
    const int item = q / nparts, part = q % nparts;
    const int env = queue ? ge_queue_slot(P, pre, item) : (list ? P.buf.work_list[item] : item);
    if constexpr (RAGGED) {
      const int cls = ge_slot_class(R, env);
      const GeParams &C = R.classes[cls];
      if (bucket >= 0 && C.bucket != bucket) continue;  // one launch per LDS bucket of size classes (see ge_k_reset)
      if (queue && part == 0 && ge_tid() == 0) ge_finish_item(P, run, env);
      if (part < ge_feat_workgroups(C.feat_parts)) ge_features_generic_env(C, env - R.class_start[cls], part, C.feat_parts);  // uniform per workgroup
    } else {
      if (queue && part == 0 && ge_tid() == 0) ge_finish_item(P, run, env);  // seed[] / episode[] now name the new episode (refill: the image is valid)
      ge_features_generic_env(P, env, part, fparts);
    }
  }
}

// betweenness of multi-part slots: parts added in part order, then the 1/((n-1)(n-2)) rescale and the float32 cast.  Multi-class
// engine: items are (slot, node) over the widest class's n; a thread looks up its slot's class and skips what does not exist there.
template <bool RAGGED>
GE_KERNEL ge_k_feat_combine(GeParams P, GeRagged R, GeRun run) {
  int *pre = (int *)ge_dyn_smem();
  const bool queue = run.items == GE_ITEMS_QUEUE, list = run.items == GE_ITEMS_LIST;
  int count = list ? P.buf.work_count[0] : P.B;
  if (queue) {
    if (ge_tid() < GE_WAVE) ge_queue_prefix_wave(P, pre, ge_tid());
    ge_sync();
    count = pre[(P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK];
  }
  const int nmax = P.n;
  for (int64_t g = (int64_t)ge_bid() * ge_bdim() + ge_tid(); g < (int64_t)count * nmax; g += (int64_t)ge_gdim() * ge_bdim()) {
    const int item = (int)(g / nmax), v = (int)(g % nmax);
    int env = queue ? ge_queue_slot(P, pre, item) : (list ? P.buf.work_list[item] : item);
    int cls = 0;
    if constexpr (RAGGED) { cls = R.slot_class[env]; env -= R.class_start[cls]; }
    const GeParams &C = RAGGED ? R.classes[cls] : P;
    const int n = C.n, nparts = C.feat_parts;
    if (v >= n || nparts == 1) continue;
    double acc = 0.0;
    for (int p = 0; p < nparts; p++) acc += C.buf.feat_scratch[((int64_t)env * nparts + p) * n + v];
    if (n > 2) acc *= 1.0 / (double)((int64_t)(n - 1) * (int64_t)(n - 2));
    C.buf.x[((int64_t)env * n + v) * C.F + C.nflag + 1] = (float)acc;
  }
}

// n <= 64 fast path over every slot / the queue.  Multi-class engine: a slot of a class with n > 64 goes straight to work_list.
template <bool RAGGED>
GE_KERNEL ge_k_features64(GeParams P, GeRagged R, GeRun run, int pre_off) {
  const int nblk = (P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK;
  int *pre = (int *)(ge_dyn_smem() + pre_off);
  const bool queue = run.items == GE_ITEMS_QUEUE;
  int count = P.B;
  if (queue) {  // the prefix scan is one wave wide
    if (ge_tid() < GE_WAVE) ge_queue_prefix_wave(P, pre, ge_tid());
    ge_sync();
    count = pre[nblk];
  }
  for (int q = ge_bid(); q < count; q += ge_gdim()) {
    const int env = queue ? ge_queue_slot(P, pre, q) : q;
    if (queue && ge_tid() == 0) ge_finish_item(P, run, env);  // seed[] / episode[] now name the new episode (refill: the image is valid)
    if constexpr (RAGGED) {
      const int cls = ge_slot_class(R, env);
      const GeParams &C = R.classes[cls];
      if (C.n > 64) {  // uniform: the generic kernel takes the slot (global id)
        if (ge_tid() == 0) { int k = atomicAdd(&P.buf.work_count[0], 1); P.buf.work_list[k] = env; }
        continue;
      }
      ge_features64_env(C, env - R.class_start[cls], pre + nblk + 1, P.buf.work_count, P.buf.work_list, env);
    } else {
      ge_features64_env(P, env, pre + nblk + 1, P.buf.work_count, P.buf.work_list, env);
    }
  }
}
