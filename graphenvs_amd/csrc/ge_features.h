// Structural-feature kernels: the five columns of feature_extraction.py:6-37 for freshly generated slots.
//   ge_features_generic_env : any n; graph staged in LDS, level-synchronous Brandes, one source at a time.
//   ge_f64_walk_env / ge_f64_node_env : n <= 64; eight lanes per BFS source, path counts and dependencies pulled level by level
//                             (no atomics); clustering and pagerank one lane per node, as an item of their own.
// Both produce float64 results in the reference's operation order where the order is observable, then round to
// float32 exactly once (sf = torch.tensor(sf)).
#pragma once
#include "ge_params.h"
#include "ge_platform.h"
#include "ge_reset.h"

struct GeFctx {
  uint64_t *abits; int *rowptr; uint16_t *scw;
  uint64_t *quads; uint32_t *rq; int zq;        // Brandes role: the rows as quads of neighbour ids, {first quad | quads << 16} per node, the all-padding quad
  uint8_t *c8;                                  // complete graph on all n nodes: the weight-code bytes (in the quads' place)
  double *sigma, *delta, *coeff, *bcw;          // per-wave scratch of the Brandes pass (this wave's area); coeff[n] = 0.0, the zero node
  double *bcw0; int wave_f64;                   // bcw of wave 0 and the float64 stride between waves (partial sums are combined in wave order)
  double *bc, *clos;                            // common
  double *prx, *prn, *sinv, *diff, *diff2, *clus;  // node role
  uint8_t *mark;                                // per wave: mark[w] = 1: w is a neighbour of the level just walked
  uint16_t *ord, *lvl;                          // per wave: BFS order of the current source; lvl[d] = where level d starts in it
};

GE_DEV GeFctx ge_carve_f(const GeParams &P, int tid) {
  unsigned char *s = ge_dyn_smem();
  const GeLdsF &L = P.ldsf;
  const int n = P.n, an = ((n * 8 + 15) & ~15) / 8;
  GeFctx c;
  c.rowptr = (int *)(s + L.rowptr); c.quads = (uint64_t *)(s + L.colw); c.c8 = (uint8_t *)(s + L.colw); c.rq = (uint32_t *)(s + L.rq); c.zq = L.zq;
  c.bc = (double *)(s + L.bc); c.clos = (double *)(s + L.clos);
  unsigned char *w = s + L.wave0 + (tid >> 6) * L.wave_stride;
  // (P.W <= GE_BCW_REG_W: the wave's betweenness partial sums live in registers while it runs and are handed over in delta[])
  const int bcw_at = P.W <= GE_BCW_REG_W ? n : 3 * n + 1;
  c.sigma = (double *)(w + L.w_sigma); c.delta = c.sigma + n; c.coeff = c.sigma + 2 * n; c.bcw = c.sigma + bcw_at;
  c.mark = (uint8_t *)(w + L.w_mark); c.ord = (uint16_t *)(w + L.w_ord); c.lvl = (uint16_t *)(w + L.w_lvl);
  c.bcw0 = (double *)(s + L.wave0 + L.w_sigma) + bcw_at; c.wave_f64 = L.wave_stride / 8;
  c.abits = (uint64_t *)(s + L.abits); c.scw = (uint16_t *)(s + L.scw);
  c.prx = (double *)(s + L.prx); c.prn = c.prx + an; c.sinv = c.prx + 2 * an; c.diff = c.prx + 3 * an; c.diff2 = c.prx + 4 * an; c.clus = (double *)(s + L.clus);
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
                   // walk, 2 backward coefficient pass, 4 backward pull, 8 pagerank iterations; 0 when shipped
#endif
// Diagnostic build only (-DGE_STAMPS, never shipped; tools/feat_phase_clocks.py): shader-clock cycles wave 0 of slot 0's first Brandes
// workgroup spends in the phases of the generic kernel -- [k] = cycles from stamp k to the next stamp, summed over sources and levels:
// 0 new level + front update, 1 forward walk, 3 coefficients, 4 dependency pull, 5 end of a source, 6 set-up of a source;
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
// One walk of the Brandes pass over the nodes ord[k0 .. k1) of a level, a lane per node, four row entries (one quad) per trip.  The
// kernel is bound by vector-instruction issue (0.20 instructions per SIMD-cycle at four waves per SIMD, profiles/r04_generic_*), so a
// row entry is made to cost as few instructions as it can: the row is a run of 8-byte quads of neighbour ids, padded with the zero
// node n (coeff[n] = 0.0 always) -- no bound test, no select, one LDS read per four ids --, and a lane whose row has ended reads the
// all-padding quad until the longest row of the trip is done.
//  forward (BWD false): sigma[v] = sum of coeff[w] -- coeff[] holds sigma(u) on the level above and 0 everywhere else, so the
//    test "is w one level up" is gone; the counts are integers, exact in any order -- and every neighbour is MARKED (a plain byte
//    store: everybody writes the same 1), which is how the next level is found: no second walk over adjacency bit rows;
//  backward (BWD true): delta[v] = sum of sigma[v] * coeff[w] IN ROW ORDER -- coeff[] holds (1 + delta(w)) / sigma(w) on the level
//    below and 0 everywhere else; a padding entry or a neighbour that is not one level deeper adds sigma[v] * 0.0 = +0.0.
#ifndef GE_BW_TWO_ABOVE
#define GE_BW_TWO_ABOVE 256  // graphs above this many nodes run fewer than 16 waves per CU (a wave's area is 29 n bytes of LDS)
#endif
#ifndef GE_BW_U
#define GE_BW_U 1  // rows a lane takes per trip of a walk over a level of more than 64 nodes, where the geometry runs fewer than 16 waves per CU
                   // (n > GE_BW_TWO_ABOVE).  Two rows double the reads in flight; measured (round 4, same box): n = 320: 4.02 us per slot with one row,
                   // 4.17 with two; n = 400: 5.93 / 5.98; n = 512: 10.63 / 10.65; config 5: 19.5 M either way -- one row shipped
#endif
template <bool BWD, int U>
GE_DEV void ge_brandes_walk_u(const GeFctx &c, int k0, int k1, int lane, bool store) {
  for (int kb = k0; kb < k1; kb += U * GE_WAVE) {
    bool on[U]; int v[U]; uint32_t q0[U], nq[U]; double sv[U], acc[U]; uint64_t q[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int k = kb + u * GE_WAVE + lane; on[u] = k < k1; v[u] = (int)c.ord[on[u] ? k : k0]; }
#pragma unroll
    for (int u = 0; u < U; u++) { const uint32_t r = c.rq[v[u]]; q0[u] = r & 0xffffu; nq[u] = on[u] ? (r >> 16) : 0u; sv[u] = BWD ? c.sigma[v[u]] : 1.0; acc[u] = 0.0; }
#pragma unroll
    for (int u = 0; u < U; u++) q[u] = c.quads[nq[u] ? q0[u] : (uint32_t)c.zq];
    for (uint32_t i = 0; ge_ballot(i < nq[0] || (U > 1 && i < nq[U - 1])) != 0ull;) {
      i++;
      uint64_t qn[U];
#pragma unroll
      for (int u = 0; u < U; u++) qn[u] = c.quads[i < nq[u] ? q0[u] + i : (uint32_t)c.zq];  // the next quad is on its way while this one's coefficients are read
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t lo = (uint32_t)q[u], hi = (uint32_t)(q[u] >> 32);
        const int w0 = (int)(lo & 0xffffu), w1 = (int)(lo >> 16), w2 = (int)(hi & 0xffffu), w3 = (int)(hi >> 16);
        const double f0 = c.coeff[w0], f1 = c.coeff[w1], f2 = c.coeff[w2], f3 = c.coeff[w3];
        if (!BWD) { c.mark[w0] = 1; c.mark[w1] = 1; c.mark[w2] = 1; c.mark[w3] = 1; }
        if (BWD) { acc[u] += sv[u] * f0; acc[u] += sv[u] * f1; acc[u] += sv[u] * f2; acc[u] += sv[u] * f3; }
        else { acc[u] += f0; acc[u] += f1; acc[u] += f2; acc[u] += f3; }
        q[u] = qn[u];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) if (on[u] && store) (BWD ? c.delta : c.sigma)[v[u]] = acc[u];
  }
}
template <bool BWD>
GE_DEV void ge_brandes_walk(const GeFctx &c, int k0, int k1, int lane, bool store, bool two) {
  if (GE_BW_U > 1 && two && k1 - k0 > GE_WAVE) ge_brandes_walk_u<BWD, GE_BW_U>(c, k0, k1, lane, store); else ge_brandes_walk_u<BWD, 1>(c, k0, k1, lane, store);
}

GE_DEV void ge_features_generic_env(const GeParams &P, int env, int part, int nparts) {
  const int tid = ge_tid_fresh(), nthreads = ge_bdim();
  // nwaves: the waves that run searches -- as many per-wave areas as THIS geometry holds (GeLdsF.waves).  A multi-class launch has
  // the threads of its bucket's widest wave count; the waves beyond a class's own count stage, join the barriers and do node work
  const int lane = tid & (GE_WAVE - 1), wv = tid >> 6, nwaves = P.ldsf.waves;
  const bool search_wave = wv < nwaves;
  const int n = P.n, W = P.W, E = P.E, F = P.F, t = P.env_type;
  const ge_buffers &G = P.buf;
  const int64_t nbase = (int64_t)env * n, ebase = (int64_t)env * E;
  GeFctx c = ge_carve_f(P, tid);
  // Two roles (GeLdsF): the Brandes workgroups (a share of the BFS sources each) and the node-level workgroup (clustering, pagerank,
  // degrees); with nparts == 1 one workgroup is both.  The node role's arrays -- adjacency bit rows, sorted edge copy, pagerank
  // vectors -- overlay the Brandes role's per-wave areas when the roles are different workgroups, so a Brandes workgroup holds only
  // what it uses (n = 256: 7 waves where 5 fitted, n = 512: 7 where 4 did); the searches need no adjacency bit rows at all.
  const bool node_part = nparts > 1 && part == nparts;  // this workgroup only does the node-level work
  const bool node_role = node_part || nparts == 1, brandes_role = !node_part;
  // stage the slot's graph in LDS
  if (node_role) { const uint64_t *arows = G.adj_bits + nbase * W; for (int i = tid; i < n * W; i += nthreads) c.abits[i] = arows[i]; }
  // complete graph on all n nodes: row i holds every other node in ascending order -- entry q of the row is node q (q < i) or q + 1 --
  // and only the weight codes are staged, a byte each (scode: ascending-neighbour order)
  const bool closed = P.complete && P.ng == n;
  // complete graph on all n nodes (TSP config 3): every pair is adjacent, so no shortest path has an interior node
  // (betweenness is a sum of zeros) and every BFS has one level of n-1 nodes (closeness (n-1)/(n-1) * (n-1)/(n-1))
  const bool trivial = closed || node_part;
  uint8_t *c8 = c.c8;
  const int32_t *grow = G.row_ptr + (int64_t)env * (n + 1);
  const uint16_t *gcolw = G.colw + ebase;
  for (int v = tid; v <= n; v += nthreads) c.rowptr[v] = grow[v];
  if (closed) { for (int idx = tid; idx < E; idx += nthreads) c8[idx] = G.scode[ebase + idx]; }
  else if (brandes_role) {
    // the rows as quads of neighbour ids.  Row v starts at quad (rowptr[v] + 3 v) / 4: consecutive starts are at least ceil(deg / 4)
    // apart (floor(a + b) - floor(a) >= floor(b)) and the last row ends below (E + 3 n) / 4 -- a closed form instead of a scan
    uint16_t *q16 = (uint16_t *)c.quads;
    for (int v = tid; v < n; v += nthreads) {
      const int r0 = grow[v], r1 = grow[v + 1], q0 = (r0 + 3 * v) >> 2, nq = (r1 - r0 + 3) >> 2;
      c.rq[v] = (uint32_t)q0 | ((uint32_t)nq << 16);
      for (int i = 0; i < nq; i++) {  // a quad per trip: its four global reads in flight together, one 8-byte LDS store
        uint32_t e[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const int k = r0 + 4 * i + j; const uint32_t x = gcolw[k < r1 ? k : r0]; e[j] = k < r1 ? (x >> 4) : (uint32_t)n; }
        c.quads[q0 + i] = (uint64_t)(e[0] | (e[1] << 16)) | ((uint64_t)(e[2] | (e[3] << 16)) << 32);
      }
    }
    if (tid < 4) q16[4 * c.zq + tid] = (uint16_t)n;
  }
  ge_sync();
  // rows in ascending-column order (scipy canonical CSR): position by rank in the bit row
  if (node_role && !closed) for (int v = tid; v < n; v += nthreads)
    for (int k = c.rowptr[v]; k < c.rowptr[v + 1]; k++) {
      const uint16_t e = gcolw[k];
      c.scw[c.rowptr[v] + ge_rank_below(c.abits + v * W, e >> 4)] = e;
    }
  const bool bcw_reg = W <= GE_BCW_REG_W;  // the wave's betweenness partial sums in registers: node lane + 64 j in bcr[j]
  double bcr[GE_BCW_REG_W];
#pragma unroll
  for (int j = 0; j < GE_BCW_REG_W; j++) bcr[j] = 0.0;
  if (brandes_role && !bcw_reg && search_wave) for (int v = lane; v < n; v += GE_WAVE) c.bcw[v] = 0.0;
  ge_sync();
  // Brandes betweenness + closeness: one level-synchronous BFS per source, sources dealt round-robin to the waves
  if (trivial && !node_part) for (int v = tid; v < n; v += nthreads) c.clos[v] = (((double)n - 1.0) / (double)(n - 1)) * (((double)n - 1.0) / (double)(n - 1));
  // One BFS per source, level by level, the nodes kept in discovery order (ord) with the start of every level (lvl): each pass
  // touches the nodes of ONE level and their rows -- O(n + E) per source -- in two phases per level and direction:
  //  forward  (a) walk the level: path counts by pull and a mark on every neighbour (ge_brandes_walk);
  //           (b) the next level = the marked nodes without a path count yet (sigma == 0: the high word of the float64 is enough),
  //               appended to ord by ballot / prefix count; in the same phase the front moves on: coeff[] = 0 on the level above,
  //               = sigma on the level just walked;
  //  backward (a) coeff[] = (1 + delta) / sigma on level lev, 0 again on level lev + 1;
  //           (b) walk level lev - 1: delta[v] = sum of sigma[v] * coeff[w] in row order.
  // No atomics, no per-node level array, and a neighbour's level is never tested: coeff[] is zero wherever a term must not count.
  const bool two = n > GE_BW_TWO_ABOVE;  // (geometries of fewer than 16 waves per CU: GE_BW_U)
  GE_FSTAMP_DECL;
  for (int s = part * nwaves + wv; s < n && !trivial && search_wave; s += nwaves * nparts) {
    GE_FSTAMP(6);
    for (int v = lane; v < n; v += GE_WAVE) { c.sigma[v] = (v == s) ? 1.0 : 0.0; c.delta[v] = 0.0; c.coeff[v] = (v == s) ? 1.0 : 0.0; c.mark[v] = 0; }
    // level 1 is the source's row as it stands (a simple graph: every neighbour once): no walk over level 0, no search for its marks.
    // The walk over level 1 then finds sigma = coeff[s] = 1 for each of them and marks level 2.
    const uint32_t rs = c.rq[s];
    const int deg_s = c.rowptr[s + 1] - c.rowptr[s];
    { const uint16_t *row = (const uint16_t *)(c.quads + (rs & 0xffffu)); for (int i = lane; i < deg_s; i += GE_WAVE) c.ord[1 + i] = row[i]; }
    if (lane == 0) { c.coeff[n] = 0.0; c.mark[n] = 0; c.ord[0] = (uint16_t)s; c.lvl[0] = 0; c.lvl[1] = 1; c.lvl[2] = (uint16_t)(1 + deg_s); }
    ge_wave_sync();
    // level d = ord[lo .. hi), level d - 1 = ord[lp .. lo); coeff[] = sigma on level d - 1
    int d = deg_s ? 1 : 0, reach = 1 + deg_s, lp = 0, lo = deg_s ? 1 : 0, hi = 1 + deg_s; int64_t tot = deg_s;
    const uint32_t *sig_hi = (const uint32_t *)c.sigma + 1;
    for (;;) {
      GE_FSTAMP(1);
      if (!(GE_FABL & 1) && d > 0) ge_brandes_walk<false>(c, lo, hi, lane, true, two);
      ge_wave_sync();
      GE_FSTAMP(0);
      int found = 0;
      for (int w0 = 0; w0 < n; w0 += 4 * GE_WAVE) {  // four chunks of 64 nodes per trip: their reads in flight together
        uint32_t m[4], sh[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const int w = w0 + GE_WAVE * j + lane; const bool in = w < n; m[j] = c.mark[in ? w : n]; sh[j] = sig_hi[2 * (in ? w : 0)]; }
#pragma unroll
        for (int j = 0; j < 4; j++) if (w0 + GE_WAVE * j < n) {  // (wave-uniform)
          const int w = w0 + GE_WAVE * j + lane; const bool in = w < n;
          const bool isnew = in && m[j] != 0u && sh[j] == 0u;
          const uint64_t b = ge_ballot(isnew);
          if (isnew) c.ord[hi + found + ge_mbcnt(b)] = (uint16_t)w;
          if (in) c.mark[w] = 0;
          found += ge_popc64(b);
        }
      }
      if (!found) break;
      for (int k = lp + lane; k < lo; k += GE_WAVE) c.coeff[c.ord[k]] = 0.0;                           // the front moves on
      for (int k = lo + lane; k < hi; k += GE_WAVE) { const int v = c.ord[k]; c.coeff[v] = c.sigma[v]; }
      d++; lp = lo; lo = hi; hi += found; reach += found; tot += (int64_t)d * found;
      if (lane == 0) c.lvl[d + 1] = (uint16_t)hi;
      ge_wave_sync();
    }
    for (int k = lp + lane; k < lo; k += GE_WAVE) c.coeff[c.ord[k]] = 0.0;  // coeff[] is all zero again
    ge_wave_sync();
    // backward accumulation, level by level.  Level 1 is where it ends: its coefficients would only serve delta[s], which nobody reads
    for (int lev = d; lev >= 2; lev--) {
      GE_FSTAMP(3);
      const int l0 = c.lvl[lev], l1 = c.lvl[lev + 1], l2 = lev < d ? (int)c.lvl[lev + 2] : l1, p0 = c.lvl[lev - 1];
      for (int k = l1 + lane; k < l2; k += GE_WAVE) c.coeff[c.ord[k]] = 0.0;
      for (int k = l0 + lane; k < l1 && !(GE_FABL & 2); k += GE_WAVE) { const int v = c.ord[k]; c.coeff[v] = (1.0 + c.delta[v]) / c.sigma[v]; if (!bcw_reg) c.bcw[v] += c.delta[v]; }
      ge_wave_sync();
      GE_FSTAMP(4);
      if (!(GE_FABL & 4)) ge_brandes_walk<true>(c, p0, l0, lane, true, two);  // (+0.0 for a neighbour that is not one level deeper)
      ge_wave_sync();
    }
    GE_FSTAMP(5);
    if (!bcw_reg) { const int l1 = c.lvl[2]; for (int k = 1 + lane; k < l1; k += GE_WAVE) { const int v = c.ord[k]; c.bcw[v] += c.delta[v]; } }
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
  if (bcw_reg && brandes_role && search_wave) {  // (this wave's delta[] is free now: every one of its searches is over)
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
      if (closed) { const int q = k - c.rowptr[i]; j = q < i ? q : q + 1; } else j = c.scw[k] >> 4;
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
  // ONE barrier per iteration: x and |x_new - x| are double-buffered, so an iteration writes arrays nobody is reading (a wave still
  // summing the previous iteration's differences reads the other pair), and the copy x <- x_new is a pointer swap.  A node-level
  // workgroup holds a whole CU's LDS while it iterates (20 % of a full reset at n = 512 with three barriers per iteration)
  double *xc = c.prx, *xw = c.prn, *dc = c.diff, *dw = c.diff2;
  bool conv = false;
  for (int it = 0; it < ((GE_FABL & 8) ? 1 : 100) && !conv; it++) {
    double dsum = 0.0;
    if (ndang) { bool first = true; for (int i = 0; i < n; i++) if (c.rowptr[i + 1] == c.rowptr[i]) { dsum = first ? xc[i] : dsum + xc[i]; first = false; } }
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
          sv[q] = c.sinv[jn[q]]; xv[q] = xc[jn[q]];
          wv[q] = prw ? (P.spatial ? G.sw64[ebase + (k0 + q < r1 ? k0 + q : r0)] : wl[cd[q]]) : 1.0;
        }
#pragma unroll
        for (int q = 0; q < GE_PR_CH; q++) if (k0 + q < r1) acc += (sv[q] * wv[q]) * xv[q];
      }
      double xn = alpha * (acc + dsum * pinit) + oma * pinit;
      xw[i] = xn;
      dc[i] = __builtin_fabs(xn - xc[i]);
    }
    ge_sync();
    const double err = ge_pw<5>(dc, n, lane);
    { double *t_ = xc; xc = xw; xw = t_; t_ = dc; dc = dw; dw = t_; }
    if (err < (double)n * tol) conv = true;
  }

  // sf = torch.tensor(sf) -> float32; x[:, -5:] = sf
  for (int v = tid; v < n; v += nthreads) {
    float *xr = G.x + (nbase + v) * F + P.nflag;
    xr[0] = (float)(2.0 * (double)(c.rowptr[v + 1] - c.rowptr[v]));
    if (nparts == 1) { xr[1] = (float)c.bc[v]; xr[2] = (float)c.clos[v]; }
    xr[3] = (float)xc[v]; xr[4] = (float)clus[v];
  }
  ge_sync();
}

// ------------------------------------------------------------------------------------------------
// n <= 64 fast path.  A 256-thread workgroup (four waves) works on ITEMS of two kinds:
//   walk item (one slot): betweenness and closeness.  The 64 BFS sources are dealt to the four waves in two rounds of eight sources
//     per wave, EIGHT LANES per source (an octet).  Per round and wave: (1) the levels of the eight searches by bit-row arithmetic
//     -- lane o of an octet holds the adjacency rows of nodes 8o .. 8o+7 in registers, ORs the rows of its nodes of the current
//     level, and three DPP moves combine the octet -- which also leaves every search's nodes in (level, node) order; (2) path counts
//     by PULL, level by level: lane o takes the o-th, (o+8)-th ... node of the level, intersects its adjacency row with the level
//     above and adds the counts of those nodes (plain LDS reads: no atomics, a node's count is written once, by its lane);
//     (3) dependencies the same way from the deepest level up: S(v) = sum over v's successors w of (1 / sigma(w) + S(w)), in
//     ascending w; (4) wave 0 adds sigma(v) S(v) to the betweenness of v over the round's 32 sources IN SOURCE ORDER (networkx
//     adds source after source: the float64 order of that sum is kept).  A wave keeps eight sources in LDS at a time (S 4 KB, counts
//     1 KB, level sets, order): 28 KB per workgroup where the walk of all 64 sources at once took 52 KB -- five workgroups per CU
//     instead of three -- and no LDS atomics (ds_add_f64 is one wave-instruction per 8 cycles of a CU's LDS,
//     tools/micro/valu_issue.hip).
//   node item (four slots, one per wave, one lane per node): clustering, scipy-order pagerank, degrees.
// The two kinds write different columns of x and share nothing, so they are independent items of one launch; a launch over c
// slots has c walk items followed by ceil(c / 4) node items.  Slots deeper than GE_F64_LV levels or with a path count of GE_F64_INV or
// more go to work_list: the generic kernel recomputes all five columns of such a slot.
#ifndef GE_F64_LV
#define GE_F64_LV 12   // BFS levels kept per source; deeper graphs take the generic path
#endif
#ifndef GE_F64_INV
#define GE_F64_INV 128  // reciprocals 1/k kept in LDS for the path counts k < GE_F64_INV; a slot with a larger count takes the generic path
                        // (G(64, 192): the largest count of a graph is 25 in the median, 47 at the 99.9th percentile)
#endif
#ifndef GE_F64_ABL
#define GE_F64_ABL 0  // diagnostic ablation bits (tools/f64_variants.py "name=-DGE_F64_ABL=3"; the results are wrong by construction): 1 forward pull, 2 backward pull, 4 pagerank
                      // iterations, 8 clustering, 16 betweenness reduction, 32 level search; 0 when shipped
#endif
#ifndef GE_F64_U
#define GE_F64_U 2    // entries of a predecessor / successor set a pull reads per trip, their LDS reads in flight together (headline loop, feature
                      // kernel alone: 1 -> 172 us, 2 -> 151, 3 -> 161, 4 -> 172, 6 -> 197: the kernel is bound by vector issue, and entries
                      // that do not exist cost their instructions)
#endif
#define GE_F64_THREADS 256
#define GE_F64_WAVES 4
#define GE_F64_SB 8                      // sources of a wave's sub-batch: eight lanes each
#define GE_F64_LST 16                    // level starts kept per source (GE_F64_LV + 1 <= GE_F64_LST)
#define GE_F64_ITEMS 16                  // items a workgroup looks up per queue-prefix rebuild
static_assert(GE_F64_LV + 1 <= GE_F64_LST, "level starts");
static_assert(GE_F64_INV <= 256, "path counts are kept as bytes");
// per-wave area of a walk item
// (rows of S and sigma carry a 65th entry that stays zero: the pulls read GE_F64_U entries of a set per trip and point the ones that
// do not exist at it -- coeff = inv[0] + 0.0 = +0.0 leaves a sum as it is)
#define GE_F64_SS 65                                              // row stride of S (float64)
#define GE_F64_SG 68                                              // row stride of sigma (bytes: every count below GE_F64_INV <= 256; whole dwords)
#define GE_F64_A_S 0                                              // double  S[8][65]
#define GE_F64_A_SIG (GE_F64_A_S + GE_F64_SB * GE_F64_SS * 8)     // u8 sigma[8][68]
#define GE_F64_A_LVL (GE_F64_A_SIG + GE_F64_SB * GE_F64_SG)       // u64 lvl[GE_F64_LV][8]
#define GE_F64_A_ORD (GE_F64_A_LVL + GE_F64_LV * GE_F64_SB * 8)   // u8 ord[8][64]: the nodes of a search in (level, node) order
#define GE_F64_A_LST (GE_F64_A_ORD + GE_F64_SB * 64)              // u8 lst[8][GE_F64_LST]: where level d starts in ord
#define GE_F64_A_BYTES (GE_F64_A_LST + GE_F64_SB * GE_F64_LST)
// walk item: abits 512 | inv | clos 512 | four wave areas
#define GE_F64_W_INV 512
#define GE_F64_W_CLOS (GE_F64_W_INV + GE_F64_INV * 8)
#define GE_F64_W_AREA (GE_F64_W_CLOS + 512)
#define GE_F64_W_BYTES (GE_F64_W_AREA + GE_F64_WAVES * GE_F64_A_BYTES)

// node item, per wave: abits 512 | x, y, sinv, diff 4 x 512 | scode E (TSP only: weighted pagerank)
GE_HOSTDEV int ge_f64_node_wave_bytes(int E, int tsp) { return 512 + 4 * 512 + (tsp ? ((E + 15) & ~15) : 0); }
// bytes of the item bodies (the queue prefix of a launch overlays them: nblk + 2 ints) = offset of the tail {items[64], flag}
GE_HOSTDEV int ge_f64_pre_off(int E, int tsp, int nblk) {
  int o = GE_F64_W_BYTES;
  const int nd = GE_F64_WAVES * ge_f64_node_wave_bytes(E, tsp), pf = (nblk + 2) * 4;
  if (nd > o) o = nd;
  if (pf > o) o = pf;
  return (o + 15) & ~15;
}
GE_HOSTDEV int ge_f64_bytes(int E, int tsp, int nblk) { return ge_f64_pre_off(E, tsp, nblk) + GE_F64_ITEMS * 4 * 4 + 16; }

// index of the lowest set bit of b, GE_F64_U times over (b loses them); a set that runs out gives `none` (the row's zero entry)
template <int BASE>
GE_DEV void ge_f64_next_bits(uint32_t &b, int (&idx)[GE_F64_U]) {
#pragma unroll
  for (int j = 0; j < GE_F64_U; j++) {
#ifdef GE_EMU
    idx[j] = b ? BASE + __builtin_ctz(b) : 64;
#else
    uint32_t z;  // v_ffbl_b32 gives 0xffffffff for an empty set: the minimum turns that into the zero entry's index
    asm("v_ffbl_b32 %0, %1" : "=v"(z) : "v"(b));
    idx[j] = (int)((z < (uint32_t)(64 - BASE) ? z : (uint32_t)(64 - BASE)) + (uint32_t)BASE);
#endif
    b &= b - 1u;
  }
}
// sum of the path counts row[i] over the set bits i of the 64-bit set p (integers: exact in any order)
GE_DEV uint32_t ge_f64_sum_counts(const uint8_t *row, uint64_t p) {
  uint32_t acc = 0u;
  for (uint32_t b = (uint32_t)p; b;) {
    int ix[GE_F64_U]; uint32_t r[GE_F64_U];
    ge_f64_next_bits<0>(b, ix);
#pragma unroll
    for (int j = 0; j < GE_F64_U; j++) r[j] = row[ix[j]];
#pragma unroll
    for (int j = 0; j < GE_F64_U; j++) acc += r[j];
  }
  for (uint32_t b = (uint32_t)(p >> 32); b;) {
    int ix[GE_F64_U]; uint32_t r[GE_F64_U];
    ge_f64_next_bits<32>(b, ix);
#pragma unroll
    for (int j = 0; j < GE_F64_U; j++) r[j] = row[ix[j]];
#pragma unroll
    for (int j = 0; j < GE_F64_U; j++) acc += r[j];
  }
  return acc;
}
// sum over the set bits w of sc, ascending, of coeff(w) = 1 / sigma(w) + S(w)
#ifndef GE_F64_COEFF
#define GE_F64_COEFF 0  // diagnostic variant (not shipped): the float64 buffer holds coeff(w) = 1 / sigma(w) + S(w) instead of S(w) -- one LDS read per
                        // successor instead of three -- and the reduction recovers S as coeff - 1 / sigma, which is S up to the rounding
                        // of that one addition (exact for a node without successors)
#endif
template <int BASE>
GE_DEV double ge_f64_sum_coeff_half(const uint8_t *sigrow, const double *Srow, const double *inv, uint32_t b, double acc) {
#if GE_F64_COEFF
  while (b) {
    int ix[GE_F64_U]; double sw[GE_F64_U];
    ge_f64_next_bits<BASE>(b, ix);
#pragma unroll
    for (int j = 0; j < GE_F64_U; j++) sw[j] = Srow[ix[j]];
#pragma unroll
    for (int j = 0; j < GE_F64_U; j++) acc += sw[j];
  }
  return acc;
#endif
  while (b) {
    int ix[GE_F64_U]; uint32_t sg[GE_F64_U]; double sw[GE_F64_U], iv[GE_F64_U];
    ge_f64_next_bits<BASE>(b, ix);
#pragma unroll
    for (int j = 0; j < GE_F64_U; j++) { sg[j] = sigrow[ix[j]]; sw[j] = Srow[ix[j]]; }
#pragma unroll
    for (int j = 0; j < GE_F64_U; j++) iv[j] = inv[sg[j]];  // (every count of a slot that is pulled is below GE_F64_INV)
#pragma unroll
    for (int j = 0; j < GE_F64_U; j++) acc += iv[j] + sw[j];
  }
  return acc;
}

// Betweenness and closeness of slot `env` (one workgroup, four waves).
GE_DEV void ge_f64_walk_env(const GeParams &P, int env, int *ovf_flag, int32_t *work_count, int32_t *work_list, int env_global) {
  const int tid = ge_tid_fresh();
  const int lane = tid & (GE_WAVE - 1), wv = tid >> 6;
  const int n = P.n, F = P.F;
  const ge_buffers &G = P.buf;
  const int64_t nbase = (int64_t)env * n;
  unsigned char *const sm = ge_dyn_smem();
  uint64_t *const abits = (uint64_t *)sm;
  double *const inv = (double *)(sm + GE_F64_W_INV), *const clos = (double *)(sm + GE_F64_W_CLOS);
  unsigned char *const area = sm + GE_F64_W_AREA + wv * GE_F64_A_BYTES;
  double *const S = (double *)(area + GE_F64_A_S);
  uint8_t *const sig = (uint8_t *)(area + GE_F64_A_SIG);
  uint64_t *const lvl = (uint64_t *)(area + GE_F64_A_LVL);
  uint8_t *const ord = (uint8_t *)(area + GE_F64_A_ORD), *const lst = (uint8_t *)(area + GE_F64_A_LST);
  GE_STAMP(11);
  if (wv == 0) abits[lane] = lane < n ? G.adj_bits[nbase + lane] : 0ull;
  if (tid == 0) *ovf_flag = 0;
  for (int i = tid; i < GE_F64_INV; i += GE_F64_THREADS) inv[i] = i > 0 ? 1.0 / (double)i : 0.0;  // correctly rounded reciprocals; [0]: the zero entry of a row
  ge_sync();
  GE_STAMP(12);
  const int sl = lane >> 3, o = lane & 7;  // source of the sub-batch, lane of its octet
  const uint64_t below_o = (1ull << (8 * o)) - 1ull;  // the nodes of the octet lanes below this one
  uint8_t *const sigrow = sig + sl * GE_F64_SG; double *const Srow = S + sl * GE_F64_SS;
  uint8_t *const ordrow = ord + sl * 64, *const lstrow = lst + sl * GE_F64_LST;
  double bc = 0.0;  // wave 0: betweenness of node `lane`, sources added in order
  bool ovf = false;
  for (int r = 0; r < 2; r++) {
    const int s = 32 * r + GE_F64_SB * wv + sl;  // this octet's source
    const bool src = s < n;
    // ---- (1) levels: lvl[d][sl] = the nodes at distance d from s; ord / lst = the same nodes in (level, node) order
    uint64_t vis = src ? (1ull << s) : 0ull, cur = vis;
    int D = 0, reach = 1, start = src ? 1 : 0, Dw = 0; int64_t tot = 0;
    if (o == 0) { lvl[sl] = cur; lstrow[0] = 0; lstrow[1] = (uint8_t)start; if (src) ordrow[0] = (uint8_t)s; }
    // the adjacency rows of this lane's eight nodes (registers during the level search only: the pulls need the room)
    uint32_t rlo[8], rhi[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { const uint64_t rw = abits[8 * o + j]; rlo[j] = (uint32_t)rw; rhi[j] = (uint32_t)(rw >> 32); }
    for (int d = 1; !(GE_F64_ABL & 32); d++) {
      const uint32_t cb = (uint32_t)(cur >> (8 * o)) & 0xffu;  // which of my eight nodes are in the current level
      uint32_t nlo = 0u, nhi = 0u;
#pragma unroll
      for (int j = 0; j < 8; j++) { const uint32_t m = 0u - ((cb >> j) & 1u); nlo |= m & rlo[j]; nhi |= m & rhi[j]; }
      nlo = ge_oct_or32(nlo); nhi = ge_oct_or32(nhi);
      const uint64_t nw = (((uint64_t)nhi << 32) | (uint64_t)nlo) & ~vis;  // (the same in the eight lanes)
      if (!ge_ballot(nw != 0ull)) break;  // no search of this wave found a new node
      Dw = d;
      vis |= nw; cur = nw;
      if (nw) {
        const int cnt = ge_popc64(nw);
        D = d; reach += cnt; tot += (int64_t)d * cnt;
        if (d < GE_F64_LV) {
          if (o == 0) lvl[d * GE_F64_SB + sl] = nw;
          int pos = start + ge_popc64(nw & below_o);
          for (uint32_t nb = (uint32_t)(nw >> (8 * o)) & 0xffu; nb; nb &= nb - 1u) ordrow[pos++] = (uint8_t)(8 * o + __builtin_ctz(nb));
        } else ovf = true;
        start += cnt;
      }
      if (o == 0 && d + 1 < GE_F64_LST) lstrow[d + 1] = (uint8_t)start;  // (a search that ended earlier: an empty level)
    }
    if (o == 0 && src) {  // closeness_centrality, wf_improved
      double cl = 0.0;
      if (tot > 0 && n > 1) { cl = ((double)reach - 1.0) / (double)tot; cl *= ((double)reach - 1.0) / (double)(n - 1); }
      clos[s] = cl;
    }
    if (r > 0) ge_sync();  // wave 0 has added the previous round's dependencies: S and sigma may be rewritten
    // ---- zero the round's counts and dependency sums (a node no search reaches keeps 0 x 0.0)
    for (int i = lane; i < GE_F64_SB * GE_F64_SG / 4; i += GE_WAVE) ((uint32_t *)sig)[i] = 0u;
    for (int i = lane; i < GE_F64_SB * GE_F64_SS; i += GE_WAVE) S[i] = 0.0;
    ge_wave_sync();
    if (o == 0 && src) sigrow[s] = 1;
    // level 1 is the source's neighbours: one path each, no pull (a trip of the loop below for every search otherwise)
    { const int lo = lstrow[1], hi = Dw >= 1 ? (int)lstrow[2] : lo; for (int k = lo + o; k < hi; k += 8) sigrow[ordrow[k]] = 1; }  // (Dw == 0: no search of this wave has a level 1, lstrow[2] was not written)
    const bool deep = ge_ballot(ovf) != 0ull;  // the slot goes to the generic kernel: skip the pulls (their level arrays would overrun)
    ge_wave_sync();
    // ---- (2) path counts: sigma(v) = sum of sigma(u) over v's neighbours u in the level above
    for (int d = 2; d <= Dw && !deep && !(GE_F64_ABL & 1); d++) {
      const int lo = lstrow[d], hi = lstrow[d + 1];
      const uint64_t prev = lvl[(d - 1) * GE_F64_SB + sl];
      for (int k = lo + o; ge_ballot(k < hi) != 0ull; k += 8) {
        if (k < hi) {
          const int v = ordrow[k];
          const uint64_t p = abits[v] & prev;
          const uint32_t acc = ge_f64_sum_counts(sigrow, p);
          sigrow[v] = (uint8_t)(acc < 255u ? acc : 255u);
          if (acc >= (uint32_t)GE_F64_INV) ovf = true;  // no reciprocal in the table: the generic kernel takes the slot
        }
      }
      ge_wave_sync();
    }
    // ---- (3) dependencies, deepest level first: S(v) = sum over v's neighbours w one level down of coeff(w),
    // coeff(w) = (1 + delta(w)) / sigma(w) with delta(w) = sigma(w) S(w), i.e. 1 / sigma(w) + S(w): the reciprocal of the (small,
    // integer) path count comes from the LDS table, one float64 add instead of a multiply, an add and a division
    const bool deep2 = ge_ballot(ovf) != 0ull;
#if GE_F64_COEFF
    if (!deep2) for (int i = lane; i < GE_F64_SB * 64; i += GE_WAVE) { const int a = (i >> 6) * GE_F64_SS + (i & 63), g = (i >> 6) * GE_F64_SG + (i & 63); S[a] = inv[sig[g] < GE_F64_INV ? sig[g] : 0]; }
    ge_wave_sync();
#endif
    for (int d = Dw - 1; d >= 1 && !deep2 && !(GE_F64_ABL & 2); d--) {
      const bool mine = d < D;  // (level D of this search has no successors: S stays 0)
      const int lo = lstrow[d], hi = mine ? lstrow[d + 1] : lo;
      const uint64_t next = lvl[(d + 1) * GE_F64_SB + sl];
      for (int k = lo + o; ge_ballot(k < hi) != 0ull; k += 8) {
        if (k < hi) {
          const int v = ordrow[k];
          const uint64_t sc = abits[v] & next;
          double acc = ge_f64_sum_coeff_half<0>(sigrow, Srow, inv, (uint32_t)sc, 0.0);
          acc = ge_f64_sum_coeff_half<32>(sigrow, Srow, inv, (uint32_t)(sc >> 32), acc);
#if GE_F64_COEFF
          acc += inv[sigrow[v]];
#endif
          Srow[v] = acc;
        }
      }
      ge_wave_sync();
    }
    if (ovf) *ovf_flag = 1;
    ge_sync();  // every wave's S and sigma of this round are complete
    // ---- (4) betweenness[v] += delta_s(v) = sigma_s(v) S_s(v), s != v, source after source (a node a search did not reach, the
    // source itself and a source that does not exist add +0.0)
    if (wv == 0 && !(GE_F64_ABL & 16)) {
#pragma unroll 1
      for (int w = 0; w < GE_F64_WAVES; w++) {
        const unsigned char *const aw = sm + GE_F64_W_AREA + w * GE_F64_A_BYTES;
        const uint8_t *const sgw = (const uint8_t *)(aw + GE_F64_A_SIG) + lane; const double *const Sw = (const double *)(aw + GE_F64_A_S) + lane;
        uint32_t sg[GE_F64_SB]; double sv[GE_F64_SB];
#pragma unroll
        for (int k = 0; k < GE_F64_SB; k++) { sg[k] = sgw[GE_F64_SG * k]; sv[k] = Sw[GE_F64_SS * k]; }
#if GE_F64_COEFF
#pragma unroll
        for (int k = 0; k < GE_F64_SB; k++) sv[k] -= inv[sg[k] < GE_F64_INV ? sg[k] : 0u];
#endif
#pragma unroll
        for (int k = 0; k < GE_F64_SB; k++) bc += (32 * r + GE_F64_SB * w + k != lane) ? (double)sg[k] * sv[k] : 0.0;
      }
    }
  }
  GE_STAMP(14);
  if (*ovf_flag) {  // uniform: hand the slot to the generic kernel (it recomputes all five columns)
    if (tid == 0) { int k = atomicAdd(&work_count[0], 1); work_list[k] = env_global; }
    ge_sync();
    return;
  }
  if (wv == 0 && lane < n) {
    if (n > 2) bc *= 1.0 / (double)((int64_t)(n - 1) * (int64_t)(n - 2));
    float *xr = G.x + (nbase + lane) * F + P.nflag;
    xr[1] = (float)bc; xr[2] = (float)clos[lane];
  }
  ge_sync();  // (the next item rewrites the LDS)
  GE_STAMP(17);
}

// Clustering, pagerank and degrees of slot `env`: ONE wave, a lane per node, on the wave's own LDS area (no workgroup barrier).
GE_DEV void ge_f64_node_env(const GeParams &P, int env, unsigned char *area, int lane) {
  const int n = P.n, E = P.E, F = P.F, t = P.env_type;
  const ge_buffers &G = P.buf;
  const int64_t nbase = (int64_t)env * n, ebase = (int64_t)env * E;
  const bool prw = (t == GE_TSP);
  uint64_t *const abits = (uint64_t *)area;
  double *const cx = (double *)(area + 512), *const cy = cx + 64, *const csinv = cx + 128, *const cdiff = cx + 192;
  uint8_t *const scode = area + 512 + 4 * 512;
  const bool live = lane < n;
  const uint64_t adj = live ? G.adj_bits[nbase + lane] : 0ull;
  const int deg = ge_popc64(adj);
  abits[lane] = adj;
  if (prw) for (int i = lane; i < E; i += GE_WAVE) scode[i] = G.scode[ebase + i];
  ge_wave_sync();
  double clus = 0.0, x = 0.0;
  // clustering (directed formula on the symmetric graph)
  if (live && !(GE_F64_ABL & 8)) {
    int64_t common = 0;
    for (uint64_t r = adj; r; r &= r - 1) common += ge_popc64(adj & abits[ge_ctz64(r)]);
    const int64_t t8 = 8 * common, dt = 2 * (int64_t)deg, db = deg;
    clus = (t8 == 0) ? 0.0 : (double)t8 / (double)((dt * (dt - 1) - 2 * db) * 2);
  }
  // pagerank ([nx] _pagerank_scipy): x_new[i] = sum over in-neighbours j ascending of (sinv[j]*w_ji) * x[j]
  int rp = 0;  // row start in ascending-neighbour order = exclusive scan of degrees
  if (prw) { int incl = ge_wave_incl_scan(deg, lane); rp = incl - deg; }
  double S = 0.0;
  { int k = 0; for (uint64_t r = adj; r; r &= r - 1, k++) S += (prw ? ge_wlut(scode[rp + k]) : 1.0) * 1.0; }
  const double sinv = (S != 0.0) ? 1.0 / S : 0.0;
  const double pinit = 1.0 / (double)n;
  const double alpha = 0.85, oma = 1 - alpha, tol = 1.0e-6;
  const uint64_t dangling = ge_ballot(live && deg == 0);
  x = pinit;
  csinv[lane] = sinv;
  // the 16 smallest neighbours as byte indices in four registers: the pull of an iteration becomes 16 unrolled (predicated) LDS
  // reads and adds in ascending-neighbour order instead of a ctz / clear-lowest-bit loop of max-degree trips; `rest` = what is left
  // of a row with more than 16 neighbours
  uint32_t nb4[4] = {0u, 0u, 0u, 0u}; uint64_t rest = adj;
  { int k = 0; for (; rest && k < 16; rest &= rest - 1, k++) nb4[k >> 2] |= (uint32_t)ge_ctz64(rest) << (8 * (k & 3)); }
  bool conv = false;
  for (int it = 0; it < ((GE_F64_ABL & 4) ? 1 : 100) && !conv; it++) {
    cx[lane] = x; cy[lane] = sinv * x;  // unweighted: data'[j->i] * x[j] is the same product for every i
    ge_wave_sync();
    double dsum = 0.0;
    { bool first = true; for (uint64_t r = dangling; r; r &= r - 1) { double xv = cx[ge_ctz64(r)]; dsum = first ? xv : dsum + xv; first = false; } }
    double xn = 0.0;
    if (live) {
      double acc = 0.0;
      if (prw) { int k = 0; for (uint64_t r = adj; r; r &= r - 1, k++) { const int j = ge_ctz64(r); acc += (csinv[j] * ge_wlut(scode[rp + k])) * cx[j]; } }
      else {
#pragma unroll
        for (int k = 0; k < 16; k++) { const double yv = cy[(nb4[k >> 2] >> (8 * (k & 3))) & 63u]; acc += (k < deg) ? yv : 0.0; }  // x + 0.0 == x
        for (uint64_t r = rest; r; r &= r - 1) acc += cy[ge_ctz64(r)];
      }
      xn = alpha * (acc + dsum * pinit) + oma * pinit;
    }
    cdiff[lane] = live ? __builtin_fabs(xn - x) : 0.0;
    ge_wave_sync();
    const double err = ge_pw<0>(cdiff, n, lane);
    x = xn;
    ge_wave_sync();
    if (err < (double)n * tol) conv = true;
  }
  if (live) {
    float *xr = G.x + (nbase + lane) * F + P.nflag;
    xr[0] = (float)(2.0 * (double)deg); xr[3] = (float)x; xr[4] = (float)clus;
  }
  ge_wave_sync();
}


// the class of a global slot (multi-class engine), as a wave-uniform value
GE_DEV int ge_slot_class(const GeRagged &R, int env) { return (int)ge_uniform_u32((uint32_t)R.slot_class[env]); }

// run.items GE_ITEMS_ALL: every slot; GE_ITEMS_QUEUE: the slots of P.buf.reset_list; GE_ITEMS_LIST: work_list (fallback of the
// fast path).  pre_off: byte offset of the queue prefix inside the dynamic LDS (behind the largest class's scratch in a multi-class
// engine).
#ifndef GE_GEN_WPS
#define GE_GEN_WPS 0  // waves per SIMD the register allocation of the generic feature kernel is held to (0: the compiler's choice)
#endif
#if GE_GEN_WPS > 0
#define GE_GEN_KERNEL GE_KERNEL_LB(512, GE_GEN_WPS)
#else
#define GE_GEN_KERNEL GE_KERNEL
#endif
template <bool RAGGED>
GE_GEN_KERNEL ge_k_features(GeParams P, GeRagged R, GeRun run, int pre_off, int bucket) {
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

// n <= 64 fast path over every slot / the queue: walk items [0, count), then node items (four slots each).  Multi-class engine: a
// slot of a class with n > 64 goes straight to work_list.  A workgroup takes the items bid, bid + grid, ...; in queue mode it looks
// up the slots of GE_F64_ITEMS of them at a time (the queue prefix overlays the item bodies in LDS and is rebuilt per chunk).
#ifndef GE_F64_WPS
#define GE_F64_WPS 6  // waves per SIMD the register allocation of the n <= 64 feature kernel is held to (0: the compiler's choice)
#endif
#if GE_F64_WPS > 0
#define GE_F64_KERNEL GE_KERNEL_LB(GE_F64_THREADS, GE_F64_WPS)
#else
#define GE_F64_KERNEL GE_KERNEL
#endif
template <bool RAGGED>
GE_F64_KERNEL ge_k_features64(GeParams P, GeRagged R, GeRun run, int tail_off) {
  const int nblk = (P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK;
  int *const pre = (int *)ge_dyn_smem();
  int *const items = (int *)(ge_dyn_smem() + tail_off);  // [GE_F64_ITEMS][4] slots (walk item: the first; -1 = none)
  int *const ovf_flag = items + GE_F64_ITEMS * 4;
  const bool queue = run.items == GE_ITEMS_QUEUE;
  int count = P.B;
  for (int base = 0;; base += GE_F64_ITEMS) {
    const int tid = ge_tid_fresh();
    if (queue) {  // the prefix scan is one wave wide
      if (tid < GE_WAVE) ge_queue_prefix_wave(P, pre, tid);
      ge_sync();
      count = pre[nblk];
    }
    const int total = count + (count + 3) / 4;
    if (ge_bid() + (int64_t)base * ge_gdim() >= total) break;  // (uniform)
    if (tid < GE_F64_ITEMS * 4) {
      const int64_t q = ge_bid() + (int64_t)(base + (tid >> 2)) * ge_gdim();
      const int sub = tid & 3;
      int qi = -1;
      if (q < count) qi = sub == 0 ? (int)q : -1;
      else if (q < total) { const int64_t k = 4 * (q - count) + sub; qi = k < count ? (int)k : -1; }
      items[tid] = qi < 0 ? -1 : (queue ? ge_queue_slot(P, pre, qi) : qi);
    }
    ge_sync();
    for (int i = 0; i < GE_F64_ITEMS; i++) {
      const int64_t q = ge_bid() + (int64_t)(base + i) * ge_gdim();
      if (q >= total) break;  // (uniform)
      if (q < count) {  // walk item
        const int env = items[4 * i];
        if (queue && ge_tid() == 0) ge_finish_item(P, run, env);  // seed[] / episode[] now name the new episode (refill: the image is valid)
        if constexpr (RAGGED) {
          const int cls = ge_slot_class(R, env);
          const GeParams &C = R.classes[cls];
          if (C.n > 64) {  // uniform: the generic kernel takes the slot (global id)
            if (ge_tid() == 0) { int k = atomicAdd(&P.buf.work_count[0], 1); P.buf.work_list[k] = env; }
            continue;
          }
          ge_f64_walk_env(C, env - R.class_start[cls], ovf_flag, P.buf.work_count, P.buf.work_list, env);
        } else {
          ge_f64_walk_env(P, env, ovf_flag, P.buf.work_count, P.buf.work_list, env);
        }
      } else {  // node item: a slot per wave
        const int t2 = ge_tid_fresh();
        const int wv = t2 >> 6, lane = t2 & (GE_WAVE - 1);
        const int env = items[4 * i + wv];
        if (env >= 0) {
          if constexpr (RAGGED) {
            const int cls = ge_slot_class(R, env);
            const GeParams &C = R.classes[cls];
            if (C.n <= 64) ge_f64_node_env(C, env - R.class_start[cls], ge_dyn_smem() + wv * ge_f64_node_wave_bytes(C.E, 0), lane);
          } else {
            ge_f64_node_env(P, env, ge_dyn_smem() + wv * ge_f64_node_wave_bytes(P.E, P.env_type == GE_TSP), lane);
          }
        }
        ge_sync();  // (the next item, or the next chunk's queue prefix, rewrites the LDS)
      }
    }
    ge_sync();
  }
}
