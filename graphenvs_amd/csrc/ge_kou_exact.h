/* SteinerTree baseline (is_eval_env, 1 < n_dests < n - 1): steiner_tree.py:84-87,
 *     T = nx.algorithms.approximation.steinertree.steiner_tree(G, self.dests, weight='delay', method='kou')
 *     approx_solution = sum([G[u][v]['delay'] for u, v in T.edges()])
 * EXACTLY as networkx 3.4.2 on CPython 3.10 computes it, float64 sum order included.  What decides the value ([nx] = networkx):
 *   - [nx] _dijkstra_multisource: heap order (distance, push counter), a path is replaced only by a strictly shorter one;
 *   - [nx] metric_closure: edge (u, v), u < v, carries the distance and the path FROM u;
 *   - [nx] kruskal_mst_edges: a stable sort by weight over G.edges(data=True) of a subgraph VIEW, whose node order is the iteration
 *     order of the view's node SET when that holds less than half of the parent's nodes ([nx] FilterAtlas / FilterAdjacency);
 *   - the second spanning tree is asked for weight="weight", an attribute these graphs do not have: every edge weighs 1 and the
 *     tree is the first acyclic edges in view order;
 *   - [nx] edge_subgraph: set(edges) of (u, v) tuples, nodes.update(e[:2]) over it, set(nodes) -- CPython set tables of tuples and of
 *     ints ([py] setobject.c, tupleobject.c's hash);
 *   - T_H = G.edge_subgraph(T_S).copy() re-inserts adjacency rows (earlier nodes first), non-terminal leaves are pruned, and the
 *     final sum runs over the edges of yet another view.
 * Plain C, one sequential thread per slot, no recursion, no allocation (arena of ge_clique_removal.h); engine and CPU checker
 * compile this one text, tests/ pin it against networkx itself on random graphs and against the reference's fixtures. */
#ifndef GE_KOU_EXACT_H
#define GE_KOU_EXACT_H
#include "ge_clique_removal.h"

/* [py] tuplehash of a 2-tuple of small non-negative ints (Objects/tupleobject.c, xxHash-derived; hash(int) == int) */
GE_CR_FN uint64_t ge_py_tuple2_hash(uint64_t a, uint64_t b) {
  const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
  uint64_t acc = P5;
  acc += a * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
  acc += b * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
  acc += 2ULL ^ (P5 ^ 3527539ULL);
  return acc == (uint64_t)-1 ? 1546275796ULL : acc;
}

/* ------------------------------------------------------------------ set of (u, v) tuples: add, membership, table order */
typedef struct { int32_t *key; uint64_t *hash; int32_t mask, fill, used; } ge_pyset2;  /* key = u * 65536 + v, -1 = unused */

GE_CR_FN void ge_pyset2_alloc(ge_cr_arena *a, ge_pyset2 *s, int32_t size) {
  s->key = (int32_t *)ge_cr_alloc(a, (uint64_t)size * 4);
  s->hash = (uint64_t *)ge_cr_alloc(a, (uint64_t)size * 8);
  s->mask = size - 1;
  if (!a->err) for (int32_t i = 0; i < size; i++) s->key[i] = -1;
}
GE_CR_FN void ge_pyset2_init(ge_cr_arena *a, ge_pyset2 *s) { ge_pyset2_alloc(a, s, 8); s->fill = 0; s->used = 0; }
GE_CR_FN void ge_pyset2_insert_clean(ge_pyset2 *s, int32_t key, uint64_t hash) {
  uint64_t perturb = hash; uint32_t i = (uint32_t)hash & (uint32_t)s->mask;
  for (;;) {
    if (s->key[i] < 0) { s->key[i] = key; s->hash[i] = hash; return; }
    if (i + 9 <= (uint32_t)s->mask) for (int j = 1; j <= 9; j++) if (s->key[i + j] < 0) { s->key[i + j] = key; s->hash[i + j] = hash; return; }
    perturb >>= 5;
    i = (uint32_t)(((uint64_t)i * 5 + 1 + perturb) & (uint64_t)(uint32_t)s->mask);
  }
}
GE_CR_FN int ge_pyset2_has(const ge_pyset2 *s, int u, int v) {
  const int32_t key = u * 65536 + v; const uint64_t hash = ge_py_tuple2_hash((uint64_t)u, (uint64_t)v);
  uint64_t perturb = hash; uint32_t i = (uint32_t)hash & (uint32_t)s->mask;
  for (;;) {
    int probes = (i + 9 <= (uint32_t)s->mask) ? 9 : 0; uint32_t e = i;
    do { if (s->key[e] < 0) return 0; if (s->hash[e] == hash && s->key[e] == key) return 1; e++; } while (probes--);
    perturb >>= 5;
    i = (uint32_t)(((uint64_t)i * 5 + 1 + perturb) & (uint64_t)(uint32_t)s->mask);
  }
}
GE_CR_FN void ge_pyset2_add(ge_cr_arena *a, ge_pyset2 *s, int u, int v) {
  const int32_t key = u * 65536 + v; const uint64_t hash = ge_py_tuple2_hash((uint64_t)u, (uint64_t)v);
  uint64_t perturb = hash; uint32_t i = (uint32_t)hash & (uint32_t)s->mask;
  for (;;) {
    int probes = (i + 9 <= (uint32_t)s->mask) ? 9 : 0; uint32_t e = i;
    do {
      if (s->key[e] < 0) {
        s->key[e] = key; s->hash[e] = hash; s->fill++; s->used++;
        if ((int64_t)s->fill * 5 >= (int64_t)s->mask * 3) {  /* set_table_resize(used > 50000 ? used * 2 : used * 4) */
          const int32_t minused = s->used > 50000 ? s->used * 2 : s->used * 4;
          int32_t newsize = 8; while (newsize <= minused) newsize <<= 1;
          ge_pyset2 t; ge_pyset2_alloc(a, &t, newsize);
          if (a->err) return;
          for (int32_t k = 0; k <= s->mask; k++) if (s->key[k] >= 0) ge_pyset2_insert_clean(&t, s->key[k], s->hash[k]);
          s->key = t.key; s->hash = t.hash; s->mask = t.mask; s->fill = s->used;
        }
        return;
      }
      if (s->hash[e] == hash && s->key[e] == key) return;
      e++;
    } while (probes--);
    perturb >>= 5;
    i = (uint32_t)(((uint64_t)i * 5 + 1 + perturb) & (uint64_t)(uint32_t)s->mask);
  }
}

/* ------------------------------------------------------------------ the graph and small helpers */
typedef struct {
  int n, m, T;
  const int32_t *off;   /* [n+1] insertion-order adjacency of nodes 0 .. n-1 */
  const uint16_t *adj;  /* [2m] */
  const double *w;      /* [2m] G[u][v]['delay'] of every directed entry */
  const int32_t *terms; /* [T] self.dests: the source, then the destinations, as np.random.choice drew them */
} ge_kou_in;

GE_CR_FN double ge_kou_w(const ge_kou_in *g, int a, int b) {
  for (int32_t e = g->off[a]; e < g->off[a + 1]; e++) if (g->adj[e] == b) return g->w[e];
  return 0.0;
}
GE_CR_FN int ge_kou_bit(const uint64_t *bits, int v) { return (int)((bits[v >> 6] >> (v & 63)) & 1ull); }
GE_CR_FN int ge_kou_find(uint16_t *uf, int x) { while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; } return x; }

/* G.edge_subgraph(edges): E = set(edges) is given; N = set(); for e in E: N.update(e[:2]); S = set(N) (show_nodes).  Fills sbits
   (membership) and order[] = the node order of the view over G ([nx] FilterAtlas.__iter__: the set's iteration order when it holds
   less than half of G's nodes, else G's own order); returns |S|. */
GE_CR_FN int32_t ge_kou_view_nodes(ge_cr_arena *a, int n, const ge_pyset2 *E, uint64_t *sbits, uint16_t *order) {
  const uint64_t mark = a->top;
  ge_pyset N, S;
  ge_pyset_init(a, &N);
  for (int32_t i = 0; i <= E->mask && !a->err; i++) if (E->key[i] >= 0) { ge_pyset_add(a, &N, E->key[i] >> 16); ge_pyset_add(a, &N, E->key[i] & 65535); }
  ge_pyset_copy(a, &N, &S);
  for (int w = 0; w < (n + 63) / 64; w++) sbits[w] = 0;
  int32_t cnt = 0;
  if (!a->err) {
    for (int32_t i = 0; i <= S.mask; i++) if (S.tab[i] >= 0) { sbits[S.tab[i] >> 6] |= 1ull << (S.tab[i] & 63); if (2 * S.used < n) order[cnt++] = (uint16_t)S.tab[i]; }
    if (2 * S.used >= n) for (int v = 0; v < n; v++) if (ge_kou_bit(sbits, v)) order[cnt++] = (uint16_t)v;
  }
  a->top = mark;
  return cnt;
}

/* edges of the view G.edge_subgraph(E) as EdgeView yields them: rows in view order, columns in G's adjacency order, an edge at its
   first endpoint */
GE_CR_FN int32_t ge_kou_view_edges(const ge_kou_in *g, const ge_pyset2 *E, const uint64_t *sbits, const uint16_t *order, int32_t cnt,
                                   uint64_t *seen, uint16_t *ea, uint16_t *eb) {
  for (int w = 0; w < (g->n + 63) / 64; w++) seen[w] = 0;
  int32_t k = 0;
  for (int32_t i = 0; i < cnt; i++) {
    const int a = order[i];
    for (int32_t e = g->off[a]; e < g->off[a + 1]; e++) {
      const int b = g->adj[e];
      if (ge_kou_bit(sbits, b) && !ge_kou_bit(seen, b) && (ge_pyset2_has(E, a, b) || ge_pyset2_has(E, b, a))) { ea[k] = (uint16_t)a; eb[k] = (uint16_t)b; k++; }
    }
    seen[a >> 6] |= 1ull << (a & 63);
  }
  return k;
}

/* bytes of arena the call below needs at most (T terminals) */
GE_CR_HD uint64_t ge_kou_arena_bytes(int n, int m, int T) {
  const uint64_t pairs = (uint64_t)T * (uint64_t)(T - 1) / 2;
  uint64_t tuples = 8; while (tuples <= 8ull * (uint64_t)(n + 2)) tuples <<= 1;   /* a tuple set of tree edges: at most n - 1 */
  uint64_t chain = 8; while (chain <= 8ull * (uint64_t)(2 * m + 2)) chain <<= 1;  /* the first one: the closure paths' edges, at most 2m */
  const uint64_t bytes = (uint64_t)T * (uint64_t)n * 2 + (uint64_t)T * (uint64_t)T * 8 + pairs * 20 + (uint64_t)(2 * m + 2) * 18 + (chain + 2 * tuples) * 24 + (uint64_t)n * 264 + 8192;
  return (bytes + 15u) & ~(uint64_t)15u;  /* slot blocks follow one another: keep them aligned */
}

/* the value; *err = 1 when the arena was too small */
GE_CR_FN double ge_kou_exact(const ge_kou_in *g, ge_cr_arena *a, int *err) {
  const int n = g->n, T = g->T, W = (n + 63) / 64;
  *err = 0;
  uint64_t *tbits = (uint64_t *)ge_cr_alloc(a, (uint64_t)W * 8), *sbits = (uint64_t *)ge_cr_alloc(a, (uint64_t)W * 8), *seen = (uint64_t *)ge_cr_alloc(a, (uint64_t)W * 8);
  uint16_t *ts = (uint16_t *)ge_cr_alloc(a, (uint64_t)T * 2);          /* terminals ascending */
  uint16_t *tix = (uint16_t *)ge_cr_alloc(a, (uint64_t)n * 2);         /* node -> rank among the terminals */
  uint16_t *par = (uint16_t *)ge_cr_alloc(a, (uint64_t)T * n * 2);     /* par[i][v]: predecessor of v on the path from ts[i] */
  double *dist = (double *)ge_cr_alloc(a, (uint64_t)n * 8), *sn = (double *)ge_cr_alloc(a, (uint64_t)n * 8);
  uint8_t *state = (uint8_t *)ge_cr_alloc(a, (uint64_t)n);             /* 0 unseen, 1 seen, 2 final */
  const int32_t hcap = 2 * g->m + 2;
  double *hd = (double *)ge_cr_alloc(a, (uint64_t)hcap * 8); int32_t *hc = (int32_t *)ge_cr_alloc(a, (uint64_t)hcap * 4); uint16_t *hv = (uint16_t *)ge_cr_alloc(a, (uint64_t)hcap * 2);
  const int32_t pairs = T * (T - 1) / 2;
  double *D = (double *)ge_cr_alloc(a, (uint64_t)T * T * 8);           /* D[i][j], i < j (ranks): distance from ts[i] */
  double *pd = (double *)ge_cr_alloc(a, (uint64_t)pairs * 8); uint16_t *pa = (uint16_t *)ge_cr_alloc(a, (uint64_t)pairs * 2), *pb = (uint16_t *)ge_cr_alloc(a, (uint64_t)pairs * 2);
  int32_t *idx = (int32_t *)ge_cr_alloc(a, (uint64_t)pairs * 4), *idx2 = (int32_t *)ge_cr_alloc(a, (uint64_t)pairs * 4);
  uint16_t *uf = (uint16_t *)ge_cr_alloc(a, (uint64_t)n * 2), *order = (uint16_t *)ge_cr_alloc(a, (uint64_t)n * 2), *stk = (uint16_t *)ge_cr_alloc(a, (uint64_t)n * 2);
  uint16_t *ea = (uint16_t *)ge_cr_alloc(a, (uint64_t)(2 * g->m + 2) * 2), *eb = (uint16_t *)ge_cr_alloc(a, (uint64_t)(2 * g->m + 2) * 2);
  if (a->err) { *err = 1; return 0.0; }

  for (int w = 0; w < W; w++) tbits[w] = 0;
  for (int i = 0; i < T; i++) tbits[g->terms[i] >> 6] |= 1ull << (g->terms[i] & 63);
  { int k = 0; for (int v = 0; v < n; v++) if (ge_kou_bit(tbits, v)) { tix[v] = (uint16_t)k; ts[k++] = (uint16_t)v; } }

  /* ---- [nx] all_pairs_dijkstra rows of the terminals (the largest one's row is never used): binary heap on (distance, counter) */
  for (int i = 0; i + 1 < T; i++) {
    const int src = ts[i];
    uint16_t *pr = par + (uint64_t)i * n;
    for (int v = 0; v < n; v++) state[v] = 0;
    int32_t hn = 0, counter = 0, left = T - 1 - i;  /* terminals above src still to be settled */
    hd[0] = 0.0; hc[0] = counter++; hv[0] = (uint16_t)src; hn = 1; state[src] = 1; sn[src] = 0.0; pr[src] = (uint16_t)src;
    while (hn > 0 && left > 0) {
      const double d = hd[0]; const int v = hv[0];
      /* pop: the last entry sifts down from the root */
      hn--;
      if (hn > 0) {
        const double xd = hd[hn]; const int32_t xc = hc[hn]; const uint16_t xv = hv[hn];
        int32_t p = 0;
        for (;;) {
          int32_t ch = 2 * p + 1;
          if (ch >= hn) break;
          if (ch + 1 < hn && (hd[ch + 1] < hd[ch] || (hd[ch + 1] == hd[ch] && hc[ch + 1] < hc[ch]))) ch++;
          if (!(hd[ch] < xd || (hd[ch] == xd && hc[ch] < xc))) break;
          hd[p] = hd[ch]; hc[p] = hc[ch]; hv[p] = hv[ch]; p = ch;
        }
        hd[p] = xd; hc[p] = xc; hv[p] = xv;
      }
      if (state[v] == 2) continue;
      state[v] = 2; dist[v] = d;
      if (ge_kou_bit(tbits, v) && v > src) { D[(uint64_t)i * T + tix[v]] = d; left--; }
      for (int32_t e = g->off[v]; e < g->off[v + 1]; e++) {
        const int u = g->adj[e];
        if (state[u] == 2) continue;
        const double vu = d + g->w[e];
        if (state[u] == 0 || vu < sn[u]) {
          state[u] = 1; sn[u] = vu; pr[u] = (uint16_t)v;
          int32_t p = hn++;
          if (hn > hcap) { *err = 1; return 0.0; }
          const int32_t c = counter++;
          while (p > 0) {
            const int32_t q = (p - 1) >> 1;
            if (!(vu < hd[q] || (vu == hd[q] && c < hc[q]))) break;
            hd[p] = hd[q]; hc[p] = hc[q]; hv[p] = hv[q]; p = q;
          }
          hd[p] = vu; hc[p] = c; hv[p] = (uint16_t)u;
        }
      }
    }
  }

  /* ---- H = M.subgraph(terminals): edges in EdgeDataView order, then [nx] kruskal_mst_edges(weight='distance') */
  int32_t np_ = 0;
  {
    const uint64_t mark = a->top;
    int32_t oc = 0;
    if (2 * T < n) {  /* the view's node set, built in self.dests order, iterated in table order */
      ge_pyset S; ge_pyset_init(a, &S);
      for (int i = 0; i < T && !a->err; i++) ge_pyset_add(a, &S, g->terms[i]);
      if (a->err) { *err = 1; return 0.0; }
      for (int32_t i = 0; i <= S.mask; i++) if (S.tab[i] >= 0) order[oc++] = (uint16_t)S.tab[i];
    } else for (int i = 0; i < T; i++) order[oc++] = ts[i];
    a->top = mark;
    for (int w = 0; w < W; w++) seen[w] = 0;
    for (int32_t i = 0; i < oc; i++) {
      const int x = order[i];
      for (int j = 0; j < T; j++) {  /* M.adj[x]: ascending */
        const int y = ts[j];
        if (y == x || ge_kou_bit(seen, y)) continue;
        const int lo = x < y ? x : y, hi = x < y ? y : x;
        pd[np_] = D[(uint64_t)tix[lo] * T + tix[hi]]; pa[np_] = (uint16_t)x; pb[np_] = (uint16_t)y; np_++;
      }
      seen[x >> 6] |= 1ull << (x & 63);
    }
  }
  /* sorted(edges, key=weight): stable; bottom-up merge sort of the index array */
  for (int32_t i = 0; i < np_; i++) idx[i] = i;
  for (int32_t width = 1; width < np_; width <<= 1) {
    for (int32_t lo = 0; lo < np_; lo += 2 * width) {
      int32_t mid = lo + width < np_ ? lo + width : np_, hi = lo + 2 * width < np_ ? lo + 2 * width : np_;
      int32_t i = lo, j = mid, k = lo;
      while (i < mid && j < hi) idx2[k++] = (pd[idx[j]] < pd[idx[i]]) ? idx[j++] : idx[i++];
      while (i < mid) idx2[k++] = idx[i++];
      while (j < hi) idx2[k++] = idx[j++];
    }
    { int32_t *t = idx; idx = idx2; idx2 = t; }
  }
  ge_pyset2 E1; ge_pyset2_init(a, &E1);
  for (int v = 0; v < n; v++) uf[v] = (uint16_t)v;
  for (int32_t q = 0; q < np_ && !a->err; q++) {
    const int x = pa[idx[q]], y = pb[idx[q]];
    const int rx = ge_kou_find(uf, x), ry = ge_kou_find(uf, y);
    if (rx == ry) continue;
    uf[rx] = (uint16_t)ry;
    /* pairwise(d['path']): the path runs from the smaller terminal to the larger one */
    const int lo = x < y ? x : y, hi = x < y ? y : x;
    const uint16_t *pr = par + (uint64_t)tix[lo] * n;
    int len = 0;
    for (int v = hi; v != lo; v = pr[v]) { if (len >= n) { *err = 1; return 0.0; } stk[len++] = (uint16_t)v; }
    int prev = lo;
    for (int k = len - 1; k >= 0; k--) { ge_pyset2_add(a, &E1, prev, stk[k]); prev = stk[k]; }
  }
  if (a->err) { *err = 1; return 0.0; }

  /* ---- G_S = G.edge_subgraph(those edges); [nx] kruskal with weight='weight': absent, so every edge weighs 1 -- view order decides */
  int32_t cnt = ge_kou_view_nodes(a, n, &E1, sbits, order);
  int32_t ne = ge_kou_view_edges(g, &E1, sbits, order, cnt, seen, ea, eb);
  if (a->err) { *err = 1; return 0.0; }
  ge_pyset2 E2; ge_pyset2_init(a, &E2);
  for (int v = 0; v < n; v++) uf[v] = (uint16_t)v;
  for (int32_t q = 0; q < ne && !a->err; q++) {
    const int rx = ge_kou_find(uf, ea[q]), ry = ge_kou_find(uf, eb[q]);
    if (rx == ry) continue;
    uf[rx] = (uint16_t)ry;
    ge_pyset2_add(a, &E2, ea[q], eb[q]);  /* T_S yields (u, v) as the view reported it */
  }
  if (a->err) { *err = 1; return 0.0; }

  /* ---- T_H = G.edge_subgraph(T_S).copy(): rows re-inserted (earlier nodes of the copy first, then G's adjacency order) */
  cnt = ge_kou_view_nodes(a, n, &E2, sbits, order);
  if (a->err) { *err = 1; return 0.0; }
  int16_t *pos = (int16_t *)ge_cr_alloc(a, (uint64_t)n * 2);
  int32_t *coff = (int32_t *)ge_cr_alloc(a, (uint64_t)(cnt + 1) * 4), *fill = (int32_t *)ge_cr_alloc(a, (uint64_t)(cnt + 1) * 4);
  uint16_t *cadj = (uint16_t *)ge_cr_alloc(a, (uint64_t)(2 * cnt + 2) * 2);  /* a tree: 2 (cnt - 1) entries */
  uint8_t *alive = (uint8_t *)ge_cr_alloc(a, (uint64_t)cnt + 1); int32_t *deg = (int32_t *)ge_cr_alloc(a, (uint64_t)(cnt + 1) * 4);
  if (a->err) { *err = 1; return 0.0; }
  for (int v = 0; v < n; v++) pos[v] = -1;
  for (int32_t i = 0; i < cnt; i++) pos[order[i]] = (int16_t)i;
  { int32_t tot = 0;
    for (int32_t i = 0; i < cnt; i++) {
      coff[i] = tot;
      const int x = order[i];
      for (int32_t e = g->off[x]; e < g->off[x + 1]; e++) { const int y = g->adj[e]; if (pos[y] >= 0 && (ge_pyset2_has(&E2, x, y) || ge_pyset2_has(&E2, y, x))) tot++; }
    }
    coff[cnt] = tot;
    if (tot > 2 * cnt) { *err = 1; return 0.0; } }
  for (int32_t i = 0; i < cnt; i++) fill[i] = coff[i];
  for (int32_t i = 0; i < cnt; i++) {
    const int x = order[i];
    for (int32_t e = g->off[x]; e < g->off[x + 1]; e++) {
      const int y = g->adj[e]; const int p = pos[y];
      if (p > i && (ge_pyset2_has(&E2, x, y) || ge_pyset2_has(&E2, y, x))) { cadj[fill[i]++] = (uint16_t)y; cadj[fill[p]++] = (uint16_t)x; }
    }
  }
  /* _remove_nonterminal_leaves: whatever the order of removal, the same nodes go; the dicts keep the order of what stays */
  for (int32_t i = 0; i < cnt; i++) { alive[i] = 1; deg[i] = coff[i + 1] - coff[i]; }
  { int32_t top = 0;
    for (int32_t i = 0; i < cnt; i++) if (deg[i] == 1 && !ge_kou_bit(tbits, order[i])) stk[top++] = (uint16_t)i;
    while (top > 0) {
      const int32_t i = stk[--top];
      if (!alive[i]) continue;
      alive[i] = 0;
      for (int32_t e = coff[i]; e < coff[i + 1]; e++) {
        const int32_t p = pos[cadj[e]];
        if (!alive[p]) continue;
        if (--deg[p] == 1 && !ge_kou_bit(tbits, order[p])) stk[top++] = (uint16_t)p;
      }
    } }
  /* return T_H.edges(): EdgeView of the copy */
  ge_pyset2 E3; ge_pyset2_init(a, &E3);
  for (int w = 0; w < W; w++) seen[w] = 0;
  for (int32_t i = 0; i < cnt && !a->err; i++) {
    if (!alive[i]) continue;
    const int x = order[i];
    for (int32_t e = coff[i]; e < coff[i + 1]; e++) { const int y = cadj[e]; if (alive[pos[y]] && !ge_kou_bit(seen, y)) ge_pyset2_add(a, &E3, x, y); }
    seen[x >> 6] |= 1ull << (x & 63);
  }
  if (a->err) { *err = 1; return 0.0; }

  /* ---- steiner_tree returns G.edge_subgraph(those); the env sums G[u][v]['delay'] over its edges, left to right from int 0 */
  cnt = ge_kou_view_nodes(a, n, &E3, sbits, order);
  ne = ge_kou_view_edges(g, &E3, sbits, order, cnt, seen, ea, eb);
  if (a->err) { *err = 1; return 0.0; }
  double total = 0.0;
  for (int32_t q = 0; q < ne; q++) total = total + ge_kou_w(g, ea[q], eb[q]);
  return total;
}

#endif
