/* TSP baseline (is_eval_env): Christofides' tour on the metric closure of a slot's graph -- the part that runs as ONE
 * sequential thread per slot.  Plain C (no HIP constructs): the device kernel (ge_tsp_eval.h) runs it with one lane per slot on
 * the slot's scratch block; the CPU checker compiles the same text, so the two agree by construction, and tests/ check the
 * algorithm itself (a perfect matching of minimum weight against a subset DP, tour <= 1.5 OPT against brute force).
 *
 * The reference calls nx.approximation.traveling_salesman_problem(G, weight='weight', cycle=True) (tsp.py:114-117): all-pairs
 * Dijkstra, Christofides on the complete graph of distances (minimum spanning tree, minimum-weight perfect matching of the
 * odd-degree nodes, Eulerian circuit, shortcutting), hops expanded back to paths of G.  Which tree, matching and circuit networkx
 * picks among equals depends on dict / set iteration orders deep inside the library; this file makes its own deterministic choices
 * (lowest index first), so the tour is A Christofides tour -- same guarantee, not the same tour.
 *
 * All weights are integers (weight codes in tenths, or fixed-point Euclidean lengths): the matching is exact, no tolerance.
 * The matching is Edmonds' blossom algorithm with dual variables (Galil's O(k^3) formulation) on the complete graph of the k odd
 * nodes, maximising BIG - d(u, v): every weight is positive, so a maximum-weight matching is perfect and has minimum total d. */
#ifndef GE_CHRISTOFIDES_H
#define GE_CHRISTOFIDES_H
#include <stdint.h>

#ifndef GE_CH_FN
#define GE_CH_FN static inline
#endif
#ifndef GE_CH_HD  /* the sizing helpers, which a host sizes buffers with as well */
#define GE_CH_HD GE_CH_FN
#endif

/* ------------------------------------------------------------------ scratch carving */
typedef struct {
  int n, k, NX, FS;      /* nodes; odd-degree nodes (even); row stride 2k+1 of the per-node tables; row stride k+2 of the flower tables */
  const int32_t *D;      /* [n*n] closure distances */
  /* spanning tree */
  int32_t *key; int16_t *par; uint8_t *intree; uint16_t *deg, *odd;
  /* matching (1-based: vertices 1..k, blossoms k+1..nx) */
  int nx;
  int32_t *W;            /* [(k+1)*(k+1)] */
  uint16_t *gu, *gv;     /* [NX*NX] endpoints (vertices) of the tightest known edge between two nodes; gu == 0: none */
  uint16_t *ff;          /* [NX*(k+1)] blossom b, vertex x -> the sub-node of b that holds x (0: x not in b) */
  uint16_t *fl, *fln;    /* [NX*FS] sub-nodes of a blossom round its cycle, base first; [NX] their number */
  int64_t *lab;          /* duals */
  uint16_t *match, *slack, *st, *pa; int8_t *S; int32_t *vis; int32_t vt;
  uint16_t *q; int qh, qt, qcap;
  uint16_t *stk, *tmp;
  /* circuit */
  int32_t *ehead; uint16_t *eto; int32_t *eid; uint8_t *eused; uint16_t *cstk, *circ; int32_t *eptr;
  int err;               /* 1: a capacity of this scratch was exceeded (never seen; the caller then keeps the double-tree walk) */
} ge_ch;

GE_CH_HD uint64_t ge_ch_align(uint64_t x) { return (x + 15u) & ~(uint64_t)15u; }

/* bytes of one slot's scratch behind its closure matrix (which is n*n*4 bytes); k is taken as its maximum, n */
GE_CH_HD uint64_t ge_ch_carve(ge_ch *c, uint8_t *base, int n) {
  const uint64_t K = (uint64_t)(n + (n & 1)), NX = 2 * K + 1, FS = K + 2, M2 = (uint64_t)n + K / 2 + 2;
  uint64_t off = 0;
#define GE_CH_TAKE(field, type, count) do { if (c) c->field = (type *)(base + off); off = ge_ch_align(off + sizeof(type) * (uint64_t)(count)); } while (0)
  GE_CH_TAKE(key, int32_t, n); GE_CH_TAKE(par, int16_t, n); GE_CH_TAKE(intree, uint8_t, n); GE_CH_TAKE(deg, uint16_t, n); GE_CH_TAKE(odd, uint16_t, K + 1);
  GE_CH_TAKE(W, int32_t, (K + 1) * (K + 1));
  GE_CH_TAKE(gu, uint16_t, NX * NX); GE_CH_TAKE(gv, uint16_t, NX * NX);
  GE_CH_TAKE(ff, uint16_t, NX * (K + 1));
  GE_CH_TAKE(fl, uint16_t, NX * FS); GE_CH_TAKE(fln, uint16_t, NX);
  GE_CH_TAKE(lab, int64_t, NX);
  GE_CH_TAKE(match, uint16_t, NX); GE_CH_TAKE(slack, uint16_t, NX); GE_CH_TAKE(st, uint16_t, NX); GE_CH_TAKE(pa, uint16_t, NX);
  GE_CH_TAKE(S, int8_t, NX); GE_CH_TAKE(vis, int32_t, NX);
  GE_CH_TAKE(q, uint16_t, 4 * NX); GE_CH_TAKE(stk, uint16_t, 4 * NX); GE_CH_TAKE(tmp, uint16_t, FS);
  GE_CH_TAKE(ehead, int32_t, n + 1); GE_CH_TAKE(eto, uint16_t, 2 * M2); GE_CH_TAKE(eid, int32_t, 2 * M2); GE_CH_TAKE(eused, uint8_t, M2);
  GE_CH_TAKE(cstk, uint16_t, M2 + 1); GE_CH_TAKE(circ, uint16_t, M2 + 1); GE_CH_TAKE(eptr, int32_t, n + 1);
#undef GE_CH_TAKE
  if (c) { c->n = n; c->qcap = (int)(4 * NX); }
  return off;
}

/* one slot's scratch block: the closure matrix, then the tables above */
GE_CH_HD uint64_t ge_ch_slot_bytes(int n) { return ge_ch_align((uint64_t)n * (uint64_t)n * 4u) + ge_ch_carve((ge_ch *)0, (uint8_t *)0, n); }

/* ------------------------------------------------------------------ maximum-weight matching (blossom algorithm with duals) */
#define GE_BL_G(c, a, b) ((uint32_t)(a) * (uint32_t)(c)->NX + (uint32_t)(b))
#define GE_BL_FL(c, b, i) ((c)->fl[(uint32_t)(b) * (uint32_t)(c)->FS + (uint32_t)(i)])
#define GE_BL_FF(c, b, x) ((c)->ff[(uint32_t)(b) * (uint32_t)((c)->k + 1) + (uint32_t)(x)])

/* slack of the recorded edge between nodes a and b: dual(u) + dual(v) - 2 w(u, v) over its endpoint vertices */
GE_CH_FN int64_t ge_bl_delta(const ge_ch *c, int a, int b) {
  const int u = c->gu[GE_BL_G(c, a, b)], v = c->gv[GE_BL_G(c, a, b)];
  return c->lab[u] + c->lab[v] - 2 * (int64_t)c->W[u * (c->k + 1) + v];
}

GE_CH_FN void ge_bl_update_slack(ge_ch *c, int u, int x) {
  if (!c->slack[x] || ge_bl_delta(c, u, x) < ge_bl_delta(c, c->slack[x], x)) c->slack[x] = (uint16_t)u;
}

GE_CH_FN void ge_bl_set_slack(ge_ch *c, int x) {
  c->slack[x] = 0;
  for (int u = 1; u <= c->k; u++)
    if (c->gu[GE_BL_G(c, u, x)] && c->st[u] != x && c->S[c->st[u]] == 0) ge_bl_update_slack(c, u, x);
}

/* the vertices of node x, in cycle order, to the back of the queue */
GE_CH_FN void ge_bl_q_push(ge_ch *c, int x) {
  int top = 0;
  c->stk[top++] = (uint16_t)x;
  while (top > 0) {
    const int y = c->stk[--top];
    if (y <= c->k) {
      if (c->qt >= c->qcap) { c->err = 1; return; }
      c->q[c->qt++] = (uint16_t)y;
      continue;
    }
    const int len = c->fln[y];
    if (top + len > c->qcap) { c->err = 1; return; }
    for (int i = len - 1; i >= 0; i--) c->stk[top++] = GE_BL_FL(c, y, i);
  }
}

GE_CH_FN void ge_bl_set_st(ge_ch *c, int x, int b) {
  int top = 0;
  c->stk[top++] = (uint16_t)x;
  while (top > 0) {
    const int y = c->stk[--top];
    c->st[y] = (uint16_t)b;
    if (y <= c->k) continue;
    const int len = c->fln[y];
    if (top + len > c->qcap) { c->err = 1; return; }
    for (int i = 0; i < len; i++) c->stk[top++] = GE_BL_FL(c, y, i);
  }
}

/* position of sub-node xr in blossom b's cycle, walking in the direction that makes it even (the cycle is reversed if needed) */
GE_CH_FN int ge_bl_get_pr(ge_ch *c, int b, int xr) {
  const int len = c->fln[b];
  int pr = 0;
  while (pr < len && GE_BL_FL(c, b, pr) != xr) pr++;
  if (pr % 2 == 1) {
    for (int i = 1, j = len - 1; i < j; i++, j--) { const uint16_t t = GE_BL_FL(c, b, i); GE_BL_FL(c, b, i) = GE_BL_FL(c, b, j); GE_BL_FL(c, b, j) = t; }
    return len - pr;
  }
  return pr;
}

/* node u becomes matched along the recorded edge (u, v); inside a blossom the matching is rotated so that the sub-node holding
   the edge's endpoint is the new base */
GE_CH_FN void ge_bl_set_match(ge_ch *c, int u0, int v0) {
  int top = 0;
  c->stk[top++] = (uint16_t)u0; c->stk[top++] = (uint16_t)v0;
  while (top > 0) {
    const int v = c->stk[--top], u = c->stk[--top];
    c->match[u] = c->gv[GE_BL_G(c, u, v)];
    if (u <= c->k) continue;
    const int eu = c->gu[GE_BL_G(c, u, v)];
    const int xr = GE_BL_FF(c, u, eu), pr = ge_bl_get_pr(c, u, xr), len = c->fln[u];
    if (top + 2 * (pr + 1) > c->qcap) { c->err = 1; return; }
    for (int i = 0; i < pr; i++) { c->stk[top++] = GE_BL_FL(c, u, i); c->stk[top++] = GE_BL_FL(c, u, i ^ 1); }
    c->stk[top++] = (uint16_t)xr; c->stk[top++] = (uint16_t)v;
    /* rotate the cycle left by pr: xr first */
    for (int i = 0; i < len; i++) c->tmp[i] = GE_BL_FL(c, u, (i + pr) % len);
    for (int i = 0; i < len; i++) GE_BL_FL(c, u, i) = c->tmp[i];
  }
}

GE_CH_FN void ge_bl_augment(ge_ch *c, int u, int v) {
  for (int guard = 0;; guard++) {
    if (guard > c->NX || c->err) { c->err = 1; return; }  /* an alternating tree is at most NX deep */
    const int xnv = c->st[c->match[u]];
    ge_bl_set_match(c, u, v);
    if (!xnv) return;
    ge_bl_set_match(c, xnv, c->st[c->pa[xnv]]);
    u = c->st[c->pa[xnv]]; v = xnv;
  }
}

GE_CH_FN int ge_bl_get_lca(ge_ch *c, int u, int v) {
  int guard = 0;
  for (++c->vt; u || v;) {
    if (++guard > 4 * c->NX) { c->err = 1; return 0; }
    if (u) {
      if (c->vis[u] == c->vt) return u;
      c->vis[u] = c->vt;
      u = c->st[c->match[u]];
      if (u) u = c->st[c->pa[u]];
    }
    { const int t = u; u = v; v = t; }
  }
  return 0;
}

GE_CH_FN void ge_bl_add_blossom(ge_ch *c, int u, int lca, int v) {
  int b = c->k + 1;
  while (b <= c->nx && c->st[b]) b++;
  if (b > c->nx) c->nx++;
  if (c->nx >= c->NX) { c->err = 1; return; }
  c->lab[b] = 0; c->S[b] = 0;
  c->match[b] = c->match[lca];
  int len = 0;
  GE_BL_FL(c, b, len++) = (uint16_t)lca;
  for (int x = u, y; x != lca; x = c->st[c->pa[y]]) {
    if (len + 2 > c->FS) { c->err = 1; return; }
    GE_BL_FL(c, b, len++) = (uint16_t)x; y = c->st[c->match[x]]; GE_BL_FL(c, b, len++) = (uint16_t)y; ge_bl_q_push(c, y);
  }
  for (int i = 1, j = len - 1; i < j; i++, j--) { const uint16_t t = GE_BL_FL(c, b, i); GE_BL_FL(c, b, i) = GE_BL_FL(c, b, j); GE_BL_FL(c, b, j) = t; }
  for (int x = v, y; x != lca; x = c->st[c->pa[y]]) {
    if (len + 2 > c->FS) { c->err = 1; return; }
    GE_BL_FL(c, b, len++) = (uint16_t)x; y = c->st[c->match[x]]; GE_BL_FL(c, b, len++) = (uint16_t)y; ge_bl_q_push(c, y);
  }
  c->fln[b] = (uint16_t)len;
  ge_bl_set_st(c, b, b);
  for (int x = 1; x <= c->nx; x++) { c->gu[GE_BL_G(c, b, x)] = 0; c->gu[GE_BL_G(c, x, b)] = 0; }
  for (int x = 1; x <= c->k; x++) GE_BL_FF(c, b, x) = 0;
  for (int i = 0; i < len; i++) {
    const int xs = GE_BL_FL(c, b, i);
    for (int x = 1; x <= c->nx; x++)
      if (c->gu[GE_BL_G(c, xs, x)] && (!c->gu[GE_BL_G(c, b, x)] || ge_bl_delta(c, xs, x) < ge_bl_delta(c, b, x))) {
        c->gu[GE_BL_G(c, b, x)] = c->gu[GE_BL_G(c, xs, x)]; c->gv[GE_BL_G(c, b, x)] = c->gv[GE_BL_G(c, xs, x)];
        c->gu[GE_BL_G(c, x, b)] = c->gu[GE_BL_G(c, x, xs)]; c->gv[GE_BL_G(c, x, b)] = c->gv[GE_BL_G(c, x, xs)];
      }
    for (int x = 1; x <= c->k; x++) if (GE_BL_FF(c, xs, x)) GE_BL_FF(c, b, x) = (uint16_t)xs;
  }
  ge_bl_set_slack(c, b);
}

GE_CH_FN void ge_bl_expand_blossom(ge_ch *c, int b) {
  const int len = c->fln[b];
  for (int i = 0; i < len; i++) ge_bl_set_st(c, GE_BL_FL(c, b, i), GE_BL_FL(c, b, i));
  const int xr = GE_BL_FF(c, b, c->gu[GE_BL_G(c, b, c->pa[b])]), pr = ge_bl_get_pr(c, b, xr);
  for (int i = 0; i < pr; i += 2) {
    const int xs = GE_BL_FL(c, b, i), xns = GE_BL_FL(c, b, i + 1);
    c->pa[xs] = c->gu[GE_BL_G(c, xns, xs)];
    c->S[xs] = 1; c->S[xns] = 0;
    c->slack[xs] = 0; ge_bl_set_slack(c, xns);
    ge_bl_q_push(c, xns);
  }
  c->S[xr] = 1; c->pa[xr] = c->pa[b];
  for (int i = pr + 1; i < len; i++) { const int xs = GE_BL_FL(c, b, i); c->S[xs] = -1; ge_bl_set_slack(c, xs); }
  c->st[b] = 0;
}

/* the tight edge recorded between nodes a and b was found; returns 1 when it closed an augmenting path */
GE_CH_FN int ge_bl_on_found_edge(ge_ch *c, int a, int b) {
  const int eu = c->gu[GE_BL_G(c, a, b)], ev = c->gv[GE_BL_G(c, a, b)];
  const int u = c->st[eu], v = c->st[ev];
  if (c->S[v] == -1) {
    c->pa[v] = (uint16_t)eu; c->S[v] = 1;
    const int nu = c->st[c->match[v]];
    c->slack[v] = 0; c->slack[nu] = 0;
    c->S[nu] = 0; ge_bl_q_push(c, nu);
  } else if (c->S[v] == 0) {
    const int lca = ge_bl_get_lca(c, u, v);
    if (!lca) { ge_bl_augment(c, u, v); ge_bl_augment(c, v, u); return 1; }
    ge_bl_add_blossom(c, u, lca, v);
  }
  return 0;
}

/* one phase: grow alternating trees from every free node until an augmenting path is found (1) or no dual step is left (0) */
GE_CH_FN int ge_bl_phase(ge_ch *c) {
  for (int x = 1; x <= c->nx; x++) { c->S[x] = -1; c->slack[x] = 0; }
  c->qh = c->qt = 0;
  for (int x = 1; x <= c->nx; x++) if (c->st[x] == x && !c->match[x]) { c->pa[x] = 0; c->S[x] = 0; ge_bl_q_push(c, x); }
  if (c->qh == c->qt) return 0;
  for (int guard = 0;; guard++) {
    if (guard > 8 * c->NX + 64) c->err = 1;  /* a phase makes O(k) dual steps */
    if (c->err) return 0;
    while (c->qh < c->qt) {
      const int u = c->q[c->qh++];
      if (c->S[c->st[u]] == 1) continue;
      for (int v = 1; v <= c->k; v++)
        if (c->gu[GE_BL_G(c, u, v)] && c->st[u] != c->st[v]) {
          if (ge_bl_delta(c, u, v) == 0) { if (ge_bl_on_found_edge(c, u, v)) return 1; if (c->err) return 0; }
          else ge_bl_update_slack(c, u, c->st[v]);
        }
    }
    int64_t d = INT64_MAX;
    for (int b = c->k + 1; b <= c->nx; b++) if (c->st[b] == b && c->S[b] == 1 && c->lab[b] / 2 < d) d = c->lab[b] / 2;
    for (int x = 1; x <= c->nx; x++)
      if (c->st[x] == x && c->slack[x]) {
        const int64_t e = ge_bl_delta(c, c->slack[x], x);
        if (c->S[x] == -1) { if (e < d) d = e; }
        else if (c->S[x] == 0) { if (e / 2 < d) d = e / 2; }
      }
    for (int u = 1; u <= c->k; u++) {
      if (c->S[c->st[u]] == 0) { if (c->lab[u] <= d) return 0; c->lab[u] -= d; }
      else if (c->S[c->st[u]] == 1) c->lab[u] += d;
    }
    for (int b = c->k + 1; b <= c->nx; b++)
      if (c->st[b] == b) {
        if (c->S[b] == 0) c->lab[b] += d * 2;
        else if (c->S[b] == 1) c->lab[b] -= d * 2;
      }
    c->qh = c->qt = 0;
    for (int x = 1; x <= c->nx; x++)
      if (c->st[x] == x && c->slack[x] && c->st[c->slack[x]] != x && ge_bl_delta(c, c->slack[x], x) == 0) {
        if (ge_bl_on_found_edge(c, c->slack[x], x)) return 1;
        if (c->err) return 0;
      }
    for (int b = c->k + 1; b <= c->nx; b++) if (c->st[b] == b && c->S[b] == 1 && c->lab[b] == 0) ge_bl_expand_blossom(c, b);
  }
}

/* maximum-weight matching of the complete graph on vertices 1..k with weights c->W (positive off the diagonal); match[] on return */
GE_CH_FN void ge_bl_solve(ge_ch *c) {
  const int k = c->k;
  c->NX = 2 * k + 1; c->FS = k + 2; c->nx = k; c->vt = 0; c->err = 0;
  int32_t wmax = 0;
  for (int u = 0; u <= 2 * k; u++) { c->st[u] = (uint16_t)(u <= k ? u : 0); c->fln[u] = 0; c->match[u] = 0; c->vis[u] = 0; c->lab[u] = 0; c->slack[u] = 0; c->pa[u] = 0; c->S[u] = -1; }
  for (int u = 1; u <= k; u++)
    for (int v = 1; v <= k; v++) {
      c->gu[GE_BL_G(c, u, v)] = (uint16_t)(u == v ? 0 : u); c->gv[GE_BL_G(c, u, v)] = (uint16_t)(u == v ? 0 : v);
      GE_BL_FF(c, u, v) = (uint16_t)(u == v ? u : 0);
      if (u != v && c->W[u * (k + 1) + v] > wmax) wmax = c->W[u * (k + 1) + v];
    }
  for (int u = 1; u <= k; u++) c->lab[u] = wmax;
  for (int phases = 0; phases <= k && ge_bl_phase(c); phases++) {}
}

/* ------------------------------------------------------------------ the tour */
/* c->D holds the closure; returns the length of the tour in the units of D (or -1: scratch capacity exceeded) */
GE_CH_FN int64_t ge_christofides_tour(ge_ch *c) {
  const int n = c->n;
  const int32_t *D = c->D;
  if (n < 2) return 0;
  if (n == 2) return 2 * (int64_t)D[1];
  /* minimum spanning tree of the closure (Prim from node 0; among equal keys the lowest index) */
  for (int v = 0; v < n; v++) { c->key[v] = INT32_MAX; c->par[v] = -1; c->intree[v] = 0; c->deg[v] = 0; }
  c->key[0] = 0;
  for (int it = 0; it < n; it++) {
    int v = -1; int32_t best = INT32_MAX;
    for (int u = 0; u < n; u++) if (!c->intree[u] && c->key[u] < best) { best = c->key[u]; v = u; }
    if (v < 0) return -1;  /* unreachable node: the graphs are connected */
    c->intree[v] = 1;
    if (c->par[v] >= 0) { c->deg[v]++; c->deg[c->par[v]]++; }
    for (int u = 0; u < n; u++) if (!c->intree[u] && D[v * n + u] < c->key[u]) { c->key[u] = D[v * n + u]; c->par[u] = (int16_t)v; }
  }
  /* minimum-weight perfect matching of the odd-degree nodes */
  int k = 0;
  for (int v = 0; v < n; v++) if (c->deg[v] & 1) c->odd[k++] = (uint16_t)v;
  c->k = k;
  if (k > 0) {
    int32_t dmax = 0;
    for (int a = 0; a < k; a++) for (int b = 0; b < k; b++) if (a != b && D[c->odd[a] * n + c->odd[b]] > dmax) dmax = D[c->odd[a] * n + c->odd[b]];
    for (int a = 0; a <= k; a++) for (int b = 0; b <= k; b++) c->W[a * (k + 1) + b] = (a && b && a != b) ? dmax + 1 - D[c->odd[a - 1] * n + c->odd[b - 1]] : 0;
    ge_bl_solve(c);
    if (c->err) return -1;
    for (int a = 1; a <= k; a++) if (!c->match[a] || c->match[c->match[a]] != a) return -1;
  }
  /* multigraph tree + matching: every degree is even */
  const int m2 = n - 1 + k / 2;
  for (int v = 0; v <= n; v++) c->ehead[v] = 0;
  for (int v = 1; v < n; v++) { c->ehead[v + 1]++; c->ehead[c->par[v] + 1]++; }
  for (int a = 1; a <= k; a++) if (c->match[a] > a) { c->ehead[c->odd[a - 1] + 1]++; c->ehead[c->odd[c->match[a] - 1] + 1]++; }
  for (int v = 0; v < n; v++) c->ehead[v + 1] += c->ehead[v];
  for (int v = 0; v < n; v++) c->eptr[v] = c->ehead[v];
  int e = 0;
  for (int v = 1; v < n; v++, e++) {
    const int p = c->par[v];
    c->eto[c->eptr[v]] = (uint16_t)p; c->eid[c->eptr[v]++] = e;
    c->eto[c->eptr[p]] = (uint16_t)v; c->eid[c->eptr[p]++] = e;
  }
  for (int a = 1; a <= k; a++) if (c->match[a] > a) {
    const int u = c->odd[a - 1], v = c->odd[c->match[a] - 1];
    c->eto[c->eptr[u]] = (uint16_t)v; c->eid[c->eptr[u]++] = e;
    c->eto[c->eptr[v]] = (uint16_t)u; c->eid[c->eptr[v]++] = e;
    e++;
  }
  for (int i = 0; i < m2; i++) c->eused[i] = 0;
  for (int v = 0; v < n; v++) c->eptr[v] = c->ehead[v];
  /* Eulerian circuit from node 0 (Hierholzer), then shortcut: keep the first visit of every node */
  int top = 0, nc = 0;
  c->cstk[top++] = 0;
  while (top > 0) {
    const int v = c->cstk[top - 1];
    while (c->eptr[v] < c->ehead[v + 1] && c->eused[c->eid[c->eptr[v]]]) c->eptr[v]++;
    if (c->eptr[v] == c->ehead[v + 1]) { c->circ[nc++] = (uint16_t)v; top--; }
    else { c->eused[c->eid[c->eptr[v]]] = 1; c->cstk[top++] = c->eto[c->eptr[v]++]; }
  }
  if (nc != m2 + 1) return -1;
  for (int v = 0; v < n; v++) c->intree[v] = 0;
  int64_t total = 0; int prev = -1, first = -1, seen = 0;
  for (int i = nc - 1; i >= 0; i--) {
    const int v = c->circ[i];
    if (c->intree[v]) continue;
    c->intree[v] = 1; seen++;
    if (prev >= 0) total += D[prev * n + v]; else first = v;
    prev = v;
  }
  if (seen != n) return -1;
  return total + D[prev * n + first];
}

#endif
