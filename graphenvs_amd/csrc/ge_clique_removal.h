/* MaxIndependentSet baseline (is_eval_env, unweighted): len(nx.approximation.maximum_independent_set(G))
 * (max_independent_set.py:63-67) = the largest independent set Boppana-Halldorsson clique removal meets, EXACTLY as networkx 3.4.2
 * on CPython 3.10 computes it.  The value depends on which node every ramsey_R2 call picks first, and that is decided by
 *   - the insertion orders of networkx's node and adjacency dicts through G.copy(), G.subgraph(...).copy(), remove_nodes_from;
 *   - the iteration order of CPython sets of small ints: FilterAtlas iterates the node SET of a subgraph view when it holds less
 *     than half of the parent's nodes ([nx] coreviews.FilterAtlas.__iter__), and non_neighbors() is
 *     `G._adj.keys() - G._adj[node].keys() - {node}` ([nx] function.non_neighbors), a set built by PySet_New(dict),
 *     difference_update and set.__sub__ ([py] Objects/setobject.c, Objects/dictobject.c dictviews_sub).
 * Both are restated here: ordered node / adjacency lists, and the open-addressing table of setobject.c (linear probes, perturbation,
 * dummies, the three resize rules, set_merge's copy paths) for int keys (hash(i) == i).
 *
 * Plain C, one sequential thread per slot, no recursion (explicit frames), no allocation (a caller-provided arena): the device
 * runs it with one lane per slot (ge_mis_eval.h) and the CPU checker compiles the same text; tests/ compare the set emulation
 * with the interpreter's own sets and the result with networkx on random graphs and with the reference's fixtures. */
#ifndef GE_CLIQUE_REMOVAL_H
#define GE_CLIQUE_REMOVAL_H
#include <stdint.h>

#ifndef GE_CR_FN
#define GE_CR_FN static inline
#endif
#ifndef GE_CR_HD
#define GE_CR_HD GE_CR_FN
#endif

#define GE_CR_EMPTY (-1)
#define GE_CR_DUMMY (-2)
#define GE_CR_MAXW 64 /* clique bit sets: n <= 4096 */

/* ------------------------------------------------------------------ arena (stack discipline) */
typedef struct { uint8_t *base; uint64_t top, cap, peak; int err; } ge_cr_arena;

GE_CR_FN void *ge_cr_alloc(ge_cr_arena *a, uint64_t bytes) {
  const uint64_t at = (a->top + 7u) & ~(uint64_t)7u;
  if (at + bytes > a->cap) { a->err = 1; return (void *)a->base; }  /* callers check err before they trust a result */
  a->top = at + bytes;
  if (a->top > a->peak) a->peak = a->top;
  return (void *)(a->base + at);
}

/* ------------------------------------------------------------------ [py] setobject.c for int keys */
typedef struct { int32_t *tab; int32_t mask, fill, used; } ge_pyset;

GE_CR_FN void ge_pyset_init(ge_cr_arena *a, ge_pyset *s) {  /* make_new_set: the 8-entry small table */
  s->tab = (int32_t *)ge_cr_alloc(a, 8 * 4);
  s->mask = 7; s->fill = 0; s->used = 0;
  if (!a->err) for (int i = 0; i < 8; i++) s->tab[i] = GE_CR_EMPTY;
}

GE_CR_FN void ge_pyset_insert_clean(int32_t *tab, int32_t mask, int32_t key) {  /* set_insert_clean */
  uint32_t perturb = (uint32_t)key, i = (uint32_t)key & (uint32_t)mask;
  for (;;) {
    if (tab[i] == GE_CR_EMPTY) { tab[i] = key; return; }
    if (i + 9 <= (uint32_t)mask) for (int j = 1; j <= 9; j++) if (tab[i + j] == GE_CR_EMPTY) { tab[i + j] = key; return; }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & (uint32_t)mask;
  }
}

GE_CR_FN void ge_pyset_resize(ge_cr_arena *a, ge_pyset *s, int32_t minused) {  /* set_table_resize */
  int32_t newsize = 8;
  while (newsize <= minused) newsize <<= 1;
  int32_t *nt = (int32_t *)ge_cr_alloc(a, (uint64_t)newsize * 4);
  if (a->err) return;
  for (int32_t i = 0; i < newsize; i++) nt[i] = GE_CR_EMPTY;
  for (int32_t i = 0; i <= s->mask; i++) if (s->tab[i] >= 0) ge_pyset_insert_clean(nt, newsize - 1, s->tab[i]);
  s->tab = nt; s->mask = newsize - 1; s->fill = s->used;
}

/* the entry that holds key, or -1 (set_lookkey) */
GE_CR_FN int32_t ge_pyset_find(const ge_pyset *s, int32_t key) {
  uint32_t perturb = (uint32_t)key, i = (uint32_t)key & (uint32_t)s->mask;
  for (;;) {
    int probes = (i + 9 <= (uint32_t)s->mask) ? 9 : 0;
    uint32_t e = i;
    do {
      if (s->tab[e] == GE_CR_EMPTY) return -1;
      if (s->tab[e] == key) return (int32_t)e;
      e++;
    } while (probes--);
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & (uint32_t)s->mask;
  }
}

GE_CR_FN void ge_pyset_add(ge_cr_arena *a, ge_pyset *s, int32_t key) {  /* set_add_entry */
  uint32_t perturb = (uint32_t)key, i = (uint32_t)key & (uint32_t)s->mask;
  int32_t freeslot = -1;
  for (;;) {
    int probes = (i + 9 <= (uint32_t)s->mask) ? 9 : 0;
    uint32_t e = i;
    do {
      if (s->tab[e] == GE_CR_EMPTY) {
        if (freeslot >= 0) { s->tab[freeslot] = key; s->used++; return; }
        s->tab[e] = key; s->fill++; s->used++;
        if ((int64_t)s->fill * 5 >= (int64_t)s->mask * 3) ge_pyset_resize(a, s, s->used > 50000 ? s->used * 2 : s->used * 4);
        return;
      }
      if (s->tab[e] == key) return;
      if (s->tab[e] == GE_CR_DUMMY) freeslot = (int32_t)e;
      e++;
    } while (probes--);
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & (uint32_t)s->mask;
  }
}

GE_CR_FN void ge_pyset_discard(ge_pyset *s, int32_t key) {  /* set_discard_entry */
  const int32_t e = ge_pyset_find(s, key);
  if (e >= 0) { s->tab[e] = GE_CR_DUMMY; s->used--; }
}

/* the tail of set_difference_update_internal: "If more than 1/4th are dummies, then resize them away." */
GE_CR_FN void ge_pyset_drop_dummies(ge_cr_arena *a, ge_pyset *s) {
  if ((uint32_t)(s->fill - s->used) <= (uint32_t)s->mask / 4) return;
  ge_pyset_resize(a, s, s->used > 50000 ? s->used * 2 : s->used * 4);
}

/* PySet_New(dict) (set_update_internal, PyDict_CheckExact branch): one resize up front, then the keys in dict order */
GE_CR_FN void ge_pyset_from_dict(ge_cr_arena *a, ge_pyset *s, const uint16_t *keys, int32_t count) {
  ge_pyset_init(a, s);
  if ((int64_t)(s->fill + count) * 5 >= (int64_t)s->mask * 3) ge_pyset_resize(a, s, (s->used + count) * 2);
  for (int32_t i = 0; i < count && !a->err; i++) ge_pyset_add(a, s, keys[i]);
}

/* set(so) / so.copy(): set_merge into a new, empty set */
GE_CR_FN void ge_pyset_copy(ge_cr_arena *a, const ge_pyset *so, ge_pyset *res) {
  ge_pyset_init(a, res);
  if (a->err || !so->used) return;
  if ((int64_t)(res->fill + so->used) * 5 >= (int64_t)res->mask * 3) ge_pyset_resize(a, res, (res->used + so->used) * 2);
  if (a->err) return;
  if (res->mask == so->mask && so->fill == so->used) {  /* same size, no dummies: the table is copied as it is */
    for (int32_t i = 0; i <= so->mask; i++) res->tab[i] = so->tab[i];
    res->fill = so->fill; res->used = so->used;
  } else {  /* empty target: set_insert_clean in table order */
    res->fill = so->used; res->used = so->used;
    for (int32_t i = 0; i <= so->mask; i++) if (so->tab[i] >= 0) ge_pyset_insert_clean(res->tab, res->mask, so->tab[i]);
  }
}

/* so - {key} (set_sub -> set_difference with a one-element set) */
GE_CR_FN void ge_pyset_minus_one(ge_cr_arena *a, const ge_pyset *so, int32_t key, ge_pyset *res) {
  if ((so->used >> 2) > 1) {  /* set_copy_and_difference: set_copy, then discard */
    ge_pyset_copy(a, so, res);
    if (a->err) return;
    ge_pyset_discard(res, key);
    ge_pyset_drop_dummies(a, res);
    return;
  }
  ge_pyset_init(a, res);
  for (int32_t i = 0; i <= so->mask && !a->err; i++) if (so->tab[i] >= 0 && so->tab[i] != key) ge_pyset_add(a, res, so->tab[i]);
}

/* ------------------------------------------------------------------ graphs with networkx's dict orders */
typedef struct { int32_t k; uint16_t *nodes; int32_t *off; uint16_t *adj; } ge_cr_graph;  /* off[i] .. off[i+1]: neighbours of nodes[i], in dict order */

typedef struct {
  int n, W;
  int16_t *pos;        /* [n] scratch: position of a node in the copy being built, -1 = not a member */
  ge_cr_arena ar;
} ge_cr;

/* G.subgraph(members).copy(): `members` arrive in iteration order (a list or the table order of a set); set_order != NULL means
 * "use this order when the member set is less than half of the parent" ([nx] FilterAtlas.__iter__), else the parent's order. */
GE_CR_FN void ge_cr_subcopy(ge_cr *c, const ge_cr_graph *g, const uint16_t *members, int32_t cnt, ge_cr_graph *out) {
  ge_cr_arena *a = &c->ar;
  out->k = cnt;
  out->nodes = (uint16_t *)ge_cr_alloc(a, (uint64_t)(cnt + 1) * 2);
  out->off = (int32_t *)ge_cr_alloc(a, (uint64_t)(cnt + 1) * 4);
  if (a->err) return;
  if (2 * cnt < g->k) {  /* the node SET's iteration order: set(members) built by insertion */
    const uint64_t mark = a->top;
    ge_pyset s; ge_pyset_init(a, &s);
    for (int32_t i = 0; i < cnt && !a->err; i++) ge_pyset_add(a, &s, members[i]);
    if (a->err) return;
    int32_t w = 0;
    for (int32_t i = 0; i <= s.mask; i++) if (s.tab[i] >= 0) out->nodes[w++] = (uint16_t)s.tab[i];
    a->top = mark;
  } else {
    for (int32_t i = 0; i < cnt; i++) c->pos[members[i]] = 0;
    int32_t w = 0;
    for (int32_t i = 0; i < g->k; i++) if (c->pos[g->nodes[i]] == 0) out->nodes[w++] = g->nodes[i];
    for (int32_t i = 0; i < cnt; i++) c->pos[members[i]] = -1;
  }
  for (int32_t i = 0; i < cnt; i++) c->pos[out->nodes[i]] = (int16_t)i;
  /* degrees inside the member set (parent rows are found through the parent's own position table: rebuilt per call) */
  int32_t *ppos = (int32_t *)ge_cr_alloc(a, (uint64_t)cnt * 4);  /* parent row of every member */
  if (a->err) { for (int32_t i = 0; i < cnt; i++) c->pos[out->nodes[i]] = -1; return; }
  for (int32_t i = 0; i < g->k; i++) { const int p = c->pos[g->nodes[i]]; if (p >= 0) ppos[p] = i; }
  int32_t tot = 0;
  for (int32_t i = 0; i < cnt; i++) {
    out->off[i] = tot;
    const int32_t r = ppos[i];
    for (int32_t e = g->off[r]; e < g->off[r + 1]; e++) if (c->pos[g->adj[e]] >= 0) tot++;
  }
  out->off[cnt] = tot;
  out->adj = (uint16_t *)ge_cr_alloc(a, (uint64_t)(tot + 1) * 2);
  int32_t *fill = (int32_t *)ge_cr_alloc(a, (uint64_t)(cnt + 1) * 4);
  if (!a->err) {
    /* add_edges_from over the copy's node order: a row first receives its earlier neighbours (in their order), then, at its own
       turn, the later ones in the parent's adjacency order */
    for (int32_t i = 0; i < cnt; i++) fill[i] = out->off[i];
    for (int32_t i = 0; i < cnt; i++) {
      const int32_t r = ppos[i];
      for (int32_t e = g->off[r]; e < g->off[r + 1]; e++) {
        const int p = c->pos[g->adj[e]];
        if (p > i) { out->adj[fill[i]++] = g->adj[e]; out->adj[fill[p]++] = out->nodes[i]; }
      }
    }
  }
  for (int32_t i = 0; i < cnt; i++) c->pos[out->nodes[i]] = -1;
  /* ppos / fill stay allocated until the caller releases the frame: they sit above the graph and are small */
}

/* iteration order of nx.non_neighbors(g, g.nodes[0]) = keys(g) - keys(adj[node]) - {node}; returns the count */
GE_CR_FN int32_t ge_cr_non_neighbors(ge_cr *c, const ge_cr_graph *g, uint16_t *out) {
  ge_cr_arena *a = &c->ar;
  const uint64_t mark = a->top;
  ge_pyset s, r;
  ge_pyset_from_dict(a, &s, g->nodes, g->k);
  if (a->err) return 0;
  for (int32_t e = g->off[0]; e < g->off[1]; e++) ge_pyset_discard(&s, g->adj[e]);  /* difference_update over the dict view, in its order */
  ge_pyset_drop_dummies(a, &s);
  ge_pyset_minus_one(a, &s, g->nodes[0], &r);
  int32_t w = 0;
  if (!a->err) for (int32_t i = 0; i <= r.mask; i++) if (r.tab[i] >= 0) out[w++] = (uint16_t)r.tab[i];
  a->top = mark;
  return w;
}

/* ------------------------------------------------------------------ [nx] ramsey_R2, iteratively */
typedef struct {
  ge_cr_graph g; uint64_t mark;  /* the arena is released to `mark` when this call returns */
  int state; int32_t c1size, i1;
  uint64_t c1[GE_CR_MAXW];
} ge_cr_frame;

/* (largest clique found, size of the largest independent set found) of g; frames: caller-provided, depth n + 2 */
GE_CR_FN void ge_cr_ramsey(ge_cr *c, const ge_cr_graph *g0, ge_cr_frame *frames, uint64_t *clique, int32_t *csize, int32_t *isize) {
  ge_cr_arena *a = &c->ar;
  const int W = c->W;
  int depth = 0;
  frames[0].g = *g0; frames[0].mark = a->top; frames[0].state = 0;
  uint64_t ret_c[GE_CR_MAXW]; int32_t ret_cs = 0, ret_is = 0;
  for (int w = 0; w < W; w++) ret_c[w] = 0;
  while (depth >= 0) {
    ge_cr_frame *f = &frames[depth];
    if (a->err) break;
    if (f->state == 0) {
      if (f->g.k == 0) { for (int w = 0; w < W; w++) ret_c[w] = 0; ret_cs = 0; ret_is = 0; a->top = f->mark; depth--; continue; }
      /* neighbours of the first node, in adjacency order (no self loops in these graphs) */
      ge_cr_frame *ch = &frames[depth + 1];
      ch->mark = a->top; ch->state = 0;
      ge_cr_subcopy(c, &f->g, f->g.adj + f->g.off[0], f->g.off[1] - f->g.off[0], &ch->g);
      f->state = 1; depth++;
    } else if (f->state == 1) {
      for (int w = 0; w < W; w++) f->c1[w] = ret_c[w];
      f->c1size = ret_cs; f->i1 = ret_is;
      ge_cr_frame *ch = &frames[depth + 1];
      ch->mark = a->top; ch->state = 0;
      uint16_t *nn = (uint16_t *)ge_cr_alloc(a, (uint64_t)(f->g.k + 1) * 2);
      if (a->err) break;
      const int32_t cnt = ge_cr_non_neighbors(c, &f->g, nn);
      ge_cr_subcopy(c, &f->g, nn, cnt, &ch->g);
      f->state = 2; depth++;
    } else {
      const int node = f->g.nodes[0];
      f->c1[node >> 6] |= 1ull << (node & 63); f->c1size++;
      const int32_t i2 = ret_is + 1;
      if (f->c1size >= ret_cs) { for (int w = 0; w < W; w++) ret_c[w] = f->c1[w]; ret_cs = f->c1size; }  /* max(c_1, c_2, key=len): the first of equals */
      ret_is = f->i1 >= i2 ? f->i1 : i2;
      a->top = f->mark; depth--;
    }
  }
  for (int w = 0; w < W; w++) clique[w] = ret_c[w];
  *csize = ret_cs; *isize = ret_is;
}

/* ------------------------------------------------------------------ [nx] clique_removal */
/* bytes of work space for a graph of n nodes and m edges: two graph buffers, the frames, and a stack arena several times the peak
 * seen on random graphs (the worst case is ~20x larger); when it does not suffice the call reports -1 and the caller keeps its
 * greedy value. */
GE_CR_HD uint64_t ge_cr_frames_bytes(int n) { return (uint64_t)(n + 3) * sizeof(ge_cr_frame); }
GE_CR_HD uint64_t ge_cr_graph_bytes(int n, int m) { return (((uint64_t)(n + 1) * 2 + 7) & ~7ull) + (((uint64_t)(n + 1) * 4 + 7) & ~7ull) + (((uint64_t)(2 * m + 1) * 2 + 7) & ~7ull); }
GE_CR_HD uint64_t ge_cr_arena_bytes(int n, int m) {
  /* measured on random G(n, m), sparse to dense, n = 64 .. 512: the stack of graph copies and set tables peaks near 4.4 n^2 bytes */
  return 24ull * (uint64_t)n * (uint64_t)n + 16ull * (uint64_t)(n + 2 * m) + 65536ull;
}
GE_CR_HD uint64_t ge_cr_slot_bytes(int n, int m) {
  const uint64_t bytes = 2 * ge_cr_graph_bytes(n, m) + ge_cr_frames_bytes(n) + (((uint64_t)n * 2 + 7) & ~7ull) + ge_cr_arena_bytes(n, m);
  return (bytes + 15u) & ~(uint64_t)15u;  /* slot blocks follow one another: keep them aligned */
}

/* Work space of one slot, carved: the caller fills g0 (nodes 0 .. n-1 in order, off[] / adj[] = the graph's insertion-order
 * adjacency), then runs ge_cr_solve. */
typedef struct { ge_cr c; ge_cr_graph ga, gb; ge_cr_frame *frames; } ge_cr_work;

GE_CR_FN void ge_cr_carve(ge_cr_work *w, uint8_t *work, int n, int m) {
  uint64_t at = 0;
  w->c.n = n; w->c.W = (n + 63) / 64;
  w->ga.nodes = (uint16_t *)(work + at); at += ((uint64_t)(n + 1) * 2 + 7) & ~7ull;
  w->ga.off = (int32_t *)(work + at); at += ((uint64_t)(n + 1) * 4 + 7) & ~7ull;
  w->ga.adj = (uint16_t *)(work + at); at += ((uint64_t)(2 * m + 1) * 2 + 7) & ~7ull;
  w->gb.nodes = (uint16_t *)(work + at); at += ((uint64_t)(n + 1) * 2 + 7) & ~7ull;
  w->gb.off = (int32_t *)(work + at); at += ((uint64_t)(n + 1) * 4 + 7) & ~7ull;
  w->gb.adj = (uint16_t *)(work + at); at += ((uint64_t)(2 * m + 1) * 2 + 7) & ~7ull;
  w->frames = (ge_cr_frame *)(work + at); at += ge_cr_frames_bytes(n);
  w->c.pos = (int16_t *)(work + at); at += ((uint64_t)n * 2 + 7) & ~7ull;
  w->c.ar.base = work + at; w->c.ar.top = 0; w->c.ar.peak = 0; w->c.ar.cap = ge_cr_arena_bytes(n, m); w->c.ar.err = 0;
  w->ga.k = n;
  for (int v = 0; v < n; v++) { w->c.pos[v] = -1; w->ga.nodes[v] = (uint16_t)v; }
}

/* len(nx.approximation.maximum_independent_set(G)), or -1 when the work space was too small */
GE_CR_FN int32_t ge_cr_solve(ge_cr_work *w) {
  ge_cr *c = &w->c;
  ge_cr_graph *cur = &w->gb, *nxt = &w->ga;
  uint64_t clique[GE_CR_MAXW]; int32_t cs, is, best;
  { /* graph = G.copy(): add_edges_from over G's adjacency re-inserts every row as "earlier nodes first (in node order), then the
       later ones in G's adjacency order" -- the same rule as a subgraph copy; here the frames' memory serves as the fill counters */
    const ge_cr_graph *g = &w->ga;
    const int32_t n = g->k;
    int32_t *fill = (int32_t *)w->frames;
    cur->k = n;
    for (int32_t i = 0; i <= n; i++) cur->off[i] = g->off[i];
    for (int32_t i = 0; i < n; i++) { cur->nodes[i] = g->nodes[i]; fill[i] = g->off[i]; }
    for (int32_t i = 0; i < n; i++)
      for (int32_t e = g->off[i]; e < g->off[i + 1]; e++) {
        const int32_t p = g->adj[e];  /* nodes are 0 .. n-1 in order: a node is its own position */
        if (p > i) { cur->adj[fill[i]++] = (uint16_t)p; cur->adj[fill[p]++] = (uint16_t)i; }
      }
  }
  ge_cr_ramsey(c, cur, w->frames, clique, &cs, &is);
  if (c->ar.err) return -1;
  best = is;
  while (cur->k > 0) {
    /* graph.remove_nodes_from(c_i): the dicts keep their order */
    int32_t k2 = 0, tot = 0;
    for (int32_t i = 0; i < cur->k; i++) {
      const int v = cur->nodes[i];
      if ((clique[v >> 6] >> (v & 63)) & 1ull) continue;
      nxt->nodes[k2] = (uint16_t)v; nxt->off[k2] = tot; k2++;
      for (int32_t e = cur->off[i]; e < cur->off[i + 1]; e++) { const int u = cur->adj[e]; if (!((clique[u >> 6] >> (u & 63)) & 1ull)) nxt->adj[tot++] = (uint16_t)u; }
    }
    if (k2 == cur->k) return -1;  /* cannot happen: a non-empty graph yields a non-empty clique */
    nxt->off[k2] = tot; nxt->k = k2;
    { ge_cr_graph *t = cur; cur = nxt; nxt = t; }
    c->ar.top = 0;
    ge_cr_ramsey(c, cur, w->frames, clique, &cs, &is);
    if (c->ar.err) return -1;
    if (is > best) best = is;  /* isets.append(i_i); max(isets, key=len) */
  }
  return best;
}

#endif
