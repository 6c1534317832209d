// C ABI of libgraphenvs_hip.so (include/graphenvs.h): config validation, buffer binding and
// kernel launches.  No device allocation, no synchronisation (except ge_timed_rollout).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "ge_params.h"
#include "ge_platform.h"
#include "ge_reset.h"
#include "ge_features.h"
#include "ge_step.h"
#include "ge_tsp_eval.h"
#include "ge_spare.h"

// Multi-class engine: size classes are grouped into LDS buckets (by n_nodes: <= 128, <= 256, <= 512, larger), and the graph kernel
// and the generic feature kernel are launched once per bucket with the dynamic LDS -- hence the residency -- of the bucket's largest
// class (round 2 ran every class at the occupancy of n = 512: one workgroup per CU)
#define GE_MAX_BUCKETS 4
struct GeBucket {
  bool used;
  int reset_lds, reset_grid;                     // graph kernel
  int gen_lds, gen_pre_off, gen_waves, gen_grid;  // generic feature kernel (classes with n > 64)
  bool gen_used;
};
static int bucket_of(int n) { return n <= 128 ? 0 : (n <= 256 ? 1 : (n <= 512 ? 2 : 3)); }

struct ge_engine {
  GeParams P;
  ge_config cfg;
  int reset_grid;     // workgroups of the queue-mode reset launch
  int lds_bytes;
  int feat_lds, feat_grid, feat_fast, gen_grid, gen_lds;  // structural-feature kernel launch geometry
  hipEvent_t ev[4];
  bool have_events;
  int nseed;      // seeding workgroups at the head of the queue-mode reset launch (64 queued slots each)
  // multi-class ("ragged") engine: P is then the engine-wide block (B = all slots, global queue / seed / episode / mt_state arrays,
  // n / m / W = the widest class) and R names the device copy of the class table
  int n_classes;
  GeRagged R;
  std::vector<GeParams> classes;  // host copy (ge_vectorize launches per class)
  int feat64_pre_off, gen_pre_off;
  int lds_bytes_inject;  // GeParams.nocolw engines: ge_inject_state runs the graph kernel on the full LDS carve (the injected rows need the list)
  GeBucket bk[GE_MAX_BUCKETS];
  bool loaded;    // the slots hold an episode (ge_reset or ge_inject_state ran)
  bool seeded;    // the generator-state ring is valid (ge_reset, or ge_inject_state with seeds)
  bool streams;   // stream_state holds the streams a regeneration left behind (ge_reset; a restored snapshot)
  // episode prefetch (ge_attach_spares): PS is the engine seen through its spare image -- the same geometry, generator ring,
  // seed[] / episode[] and work lists, but every per-slot slab is the image's and the queue is the refill list
  bool spares;
  GeParams PS;
  GeRagged RS;                     // multi-class engine: class table whose bufs are the images
  std::vector<GeParams> classesS;  // its host copy
  int period, swap_parts;
  int64_t pending_calls;           // ge_reset_pending calls since the last refill
  int32_t *refill_list, *refill_count;
};

// ---- what a reset-path launch does (GeRun, ge_params.h): the only places where a request becomes flags
static GeRun run_full() { GeRun r = {GE_ITEMS_ALL, 1, 0, 0, 0, 0, 0}; return r; }                         // ge_reset
static GeRun run_inject(bool seeds) { GeRun r = {GE_ITEMS_ALL, seeds ? 2 : 0, 0, 0, 1, 0, 0}; return r; }  // ge_inject_state
static GeRun run_continue() { GeRun r = {GE_ITEMS_ALL, 0, 1, 1, 0, 0, 0}; return r; }                     // ge_reset_continue
// finished slots regenerated in place.  Without spares the launch that moves a slot to episode e + 1 refills ring entry e with
// episode e + GE_SEED_DEPTH; with spares every regeneration of a slot -- refill or in place -- seeds the episode after the one it
// generates, two past the one seed[] / episode[] name
static GeRun run_queue(const ge_engine *e) { GeRun r = {GE_ITEMS_QUEUE, 0, 1, 0, 0, 0, e->spares ? 2 : GE_SEED_DEPTH}; return r; }
static GeRun run_refill() { GeRun r = {GE_ITEMS_QUEUE, 0, 1, 0, 0, 1, 2}; return r; }
static GeRun as_list(GeRun r) { r.items = GE_ITEMS_LIST; return r; }  // the feature kernels' fallback list of the same launch sequence


static thread_local char g_err[256] = "";
static int fail(int code, const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }
extern "C" const char *ge_last_error(void) { return g_err; }
extern "C" int ge_abi_version(void) { return GE_ABI_VERSION; }
#ifndef GE_SOURCE_HASH
#define GE_SOURCE_HASH "unknown"
#endif
static const char g_source_hash[] = "GE_SOURCE_HASH=" GE_SOURCE_HASH;  // the marker lets the host read the hash from the file's bytes
extern "C" const char *ge_source_hash(void) { return g_source_hash + 15; }

static const int kMaxLds = 160 * 1024;
#ifndef GE_NP_EARLY_MAX
#define GE_NP_EARLY_MAX 256  // graphs up to this size hold the n x n delay matrix in LDS (a test build lowers it to run the late path, ge_np_draws_edges, on small graphs)
#endif

// the reset kernel is instantiated per env type (ge_k_reset<ENV>): run `stmt` with ENV bound to the runtime env type
#define GE_FOR_ENV(env_type, stmt)                                                           \
  do {                                                                                       \
    switch (env_type) {                                                                      \
      case GE_SHORTEST_PATH: { constexpr int ENV = GE_SHORTEST_PATH; stmt; break; }            \
      case GE_LONGEST_PATH: { constexpr int ENV = GE_LONGEST_PATH; stmt; break; }              \
      case GE_STEINER_TREE: { constexpr int ENV = GE_STEINER_TREE; stmt; break; }              \
      case GE_TSP: { constexpr int ENV = GE_TSP; stmt; break; }                                \
      case GE_DENSEST_SUBGRAPH: { constexpr int ENV = GE_DENSEST_SUBGRAPH; stmt; break; }      \
      case GE_MAX_INDEPENDENT_SET: { constexpr int ENV = GE_MAX_INDEPENDENT_SET; stmt; break; } \
      case GE_MULTICAST_ROUTING: { constexpr int ENV = GE_MULTICAST_ROUTING; stmt; break; }    \
      case GE_DISTRIBUTION_CENTER: { constexpr int ENV = GE_DISTRIBUTION_CENTER; stmt; break; } \
      default: { constexpr int ENV = GE_PERISHABLE_DELIVERY; stmt; break; }                    \
    }                                                                                        \
  } while (0)

static int derive(const ge_config *cfg, GeParams &P, int queue_B = 0) {
  if (!cfg) return fail(GE_E_BADARG, "null config");
  memset(&P, 0, sizeof(P));
  const int t = cfg->env_type, n = cfg->n_nodes, m = cfg->n_edges;
  if (t < GE_SHORTEST_PATH || t > GE_PERISHABLE_DELIVERY) return fail(GE_E_BADARG, "unknown env_type");
  if (cfg->num_envs < 1) return fail(GE_E_BADARG, "num_envs must be >= 1");
  if (n < 3 || n > 4095) return fail(GE_E_BADARG, "n_nodes must be in [3, 4095]");
  const int ng = (t == GE_DENSEST_SUBGRAPH) ? n - 1 : n;  // densest_subgraph.py:59
  const double max_edges = (double)ng * (ng - 1) / 2.0;
  if (m < ng - 1) return fail(GE_E_BADARG, "n_edges < nodes-1: no connected graph exists (the reference would loop forever)");
  if (m > max_edges) return fail(GE_E_BADARG, "n_edges exceeds the complete graph");
  // G(n, m) is sampled by rejection until it is connected.  P(connected) ~ exp(-n e^(-2m/n)) (Erdos-Renyi): below 1e-7 the
  // reference would spin for hours per reset and a device loop of that length is a hung GPU -- refuse it loudly instead.
  if (m < max_edges && (double)ng * exp(-2.0 * (double)m / (double)ng) > 16.2)
    return fail(GE_E_UNSUPPORTED, "n_edges is so small for n_nodes that a random G(n, m) is connected with probability < 1e-7: the reference's rejection loop would not terminate in practice");
  // TSP also rejects graphs with a node of degree 1 (tsp.py:65-68): P(none) ~ exp(-n d e^(-d)) with d = 2m/n
  if (t == GE_TSP && m < max_edges && (double)n * (2.0 * m / n) * exp(-2.0 * (double)m / (double)n) > 16.2)
    return fail(GE_E_UNSUPPORTED, "n_edges is so small for n_nodes that a random G(n, m) has no degree-1 node with probability < 1e-7: the TSP rejection loop would not terminate in practice");
  // constructor asserts of the reference
  if (t == GE_SHORTEST_PATH && cfg->parenting != -1) return fail(GE_E_BADARG, "Parenting is not available for shortest path (shortest_path.py:26)");
  if (t == GE_STEINER_TREE && cfg->parenting != -1) return fail(GE_E_BADARG, "Parenting not available for this environment (steiner_tree.py:29)");
  if (t == GE_LONGEST_PATH && (cfg->parenting < 0 || cfg->parenting > 3)) return fail(GE_E_BADARG, "parenting must be in [0,1,2,3] (longest_path.py:29)");
  if (t == GE_TSP && cfg->parenting != 1 && cfg->parenting != 2) return fail(GE_E_BADARG, "Parenting must be either 1 or 2 (tsp.py:25)");
  if (t == GE_DENSEST_SUBGRAPH && cfg->parenting != 0 && cfg->parenting != 1) return fail(GE_E_BADARG, "Parenting must be 0 or 1 (densest_subgraph.py:28)");
  if (t == GE_DENSEST_SUBGRAPH && cfg->weighted) return fail(GE_E_BADARG, "Weighted graphs not supported for this env (densest_subgraph.py:29)");
  if (t == GE_TSP && cfg->spatial && !cfg->weighted) return fail(GE_E_BADARG, "Spatial TSP must be weighted (tsp.py:27)");
  if ((t == GE_STEINER_TREE || t == GE_MULTICAST_ROUTING) && (cfg->n_dests < 1 || cfg->n_dests > n - 1)) return fail(GE_E_BADARG, "n_dests must be in [1, n_nodes-1]");
  if (t == GE_PERISHABLE_DELIVERY && cfg->parenting != 1) return fail(GE_E_BADARG, "Parenting must be 1! (perishable_product_delivery.py:30)");
  if (t == GE_PERISHABLE_DELIVERY && (cfg->n_dests < 1 || cfg->n_dests > 5)) return fail(GE_E_BADARG, "Max 5 products! (perishable_product_delivery.py:35)");
  if (t == GE_PERISHABLE_DELIVERY && 2 * cfg->n_dests > n) return fail(GE_E_BADARG, "2 * n_products exceeds n_nodes: the reference would loop forever");
  if (t == GE_PERISHABLE_DELIVERY && !(cfg->dt_max >= cfg->dt_min && cfg->dt_min > 0.0)) return fail(GE_E_BADARG, "dt_min / dt_max must be the constructor's delivery-time window (0 < dt_min <= dt_max)");
  if (t == GE_DISTRIBUTION_CENTER && cfg->parenting != 1 && cfg->parenting != 2) return fail(GE_E_BADARG, "parenting must be 1 or 2 (distribution_center.py:32)");
  if (t == GE_DISTRIBUTION_CENTER && (cfg->n_dests < 0 || cfg->n_dests > n)) return fail(GE_E_BADARG, "target_count must be in [0, n_nodes]");
  if (t == GE_DISTRIBUTION_CENTER && !(cfg->max_distance >= 0.0)) return fail(GE_E_BADARG, "max_distance must be >= 0");
  if (t == GE_MULTICAST_ROUTING && (cfg->parenting < 1 || cfg->parenting > 4)) return fail(GE_E_BADARG, "Invalid parenting type (multicast_routing.py:34-35)");

  P.env_type = t; P.B = cfg->num_envs; P.n = n; P.m = m; P.E = 2 * m; P.W = (n + 63) / 64; P.ng = ng;
  const bool edge_env = (t == GE_STEINER_TREE || t == GE_MULTICAST_ROUTING);
  P.nflag = (t == GE_TSP || t == GE_MULTICAST_ROUTING) ? 4 : (t == GE_DENSEST_SUBGRAPH ? 1 : (t == GE_DISTRIBUTION_CENTER ? 5 : (t == GE_PERISHABLE_DELIVERY ? 16 : 2)));  // utils.py:32-73
  P.F = P.nflag + 5;
  P.Fe = edge_env ? 2 : 1;
  P.A = edge_env ? P.E : n;  // steiner_tree.py:117, multicast_routing.py:155-157
  P.AW = (P.A + 63) / 64;
  P.T = edge_env ? (cfg->n_dests + 1 > 2 ? cfg->n_dests + 1 : 2) : 2;
  if (t == GE_DISTRIBUTION_CENTER) { P.T = cfg->n_dests > 2 ? cfg->n_dests : 2; P.max_distance = cfg->max_distance; }
  if (t == GE_PERISHABLE_DELIVERY) { P.T = 2 * cfg->n_dests; P.dt_min = cfg->dt_min; P.dt_max = cfg->dt_max; }
  P.weighted = cfg->weighted ? 1 : 0; P.parenting = cfg->parenting; P.n_dests = cfg->n_dests;
  P.spatial = (t == GE_TSP && cfg->spatial) ? 1 : 0;
  P.is_eval = cfg->is_eval_env ? 1 : 0; P.autoreset = cfg->autoreset == 2 ? 2 : (cfg->autoreset ? 1 : 0);
  P.complete = (m >= max_edges) ? 1 : 0;
  P.div_m = ((1ull << 40) / (uint64_t)(ng > 1 ? ng - 1 : 1)) + 1ull;
  P.n_choices = (cfg->n_choices < 0) ? floor((double)n / exp(1.0)) : cfg->n_choices;  // densest_subgraph.py:38-39
  P.env_index_base = cfg->env_index_base; P.seed_stride = cfg->seed_stride;
  P.node_id_base = cfg->node_id_base;
  P.edge_row_stride = cfg->edge_row_stride > 0 ? cfg->edge_row_stride : (int64_t)cfg->num_envs * 2 * m;
  P.np_early = (t == GE_TSP || t == GE_MAX_INDEPENDENT_SET || t == GE_DENSEST_SUBGRAPH || !cfg->weighted || n <= GE_NP_EARLY_MAX) ? 1 : 0;  // nibble matrix of n*n/2 bytes <= 32 KiB
  // few slots regenerate per step when graphs are large (long episodes): 8 workgroups share a slot's sources.  The number of
  // parts fixes the order in which a node's float64 betweenness is added up, so it depends on the geometry only, not on the
  // batch size (a shard of a batch must reproduce the unsharded run bit for bit); only a partial-sum scratch beyond 16 GiB
  // halves it
  P.feat_parts = 1;
  if (n > 64) { int parts = 8; while (parts > 1 && (int64_t)cfg->num_envs * parts * n * 8 > (16ll << 30)) parts >>= 1; P.feat_parts = parts; }
  if (P.complete && ng == n) P.feat_parts = 1;  // no BFS sources to share: betweenness and closeness of a complete graph are constants
  if (!P.complete && m > 65535) return fail(GE_E_TOOBIG, "n_edges > 65535 for a non-complete graph");
  if (P.E > (1 << 24)) return fail(GE_E_TOOBIG, "too many edges");
  if (cfg->num_envs > 8192 * GE_STEP_BLOCK) return fail(GE_E_TOOBIG, "num_envs > 2M per engine");
  P.nocolw = (t == GE_TSP && P.complete && ng == n && !P.is_eval && !P.spatial && !getenv("GE_KEEP_COLW")) ? 1 : 0;
  P.nowsort = (P.nocolw && n > 64) ? 1 : 0;
  ge_make_lds(P, queue_B > 0 ? queue_B : P.B);
  ge_make_ldsf(P, queue_B > 0 ? queue_B : P.B);
  ge_tune_feat_parts(P);
  if (P.lds.total > kMaxLds || P.ldsf.total > kMaxLds) return fail(GE_E_TOOBIG, "per-env graph does not fit 160 KiB of LDS");
  return GE_OK;
}

// per-slot work space of the sequential is_eval_env baselines (ge_tsp_eval.h); 0 = none
static uint64_t eval_slot_bytes(const GeParams &P) {
  if (!P.is_eval) return 0;
  if (P.env_type == GE_TSP) return ge_ch_slot_bytes(P.n);
  if (P.env_type == GE_MAX_INDEPENDENT_SET && !P.weighted) return ge_cr_slot_bytes(P.n, P.m);
  if (P.env_type == GE_STEINER_TREE && P.n_dests > 1 && P.n_dests < P.n - 1) return ge_steiner_slot_bytes(P.n, P.m, P.T);  // steiner_tree.py:78-87: the Kou branch
  return 0;
}

extern "C" int ge_get_layout(const ge_config *cfg, ge_layout *out) {
  GeParams P;
  int rc = derive(cfg, P);
  if (rc != GE_OK) return rc;
  if (!out) return fail(GE_E_BADARG, "null layout");
  out->F = P.F; out->Fe = P.Fe; out->A = P.A; out->W = P.W; out->E = P.E;
  out->total_nodes = (int64_t)P.B * P.n; out->total_edges = (int64_t)P.B * P.E;
  out->obs_len = (int64_t)P.n * P.F + (int64_t)P.E * P.Fe + 2 * (int64_t)P.E;
  out->reset_lds_bytes = P.lds.total;
  out->feat_parts = P.feat_parts;
  out->eval_scratch_bytes = (int64_t)((uint64_t)P.B * eval_slot_bytes(P));
  out->prune_scratch_words = ((P.env_type == GE_LONGEST_PATH || P.env_type == GE_TSP) && P.parenting >= 2 && P.W > GE_MAXW) ? (int64_t)P.B * 4 * P.W : 0;
  return GE_OK;
}

extern "C" int ge_destroy(ge_engine *e);

// the multi-class engine is instantiated for the env ids of BASELINE config 5 (ShortestPath, DensestSubgraph, MaxIndependentSet)
#define GE_FOR_RAGGED_ENV(env_type, stmt)                                                   \
  do {                                                                                      \
    switch (env_type) {                                                                     \
      case GE_SHORTEST_PATH: { constexpr int ENV = GE_SHORTEST_PATH; stmt; break; }           \
      case GE_DENSEST_SUBGRAPH: { constexpr int ENV = GE_DENSEST_SUBGRAPH; stmt; break; }     \
      default: { constexpr int ENV = GE_MAX_INDEPENDENT_SET; stmt; break; }                   \
    }                                                                                       \
  } while (0)

static int check_buffers(const GeParams &P, const ge_buffers *bufs) {
  const void *need[] = {bufs->x, bufs->edge_index, bufs->edge_attr, bufs->row_ptr, bufs->colw, bufs->scode, bufs->adj_bits, bufs->slot_rec,
                        bufs->terminals, bufs->node_bits, bufs->target_bits, bufs->counters, bufs->seed,
                        bufs->episode, bufs->heuristic, bufs->mt_state, bufs->mask, bufs->mask_bits, bufs->reward,
                        bufs->terminated, bufs->invalid, bufs->solved, bufs->final_cost, bufs->final_heur, bufs->final_len,
                        bufs->reset_list, bufs->reset_count, bufs->work_list, bufs->work_count};
  for (size_t k = 0; k < sizeof(need) / sizeof(need[0]); k++) if (!need[k]) return fail(GE_E_BADARG, "a required device buffer is null");
  if (P.env_type == GE_STEINER_TREE && !bufs->rev_edge) return fail(GE_E_BADARG, "SteinerTree needs rev_edge");
  if (P.env_type == GE_DISTRIBUTION_CENTER && (!bufs->range_bits || !bufs->cover_bits)) return fail(GE_E_BADARG, "DistributionCenter needs range_bits and cover_bits");
  if (P.env_type == GE_DISTRIBUTION_CENTER && P.n <= 64 && !bufs->aux_bits) return fail(GE_E_BADARG, "DistributionCenter with n_nodes <= 64 needs aux_bits (which rows of range_bits exist)");
  if (P.env_type == GE_MULTICAST_ROUTING && P.parenting == 2 && !bufs->rev_edge) return fail(GE_E_BADARG, "MulticastRouting parenting 2 needs rev_edge");
  if (P.env_type == GE_MULTICAST_ROUTING && P.parenting >= 3 && !bufs->node_aux) return fail(GE_E_BADARG, "MulticastRouting parenting >= 3 needs node_aux");
  if (P.feat_parts > 1 && !bufs->feat_scratch) return fail(GE_E_BADARG, "feat_scratch required (ge_layout.feat_parts > 1)");
  if (P.spatial && !bufs->sw64) return fail(GE_E_BADARG, "spatial TSP needs sw64");
  if (eval_slot_bytes(P) && !bufs->eval_scratch) return fail(GE_E_BADARG, "is_eval_env of TSP / unweighted MaxIndependentSet / SteinerTree (1 < n_dests < n - 1) needs eval_scratch (ge_layout.eval_scratch_bytes)");
  if (P.W == 1 && !bufs->node_rec) return fail(GE_E_BADARG, "n_nodes <= 64 needs node_rec");
  if ((P.env_type == GE_LONGEST_PATH || P.env_type == GE_TSP) && P.parenting >= 2 && P.W > GE_MAXW && !bufs->prune_scratch)
    return fail(GE_E_BADARG, "parenting >= 2 on more than 512 nodes needs prune_scratch ([B, 4, W] uint64)");
  return GE_OK;
}

// launch geometry and LDS limits from e->P (uniform engine) or from the class table (multi-class engine); deletes e on failure
static int finish_create(ge_engine *e, ge_engine **out) {
  const GeParams &P = e->P;
  const int nblk = (P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK;
  const bool rg = e->n_classes > 0;
  // ---- graph kernel
  int reset_lds = P.lds.total, gen_lds = P.ldsf.total, f64_body = 0;
  bool any64 = !rg && P.n <= 64 && !P.spatial;
  if (!rg) f64_body = ge_f64_pre_off(P.E, P.env_type == GE_TSP, nblk);
  for (const GeParams &C : e->classes) {
    if (C.lds.total > reset_lds) reset_lds = C.lds.total;
    if (C.ldsf.total > gen_lds) gen_lds = C.ldsf.total;
    if (C.n <= 64) { any64 = true; const int b = ge_f64_pre_off(C.E, 0, nblk); if (b > f64_body) f64_body = b; }
  }
  if (rg && e->P.lds.pre != 0) { e->P.lds.pre = ge_align16(reset_lds); reset_lds = e->P.lds.pre + (nblk + 2) * 4; }  // prefix behind every class's scratch
  e->lds_bytes = reset_lds;
  e->lds_bytes_inject = reset_lds;
  if (!rg && P.nocolw) { GeParams Pi = P; Pi.nocolw = 0; Pi.nowsort = 0; ge_make_lds(Pi, P.B); e->lds_bytes_inject = Pi.lds.total; }
  if (reset_lds > kMaxLds || gen_lds > kMaxLds || e->lds_bytes_inject > kMaxLds) { delete e; return fail(GE_E_TOOBIG, "per-env graph does not fit 160 KiB of LDS"); }
  int per_cu = kMaxLds / (reset_lds > 0 ? reset_lds : 1);
  if (per_cu > 16) per_cu = 16;
  if (per_cu < 1) per_cu = 1;
  e->reset_grid = 256 * per_cu;
  if (e->reset_grid > P.B) e->reset_grid = P.B;
  e->nseed = (e->reset_grid + 63) / 64;  // one seeding workgroup per 64 regenerating workgroups: the usual queue fits one round of both
  hipError_t hr = hipSuccess;
  if (reset_lds > 64 * 1024 || e->lds_bytes_inject > 64 * 1024) {
    const int most = reset_lds > e->lds_bytes_inject ? reset_lds : e->lds_bytes_inject;
    if (rg) GE_FOR_RAGGED_ENV(P.env_type, hr = (hipError_t)GE_SET_MAX_DYN_LDS((ge_k_reset<ENV, true>), reset_lds));
    else GE_FOR_ENV(P.env_type, hr = (hipError_t)GE_SET_MAX_DYN_LDS((ge_k_reset<ENV, false>), most));
    if (hr != hipSuccess) { delete e; return fail(GE_E_LAUNCH, "cannot raise the dynamic LDS limit of the reset kernel"); }
  }
  // ---- quad-per-slot step kernel of the edge-action envs: its LDS stage (mask rows + node sets of 256 slots) passes 64 KB
  if (!rg && (P.env_type == GE_STEINER_TREE || P.env_type == GE_MULTICAST_ROUTING) && ge_edge_fits(P.AW, P.W) && ge_edge_lds_bytes(P.AW, P.W) > 64 * 1024) {
    const int bytes = (int)ge_edge_lds_bytes(P.AW, P.W);
    if (P.env_type == GE_STEINER_TREE) { hr = (hipError_t)GE_SET_MAX_DYN_LDS((ge_k_step_edge<GE_STEINER_TREE, true>), bytes); if (hr == hipSuccess) hr = (hipError_t)GE_SET_MAX_DYN_LDS((ge_k_step_edge<GE_STEINER_TREE, false>), bytes); }
    else { hr = (hipError_t)GE_SET_MAX_DYN_LDS((ge_k_step_edge<GE_MULTICAST_ROUTING, true>), bytes); if (hr == hipSuccess) hr = (hipError_t)GE_SET_MAX_DYN_LDS((ge_k_step_edge<GE_MULTICAST_ROUTING, false>), bytes); }
    if (hr != hipSuccess) { delete e; return fail(GE_E_LAUNCH, "cannot raise the dynamic LDS limit of the edge step kernel"); }
  }
  // ---- thread-per-slot step kernel: its LDS stage (one node set per slot of the workgroup) passes 64 KB above 2 048 nodes
  if (!rg && (size_t)GE_STEP_BLOCK * P.W * 8 + GE_STEP_BLOCK + 64 > 64 * 1024) {
    const int bytes = (int)((size_t)GE_STEP_BLOCK * P.W * 8 + GE_STEP_BLOCK + 64);
    const bool pr = (P.env_type == GE_LONGEST_PATH || P.env_type == GE_TSP) && P.parenting >= 2;
    if (pr && P.env_type == GE_TSP) { hr = (hipError_t)GE_SET_MAX_DYN_LDS((ge_k_step<GE_TSP, true, false, 2>), bytes); if (hr == hipSuccess) hr = (hipError_t)GE_SET_MAX_DYN_LDS((ge_k_step<GE_TSP, false, false, 2>), bytes); }
    else if (pr) { hr = (hipError_t)GE_SET_MAX_DYN_LDS((ge_k_step<GE_LONGEST_PATH, true, false, 2>), bytes); if (hr == hipSuccess) hr = (hipError_t)GE_SET_MAX_DYN_LDS((ge_k_step<GE_LONGEST_PATH, false, false, 2>), bytes); }
    else { GE_FOR_ENV(P.env_type, hr = (hipError_t)GE_SET_MAX_DYN_LDS((ge_k_step<ENV, true, false, 0>), bytes)); if (hr == hipSuccess) GE_FOR_ENV(P.env_type, hr = (hipError_t)GE_SET_MAX_DYN_LDS((ge_k_step<ENV, false, false, 0>), bytes)); }
    if (hr != hipSuccess) { delete e; return fail(GE_E_LAUNCH, "cannot raise the dynamic LDS limit of the step kernel"); }
  }
  // ---- feature kernels
  e->feat_fast = any64 ? 1 : 0;  // (spatial TSP: float64 weights do not fit the fast path's LDS)
  e->feat64_pre_off = ge_align16(f64_body);  // the item bodies (the queue prefix overlays them), then the tail {item slots, overflow flag}
  e->feat_lds = e->feat_fast ? e->feat64_pre_off + GE_F64_ITEMS * 4 * 4 + 16 : gen_lds;
  e->gen_lds = gen_lds;
  e->gen_pre_off = P.ldsf.pre;
  if (rg) { e->gen_pre_off = ge_align16(gen_lds); e->gen_lds = e->gen_pre_off + (nblk + 2) * 4; }
  if (e->feat_lds > kMaxLds || e->gen_lds > kMaxLds) { delete e; return fail(GE_E_TOOBIG, "feature kernel does not fit LDS"); }
  if (e->gen_lds > 64 * 1024) {
    hr = rg ? (hipError_t)GE_SET_MAX_DYN_LDS(ge_k_features<true>, e->gen_lds) : (hipError_t)GE_SET_MAX_DYN_LDS(ge_k_features<false>, e->gen_lds);
    if (hr != hipSuccess) { delete e; return fail(GE_E_LAUNCH, "cannot raise the dynamic LDS limit of the feature kernel"); }
  }
  if (e->feat_fast && e->feat_lds > 64 * 1024) {
    hr = rg ? (hipError_t)GE_SET_MAX_DYN_LDS(ge_k_features64<true>, e->feat_lds) : (hipError_t)GE_SET_MAX_DYN_LDS(ge_k_features64<false>, e->feat_lds);
    if (hr != hipSuccess) { delete e; return fail(GE_E_LAUNCH, "cannot raise the dynamic LDS limit of the n<=64 feature kernel"); }
  }
  { int per = kMaxLds / e->gen_lds; if (per > 16) per = 16; if (per < 1) per = 1; e->gen_grid = 256 * per; if (e->gen_grid > P.B) e->gen_grid = P.B; }
  { int per = kMaxLds / e->feat_lds; if (per > 16) per = 16; if (per < 1) per = 1; e->feat_grid = 256 * per; if (e->feat_grid > P.B) e->feat_grid = P.B; }
  *out = e;
  return GE_OK;
}

extern "C" int ge_create(const ge_config *cfg, const ge_buffers *bufs, ge_engine **out) {
  if (!bufs || !out) return fail(GE_E_BADARG, "null argument");
  GeParams P;
  int rc = derive(cfg, P);
  if (rc != GE_OK) return rc;
  rc = check_buffers(P, bufs);
  if (rc != GE_OK) return rc;
  P.buf = *bufs;
  ge_engine *e = new (std::nothrow) ge_engine();
  if (!e) return fail(GE_E_BADARG, "out of host memory");
  e->P = P; e->cfg = *cfg; e->have_events = false;
  e->loaded = false; e->seeded = false; e->streams = false;
  e->spares = false; e->period = 0; e->swap_parts = 1; e->pending_calls = 0; memset(&e->RS, 0, sizeof(e->RS));
  e->n_classes = 0; memset(&e->R, 0, sizeof(e->R));
  return finish_create(e, out);
}

extern "C" int64_t ge_ragged_table_bytes(int32_t n_classes) { return (int64_t)sizeof(GeParams) * (n_classes > 0 ? n_classes : 0); }

extern "C" int ge_create_ragged(const ge_config *cfgs, const ge_buffers *bufs, int32_t n_classes, void *class_table,
                                int32_t *slot_class, int32_t *class_start, ge_engine **out) {
  if (!cfgs || !bufs || !out || !class_table || !slot_class || !class_start || n_classes < 1) return fail(GE_E_BADARG, "null argument");
  const int t = cfgs[0].env_type;
  if (t != GE_SHORTEST_PATH && t != GE_DENSEST_SUBGRAPH && t != GE_MAX_INDEPENDENT_SET)
    return fail(GE_E_UNSUPPORTED, "the multi-class engine is built for ShortestPath, DensestSubgraph and MaxIndependentSet (BASELINE config 5)");
  int64_t total = 0;
  for (int c = 0; c < n_classes; c++) total += cfgs[c].num_envs;
  if (cfgs[0].is_eval_env && t == GE_MAX_INDEPENDENT_SET && !cfgs[0].weighted)
    return fail(GE_E_UNSUPPORTED, "is_eval_env of unweighted MaxIndependentSet (the clique-removal baseline, a launch per uniform engine) is not built for the multi-class engine");
  if (total > 8192 * GE_STEP_BLOCK) return fail(GE_E_TOOBIG, "num_envs > 2M per engine");
  ge_engine *e = new (std::nothrow) ge_engine();
  if (!e) return fail(GE_E_BADARG, "out of host memory");
  e->classes.resize(n_classes);
  std::vector<int32_t> start(n_classes + 1, 0), cls_of((size_t)total);
  int wmin = 8, widest = 0;
  for (int c = 0; c < n_classes; c++) {
    GeParams &C = e->classes[c];
    int rc = derive(&cfgs[c], C, (int)total);
    if (rc == GE_OK) rc = check_buffers(C, &bufs[c]);
    if (rc == GE_OK && (cfgs[c].env_type != t || cfgs[c].autoreset != cfgs[0].autoreset || cfgs[c].seed_stride != cfgs[0].seed_stride))
      rc = fail(GE_E_BADARG, "the classes of a multi-class engine share env_type, autoreset and seed_stride");
    if (rc == GE_OK && (cfgs[c].env_index_base != cfgs[0].env_index_base + start[c] || bufs[c].seed != bufs[0].seed + start[c] ||
                        bufs[c].episode != bufs[0].episode + start[c] || bufs[c].mt_state != bufs[0].mt_state + (int64_t)start[c] * GE_SEED_DEPTH * 2 * GE_MT_N))
      rc = fail(GE_E_BADARG, "classes follow one another in slot order: env_index_base, seed, episode and mt_state of class c start at its first global slot");
    // the engine-wide kernels take the queues and the work lists from class 0 and index slot_rec by global slot through the class's pointer
    if (rc == GE_OK && (bufs[c].reset_list != bufs[0].reset_list || bufs[c].reset_count != bufs[0].reset_count || bufs[c].work_list != bufs[0].work_list ||
                        bufs[c].work_count != bufs[0].work_count || bufs[c].slot_rec != bufs[0].slot_rec + 2 * (int64_t)start[c]))
      rc = fail(GE_E_BADARG, "reset_list, reset_count, work_list and work_count are engine-wide (the same pointers in every class), and slot_rec of class c starts at its first global slot");
    if (rc != GE_OK) { delete e; return rc; }
    C.buf = bufs[c];
    start[c + 1] = start[c] + cfgs[c].num_envs;
    for (int i = start[c]; i < start[c + 1]; i++) cls_of[(size_t)i] = c;
    if (C.n > e->classes[widest].n) widest = c;
  }
  (void)wmin;
  // one launch geometry of the generic feature kernel per LDS bucket: the wave count every class of the bucket can hold -- two
  // workgroups per CU where that leaves at least four waves, else one
  memset(e->bk, 0, sizeof(e->bk));
  const int nblk_all = (int)((total + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK);
  for (int b = 0; b < GE_MAX_BUCKETS; b++) {
    // every class takes the wave count it would choose as a uniform engine; if the bucket's widest allocation then leaves room for
    // ONE workgroup per CU only, the smaller classes are re-derived for the whole CU (up to 16 waves: idle LDS otherwise).  The
    // launch has the threads of the largest count; a class's surplus waves do not search (ge_features_generic_env)
    int waves = 1; bool any = false, anygen = false;
    GeBucket &K = e->bk[b];
    for (int pass = 0; pass < 2; pass++) {
      waves = 1; K.gen_lds = 0; K.reset_lds = 0;
      for (GeParams &C : e->classes) if (bucket_of(C.n) == b) {
        any = true;
        C.bucket = b;
        if (pass == 0) ge_make_ldsf(C, (int)total); else ge_make_ldsf(C, (int)total, 0, 160 * 1024 - 2048);
        ge_tune_feat_parts(C);
        if (C.lds.total > K.reset_lds) K.reset_lds = C.lds.total;
        if (C.n > 64) { anygen = true; if (C.ldsf.total > K.gen_lds) K.gen_lds = C.ldsf.total; if (C.ldsf.waves > waves) waves = C.ldsf.waves; }
      }
      if (2 * (K.gen_lds + (nblk_all + 2) * 4) <= kMaxLds) break;  // two workgroups per CU: the classes keep their own choice
    }
    K.used = any; K.gen_used = anygen; K.gen_waves = waves;
    if (!any) continue;
    if (K.reset_lds < GE_SEED_LDS_BYTES) K.reset_lds = GE_SEED_LDS_BYTES;
    { int per = kMaxLds / K.reset_lds; if (per > 16) per = 16; if (per < 1) per = 1; K.reset_grid = 256 * per; if (K.reset_grid > total) K.reset_grid = (int)total; }
    if (anygen) {
      K.gen_pre_off = ge_align16(K.gen_lds); K.gen_lds = K.gen_pre_off + (nblk_all + 2) * 4;
      int per = kMaxLds / K.gen_lds; if (per > 16) per = 16; if (per < 1) per = 1; K.gen_grid = 256 * per; if (K.gen_grid > total) K.gen_grid = (int)total;
    }
  }
  // engine-wide block: the widest class's geometry (LDS stage of the step kernel), all slots, the global arrays of class 0
  e->P = e->classes[widest];
  e->P.B = (int32_t)total;
  for (const GeParams &C : e->classes) if (C.feat_parts > e->P.feat_parts) e->P.feat_parts = C.feat_parts;
  e->P.buf = bufs[0];
  e->P.env_index_base = cfgs[0].env_index_base;
  e->cfg = cfgs[0]; e->cfg.num_envs = (int32_t)total;
  e->have_events = false; e->loaded = false; e->seeded = false; e->streams = false;
  e->spares = false; e->period = 0; e->swap_parts = 1; e->pending_calls = 0; memset(&e->RS, 0, sizeof(e->RS));
  e->n_classes = n_classes;
  if (hipMemcpy(class_table, e->classes.data(), sizeof(GeParams) * (size_t)n_classes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(slot_class, cls_of.data(), sizeof(int32_t) * (size_t)total, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(class_start, start.data(), sizeof(int32_t) * (size_t)(n_classes + 1), hipMemcpyHostToDevice) != hipSuccess) {
    delete e; return fail(GE_E_LAUNCH, "cannot copy the class table to the device");
  }
  e->R.classes = (const GeParams *)class_table; e->R.slot_class = slot_class; e->R.class_start = class_start; e->R.n_classes = n_classes;
  return finish_create(e, out);
}

// the per-slot slabs an image must hold for this (sub-)engine
static int check_image(const GeParams &P, const ge_buffers &I) {
  const void *need[] = {I.x, I.edge_index, I.edge_attr, I.row_ptr, I.colw, I.scode, I.adj_bits, I.slot_rec, I.terminals, I.node_bits,
                        I.target_bits, I.counters, I.heuristic, I.mask, I.mask_bits};
  for (size_t k = 0; k < sizeof(need) / sizeof(need[0]); k++) if (!need[k]) return fail(GE_E_BADARG, "a required slab of the spare image is null");
  const ge_buffers &G = P.buf;
  if ((G.sw64 && !I.sw64) || (G.node_rec && !I.node_rec) || (G.rev_edge && !I.rev_edge) || (G.aux_bits && !I.aux_bits) || (G.node_aux && !I.node_aux) ||
      (G.range_bits && !I.range_bits) || (G.cover_bits && !I.cover_bits))
    return fail(GE_E_BADARG, "the spare image lacks an optional slab the engine has (sw64 / node_rec / rev_edge / aux_bits / node_aux / range_bits / cover_bits)");
  return GE_OK;
}
// V = P seen through image I: per-slot slabs from the image, everything that sequences the engine shared with the live view
static GeParams image_view(const GeParams &P, const ge_buffers &I, int32_t *refill_list, int32_t *refill_count) {
  GeParams V = P;
  ge_buffers B = I;
  const ge_buffers &G = P.buf;
  B.seed = G.seed; B.episode = G.episode; B.mt_state = G.mt_state;
  B.work_list = G.work_list; B.work_count = G.work_count; B.feat_scratch = G.feat_scratch; B.eval_scratch = G.eval_scratch;
  B.reset_list = refill_list; B.reset_count = refill_count;
  B.reward = G.reward; B.terminated = G.terminated; B.invalid = G.invalid; B.solved = G.solved;
  B.final_cost = G.final_cost; B.final_heur = G.final_heur; B.final_len = G.final_len;  // (not written by a refill)
  B.actions_out = nullptr; B.stream_state = nullptr;
  V.buf = B;
  return V;
}

extern "C" int ge_attach_spares(ge_engine *e, const ge_spares *sp, void *class_table_spare) {
  if (!e || !sp) return fail(GE_E_BADARG, "null argument");
  if (e->spares) return fail(GE_E_STATE, "spares are already attached");
  if (!e->P.autoreset) return fail(GE_E_BADARG, "spares serve autoreset: this engine freezes finished slots");
  if (e->P.buf.stream_state) return fail(GE_E_UNSUPPORTED, "spares and stream_state (reset(seed=None) continuing the streams) exclude each other: an image generated ahead of time would leave the streams of the wrong episode behind");
  if (!sp[0].state || !sp[0].swap_list || !sp[0].swap_count || !sp[0].refill_list || !sp[0].refill_count) return fail(GE_E_BADARG, "state / swap_list / swap_count / refill_list / refill_count are required");
  if (sp[0].period < 1) return fail(GE_E_BADARG, "period must be >= 1");
  if (e->n_classes > 0 && !class_table_spare) return fail(GE_E_BADARG, "a multi-class engine needs class_table_spare (ge_ragged_table_bytes)");
  int rc = GE_OK;
  if (e->n_classes == 0) rc = check_image(e->P, sp[0].image);
  for (int c = 0; c < e->n_classes && rc == GE_OK; c++) rc = check_image(e->classes[c], sp[c].image);
  if (rc != GE_OK) return rc;
  e->refill_list = sp[0].refill_list; e->refill_count = sp[0].refill_count;
  e->PS = image_view(e->P, sp[0].image, e->refill_list, e->refill_count);
  if (e->n_classes > 0) {
    e->classesS.resize(e->n_classes);
    for (int c = 0; c < e->n_classes; c++) e->classesS[c] = image_view(e->classes[c], sp[c].image, e->refill_list, e->refill_count);
    if (hipMemcpy(class_table_spare, e->classesS.data(), sizeof(GeParams) * (size_t)e->n_classes, hipMemcpyHostToDevice) != hipSuccess)
      return fail(GE_E_LAUNCH, "cannot copy the spare class table to the device");
    e->RS = e->R; e->RS.classes = (const GeParams *)class_table_spare;
  }
  e->P.spare_state = sp[0].state; e->P.swap_list = sp[0].swap_list; e->P.swap_count = sp[0].swap_count;
  e->PS.spare_state = sp[0].state;
  e->period = sp[0].period; e->pending_calls = 0;
  // a slot's image is some tens of KB: split it over workgroups so that a handful of finished slots is not a handful of workgroups
  const GeParams &P = e->P;
  const int64_t slot_bytes = (int64_t)P.n * P.F * 4 + (int64_t)P.E * (16 + 4 * P.Fe + 2 + 1 + (P.buf.rev_edge ? 4 : 0)) + (int64_t)P.n * P.W * 8 * (P.buf.range_bits ? 2 : 1) + P.A;
  int parts = (int)(slot_bytes / 8192); if (parts < 1) parts = 1; if (parts > 16) parts = 16;
  if (parts >= 4) parts = GE_SWAP_GROUPS;  // a large image: one workgroup per (group of) whole arrays
  e->swap_parts = parts;
  e->spares = true;
  // (the caller zeroes `state`; ge_reset / ge_inject_state clear it and refill every image)
  return GE_OK;
}

extern "C" int ge_destroy(ge_engine *e) {
  if (!e) return GE_OK;
  if (e->have_events) for (int k = 0; k < 4; k++) (void)hipEventDestroy(e->ev[k]);
  delete e;
  return GE_OK;
}

static int check_launch(const char *what) {
  // GE_DEBUG_SYNC=1: wait for every launch and name it (a kernel fault is otherwise reported at some later synchronisation)
  static const bool debug_sync = getenv("GE_DEBUG_SYNC") != nullptr;
  if (debug_sync) {
    fprintf(stderr, "[graphenvs] %s ...\n", what); fflush(stderr);
    const hipError_t hs = hipDeviceSynchronize();  // an asynchronous fault of THIS launch surfaces here
    fprintf(stderr, "[graphenvs] %s %s\n", what, hs == hipSuccess ? "done" : hipGetErrorString(hs)); fflush(stderr);
    if (hs != hipSuccess) { snprintf(g_err, sizeof(g_err), "%s: %s (GE_DEBUG_SYNC)", what, hipGetErrorString(hs)); return GE_E_LAUNCH; }
  }
  hipError_t hr = hipGetLastError();
  if (hr != hipSuccess) { snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(hr)); return GE_E_LAUNCH; }
  return GE_OK;
}

// ring entries jlo .. GE_SEED_DEPTH - 1 of every slot from seeds[] (+ j * seed_stride), on the caller's stream
static int launch_seed(ge_engine *e, const uint32_t *seeds, int jlo, void *stream) {
  const int64_t items = (int64_t)(GE_SEED_DEPTH - jlo) * e->P.B;
  int64_t grid = (items + GE_WAVE - 1) / GE_WAVE;
  if (grid > 8192) grid = 8192;
  GE_LAUNCH(ge_k_seed, (int)grid, 2 * GE_WAVE, GE_SEED_LDS_BYTES, stream, e->P, seeds, jlo);
  return check_launch("seed kernel");
}

// V: the engine (e->P) or its spare-image view (e->PS); VR: the matching class table
static int launch_combine(ge_engine *e, const GeParams &V, const GeRagged &VR, GeRun run, bool small, void *stream) {
  const bool rg = e->n_classes > 0;
  size_t lds = (size_t)((V.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK + 2) * 4;
  int64_t items = (int64_t)(run.items == GE_ITEMS_ALL ? V.B : (small ? 64 : 4096)) * V.n;
  int grid = (int)((items + 255) / 256); if (grid > 8192) grid = 8192;
  if (rg) GE_LAUNCH(ge_k_feat_combine<true>, grid, 256, lds, stream, V, VR, run);
  else GE_LAUNCH(ge_k_feat_combine<false>, grid, 256, lds, stream, V, VR, run);
  return check_launch("feature combine kernel");
}

// `small`: the queue is expected to be short (the in-place regenerations of an engine with spares): a fraction of the grid
static int launch_features(ge_engine *e, const GeParams &V, const GeRagged &VR, GeRun run, bool small, void *stream) {
  int rc = GE_OK;
  const bool rg = e->n_classes > 0, queue = run.items == GE_ITEMS_QUEUE;
  int fgrid = queue ? e->feat_grid : (V.B < e->feat_grid * 4 ? V.B : e->feat_grid * 4);
  if (small && fgrid > 64) fgrid = 64;
  const int gen_threads = GE_WAVE * (rg ? e->classes[0].ldsf.waves : V.ldsf.waves);
  if (e->feat_fast) {
    if (rg) GE_LAUNCH(ge_k_features64<true>, fgrid, GE_F64_THREADS, e->feat_lds, stream, V, VR, run, e->feat64_pre_off);
    else GE_LAUNCH(ge_k_features64<false>, fgrid, GE_F64_THREADS, e->feat_lds, stream, V, VR, run, e->feat64_pre_off);
    rc = check_launch("feature kernel (n <= 64)");
    if (rc != GE_OK) return rc;
    // the fast path's fallback list (normally empty); multi-class engine: every slot of a class with n > 64, feat_parts workgroups each
    if (rg) {
      for (int b = 0; b < GE_MAX_BUCKETS && rc == GE_OK; b++) {  // one launch per LDS bucket that has classes with n > 64
        const GeBucket &K = e->bk[b];
        if (!K.gen_used) continue;
        int64_t want = (int64_t)K.gen_grid * ge_feat_workgroups(V.feat_parts) * (queue ? 1 : 4);
        if (small && want > 288) want = 288;
        if (want > 65535 * 16) want = 65535 * 16;
        GE_LAUNCH(ge_k_features<true>, (int)want, GE_WAVE * K.gen_waves, K.gen_lds, stream, V, VR, as_list(run), K.gen_pre_off, b);
        rc = check_launch("feature kernel (list)");
      }
      return (rc == GE_OK && V.feat_parts > 1) ? launch_combine(e, V, VR, as_list(run), small, stream) : rc;
    }
    int g2 = e->gen_grid < 8 ? e->gen_grid : 8;  // (the list is normally empty and a handful of slots at most: eight workgroups dispatch in less time than 64 -- the launch is on every step's critical path)
    GE_LAUNCH(ge_k_features<false>, g2, gen_threads, e->gen_lds, stream, V, VR, as_list(run), e->gen_pre_off, -1);
    return check_launch("feature kernel (fallback list)");
  }
  {
    int64_t want = (int64_t)fgrid * ge_feat_workgroups(V.feat_parts);
    if (queue && !run.refill && want > 4608) want = 4608;  // the list is short, workgroups stride over it
    if (small && want > 288) want = 288;
    if (want > 65535 * 16) want = 65535 * 16;
    if (rg) {  // (no class with n <= 64: every slot takes the generic kernel) one launch per LDS bucket
      for (int b = 0; b < GE_MAX_BUCKETS && rc == GE_OK; b++) {
        const GeBucket &K = e->bk[b];
        if (!K.gen_used) continue;
        GE_LAUNCH(ge_k_features<true>, (int)want, GE_WAVE * K.gen_waves, K.gen_lds, stream, V, VR, run, K.gen_pre_off, b);
        rc = check_launch("feature kernel");
      }
    }
    else GE_LAUNCH(ge_k_features<false>, (int)want, gen_threads, e->gen_lds, stream, V, VR, run, e->gen_pre_off, -1);
  }
  rc = check_launch("feature kernel");
  if (rc != GE_OK || V.feat_parts == 1) return rc;
  return launch_combine(e, V, VR, run, small, stream);
}

// is_eval_env baselines that are sequential programs (ge_tsp_eval.h: TSP Christofides, MaxIndependentSet clique removal) for the
// regenerated slots, on the slabs the graph kernel wrote
static int launch_seq_baseline(ge_engine *e, const GeParams &P, int queue, void *stream) {
  const uint64_t slot_bytes = eval_slot_bytes(P);
  const int nblk = (P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK;
  (void)e;
  if (P.env_type == GE_MAX_INDEPENDENT_SET) {
    int g = (P.B + GE_TSP_EVAL_THREADS - 1) / GE_TSP_EVAL_THREADS; if (g > 4096) g = 4096;
    GE_LAUNCH(ge_k_mis_baseline, g, GE_TSP_EVAL_THREADS, (nblk + 2) * 4, stream, P, queue, (uint8_t *)P.buf.eval_scratch, slot_bytes);
    return check_launch("MaxIndependentSet baseline kernel");
  }
  if (P.env_type == GE_STEINER_TREE) {
    int g = (P.B + GE_TSP_EVAL_THREADS - 1) / GE_TSP_EVAL_THREADS; if (g > 4096) g = 4096;
    GE_LAUNCH(ge_k_steiner_baseline, g, GE_TSP_EVAL_THREADS, (nblk + 2) * 4, stream, P, queue, (uint8_t *)P.buf.eval_scratch, slot_bytes);
    return check_launch("SteinerTree baseline kernel");
  }
  const int pre_off = GE_WAVE * P.W * 8;
  int grid = P.B < 2048 ? P.B : 2048;
  GE_LAUNCH(ge_k_tsp_closure, grid, GE_TSP_EVAL_THREADS, pre_off + (nblk + 2) * 4, stream, P, queue, (uint8_t *)P.buf.eval_scratch, slot_bytes, pre_off);
  int rc = check_launch("TSP baseline: closure kernel");
  if (rc != GE_OK) return rc;
  grid = (P.B + GE_TSP_EVAL_THREADS - 1) / GE_TSP_EVAL_THREADS; if (grid > 4096) grid = 4096;
  GE_LAUNCH(ge_k_tsp_tour, grid, GE_TSP_EVAL_THREADS, (nblk + 2) * 4, stream, P, queue, (uint8_t *)P.buf.eval_scratch, slot_bytes);
  return check_launch("TSP baseline: tour kernel");
}

// One pass of the reset path over the items `run` names, on the engine (V = e->P) or on its spare image (V = e->PS, run.refill).
// Everything on the caller's stream, in order: generator states (full reset: the ring of every slot; queue launches: seeding
// workgroups inside the reset launch), graph kernel, sequential baselines, feature kernel(s).
static int launch_reset(ge_engine *e, const GeParams &V, const GeRagged &VR, const uint32_t *seeds, GeRun run, const GeInject &inj, bool small, void *stream) {
  int rc = GE_OK;
  const bool queue = run.items == GE_ITEMS_QUEUE;
  if (run.restart == 1) rc = launch_seed(e, seeds, 0, stream);
  else if (run.restart == 2) rc = launch_seed(e, inj.seeds, 1, stream);  // the injected episode needs no states of its own
  if (rc != GE_OK) return rc;
  if (run.cont) {
    int64_t g = ((int64_t)V.B + GE_WAVE - 1) / GE_WAVE; if (g > 8192) g = 8192;
    GE_LAUNCH(ge_k_seed_next, (int)g, 2 * GE_WAVE, GE_SEED_LDS_BYTES, stream, V);
    rc = check_launch("seed kernel (next ring entry)");
    if (rc != GE_OK) return rc;
  }
  int rgrid = e->reset_grid, nseed = e->nseed;
  if (small && rgrid > 128) { rgrid = 128; nseed = 2; }
  if (!queue) nseed = 0;
  const int grid = queue ? rgrid + nseed : (V.B < e->reset_grid * 4 ? V.B : e->reset_grid * 4);
  if (e->n_classes > 0) {
    bool first = true;
    for (int b = 0; b < GE_MAX_BUCKETS && rc == GE_OK; b++) {  // one launch per LDS bucket; the seeding workgroups ride in the first
      const GeBucket &K = e->bk[b];
      if (!K.used) continue;
      int bg = K.reset_grid; if (small && bg > 128) bg = 128;
      const int ns = first ? nseed : 0;
      const int g = queue ? bg + ns : (V.B < K.reset_grid * 4 ? V.B : K.reset_grid * 4);
      int lds = K.reset_lds;
      GeParams V2 = V;
      if (V.lds.pre != 0) { V2.lds.pre = ge_align16(K.reset_lds); lds = V2.lds.pre + ((V.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK + 2) * 4; }  // queue prefix behind the bucket's scratch
      GE_FOR_RAGGED_ENV(V.env_type, GE_LAUNCH((ge_k_reset<ENV, true>), g, GE_RESET_THREADS, lds, stream, V2, VR, seeds, run, inj, ns, b));
      rc = check_launch("reset kernel");
      first = false;
    }
  } else if (run.inject && V.nocolw) {  // the injected rows come in the caller's order: this launch keeps the {neighbour, code} list (the full LDS carve)
    GeParams Vi = V; Vi.nocolw = 0; Vi.nowsort = 0; ge_make_lds(Vi, V.B);
    GE_FOR_ENV(V.env_type, GE_LAUNCH((ge_k_reset<ENV, false>), grid, GE_RESET_THREADS, e->lds_bytes_inject, stream, Vi, VR, seeds, run, inj, nseed, -1));
  } else GE_FOR_ENV(V.env_type, GE_LAUNCH((ge_k_reset<ENV, false>), grid, GE_RESET_THREADS, e->lds_bytes, stream, V, VR, seeds, run, inj, nseed, -1));
  if (rc == GE_OK) rc = check_launch("reset kernel");
  if (rc != GE_OK) return rc;
  if (!run.inject && e->n_classes == 0 && eval_slot_bytes(V)) rc = launch_seq_baseline(e, V, queue ? 1 : 0, stream);
  if (rc != GE_OK) return rc;
  if (!run.inject) rc = launch_features(e, V, VR, run, small, stream);
  if (rc == GE_OK && run.restart) e->seeded = true;
  if (rc == GE_OK && run.restart == 1 && e->P.buf.stream_state) e->streams = true;
  return rc;
}

// every image is empty; with seeded generator states refill them all right away (one pass at full occupancy)
static int refill_spares(ge_engine *e, void *stream) {
  const int nblk = (e->P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK;
  GE_LAUNCH(ge_k_refill_list, nblk, GE_STEP_BLOCK, 64, stream, e->P, e->refill_list, e->refill_count);
  int rc = check_launch("refill list kernel");
  if (rc != GE_OK) return rc;
  GeInject none = {nullptr, nullptr, nullptr, nullptr, nullptr};
  e->pending_calls = 0;
  return launch_reset(e, e->PS, e->RS, nullptr, run_refill(), none, false, stream);
}
static int invalidate_spares(ge_engine *e, void *stream) {
  if (!e->spares) return GE_OK;
  if (hipMemsetAsync(e->P.spare_state, 0, (size_t)e->P.B, (hipStream_t)stream) != hipSuccess) return fail(GE_E_LAUNCH, "hipMemsetAsync failed");
  return (e->seeded && e->P.autoreset) ? refill_spares(e, stream) : GE_OK;
}

// a full reset / injection leaves the finished-slot queues empty: the regeneration queue and, with spares, the swap queue (next-step
// autoreset consumes it at the START of the next ge_step: a stale entry would copy an image over the slot that was just reset)
static int clear_queue(ge_engine *e, void *stream) {
  const size_t bytes = sizeof(int32_t) * (size_t)((e->P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK);
  if (hipMemsetAsync(e->P.buf.reset_count, 0, bytes, (hipStream_t)stream) != hipSuccess) return fail(GE_E_LAUNCH, "hipMemsetAsync failed");
  if (e->spares && hipMemsetAsync(e->P.swap_count, 0, bytes, (hipStream_t)stream) != hipSuccess) return fail(GE_E_LAUNCH, "hipMemsetAsync failed");
  return GE_OK;
}

extern "C" int ge_reset(ge_engine *e, const uint32_t *seeds, void *stream) {
  if (!e || !seeds) return fail(GE_E_BADARG, "null argument");
  GeInject none = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int rc = clear_queue(e, stream);
  if (rc != GE_OK) return rc;
  rc = launch_reset(e, e->P, e->R, seeds, run_full(), none, false, stream);
  if (rc == GE_OK) e->loaded = true;
  if (rc == GE_OK) rc = invalidate_spares(e, stream);  // ... and refill: every slot's episode 1 waits in its image
  return rc;
}

extern "C" int ge_reset_continue(ge_engine *e, void *stream) {
  if (!e) return fail(GE_E_BADARG, "null argument");
  if (e->n_classes > 0) return fail(GE_E_UNSUPPORTED, "ge_reset_continue is not built for the multi-class engine");
  if (e->spares) return fail(GE_E_UNSUPPORTED, "ge_reset_continue on an engine with spares: an image generated ahead of time would leave the streams of the wrong episode behind");
  if (!e->P.buf.stream_state) return fail(GE_E_STATE, "ge_reset_continue needs ge_buffers.stream_state (the streams every reset leaves behind)");
  if (!e->loaded || !e->seeded || !e->streams) return fail(GE_E_STATE, "ge_reset_continue before ge_reset: there is no stream to continue");
  GeInject none = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int rc = clear_queue(e, stream);
  if (rc != GE_OK) return rc;
  return launch_reset(e, e->P, e->R, nullptr, run_continue(), none, false, stream);
}

extern "C" int ge_inject_state(ge_engine *e, const int64_t *links, const uint8_t *wcode, const float *x,
                               const int32_t *terminals, const uint32_t *seeds, void *stream) {
  if (!e || !links || !wcode || !x) return fail(GE_E_BADARG, "null argument");
  if (e->n_classes > 0) return fail(GE_E_UNSUPPORTED, "ge_inject_state is not built for the multi-class engine (inject into uniform engines)");
  const int t = e->P.env_type;
  if ((t == GE_SHORTEST_PATH || t == GE_LONGEST_PATH || t == GE_STEINER_TREE || t == GE_MULTICAST_ROUTING || t == GE_DISTRIBUTION_CENTER ||
       t == GE_PERISHABLE_DELIVERY) && !terminals)
    return fail(GE_E_BADARG, "terminals required (source / destinations, targets, or pickups then drop-offs)");
  if (!seeds && e->P.autoreset && !e->seeded)
    return fail(GE_E_STATE, "ge_inject_state without seeds on an engine with autoreset whose generator states were never seeded: pass seeds, or call ge_reset first");
  GeInject inj = {links, wcode, x, terminals, seeds};
  int rc = clear_queue(e, stream);  // slots queued before the injection must not be regenerated over the injected state
  if (rc != GE_OK) return rc;
  rc = launch_reset(e, e->P, e->R, nullptr, run_inject(seeds != nullptr), inj, false, stream);
  if (rc == GE_OK) e->loaded = true;
  if (rc == GE_OK) rc = invalidate_spares(e, stream);  // the images belong to the episodes that follow the injected ones
  return rc;
}

static size_t step_lds(const ge_engine *e) { return (size_t)GE_STEP_BLOCK * e->P.W * 8 + GE_STEP_BLOCK + 64; }

extern "C" int ge_sample_actions(ge_engine *e, uint64_t policy_seed, int64_t *actions, void *stream);

// parenting >= 2 of LongestPath / TSP: the instantiation of the step kernel that carries the residual-graph walks
static bool prunes(const ge_engine *e) { return (e->P.env_type == GE_LONGEST_PATH || e->P.env_type == GE_TSP) && e->P.parenting >= 2; }
#define GE_LAUNCH_STEP(SAMPLE, actions_arg, seed_arg)                                                                                   \
  do {                                                                                                                                  \
    if (e->n_classes > 0) GE_FOR_RAGGED_ENV(e->P.env_type, GE_LAUNCH((ge_k_step<ENV, SAMPLE, true, 0>), grid, GE_STEP_BLOCK, step_lds(e), stream, e->P, e->R, actions_arg, seed_arg)); \
    else if (path64(e) && e->spares) GE_LAUNCH((ge_k_step_path64<SAMPLE, true>), grid, GE_STEP_BLOCK, step_lds(e), stream, e->P, actions_arg, seed_arg);  \
    else if (path64(e)) GE_LAUNCH((ge_k_step_path64<SAMPLE, false>), grid, GE_STEP_BLOCK, step_lds(e), stream, e->P, actions_arg, seed_arg);  \
    else if (edge_quad(e) && e->P.env_type == GE_STEINER_TREE) GE_LAUNCH((ge_k_step_edge<GE_STEINER_TREE, SAMPLE>), grid, GE_EDGE_THREADS, ge_edge_lds_bytes(e->P.AW, e->P.W), stream, e->P, actions_arg, seed_arg); \
    else if (edge_quad(e)) GE_LAUNCH((ge_k_step_edge<GE_MULTICAST_ROUTING, SAMPLE>), grid, GE_EDGE_THREADS, ge_edge_lds_bytes(e->P.AW, e->P.W), stream, e->P, actions_arg, seed_arg); \
    else if (prunes(e) && e->P.env_type == GE_TSP && e->P.W > GE_MAXW) GE_LAUNCH((ge_k_step<GE_TSP, SAMPLE, false, 2>), grid, GE_STEP_BLOCK, step_lds(e), stream, e->P, e->R, actions_arg, seed_arg); \
    else if (prunes(e) && e->P.W > GE_MAXW) GE_LAUNCH((ge_k_step<GE_LONGEST_PATH, SAMPLE, false, 2>), grid, GE_STEP_BLOCK, step_lds(e), stream, e->P, e->R, actions_arg, seed_arg); \
    else if (prunes(e) && e->P.env_type == GE_TSP) GE_LAUNCH((ge_k_step<GE_TSP, SAMPLE, false, 1>), grid, GE_STEP_BLOCK, step_lds(e), stream, e->P, e->R, actions_arg, seed_arg); \
    else if (prunes(e)) GE_LAUNCH((ge_k_step<GE_LONGEST_PATH, SAMPLE, false, 1>), grid, GE_STEP_BLOCK, step_lds(e), stream, e->P, e->R, actions_arg, seed_arg); \
    else GE_FOR_ENV(e->P.env_type, GE_LAUNCH((ge_k_step<ENV, SAMPLE, false, 0>), grid, GE_STEP_BLOCK, step_lds(e), stream, e->P, e->R, actions_arg, seed_arg)); \
  } while (0)

static bool path64(const ge_engine *e);
static size_t step_lds(const ge_engine *e);
// SteinerTree / MulticastRouting whose mask rows fit the LDS stage of the quad-per-slot kernel (ge_step.h, ge_k_step_edge)
static bool edge_quad(const ge_engine *e) {
  static const bool off = getenv("GE_NO_EDGE_QUAD") != nullptr;  // diagnostic: the thread-per-slot kernel (before / after measurements)
  return !off && e->n_classes == 0 && (e->P.env_type == GE_STEINER_TREE || e->P.env_type == GE_MULTICAST_ROUTING) && ge_edge_fits(e->P.AW, e->P.W);
}

static bool path64(const ge_engine *e) {
  return e->n_classes == 0 && (e->P.env_type == GE_SHORTEST_PATH || e->P.env_type == GE_LONGEST_PATH) && e->P.W == 1 && e->P.parenting < 2;
}

// call-order guard (the reference raises in the same situations): stepping needs an episode in the slots, and autoreset needs a
// seeded generator ring -- an unseeded MT19937 state would draw the same node pair for ever
static int check_state(const ge_engine *e) {
  if (!e->loaded) return fail(GE_E_STATE, "the engine holds no episode yet: call ge_reset (or ge_inject_state) first");
  if (e->P.autoreset && !e->seeded) return fail(GE_E_STATE, "autoreset needs seeded generator states: call ge_reset, or ge_inject_state with seeds");
  return GE_OK;
}

extern "C" int ge_step_only(ge_engine *e, const int64_t *actions, void *stream) {
  if (!e || !actions) return fail(GE_E_BADARG, "null argument");
  int rc = check_state(e);
  if (rc != GE_OK) return rc;
  if (e->P.env_type == GE_DISTRIBUTION_CENTER && e->P.n <= 64) {  // the chosen centres' coverage ranges, computed when they are chosen
    GE_LAUNCH(ge_k_dc_range, (e->P.B + GE_WAVE - 1) / GE_WAVE, GE_WAVE, (size_t)e->P.n * GE_WAVE * 8 + GE_WAVE * 64, stream, e->P, actions);
    rc = check_launch("coverage range kernel");
    if (rc != GE_OK) return rc;
  }
  int grid = (e->P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK;
  GE_LAUNCH_STEP(false, actions, (uint64_t)0);
  return check_launch("step kernel");
}

// sample + step in one launch where the fused kernel exists (the drawn actions go to ge_buffers.actions_out when that is set),
// else two launches through `scratch`
static int sample_and_step(ge_engine *e, uint64_t policy_seed, int64_t *scratch, void *stream) {
  int rc = check_state(e);
  if (rc != GE_OK) return rc;
  if (e->P.env_type == GE_DISTRIBUTION_CENTER && e->P.n <= 64) {  // the coverage range kernel sits between the policy and the step
    if (!scratch) return fail(GE_E_BADARG, "this env type needs actions_scratch");
    rc = ge_sample_actions(e, policy_seed, scratch, stream);
    return rc == GE_OK ? ge_step_only(e, scratch, stream) : rc;
  }
  int grid = (e->P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK;
  GE_LAUNCH_STEP(true, (const int64_t *)nullptr, policy_seed);
  return check_launch("fused sample+step kernel");
}

// regenerate the slots the most recent step launch queued
extern "C" int ge_reset_pending(ge_engine *e, void *stream) {
  if (!e) return fail(GE_E_BADARG, "null argument");
  int rc = check_state(e);
  if (rc != GE_OK) return rc;
  if (e->P.autoreset) {
    GeInject none = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (e->spares) {  // finished slots with a valid image: one streaming copy each
      const int nblk = (e->P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK;
      int64_t grid = 1024;  // workgroups stride over (slot, part) items; most steps have a few hundred
      if (grid > (int64_t)e->P.B * e->swap_parts) grid = (int64_t)e->P.B * e->swap_parts;
      if (e->n_classes > 0) GE_LAUNCH(ge_k_swap<true>, (int)grid, 256, (nblk + 2) * 4, stream, e->P, e->R, e->RS, e->PS.buf, e->swap_parts);
      else GE_LAUNCH(ge_k_swap<false>, (int)grid, 256, (nblk + 2) * 4, stream, e->P, e->R, e->RS, e->PS.buf, e->swap_parts);
      rc = check_launch("swap kernel");
      if (rc != GE_OK) return rc;
    }
    rc = launch_reset(e, e->P, e->R, nullptr, run_queue(e), none, e->spares, stream);  // (with spares: the slots that finished again before their image was refilled)
    if (rc == GE_OK && e->spares && ++e->pending_calls >= e->period) rc = refill_spares(e, stream);
  }
  return rc;
}

extern "C" int ge_step(ge_engine *e, const int64_t *actions, void *stream) {
  if (!e || !actions) return fail(GE_E_BADARG, "null argument");
  if (e->P.autoreset == 2) {  // next-step mode: the slots that finished in the previous step are regenerated first
    int rc = ge_reset_pending(e, stream);
    return rc == GE_OK ? ge_step_only(e, actions, stream) : rc;
  }
  int rc = ge_step_only(e, actions, stream);
  if (rc != GE_OK) return rc;
  return ge_reset_pending(e, stream);
}

// Checkpointing: the slabs are the whole state.  An engine whose slabs were restored from a snapshot of a reset engine holds an
// episode and a seeded generator ring.
extern "C" int ge_mark_restored(ge_engine *e) {
  if (!e) return GE_E_BADARG;
  e->loaded = true; e->seeded = true; e->streams = e->P.buf.stream_state != nullptr;
  e->pending_calls = e->period;  // (the caller restores or clears spare_state with the other slabs; refill at the next opportunity)
  return GE_OK;
}

extern "C" int ge_vectorize(ge_engine *e, float *out, void *stream) {
  if (!e || !out) return fail(GE_E_BADARG, "null argument");
  if (e->n_classes > 0) {  // multi-class engine: the classes' flat vectors follow one another, class after class
    for (const GeParams &C : e->classes) {
      const int64_t Lc = (int64_t)C.n * C.F + (int64_t)C.E * C.Fe + 2 * (int64_t)C.E, tot = (int64_t)C.B * Lc;
      int64_t blocks = (tot + 255) / 256; if (blocks > 256 * 32) blocks = 256 * 32;
      GE_LAUNCH(ge_k_vectorize, (int)blocks, 256, 0, stream, C, out);
      int rc = check_launch("vectorize kernel");
      if (rc != GE_OK) return rc;
      out += tot;
    }
    return GE_OK;
  }
  int64_t L = (int64_t)e->P.n * e->P.F + (int64_t)e->P.E * e->P.Fe + 2 * (int64_t)e->P.E;
  int64_t total = (int64_t)e->P.B * L;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  GE_LAUNCH(ge_k_vectorize, (int)blocks, 256, 0, stream, e->P, out);
  return check_launch("vectorize kernel");
}

extern "C" int ge_sample_actions(ge_engine *e, uint64_t policy_seed, int64_t *actions, void *stream) {
  if (!e || !actions) return fail(GE_E_BADARG, "null argument");
  int grid = (e->P.B + 255) / 256;
  if (e->n_classes > 0) GE_LAUNCH(ge_k_sample<true>, grid, 256, 0, stream, e->P, e->R, policy_seed, actions);
  else GE_LAUNCH(ge_k_sample<false>, grid, 256, 0, stream, e->P, e->R, policy_seed, actions);
  return check_launch("sample kernel");
}

extern "C" int ge_random_rollout(ge_engine *e, uint64_t policy_seed, int32_t n_steps, int64_t *scratch, void *stream) {
  if (!e) return fail(GE_E_BADARG, "null argument");
  for (int s = 0; s < n_steps; s++) {
    int rc = GE_OK;
    if (e->P.autoreset == 2) rc = ge_reset_pending(e, stream);
    if (rc == GE_OK) rc = sample_and_step(e, policy_seed, scratch, stream);
    if (rc == GE_OK && e->P.autoreset != 2) rc = ge_reset_pending(e, stream);
    if (rc != GE_OK) return rc;
  }
  return GE_OK;
}

extern "C" int ge_timed_rollout(ge_engine *e, uint64_t policy_seed, int32_t n_steps, int64_t *scratch, void *stream,
                                double *step_ms, double *reset_ms, double *policy_ms) {
  if (!e || (!scratch && !path64(e))) return fail(GE_E_BADARG, "null argument");
  if (!e->have_events) { for (int k = 0; k < 4; k++) if (hipEventCreate(&e->ev[k]) != hipSuccess) return fail(GE_E_LAUNCH, "hipEventCreate failed"); e->have_events = true; }
  double ts = 0, tr = 0, tp = 0;
  hipStream_t st = (hipStream_t)stream;
  for (int s = 0; s < n_steps; s++) {
    int rc = GE_OK;
    if (e->P.autoreset == 2) rc = ge_reset_pending(e, stream);  // next-step mode: counted with nothing (the timed split is for same-step runs)
    (void)hipEventRecord(e->ev[0], st);
    if (rc == GE_OK && !path64(e)) rc = ge_sample_actions(e, policy_seed, scratch, stream);
    (void)hipEventRecord(e->ev[1], st);
    if (rc == GE_OK) rc = path64(e) ? sample_and_step(e, policy_seed, scratch, stream) : ge_step_only(e, scratch, stream);
    (void)hipEventRecord(e->ev[2], st);
    if (rc == GE_OK && e->P.autoreset != 2) rc = ge_reset_pending(e, stream);
    (void)hipEventRecord(e->ev[3], st);
    if (rc != GE_OK) return rc;
    if (hipEventSynchronize(e->ev[3]) != hipSuccess) return fail(GE_E_LAUNCH, "hipEventSynchronize failed");
    float a = 0, b = 0, c = 0;
    (void)hipEventElapsedTime(&a, e->ev[0], e->ev[1]); (void)hipEventElapsedTime(&b, e->ev[1], e->ev[2]); (void)hipEventElapsedTime(&c, e->ev[2], e->ev[3]);
    tp += a; ts += b; tr += c;
  }
  if (step_ms) *step_ms = ts;
  if (reset_ms) *reset_ms = tr;
  if (policy_ms) *policy_ms = tp;
  return GE_OK;
}

extern "C" int ge_timed_step_burst(ge_engine *e, uint64_t policy_seed, int32_t k, int64_t *scratch, void *stream, double *burst_ms) {
  if (!e || !burst_ms) return fail(GE_E_BADARG, "null argument");
  if (!e->have_events) { for (int j = 0; j < 4; j++) if (hipEventCreate(&e->ev[j]) != hipSuccess) return fail(GE_E_LAUNCH, "hipEventCreate failed"); e->have_events = true; }
  hipStream_t st = (hipStream_t)stream;
  (void)hipEventRecord(e->ev[0], st);
  for (int j = 0; j < k; j++) { int rc = sample_and_step(e, policy_seed, scratch, stream); if (rc != GE_OK) return rc; }
  (void)hipEventRecord(e->ev[1], st);
  if (hipEventSynchronize(e->ev[1]) != hipSuccess) return fail(GE_E_LAUNCH, "hipEventSynchronize failed");
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e->ev[0], e->ev[1]);
  *burst_ms = ms;
  return GE_OK;
}

// the launch floor of the step kernel's shape (bench.py: reported beside the kernel's own duration, never subtracted from it)
GE_KERNEL ge_k_empty(int) {}
extern "C" int ge_timed_empty_burst(ge_engine *e, int32_t k, void *stream, double *burst_ms) {
  if (!e || !burst_ms) return fail(GE_E_BADARG, "null argument");
  if (!e->have_events) { for (int j = 0; j < 4; j++) if (hipEventCreate(&e->ev[j]) != hipSuccess) return fail(GE_E_LAUNCH, "hipEventCreate failed"); e->have_events = true; }
  hipStream_t st = (hipStream_t)stream;
  const int grid = (e->P.B + GE_STEP_BLOCK - 1) / GE_STEP_BLOCK;
  const bool quad = edge_quad(e);
  (void)hipEventRecord(e->ev[0], st);
  for (int j = 0; j < k; j++) GE_LAUNCH(ge_k_empty, grid, quad ? GE_EDGE_THREADS : GE_STEP_BLOCK, quad ? ge_edge_lds_bytes(e->P.AW, e->P.W) : step_lds(e), stream, 0);
  (void)hipEventRecord(e->ev[1], st);
  if (hipEventSynchronize(e->ev[1]) != hipSuccess) return fail(GE_E_LAUNCH, "hipEventSynchronize failed");
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e->ev[0], e->ev[1]);
  *burst_ms = ms;
  return check_launch("empty kernel");
}

#if !defined(GE_EMU)
// diagnostic (tools/occupancy.py): resident workgroups per CU the runtime reports for the reset-path kernels
extern "C" int ge_debug_occupancy(ge_engine *e, int *out4) {
  if (!e || !out4) return GE_E_BADARG;
  int a = -1, b = -1, c = -1, d = -1;
  GE_FOR_ENV(e->P.env_type, (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, (ge_k_reset<ENV, false>), GE_RESET_THREADS, e->lds_bytes));
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, ge_k_features64<false>, GE_F64_THREADS, e->feat_lds);
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, ge_k_features<false>, GE_WAVE * e->P.ldsf.waves, e->P.ldsf.total);
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&d, (ge_k_step_path64<true, false>), GE_STEP_BLOCK, step_lds(e));
  out4[0] = a; out4[1] = b; out4[2] = c; out4[3] = d;
  return GE_OK;
}
#endif

#if defined(GE_STAMPS) && !defined(GE_EMU)
// diagnostic build only: copy out the phase timestamps of slot 0 (synchronises)
extern "C" int ge_debug_read_stamps(unsigned long long *out32) {
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(ge_stamp_buf), 32 * sizeof(unsigned long long)) == hipSuccess ? GE_OK : GE_E_LAUNCH;
}
#endif
