/*
 * graphenvs.h -- C ABI of the MI355X-native batched graph-env engine (libgraphenvs_hip.so).
 *
 * Drop-in boundary for the reference's hot path (teshnizi/GraphEnvs): the reference has no
 * FFI; its boundary is the Gymnasium protocol of six env classes.  Each entry point below
 * names the reference interface it replaces (paths under /root/reference/graph_envs/).
 *
 * Conventions
 *  - plain C types only; every buffer pointer is a DEVICE pointer owned by the caller
 *    (the Python host allocates them with torch); the library never allocates device memory;
 *  - every launch goes to the hipStream_t passed as `stream` (void*, NULL = default stream); the engine owns no stream
 *    of its own and no call synchronises the stream or the device.  A pointer the caller passes to a call (seeds,
 *    actions, ...) is consumed on `stream` only and may be reused once `stream` has passed the call;
 *  - int return code: GE_OK or a negative GE_E_* value; no exceptions cross the ABI;
 *  - gfx950 only.
 */
#ifndef GRAPHENVS_H
#define GRAPHENVS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GE_ABI_VERSION 5

/* env ids of graph_envs/__init__.py:9-56 that are on the hot path */
enum {
  GE_SHORTEST_PATH = 0,       /* shortest_path.py  ShortestPathEnv   */
  GE_LONGEST_PATH = 1,        /* longest_path.py   LongestPathEnv    */
  GE_STEINER_TREE = 2,        /* steiner_tree.py   SteinerTreeEnv    */
  GE_TSP = 3,                 /* tsp.py            TSPEnv            */
  GE_DENSEST_SUBGRAPH = 4,    /* densest_subgraph.py DensestSubgraphEnv */
  GE_MAX_INDEPENDENT_SET = 5, /* max_independent_set.py MaxIndependentSet */
  GE_MULTICAST_ROUTING = 6,   /* multicast_routing.py MulticastRoutingEnv */
  GE_DISTRIBUTION_CENTER = 7, /* distribution_center.py DistributionCenterEnv */
  GE_PERISHABLE_DELIVERY = 8  /* perishable_product_delivery.py PerishableProductDeliveryEnv */
};

enum {
  GE_OK = 0,
  GE_E_BADARG = -1,      /* invalid config / null pointer */
  GE_E_UNSUPPORTED = -2, /* valid in the reference but not built yet */
  GE_E_LAUNCH = -3,      /* HIP launch/runtime error (hipGetLastError) */
  GE_E_TOOBIG = -4,      /* per-env graph does not fit the LDS-resident reset kernel */
  GE_E_STATE = -5        /* call order: the engine holds no episode yet (ge_reset / ge_inject_state first), or autoreset needs
                            generator states that were never seeded (ge_reset, or ge_inject_state with seeds) */
};

/* slot_rec[2*i + 1]: packed scalar state of slot i */
#define GE_REC_HEAD_MASK   0xffffull        /* bits 0-15: path head / TSP head / delivery vehicle; 0xffff = none */
#define GE_REC_STATUS_SHIFT 16              /* bits 16-23: 0 running, 1 finished (autoreset off), 2 needs reset,
                                               3 regenerated in this ge_step (next-step mode), 4 generation failed (see work_count[1]) */
#define GE_REC_AUX_SHIFT   24               /* bits 24-31: ShortestPath / LongestPath with n <= 64: the destination node */
#define GE_REC_TSTEP_SHIFT 32               /* bits 32-63: transitions executed by the slot since the last ge_reset (mod 2^32) */
#define GE_STREAM_WORDS 640                 /* one saved generator stream in ge_buffers.stream_state: 624 state words, the read position, padding */
#define GE_SEED_DEPTH 3                     /* ring of generator states per slot: episode e lives in entry e mod 3; while a slot runs episode e the
                                               entries of e+1 and e+2 are valid and the entry of e is being refilled with e+3 */

/* Constructor kwargs of the reference env classes, plus batch geometry.
 * shortest_path.py:23, longest_path.py:26, steiner_tree.py:26, tsp.py:22,
 * densest_subgraph.py:25, max_independent_set.py:25. */
typedef struct {
  int32_t env_type;
  int32_t num_envs;        /* B: env slots on this GPU */
  int32_t n_nodes;         /* uniform geometry */
  int32_t n_edges;         /* undirected edge count m (E = 2m directed) */
  int32_t weighted;
  int32_t parenting;
  int32_t n_dests;         /* SteinerTree, MulticastRouting; DistributionCenter: target_count; PerishableProductDelivery: n_products */
  int32_t spatial;         /* TSP: node coordinates rand()*10, Euclidean float64 edge weights (tsp.py:79-86) */
  int32_t is_eval_env;
  int32_t autoreset;       /* 0: finished slots freeze until ge_reset; 1: same-step autoreset (the step that ends an episode returns
                              the new episode's observation and mask beside the old episode's reward / final_*); 2: next-step
                              autoreset (gymnasium's default: that step returns the final observation, the NEXT ge_step
                              regenerates the slot, ignores its action and returns reward 0, terminated 0) */
  double n_choices;        /* DensestSubgraph; < 0 -> floor(n / e) as densest_subgraph.py:38-39 */
  int64_t env_index_base;  /* global index of slot 0 (multi-GPU shard of the batch dimension) */
  int64_t seed_stride;     /* episode k of a slot seeded s0 runs reset(seed=(s0 + k*seed_stride) mod 2^32) */
  int64_t node_id_base;    /* added to every node id written to edge_index: lets several engines of different
                              geometry share one PyG slab (ragged batch: variable-size CSR packing) */
  int64_t edge_row_stride; /* elements between the two rows of edge_index; 0 = num_envs * 2 * n_edges */
  double max_distance;     /* DistributionCenter: coverage radius (distribution_center.py:29, default 1) */
  double dt_min, dt_max;   /* PerishableProductDelivery: delivery-time window.  The reference computes it in its constructor
                              with numpy (perishable_product_delivery.py:53-61: avg_dist = np.log(n) / np.log(2m/n), times
                              0.65 when weighted; dt_min = 0.6 avg_dist, dt_max = 1.4 avg_dist); the caller passes the values */
} ge_config;

/* Sizes (in elements) of every caller-allocated device buffer for a config. */
typedef struct {
  int32_t F, Fe, A, W, E;      /* node feats, edge feats, mask length per env, u64 words per node set, directed edges */
  int64_t total_nodes, total_edges;
  int64_t obs_len;             /* flat obs length per env: n*F + E*Fe + 2E (utils.py:87-88) */
  int64_t reset_lds_bytes;     /* dynamic LDS one reset workgroup needs */
  int32_t feat_parts;          /* workgroups that share one slot in the n > 64 feature kernel (sizes feat_scratch) */
  int64_t eval_scratch_bytes;  /* bytes of ge_buffers.eval_scratch: work space of the is_eval_env baselines that run as sequential
                                  programs (TSP: closure matrix + matching tables per slot; unweighted MaxIndependentSet: graph
                                  copies + set tables of clique removal); else 0 */
  int64_t prune_scratch_words; /* uint64 words of ge_buffers.prune_scratch (num_envs x 4 x W for parenting >= 2 on more than 512 nodes); else 0 */
} ge_layout;

/* Device buffers.  B = num_envs, Nn = B*n, Ne = B*E, W = ceil(n/64).
 * Observation slabs are kept in PyG layout (utils.py:26-29 to_pyg_graph): */
typedef struct {
  /* --- observation (PyG Batch layout) */
  float *x;             /* [Nn, F]   GraphInstance.nodes of every env, row = global node id */
  int64_t *edge_index;  /* [2, Ne]   edge_links transposed, node ids offset by slot*n       */
  float *edge_attr;     /* [Ne, Fe]  GraphInstance.edges                                    */
  /* --- graph slab (CSR, insertion-order columns) */
  int32_t *row_ptr;     /* [B, n+1]  local offsets into the slot's edge segment             */
  uint16_t *colw;       /* [Ne]      (col << 4) | weight code k, weight = k/10.0 (k=10: 1.0) */
  uint8_t *scode;       /* [Ne]      weight codes in ascending-neighbour order per row: the code of u->v sits at
                                     row_ptr[u] + popcount(adj_bits[u] & below(v)) -- a lookup without a row scan */
  double *sw64;         /* [Ne]      spatial TSP only: float64 edge weights in ascending-neighbour order (else NULL) */
  uint64_t *adj_bits;   /* [Nn, W]   adjacency bit rows                                     */
  uint64_t *node_rec;   /* [Nn, 2]   n <= 64 only: {bit row, weight codes of the 16 smallest neighbours as nibbles}:
                                     everything a step needs about a node in one 16-byte gather (else NULL).  The graphs are
                                     undirected (weight(u,v) == weight(v,u)): the n <= 64 path kernel reads the weight of the
                                     move head -> a from a's record, so the head's record is never fetched */
  int32_t *rev_edge;    /* [Ne]      SteinerTree, MulticastRouting parenting 2: local index of the reverse directed edge (else NULL) */
  /* --- per-slot dynamic state */
  uint64_t *slot_rec;   /* [B, 2] {solution_cost accumulator as float64 bits (f32 semantics for Steiner/MIS), packed word GE_REC_*}:
                                 the scalar state of a slot in ONE coalesced 16-byte record */
  int32_t *terminals;   /* [B, T] T = max(2, n_dests+1): src, dest(s)                        */
  uint64_t *node_bits;  /* [B, W] visited / in-tree / taken set                              */
  uint64_t *target_bits;/* [B, W] SteinerTree: targets                                       */
  int32_t *counters;    /* [B, 2] taken count, edge count (Densest) / steps in episode (unused by the n <= 64 path kernel:
                                 an episode's length is the size of its visited set)          */
  uint32_t *seed;       /* [B]  seed of the slot's current episode (advanced by the feature kernel, the last kernel of a regeneration) */
  int64_t *episode;     /* [B]  episode index k of the slot (likewise) */
  double *heuristic;    /* [B]  heuristic_solution of the current episode (is_eval_env)      */
  uint32_t *mt_state;   /* [B, GE_SEED_DEPTH, 2, 624] MT19937 states (python stream, numpy stream) of the slot's coming episodes, episode e
                                 in ring entry e mod GE_SEED_DEPTH: seeded one lane per slot, two episodes before their use, by
                                 seeding workgroups that ride in the reset launch of an earlier regeneration of the slot */
  uint64_t *aux_bits;   /* [B]  DistributionCenter with n <= 64: which rows of range_bits exist (else NULL) */
  /* --- outputs of the last step / reset */
  uint8_t *mask;        /* [B, A] info['mask'] as bool bytes                                 */
  uint64_t *mask_bits;  /* [B, ceil(A/64)] same, packed                                      */
  double *reward;       /* [B]                                                               */
  uint8_t *terminated;  /* [B]                                                               */
  uint8_t *invalid;     /* [B]  1 where the reference would have raised AssertionError       */
  int8_t *solved;       /* [B]  info['solved']: -1 absent, 0/1                               */
  double *final_cost;   /* [B]  info['solution_cost'] where terminated                       */
  double *final_heur;   /* [B]  info['heuristic_solution'] where terminated (same-step autoreset: copied by the reset kernel
                                 before it overwrites `heuristic`)                             */
  int32_t *final_len;   /* [B]  episode length where terminated                              */
  /* --- reset work queue */
  int32_t *reset_list;  /* [B]  step workgroup g (256 slots) lists its finished slots at [256g, 256g+count) */
  int32_t *reset_count; /* [ceil(B/256)] finished slots per step workgroup, rewritten by every step    */
  int32_t *work_list;   /* [B]  slots the n<=64 feature fast path hands to the generic feature kernel  */
  int32_t *work_count;  /* [4]  [0] = entries in work_list; [1] = device error flags (bit 0: a G(n,m) rejection loop hit its
                                 round cap -- unseeded or corrupted generator state; the slot's status is 4) */
  double *feat_scratch; /* [B, parts, n] n > 64 only: betweenness partial sums when several workgroups share a slot
                                 (parts = ge_layout.feat_parts; NULL when parts == 1) */
  int32_t *node_aux;    /* [B, n] MulticastRouting parenting >= 3: the one selectable edge into each node outside the tree
                                 (argmin of distance-from-source, multicast_routing.py:164-186), -1 = none; else NULL */
  uint64_t *range_bits; /* [B, n, W] DistributionCenter: nodes within max_distance of each node, from that node as the
                                 Dijkstra source (distribution_center.py:25-26); else NULL */
  uint64_t *cover_bits; /* [B, W] DistributionCenter: covered nodes; else NULL */
  int64_t *actions_out; /* [B]  optional (may be NULL): where the fused policy+step launches of ge_random_rollout / ge_timed_*
                                 record the actions they drew; NULL = not recorded (8 bytes per slot and step less to write) */
  uint32_t *stream_state; /* [B, 2, GE_STREAM_WORDS] optional (may be NULL): the python and the numpy MT19937 stream of every slot as its
                                 last regeneration LEFT them (624 state words, then the read position), written by every reset
                                 when present -- what ge_reset_continue (reset(seed=None), shortest_path.py:49-52) resumes from */
  uint8_t *eval_scratch;  /* [ge_layout.eval_scratch_bytes] is_eval_env work space: TSP's Christofides baseline (tsp.py:114-117),
                                 MaxIndependentSet's clique removal (max_independent_set.py:63-67); else NULL */
  uint64_t *prune_scratch;/* [B, 4, W] LongestPath / TSP with parenting >= 2 on graphs above 512 nodes: node sets of the residual-graph
                                 walks of step() (longest_path.py:134-143, tsp.py:181-194), which smaller graphs keep in registers; else NULL */
} ge_buffers;

typedef struct ge_engine ge_engine;

int ge_abi_version(void);

/* hash of the HIP sources + this header the binary was compiled from (-DGE_SOURCE_HASH, set by graphenvs_amd/_lib.py build()):
 * the host compares it with the sources on disk and refuses (or rebuilds) a stale binary */
const char *ge_source_hash(void);

/* sizes for a config; replaces the observation_space/action_space arithmetic of the
 * constructors (shortest_path.py:40-42, steiner_tree.py:43-45, tsp.py:41-42). */
int ge_get_layout(const ge_config *cfg, ge_layout *out);

/* gym.make(id, **kwargs) for a batch (graph_envs/__init__.py:9-56 + the constructors). */
int ge_create(const ge_config *cfg, const ge_buffers *bufs, ge_engine **out);
int ge_destroy(ge_engine *e);

/* Multi-class ("ragged") engine -- BASELINE config 5: env instances of ONE id but different (n_nodes, n_edges) stepped by one launch
 * sequence.  Every reference instance has a fixed geometry (shortest_path.py:23-45, densest_subgraph.py:25-50,
 * max_independent_set.py:25-38), so a ragged batch is a list of size classes; class c is described exactly like a uniform engine
 * (cfgs[c], bufs[c]) and owns slots [start_c, start_c + num_envs_c) of the global slot order, start_c = sum of the earlier classes:
 *  - cfgs[c].env_index_base = cfgs[0].env_index_base + start_c; env_type, autoreset, seed_stride equal in all classes;
 *  - seed, episode, mt_state are ENGINE-wide arrays in slot order (bufs[c].seed = bufs[0].seed + start_c, ...); reset_list,
 *    reset_count, work_list, work_count (and actions_out) are taken from bufs[0] and sized for all slots; callers normally make
 *    every per-slot array engine-wide the same way and let the classes share the observation slabs through node_id_base /
 *    edge_row_stride (variable-size CSR packing: the policy sees ONE PyG Batch);
 *  - class_table (ge_ragged_table_bytes(n_classes) bytes), slot_class [all slots] int32 and class_start [n_classes + 1] int32 are
 *    device buffers the engine fills once, here (a synchronous copy).
 * ge_reset / ge_step / ge_sample_actions / ge_random_rollout / ge_vectorize then take arrays over ALL slots (ge_vectorize: the classes'
 * flat vectors one class after the other).  Every kernel maps a slot to its class and runs the class's code path; classes with
 * n_nodes <= 64 use the fast feature kernel, the others the generic one, inside the same launch sequence.  Built for ShortestPath,
 * DensestSubgraph and MaxIndependentSet; ge_inject_state is not available. */
int64_t ge_ragged_table_bytes(int32_t n_classes);
int ge_create_ragged(const ge_config *cfgs, const ge_buffers *bufs, int32_t n_classes, void *class_table,
                     int32_t *slot_class, int32_t *class_start, ge_engine **out);

/* Episode prefetch ("spares").  reset() of the reference (shortest_path.py:47-98 and siblings) is a pure function of its seed, and the
 * k-th autoreset episode of a slot is seeded s0 + k * seed_stride: it can be produced before the slot needs it.  With spares attached
 * every slot owns an IMAGE -- a second set of the per-slot slabs (observation, CSR, masks, scalar state) -- that the reset path fills
 * for many slots per launch, every `period` calls of ge_step; a slot that finishes with a valid image gets it by one streaming copy
 * inside the same ge_step (same-step autoreset) / the next one (next-step autoreset) instead of a regeneration whose latency the whole
 * step would wait for; a slot that finishes again before its image was refilled is regenerated in place as without spares.  Every
 * output of every call is the same with and without spares, whatever the period.
 *  - image: like the ge_buffers of ge_create, but only the per-slot slabs are read: x, edge_index, edge_attr, row_ptr, colw, scode,
 *    adj_bits, slot_rec, terminals, node_bits, target_bits, counters, heuristic, mask, mask_bits and, where the engine has them,
 *    sw64, node_rec, rev_edge, aux_bits, node_aux, range_bits, cover_bits -- same sizes as the live ones.  (The image of a class of
 *    a multi-class engine packs its observation slabs exactly like the live ones: same node_id_base / edge_row_stride.)
 *  - state [B] uint8 (zeroed by the caller), swap_list / refill_list [B] int32, swap_count / refill_count [ceil(B/256)] int32.
 * Multi-class engine: sp is an array of n_classes entries (entry c: the image of class c; state / lists / period are taken from
 * entry 0 and cover ALL slots) and class_table_spare is a second device buffer of ge_ragged_table_bytes(n_classes) bytes; else
 * sp points at one entry and class_table_spare is NULL.  Not available together with ge_buffers.stream_state. */
typedef struct {
  ge_buffers image;
  uint8_t *state;
  int32_t *swap_list, *swap_count;
  int32_t *refill_list, *refill_count;
  int32_t period;
} ge_spares;
int ge_attach_spares(ge_engine *e, const ge_spares *sp, void *class_table_spare);

/* env.reset(seed=s) for every slot (shortest_path.py:47-98 and the five siblings):
 * seeds [B] uint32 on device = first-episode seed per slot (read on `stream` only); sets episode = 0.
 * Runs graph sampling, weights (a8), terminals (a9), structural features (a6),
 * baselines (a16), first mask; fills the observation slabs. */
int ge_reset(ge_engine *e, const uint32_t *seeds, void *stream);

/* env.reset() WITHOUT a seed for every slot (shortest_path.py:49-52: the process-global `random` / `np.random` streams are
 * not re-seeded, the new graph is drawn from where the previous reset left them).  Needs ge_buffers.stream_state and an
 * earlier ge_reset on this engine (GE_E_STATE otherwise).  Every slot moves to its next episode index; seed[] advances by
 * seed_stride so that the autoreset episodes that follow are the ones a fresh ge_reset would have been followed by. */
int ge_reset_continue(ge_engine *e, void *stream);

/* env.step(a) for every slot (shortest_path.py:111-141 and siblings).  actions [B] int64 on
 * device.  Writes reward/terminated/invalid/solved/final_* and the next mask; with autoreset
 * it then regenerates finished slots with the seed of their next episode. */
int ge_step(ge_engine *e, const int64_t *actions, void *stream);

/* the two halves of ge_step, exposed for profiling and for callers that overlap them */
int ge_step_only(ge_engine *e, const int64_t *actions, void *stream);
int ge_reset_pending(ge_engine *e, void *stream);

/* Parity path: load a post-reset state produced elsewhere instead of sampling it
 * (SURVEY 7 "injecting the oracle's post-reset state").  links [B,E,2] int64 local node ids in
 * edge_links order, wcode [B,E] uint8 weight codes (symmetric: the code of u->v equals the code of v->u),
 * x [B,n,F] float32, terminals [B,T] int32.  seeds [B] uint32 or NULL: the seed each injected episode is taken to have
 * -- sets seed[] and episode = 0 and pre-seeds the generator states of the following episodes, so that autoreset continues
 * with reset(seed + seed_stride), ...; with NULL the engine must have been seeded by an earlier ge_reset (or run with
 * autoreset off), and the episodes that follow continue that earlier sequence. */
int ge_inject_state(ge_engine *e, const int64_t *links, const uint8_t *wcode, const float *x,
                    const int32_t *terminals, const uint32_t *seeds, void *stream);

/* Checkpointing: the slabs are the whole state of an engine.  After restoring every slab from a snapshot of an engine that
 * had been reset, tell the engine so (it then holds an episode and a seeded generator ring; the call-order guard of
 * ge_step needs to know). */
int ge_mark_restored(ge_engine *e);

/* utils.vectorize_graph for the whole batch (utils.py:87-88): out [B, obs_len] float32. */
int ge_vectorize(ge_engine *e, float *out, void *stream);

/* Uniform random valid action per slot from mask_bits (bench/test policy; the same function is
 * restated in oracle/ge_oracle.c oge_policy_pick).  actions [B] int64; -1 where the mask is empty. */
int ge_sample_actions(ge_engine *e, uint64_t policy_seed, int64_t *actions, void *stream);

/* n_steps x (sample, step[, autoreset]) without host involvement between launches.  actions_scratch [B] int64 is
 * only needed by env types without a fused policy+step kernel (it may be NULL otherwise). */
int ge_random_rollout(ge_engine *e, uint64_t policy_seed, int32_t n_steps, int64_t *actions_scratch,
                      void *stream);

/* Same loop with hipEvents around every step-kernel launch; returns summed milliseconds of the
 * step kernel and of the reset kernel (synchronises the stream at the end; profiling only). */
int ge_timed_rollout(ge_engine *e, uint64_t policy_seed, int32_t n_steps, int64_t *actions_scratch,
                     void *stream, double *step_ms_sum, double *reset_ms_sum, double *policy_ms_sum);

/* k back-to-back launches of the (sample +) step kernel between ONE pair of hipEvents, without the autoreset
 * kernels in between (finished slots simply stop moving): the event overhead is amortised over k launches.
 * Returns the elapsed milliseconds of the k launches (synchronises; profiling only). */
int ge_timed_step_burst(ge_engine *e, uint64_t policy_seed, int32_t k, int64_t *actions_scratch, void *stream,
                        double *burst_ms);

/* the launch floor of the step kernel: k back-to-back launches of an EMPTY kernel with the step kernel's grid, block and dynamic
 * LDS between one pair of hipEvents (synchronises; profiling only: bench.py reports it beside the step kernel's own duration). */
int ge_timed_empty_burst(ge_engine *e, int32_t k, void *stream, double *burst_ms);

const char *ge_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
