/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the graph-env hot path.
 *
 * A plain-C, single-env restatement of the reference's reset()/step() semantics
 * (teshnizi/GraphEnvs, graph_envs/*.py) and of the third-party arithmetic it calls
 * (CPython 3.10 `random`, numpy legacy RandomState, networkx 3.4.2, scipy 1.15.3).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (graphenvs_amd/, libgraphenvs_hip.so) never links, imports or calls it.
 *
 * Parity status: PINNED (one exception, the TSP baseline: an own Christofides tour, bound-checked; networkx's Kou tree and clique removal
 * are restated exactly, see ge_oracle.c "baselines") -- oracle/gen_golden.py runs the real reference in the build
 * container and tests/test_oracle_golden.py checks this file against those fixtures
 * (tests/golden/*.npz) and against the known answers of SURVEY.md section 10.
 */
#ifndef GE_ORACLE_H
#define GE_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { OGE_SHORTEST_PATH = 0, OGE_LONGEST_PATH = 1, OGE_STEINER_TREE = 2, OGE_TSP = 3,
       OGE_DENSEST_SUBGRAPH = 4, OGE_MAX_INDEPENDENT_SET = 5, OGE_MULTICAST_ROUTING = 6,
       OGE_DISTRIBUTION_CENTER = 7, OGE_PERISHABLE_DELIVERY = 8 };

typedef struct {
  int32_t env_type;
  int32_t n_nodes;
  int32_t n_edges;
  int32_t weighted;
  int32_t parenting;
  int32_t n_dests;     /* SteinerTree, MulticastRouting; DistributionCenter: target_count; PerishableProductDelivery: n_products */
  int32_t spatial;     /* TSP only */
  int32_t is_eval_env;
  double n_choices;    /* DensestSubgraph only; <0 -> n_nodes // e (reference default) */
  double max_distance; /* DistributionCenter only (distribution_center.py:29) */
  double dt_min, dt_max; /* PerishableProductDelivery: delivery-time window, computed by the constructor with numpy
                            (perishable_product_delivery.py:53-61) */
} oge_cfg;

typedef struct oge_env oge_env;

/* status codes returned by oge_step */
enum { OGE_OK = 0, OGE_INVALID_ACTION = 1 };

oge_env *oge_create(const oge_cfg *cfg);
void oge_destroy(oge_env *e);

/* seed >= 0: reseed both MT19937 streams like reset(seed=s); seed < 0: continue the streams
 * (reset(seed=None)). Returns the number of G(n,m) rejection attempts. */
int oge_reset(oge_env *e, int64_t seed);

/* one transition. reward is returned as double; Steiner/MIS rewards are float32 values
 * (exactly representable). solved: -1 = key absent from info, 0/1 otherwise. */
int oge_step(oge_env *e, int64_t action, double *reward, int32_t *done, int32_t *solved);

/* geometry */
int oge_num_node_features(const oge_env *e);  /* F  */
int oge_num_edge_features(const oge_env *e);  /* Fe */
int oge_num_directed_edges(const oge_env *e); /* E = 2m */
int oge_mask_size(const oge_env *e);          /* n, or 2m for SteinerTree / MulticastRouting */
int oge_obs_size(const oge_env *e);           /* n*F + E*Fe + 2E */

/* state readers (copy out) */
void oge_get_nodes(const oge_env *e, float *x);               /* [n,F]   */
void oge_get_edges(const oge_env *e, float *ef);              /* [E,Fe]  */
void oge_get_edge_links(const oge_env *e, int64_t *links);    /* [E,2]   */
void oge_get_obs(const oge_env *e, float *obs);               /* vectorize_graph */
void oge_get_mask(const oge_env *e, uint8_t *mask);           /* info['mask'] as of the last reset/step */
void oge_get_features64(const oge_env *e, double *sf);        /* [n,5] float64 features before the f32 cast */
double oge_solution_cost(const oge_env *e);
double oge_heuristic_solution(const oge_env *e);              /* info['heuristic_solution'] of the current episode */
int oge_head(const oge_env *e);
void oge_get_terminals(const oge_env *e, int32_t *out);       /* [0]=src, [1..] = dest(s); count = 1 + n_targets */
int oge_num_targets(const oge_env *e);

/* inject a post-reset state produced elsewhere (parity path for step-only tests).
 * links [E,2] int64 (row-major by source), weights f64 per directed edge, x [n,F] f32. */
int oge_inject(oge_env *e, const int64_t *links, const double *w64, const float *x,
               const int32_t *terminals, int n_terminals);

/* reproducible uniform choice among valid actions; the same function is implemented
 * independently in the HIP library (ge_sample_actions). Returns -1 if the mask is empty. */
int64_t oge_policy_pick(const uint8_t *mask, int n, uint64_t policy_seed, uint64_t env_index,
                        uint64_t t);

/* Batched random-policy rollout with autoreset, OpenMP over envs (cpu_baseline leg).
 * Env i starts with seed (first_seed + i) mod 2^32; episode k reseeds with
 * (first_seed + i + k*seed_stride) mod 2^32.  Runs n_steps transitions per env.
 * Returns total transitions; out_sum_reward/out_episodes optional accumulators (may be NULL). */
int64_t oge_rollout(const oge_cfg *cfg, int64_t first_seed, int64_t seed_stride, int32_t n_envs,
                    int32_t n_steps, uint64_t policy_seed, int32_t n_threads,
                    double *out_sum_reward, int64_t *out_episodes, double *out_reset_seconds);

/* test hooks for the own baselines: the independent set (n flags) / the Steiner tree (n*n symmetric flags)
 * recomputed on the graph of the last reset; both return the heuristic value */
double oge_debug_greedy_mis(oge_env *e, uint8_t *out_n);
double oge_debug_steiner_tree(oge_env *e, uint8_t *out_nn);
/* the TSP baseline's pieces: tour length (integer units) of the Christofides tour on a closure matrix D [n*n]; minimum weight
 * of a perfect matching of dist [k*k] (k even) with every vertex's partner in match_out [k] */
int64_t oge_debug_christofides(int32_t n, const int32_t *D);
int64_t oge_debug_min_matching(int32_t k, const int32_t *dist, int32_t *match_out);
/* the MIS baseline's pieces (graphenvs_amd/csrc/ge_clique_removal.h): iteration order of a CPython set of ints built by insertion;
 * of nx.non_neighbors(G, nodes[0]) for a graph whose node dict is `nodes` and whose first adjacency dict is `first_adj`; the value
 * len(nx.approximation.maximum_independent_set(G)) for G as insertion-order CSR (-1: work space too small) */
int32_t oge_debug_pyset_int_order(const int32_t *keys, int32_t count, int32_t *out);
int32_t oge_debug_non_neighbors(const int32_t *nodes, int32_t k, const int32_t *first_adj, int32_t deg, int32_t *out);
int32_t oge_debug_clique_removal(int32_t n, int32_t m, const int32_t *row_ptr, const int32_t *col);
/* the SteinerTree baseline (graphenvs_amd/csrc/ge_kou_exact.h) on a graph given as insertion-order CSR with float64 weights per
 * directed entry and the terminals in self.dests order; NaN: work space too small */
double oge_debug_kou_exact(int32_t n, int32_t m, int32_t T, const int32_t *off, const int32_t *col, const double *w, const int32_t *terms);

/* iteration order of a CPython 3.10 set after adding the int pairs (u[i], v[i]) in order (multicast baseline sums a set) */
int oge_pyset_order(const int32_t *u, const int32_t *v, int count, int32_t *out_u, int32_t *out_v);

/* raw MT19937 access, for pinning the two generators themselves */
void oge_mt_py_seed(uint32_t *state625, uint32_t seed);
void oge_mt_np_seed(uint32_t *state625, uint32_t seed);
uint32_t oge_mt_next(uint32_t *state625);

#ifdef __cplusplus
}
#endif
#endif
