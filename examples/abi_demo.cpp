// A C++ host that drives libgraphenvs_hip.so through include/graphenvs.h alone: no Python, no torch.
// It allocates every buffer of ge_buffers with hipMalloc as ge_get_layout sizes them, resets B ShortestPath-v0 slots with
// seeds 0..B-1 and rolls K steps with the device policy, then prints counters a test compares with the Python host.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/abi_demo.cpp -Lgraphenvs_amd -lgraphenvs_hip -Wl,-rpath,$PWD/graphenvs_amd -o examples/abi_demo
//   examples/abi_demo [B] [K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "graphenvs.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
#define GE(x) do { int rc_ = (x); if (rc_ != GE_OK) { fprintf(stderr, "graphenvs error %d (%s) at line %d\n", rc_, ge_last_error(), __LINE__); exit(3); } } while (0)

template <class T> static T *dev(size_t count) { T *p = nullptr; CK(hipMalloc(&p, count * sizeof(T))); CK(hipMemset(p, 0, count * sizeof(T))); return p; }

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 100, n = 64, m = 192;
  ge_config cfg; memset(&cfg, 0, sizeof(cfg));
  cfg.env_type = GE_SHORTEST_PATH; cfg.num_envs = B; cfg.n_nodes = n; cfg.n_edges = m; cfg.weighted = 1; cfg.parenting = -1;
  cfg.autoreset = 1; cfg.n_choices = -1.0; cfg.env_index_base = 0; cfg.seed_stride = B;
  ge_layout L; GE(ge_get_layout(&cfg, &L));
  const size_t Nn = (size_t)B * n, Ne = (size_t)B * L.E, W = L.W, AW = (L.A + 63) / 64;
  ge_buffers b; memset(&b, 0, sizeof(b));
  b.x = dev<float>(Nn * L.F); b.edge_index = dev<int64_t>(2 * Ne); b.edge_attr = dev<float>(Ne * L.Fe);
  b.row_ptr = dev<int32_t>((size_t)B * (n + 1)); b.colw = dev<uint16_t>(Ne); b.scode = dev<uint8_t>(Ne);
  b.adj_bits = dev<uint64_t>(Nn * W); b.node_rec = dev<uint64_t>(Nn * 2); b.slot_rec = dev<uint64_t>((size_t)B * 2);
  b.terminals = dev<int32_t>((size_t)B * 2); b.node_bits = dev<uint64_t>(B * W); b.target_bits = dev<uint64_t>(B * W);
  b.counters = dev<int32_t>((size_t)B * 2); b.seed = dev<uint32_t>(B); b.episode = dev<int64_t>(B);
  b.heuristic = dev<double>(B); b.mt_state = dev<uint32_t>((size_t)B * GE_SEED_DEPTH * 2 * 624);
  b.mask = dev<uint8_t>((size_t)B * L.A); b.mask_bits = dev<uint64_t>(B * AW); b.reward = dev<double>(B); b.terminated = dev<uint8_t>(B);
  b.invalid = dev<uint8_t>(B); b.solved = dev<int8_t>(B); b.final_cost = dev<double>(B); b.final_heur = dev<double>(B);
  b.final_len = dev<int32_t>(B); b.reset_list = dev<int32_t>(B); b.reset_count = dev<int32_t>((B + 255) / 256);
  b.work_list = dev<int32_t>(B); b.work_count = dev<int32_t>(4);
  ge_engine *e = nullptr; GE(ge_create(&cfg, &b, &e));

  std::vector<uint32_t> seeds(B); for (int i = 0; i < B; i++) seeds[i] = (uint32_t)i;
  uint32_t *dseeds = dev<uint32_t>(B); CK(hipMemcpy(dseeds, seeds.data(), B * sizeof(uint32_t), hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  if (ge_random_rollout(e, 1, 1, nullptr, st) != GE_E_STATE) { fprintf(stderr, "stepping before ge_reset must be refused\n"); return 4; }
  GE(ge_reset(e, dseeds, st));
  CK(hipStreamSynchronize(st));
  CK(hipMemset(dseeds, 0xff, B * sizeof(uint32_t)));  // the seeds buffer belongs to the caller again once the stream has passed ge_reset
  GE(ge_random_rollout(e, /*policy_seed=*/1, K, /*actions_scratch: not needed by the fused ShortestPath kernel*/ nullptr, st));
  CK(hipStreamSynchronize(st));

  std::vector<int64_t> episode(B); std::vector<uint64_t> rec((size_t)B * 2); std::vector<float> x(Nn * L.F);
  CK(hipMemcpy(episode.data(), b.episode, B * sizeof(int64_t), hipMemcpyDeviceToHost));
  CK(hipMemcpy(rec.data(), b.slot_rec, rec.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  CK(hipMemcpy(x.data(), b.x, x.size() * sizeof(float), hipMemcpyDeviceToHost));
  long long episodes = 0, steps = 0; double csum = 0.0, xsum = 0.0;
  for (int i = 0; i < B; i++) {  // slot_rec: {cost as float64 bits, head | status << 16 | aux << 24 | tstep << 32}
    double c; memcpy(&c, &rec[2 * (size_t)i], 8);
    episodes += episode[i]; steps += (long long)(rec[2 * (size_t)i + 1] >> GE_REC_TSTEP_SHIFT); csum += c;
  }
  for (float v : x) xsum += (double)v;
  printf("{\"envs\": %d, \"steps\": %d, \"episodes\": %lld, \"transitions\": %lld, \"cost_sum\": %.17g, \"x_sum\": %.17g}\n", B, K, episodes, steps, csum, xsum);
  GE(ge_destroy(e));
  return 0;
}
