// Diagnostic microbenchmark (not part of the library): launch time of a kernel in which every thread reads NA
// separate 8-byte streams and writes NW separate 8-byte streams (the access shape of the step kernel's SoA slot
// state), as a function of thread count, stream count and how the arrays were allocated.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct Ptrs { unsigned long long *r[32]; unsigned long long *w[32]; };
template <int NA, int NW>
__global__ void k(Ptrs p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long acc = 0;
#pragma unroll
  for (int a = 0; a < NA; a++) acc += p.r[a][i];
#pragma unroll
  for (int a = 0; a < NW; a++) p.w[a][i] = acc + a;
}
template <int NA, int NW>
static void run(Ptrs p, int n, const char *tag) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 8; rep++) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int j = 0; j < 5; j++) hipLaunchKernelGGL((k<NA, NW>), dim3((n + 255) / 256), dim3(256), 0, 0, p, n);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  double us = best * 1e3 / 5, bytes = (double)n * 8 * (NA + NW);
  printf("%-10s n=%8d reads=%2d writes=%2d  %8.2f us/launch  %7.1f GB/s\n", tag, n, NA, NW, us, bytes / us * 1e-3);
}
int main(int argc, char **argv) {
  for (int mode = 0; mode < 2; mode++) {
    for (int n : {65536, 262144, 1048576}) {
      Ptrs p; std::vector<void *> owned;
      if (mode == 0) { for (int a = 0; a < 32; a++) { CK(hipMalloc(&p.r[a], (size_t)n * 8)); CK(hipMalloc(&p.w[a], (size_t)n * 8)); owned.push_back(p.r[a]); owned.push_back(p.w[a]); CK(hipMemset(p.r[a], 1, (size_t)n * 8)); } }
      else { char *slab; CK(hipMalloc(&slab, (size_t)n * 8 * 64)); owned.push_back(slab); CK(hipMemset(slab, 1, (size_t)n * 8 * 64)); for (int a = 0; a < 32; a++) { p.r[a] = (unsigned long long *)(slab + (size_t)a * n * 8); p.w[a] = (unsigned long long *)(slab + (size_t)(32 + a) * n * 8); } }
      const char *tag = mode ? "one-slab" : "separate";
      run<1, 1>(p, n, tag); run<4, 4>(p, n, tag); run<11, 6>(p, n, tag); run<12, 14>(p, n, tag); run<24, 24>(p, n, tag);
      for (void *q : owned) CK(hipFree(q));
    }
  }
  return 0;
}
