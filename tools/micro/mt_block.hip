// Diagnostic micro-benchmark (not shipped): cycles per MT19937 block of the register-resident generator, one wave alone on a CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I graphenvs_amd/csrc tools/micro/mt_block.hip -o gpurun_out/mt_block && gpurun_out/mt_block
#include "ge_reset.h"
#include <cstdio>
template <int MODE>
__global__ void k(uint32_t *out, unsigned long long *cyc, int blocks) {
  const int lane = threadIdx.x & 63;
  __shared__ uint32_t mt[GE_MT_N];
  for (int i = lane; i < GE_MT_N; i += 64) mt[i] = 1812433253u * (i + 1) ^ (i << 7);
  __syncthreads();
  uint32_t R[GE_MT_ROWS];
  ge_mt_to_regs(R, mt, lane);
  int tot = 0;
  const unsigned long long t0 = clock64();
  for (int b = 0; b < blocks; b++) {
    if (MODE == 2) { ge_mt_from_regs(mt, R, lane); ge_wave_sync(); ge_mt_twist(mt, lane); ge_mt_to_regs(R, mt, lane); }
    else ge_mt_twist_regs(R, lane);
    if (MODE >= 1) {
#pragma unroll
      for (int kk = 0; kk < GE_MT_ROWS; kk++) tot += ge_popc64(ge_ballot(GE_WAVE * kk + lane < GE_MT_N && (ge_temper(R[kk]) & 7u) < 7u));
    }
  }
  const unsigned long long t1 = clock64();
  if (lane == 0) { cyc[0] = t1 - t0; out[0] = tot; }
  out[1 + lane] = R[0] ^ R[9];
}
int main() {
  uint32_t *out; unsigned long long *cyc;
  hipMalloc(&out, 4 * 128); hipMalloc(&cyc, 8);
  const int blocks = 20000;
  for (int mode = 0; mode < 3; mode++) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(a);
      if (mode == 0) k<0><<<1, 64>>>(out, cyc, blocks); else if (mode == 1) k<1><<<1, 64>>>(out, cyc, blocks); else k<2><<<1, 64>>>(out, cyc, blocks);
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long c; uint32_t o[2]; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(o, out, 8, hipMemcpyDeviceToHost);
    printf("mode %d (%s): %.3f us per block, %.0f clock64 ticks per block (tot %u)\n", mode, mode == 0 ? "twist in registers" : mode == 1 ? "twist in registers + count" : "LDS twist + count", ms * 1e3 / blocks, (double)c / blocks, o[0]);
  }
  return 0;
}
