// Diagnostic micro-benchmark (not shipped): what one SIMD of gfx950 issues per cycle for the instruction kinds of the feature
// kernels, with 1 / 2 / 4 / 8 waves resident on it.  One workgroup on one CU, 256 x k threads = k waves per SIMD; every wave runs the
// same loop of independent instructions (eight accumulators, no dependency shorter than eight instructions) and reads the shader
// clock (s_memtime) around it.  Printed: wave64 instructions per SIMD-cycle = (waves on the SIMD x instructions) / cycles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/valu_issue.hip -o gpurun_out/valu_issue && gpurun_out/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum { K_ADD32, K_AND32, K_ADD64, K_ADDF64, K_FMAF64, K_FFBL, K_BCNT, K_DPP, K_LDSRD64, K_LDSADD32, K_LDSADDF64, K_CNDMASK, K_MULLO, K_LSHL64, K_N };
static const char *kname[K_N] = {"v_add_u32", "v_and_b32", "64-bit add (v_add_co + v_addc_co)", "v_add_f64", "v_fma_f64", "v_ffbl_b32", "v_bcnt_u32_b32",
                                 "v_mov_b32 dpp quad_perm", "ds_read_b64 (conflict-free)", "ds_add_u32 (conflict-free)", "ds_add_f64 (conflict-free)",
                                 "v_cndmask_b32", "v_mul_lo_u32", "v_lshlrev_b64"};

template <int KIND>
__global__ void __launch_bounds__(1024) k(unsigned long long *cyc, uint32_t *sink, int iters) {
  __shared__ double lds[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  uint32_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3, a4 = tid + 4, a5 = tid + 5, a6 = tid + 6, a7 = tid + 7, b = tid | 1;
  uint64_t q0 = tid, q1 = tid + 1, q2 = tid + 2, q3 = tid + 3, q4 = 4, q5 = 5, q6 = 6, q7 = 7, qb = 0x100000001ull * (tid | 1);
  double d0 = tid, d1 = 1, d2 = 2, d3 = 3, d4 = 4, d5 = 5, d6 = 6, d7 = 7, db = 1.0000001;
  const uint32_t la = (uint32_t)((tid & 63) * 8 + (tid >> 6) * 512);  // byte address: a wave's 64 lanes on 64 consecutive 8-byte words
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (KIND == K_ADD32) asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      if (KIND == K_AND32) asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
      if (KIND == K_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      if (KIND == K_FFBL) asm volatile("v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3\n v_ffbl_b32 %4, %4\n v_ffbl_b32 %5, %5\n v_ffbl_b32 %6, %6\n v_ffbl_b32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      if (KIND == K_BCNT) asm volatile("v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      if (KIND == K_DPP) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      if (KIND == K_ADD64) { q0 += qb; q1 += qb; q2 += qb; q3 += qb; q4 += qb; q5 += qb; q6 += qb; q7 += qb; asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7)); }
      if (KIND == K_LSHL64) asm volatile("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3\n v_lshlrev_b64 %4, 1, %4\n v_lshlrev_b64 %5, 1, %5\n v_lshlrev_b64 %6, 1, %6\n v_lshlrev_b64 %7, 1, %7" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
      if (KIND == K_ADDF64) asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db));
      if (KIND == K_FMAF64) asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db));
      if (KIND == K_LDSRD64) asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8192\n ds_read_b64 %2, %8 offset:16384\n ds_read_b64 %3, %8 offset:24576\n ds_read_b64 %4, %8\n ds_read_b64 %5, %8 offset:8192\n ds_read_b64 %6, %8 offset:16384\n ds_read_b64 %7, %8 offset:24576\n s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(la) : "memory");
      if (KIND == K_LDSADD32) asm volatile("ds_add_u32 %0, %1\n ds_add_u32 %0, %1 offset:8192\n ds_add_u32 %0, %1 offset:16384\n ds_add_u32 %0, %1 offset:24576\n ds_add_u32 %0, %1 offset:4\n ds_add_u32 %0, %1 offset:8196\n ds_add_u32 %0, %1 offset:16388\n ds_add_u32 %0, %1 offset:24580" : : "v"(la), "v"(b) : "memory");
      if (KIND == K_LDSADDF64) asm volatile("ds_add_f64 %0, %1\n ds_add_f64 %0, %1 offset:8192\n ds_add_f64 %0, %1 offset:16384\n ds_add_f64 %0, %1 offset:24576\n ds_add_f64 %0, %1\n ds_add_f64 %0, %1 offset:8192\n ds_add_f64 %0, %1 offset:16384\n ds_add_f64 %0, %1 offset:24576" : : "v"(la), "v"(db) : "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = clock64();
  if ((tid & 63) == 0) cyc[tid >> 6] = t1 - t0;
  sink[tid] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^ (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) ^ (uint32_t)lds[tid];
}

template <int KIND>
static void run(unsigned long long *cyc, uint32_t *sink) {
  const int iters = 4000, per_iter = 32;
  printf("%-36s", kname[KIND]);
  for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD (one workgroup of 256 * wps threads: its waves go round-robin over the four SIMDs)
    k<KIND><<<1, 256 * wps>>>(cyc, sink, iters);
    k<KIND><<<1, 256 * wps>>>(cyc, sink, iters);
    hipDeviceSynchronize();
    unsigned long long c[16]; hipMemcpy(c, cyc, 8 * 4 * wps, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int w = 0; w < 4 * wps; w++) mx = c[w] > mx ? c[w] : mx;
    const double n = (double)iters * per_iter * ((KIND == K_ADD64) ? 2 : 1);
    printf("  %d w/SIMD: %5.2f cyc/inst/wave, %.3f inst/SIMD-cyc |", wps, (double)mx / n, wps * n / (double)mx);
  }
  printf("\n");
}

int main() {
  unsigned long long *cyc; uint32_t *sink;
  hipMalloc(&cyc, 8 * 64); hipMalloc(&sink, 4 * 1024);
  run<K_ADD32>(cyc, sink); run<K_AND32>(cyc, sink); run<K_CNDMASK>(cyc, sink); run<K_MULLO>(cyc, sink); run<K_ADD64>(cyc, sink); run<K_LSHL64>(cyc, sink);
  run<K_FFBL>(cyc, sink); run<K_BCNT>(cyc, sink); run<K_DPP>(cyc, sink); run<K_ADDF64>(cyc, sink); run<K_FMAF64>(cyc, sink);
  run<K_LDSRD64>(cyc, sink); run<K_LDSADD32>(cyc, sink); run<K_LDSADDF64>(cyc, sink);
  return 0;
}
