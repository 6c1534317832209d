// Thin names for the gfx950 wave64 primitives the kernels use.  The only alternative
// definition of these names is the CPU sanitizer harness in tests/emu/hip_emu.h (-DGE_EMU),
// which exists because GPU sanitizers are not available; the shipped library is always built
// from the HIP definitions below.
#pragma once
#ifdef GE_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GE_DEV static __device__ __forceinline__
#define GE_DEVFN __device__
#define GE_KERNEL __global__ void
#define GE_KERNEL_LB(threads, waves_per_simd) __global__ void __launch_bounds__(threads, waves_per_simd)
#define GE_HOSTDEV __host__ __device__ inline
#define GE_CONSTANT static __constant__ const

GE_DEV int ge_tid() { return (int)threadIdx.x; }
// The same value behind an opaque move.  Read at the top of the per-slot body of a persistent (slot-loop) kernel, it keeps the
// compiler from hoisting everything it derives from the thread id -- lane masks, shuffle sources, LDS addresses -- out of the slot
// loop, where those values stay live across the whole body and are spilled to scratch (ge_k_reset: 78 spilled VGPRs without it).
GE_DEV int ge_tid_fresh() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; }
GE_DEV int ge_bid() { return (int)blockIdx.x; }
GE_DEV int ge_bdim() { return (int)blockDim.x; }
GE_DEV int ge_gdim() { return (int)gridDim.x; }
GE_DEV unsigned char *ge_dyn_smem() {
  extern __shared__ __align__(16) unsigned char ge_smem_raw[];
  return ge_smem_raw;
}
GE_DEV void ge_sync() { __syncthreads(); }
// issue priority of this wave among the waves of its SIMD (s_setprio, 0 = default .. 3)
GE_DEV void ge_wave_priority(int p) { if (p) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
// LDS hand-off between lanes of ONE wave (the other waves of the workgroup do not take part)
GE_DEV void ge_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }
// Ordering point between the four lanes of a quad that execute identical control flow: LDS operations of one wave
// are performed in issue order, so only the compiler has to be kept from moving accesses across this point.
GE_DEV void ge_quad_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
// value of the lane (lane ^ 1) / (lane ^ 2) inside the quad: one DPP move (quad_perm), no LDS
GE_DEV uint32_t ge_quad_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true); }
GE_DEV uint32_t ge_quad_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true); }
// the 16-bit values of the four lanes of a quad as one 64-bit word (lane k of the quad in bits [16k, 16k+16)): four quad_perm broadcasts
GE_DEV uint64_t ge_quad_gather16(uint32_t v) {
  const uint32_t b0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xf, 0xf, true), b1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x55, 0xf, 0xf, true);
  const uint32_t b2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xAA, 0xf, 0xf, true), b3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xFF, 0xf, 0xf, true);
  return (uint64_t)((b0 & 0xffffu) | (b1 << 16)) | ((uint64_t)((b2 & 0xffffu) | (b3 << 16)) << 32);
}
// OR over the eight lanes of an octet (lanes 8k .. 8k+7 of the wave), the result in every one of them: two quad_perm exchanges and a
// row_half_mirror (lane i of the eight <-> lane 7 - i), three DPP moves, no LDS.  Every lane of the octet must execute it.
GE_DEV uint32_t ge_oct_or32(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);
  v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);
  v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true);
  return v;
}
// fire-and-forget LDS adds (ds_add_u32 / ds_add_f64, no return value, nothing to wait for)
GE_DEV void ge_lds_add_u32(uint32_t *p, uint32_t v) { atomicAdd(p, v); }
GE_DEV void ge_lds_add_f64(double *p, double v) { unsafeAtomicAdd(p, v); }
// every outstanding load has returned (s_waitcnt vmcnt(0)): placed where that is already true, in front of a block of
// stores, it keeps the compiler's conservative per-register waits at control-flow joins from landing between the
// stores, where they would wait for store acknowledgements
GE_DEV void ge_wait_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }
GE_DEV uint64_t ge_ballot(bool p) { return (uint64_t)__ballot(p ? 1 : 0); }
GE_DEV int ge_shfl_i32(int v, int src) { return __shfl(v, src, 64); }
GE_DEV uint32_t ge_shfl_u32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
GE_DEV uint64_t ge_shfl_u64(uint64_t v, int src) {
  int lo = __shfl((int)(uint32_t)v, src, 64), hi = __shfl((int)(uint32_t)(v >> 32), src, 64);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
GE_DEV double ge_shfl_f64(double v, int src) { return __longlong_as_double((long long)ge_shfl_u64((uint64_t)__double_as_longlong(v), src)); }
// value of lane `idx` where idx is wave-uniform: v_readlane_b32, no LDS
GE_DEV uint32_t ge_readlane_u32(uint32_t v, int idx) { return (uint32_t)__builtin_amdgcn_readlane((int)v, idx); }
// v with lane `idx` replaced by `val` (idx and val wave-uniform): a compare against the lane id and a select (v_writelane_b32
// would need the lane index in M0 beside the scalar value: one constant-bus operand only)
GE_DEV uint32_t ge_writelane_u32(uint32_t v, uint32_t val, int idx) { return ((int)(threadIdx.x & 63u) == idx) ? val : v; }
// value known to be the same in every lane: keep it (and what is computed from it) on the scalar unit
GE_DEV uint32_t ge_uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
GE_DEV double ge_u64_as_f64(uint64_t v) { return __longlong_as_double((long long)v); }
GE_DEV uint64_t ge_f64_as_u64(double v) { return (uint64_t)__double_as_longlong(v); }
GE_DEV int ge_popc64(uint64_t v) { return __popcll(v); }
// set bits of the (wave-uniform) mask below this lane: v_mbcnt_lo / v_mbcnt_hi, no lane mask to build
GE_DEV int ge_mbcnt(uint64_t m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
GE_DEV int ge_ctz64(uint64_t v) { return v ? (int)__builtin_ctzll(v) : 64; }
GE_DEV int ge_clz32(uint32_t v) { return v ? (int)__builtin_clz(v) : 32; }

#define GE_LAUNCH(kernel, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3((unsigned)(block)), (size_t)(smem), (hipStream_t)(stream), __VA_ARGS__)
#define GE_SET_MAX_DYN_LDS(kernel, bytes) \
  hipFuncSetAttribute((const void *)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))
#endif
