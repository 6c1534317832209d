// TEST INFRASTRUCTURE ONLY -- CPU sanitizer harness for the HIP kernels.
//
// Compiles graphenvs_amd/csrc/*.hip with g++ (-DGE_EMU) so the kernels can run under
// UBSan/ASan and bounds checks in this GPU-less container (GPU sanitizers are unavailable on
// the MI355X pool).  Every GPU thread is a ucontext fiber; __syncthreads / ballot / shuffle are
// rendezvous points, and a rendezvous that not every live lane reaches aborts with a message
// (divergent-barrier detector).  Lanes run one after another between rendezvous points, in
// ascending or (GE_EMU_REVERSE=1) descending order, so a missing barrier shows up as a result
// that depends on the order.  The product never loads the emu library: graphenvs_amd/_lib.py
// only opens libgraphenvs_hip.so and raises if it is missing.
#pragma once
#include <ucontext.h>
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#endif

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define GE_DEV static inline
#define GE_DEVFN inline
#define GE_KERNEL static void
#define GE_KERNEL_LB(threads, waves_per_simd) static void
#define GE_HOSTDEV inline
#define GE_CONSTANT static const

// ---- minimal HIP runtime surface used by ge_api
typedef void *hipStream_t;
typedef int hipError_t;
typedef struct ge_emu_event { double t; } *hipEvent_t;
#define hipSuccess 0
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
#define hipMemcpyHostToDevice 1
static inline hipError_t hipMemcpy(void *dst, const void *src, size_t n, int) { memcpy(dst, src, n); return hipSuccess; }
#define hipStreamNonBlocking 1
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new ge_emu_event{0}; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new ge_emu_event{0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

namespace ge_emu {

constexpr int kMaxThreads = 1024;
constexpr int kWave = 64;
#ifndef GE_F64_QL
#define GE_F64_QL 4
#endif
constexpr int kQuad = GE_F64_QL;  // lanes that run identical control flow in the feature walk
constexpr size_t kStack = 256 * 1024;

struct Fiber {
  ucontext_t ctx;
  char *stack = nullptr;
  bool done = true;
  int waiting = 0;  // 0 runnable, 1 block barrier, 2 wave collective, 3 quad rendezvous, 4 octet rendezvous
};

struct Block {
  int nthreads = 0, bid = 0, gdim = 0;
  int cur = 0;
  Fiber fib[kMaxThreads];
  ucontext_t sched;
  uint64_t slot[2][kMaxThreads];
  bool part[2][kMaxThreads];
  int wave_gen[kMaxThreads / kWave];
  unsigned char *smem = nullptr;
  size_t smem_bytes = 0;
  std::function<void()> body;
};

inline Block &blk() { static Block b; return b; }

// AddressSanitizer build (build_emu.py asan=True): every switch between the scheduler's stack and a lane's fiber stack is
// announced, and the dynamic LDS of a block is a heap allocation of exactly the requested size (its red zone is the guard).
#if defined(__SANITIZE_ADDRESS__)
#define GE_EMU_ASAN 1
struct AsanSched { const void *bottom = nullptr; size_t size = 0; };
inline AsanSched &asan_sched() { static AsanSched a; return a; }
inline void switch_to_fiber(Block &b, Fiber &f) {
  void *fake = nullptr;
  __sanitizer_start_switch_fiber(&fake, f.stack, kStack);
  swapcontext(&b.sched, &f.ctx);
  __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
}
inline void switch_to_sched(Block &b, bool last) {
  void *fake = nullptr;
  __sanitizer_start_switch_fiber(last ? nullptr : &fake, asan_sched().bottom, asan_sched().size);
  swapcontext(&b.fib[b.cur].ctx, &b.sched);
  __sanitizer_finish_switch_fiber(fake, &asan_sched().bottom, &asan_sched().size);
}
inline void fiber_entered() { __sanitizer_finish_switch_fiber(nullptr, &asan_sched().bottom, &asan_sched().size); }
#else
inline void switch_to_fiber(Block &b, Fiber &f) { swapcontext(&b.sched, &f.ctx); }
inline void switch_to_sched(Block &b, bool) { swapcontext(&b.fib[b.cur].ctx, &b.sched); }
inline void fiber_entered() {}
#endif

inline void yield_to_sched() { switch_to_sched(blk(), false); }

inline void trampoline() {
  fiber_entered();
  Block &b = blk();
  b.body();
  b.fib[b.cur].done = true;
  switch_to_sched(b, true);
}

inline void die(const char *msg) { fprintf(stderr, "[hip_emu] %s\n", msg); abort(); }

inline void run_block(int bid, int gdim, int nthreads, size_t smem_bytes) {
  Block &b = blk();
  static bool reverse = getenv("GE_EMU_REVERSE") && atoi(getenv("GE_EMU_REVERSE"));
  b.nthreads = nthreads; b.bid = bid; b.gdim = gdim;
#ifdef GE_EMU_ASAN
  static size_t shortfall = getenv("GE_EMU_LDS_SHORTFALL") ? (size_t)atoi(getenv("GE_EMU_LDS_SHORTFALL")) : 0;  // self-test of the detector: allocate less than asked
  if (shortfall && smem_bytes > shortfall) smem_bytes -= shortfall;
  free(b.smem); b.smem = nullptr; if (posix_memalign((void **)&b.smem, 64, smem_bytes ? smem_bytes : 1)) die("out of memory"); b.smem_bytes = smem_bytes;  // exact size: the red zone follows the last byte
#else
  if (b.smem_bytes < smem_bytes + 64) { free(b.smem); b.smem = (unsigned char *)aligned_alloc(64, ((smem_bytes + 64 + 63) / 64) * 64); b.smem_bytes = smem_bytes + 64; }
#endif
  memset(b.smem, 0xCD, b.smem_bytes);  // poison: LDS is uninitialised on the GPU
  memset(b.wave_gen, 0, sizeof(b.wave_gen));
  memset(b.part, 0, sizeof(b.part));
  for (int t = 0; t < nthreads; t++) {
    Fiber &f = b.fib[t];
    if (!f.stack) f.stack = (char *)malloc(kStack);
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStack; f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())trampoline, 0);
    f.done = false; f.waiting = 0;
  }
  for (;;) {
    bool progressed = false; int live = 0;
    for (int k = 0; k < nthreads; k++) {
      int t = reverse ? nthreads - 1 - k : k;
      Fiber &f = b.fib[t];
      if (f.done) continue;
      live++;
      if (f.waiting) continue;
      b.cur = t; switch_to_fiber(b, f); progressed = true;
    }
    if (!live) break;
    // release rendezvous groups whose every live member has arrived
    bool released = false;
    int nlive = 0, nbar = 0;
    for (int t = 0; t < nthreads; t++) if (!b.fib[t].done) { nlive++; if (b.fib[t].waiting == 1) nbar++; }
    if (nlive && nbar == nlive) { for (int t = 0; t < nthreads; t++) b.fib[t].waiting = 0; released = true; }
    for (int w = 0; w * kWave < nthreads; w++) {
      int wl = 0, ww = 0;
      for (int t = w * kWave; t < nthreads && t < (w + 1) * kWave; t++) if (!b.fib[t].done) { wl++; if (b.fib[t].waiting == 2) ww++; }
      if (wl && ww == wl) { for (int t = w * kWave; t < nthreads && t < (w + 1) * kWave; t++) if (b.fib[t].waiting == 2) b.fib[t].waiting = 0; b.wave_gen[w]++; released = true; }
    }
    for (int q0 = 0; q0 < nthreads; q0 += kQuad) {  // groups of consecutive lanes running identical control flow
      int ql = 0, qw = 0;
      for (int t = q0; t < nthreads && t < q0 + kQuad; t++) if (!b.fib[t].done) { ql++; if (b.fib[t].waiting == 3) qw++; }
      if (ql && qw == ql) { for (int t = q0; t < nthreads && t < q0 + kQuad; t++) if (b.fib[t].waiting == 3) b.fib[t].waiting = 0; released = true; }
    }
    for (int q0 = 0; q0 < nthreads; q0 += 8) {  // octets (the n <= 64 feature kernel: eight lanes per BFS source)
      int ql = 0, qw = 0;
      for (int t = q0; t < nthreads && t < q0 + 8; t++) if (!b.fib[t].done) { ql++; if (b.fib[t].waiting == 4) qw++; }
      if (ql && qw == ql) { for (int t = q0; t < nthreads && t < q0 + 8; t++) if (b.fib[t].waiting == 4) b.fib[t].waiting = 0; released = true; }
    }
    if (!progressed && !released) {
      if (getenv("GE_EMU_DEBUG_DEADLOCK"))  // which lanes wait for what (1 block barrier, 2 wave collective, 3 quad rendezvous)
        for (int t = 0; t < nthreads && t < 64; t++) if (!b.fib[t].done) fprintf(stderr, "[hip_emu] block %d lane %d waits for %d\n", bid, t, b.fib[t].waiting);
      die("deadlock: a barrier or wave collective was not reached by every live lane (divergent rendezvous)");
    }
  }
  // LDS overrun detector: the 64 bytes behind the requested dynamic LDS were poisoned above and must still be
  for (size_t k = smem_bytes; k < b.smem_bytes; k++)
    if (b.smem[k] != 0xCD) { fprintf(stderr, "[hip_emu] block %d wrote past its %zu bytes of dynamic LDS (offset %zu)\n", bid, smem_bytes, k); abort(); }
}

template <class Fn>
inline void launch(int grid, int block, size_t smem, Fn fn) {
  if (block > kMaxThreads) die("block too large");
  blk().body = fn;
  for (int b = 0; b < grid; b++) run_block(b, grid, block, smem);
}

inline void barrier() { Block &b = blk(); b.fib[b.cur].waiting = 1; yield_to_sched(); }

// wave collective: publish v, wait for the wave, return the parity of the finished generation
inline int wave_rendezvous(uint64_t v) {
  Block &b = blk(); int w = b.cur / kWave; int g = b.wave_gen[w] & 1;
  b.slot[g][b.cur] = v; b.part[g][b.cur] = true;
  b.fib[b.cur].waiting = 2; yield_to_sched();
  return g;
}
inline void wave_leave(int g) { (void)g; }

}  // namespace ge_emu

GE_DEV int ge_tid() { return ge_emu::blk().cur; }
GE_DEV int ge_tid_fresh() { return ge_emu::blk().cur; }
GE_DEV int ge_bid() { return ge_emu::blk().bid; }
GE_DEV int ge_bdim() { return ge_emu::blk().nthreads; }
GE_DEV int ge_gdim() { return ge_emu::blk().gdim; }
GE_DEV unsigned char *ge_dyn_smem() { return ge_emu::blk().smem; }
GE_DEV void ge_sync() { ge_emu::barrier(); }
GE_DEV void ge_wave_priority(int) {}
GE_DEV void ge_wave_sync() { ge_emu::wave_rendezvous(0); }
GE_DEV void ge_quad_sync() { ge_emu::Block &b = ge_emu::blk(); b.fib[b.cur].waiting = 3; ge_emu::yield_to_sched(); }

GE_DEV void ge_wait_loads() {}
GE_DEV uint64_t ge_ballot(bool p) {
  using namespace ge_emu;
  int g = wave_rendezvous(p ? 1 : 0);
  Block &b = blk(); int w0 = (b.cur / kWave) * kWave; uint64_t m = 0;
  for (int l = 0; l < kWave && w0 + l < b.nthreads; l++) if (!b.fib[w0 + l].done && b.part[g][w0 + l] && b.slot[g][w0 + l]) m |= 1ull << l;
  // second rendezvous so nobody overwrites this generation's slots before all lanes read them
  int g2 = wave_rendezvous(0); (void)g2;
  return m;
}
GE_DEV uint64_t ge_shfl_u64(uint64_t v, int src) {
  using namespace ge_emu;
  int g = wave_rendezvous(v);
  Block &b = blk(); int w0 = (b.cur / kWave) * kWave;
  if (src < 0 || src >= kWave || w0 + src >= b.nthreads) die("shfl source lane out of range");
  uint64_t r = b.slot[g][w0 + src];
  wave_rendezvous(0);
  return r;
}
GE_DEV int ge_shfl_i32(int v, int src) { return (int)(int64_t)ge_shfl_u64((uint64_t)(int64_t)v, src); }
GE_DEV uint32_t ge_shfl_u32(uint32_t v, int src) { return (uint32_t)ge_shfl_u64(v, src); }
GE_DEV uint32_t ge_readlane_u32(uint32_t v, int idx) { return ge_shfl_u32(v, idx); }
GE_DEV uint32_t ge_writelane_u32(uint32_t v, uint32_t val, int idx) { return ((ge_tid() & 63) == idx) ? val : v; }
GE_DEV double ge_shfl_f64(double v, int src) { uint64_t u; memcpy(&u, &v, 8); u = ge_shfl_u64(u, src); memcpy(&v, &u, 8); return v; }

GE_DEV uint32_t ge_quad_xchg(uint32_t v, int xr) {
  ge_emu::Block &b = ge_emu::blk();
  static uint32_t qslot[ge_emu::kMaxThreads];
  qslot[b.cur] = v;
  int me = b.cur;
  ge_quad_sync();
  uint32_t r = qslot[me ^ xr];
  ge_quad_sync();
  return r;
}
GE_DEV uint64_t ge_quad_gather16(uint32_t v) {
  ge_emu::Block &b = ge_emu::blk();
  static uint32_t gslot[ge_emu::kMaxThreads];
  gslot[b.cur] = v;
  const int q0 = b.cur & ~3;
  ge_quad_sync();
  uint64_t r = 0;
  for (int k = 0; k < 4; k++) r |= (uint64_t)(gslot[q0 + k] & 0xffffu) << (16 * k);
  ge_quad_sync();
  return r;
}
GE_DEV void ge_oct_sync() { ge_emu::Block &b = ge_emu::blk(); b.fib[b.cur].waiting = 4; ge_emu::yield_to_sched(); }
GE_DEV uint32_t ge_oct_or32(uint32_t v) {
  ge_emu::Block &b = ge_emu::blk();
  static uint32_t oslot[ge_emu::kMaxThreads];
  oslot[b.cur] = v;
  const int q0 = b.cur & ~7;
  ge_oct_sync();
  uint32_t r = 0;
  for (int k = 0; k < 8; k++) r |= oslot[q0 + k];
  ge_oct_sync();
  return r;
}
GE_DEV uint32_t ge_quad_xor1(uint32_t v) { return ge_quad_xchg(v, 1); }
GE_DEV uint32_t ge_quad_xor2(uint32_t v) { return ge_quad_xchg(v, 2); }
GE_DEV void ge_lds_add_u32(uint32_t *p, uint32_t v) { *p += v; }
GE_DEV void ge_lds_add_f64(double *p, double v) { *p += v; }
GE_DEV uint32_t ge_uniform_u32(uint32_t v) { return v; }
GE_DEV uint32_t ge_readlane_u32(uint32_t v, int idx);
GE_DEV double ge_u64_as_f64(uint64_t v) { double d; memcpy(&d, &v, 8); return d; }
GE_DEV uint64_t ge_f64_as_u64(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }
GE_DEV int ge_popc64(uint64_t v) { return __builtin_popcountll(v); }
GE_DEV int ge_mbcnt(uint64_t m) { return __builtin_popcountll(m & ((1ull << (ge_tid() & 63)) - 1ull)); }
GE_DEV int ge_ctz64(uint64_t v) { return v ? __builtin_ctzll(v) : 64; }
GE_DEV int ge_clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }

template <class T> GE_DEV T atomicAdd(T *p, T v) { T o = *p; *p = (T)(o + v); return o; }
template <class T> GE_DEV T atomicOr(T *p, T v) { T o = *p; *p = (T)(o | v); return o; }
template <class T> GE_DEV T atomicAnd(T *p, T v) { T o = *p; *p = (T)(o & v); return o; }
template <class T> GE_DEV T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }

#define GE_LAUNCH(kernel, grid, block, smem, stream, ...) \
  ge_emu::launch((int)(grid), (int)(block), (size_t)(smem), [=]() { kernel(__VA_ARGS__); })
#define GE_SET_MAX_DYN_LDS(kernel, bytes) (0)

struct ulonglong2 { unsigned long long x, y; };
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }
