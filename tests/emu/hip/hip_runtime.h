// Host stand-in for <hip/hip_runtime.h>, TEST INFRASTRUCTURE ONLY: lets the kernels of
// vector_quantization_amd/csrc/vq_hip.hip run on CPU threads so that their indexing, barriers and
// cross-lane logic can be checked (and debugged with gdb) without a GPU.  One OS thread per GPU
// thread; the workgroups of a launch run one after another; a wave is 64 consecutive threads and
// every cross-lane operation is a rendezvous of those 64 threads.  Nothing here is part of the
// product: the package never includes or loads it (tests/emu/build_emu.py builds a separate library).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __align__(n) __attribute__((aligned(n)))
#define __shared__ static      // one workgroup runs at a time: a function-level static is shared by its threads

struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; } __attribute__((aligned(16)));
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

typedef int hipError_t;
typedef void *hipStream_t;
constexpr hipError_t hipSuccess = 0;
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8, hipDeviceAttributeMultiprocessorCount = 63,
       hipDeviceAttributeMaxSharedMemoryPerBlock = 74 };

namespace emu {
constexpr int WAVE = 64;
constexpr size_t LDS_BYTES = 160 * 1024;

struct Barrier {                       // reusable; threads that finish early leave with drop()
    // the bookkeeping sits under a mutex; waiters spin on the generation counter and yield (hundreds of
    // threads share a few cores: a condition variable's wake-ups cost far more than the yields)
    std::mutex m;
    int count = 0, waiting = 0;
    std::atomic<uint64_t> gen{0};
    void init(int n) { count = n; waiting = 0; }
    void wait() {
        uint64_t g;
        {
            std::lock_guard<std::mutex> l(m);
            g = gen.load(std::memory_order_relaxed);
            if (++waiting == count) {
                waiting = 0;
                gen.store(g + 1, std::memory_order_release);
                return;
            }
        }
        while (gen.load(std::memory_order_acquire) == g) std::this_thread::yield();
    }
    void drop() {
        std::lock_guard<std::mutex> l(m);
        --count;
        if (count > 0 && waiting == count) {
            waiting = 0;
            gen.store(gen.load(std::memory_order_relaxed) + 1, std::memory_order_release);
        }
    }
};
struct WaveState { Barrier bar; uint64_t x[WAVE]; };

struct Block {
    Barrier bar;
    std::vector<WaveState> waves;
};
extern Block *g_block;                                 // the workgroup that is running
extern unsigned char g_lds[LDS_BYTES] __attribute__((aligned(256)));
extern int g_cus;
struct Idx { unsigned x, y, z; };
}  // namespace emu

extern thread_local emu::Idx threadIdx, blockIdx, blockDim, gridDim;

namespace emu {
inline int lane() { return (int)(threadIdx.x & (WAVE - 1)); }
inline WaveState &wave() { return g_block->waves[threadIdx.x / WAVE]; }
template <class T> inline T lane_read(T v, int src) {  // value of lane `src` (every lane of the wave must call)
    static_assert(sizeof(T) <= 8, "");
    WaveState &w = wave();
    uint64_t b = 0;
    memcpy(&b, &v, sizeof(T));
    w.x[lane()] = b;
    w.bar.wait();
    const uint64_t r = w.x[src & (WAVE - 1)];
    w.bar.wait();
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}
inline int dpp_source(int ctrl) {
    const int l = lane();
    switch (ctrl) {
        case 0xB1: return l ^ 1;                              // quad_perm [1,0,3,2]
        case 0x4E: return l ^ 2;                              // quad_perm [2,3,0,1]
        case 0x141: return (l & ~7) | (7 - (l & 7));          // row_half_mirror
        case 0x140: return (l & ~15) | (15 - (l & 15));       // row_mirror
        default: fprintf(stderr, "emu: DPP control %#x is not modelled\n", ctrl); abort();
    }
}
template <class F> void launch(dim3 grid, dim3 block, size_t lds, F body) {
    if (lds > LDS_BYTES || block.x == 0 || block.x > 1024 || block.x % WAVE != 0 || block.y != 1 || block.z != 1) {
        fprintf(stderr, "emu: launch shape not modelled (block %u, lds %zu)\n", block.x, lds);
        abort();
    }
    // block.x OS threads serve all workgroups of the launch, one workgroup after another: before each one,
    // thread 0 re-arms the workgroup's barriers and fills the LDS with garbage (as on the GPU, it comes up
    // undefined); `outer` keeps the threads together between workgroups
    const unsigned nthreads = block.x;
    Block blk;
    blk.waves = std::vector<WaveState>(nthreads / WAVE);
    Barrier outer;
    outer.init((int)nthreads);
    g_block = &blk;
    std::vector<std::thread> th;
    th.reserve(nthreads);
    for (unsigned t = 0; t < nthreads; ++t)
        th.emplace_back([&, t] {
            for (unsigned bz = 0; bz < grid.z; ++bz)
                for (unsigned by = 0; by < grid.y; ++by)
                    for (unsigned bx = 0; bx < grid.x; ++bx) {
                        if (t == 0) {
                            blk.bar.init((int)nthreads);
                            for (auto &w : blk.waves) w.bar.init(WAVE);
                            memset(g_lds, 0xCD, LDS_BYTES);
                        }
                        outer.wait();
                        threadIdx = Idx{t, 0, 0};
                        blockIdx = Idx{bx, by, bz};
                        blockDim = Idx{block.x, 1, 1};
                        gridDim = Idx{grid.x, grid.y, grid.z};
                        body();
                        blk.waves[t / WAVE].bar.drop();
                        blk.bar.drop();
                        outer.wait();
                    }
        });
    for (auto &t : th) t.join();
    g_block = nullptr;
}
}  // namespace emu

template <class K, class... A>
inline void hipLaunchKernelGGL(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t, A... args) {
    emu::launch(grid, block, lds, [=] { kern(args...); });
}
inline void __syncthreads() { emu::g_block->bar.wait(); }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ---- cross-lane operations -------------------------------------------------------------------------
template <class T> inline T __shfl_xor(T v, int mask, int = 64) { return emu::lane_read(v, emu::lane() ^ mask); }
template <class T> inline T __shfl_up(T v, unsigned delta, int = 64) {
    const int src = emu::lane() - (int)delta;
    const T r = emu::lane_read(v, src < 0 ? emu::lane() : src);
    return r;
}
inline unsigned long long __ballot(int pred) {
    emu::WaveState &w = emu::wave();
    w.x[emu::lane()] = pred ? 1u : 0u;
    w.bar.wait();
    unsigned long long m = 0;
    for (int i = 0; i < emu::WAVE; ++i) m |= (unsigned long long)(w.x[i] & 1u) << i;
    w.bar.wait();
    return m;
}
inline int emu_update_dpp(int, int src, int ctrl, int, int, bool) { return emu::lane_read(src, emu::dpp_source(ctrl)); }
inline uint32_t emu_readlane(uint32_t v, int l) { return emu::lane_read(v, l); }
inline uint32_t emu_readfirstlane(uint32_t v) { return emu::lane_read(v, 0); }
inline uint32_t emu_sad_u16(uint32_t a, uint32_t b, uint32_t acc) {
    const int32_t a0 = a & 0xffff, a1 = a >> 16, b0 = b & 0xffff, b1 = b >> 16;
    return acc + (uint32_t)std::abs(a0 - b0) + (uint32_t)std::abs(a1 - b1);
}
inline uint32_t emu_sad_u8(uint32_t a, uint32_t b, uint32_t acc) {
    for (int i = 0; i < 4; ++i) acc += (uint32_t)std::abs((int)((a >> (8 * i)) & 0xff) - (int)((b >> (8 * i)) & 0xff));
    return acc;
}
inline uint32_t emu_perm(uint32_t s0, uint32_t s1, uint32_t sel) {     // v_perm_b32, selector values 0..7 and 12 (zero byte)
    const uint64_t both = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t k = (sel >> (8 * i)) & 0xff;
        if (k == 12) continue;                                          // selector 0x0c: constant 0x00
        if (k > 7) { fprintf(stderr, "emu: v_perm selector %u is not modelled\n", k); abort(); }
        r |= (uint32_t)((both >> (8 * k)) & 0xff) << (8 * i);
    }
    return r;
}
#define __builtin_amdgcn_update_dpp emu_update_dpp
#define __builtin_amdgcn_readlane emu_readlane
#define __builtin_amdgcn_readfirstlane emu_readfirstlane
#define __builtin_amdgcn_sad_u16 emu_sad_u16
#define __builtin_amdgcn_sad_u8 emu_sad_u8
#define __builtin_amdgcn_perm emu_perm
#define __builtin_amdgcn_sched_barrier(x) ((void)0)

// ---- scalar helpers of the HIP device library ------------------------------------------------------------
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline long long __double_as_longlong(double d) { long long r; memcpy(&r, &d, 8); return r; }
inline double __longlong_as_double(long long v) { double r; memcpy(&r, &v, 8); return r; }
inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }

// ---- atomics (LDS and global alike: plain memory here) ----------------------------------------------------
inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline uint32_t atomicMax(uint32_t *p, uint32_t v) {
    uint32_t o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
inline uint32_t atomicMin(uint32_t *p, uint32_t v) {
    uint32_t o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o > v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
inline unsigned long long atomicCAS(unsigned long long *p, unsigned long long cmp, unsigned long long v) {
    __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return cmp;
}

// ---- the few runtime calls the host side of the library makes ----------------------------------------------
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int *v, int attr, int) {
    *v = attr == hipDeviceAttributeMultiprocessorCount ? emu::g_cus : (int)emu::LDS_BYTES;
    return hipSuccess;
}
inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
typedef void *hipEvent_t;
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
#define HIP_SYMBOL(x) (x)
inline hipError_t hipGetSymbolAddress(void **p, const void *symbol) { *p = const_cast<void *>(symbol); return hipSuccess; }
inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
