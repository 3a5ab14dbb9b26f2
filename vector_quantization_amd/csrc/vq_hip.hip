// vq_hip.hip -- HIP kernels (gfx950 / CDNA4) and the C ABI of libvqhip.so.
//
// Hot path of VQ_Full.py rebuilt for MI355X:
//   block extraction (VQ_Full.py:23-33, 51-57), L1 nearest-codeword search (:77-78, :124-125),
//   LBG split / centroid update (:67-73, :79-90), decode (:157-169), label bit-packing (:134-147).
//
// Nearest-codeword search, FAST mode.  The reference's distance is a left-to-right fp64 sum of
// |x_e - c_e| with x_e an integer in 0..255.  The search runs in two tiers:
//   1. Filter.  Codebook entries are rounded to 16-bit fixed point C = rint(c * 2^s) (s <= 8 chosen
//      so the largest entry fits) and pixels are scaled exactly, X = x << s.  v_sad_u16 adds two
//      |X - C| terms per instruction into a 32-bit sum S_k.  The exact scaled distance T_k obeys
//      |T_k - S_k| <= E_k = sum_e |c_e * 2^s - C_e|, and the kernel knows Emax >= every E_k.
//      Each lane tracks the two smallest keys (S_k << kbits | k); when S_second - S_first >
//      2 * Emax the first key's codeword is the unique fp64 argmin and no fp64 work is needed.
//   2. Re-check.  Otherwise the vector's index is appended to a queue in the workspace, and a second
//      kernel lets one whole wave evaluate each queued vector against every codeword in sequential
//      fp64 (lanes <-> codewords, coalesced reads of a transposed fp64 codebook), reducing
//      (distance, index) lexicographically, which is numpy's first-minimum rule.
//   A codeword whose row is bit-identical to an earlier row can never win (ties go to the lower
//   index), so the filter gives it a key that never competes; duplicated codewords (the twin that
//   the split of an all-zero centroid produces) therefore do not force re-checks.
//   Pruning.  sum_e |X_e - C_e| >= |sum_e X_e - sum_e C_e| holds exactly in integers.  With the
//   fixed-point rows kept in ascending order of their element sum and the vectors of a wave chosen
//   to have nearly equal element sums (vq_blocks_sort), a wave searches only the contiguous range of
//   rows whose element sum lies within (worst best-distance so far + 2 * Emax) of its vectors' sums;
//   every skipped codeword is provably neither the nearest nor a near-tie, so results are unchanged.
// EXACT mode evaluates every distance in sequential fp64 and is the on-device cross-check.
//
// Distortion.  The per-vector winning distances go to an array, and their sum is taken in the
// order numpy.mean uses (VQ_Full.py:90): pairwise summation inside 8192-element chunks, chunk sums
// added left to right.  The stop rule of the training loop therefore sees the reference's value.
//
// No entry point allocates memory; scratch comes in through the workspace argument.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <unordered_map>

#include "vq_hip.h"

// Scratch of vq_encode_l1_static, the entry point for callers that cannot allocate device memory: it
// takes the place of the caller's workspace.  Every GPU the library runs on holds its own copy of the
// array, at its own address, until the process ends; host code reaches it only through the address
// that hipGetSymbolAddress returns for the current device (static_scratch below), never by this name.
static __device__ __align__(256) unsigned char scratch[VQ_STATIC_SCRATCH_BYTES];

#ifdef VQ_TIMELINE
// Diagnostic builds only (never the shipped library): per wave of the last k_assign_fast launch, its start and
// end time (100 MHz counter), hardware id register and the codebook rows it searched.
static __device__ uint32_t g_timeline[8 << 18];
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// host-side error plumbing
// ------------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(VQ_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define LAUNCH_CHECK(name)                                                                         \
    do {                                                                                           \
        hipError_t e_ = hipGetLastError();                                                         \
        if (e_ != hipSuccess) return fail(VQ_E_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

constexpr int ABI_VERSION = 8;
constexpr int WAVE = 64;
constexpr int64_t NP_CHUNK = 8192;      // numpy's reduction buffer: elements per pairwise-summed chunk
constexpr int MAX_D_FAST = 192;         // largest vector dimension of the SAD filter kernels
constexpr int MAX_K = 1 << 20;

// Tuning knobs of the d <= 16 filter kernels: vectors per lane, codewords per scalar-load batch.
#ifndef VQ_R8
#define VQ_R8 4
#endif
#ifndef VQ_U8
#define VQ_U8 4
#endif
#ifndef VQ_RBIG
#define VQ_RBIG 2
#endif
// Pruned search: vectors per lane, codewords per scalar-load batch (d <= 16), codewords added per side and round.
#ifndef VQ_RP
#define VQ_RP 2
#endif
#ifndef VQ_UP
#define VQ_UP 2
#endif
#ifndef VQ_GP
#define VQ_GP 32
#endif
#ifndef VQ_G0
#define VQ_G0 32
#endif
// Pruned search for rows of 34 .. 64 bytes (0 = such rows take the LDS-tiled full search).
#ifndef VQ_PRUNE_WIDE
#define VQ_PRUNE_WIDE 1
#endif

// Sorted-codebook position table: entry t = number of codewords whose fixed-point element sum is
// below t << tshift; POSTAB_NT + 2 entries are used at most.
constexpr int POSTAB_NT = 1024;
constexpr int POSTAB_WORDS = POSTAB_NT + 8;
constexpr int PREP_SORT_CAP = 4096;     // codebooks up to this size are sorted by element sum
constexpr int SORT_WIN = 8192;          // vectors per window of the windowed vector sort
constexpr int SORT_BUCKETS = 2048;      // counting-sort buckets per window

// Ambiguous vectors are queued in NQ sub-queues, each with its own counter in its own cache line:
// the group of 64 consecutive vectors g goes to sub-queue g % NQ.  (One counter for the whole
// launch serialises some 10^4 same-address atomics and was measured to bound the kernel.)
constexpr int NQ = 64;
// The fused accumulation flushes its LDS tables into NREP replicas of the [K][d+1] table (workgroup w
// into replica w % NREP): atomics to one small table serialise on its few cache lines.
constexpr int NREP = 32;
inline size_t queue_cap(int64_t N) {                       // entries per sub-queue
    const size_t groups = ((size_t)N + 63) / 64;
    return (groups + NQ - 1) / NQ * 64;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Workspace layout (all offsets multiples of 256 bytes).
struct WsLayout {
    size_t header;     // PrepHeader
    size_t c16;        // Kpad * dp2 uint32: codebook as packed pairs of 16-bit fixed point, rows in
                       // ascending order of their element sum (Kpad = K rounded up to 8; pad rows never win)
    size_t bias;       // Kpad uint32: key bias (original row index, plus the never-wins bit for duplicate rows)
    size_t c16u;       // K * dp2 uint32: the same rows in original order (scratch of the prepare kernel)
    size_t postab;     // POSTAB_WORDS uint32: sum threshold -> position in the sorted codebook
    size_t prep_tmp;   // scratch of the prepare kernels: K uint64 row hashes, K uint32 element sums, K uint32 sorted sums
    size_t c16T;       // dp2 * K uint32: the c16 rows transposed (word p of every sorted position side by side), kept
                       // for sorted codebooks too large for the LDS of the fix-up kernel (else empty)
    size_t acc_rep;    // NREP * K * (d + 1) uint64: replicas of the [sums | count] table (fused accumulation)
    size_t cbT;        // d * K doubles: transposed fp64 codebook, columns in the order of the c16 rows
    size_t cb_u8;      // K * d bytes: codebook truncated to uint8 (decode)
    size_t chunks;     // ceil(N / 8192) + 1 doubles: per-chunk sums of the distance array
    size_t qcount;     // NQ counters, 64 bytes apart: entries in each sub-queue of ambiguous vectors
    size_t queue;      // NQ * queue_cap(N) uint32: indices of the vectors the fix-up kernel must re-check
    size_t total;
};

int dp2_bucket(int d);   // forward

// Sorted codebooks whose transposed fp64 copy exceeds the LDS budget of k_fixup_pruned take k_fixup_shortlist,
// which reads the fixed-point rows transposed.
inline bool wants_c16T(int K, int d) { return K <= PREP_SORT_CAP && (size_t)K * d * 8 > 128 * 1024; }

WsLayout ws_layout(int K, int d, int64_t N) {
    WsLayout L;
    int dp2 = dp2_bucket(d);
    if (dp2 <= 0) dp2 = (d + 1) / 2;
    size_t off = 0;
    L.header = off;   off += 256;
    const size_t kpad = ((size_t)K + 7) / 8 * 8;
    L.c16 = off;      off += align_up(kpad * dp2 * 4, 256);
    L.bias = off;     off += align_up(kpad * 4, 256);
    L.c16u = off;     off += align_up((size_t)K * dp2 * 4, 256);
    L.postab = off;   off += align_up((size_t)POSTAB_WORDS * 4, 256);
    L.prep_tmp = off; off += align_up((size_t)K * 16, 256);
    L.c16T = off;     off += wants_c16T(K, d) ? align_up((size_t)K * dp2 * 4, 256) : 0;
    L.acc_rep = off;  off += (size_t)K * (d + 1) * 4 <= 40 * 1024 ? align_up((size_t)NREP * K * (d + 1) * 8, 256) : 0;
    L.cbT = off;      off += align_up((size_t)K * d * 8, 256);
    L.cb_u8 = off;    off += align_up((size_t)K * d, 256);
    L.chunks = off;   off += align_up((size_t)((N + NP_CHUNK - 1) / NP_CHUNK + 1) * 8, 256);
    L.qcount = off;   off += (size_t)NQ * 64;
    L.queue = off;    off += align_up((size_t)NQ * queue_cap(N) * 4, 256);
    L.total = off;
    return L;
}

// Header that the prepare kernel writes and the assignment kernel reads.
struct PrepHeader {
    uint32_t shift;     // s: fixed-point scale is 2^s
    uint32_t emax;      // integer bound, strictly greater than every E_k (units of 2^-s)
    uint32_t cmax;      // ceil of the largest codebook entry, saturated
    uint32_t sorted;    // 1: rows of c16 / bias are in ascending order of element sum and postab is valid
    uint32_t tshift;    // postab granularity: entry t covers sums [t << tshift, (t + 1) << tshift)
    uint32_t nt;        // postab entries 0 .. nt + 1 are valid; entry nt + 1 = K
    uint32_t pad[58];
};
static_assert(sizeof(PrepHeader) == 256, "header is one 256-byte slot");

// Header at the end of a window-sorted buffer, written by the sort kernel and verified by the pruned
// search before it trusts the origin map (a buffer whose contents were replaced since the sort makes
// the search write label -1 everywhere instead of following stale origins).
constexpr uint32_t SORTED_MAGIC = 0x56515352u;      // "VQSR"
struct SortedHeader {
    uint32_t magic;
    uint32_t d;
    unsigned long long N;
    uint32_t pad[60];
};
static_assert(sizeof(SortedHeader) == 256, "header is one 256-byte slot");

// Gate of the kernels that vq_lbg_iterate enqueues.  The host enqueues iteration `iter` of a stage before it
// knows whether an earlier iteration met the stop rule; the stage's device-resident state records the
// iteration after which the stage stopped, and the kernels of later iterations return at once.
struct Gate {
    const int32_t *stopped_after;    // null: no gate (every other entry point)
    int iter;
};
__device__ __forceinline__ bool gate_closed(const Gate &g) { return g.stopped_after != nullptr && *g.stopped_after < g.iter; }
// the gate the launches of the calling thread carry (set for the duration of vq_lbg_iterate only)
thread_local Gate g_gate = {nullptr, 0};
// set by vq_lbg_iterate when it has already zeroed the queue counters and the replica tables of the workspace
// together with the caller's sums / counts (one launch instead of four fills)
thread_local bool g_ws_zeroed = false;

// Device-resident state of one LBG stage (vq_lbg_stage_begin / vq_lbg_iterate).
struct LbgRecord {
    double distortion;               // mean winning distance of this iteration (VQ_Full.py:90)
    uint32_t converged;              // 1: the stop rule of VQ_Full.py:92-94 held at this iteration
    uint32_t valid;                  // 1: the iteration ran (0: it was gated off)
};
struct LbgState {
    int32_t stopped_after;           // iteration at which the stop rule held (INT32_MAX while running)
    uint32_t pad0;
    double prev;                     // previous iteration's distortion (+inf before the first)
    double dist_sum;                 // numpy-order sum of the winning distances of the current iteration
    double pad1[5];
    LbgRecord rec[1];                // one per iteration of the stage
};
static_assert(offsetof(LbgState, rec) == 64, "records start at byte 64");

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Maximum / minimum of a 32-bit value over the 64 lanes of a (fully active) wave, returned as a
// wave-uniform value.  Four DPP steps reduce each row of 16 lanes, scalar code joins the rows.
template <bool IS_MAX>
__device__ __forceinline__ uint32_t wave_reduce_u32(uint32_t v) {
    auto op = [](uint32_t a, uint32_t b) { return IS_MAX ? (a > b ? a : b) : (a < b ? a : b); };
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));   // row_half_mirror
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));   // row_mirror
    const uint32_t r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const uint32_t r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    // readfirstlane marks the result as wave-uniform for the compiler: what is computed from it stays in
    // scalar registers and table lookups indexed by it become scalar loads
    return __builtin_amdgcn_readfirstlane(op(op(r0, r1), op(r2, r3)));
}

// Scalar loads issued ahead of their use.  The compiler does not track when their data arrives (it would wait for it
// at once): the registers may be read only after sload_wait, which every such load must be followed by.
#ifndef VQ_HOST_EMULATION
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sload_x16(u32x16 &r, const uint32_t *base, uint32_t byte_off) {
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(r) : "s"(base), "s"(byte_off));
}
__device__ __forceinline__ void sload_x2(u32x2 &r, const uint32_t *base, uint32_t byte_off) {
    asm volatile("s_load_dwordx2 %0, %1, %2" : "=&s"(r) : "s"(base), "s"(byte_off));
}
__device__ __forceinline__ void sload_wait(u32x16 &a, u32x2 &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); }
#else
struct u32x16 {
    uint32_t v[16];
    uint32_t operator[](int i) const { return v[i]; }
};
struct u32x2 {
    uint32_t v[2];
    uint32_t operator[](int i) const { return v[i]; }
};
inline void sload_x16(u32x16 &r, const uint32_t *base, uint32_t byte_off) {
    for (int i = 0; i < 16; ++i) r.v[i] = base[byte_off / 4 + i];
}
inline void sload_x2(u32x2 &r, const uint32_t *base, uint32_t byte_off) {
    for (int i = 0; i < 2; ++i) r.v[i] = base[byte_off / 4 + i];
}
inline void sload_wait(u32x16 &, u32x2 &) {}
#endif  // VQ_HOST_EMULATION

// Four bytes at ANY byte address as one word.  Global memory takes unaligned accesses on this hardware (one load
// instruction, served at a fraction of the aligned rate; assembling the word from its two aligned neighbours with a
// funnel shift was measured slower still), which is what lets images whose rows are not whole words (W * C % 4 != 0:
// most RGB photographs) be read word-wise like all others.
__device__ __forceinline__ uint32_t load_u32_any(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// fp64 -> uint8 the way numpy converts inside np.array(..., dtype=np.uint8): truncate toward
// zero and keep the low 8 bits; magnitudes beyond the int32 range (and NaN) give 0.
__device__ __forceinline__ uint8_t cast_u8(double v) {
    if (!(fabs(v) < 2147483648.0)) return 0;
    return (uint8_t)(uint32_t)(int32_t)v;
}

// Lexicographic (distance, index) minimum across the 64 lanes of a wave.
__device__ __forceinline__ void wave_argmin(double &best, int &bidx) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double ob = __shfl_xor(best, off, WAVE);
        int oi = __shfl_xor(bidx, off, WAVE);
        if (ob < best || (ob == best && oi < bidx)) {
            best = ob;
            bidx = oi;
        }
    }
}

// Whole-wave exact search for ONE vector (xrow: d bytes, wave-uniform address; xv0: lane e's copy
// of x[e] for e < 64, already loaded).  Lane l evaluates codewords l, l+64, ... in sequential fp64
// against the transposed codebook cbT[e*K + k] (in LDS when LDS_CB).  Loads are issued eight
// elements (x four codewords) at a time so that one memory round trip serves 32 terms; the
// additions still run in element order, one chain per codeword.
template <bool LDS_CB>
__device__ __forceinline__ void exact_search_wave(uint32_t xv0, const uint8_t *__restrict__ xrow, int d,
                                                  const double *__restrict__ cbT, const double *s_cbT, int K,
                                                  int lane, int &out_idx, double &out_dist) {
    double best = __builtin_huge_val();
    int bidx = 0x7fffffff;
    for (int k0 = 0; k0 < K; k0 += 4 * WAVE) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        int kk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int k = k0 + q * WAVE + lane;
            kk[q] = k < K ? k : K - 1;                         // clamped: loads stay in bounds
        }
        for (int ebase = 0; ebase < d; ebase += WAVE) {
            const int ecount = d - ebase < WAVE ? d - ebase : WAVE;
            uint32_t xv = xv0;                                  // lane e holds x[ebase + e]
            if (ebase > 0) xv = lane < ecount ? (uint32_t)xrow[ebase + lane] : 0u;
            int e0 = 0;
            for (; e0 + 8 <= ecount; e0 += 8) {                  // full batches: no predication
                double c[8][4];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const size_t off = (size_t)(ebase + e0 + j) * K;
#pragma unroll
                    for (int q = 0; q < 4; ++q) c[j][q] = LDS_CB ? s_cbT[off + kk[q]] : cbT[off + kk[q]];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double xe = (double)__builtin_amdgcn_readlane(xv, e0 + j);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] += fabs(xe - c[j][q]);
                }
            }
            for (; e0 < ecount; ++e0) {                          // tail of a dimension that is not a multiple of 8
                const size_t off = (size_t)(ebase + e0) * K;
                const double xe = (double)__builtin_amdgcn_readlane(xv, e0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += fabs(xe - (LDS_CB ? s_cbT[off + kk[q]] : cbT[off + kk[q]]));
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int k = k0 + q * WAVE + lane;
            if (k < K && acc[q] < best) {
                best = acc[q];
                bidx = k;
            }
        }
    }
    wave_argmin(best, bidx);
    out_idx = bidx;
    out_dist = best;
}

// ------------------------------------------------------------------------------------------------
// block extraction (VQ_Full.py:23-33, 51-57)
// ------------------------------------------------------------------------------------------------
// One thread per group of G consecutive bytes of the block matrix.  G = 16 (four 4-byte words,
// one 16-byte store) or 4 when a block row segment (bw*C bytes) is a multiple of 4, else 1.  A
// word that lies inside the image is fetched with one load (at whatever byte alignment the image row gives
// it); words in the padding are assembled byte by byte.
template <int G>
__global__ __launch_bounds__(256) void k_blocks_u8(const uint8_t *__restrict__ img, int H, int W, int C, int bh,
                                                   int bw, int pad_zero, uint8_t *__restrict__ blocks, int nc,
                                                   int64_t total_groups, int img_words_ok) {
    const int seg = bw * C;            // bytes of one block row segment
    const int d = bh * seg;
    const int64_t rowbytes = (int64_t)W * C;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total_groups;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = g * G;       // first output byte
        if constexpr (G == 1) {
            int64_t n = o / d;
            int e = (int)(o - n * d);
            int r = (int)(n / nc), q = (int)(n - (int64_t)r * nc);
            int i = e / seg, rem = e - i * seg;
            int y = r * bh + i;
            int yc = y < H ? y : H - 1;
            int j = rem / C, ch = rem - j * C;
            int x = q * bw + j;
            int xc = x < W ? x : W - 1;
            uint8_t v = img[((size_t)yc * W + xc) * C + ch];
            if (pad_zero && (y >= H || x >= W)) v = 0;
            blocks[o] = v;
        } else {
            constexpr int NW = G >= 4 ? G / 4 : 1;
            uint32_t words[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int64_t ow = o + 4 * w;
                int64_t n = ow / d;
                int e = (int)(ow - n * d);
                int r = (int)(n / nc), q = (int)(n - (int64_t)r * nc);
                int i = e / seg, rem = e - i * seg;
                int y = r * bh + i;
                int yc = y < H ? y : H - 1;
                int64_t xb = (int64_t)q * seg + rem;           // byte offset inside the (padded) image row
                uint32_t v;
                if (img_words_ok && xb + 4 <= rowbytes && (y < H || !pad_zero)) {
                    v = load_u32_any(img + (size_t)yc * rowbytes + xb);
                } else {
                    v = 0;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        int j = (rem + t) / C, ch = (rem + t) - j * C;
                        int x = q * bw + j;
                        int xc = x < W ? x : W - 1;
                        uint32_t b = img[((size_t)yc * W + xc) * C + ch];
                        if (pad_zero && (y >= H || x >= W)) b = 0;
                        v |= b << (8 * t);
                    }
                }
                words[w] = v;
            }
            if (G == 16) {
                uint4 out = make_uint4(words[0], words[1 % NW], words[2 % NW], words[3 % NW]);
                *reinterpret_cast<uint4 *>(blocks + o) = out;
            } else {
                *reinterpret_cast<uint32_t *>(blocks + o) = words[0];
            }
        }
    }
}

// Blocks whose row segments are whole words (bw * C % 4 == 0) and whose vectors are whole 16-byte pieces (d % 16 == 0,
// d <= 192) -- RGB 4x4 and 8x8, gray 8x8: a workgroup takes up to 256 consecutive blocks of one block row.  For each of
// their bh pixel rows it copies the contiguous stretch of the image row into LDS, a thread moving the bw * C bytes of
// its block to the block's place there (adjacent lanes read adjacent segments, at whatever byte alignment the image
// row has); then the vectors leave as one contiguous run of 16-byte stores.  No division per word, every image byte
// read once, every store a full line.  Words that reach into the padding are assembled byte by byte.
__global__ __launch_bounds__(256) void k_blocks_rows(const uint8_t *__restrict__ img, int H, int W, int C, int bh, int bw,
                                                     int pad_zero, uint8_t *__restrict__ blocks, int nr, int nc) {
    extern __shared__ __align__(16) uint32_t s_tile[];           // 256 * d / 4 words
    constexpr int TB = 256;
    const int seg = bw * C, segw = seg >> 2, dw = (bh * seg) >> 2;
    const int64_t rowbytes = (int64_t)W * C;
    const int tiles_per_row = (nc + TB - 1) / TB;
    const int64_t ntiles = (int64_t)nr * tiles_per_row;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {    // (workgroup-uniform)
        const int r = (int)(t / tiles_per_row);
        const int q0 = (int)(t - (int64_t)r * tiles_per_row) * TB;
        const int nb = nc - q0 < TB ? nc - q0 : TB;
        const int qq = threadIdx.x;
        if (qq < nb) {
            const int64_t xb0 = (int64_t)(q0 + qq) * seg;         // first byte of the block's segment in an image row
            for (int i = 0; i < bh; ++i) {
                const int y = r * bh + i;
                const uint8_t *src = img + (size_t)(y < H ? y : H - 1) * rowbytes;
                const bool zero_row = pad_zero && y >= H;
                uint32_t *dst = s_tile + qq * dw + i * segw;
                for (int rem = 0; rem < segw; ++rem) {
                    const int64_t xb = xb0 + 4 * rem;
                    uint32_t v = 0;
                    if (!zero_row) {
                        if (xb + 4 <= rowbytes) {
                            v = load_u32_any(src + xb);
                        } else {
#pragma unroll
                            for (int b4 = 0; b4 < 4; ++b4) {
                                const int j = (4 * rem + b4) / C, ch = (4 * rem + b4) - j * C;
                                const int x = (q0 + qq) * bw + j;
                                uint32_t b = src[(size_t)(x < W ? x : W - 1) * C + ch];
                                if (pad_zero && x >= W) b = 0;
                                v |= b << (8 * b4);
                            }
                        }
                    }
                    dst[rem] = v;
                }
            }
        }
        __syncthreads();
        uint4 *out = reinterpret_cast<uint4 *>(blocks + ((size_t)r * nc + q0) * ((size_t)dw * 4));
        const int n16 = nb * (dw >> 2);
        for (int k = threadIdx.x; k < n16; k += 256) out[k] = reinterpret_cast<const uint4 *>(s_tile)[k];
        __syncthreads();
    }
}

// Any block shape (segments that are not whole words: 2x2 and 3x3 blocks, RGB 2x2 ...): a workgroup takes a strip of
// up to TB consecutive blocks of one block row.  Phase 1 copies the strip's stretch of each of its bh image rows into
// LDS word by word (coalesced; at whatever byte alignment the image row has; words that reach past the row are
// assembled byte by byte with the padding rule).  Phase 2: a thread per block collects the block's bytes from the LDS
// rows and stores its vector -- as words when d is a multiple of four (adjacent lanes then write adjacent vectors).
// rs: bytes per LDS row, a multiple of four, at least TB * seg.
__global__ __launch_bounds__(256) void k_blocks_strip(const uint8_t *__restrict__ img, int H, int W, int C, int bh, int bw,
                                                      int pad_zero, uint8_t *__restrict__ blocks, int nr, int nc, int TB, int rs,
                                                      int word_stores) {
    extern __shared__ __align__(16) uint32_t s_rows[];           // bh rows of rs bytes
    const uint8_t *s_bytes = reinterpret_cast<const uint8_t *>(s_rows);
    const int seg = bw * C, d = bh * seg;
    const int64_t rowbytes = (int64_t)W * C;
    const int tiles_per_row = (nc + TB - 1) / TB;
    const int64_t ntiles = (int64_t)nr * tiles_per_row;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {    // (workgroup-uniform)
        const int r = (int)(t / tiles_per_row);
        const int q0 = (int)(t - (int64_t)r * tiles_per_row) * TB;
        const int nb = nc - q0 < TB ? nc - q0 : TB;
        const int64_t xb_start = (int64_t)q0 * seg;
        const int nw = (nb * seg + 3) >> 2;
        for (int i = 0; i < bh; ++i) {
            const int y = r * bh + i;
            const uint8_t *src = img + (size_t)(y < H ? y : H - 1) * rowbytes;
            const bool zero_row = pad_zero && y >= H;
            for (int w = threadIdx.x; w < nw; w += 256) {
                const int64_t xb = xb_start + 4 * w;
                uint32_t v = 0;
                if (!zero_row) {
                    if (xb + 4 <= rowbytes) {
                        v = load_u32_any(src + xb);
                    } else {
#pragma unroll
                        for (int b4 = 0; b4 < 4; ++b4) {
                            const int64_t j = (xb + b4) / C;                 // pixel column of this byte
                            const int ch = (int)((xb + b4) - j * C);
                            uint32_t b = src[(size_t)(j < W ? j : W - 1) * C + ch];
                            if (pad_zero && j >= W) b = 0;
                            v |= b << (8 * b4);
                        }
                    }
                }
                s_rows[i * (rs >> 2) + w] = v;
            }
        }
        __syncthreads();
        for (int qq = threadIdx.x; qq < nb; qq += 256) {
            uint8_t *out = blocks + ((size_t)r * nc + q0 + qq) * (size_t)d;
            const uint8_t *col = s_bytes + qq * seg;
            int i = 0, rem = 0;                                   // element e = i * seg + rem of the vector
            if (word_stores) {
                for (int e = 0; e < d; e += 4) {
                    uint32_t v = 0;
#pragma unroll
                    for (int b4 = 0; b4 < 4; ++b4) {
                        v |= (uint32_t)col[i * rs + rem] << (8 * b4);
                        if (++rem == seg) {
                            rem = 0;
                            ++i;
                        }
                    }
                    *reinterpret_cast<uint32_t *>(out + e) = v;
                }
            } else {
                for (int e = 0; e < d; ++e) {
                    out[e] = col[i * rs + rem];
                    if (++rem == seg) {
                        rem = 0;
                        ++i;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// 4x4 blocks of one-byte pixels, no padding, image rows multiples of 16 aligned bytes: a thread reads four
// horizontally adjacent blocks with four 16-byte loads (one per pixel row; a wave's loads are contiguous
// runs of the four rows) and regroups the words into the four 16-byte vectors.  The vectors go through LDS so
// that every store instruction of a wave writes 1 KB of consecutive vectors.
__global__ __launch_bounds__(256) void k_blocks_quads(const uint8_t *__restrict__ img, int W, uint8_t *__restrict__ blocks,
                                                      int nc, int64_t total_quads) {
    __shared__ uint4 s_vec[1024];
    const int tid = threadIdx.x;
    for (int64_t q0 = (int64_t)blockIdx.x * 256; q0 < total_quads; q0 += (int64_t)gridDim.x * 256) {   // (uniform)
        const int64_t qg = q0 + tid;
        if (qg < total_quads) {
            const uint32_t n0 = (uint32_t)qg * 4, r = n0 / (uint32_t)nc, q = n0 - r * (uint32_t)nc;
            const uint8_t *base = img + ((size_t)r * 4) * (size_t)W + (size_t)q * 4;
            uint4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const uint4 *>(base + (size_t)i * W);
            s_vec[tid * 4 + 0] = make_uint4(v[0].x, v[1].x, v[2].x, v[3].x);
            s_vec[tid * 4 + 1] = make_uint4(v[0].y, v[1].y, v[2].y, v[3].y);
            s_vec[tid * 4 + 2] = make_uint4(v[0].z, v[1].z, v[2].z, v[3].z);
            s_vec[tid * 4 + 3] = make_uint4(v[0].w, v[1].w, v[2].w, v[3].w);
        }
        __syncthreads();
        const int64_t rem = (total_quads - q0) * 4;
        const int count = (int)(rem < 1024 ? rem : 1024);
        uint4 *out = reinterpret_cast<uint4 *>(blocks) + q0 * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k * 256 + tid < count) out[k * 256 + tid] = s_vec[k * 256 + tid];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// column sums (initial centroid, VQ_Full.py:67)
// ------------------------------------------------------------------------------------------------
// Thread t of the grid owns column (t % d) and walks rows t/d, t/d + rows_per_pass, ...; a
// workgroup merges its threads' sums in LDS and issues one 64-bit atomic per column.
__global__ __launch_bounds__(256) void k_colsum_u8(const uint8_t *__restrict__ blocks, int64_t N, int d,
                                                   unsigned long long *__restrict__ sums, int64_t rows_per_pass) {
    extern __shared__ unsigned long long s_col[];
    for (int i = threadIdx.x; i < d; i += blockDim.x) s_col[i] = 0ull;
    __syncthreads();
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < rows_per_pass * d) {
        int64_t row = t / d;
        int e = (int)(t - row * d);
        unsigned long long acc = 0ull;
        for (int64_t n = row; n < N; n += rows_per_pass) acc += blocks[(size_t)n * d + e];
        if (acc) atomicAdd(&s_col[e], acc);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < d; i += blockDim.x)
        if (s_col[i]) atomicAdd(&sums[i], s_col[i]);
}

// ------------------------------------------------------------------------------------------------
// split / finalize (VQ_Full.py:71-73, :81-86)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_split(const double *__restrict__ cb, int64_t n, double fplus, double fminus,
                                               double *__restrict__ cb2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double c = cb[i];
        cb2[i] = c * fplus;
        cb2[n + i] = c * fminus;
    }
}

__global__ __launch_bounds__(256) void k_finalize(const unsigned long long *__restrict__ sums,
                                                  const unsigned long long *__restrict__ counts,
                                                  const double *cb_old, int K, int d, double *cb_new) {
    int64_t n = (int64_t)K * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long c = counts[i / d];
        double old = cb_old[i];
        cb_new[i] = c ? (double)sums[i] / (double)c : old;
    }
}

// The stop rule of VQ_Full.py:89-98 for iteration `iter` of a stage (one thread): the distortion D = dist_sum /
// n_total is recorded, and the stage stops when |prev - D| / prev < threshold.
__device__ __forceinline__ void apply_stop_rule(LbgState *state, double dist_sum, double n_total, double threshold,
                                                int iter) {
    const double D = dist_sum / n_total;
    const double prev = state->prev;
    uint32_t conv = 0;
    if (prev != __builtin_huge_val()) {
        const double change = fabs(prev - D) / prev;
        conv = change < threshold ? 1u : 0u;
    }
    state->rec[iter].distortion = D;
    state->rec[iter].converged = conv;
    state->rec[iter].valid = 1u;
    state->prev = D;
    if (conv) state->stopped_after = iter;                    // later iterations of this stage are gated off
}

// Centroid update in place plus the stop rule (VQ_Full.py:81-98), for iteration g.iter of a stage: cb[k] =
// sums[k] / counts[k] for non-empty clusters (empty ones keep their codeword); thread 0 of workgroup 0 turns
// the distance sum into the distortion D = sum / n_total, records it, and applies the rule
// |prev - D| / prev < threshold (never at the first iteration, where prev is +inf; a zero prev gives NaN
// or +inf, which never satisfies it -- as in the reference).
__global__ __launch_bounds__(256) void k_finalize_rule(const unsigned long long *__restrict__ sums,
                                                       const unsigned long long *__restrict__ counts, double *cb, int K,
                                                       int d, double n_total, double threshold, LbgState *state, Gate g) {
    if (gate_closed(g)) return;
    int64_t n = (int64_t)K * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long c = counts[i / d];
        if (c) cb[i] = (double)sums[i] / (double)c;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) apply_stop_rule(state, state->dist_sum, n_total, threshold, g.iter);
}

// Zeroes up to four regions (16-byte aligned starts, sizes multiples of 16 bytes) in one launch.
struct ZeroRegions {
    uint4 *p[4];
    size_t n16[4];                   // 16-byte pieces per region
};
__global__ __launch_bounds__(256) void k_zero_regions(const ZeroRegions z) {
    const uint4 zero = make_uint4(0, 0, 0, 0);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < z.n16[r]; i += stride) z.p[r][i] = zero;
}

__global__ void k_lbg_stage_begin(LbgState *state, int max_iterations) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        state->stopped_after = 0x7fffffff;
        state->prev = __builtin_huge_val();
        state->dist_sum = 0.0;
    }
    if (i < max_iterations) {
        state->rec[i].distortion = 0.0;
        state->rec[i].converged = 0u;
        state->rec[i].valid = 0u;
    }
}

// ------------------------------------------------------------------------------------------------
// codebook preparation for the SAD filter (one workgroup)
// ------------------------------------------------------------------------------------------------
constexpr int PREP_HASH_CAP = 4096;     // codebooks up to this size get the duplicate-row detection

__device__ __forceinline__ unsigned long long mix64(unsigned long long h, unsigned long long v) {
    h ^= v;
    h *= 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    return h;
}

// One codebook entry of the fixed-point conversion: rounds cv * scale into the 16-bit half hh of `pair`, adds
// the rounding error to err, the fixed-point value to sc and the entry's bits to the row hash.
__device__ __forceinline__ void prep_quant(double cv, int hh, double scale, double qmax, uint32_t &pair, double &err,
                                           uint32_t &sc, unsigned long long &h) {
    double t = cv * scale;
    double q = rint(t);
    if (!(q >= 0.0)) q = 0.0;          // also catches NaN
    if (q > qmax) q = qmax;
    double de = fabs(t - q);
    if (!(de == de)) de = 4.0e9;
    err += de;
    pair |= ((uint32_t)q) << (16 * hh);
    sc += (uint32_t)q;
    h = mix64(h, (unsigned long long)__double_as_longlong(cv));
}

__device__ __forceinline__ uint32_t prep_ceil_sat(double c) {       // ceil, saturated; non-finite counts as huge
    if (!(c == c) || c > 1.0e9 || c < -1.0e9) return 0x7fffffffu;
    if (c <= 0.0) return 0u;
    return (uint32_t)ceil(c);
}

// ---- codebook preparation across the chip (any K, any d <= 192) ------------------------------------------
// Four launches: k_prep_max (largest entry -> the fixed-point scale), k_prep_rows (one wave per codeword:
// fixed-point row, rounding-error bound, element sum, row hash), k_prep_rank (codebooks up to PREP_SORT_CAP
// rows: position in ascending (element sum, index) order by counting, duplicate rows, rows and transposed
// fp64 copy moved to their sorted place), k_prep_finish (position table, pad rows, header).  The header's
// cmax and emax are zeroed before the first launch and filled with atomic maxima.
__global__ __launch_bounds__(256) void k_prep_max(const double *__restrict__ cb, int64_t n, PrepHeader *hdr) {
    uint32_t lmax = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t m = prep_ceil_sat(cb[i]);
        lmax = m > lmax ? m : lmax;
    }
    lmax = wave_reduce_u32<true>(lmax);
    if ((threadIdx.x & (WAVE - 1)) == 0 && lmax) atomicMax(&hdr->cmax, lmax);
}

// largest scale 2^s (s <= 8) that keeps both the pixels (<= 255) and the largest codebook entry within clim,
// the fixed-point limit the key layout allows
__device__ __forceinline__ int prep_shift(uint32_t cmax, uint32_t clim) {
    const uint64_t top = cmax > 255u ? cmax : 255u;
    int s = 8;
    while (s > 0 && (top << s) > (uint64_t)clim) --s;
    return s;
}

// rows_out: c16u (original order, moved by k_prep_rank) for codebooks that get sorted, else c16 itself; in that
// case the transposed copy and the biases are written here too.
__global__ __launch_bounds__(256) void k_prep_rows(const double *__restrict__ cb, int K, int d, int dp2, uint32_t clim,
                                                   PrepHeader *hdr, uint32_t *__restrict__ rows_out,
                                                   uint32_t *__restrict__ sc, unsigned long long *__restrict__ hash,
                                                   double *__restrict__ cbT_plain, uint32_t *__restrict__ bias_plain) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int s = prep_shift(hdr->cmax, clim);
    const double scale = (double)(1u << s);
    const double qmax = (double)clim;
    const int nwaves = gridDim.x * 4;
    for (int k = blockIdx.x * 4 + (threadIdx.x >> 6); k < K; k += nwaves) {        // (wave-uniform)
        const double *row = cb + (size_t)k * d;
        double err = 0.0;
        uint32_t scv = 0;
        unsigned long long h = 0;
        for (int e0 = 0; e0 < 2 * dp2; e0 += WAVE) {
            const int e = e0 + lane;
            uint32_t q = 0;
            if (e < d) {
                const double cv = row[e];
                const double t = cv * scale;
                double qd = rint(t);
                if (!(qd >= 0.0)) qd = 0.0;          // also catches NaN
                if (qd > qmax) qd = qmax;
                double de = fabs(t - qd);
                if (!(de == de)) de = 4.0e9;
                err += de;
                q = (uint32_t)qd;
                scv += q;
                h += mix64(0x243F6A8885A308D3ull + (unsigned long long)e, (unsigned long long)__double_as_longlong(cv));
                if (cbT_plain) cbT_plain[(size_t)e * K + k] = cv;
            }
            const uint32_t qn = __shfl_xor(q, 1, WAVE);                              // the odd neighbour's entry
            if (!(lane & 1) && e < 2 * dp2) rows_out[(size_t)k * dp2 + (e >> 1)] = q | (qn << 16);
        }
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {                                  // the same value in every lane
            err += __shfl_xor(err, off, WAVE);
            scv += __shfl_xor(scv, off, WAVE);
            h += __shfl_xor(h, off, WAVE);
        }
        if (lane == 0) {
            h = mix64(h, 0x9E3779B97F4A7C15ull);
            sc[k] = scv;
            hash[k] = h;
            // strict integer upper bound of err (plus slack for the fp64 rounding of err itself)
            const uint32_t ek = (err < 2.0e9) ? (uint32_t)err + 2u : 0x7fffffffu;
            atomicMax(&hdr->emax, ek);
            if (bias_plain) bias_plain[k] = (uint32_t)k;
        }
    }
}

// Dynamic LDS: K element sums (uint32, padded to a multiple of 4), then K row hashes (uint64).  A workgroup
// takes PREP_RANK_ROWS codewords, 16 threads each: they split the scan that ranks the codeword and looks for the
// first earlier row with its sum and hash (verified bit by bit: identical rows give identical distances, so
// the later one can never win -- ties go to the lower index -- and gets the never-competes bit; a missed
// duplicate only costs re-checks); then each wave moves four of the rows.
constexpr int PREP_RANK_ROWS = 16;
__global__ __launch_bounds__(256) void k_prep_rank(const double *__restrict__ cb, int K, int d, int dp2,
                                                   const uint32_t *__restrict__ sc, const unsigned long long *__restrict__ hash,
                                                   const uint32_t *__restrict__ c16u, uint32_t *__restrict__ c16,
                                                   uint32_t *__restrict__ bias, double *__restrict__ cbT,
                                                   uint32_t *__restrict__ scs, uint32_t *__restrict__ c16T) {
    extern __shared__ __align__(16) unsigned long long s_prep[];
    __shared__ uint32_t s_pos[PREP_RANK_ROWS];
    const int kq = (K + 3) & ~3;
    uint32_t *s_sc = reinterpret_cast<uint32_t *>(s_prep);                      // K sums, padded to a multiple of 4
    unsigned long long *s_h = s_prep + kq / 2;                                  // K hashes
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    for (int i = tid; i < kq; i += 256) s_sc[i] = i < K ? sc[i] : 0xffffffffu;
    for (int i = tid; i < K; i += 256) s_h[i] = hash[i];
    __syncthreads();
    const int r = tid >> 4, sub = tid & 15;
    const int k = blockIdx.x * PREP_RANK_ROWS + r;
    const bool live = k < K;
    const int kc = live ? k : K - 1;
    const uint32_t me = s_sc[kc];
    const unsigned long long mh = s_h[kc];
    uint32_t pos = 0;
    int first = kc;
    {
        const uint4 *sv = reinterpret_cast<const uint4 *>(s_sc);
#pragma unroll 4
        for (int j4 = sub; j4 < kq / 4; j4 += 16) {
            const uint4 o = sv[j4];
            const int j = 4 * j4;
            pos += (o.x < me || (o.x == me && j < kc)) ? 1u : 0u;
            pos += (o.y < me || (o.y == me && j + 1 < kc)) ? 1u : 0u;
            pos += (o.z < me || (o.z == me && j + 2 < kc)) ? 1u : 0u;
            pos += (o.w < me || (o.w == me && j + 3 < kc)) ? 1u : 0u;
            if (o.x == me || o.y == me || o.z == me || o.w == me) {             // rare: look at the hashes
                const uint32_t ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (ov[t] == me && j + t < first && s_h[j + t] == mh) first = j + t;
            }
        }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            pos += __shfl_xor(pos, m, WAVE);
            const int of = __shfl_xor(first, m, WAVE);
            first = of < first ? of : first;
        }
    }
    if (live && sub == 0) {
        uint32_t dup = 0;
        if (first < k) {
            const unsigned long long *rowb = reinterpret_cast<const unsigned long long *>(cb + (size_t)k * d);
            const unsigned long long *rj = reinterpret_cast<const unsigned long long *>(cb + (size_t)first * d);
            int e = 0;
            while (e < d && rj[e] == rowb[e]) ++e;
            if (e == d) dup = 1u;
        }
        scs[pos] = me;
        bias[pos] = (uint32_t)k | (dup << 31);
        s_pos[r] = pos;
    }
    __syncthreads();
    // the fixed-point rows and the transposed fp64 copy go to their sorted place: c16[pos], cbT[e][pos]
    const int wv = tid >> 6;
#pragma unroll
    for (int t = 0; t < PREP_RANK_ROWS / 4; ++t) {
        const int rr = wv * (PREP_RANK_ROWS / 4) + t;
        const int k2 = blockIdx.x * PREP_RANK_ROWS + rr;
        if (k2 < K) {                                                           // (wave-uniform)
            const uint32_t p2 = s_pos[rr];
            for (int p = lane; p < dp2; p += WAVE) {
                const uint32_t w = c16u[(size_t)k2 * dp2 + p];
                c16[(size_t)p2 * dp2 + p] = w;
                if (c16T) c16T[(size_t)p * K + p2] = w;                        // (nullable: only large codebooks keep it)
            }
            for (int e = lane; e < d; e += WAVE) cbT[(size_t)e * K + p2] = cb[(size_t)k2 * d + e];
        }
    }
}

// One workgroup: position table over the sums a vector can have (sorted codebooks), pad rows, header.
__global__ __launch_bounds__(1024) void k_prep_finish(int K, int d, int dp2, uint32_t clim, int do_sort, PrepHeader *hdr,
                                                      const uint32_t *__restrict__ scs, uint32_t *__restrict__ c16,
                                                      uint32_t *__restrict__ bias, uint32_t *__restrict__ postab) {
    __shared__ uint32_t s_scs[PREP_SORT_CAP];
    const int tid = threadIdx.x;
    const int s = prep_shift(hdr->cmax, clim);
    if (do_sort)
        for (int i = tid; i < K; i += 1024) s_scs[i] = scs[i];
    // pad rows: zeros with the never-competes bias, so that batches of codewords may run past K
    const int kpad = (K + 7) / 8 * 8;
    for (int i = tid; i < (kpad - K) * dp2; i += 1024) c16[(size_t)K * dp2 + i] = 0u;
    for (int i = K + tid; i < kpad; i += 1024) bias[i] = 0x80000000u;
    // position table over the sums a vector can have: 0 .. d * 255 << s
    const uint32_t mtab = (uint32_t)d * (255u << s) + 1u;
    uint32_t tshift = 0;
    while ((mtab >> tshift) > (uint32_t)POSTAB_NT - 1u) ++tshift;
    const uint32_t nt = (mtab >> tshift) + 1u;                 // entries 0 .. nt cover every vector sum
    __syncthreads();
    if (do_sort) {
        for (uint32_t t = tid; t <= nt + 1u; t += 1024) {
            uint32_t lo = 0, hi = (uint32_t)K;                 // lower bound of (t << tshift) in s_scs
            const uint64_t target = (uint64_t)t << tshift;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((uint64_t)s_scs[mid] < target) lo = mid + 1; else hi = mid;
            }
            postab[t] = t == nt + 1u ? (uint32_t)K : lo;
        }
    }
    if (tid == 0) {
        hdr->shift = (uint32_t)s;
        hdr->sorted = do_sort ? 1u : 0u;
        hdr->tshift = tshift;
        hdr->nt = nt;
    }
}

// Codebook preparation for short rows and small codebooks (d <= 16, K * TPC <= 1024: the LBG stages of small
// blocks, where this launch sits on the critical path of every iteration).  Same outputs as the four-launch form above, with TPC
// adjacent threads per codeword: each holds 16 / TPC entries in registers from the single read of the
// codebook to the writes at the sorted place, partial sums are combined across the TPC lanes, and the ranking
// scan is split between them.  Global stores come last, so that nothing waits for them.
// (A device function: k_prepare_small runs it on its own, k_lbg_finish_t behind the centroid update of an LBG
// iteration, for the next one.  s_keys: the workgroup's dynamic LDS.)
template <int TPC>
__device__ __forceinline__ void prepare_small_body(const double *cb, int K, int d, int dp2, uint32_t clim, PrepHeader *hdr,
                                                   uint32_t *c16, uint32_t *bias, double *cbT, uint32_t *postab,
                                                   unsigned long long *s_keys) {
    __shared__ uint32_t s_cmax;
    __shared__ uint32_t s_emax;
    __shared__ __align__(16) uint32_t s_sc[1024 + 4];            // element sum of each fixed-point row, original order
    constexpr int EPT = 16 / TPC;                                // entries per thread
    constexpr int PPT = EPT / 2;                                 // fixed-point words per thread
    uint32_t *s_scs = reinterpret_cast<uint32_t *>(s_keys);      // the sums in ascending order (after the duplicate scan)
    const bool scan_dups = K > 1;
    const uint32_t T = scan_dups ? 2u * (uint32_t)K : 0u;        // table slots
    unsigned long long *s_rowhash = s_keys + T;                  // K entries
    uint32_t *s_first = reinterpret_cast<uint32_t *>(s_rowhash + (scan_dups ? K : 0));
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int k = tid / TPC, sub = tid % TPC;
    const bool live = k < K;
    const int kc = live ? k : K - 1;
    if (tid == 0) {
        s_cmax = 0;
        s_emax = 0;
    }
    if (tid < 4) s_sc[K + tid] = 0xffffffffu;
    for (uint32_t i = tid; i < T; i += 1024) {
        s_keys[i] = 0ull;
        s_first[i] = 0xffffffffu;
    }
    // phase A: this thread's entries; largest entry of the codebook
    double creg[EPT];
    {
        const double *row = cb + (size_t)kc * d;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int ee = sub * EPT + e;
            creg[e] = row[ee < d ? ee : d - 1];
        }
    }
    uint32_t lmax = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const uint32_t m = prep_ceil_sat(creg[e]);
        lmax = m > lmax ? m : lmax;
    }
    lmax = wave_reduce_u32<true>(lmax);
    __syncthreads();                                             // s_cmax is initialised
    if (lane == 0) atomicMax(&s_cmax, lmax);
    __syncthreads();
    const uint32_t cmax = s_cmax;
    const uint64_t top = cmax > 255u ? cmax : 255u;
    int s = 8;
    while (s > 0 && (top << s) > (uint64_t)clim) --s;
    const double scale = (double)(1u << s);
    const double qmax = (double)clim;
    // phase B: fixed-point words, rounding-error bound, element sum and row hash, combined over the TPC lanes
    uint32_t preg[PPT];
    double err = 0.0;
    uint32_t sc = 0;
    unsigned long long h = 0x243F6A8885A308D3ull + (unsigned long long)sub;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        uint32_t pair = 0;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            if (sub * EPT + 2 * p + hh < d) prep_quant(creg[2 * p + hh], hh, scale, qmax, pair, err, sc, h);
        preg[p] = pair;
    }
#pragma unroll
    for (int m = 1; m < TPC; m <<= 1) {
        err += __shfl_xor(err, m, WAVE);
        sc += __shfl_xor(sc, m, WAVE);
        h += __shfl_xor(h, m, WAVE);                             // (a sum: the same in every lane of the group)
    }
    h = mix64(h, 0x9E3779B97F4A7C15ull);
    if (h == 0ull) h = 1ull;
    // strict integer upper bound of err (plus slack for the fp64 rounding of err itself)
    uint32_t ek = (err < 2.0e9) ? (uint32_t)err + 2u : 0x7fffffffu;
    ek = wave_reduce_u32<true>(live ? ek : 0u);
    if (lane == 0) atomicMax(&s_emax, ek);
    if (live && sub == 0) {
        s_sc[k] = sc;
        // record the smallest index of each distinct row hash (0 marks an empty slot)
        if (scan_dups) {
            s_rowhash[k] = h;
            uint32_t slot = (uint32_t)(h % T);
            for (uint32_t probe = 0; probe < T; ++probe) {
                unsigned long long old = atomicCAS(&s_keys[slot], 0ull, h);
                if (old == 0ull || old == h) {
                    atomicMin(&s_first[slot], (uint32_t)k);
                    break;
                }
                slot = slot + 1 == T ? 0 : slot + 1;
            }
        }
    }
    __syncthreads();
    // phase C: row k is a duplicate when the first row with its hash is an earlier, bit-identical row
    uint32_t dup = 0;
    if (scan_dups && live && sub == 0) {
        uint32_t slot = (uint32_t)(h % T);
        uint32_t first = 0xffffffffu;
        for (uint32_t probe = 0; probe < T; ++probe) {
            unsigned long long key = s_keys[slot];
            if (key == h) {
                first = s_first[slot];
                break;
            }
            if (key == 0ull) break;
            slot = slot + 1 == T ? 0 : slot + 1;
        }
        if (first < (uint32_t)k) {
            const unsigned long long *rowb = reinterpret_cast<const unsigned long long *>(cb + (size_t)k * d);
            const unsigned long long *rj = reinterpret_cast<const unsigned long long *>(cb + (size_t)first * d);
            int e = 0;
            while (e < d && rj[e] == rowb[e]) ++e;
            if (e == d) dup = 1u;
        }
    }
    __syncthreads();                                             // the hash table is free: s_scs may be written
    // phase D: position of the row in ascending (element sum, index) order, by counting; the TPC lanes of a
    // codeword take every TPC-th group of four sums
    const uint32_t me = s_sc[kc];
    uint32_t pos = 0;
    {
        const uint4 *sv = reinterpret_cast<const uint4 *>(s_sc);
        for (int j4 = sub; j4 < (K + 3) / 4; j4 += TPC) {
            const uint4 o = sv[j4];
            const int j = 4 * j4;
            pos += (o.x < me || (o.x == me && j < kc)) ? 1u : 0u;
            pos += (o.y < me || (o.y == me && j + 1 < kc)) ? 1u : 0u;
            pos += (o.z < me || (o.z == me && j + 2 < kc)) ? 1u : 0u;
            pos += (o.w < me || (o.w == me && j + 3 < kc)) ? 1u : 0u;
        }
#pragma unroll
        for (int m = 1; m < TPC; m <<= 1) pos += __shfl_xor(pos, m, WAVE);
    }
    if (live && sub == 0) s_scs[pos] = me;
    __syncthreads();
    // position table over the sums a vector can have: 0 .. d * 255 << s
    const uint32_t mtab = (uint32_t)d * (255u << s) + 1u;
    uint32_t tshift = 0;
    while ((mtab >> tshift) > (uint32_t)POSTAB_NT - 1u) ++tshift;
    const uint32_t nt = (mtab >> tshift) + 1u;                 // entries 0 .. nt cover every vector sum
    for (uint32_t t = tid; t <= nt + 1u; t += 1024) {
        uint32_t lo = 0, hi = (uint32_t)K;                     // lower bound of (t << tshift) in s_scs
        const uint64_t target = (uint64_t)t << tshift;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint64_t)s_scs[mid] < target) lo = mid + 1; else hi = mid;
        }
        postab[t] = t == nt + 1u ? (uint32_t)K : lo;
    }
    // the fixed-point row and the transposed fp64 copy at their sorted place: c16[pos], cbT[e][pos]
    if (live) {
#pragma unroll
        for (int p = 0; p < PPT; ++p)
            if (sub * PPT + p < dp2) c16[(size_t)pos * dp2 + sub * PPT + p] = preg[p];
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (sub * EPT + e < d) cbT[(size_t)(sub * EPT + e) * K + pos] = creg[e];
        if (sub == 0) bias[pos] = (uint32_t)k | (dup << 31);
    }
    // pad rows: zeros with the never-competes bias, so that batches of codewords may run past K
    const int kpad = (K + 7) / 8 * 8;
    for (int i = tid; i < (kpad - K) * dp2; i += 1024) c16[(size_t)K * dp2 + i] = 0u;
    for (int i = K + tid; i < kpad; i += 1024) bias[i] = 0x80000000u;
    if (tid == 0) {
        hdr->shift = (uint32_t)s;
        hdr->emax = s_emax;
        hdr->cmax = cmax;
        hdr->sorted = 1u;
        hdr->tshift = tshift;
        hdr->nt = nt;
    }
}

template <int TPC>
__global__ __launch_bounds__(1024) void k_prepare_small(const double *__restrict__ cb, int K, int d, int dp2,
                                                        uint32_t clim, PrepHeader *__restrict__ hdr,
                                                        uint32_t *__restrict__ c16, uint32_t *__restrict__ bias,
                                                        double *__restrict__ cbT, uint32_t *__restrict__ postab) {
    extern __shared__ __align__(16) unsigned long long s_keys[];
    prepare_small_body<TPC>(cb, K, d, dp2, clim, hdr, c16, bias, cbT, postab, s_keys);
}

// ------------------------------------------------------------------------------------------------
// where the window sort takes its vectors from
// ------------------------------------------------------------------------------------------------
// The (N, d) block matrix in memory.
struct SrcBlocks {
    const uint8_t *p;
    int d;
    // the direct-addressing interface of k_window_sort (a vector's first byte as an offset from origin(), the
    // offset of its w-th word from there): not used for the block matrix, whose word() is one multiply already
    __device__ __forceinline__ bool fast_words() const { return false; }
    __device__ __forceinline__ const uint8_t *origin() const { return p; }
    __device__ __forceinline__ uint32_t vec_base(int64_t, bool &inside) const {
        inside = false;
        return 0u;
    }
    __device__ __forceinline__ uint32_t word_off(int) const { return 0u; }
    __device__ __forceinline__ bool quads() const { return false; }
    __device__ __forceinline__ void quad_words(int64_t, uint32_t (&)[4][4]) const {}
    __device__ __forceinline__ uint32_t byte(int64_t n, int e) const { return p[(size_t)n * d + e]; }
    // word w of vector n (rows are whole aligned words)
    __device__ __forceinline__ uint32_t word(int64_t n, int w) const {
        return reinterpret_cast<const uint32_t *>(p + (size_t)n * d)[w];
    }
    // DW aligned words of vector n (d = 4 * DW)
    template <int DW>
    __device__ __forceinline__ void row_words(int64_t n, uint32_t (&w)[DW]) const {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(p) + (size_t)n * DW;
        if (DW == 4) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src);
            w[0] = v.x; w[1 % DW] = v.y; w[2 % DW] = v.z; w[3 % DW] = v.w;
        } else if (DW == 2) {
            const uint2 v = *reinterpret_cast<const uint2 *>(src);
            w[0] = v.x; w[1 % DW] = v.y;
        } else {
#pragma unroll
            for (int i = 0; i < DW; ++i) w[i] = src[i];
        }
    }
};

// The image itself: vector n is block (n / nc, n % nc) of the image padded on the bottom / right to
// multiples of (bh, bw), element e = (i * bw + j) * C + ch (VQ_Full.py:23-33, 51-57); nothing is
// materialised, padding is resolved by clamping (edge) or zeroing the coordinate.
struct SrcImage {
    const uint8_t *img;
    int H, W, C, bh, bw, pad_zero;
    uint32_t nc, seg, d;          // blocks per block row, bytes per block row segment (bw * C), bh * seg
    int words_ok;                 // words inside the image may be read with one load (at any byte alignment: load_u32_any)
    int quads_ok;                 // 4x4 blocks of 1-byte pixels, no padding, image rows multiples of 16 aligned bytes
    int fast32;                   // every byte offset into the image fits 32 bits
    // direct addressing: blocks that need no padding are read at origin() + vec_base + word_off
    __device__ __forceinline__ bool fast_words() const { return fast32 != 0 && words_ok != 0 && (seg & 3u) == 0; }
    __device__ __forceinline__ const uint8_t *origin() const { return img; }
    __device__ __forceinline__ uint32_t vec_base(int64_t n, bool &inside) const {
        const uint32_t r = (uint32_t)n / nc, q = (uint32_t)n - r * nc;
        inside = (r + 1u) * (uint32_t)bh <= (uint32_t)H && (q + 1u) * (uint32_t)bw <= (uint32_t)W;
        return r * (uint32_t)bh * (uint32_t)(W * C) + q * seg;
    }
    __device__ __forceinline__ uint32_t word_off(int w) const {
        const uint32_t wps = seg >> 2, i = (uint32_t)w / wps, c = (uint32_t)w - i * wps;
        return i * (uint32_t)(W * C) + 4u * c;
    }
    __device__ __forceinline__ bool quads() const { return quads_ok != 0; }
    // quads_ok: the four consecutive vectors n0 .. n0 + 3 (n0 a multiple of 4: four horizontally adjacent blocks)
    // from four 16-byte loads, one per pixel row; w[m] = the four row words of vector n0 + m
    __device__ __forceinline__ void quad_words(int64_t n0, uint32_t (&w)[4][4]) const {
        const uint32_t r = (uint32_t)n0 / nc, q = (uint32_t)n0 - r * nc;
        const uint8_t *base = img + ((size_t)r * 4) * (size_t)W + (size_t)q * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 v = *reinterpret_cast<const uint4 *>(base + (size_t)i * W);
            w[0][i] = v.x; w[1][i] = v.y; w[2][i] = v.z; w[3][i] = v.w;
        }
    }
    __device__ __forceinline__ uint32_t byte_at(uint32_t r, uint32_t q, uint32_t e) const {
        const uint32_t i = e / seg, rem = e - i * seg;
        const uint32_t j = rem / (uint32_t)C, ch = rem - j * (uint32_t)C;
        const int y = (int)(r * (uint32_t)bh + i), x = (int)(q * (uint32_t)bw + j);
        const int yc = y < H ? y : H - 1, xc = x < W ? x : W - 1;
        uint32_t v = img[((size_t)yc * W + xc) * C + ch];
        if (pad_zero && (y >= H || x >= W)) v = 0;
        return v;
    }
    __device__ __forceinline__ uint32_t byte(int64_t n, int e) const {
        const uint32_t r = (uint32_t)n / nc, q = (uint32_t)n - r * nc;
        return byte_at(r, q, (uint32_t)e);
    }
    // bytes e .. e + 3 of block (r, q) as one word
    __device__ __forceinline__ uint32_t word_at(uint32_t r, uint32_t q, uint32_t e) const {
        const uint32_t i = e / seg, rem = e - i * seg;
        const int y = (int)(r * (uint32_t)bh + i);
        const int yc = y < H ? y : H - 1;
        const int64_t rowbytes = (int64_t)W * C;
        const int64_t xb = (int64_t)q * seg + rem;           // byte offset inside the (padded) image row
        if (words_ok && (seg & 3u) == 0 && xb + 4 <= rowbytes && (y < H || !pad_zero))
            return load_u32_any(img + (size_t)yc * rowbytes + xb);
        uint32_t v = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) v |= byte_at(r, q, e + t) << (8 * t);
        return v;
    }
    __device__ __forceinline__ uint32_t word(int64_t n, int w) const {
        const uint32_t r = (uint32_t)n / nc, q = (uint32_t)n - r * nc;
        return word_at(r, q, 4u * (uint32_t)w);
    }
    template <int DW>
    __device__ __forceinline__ void row_words(int64_t n, uint32_t (&w)[DW]) const {
        const uint32_t r = (uint32_t)n / nc, q = (uint32_t)n - r * nc;
#pragma unroll
        for (int k = 0; k < DW; ++k) w[k] = word_at(r, q, 4u * k);
    }
};

// ------------------------------------------------------------------------------------------------
// windowed sort of the block vectors by element sum (input of the pruned search)
// ------------------------------------------------------------------------------------------------
// One workgroup per window of SORT_WIN vectors: a counting sort on the element sum (bucketed) places
// every vector of the window at its sorted position in `rows` and records where it came from in
// `perm` (position inside the window).  Vectors never leave their window, so everything the search
// later writes per vector (label, distance) lands within 32 / 64 KB of its neighbours.  The order
// inside a bucket depends on the order the atomics retire in; nothing computed from the sorted copy
// depends on it.  This kernel serves rows of any length: every thread sums its vectors and takes their
// ranks, the inverse map (sorted slot -> origin) goes to LDS, and then the whole workgroup copies the
// window in OUTPUT order -- consecutive lanes write consecutive words (WORDS: d % 4 == 0 and the source
// delivers aligned words) or bytes of consecutive sorted rows, so the stores are coalesced and the source
// rows, just read for their sums, come from the cache.  Rows of one to four words take
// k_window_sort_lds instead.
template <class SRC, bool WORDS>
__global__ __launch_bounds__(1024) void k_window_sort(const SRC src, int64_t N, int d, int bshift,
                                                      uint8_t *__restrict__ rows, uint16_t *__restrict__ perm,
                                                      SortedHeader *__restrict__ shdr, int copy_rows, int sums_given) {
    constexpr int T = 1024;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        shdr->magic = SORTED_MAGIC;
        shdr->d = (uint32_t)d;
        shdr->N = (unsigned long long)N;
    }
    constexpr int PER = SORT_WIN / T;                       // vectors per thread
    __shared__ uint32_t s_cnt[SORT_BUCKETS];
    __shared__ uint32_t s_wsum[T / WAVE];
    __shared__ uint16_t s_inv[SORT_WIN];                    // sorted slot -> position in the window it came from
    // WORDS with a directly addressable source: where each vector of the window starts (0xffffffff: it needs
    // padding and goes through src.word) and where its words lie from there -- the copy below then costs a
    // few instructions per word instead of the index arithmetic of a block lookup
    __shared__ uint32_t s_base[WORDS ? SORT_WIN : 1];
    __shared__ uint32_t s_woff[WORDS ? MAX_D_FAST / 4 : 1];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
    const int64_t w0 = (int64_t)blockIdx.x * SORT_WIN;
    const int count = (int)((N - w0) < SORT_WIN ? (N - w0) : SORT_WIN);      // vectors in this window
    const int dw = d >> 2;
    const bool direct = WORDS && src.fast_words();
    const uint8_t *org = src.origin();
    for (int i = tid; i < SORT_BUCKETS; i += T) s_cnt[i] = 0u;
    if (WORDS && direct)
        for (int w = tid; w < dw; w += T) s_woff[w] = src.word_off(w);
    __syncthreads();
    uint32_t bucket[PER], rank[PER];
    if (sums_given) {
        // (uniform) k_sort_sums left every vector's element sum in its `perm` entry; the origin map replaces them below
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int li = j * T + tid;
            bucket[j] = 0;
            rank[j] = 0;
            if (li < count) {
                if (WORDS && direct) {
                    bool inside = false;
                    const uint32_t b = src.vec_base(w0 + li, inside);
                    s_base[li] = inside ? b : 0xffffffffu;
                }
                bucket[j] = (uint32_t)perm[w0 + li] >> bshift;
                rank[j] = atomicAdd(&s_cnt[bucket[j]], 1u);
            }
        }
    } else if (WORDS && direct) {
        // the PER vectors of a thread side by side: one trip of the word loop has PER independent loads in flight
        // (a vector that needs padding, or lies beyond the window, reads the image's first word instead and is
        // summed through src.word afterwards)
        uint32_t base[PER], sum[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int li = j * T + tid;
            uint32_t b = 0xffffffffu;
            if (li < count) {
                bool inside = false;
                b = src.vec_base(w0 + li, inside);
                b = inside ? b : 0xffffffffu;
                s_base[li] = b;
            }
            base[j] = b;
            sum[j] = 0;
        }
#pragma unroll 4
        for (int w = 0; w < dw; ++w) {                      // (four trips' loads in flight: long rows wait on latency)
            const uint32_t off = s_woff[w];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t at = base[j] != 0xffffffffu ? base[j] + off : 0u;
                sum[j] = __builtin_amdgcn_sad_u8(load_u32_any(org + at), 0u, sum[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int li = j * T + tid;
            bucket[j] = 0;
            rank[j] = 0;
            if (li < count) {
                if (base[j] == 0xffffffffu) {
                    sum[j] = 0;
                    for (int w = 0; w < dw; ++w) sum[j] = __builtin_amdgcn_sad_u8(src.word(w0 + li, w), 0u, sum[j]);
                }
                bucket[j] = sum[j] >> bshift;
                rank[j] = atomicAdd(&s_cnt[bucket[j]], 1u);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int li = j * T + tid;
            bucket[j] = 0;
            rank[j] = 0;
            if (li < count) {
                uint32_t sum = 0;
                if (WORDS) {
                    for (int w = 0; w < dw; ++w) sum = __builtin_amdgcn_sad_u8(src.word(w0 + li, w), 0u, sum);
                } else {
                    for (int e = 0; e < d; ++e) sum += src.byte(w0 + li, e);
                }
                bucket[j] = sum >> bshift;
                rank[j] = atomicAdd(&s_cnt[bucket[j]], 1u);
            }
        }
    }
    __syncthreads();
    // exclusive scan of the bucket counts: two buckets per thread, wave scan, then the wave totals
    const uint32_t c0 = s_cnt[2 * tid], c1 = s_cnt[2 * tid + 1];
    uint32_t incl = c0 + c1;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, WAVE);
        if (lane >= off) incl += o;
    }
    if (lane == WAVE - 1) s_wsum[wv] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int i = 0; i < wv; ++i) wbase += s_wsum[i];
    const uint32_t excl = wbase + incl - (c0 + c1);
    __syncthreads();
    s_cnt[2 * tid] = excl;
    s_cnt[2 * tid + 1] = excl + c0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int li = j * T + tid;
        if (li < count) s_inv[s_cnt[bucket[j]] + rank[j]] = (uint16_t)li;          // slot < count
    }
    __syncthreads();
    for (int i = tid; i < count; i += T) perm[w0 + i] = s_inv[i];
    if (!copy_rows) return;                                 // (uniform) k_sort_gather copies the rows
    if (WORDS) {
        // U words per thread and trip: their slot look-ups, their loads and their stores in groups, so that U
        // loads are in flight at once (one word per trip leaves the copy waiting on a chain of LDS and cache
        // latencies per word)
        uint32_t *out = reinterpret_cast<uint32_t *>(rows) + (size_t)w0 * dw;
        const int total = count * dw;
        constexpr int U = 8;
        const unsigned long long magic = ((1ull << 32) + (unsigned)dw - 1ull) / (unsigned)dw;   // i / dw for i * dw < 2^32
        for (int i0 = tid; i0 < total; i0 += T * U) {
            uint32_t val[U];
            int li[U], ww[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * T;
                const int ic = i < total ? i : 0;
                const int slot = (int)(((unsigned long long)(unsigned)ic * magic) >> 32);
                ww[u] = ic - slot * dw;
                li[u] = s_inv[slot];
            }
            if (direct) {
                uint32_t base[U];
#pragma unroll
                for (int u = 0; u < U; ++u) base[u] = s_base[li[u]];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t at = base[u] != 0xffffffffu ? base[u] + s_woff[ww[u]] : 0u;
                    val[u] = load_u32_any(org + at);
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (base[u] == 0xffffffffu) val[u] = src.word(w0 + (int64_t)li[u], ww[u]);
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) val[u] = src.word(w0 + (int64_t)li[u], ww[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * T;
                if (i < total) out[i] = val[u];
            }
        }
    } else {
        uint8_t *out = rows + (size_t)w0 * d;
        const int total = count * d;
        for (int i = tid; i < total; i += T) {
            const int slot = i / d, e = i - slot * d;
            out[i] = (uint8_t)src.byte(w0 + (int64_t)s_inv[slot], e);
        }
    }
}

// First launch of the generic window sort when the windows are too few to fill the chip: the element sums of the
// vectors, `parts` workgroups per window, left in the vectors' `perm` entries (a sum is at most 255 * 192 < 2^16)
// for k_window_sort (sums_given = 1), which ranks them and overwrites the entries with the origin map.
template <class SRC, bool WORDS>
__global__ __launch_bounds__(1024) void k_sort_sums(const SRC src, int64_t N, int d, int parts, uint16_t *__restrict__ perm) {
    constexpr int T = 1024;
    __shared__ uint32_t s_woff[WORDS ? MAX_D_FAST / 4 : 1];
    const int tid = threadIdx.x;
    const int64_t win = (int64_t)(blockIdx.x / (unsigned)parts);
    const int part = (int)(blockIdx.x % (unsigned)parts);
    const int64_t w0 = win * SORT_WIN;
    const int count = (int)((N - w0) < SORT_WIN ? (N - w0) : SORT_WIN);
    const int per = (count + parts - 1) / parts;            // vectors per part
    const int l0 = part * per < count ? part * per : count;
    const int l1 = l0 + per < count ? l0 + per : count;
    const int dw = d >> 2;
    const bool direct = WORDS && src.fast_words();
    const uint8_t *org = src.origin();
    if (WORDS && direct)
        for (int w = tid; w < dw; w += T) s_woff[w] = src.word_off(w);
    __syncthreads();
    for (int li = l0 + tid; li < l1; li += T) {
        uint32_t sum = 0;
        if (WORDS) {
            bool inside = false;
            const uint32_t base = direct ? src.vec_base(w0 + li, inside) : 0u;
            if (inside) {
#pragma unroll 8
                for (int w = 0; w < dw; ++w)
                    sum = __builtin_amdgcn_sad_u8(load_u32_any(org + (uint32_t)(base + s_woff[w])), 0u, sum);
            } else {
                for (int w = 0; w < dw; ++w) sum = __builtin_amdgcn_sad_u8(src.word(w0 + li, w), 0u, sum);
            }
        } else {
            for (int e = 0; e < d; ++e) sum += src.byte(w0 + li, e);
        }
        perm[w0 + li] = (uint16_t)sum;
    }
}

// Second launch of the generic window sort when the windows are too few to fill the chip (a workgroup per window
// leaves most compute units idle and the copy waits on one unit's load latency): `parts` workgroups per window copy
// the rows in output order from the origin map that k_window_sort (copy_rows = 0) left in `perm`.
template <class SRC, bool WORDS>
__global__ __launch_bounds__(1024) void k_sort_gather(const SRC src, int64_t N, int d, int parts,
                                                      uint8_t *__restrict__ rows, const uint16_t *__restrict__ perm) {
    constexpr int T = 1024;
    __shared__ uint16_t s_li[SORT_WIN];                     // slot - s0 -> position in the window it came from
    __shared__ uint32_t s_base[WORDS ? SORT_WIN : 1];       // slot - s0 -> first byte (0xffffffff: needs padding)
    __shared__ uint32_t s_woff[WORDS ? MAX_D_FAST / 4 : 1];
    const int tid = threadIdx.x;
    const int64_t win = (int64_t)(blockIdx.x / (unsigned)parts);
    const int part = (int)(blockIdx.x % (unsigned)parts);
    const int64_t w0 = win * SORT_WIN;
    const int count = (int)((N - w0) < SORT_WIN ? (N - w0) : SORT_WIN);
    const int per = (count + parts - 1) / parts;            // sorted slots per part
    const int s0 = part * per < count ? part * per : count;
    const int s1 = s0 + per < count ? s0 + per : count;
    const int dw = d >> 2;
    const bool direct = WORDS && src.fast_words();
    const uint8_t *org = src.origin();
    for (int sl = s0 + tid; sl < s1; sl += T) {
        const int li = perm[w0 + sl];
        s_li[sl - s0] = (uint16_t)li;
        if (WORDS && direct) {
            bool inside = false;
            const uint32_t b = src.vec_base(w0 + li, inside);
            s_base[sl - s0] = inside ? b : 0xffffffffu;
        }
    }
    if (WORDS && direct)
        for (int w = tid; w < dw; w += T) s_woff[w] = src.word_off(w);
    __syncthreads();
    if (WORDS) {
        uint32_t *out = reinterpret_cast<uint32_t *>(rows) + ((size_t)w0 + (size_t)s0) * dw;
        const int total = (s1 - s0) * dw;
        constexpr int U = 8;
        const unsigned long long magic = ((1ull << 32) + (unsigned)dw - 1ull) / (unsigned)dw;   // i / dw for i * dw < 2^32
        for (int i0 = tid; i0 < total; i0 += T * U) {
            uint32_t val[U];
            int rel[U], ww[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * T;
                const int ic = i < total ? i : 0;
                rel[u] = (int)(((unsigned long long)(unsigned)ic * magic) >> 32);
                ww[u] = ic - rel[u] * dw;
            }
            if (direct) {
                uint32_t base[U];
#pragma unroll
                for (int u = 0; u < U; ++u) base[u] = s_base[rel[u]];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t at = base[u] != 0xffffffffu ? base[u] + s_woff[ww[u]] : 0u;
                    val[u] = load_u32_any(org + at);
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (base[u] == 0xffffffffu) val[u] = src.word(w0 + (int64_t)s_li[rel[u]], ww[u]);
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) val[u] = src.word(w0 + (int64_t)s_li[rel[u]], ww[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * T;
                if (i < total) out[i] = val[u];
            }
        }
    } else {
        uint8_t *out = rows + ((size_t)w0 + (size_t)s0) * d;
        const int total = (s1 - s0) * d;
        for (int i = tid; i < total; i += T) {
            const int rel = i / d, e = i - rel * d;
            out[i] = (uint8_t)src.byte(w0 + (int64_t)s_li[rel], e);
        }
    }
}

// The same sort for rows of DW 32-bit words (d = 4 * DW <= 16, word-aligned): the window's rows are
// placed at their sorted positions in LDS (8192 * d bytes) and written out in order, so that every
// global access of the kernel is coalesced (a scattered store costs one L2 request per lane).
// Dynamic LDS: [SORT_WIN * DW words of rows][SORT_WIN uint16 origins].
template <int DW, class SRC>
__global__ __launch_bounds__(1024) void k_window_sort_lds(const SRC src, int64_t N, int bshift,
                                                          uint32_t *__restrict__ rows, uint16_t *__restrict__ perm,
                                                          SortedHeader *__restrict__ shdr) {
    constexpr int T = 1024;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        shdr->magic = SORTED_MAGIC;
        shdr->d = (uint32_t)(4 * DW);
        shdr->N = (unsigned long long)N;
    }
    constexpr int PER = SORT_WIN / T;                       // vectors per thread
    extern __shared__ __align__(16) uint32_t s_rows[];
    uint16_t *s_perm = reinterpret_cast<uint16_t *>(s_rows + (size_t)SORT_WIN * DW);
    __shared__ uint32_t s_cnt[SORT_BUCKETS];
    __shared__ uint32_t s_wsum[T / WAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
    const int64_t nwin = (N + SORT_WIN - 1) / SORT_WIN;
    // A workgroup walks windows blockIdx.x, + gridDim.x, ...: the rows of its NEXT window are requested before
    // the current one is written out, so that the loads of one window travel while the stores of the other do.
    uint32_t keep[PER][DW], nxt[PER][DW];
    // position inside the window of this thread's j-th vector: strided by the workgroup, or -- when the source
    // delivers four adjacent blocks from four 16-byte loads -- runs of four consecutive vectors
    const bool quads = DW == 4 && src.quads();
    auto li_of = [&](int j) { return quads ? ((j >> 2) * T + tid) * 4 + (j & 3) : j * T + tid; };
    auto load_window = [&](int64_t win, uint32_t (&dst)[PER][DW]) {
        const int64_t w0 = win * SORT_WIN;
        const int count = (int)((N - w0) < SORT_WIN ? (N - w0) : SORT_WIN);
        if constexpr (DW == 4) {
            if (quads) {                                    // (uniform) N is a multiple of 4 here
#pragma unroll
                for (int g = 0; g < PER / 4; ++g) {
                    const int li = (g * T + tid) * 4;
                    const int lc = li < count ? li : count - 4;
                    uint32_t w[4][4];
                    src.quad_words(w0 + lc, w);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) dst[g * 4 + m][i] = w[m][i];
                }
                return;
            }
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int li = j * T + tid;
            const int lc = li < count ? li : count - 1;     // clamped: loads stay in bounds
            src.template row_words<DW>(w0 + lc, dst[j]);
        }
    };
    if ((int64_t)blockIdx.x < nwin) load_window(blockIdx.x, keep);
    for (int64_t win = blockIdx.x; win < nwin; win += gridDim.x) {
        const int64_t w0 = win * SORT_WIN;
        const int count = (int)((N - w0) < SORT_WIN ? (N - w0) : SORT_WIN);  // vectors in this window
        for (int i = tid; i < SORT_BUCKETS; i += T) s_cnt[i] = 0u;
        __syncthreads();
        uint32_t bucket[PER], rank[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int li = li_of(j);
            uint32_t sum = 0;
#pragma unroll
            for (int w = 0; w < DW; ++w) sum = __builtin_amdgcn_sad_u8(keep[j][w], 0u, sum);
            bucket[j] = sum >> bshift;
            rank[j] = li < count ? atomicAdd(&s_cnt[bucket[j]], 1u) : 0u;
        }
        __syncthreads();
        // exclusive scan of the bucket counts: two buckets per thread, wave scan, then the wave totals
        const uint32_t c0 = s_cnt[2 * tid], c1 = s_cnt[2 * tid + 1];
        uint32_t incl = c0 + c1;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, WAVE);
            if (lane >= off) incl += o;
        }
        if (lane == WAVE - 1) s_wsum[wv] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int i = 0; i < wv; ++i) wbase += s_wsum[i];
        const uint32_t excl = wbase + incl - (c0 + c1);
        __syncthreads();
        s_cnt[2 * tid] = excl;
        s_cnt[2 * tid + 1] = excl + c0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int li = li_of(j);
            if (li < count) {
                const uint32_t slot = s_cnt[bucket[j]] + rank[j];            // < count
#pragma unroll
                for (int w = 0; w < DW; ++w) s_rows[(size_t)slot * DW + w] = keep[j][w];
                s_perm[slot] = (uint16_t)li;
            }
        }
        // the next window's rows start travelling now
        const int64_t next = win + gridDim.x;
        if (next < nwin) load_window(next, nxt);
        __syncthreads();
        // write the window out in order, 16 bytes per lane where the window allows it
        uint32_t *orow = rows + (size_t)w0 * DW;
        const int words = count * DW;
        if ((words & 3) == 0) {
            const uint4 *s4 = reinterpret_cast<const uint4 *>(s_rows);
            uint4 *o4 = reinterpret_cast<uint4 *>(orow);           // rows is 16-byte aligned, w0 * DW * 4 a multiple of 16
            for (int i = tid; i < words / 4; i += T) o4[i] = s4[i];
        } else {
            for (int i = tid; i < words; i += T) orow[i] = s_rows[i];
        }
        uint16_t *operm = perm + w0;
        if ((count & 7) == 0 && (reinterpret_cast<uintptr_t>(operm) & 15) == 0) {
            const uint4 *sp = reinterpret_cast<const uint4 *>(s_perm);
            uint4 *op = reinterpret_cast<uint4 *>(operm);
            for (int i = tid; i < count / 8; i += T) op[i] = sp[i];
        } else {
            for (int i = tid; i < count; i += T) operm[i] = s_perm[i];
        }
        __syncthreads();                                    // the LDS image is free for the next window
#pragma unroll
        for (int j = 0; j < PER; ++j)
#pragma unroll
            for (int w = 0; w < DW; ++w) keep[j][w] = nxt[j][w];
    }
}

// The same sort for rows of 5 .. 12 words (RGB 4x4 blocks: 48 bytes).  A window of such rows (up to 384 KB)
// is beyond the LDS, and copying it in output order straight from the source gathers 12-byte pieces out of
// 128-byte cache lines: the caches move ten times the bytes they deliver.  Here every thread keeps its 16 rows
// in registers from one coalesced read (512 threads, 16 * DW registers of rows), and the window leaves through
// the LDS in chunks of SORT_CHUNK sorted slots: the rows of a chunk are placed at their sorted positions and
// written out in order with 16-byte stores.  Dynamic LDS: SORT_CHUNK * DW words.  (Reading an RGB pixel row
// of a block with one 12-byte load was measured slower: the register triples it needs push rows into scratch.)
constexpr int SORT_CHUNK = 2048;
template <int DW, class SRC>
__global__ __launch_bounds__(512) void k_window_sort_regs(const SRC src, int64_t N, int bshift,
                                                          uint32_t *__restrict__ rows, uint16_t *__restrict__ perm,
                                                          SortedHeader *__restrict__ shdr) {
    constexpr int T = 512;
    constexpr int PER = SORT_WIN / T;                       // vectors per thread
    constexpr int BPT = SORT_BUCKETS / T;                   // buckets per thread in the scan
    static_assert(SORT_WIN % T == 0 && SORT_BUCKETS % T == 0 && SORT_WIN % SORT_CHUNK == 0 && SORT_BUCKETS <= 65536 &&
                  SORT_WIN <= 65536, "window sort geometry");
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        shdr->magic = SORTED_MAGIC;
        shdr->d = (uint32_t)(4 * DW);
        shdr->N = (unsigned long long)N;
    }
    extern __shared__ __align__(16) uint32_t s_chunk[];
    __shared__ uint32_t s_cnt[SORT_BUCKETS];
    __shared__ uint32_t s_wsum[T / WAVE];
    __shared__ uint32_t s_woff[DW];
    __shared__ __align__(16) uint16_t s_perm[SORT_WIN];     // sorted slot -> origin (before that: sums of padded vectors)
    __shared__ uint16_t s_padslot[SORT_WIN];                // sorted slot of each vector that needs padding
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
    const int64_t w0 = (int64_t)blockIdx.x * SORT_WIN;
    const int count = (int)((N - w0) < SORT_WIN ? (N - w0) : SORT_WIN);      // vectors in this window (>= 1)
    // The image source is read by direct addressing only (the host sends other images to k_window_sort); the
    // few vectors that need padding never enter the registers: their sums and, later, their rows go through
    // src.word in one rolled loop each (what they need from one phase to the next waits in the LDS), so that
    // `keep` and `slot` are only ever indexed by constants and the padding code exists once.
    constexpr bool DIRECT = !std::is_same<SRC, SrcBlocks>::value;
    const uint8_t *org = src.origin();
    for (int i = tid; i < SORT_BUCKETS; i += T) s_cnt[i] = 0u;
    if (DIRECT && tid < DW) s_woff[tid] = src.word_off(tid);
    __syncthreads();
    uint32_t keep[PER][DW];
    uint32_t padmask = 0;                                   // bit j: vector j of this thread needs padding
    if constexpr (DIRECT) {
        constexpr int GRP = PER / 4;                        // (in four groups: fewer addresses alive beside the rows)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            uint32_t base[GRP];
#pragma unroll
            for (int jj = 0; jj < GRP; ++jj) {
                const int j = h * GRP + jj;
                const int li = j * T + tid;
                uint32_t b = 0u;                            // beyond the window: the image's first block, unused
                if (li < count) {
                    bool inside = false;
                    b = src.vec_base(w0 + li, inside);
                    if (!inside) {
                        b = 0u;
                        padmask |= 1u << j;
                    }
                }
                base[jj] = b;
            }
#pragma unroll
            for (int w = 0; w < DW; ++w) {
                const uint32_t off = s_woff[w];
#pragma unroll
                for (int jj = 0; jj < GRP; ++jj)
                    keep[h * GRP + jj][w] = load_u32_any(org + (uint32_t)(base[jj] + off));
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int li = j * T + tid;
            const int lc = li < count ? li : count - 1;     // clamped: loads stay in bounds
            src.template row_words<DW>(w0 + lc, keep[j]);
        }
    }
    if (DIRECT && padmask != 0u) {
#pragma nounroll
        for (int j = 0; j < PER; ++j) {
            if (padmask >> j & 1u) {
                const int li = j * T + tid;
                uint32_t sum = 0;
                for (int w = 0; w < DW; ++w) sum = __builtin_amdgcn_sad_u8(src.word(w0 + li, w), 0u, sum);
                s_perm[li] = (uint16_t)sum;                 // <= 255 * 4 * DW; read back by this thread below
            }
        }
    }
    uint32_t slot[PER];                                     // first bucket | rank in the bucket << 16, then the sorted position
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int li = j * T + tid;
        uint32_t sum = 0;
#pragma unroll
        for (int w = 0; w < DW; ++w) sum = __builtin_amdgcn_sad_u8(keep[j][w], 0u, sum);
        if (DIRECT && (padmask >> j & 1u)) sum = s_perm[li];
        const uint32_t bucket = sum >> bshift;              // < SORT_BUCKETS
        slot[j] = bucket | (li < count ? atomicAdd(&s_cnt[bucket], 1u) : 0u) << 16;
    }
    __syncthreads();
    // exclusive scan of the bucket counts: BPT buckets per thread, wave scan, then the wave totals
    uint32_t c[BPT], incl = 0;
#pragma unroll
    for (int b = 0; b < BPT; ++b) {
        c[b] = s_cnt[BPT * tid + b];
        incl += c[b];
    }
    const uint32_t mine = incl;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, WAVE);
        if (lane >= off) incl += o;
    }
    if (lane == WAVE - 1) s_wsum[wv] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (int i = 0; i < wv; ++i) run += s_wsum[i];
#pragma unroll
    for (int b = 0; b < BPT; ++b) {
        s_cnt[BPT * tid + b] = run;
        run += c[b];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int li = j * T + tid;
        if (li < count) {
            slot[j] = s_cnt[slot[j] & 0xffffu] + (slot[j] >> 16);           // < count
            s_perm[slot[j]] = (uint16_t)li;
            if (DIRECT && (padmask >> j & 1u)) {
                s_padslot[li] = (uint16_t)slot[j];
                slot[j] = 0xffffffffu;                      // placed by the padding loop of its chunk
            }
        } else {
            slot[j] = 0xffffffffu;                          // in no chunk
        }
    }
    for (int cs = 0; cs < count; cs += SORT_CHUNK) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t rel = slot[j] - (uint32_t)cs;
            if (rel < (uint32_t)SORT_CHUNK) {
                if constexpr (DW % 4 == 0) {
                    uint4 *dst = reinterpret_cast<uint4 *>(s_chunk + (size_t)rel * DW);
#pragma unroll
                    for (int q = 0; q < DW / 4; ++q)
                        dst[q] = make_uint4(keep[j][4 * q], keep[j][4 * q + 1], keep[j][4 * q + 2], keep[j][4 * q + 3]);
                } else {
#pragma unroll
                    for (int w = 0; w < DW; ++w) s_chunk[(size_t)rel * DW + w] = keep[j][w];
                }
            }
        }
        if (DIRECT && padmask != 0u) {
#pragma nounroll
            for (int j = 0; j < PER; ++j) {
                if (padmask >> j & 1u) {
                    const int li = j * T + tid;
                    const uint32_t rel = (uint32_t)s_padslot[li] - (uint32_t)cs;
                    if (rel < (uint32_t)SORT_CHUNK)
                        for (int w = 0; w < DW; ++w) s_chunk[(size_t)rel * DW + w] = src.word(w0 + li, w);
                }
            }
        }
        __syncthreads();                                    // (the first trip also completes s_perm)
        const int n = (count - cs) < SORT_CHUNK ? (count - cs) : SORT_CHUNK;
        const int words = n * DW;
        uint32_t *orow = rows + ((size_t)w0 + (size_t)cs) * DW;     // 16-byte aligned: w0 and cs are multiples of 4 rows
        if ((words & 3) == 0) {
            const uint4 *s4 = reinterpret_cast<const uint4 *>(s_chunk);
            uint4 *o4 = reinterpret_cast<uint4 *>(orow);
            for (int i = tid; i < words / 4; i += T) o4[i] = s4[i];
        } else {
            for (int i = tid; i < words; i += T) orow[i] = s_chunk[i];
        }
        if (cs == 0) {
            uint16_t *operm = perm + w0;
            if ((count & 7) == 0 && (reinterpret_cast<uintptr_t>(operm) & 15) == 0) {
                const uint4 *sp = reinterpret_cast<const uint4 *>(s_perm);
                uint4 *op = reinterpret_cast<uint4 *>(operm);
                for (int i = tid; i < count / 8; i += T) op[i] = sp[i];
            } else {
                for (int i = tid; i < count; i += T) operm[i] = s_perm[i];
            }
        }
        __syncthreads();                                    // the chunk is free for the next slots
    }
}

// ------------------------------------------------------------------------------------------------
// nearest codeword, FAST mode (VQ_Full.py:77-78 / :124-125), optionally fused with the
// per-cluster accumulation of VQ_Full.py:81-90
// ------------------------------------------------------------------------------------------------
constexpr int KEY_PACKED = 0;   // key = (S << kbits) + (k | dupbit): needs S << kbits < 2^31
constexpr int KEY_SPLIT = 1;    // value and index tracked separately

struct AssignArgs {
    const uint8_t *blocks;
    int64_t N;
    int d;
    const double *cb;           // (K, d) row-major fp64
    const double *cbT;          // transposed fp64 copy, columns in the order of the fixed-point rows
    int K;
    int kbits;
    const PrepHeader *hdr;
    int32_t *labels;            // N; written for unambiguous vectors only
    double *min_dist;           // nullable; written for unambiguous vectors only
    uint32_t *queue;            // NQ sub-queues of qcap entries: ambiguous vectors, in no particular order (their
                                // index in `blocks`, or with PRUNE their position in the sorted copy)
    uint32_t *qcount;           // entries per sub-queue, counter q at qcount[16 * q] (zeroed before the launch);
                                // the 64-bit word at qcount[16 * q + 2] collects the pairs evaluated (statistics)
    uint32_t qcap;
    unsigned long long *acc_rep; // ACC: NREP replicas of the [K][d+1] table of [sums | count], zeroed before the launch
    int vec4;                   // how a vector is read: 0 byte by byte, 1 word by word, 2 in 16-byte pieces
                                // (row_load_mode)
    int64_t ntiles;
    // PRUNE: the search reads the window-sorted copy of the vectors and skips codewords by element sum
    const uint8_t *rows;        // N * d bytes: the vectors, each window of SORT_WIN sorted by element sum
    const uint16_t *perm;       // N: position inside its window of the vector now at sorted position i
    const uint32_t *postab;     // sum threshold -> position in the sorted codebook
    const SortedHeader *shdr;   // what the sorted copy holds (verified before the origin map is trusted)
    int64_t dist_sorted_end;    // PRUNE: the distances of the vectors at sorted positions below this go to min_dist at
                                // their SORTED position (coalesced stores; the reader un-permutes), the others by origin
    int cb_lds;                 // ACC: 1 = the fp64 codebook is staged in LDS (rows of d + 1 doubles) for the winners' distances
    int tab_reps;               // ACC: copies of the LDS table (a power of two; see the kernel)
    int persist_tiles;          // ACC: > 0 = resident workgroups; every XCD's waves share this many tiles of 64 * R vectors
    int first_rows, round_rows; // PRUNE, scalar-load codewords: rows of the first round / added per side and later round (multiples of 8)
    int heavy_first;            // PRUNE: 1 = tiles from the middle of the sort windows are dispatched first (few-round launches)
    Gate gate;
};

// Whether a filter variant settles near-ties inside the kernel (see RESOLVE in k_assign_fast).
constexpr bool resolves_in_kernel(int dp2, bool lds_cw, bool prune) { return prune && !lds_cw && dp2 <= 32; }

// Codewords per LDS tile of the LDS_CW variant (about 4 KB of fixed-point rows per tile).
constexpr int tile_k_for(int dp2) { return dp2 <= 8 ? 128 : (dp2 <= 16 ? 64 : (dp2 <= 32 ? 32 : (dp2 <= 64 ? 16 : 8))); }

// LDS_CW = false: codewords come through scalar loads (small codebooks that live in the scalar
// cache).  LDS_CW = true: the workgroup stages the fixed-point codebook through LDS in
// double-buffered tiles (wide vector loads, one barrier per tile) and every wave reads the rows
// back as broadcast LDS reads; this serves large dimensions (a row no longer fits the scalar
// registers) and codebooks far beyond the scalar cache.
//
// PRUNE (scalar-load variant only): the vectors are read from a copy in which every window of
// SORT_WIN vectors is sorted by element sum, so the 64 * R vectors of a wave have nearly the same
// sum, and the fixed-point codebook rows are in ascending order of THEIR element sum.  Since
// sum_e |X_e - C_e| >= |sum_e X_e - sum_e C_e| holds exactly in integers, a codeword whose sum lies
// further from every vector sum of the wave than the wave's worst current best distance plus the
// ambiguity window can be neither the nearest nor a near-tie for any of them: the wave searches a
// contiguous range of the sorted codebook around its own sum, widening it in steps and narrowing
// the bound as the best distances improve.  Results are identical to the full search.
// Whether the scalar loads of the 16-byte-row pruned search run one pair of rows ahead of the arithmetic (36 more scalar
// registers).  VQ_PREFETCH_ACC=0 leaves the fused LBG kernel, which is short of scalar registers, without it.
#ifndef VQ_PREFETCH_ACC
#define VQ_PREFETCH_ACC 1
#endif
#define VQ_PREFETCH_IN(acc) (!(acc) || VQ_PREFETCH_ACC)
// The fused LBG kernel runs two workgroups of 12 waves per compute unit: six waves per SIMD, i.e. at most 80 vector
// registers (the register allocator otherwise stops at 82 and the second workgroup no longer fits).
#define VQ_ACC_ATTR(acc) __attribute__((amdgpu_waves_per_eu((acc) ? 6 : 1, (acc) ? 6 : 8)))
#ifdef VQ_WAVES_PER_EU
#define VQ_ASSIGN_ATTR __attribute__((amdgpu_waves_per_eu(VQ_WAVES_PER_EU, VQ_WAVES_PER_EU)))
#elif defined(VQ_NUM_SGPR)
#define VQ_ASSIGN_ATTR __attribute__((amdgpu_num_sgpr(VQ_NUM_SGPR)))
#else
#define VQ_ASSIGN_ATTR
#endif
//
// ACC (with PRUNE): the LBG pass of VQ_Full.py:81-90 fused in.  Every unambiguous vector's bytes are
// added to a per-workgroup LDS table [K][d+1] (sums | count), flushed with 64-bit global atomics when
// the workgroup ends (integer sums: order-free).  The lanes of a sorted wave mostly share a handful
// of labels, so "lane adds its own vector" would serialise on a few LDS addresses; instead a wave
// stages its 64 rows and labels in LDS and re-reads them transposed -- 64 / EPL vectors x EPL
// elements per instruction -- so that one atomic instruction spreads over 16+ distinct addresses.
template <int DP2, int R, int KEYMODE, bool LDS_CW, bool PRUNE, bool ACC = false>
__global__ __launch_bounds__(ACC ? 1024 : 512) VQ_ASSIGN_ATTR VQ_ACC_ATTR(ACC) void k_assign_fast(const AssignArgs a, const uint32_t *__restrict__ c16,
                                                     const uint32_t *__restrict__ bias,
                                                     const double *__restrict__ cbrows) {
    static_assert(!ACC || (PRUNE && DP2 <= 16), "the fused accumulation belongs to the pruned variant");
    // dynamic LDS, LDS_CW: [2 tiles of TILE_K * DP2 words][2 x TILE_K bias words]
    //              ACC:    [(K + 1) rows of 2*DP2 byte sums | count | pad][per wave: 64 rows of EPL bytes, 64 labels]
    extern __shared__ __align__(16) uint32_t s_tab[];
    if (gate_closed(a.gate)) return;
#ifdef VQ_TIMELINE
    const uint32_t tl_t0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
    uint32_t tl_rows = 0, tl_loaded = 0, tl_searched = 0, tl_resolved = 0;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int BLOCK = blockDim.x;
    const int d = a.d, K = a.K;
    const bool want_dist = a.min_dist != nullptr;
    constexpr int EPL = 2 * DP2;                               // staged bytes per row (elements beyond d are zero)
    // The scalar-load pruned search settles its near-ties itself (no queue, no fix-up launch): the rows a wave
    // searched contain every codeword that can be nearest to any of its vectors, so the wave evaluates that
    // range in fp64 for each ambiguous vector in turn.
    constexpr bool RESOLVE = resolves_in_kernel(DP2, LDS_CW, PRUNE);
    // LDS table of the fused accumulation: K rows of TROW words -- EPL byte sums (those beyond d stay zero), the
    // member count, one pad word, so that element pairs sit on 8-byte boundaries; row K is a dummy (never flushed)
    // The table exists in a.tab_reps copies (small codebooks): the vectors that one atomic instruction adds go to
    // different copies, so that vectors with the same label -- most of a sorted wave when K is small -- do not queue
    // up on one LDS address.
    constexpr int TROW = EPL + 2;
    const int tabn = K * TROW;
    const int rep_words = tabn + TROW;                         // one copy, dummy row included
    const int tab_all = ACC ? a.tab_reps * rep_words : 0;
    uint8_t *st_rows = reinterpret_cast<uint8_t *>(s_tab + ((tab_all + 3) & ~3)) + (tid >> 6) * (WAVE * EPL + WAVE * 4);
    int32_t *st_lbl = reinterpret_cast<int32_t *>(st_rows + WAVE * EPL);
    // ACC with a.cb_lds: the fp64 codebook behind the staging areas, rows padded to d + 1 doubles (lanes read the same
    // element of different rows: the odd stride spreads them over the banks)
    double *s_cbl = reinterpret_cast<double *>(reinterpret_cast<uint8_t *>(s_tab + ((tab_all + 3) & ~3)) +
                                               (BLOCK >> 6) * (WAVE * EPL + WAVE * 4));
    if (ACC) {
        for (int i = tid; i < tab_all; i += BLOCK) s_tab[i] = 0u;
        if (a.cb_lds) {
            for (int i = tid; i < K * d; i += BLOCK) {
                const int kk = i / d;
                s_cbl[kk * (d + 1) + (i - kk * d)] = a.cb[i];
            }
        }
        __syncthreads();
    }

    const uint32_t kbits = (uint32_t)a.kbits;
    const uint32_t kmask = (1u << kbits) - 1u;
    // The shift count lives in a vector register: v_lshl_add_u32 can then take the per-codeword
    // bias as its single scalar operand (one instruction instead of shift + add).
    uint32_t kbits_v = kbits;
    asm volatile("" : "+v"(kbits_v));

    // Workgroups are handed to the 8 XCDs round-robin.  When pruning, the tiles of one sort window write
    // their labels and distances into the same 32 / 64 KB of output, so consecutive tiles are given to
    // workgroups of the same XCD (one L2 then sees all writes to a line before it is evicted).
    int64_t first_tile = blockIdx.x;
    if (PRUNE && (gridDim.x & 7) == 0) {
        const uint32_t tx = gridDim.x >> 3;
        first_tile = (int64_t)(blockIdx.x & 7) * tx + (blockIdx.x >> 3);
        // Launches of only a few rounds of workgroups (a.heavy_first): within an XCD's share, the tiles from the
        // middle of every sort window -- the most codebook rows to search -- go first and those from its two ends
        // last, so that the workgroups that start last are the short ones.
        const uint32_t TPW = SORT_WIN / (blockDim.x * R);     // tiles per sort window
        if (a.heavy_first && TPW >= 2 && tx % TPW == 0) {
            const uint32_t nwin = tx / TPW, j = blockIdx.x >> 3;
            uint32_t c = j / nwin;
            const uint32_t w = j - c * nwin;
            // a launch that is resident all at once (heavy_first = 2): the compute units of an XCD take its workgroups
            // in turn, so the workgroups that share one are a fixed number of classes apart; running every other
            // stretch of that many classes backwards puts long and short workgroups together
            if (a.heavy_first == 2 && nwin <= 32 && 32 % nwin == 0) {
                const uint32_t m = 32 / nwin;
                if (TPW % m == 0 && ((c / m) & 1)) c = (c / m) * m + (m - 1 - c % m);
            }
            // classes 0, 1, 2, ... -> the middle tile, then its right and left neighbours in turn: a permutation of the
            // window's TPW tiles whether TPW is even or odd (12-wave workgroups take 5 tiles and a third per window)
            const uint32_t mid = (TPW - 1) / 2;
            const uint32_t pos = (c & 1) ? mid + 1 + (c >> 1) : mid - (c >> 1);
            first_tile = (int64_t)(blockIdx.x & 7) * tx + (int64_t)w * TPW + pos;
        }
    }
    // One tile per workgroup (the host launches at least ntiles workgroups): nothing is carried from tile to tile, so
    // no value has to stay live across the whole kernel.
    // (base: the first vector of the tile; wave_off: where in it this wave's 64 * R vectors start)
    auto do_tile = [&](const int64_t base, const int wave_off) {
        // ---- load R vectors per lane and convert to packed 16-bit fixed point ----
        uint32_t x[R][DP2];
        uint32_t origin[R];                                    // PRUNE: where in its window the vector came from
        // position of lane's r-th vector: a wave takes 64 * R consecutive vectors when pruning (they are
        // neighbours in sorted order), otherwise the R vectors of a lane are a workgroup apart
        auto vec_index = [&](int r) -> int64_t {
            return PRUNE ? base + wave_off + r * WAVE + lane : base + (int64_t)r * BLOCK + tid;
        };
        // The vectors are requested first and looked at last: the header reads below (three more memory round trips)
        // run while they are on their way.  Word-wise rows wait in x[r][even] as they came.
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int64_t n = vec_index(r);
            if (n >= a.N) n = a.N - 1;                 // clamp: loads stay in bounds, result is discarded
            const uint8_t *row = (PRUNE ? a.rows : a.blocks) + (size_t)n * d;
            // (the fused LBG pass of a training run writes no labels and leaves its distances in sorted order: it
            // has no use for the origin map, except in the partial last chunk)
            origin[r] = 0u;
            if (PRUNE && (base + wave_off + WAVE * R > a.dist_sorted_end))
                origin[r] = (uint32_t)a.perm[n];
            // Out-of-row positions (d below the template's padded size) are loaded from a clamped
            // index and zeroed later, so that no load sits behind a branch.
            if (DP2 % 8 == 0 && a.vec4 == 2) {
                // rows of exactly 2 * DP2 bytes, 16-byte aligned: whole 16-byte pieces from one address
#pragma unroll
                for (int q = 0; q < DP2 / 8; ++q) {
                    const uint4 v = reinterpret_cast<const uint4 *>(row)[q];
                    x[r][8 * q] = v.x;
                    x[r][8 * q + 2] = v.y;
                    x[r][8 * q + 4] = v.z;
                    x[r][8 * q + 6] = v.w;
                }
            } else if (a.vec4) {
                const uint32_t *row4 = reinterpret_cast<const uint32_t *>(row);
                const int nw = d >> 2;
#pragma unroll
                for (int p = 0; p < DP2; p += 2) x[r][p] = row4[(p >> 1) < nw ? (p >> 1) : 0];
            } else {
#pragma unroll
                for (int p = 0; p < DP2; ++p) {
                    const int e0 = 2 * p, e1 = 2 * p + 1;
                    uint32_t b0 = row[e0 < d ? e0 : 0];
                    uint32_t b1 = row[e1 < d ? e1 : 0];
                    b0 = e0 < d ? b0 : 0u;
                    b1 = e1 < d ? b1 : 0u;
                    x[r][p] = (b0 | (b1 << 16)) << 8;
                    // long rows: sixteen bytes on their way at a time (a register per byte of the whole row would set
                    // the register count of the kernel)
                    if (DP2 > 8 && p % 8 == 7) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (PRUNE) {
            // (wave-uniform) the sorted copy must be the one made for these N vectors of d bytes; otherwise its
            // origin map cannot be trusted: every label of this workgroup's tile becomes -1 and nothing else
            // is written
            if (a.shdr->magic != SORTED_MAGIC || a.shdr->N != (unsigned long long)a.N || a.shdr->d != (uint32_t)a.d) {
                for (int r = 0; r < R; ++r) {
                    const int64_t n = vec_index(r);
                    if (n < a.N && a.labels) a.labels[n] = -1;
                }
                return;
            }
        }
        const uint32_t s = a.hdr->shift;
        const uint32_t emax = a.hdr->emax;
        const uint32_t window = emax >= 0x3fffffffu ? 0xffffffffu : 2u * emax;   // ambiguous iff S2 - S1 <= window
        uint32_t sh8_v = 8u - s;                               // (a vector register too: shifts by a scalar issue slower)
        asm volatile("" : "+v"(sh8_v));
        // bytes (0, 1) / (2, 3) of a word as a pair of 16-bit fixed-point values x << s: v_perm puts each byte into the
        // upper byte of its half (x << 8), one right shift takes the pair down to x << s (s <= 8)
        if (a.vec4) {
            const int nw = d >> 2;
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int p = 0; p < DP2; p += 2) {
                    const uint32_t w = (p >> 1) < nw ? x[r][p] : 0u;
                    x[r][p] = __builtin_amdgcn_perm(0u, w, 0x010c000cu) >> sh8_v;
                    x[r][p + 1] = __builtin_amdgcn_perm(0u, w, 0x030c020cu) >> sh8_v;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int p = 0; p < DP2; ++p) x[r][p] >>= sh8_v;
            }
        }
#ifdef VQ_TIMELINE
        __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0): the vectors have arrived
        tl_loaded = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
        // ---- filter: two smallest keys per vector ----
        uint32_t rows_searched = 0;                            // codebook rows this wave evaluated (wave-uniform)
        int range_lo = 0, range_hi = 0;                        // RESOLVE: the rows [range_lo, range_hi) it searched
        uint32_t m1[R], m2[R], bi[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            m1[r] = 0xffffffffu;
            m2[r] = 0xffffffffu;
            bi[r] = 0;
        }
        // fold one codeword's SAD sums into the two smallest keys of each vector
        auto finish = [&](const uint32_t *acc, uint32_t b, int k) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (KEYMODE == KEY_PACKED) {
                    uint32_t key = (acc[r] << kbits_v) + b;
                    m2[r] = umed3(key, m1[r], m2[r]);
                    m1[r] = min(key, m1[r]);
                } else {
                    uint32_t v = acc[r] + ((b >> 31) << 30);   // duplicate rows never compete
                    m2[r] = umed3(v, m1[r], m2[r]);
                    bi[r] = (v < m1[r]) ? (uint32_t)k : bi[r];
                    m1[r] = min(v, m1[r]);
                }
            }
        };
        auto step = [&](const uint32_t *cw, uint32_t b, int k) {
            uint32_t acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[r] = 0;
#pragma unroll
                for (int p = 0; p < DP2; ++p) acc[r] = __builtin_amdgcn_sad_u16(x[r][p], cw[p], acc[r]);
            }
            finish(acc, b, k);
        };
        // Two codewords at once (packed keys): with keys a and b, the new smallest is min3(a, b, m1) and the new
        // second smallest min(med3(a, b, m1), m2) -- five instructions for the pair instead of six.
        auto step2 = [&](const uint32_t *cw0, const uint32_t *cw1, uint32_t b0, uint32_t b1) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                uint32_t a0 = 0, a1 = 0;
#pragma unroll
                for (int p = 0; p < DP2; ++p) a0 = __builtin_amdgcn_sad_u16(x[r][p], cw0[p], a0);
#pragma unroll
                for (int p = 0; p < DP2; ++p) a1 = __builtin_amdgcn_sad_u16(x[r][p], cw1[p], a1);
                const uint32_t k0 = (a0 << kbits_v) + b0, k1 = (a1 << kbits_v) + b1;
                const uint32_t md = umed3(k0, k1, m1[r]);
                m1[r] = min(min(k0, k1), m1[r]);
                m2[r] = min(md, m2[r]);
            }
        };
        if constexpr (!LDS_CW) {
            // Codewords are fetched by scalar loads (wave-uniform addresses) in batches of U, so that
            // one scalar-memory wait is paid per U * R * (DP2 + 3) vector instructions.
            constexpr int U = DP2 <= 8 ? (PRUNE ? VQ_UP : VQ_U8) : (DP2 <= 16 ? 2 : 1);
            // rows [lo, hi) of the fixed-point codebook, hi - lo a multiple of U (rows K .. Kpad are
            // pad rows that never win)
            // 16-byte rows in pairs (the pruned search at d <= 16): the rows of the next trip are requested before this
            // trip's arithmetic starts, so that the wave does not sit out a scalar-memory round trip per batch (the
            // last trip re-reads its own rows).  The compiler folds a rotated loop of plain loads back into load-and-wait,
            // hence the explicit requests.
            auto run = [&](int lo, int hi) {
                if (lo >= hi) return;                          // (wave-uniform)
                if constexpr (DP2 == 8 && U == 2 && KEYMODE == KEY_PACKED && VQ_PREFETCH_IN(ACC)) {
                    // two 32-byte rows are one 64-byte scalar load, their biases one 8-byte load (lo is even)
                    u32x16 cur;
                    u32x2 cb2;
                    sload_x16(cur, c16, (uint32_t)lo * 32u);
                    sload_x2(cb2, bias, (uint32_t)lo * 4u);
                    sload_wait(cur, cb2);
                    for (int k = lo; k < hi; k += 2) {
                        u32x16 nxt;
                        u32x2 nb2;
                        const uint32_t kn = (uint32_t)(k + 2 < hi ? k + 2 : k);
                        sload_x16(nxt, c16, kn * 32u);
                        sload_x2(nb2, bias, kn * 4u);
                        __builtin_amdgcn_sched_barrier(0);     // the requests go out before the arithmetic, not after it
                        const uint32_t cw0[8] = {cur[0], cur[1], cur[2], cur[3], cur[4], cur[5], cur[6], cur[7]};
                        const uint32_t cw1[8] = {cur[8], cur[9], cur[10], cur[11], cur[12], cur[13], cur[14], cur[15]};
                        step2(cw0, cw1, cb2[0], cb2[1]);
                        sload_wait(nxt, nb2);
                        cur = nxt;
                        cb2 = nb2;
                    }
                    return;
                }
                for (int k = lo; k < hi; k += U) {
                    uint32_t cw[U][DP2], b[U];
#pragma unroll
                    for (int j = 0; j < U; ++j) {
#pragma unroll
                        for (int p = 0; p < DP2; ++p) cw[j][p] = c16[(size_t)(k + j) * DP2 + p];
                        b[j] = bias[k + j];
                    }
                    if constexpr (U % 2 == 0 && KEYMODE == KEY_PACKED) {
#pragma unroll
                        for (int j = 0; j < U; j += 2) step2(cw[j], cw[j + 1], b[j], b[j + 1]);
                    } else {
#pragma unroll
                        for (int j = 0; j < U; ++j) step(cw[j], b[j], k + j);
                    }
                }
            };
            const int kpad = (K + 7) & ~7;
            if constexpr (!PRUNE) {
                run(0, (K + U - 1) / U * U);
                rows_searched = (uint32_t)((K + U - 1) / U * U);
            } else {
                // wave-uniform range of the vectors' element sums (same fixed-point units as the rows)
                uint32_t sxlo = 0xffffffffu, sxhi = 0u;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    uint32_t sx = 0;
#pragma unroll
                    for (int p = 0; p < DP2; ++p) sx = __builtin_amdgcn_sad_u16(x[r][p], 0u, sx);
                    sxlo = min(sxlo, sx);
                    sxhi = max(sxhi, sx);
                }
                sxlo = wave_reduce_u32<false>(sxlo);
                sxhi = wave_reduce_u32<true>(sxhi);
                // (marked wave-uniform: the table look-ups below then are scalar loads)
                const uint32_t tshift = __builtin_amdgcn_readfirstlane(a.hdr->tshift), nt = __builtin_amdgcn_readfirstlane(a.hdr->nt);
                const int G = a.round_rows;                    // codewords added per side and round
                const int G0 = a.first_rows;                   // codewords of the first round, around the wave's own sum
                int pa = (int)(a.postab[((sxlo >> 1) + (sxhi >> 1)) >> tshift] & ~7u) - G0 / 2;
                if (pa > kpad - G0) pa = kpad - G0;
                if (pa < 0) pa = 0;
                int pb = pa + G0 < kpad ? pa + G0 : kpad;
                // segments still to search this round: [l0, h0) and [l1, h1); [pa, pb) is what has been searched
                int l0 = pa, h0 = pb, l1 = 0, h1 = 0;
                for (;;) {
#pragma nounroll
                    for (int sg = 0; sg < 2; ++sg) run(sg ? l1 : l0, sg ? h1 : h0);
                    uint32_t bm = 0;
#pragma unroll
                    for (int r = 0; r < R; ++r) bm = max(bm, m1[r]);
                    bm = wave_reduce_u32<true>(bm);
                    if (KEYMODE == KEY_PACKED) bm >>= kbits;   // worst "best SAD so far" of the wave
                    uint32_t thr = bm + window;                // a codeword further away than this matters to no vector
                    if (thr < bm) thr = 0xffffffffu;
                    const uint32_t lo_sum = sxlo > thr ? sxlo - thr : 0u;
                    const unsigned long long hi_sum = (unsigned long long)sxhi + thr;
                    unsigned long long ih = (hi_sum >> tshift) + 1ull;
                    if (ih > nt + 1ull) ih = nt + 1ull;
                    const int klo = (int)(a.postab[lo_sum >> tshift] & ~7u);
                    const int khi = (int)((a.postab[ih] + 7u) & ~7u);
                    const int na = pa > klo ? (pa - G > klo ? pa - G : klo) : pa;
                    const int nb = pb < khi ? (pb + G < khi ? pb + G : khi) : pb;
                    if (na == pa && nb == pb) break;
                    l0 = na; h0 = pa; l1 = pb; h1 = nb;
                    pa = na;
                    pb = nb;
                }
                rows_searched = (uint32_t)(pb - pa);
                range_lo = pa;
                range_hi = pb < K ? pb : K;                    // rows K .. Kpad are pad rows
            }
        } else {
            constexpr int TILE_K = tile_k_for(DP2);
            constexpr int TILE_W = TILE_K * DP2;               // words of fixed-point rows per tile
            constexpr int NV = TILE_W / 4;                     // 16-byte pieces per tile
            constexpr int MAXL = (NV + 127) / 128;             // pieces per thread (workgroups have >= 128 threads)
            uint32_t *s_cw = s_tab;
            uint32_t *s_b = s_cw + 2 * TILE_W;
            const int ntk = (K + TILE_K - 1) / TILE_K;
            const uint32_t total_w = (uint32_t)K * DP2;
            const uint4 *c16v = reinterpret_cast<const uint4 *>(c16);
            uint4 pre[MAXL];
            uint32_t preb = 0;
            // rows past the end of the codebook are staged as zeros with the never-competes bias
            auto gload = [&](int t) {
#pragma unroll
                for (int l = 0; l < MAXL; ++l) {
                    const int i = tid + l * BLOCK;
                    const uint32_t w0 = (uint32_t)t * TILE_W + (uint32_t)i * 4;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (i < NV && w0 < total_w) v = c16v[w0 >> 2];
                    if (w0 + 1 >= total_w) v.y = 0;
                    if (w0 + 2 >= total_w) v.z = 0;
                    if (w0 + 3 >= total_w) v.w = 0;
                    pre[l] = v;
                }
                const int kk = t * TILE_K + tid;
                preb = (tid < TILE_K && kk < K) ? bias[kk] : 0x80000000u;
            };
            auto lstore = [&](int slot) {
                uint4 *dst = reinterpret_cast<uint4 *>(s_cw + slot * TILE_W);
#pragma unroll
                for (int l = 0; l < MAXL; ++l) {
                    const int i = tid + l * BLOCK;
                    if (i < NV) dst[i] = pre[l];
                }
                if (tid < TILE_K) s_b[slot * TILE_K + tid] = preb;
            };
            uint32_t vz = 0;
            asm volatile("" : "+v"(vz));
            // tiles [tlo, thi) of the fixed-point codebook (workgroup-uniform range), double-buffered
            auto run_tiles = [&](int tlo, int thi) {
            if (tlo >= thi) return;
            __syncthreads();                                   // whatever used the buffers before is done with them
            gload(tlo);
            lstore(0);
            __syncthreads();
            for (int t = tlo; t < thi; ++t) {
                const int slot = (t - tlo) & 1;
                if (t + 1 < thi) gload(t + 1);
                // vz is an opaque per-lane zero: it keeps the compiler from treating these (wave-uniform)
                // LDS reads as scalar values, which it would move to scalar registers one
                // v_readfirstlane at a time
                const uint32_t *cwbuf = s_cw + slot * TILE_W + vz;
                const uint32_t *bbuf = s_b + slot * TILE_K + vz;
                if constexpr (DP2 <= 16) {
#pragma unroll 2
                    for (int kk = 0; kk < TILE_K; ++kk) {
                        uint32_t cw[DP2];
                        if constexpr (DP2 % 4 == 0) {              // rows of whole 16-byte pieces: ds_read_b128
                            const uint4 *rowv = reinterpret_cast<const uint4 *>(cwbuf + kk * DP2);
#pragma unroll
                            for (int q = 0; q < DP2 / 4; ++q) {
                                const uint4 v = rowv[q];
                                cw[4 * q] = v.x;
                                cw[4 * q + 1] = v.y;
                                cw[4 * q + 2] = v.z;
                                cw[4 * q + 3] = v.w;
                            }
                        } else {
#pragma unroll
                            for (int p = 0; p < DP2; ++p) cw[p] = cwbuf[kk * DP2 + p];
                        }
                        step(cw, bbuf[kk], t * TILE_K + kk);
                    }
                } else {
                    // long rows: read the row in chunks of CH words, one chunk ahead of the SADs that
                    // use it, so that only two chunks of the row are in registers at a time
                    // (the rows start at multiples of 16 bytes -- DP2 and the tile are multiples of four words -- and are
                    // read as 16-byte pieces: one ds_read_b128 moves in 4 LDS cycles what two ds_read2_b32 move in 8)
                    constexpr int CH = 8;
                    constexpr int NCH = DP2 / CH;
                    static_assert(DP2 % CH == 0 && TILE_W % 4 == 0, "rows are whole 16-byte pieces");
                    for (int kk = 0; kk < TILE_K; ++kk) {
                        const uint4 *rowv = reinterpret_cast<const uint4 *>(cwbuf + kk * DP2);
                        uint32_t acc[R];
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r] = 0;
                        uint4 cur0 = rowv[0], cur1 = rowv[1], nxt0 = cur0, nxt1 = cur1;
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            if (c + 1 < NCH) {
                                nxt0 = rowv[2 * (c + 1)];
                                nxt1 = rowv[2 * (c + 1) + 1];
                            }
                            const uint32_t cur[CH] = {cur0.x, cur0.y, cur0.z, cur0.w, cur1.x, cur1.y, cur1.z, cur1.w};
#pragma unroll
                            for (int r = 0; r < R; ++r) {
#pragma unroll
                                for (int p = 0; p < CH; ++p) acc[r] = __builtin_amdgcn_sad_u16(x[r][c * CH + p], cur[p], acc[r]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            cur0 = nxt0;
                            cur1 = nxt1;
                        }
                        finish(acc, bbuf[kk], t * TILE_K + kk);
                    }
                }
                if (t + 1 < thi) lstore(slot ^ 1);
                __syncthreads();
            }
            };
            if constexpr (!PRUNE) {
                run_tiles(0, ntk);
                rows_searched = (uint32_t)(ntk * TILE_K);
            } else {
                // The element-sum bound at workgroup granularity: the BLOCK * R vectors of the workgroup are
                // neighbours in sorted order, every wave walks the same tiles, and the range of sorted rows
                // still to search comes from the workgroup-wide worst "best SAD so far" (one LDS exchange and
                // two barriers per round).  All range arithmetic is workgroup-uniform.
                uint32_t *s_red = s_b + 2 * TILE_K;            // [0..15] per-wave values, [16..18] results
                const int nwv = BLOCK >> 6, wv = tid >> 6;
                auto wg_reduce = [&](uint32_t vmin, uint32_t vmax, uint32_t &omin, uint32_t &omax) {
                    vmin = wave_reduce_u32<false>(vmin);
                    vmax = wave_reduce_u32<true>(vmax);
                    __syncthreads();
                    if (lane == 0) {
                        s_red[wv] = vmin;
                        s_red[16 + wv] = vmax;
                    }
                    __syncthreads();
                    uint32_t a0 = 0xffffffffu, a1 = 0u;
                    for (int i = 0; i < nwv; ++i) {
                        a0 = min(a0, s_red[i]);
                        a1 = max(a1, s_red[16 + i]);
                    }
                    omin = __builtin_amdgcn_readfirstlane(a0);
                    omax = __builtin_amdgcn_readfirstlane(a1);
                };
                uint32_t sxlo = 0xffffffffu, sxhi = 0u;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    uint32_t sx = 0;
#pragma unroll
                    for (int p = 0; p < DP2; ++p) sx = __builtin_amdgcn_sad_u16(x[r][p], 0u, sx);
                    sxlo = min(sxlo, sx);
                    sxhi = max(sxhi, sx);
                }
                wg_reduce(sxlo, sxhi, sxlo, sxhi);
                const uint32_t tshift = a.hdr->tshift, nt = a.hdr->nt;
                // tiles added per side and round: about 32 rows, at least one tile
#ifndef VQ_GT_ROWS
#define VQ_GT_ROWS 32
#endif
                constexpr int GT = TILE_K >= VQ_GT_ROWS ? 1 : VQ_GT_ROWS / TILE_K;
                int ta = (int)(a.postab[((sxlo >> 1) + (sxhi >> 1)) >> tshift] / TILE_K) - GT / 2;
                if (ta > ntk - GT) ta = ntk - GT;
                if (ta < 0) ta = 0;
                int tb = ta + GT < ntk ? ta + GT : ntk;
                int l0 = ta, h0 = tb, l1 = 0, h1 = 0;          // tile segments still to search this round
                for (;;) {
                    run_tiles(l0, h0);
                    run_tiles(l1, h1);
                    uint32_t bm = 0, unused;
#pragma unroll
                    for (int r = 0; r < R; ++r) bm = max(bm, m1[r]);
                    wg_reduce(bm, bm, unused, bm);
                    if (KEYMODE == KEY_PACKED) bm >>= kbits;   // worst "best SAD so far" of the workgroup
                    uint32_t thr = bm + window;                // a codeword further away than this matters to no vector
                    if (thr < bm) thr = 0xffffffffu;
                    const uint32_t lo_sum = sxlo > thr ? sxlo - thr : 0u;
                    const unsigned long long hi_sum = (unsigned long long)sxhi + thr;
                    unsigned long long ih = (hi_sum >> tshift) + 1ull;
                    if (ih > nt + 1ull) ih = nt + 1ull;
                    const int tlo = (int)(a.postab[lo_sum >> tshift] / TILE_K);
                    const int thi = (int)((a.postab[ih] + TILE_K - 1) / TILE_K);
                    const int na = ta > tlo ? (ta - GT > tlo ? ta - GT : tlo) : ta;
                    const int nb = tb < thi ? (tb + GT < thi ? tb + GT : thi) : tb;
                    if (na == ta && nb == tb) break;
                    l0 = na; h0 = ta; l1 = tb; h1 = nb;
                    ta = na;
                    tb = nb;
                }
                rows_searched = (uint32_t)((tb - ta) * TILE_K);
            }
        }
#ifdef VQ_TIMELINE
        tl_searched = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
        // ---- near-ties settled in the kernel (RESOLVE) ----
        // the outcome replaces the vector's keys: the winner, and no runner-up
        if constexpr (RESOLVE) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t ns = vec_index(r);
                const uint32_t s1 = KEYMODE == KEY_PACKED ? m1[r] >> kbits : m1[r];
                const uint32_t s2 = KEYMODE == KEY_PACKED ? m2[r] >> kbits : m2[r];
                const unsigned long long amask = __ballot(ns < a.N && (s2 - s1 <= window));
                if (amask && lane == 0)                        // statistics: vectors re-checked in fp64
                    atomicAdd(a.qcount + 16 * ((uint32_t)((ns - lane) >> 6) & (NQ - 1)) + 1, (uint32_t)__popcll(amask));
                for (unsigned long long todo = amask; todo; todo &= todo - 1ull) {      // (wave-uniform)
                    const int owner = __ffsll((long long)todo) - 1;
                    uint32_t xw[DP2];                          // the vector, wave-uniform
#pragma unroll
                    for (int p = 0; p < DP2; ++p) xw[p] = __builtin_amdgcn_readlane(x[r][p], owner);
                    // The nearest codeword in fp64 is among the searched rows whose fixed-point SAD lies within
                    // the ambiguity window of the smallest one.  Lanes take one row each and recompute its SAD
                    // (the vector is a scalar operand); for every candidate the wave reads the fp64 codeword
                    // (one address for all lanes) and sums |x_e - c_e| in element order, the same value in all lanes.
                    // The lowest codeword index wins ties, as in the fix-up kernels.
                    const uint32_t s1o = __builtin_amdgcn_readlane(s1, owner);
                    const uint32_t lim = s1o + window < s1o ? 0xffffffffu : s1o + window;
                    double best = __builtin_huge_val();
                    int bk = 0x7fffffff, bpos = 0;
                    for (int p0 = range_lo; p0 < range_hi; p0 += WAVE) {
                        const int pos = p0 + lane;
                        const bool on = pos < range_hi;
                        const int pc = on ? pos : range_hi - 1;
                        const uint32_t *crow16 = c16 + (size_t)pc * DP2;
                        const uint32_t kk_lane = bias[pc] & 0x7fffffffu;
                        uint32_t sad = 0;
#pragma unroll
                        for (int p = 0; p < DP2; ++p) {
                            sad = __builtin_amdgcn_sad_u16(crow16[p], xw[p], sad);
                            if (p % 4 == 3) __builtin_amdgcn_sched_barrier(0);     // at most four row words in registers
                        }
                        for (unsigned long long cm = __ballot(on && sad <= lim); cm; cm &= cm - 1ull) {
                            const int cl = __ffsll((long long)cm) - 1;
                            const int kk = (int)__builtin_amdgcn_readlane(kk_lane, cl);
                            const double *crow = cbrows + (size_t)kk * d;       // wave-uniform: scalar loads
                            double acc = 0.0;
                            uint32_t sv = s;                   // opaque per trip: the element conversions stay inside
                            asm volatile("" : "+v"(sv));       // the loop (hoisted, they would hold 4 * DP2 registers)
                            if (d == 2 * DP2) {
                                // whole rows: no test between the elements, so the row arrives four elements per scalar
                                // load instead of one memory round trip per element
                                constexpr int CE = (2 * DP2) % 4 == 0 ? 4 : 2;
#pragma unroll
                                for (int e0 = 0; e0 < 2 * DP2; e0 += CE) {
                                    double cv[CE];
#pragma unroll
                                    for (int i = 0; i < CE; ++i) cv[i] = crow[e0 + i];
#pragma unroll
                                    for (int i = 0; i < CE; ++i) {
                                        const double xe = (double)((xw[(e0 + i) >> 1] >> (sv + 16 * (i & 1))) & 0xffu);
                                        acc += fabs(xe - cv[i]);
                                    }
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            } else {
#pragma unroll
                                for (int e = 0; e < 2 * DP2; ++e) {
                                    if (e < d) {
                                        const double xe = (double)((xw[e >> 1] >> (sv + 16 * (e & 1))) & 0xffu);
                                        acc += fabs(xe - crow[e]);
                                    }
                                }
                            }
                            if (acc < best || (acc == best && kk < bk)) {
                                best = acc;
                                bk = kk;
                                bpos = p0 + cl;
                            }
                        }
                    }
                    if (lane == owner) {
                        if (KEYMODE == KEY_PACKED) m1[r] = (m1[r] & ~kmask) | (uint32_t)bk;
                        else bi[r] = (uint32_t)bpos;
                        m2[r] = 0xffffffffu;
                    }
                }
            }
        }
#ifdef VQ_TIMELINE
        tl_rows += rows_searched;
        tl_resolved = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
        // statistics: (vector, codeword) pairs this wave evaluated, summed into the 64-bit word at byte 8 of
        // the wave's sub-queue slot
        if (lane == 0)
            atomicAdd(reinterpret_cast<unsigned long long *>(a.qcount + 16 * ((uint32_t)(vec_index(0) >> 6) & (NQ - 1)) + 2),
                      (unsigned long long)rows_searched * (unsigned long long)(WAVE * R));
        // ---- decide, measure, write, accumulate ----
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t ns = vec_index(r);                   // position in the order the vectors were read in
            const bool valid = ns < a.N;
            // index of the vector in the caller's order: the sorted copy keeps every vector inside its window
            int64_t n = ns;
            if (PRUNE) {
                n = (ns & ~(int64_t)(SORT_WIN - 1)) + (int64_t)origin[r];
                n = n < a.N ? n : a.N - 1;                     // a `sorted` buffer that was not filled must not lead out of bounds
            }
            uint32_t s1, s2;
            int idx;
            if (KEYMODE == KEY_PACKED) {
                idx = (int)(m1[r] & kmask);
                s1 = m1[r] >> kbits;
                s2 = m2[r] >> kbits;
            } else {
                idx = (int)(bias[bi[r]] & 0x7fffffffu);         // rows are stored sorted: position -> codeword index
                s1 = m1[r];
                s2 = m2[r];
            }
            // near-tie: the filter cannot rank the two best candidates; queue the vector for the
            // fp64 fix-up kernel (one atomic per wave that has any)
            const bool amb = !RESOLVE && valid && (s2 - s1 <= window);
            const unsigned long long amask = __ballot(amb);
            if (amask) {
                const int leader = __ffsll((long long)amask) - 1;
                const uint32_t q = (uint32_t)((ns - lane) >> 6) & (NQ - 1);    // sub-queue of this group of 64 vectors
                uint32_t qbase = 0;
                if (lane == leader) qbase = atomicAdd(a.qcount + 16 * q, (uint32_t)__popcll(amask));
                qbase = __builtin_amdgcn_readlane(qbase, leader);
                if (amb)
                    a.queue[(size_t)q * a.qcap + qbase + (uint32_t)__popcll(amask & ((1ull << lane) - 1ull))] =
                        PRUNE ? (uint32_t)ns : (uint32_t)n;       // pruned: position in the sorted copy
            }
            const bool settled = valid && !amb;
            if constexpr (ACC) {
                // stage this group's rows (raw bytes) and labels (as byte offsets of their table rows; row K
                // is a dummy that takes the vectors not settled here), then add them transposed
#pragma unroll
                for (int p = 0; p < DP2; p += 2) {
                    // x >> s holds two bytes at positions 0 and 2: pick [lo.0, lo.2, hi.0, hi.2]
                    const uint32_t w = __builtin_amdgcn_perm(x[r][p + 1] >> s, x[r][p] >> s, 0x06040200u);
                    *reinterpret_cast<uint32_t *>(st_rows + lane * EPL + 2 * p) = w;
                }
                // (the copy of the table this vector goes to: its slot within the instruction that adds it)
                const int my_row = ((lane & (a.tab_reps - 1)) * (K + 1) + (settled ? idx : K)) * TROW;
                st_lbl[lane] = my_row * 4;
                // the staging area belongs to this wave alone and a wave's LDS operations complete in order:
                // only the compiler has to be kept from moving them (a fence would also wait for the
                // global stores in flight)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // one 64-bit LDS add carries TWO byte sums (the low sum cannot carry into the high one: a
                // workgroup adds at most 1024 x 255 to it): lanes = VS vectors x DP2 element pairs
                constexpr int VS = WAVE / DP2;                   // vectors per instruction
                const int g = lane / DP2, pr = lane % DP2;
                char *tab_p = reinterpret_cast<char *>(s_tab) + 8 * pr;
                constexpr int STEPS = (WAVE + VS - 1) / VS;
#pragma unroll
                for (int j = 0; j < STEPS; ++j) {
                    const int src = j * VS + g;
                    const bool on = (WAVE % DP2 == 0) || (g < VS && src < WAVE);   // (DP2 divides the wave: always)
                    const int srcc = on ? src : 0;
                    const int off = st_lbl[srcc];
                    const uint32_t two = *reinterpret_cast<const uint16_t *>(st_rows + srcc * EPL + 2 * pr);
                    const unsigned long long v = (unsigned long long)(two & 0xffu) | ((unsigned long long)(two >> 8) << 32);
                    if (on) atomicAdd(reinterpret_cast<unsigned long long *>(tab_p + off), v);
                }
                atomicAdd(&s_tab[my_row + EPL], 1u);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging area is rewritten by the next group
            }
            if (!settled) continue;
            // (null: the fused LBG pass of a training run; whole chunks of a training pass: in sorted order, like
            // the distances -- full-width stores instead of a scatter through the origin map)
            if (a.labels) a.labels[PRUNE && ns < a.dist_sorted_end ? ns : n] = idx;
            if (want_dist) {
                double acc = 0.0;
                auto measure = [&](const double *crow) {
                    if (d == 2 * DP2) {
                        // whole rows: four codeword elements are fetched at a time, ahead of the additions that use
                        // them (a test per element would put a memory round trip between any two additions)
                        constexpr int CE = (2 * DP2) % 4 == 0 ? 4 : 2;     // elements per batch
#pragma unroll
                        for (int e0 = 0; e0 < 2 * DP2; e0 += CE) {
                            double cv[CE];
#pragma unroll
                            for (int i = 0; i < CE; ++i) cv[i] = crow[e0 + i];
#pragma unroll
                            for (int i = 0; i < CE; ++i) {
                                const double xv = (double)((x[r][(e0 + i) >> 1] >> (s + 16 * (i & 1))) & 0xffu);
                                acc += fabs(xv - cv[i]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        return;
                    }
#pragma unroll
                    for (int p = 0; p < DP2; ++p) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            int e = 2 * p + h;
                            if (e < d) {
                                double xv = (double)((x[r][p] >> (s + 16 * h)) & 0xffu);
                                acc += fabs(xv - crow[e]);
                            }
                        }
                        // long rows: at most eight codeword elements on their way at a time (all of them at once
                        // would set the register count of the whole kernel)
                        if (DP2 > 8 && p % 4 == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                };
                // every lane reads another codeword: from global memory that is 64 cache lines per load instruction,
                // from the LDS copy a conflict-free read
                if (ACC && a.cb_lds) measure(s_cbl + idx * (d + 1));
                else measure(a.cb + (size_t)idx * d);
                a.min_dist[PRUNE && ns < a.dist_sorted_end ? ns : n] = acc;
            }
        }
    };
    bool looped = false;
    if constexpr (ACC) {
        if (a.persist_tiles > 0) {
            // Resident workgroups whose waves fetch tiles of 64 * R vectors until their XCD's share is used up: the LDS
            // table and the LDS copy of the codebook are set up and flushed once per workgroup instead of once per
            // tile, and a wave that is done with a tile does not wait for the rest of its workgroup.  The counters
            // (word 4 of the first eight counter slots) are zeroed with the queue counters before the launch.
            // (Measured for the plain search too, d = 16 and d = 48 at two to four rounds of workgroups: not faster,
            // and the loop costs 14 to 20 vector registers.)
            looped = true;
            const uint32_t xcd = blockIdx.x & 7;
            uint32_t *ctr = a.qcount + 16 * xcd + 4;
            for (;;) {
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(ctr, 1u);
                t = __builtin_amdgcn_readfirstlane(t);
                if (t >= (uint32_t)a.persist_tiles) break;
                const int64_t wbase = ((int64_t)xcd * a.persist_tiles + t) * (WAVE * R);
                if (wbase >= a.N) break;
                do_tile(wbase, 0);
            }
        }
    }
    if (!looped && first_tile < a.ntiles) do_tile(first_tile * (int64_t)(BLOCK * R), (tid >> 6) * (WAVE * R));
#ifdef VQ_TIMELINE
    {
        const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (tid >> 6);
        if (lane == 0 && w < (1u << 18)) {
            g_timeline[8 * w] = tl_t0;
            g_timeline[8 * w + 1] = (uint32_t)__builtin_amdgcn_s_memrealtime();
            g_timeline[8 * w + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | (__builtin_amdgcn_s_getreg((3 << 11) | 20) << 28);   // HW_ID | XCC_ID << 28
            g_timeline[8 * w + 3] = tl_rows;
            g_timeline[8 * w + 4] = tl_loaded;
            g_timeline[8 * w + 5] = tl_searched;
            g_timeline[8 * w + 6] = tl_resolved;
        }
    }
#endif
    if (ACC) {
        __syncthreads();
        unsigned long long *rep = a.acc_rep + (size_t)(blockIdx.x % NREP) * (K * (d + 1));
        for (int i = tid; i < tabn; i += BLOCK) {
            uint32_t v = 0;
            for (int c = 0; c < a.tab_reps; ++c) v += s_tab[c * rep_words + i];
            const int kk = i / TROW, e = i - kk * TROW;            // replica rows are [d sums | count]
            if (v && (e < d || e == EPL)) atomicAdd(&rep[kk * (d + 1) + (e < d ? e : d)], (unsigned long long)v);
        }
    }
}

// Adds the NREP replica tables of the fused accumulation into the caller's sums / counts.
__global__ __launch_bounds__(256) void k_reduce_replicas(const unsigned long long *__restrict__ rep, int K, int d,
                                                         unsigned long long *__restrict__ sums,
                                                         unsigned long long *__restrict__ counts, Gate gate) {
    if (gate_closed(gate)) return;
    const int tabn = K * (d + 1);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tabn) return;
    unsigned long long t = 0;
#pragma unroll 8
    for (int r = 0; r < NREP; ++r) t += rep[(size_t)r * tabn + i];
    if (!t) return;
    const int kk = i / (d + 1), e = i - kk * (d + 1);
    if (e < d) sums[(size_t)kk * d + e] += t;           // no other kernel touches sums / counts while this one runs
    else counts[kk] += t;
}

// Fix-up pass of FAST mode: every queued vector is searched again by a whole wave in sequential
// fp64 (exact_search_wave); its label, its distance and -- when sums is non-null -- its
// contribution to the cluster sums are written here.  Waves stride over the queue, so the work is
// balanced however the ambiguous vectors cluster in the image; results do not depend on queue order.
// A wave fetches the queue entries and vector rows of four items at once (two memory round trips
// per four items); with LDS_CB the transposed codebook is staged in LDS once per workgroup.
template <bool LDS_CB>
__global__ __launch_bounds__(1024) void k_fixup(const uint8_t *__restrict__ blocks, const uint16_t *__restrict__ perm,
                                                int64_t N, int d, const double *__restrict__ cbT, int K,
                                                const uint32_t *__restrict__ queue,
                                                const uint32_t *__restrict__ qcount, uint32_t qcap,
                                                int32_t *__restrict__ labels, double *__restrict__ min_dist,
                                                unsigned long long *__restrict__ sums,
                                                unsigned long long *__restrict__ counts, Gate gate) {
    extern __shared__ __align__(16) double s_cbT[];
    if (gate_closed(gate)) return;
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t waves_per_wg = blockDim.x >> 6;
    const uint32_t wave = blockIdx.x * waves_per_wg + (threadIdx.x >> 6);
    // wave w serves sub-queue w % NQ, as the (w / NQ)-th of the nwaves / NQ waves that share it (the
    // host launches a multiple of NQ waves)
    const uint32_t nwaves = gridDim.x * waves_per_wg / NQ;
    const uint32_t sub = wave & (NQ - 1);
    const uint32_t rank = wave / NQ;
    // nothing for this workgroup (uniform): none of its waves' sub-queues reaches their rank
    bool any = false;
    for (uint32_t w = 0; w < waves_per_wg; ++w) {
        const uint32_t ww = blockIdx.x * waves_per_wg + w;
        any = any || qcount[16 * (ww & (NQ - 1))] > ww / NQ;
    }
    if (!any) return;
    if (LDS_CB) {
        const int n = K * d;
        for (int i = threadIdx.x; i < n; i += blockDim.x) s_cbT[i] = cbT[i];
        __syncthreads();
    }
    const uint32_t count = min(qcount[16 * sub], qcap);       // a counter nobody reset must not lead past the sub-queue
    queue += (size_t)sub * qcap;
    for (uint32_t i0 = rank; i0 < count; i0 += 4 * nwaves) {
        int64_t n[4], m[4];                                    // row to read; vector index to write
        uint32_t xv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t i = i0 + t * nwaves;
            n[t] = (int64_t)queue[i < count ? i : count - 1];  // wave-uniform
            n[t] = n[t] < N ? n[t] : N - 1;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            m[t] = n[t];
            if (perm) {                                        // `blocks` is the sorted copy: origin inside the window
                m[t] = (n[t] & ~(int64_t)(SORT_WIN - 1)) + (int64_t)perm[n[t]];
                m[t] = m[t] < N ? m[t] : N - 1;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) xv[t] = lane < d ? (uint32_t)blocks[(size_t)n[t] * d + lane] : 0u;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (i0 + t * nwaves >= count) break;
            const uint8_t *xrow = blocks + (size_t)n[t] * d;
            int ri;
            double rd;
            exact_search_wave<LDS_CB>(xv[t], xrow, d, cbT, s_cbT, K, lane, ri, rd);
            if (lane == 0) {
                labels[m[t]] = ri;
                if (min_dist) min_dist[m[t]] = rd;
            }
            if (sums) {
                for (int e = lane; e < d; e += WAVE) {
                    const unsigned v = xrow[e];
                    if (v) atomicAdd(&sums[(size_t)ri * d + e], (unsigned long long)v);
                }
                if (lane == 0) atomicAdd(&counts[ri], 1ull);
            }
        }
    }
}

// The fix-up pass with the element-sum bound of the pruned search (needs the sorted fixed-point
// codebook that the prepare kernels leave for K <= PREP_SORT_CAP; this kernel serves codebooks whose
// transposed fp64 copy fits the LDS, k_fixup_shortlist the larger ones).  A wave takes one queued vector: lanes hold
// 256 consecutive positions of the sorted codebook (four per lane), starting with those around the
// vector's own element sum; after every pass the exact best distance D gives the range of positions
// that can still hold a codeword at distance <= D (their fixed-point element sum lies within
// D * 2^s + Emax + 1 of the vector's, looked up in postab), and the wave takes the next 256 on
// whichever side is not covered yet.  A skipped codeword is strictly further away than the winner,
// so label and distance equal those of the full search, ties included (lowest codeword index).
__global__ __launch_bounds__(1024) void k_fixup_pruned(const uint8_t *__restrict__ blocks,
                                                       const uint16_t *__restrict__ perm, int64_t N, int d,
                                                       const double *__restrict__ cbT, int K,
                                                       const uint32_t *__restrict__ queue,
                                                       const uint32_t *__restrict__ qcount, uint32_t qcap,
                                                       const PrepHeader *__restrict__ hdr,
                                                       const uint32_t *bias, const uint32_t *postab,
                                                       int32_t *__restrict__ labels, double *__restrict__ min_dist,
                                                       unsigned long long *__restrict__ sums,
                                                       unsigned long long *__restrict__ counts, Gate gate) {
    extern __shared__ __align__(16) double s_cbT[];
    if (gate_closed(gate)) return;
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t waves_per_wg = blockDim.x >> 6;
    const uint32_t wave = blockIdx.x * waves_per_wg + (threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * waves_per_wg / NQ;
    const uint32_t sub = wave & (NQ - 1);
    const uint32_t rank = wave / NQ;
    bool any = false;
    for (uint32_t w = 0; w < waves_per_wg; ++w) {
        const uint32_t ww = blockIdx.x * waves_per_wg + w;
        any = any || qcount[16 * (ww & (NQ - 1))] > ww / NQ;
    }
    if (!any) return;
    // the transposed codebook, the position table and the position -> codeword map are staged in LDS, so that a
    // queued vector costs no memory round trip beyond fetching the vector itself
    const uint32_t *s_postab = reinterpret_cast<const uint32_t *>(s_cbT + (size_t)K * d);
    const uint32_t *s_bias = s_postab + POSTAB_WORDS;
    {
        const int n = K * d;
        for (int i = threadIdx.x; i < n; i += blockDim.x) s_cbT[i] = cbT[i];
        uint32_t *wp = reinterpret_cast<uint32_t *>(s_cbT + (size_t)K * d);
        for (int i = threadIdx.x; i < POSTAB_WORDS; i += blockDim.x) wp[i] = postab[i];
        for (int i = threadIdx.x; i < K; i += blockDim.x) wp[POSTAB_WORDS + i] = bias[i];
        __syncthreads();
        postab = s_postab;
        bias = s_bias;
    }
    const uint32_t shift = hdr->shift, emax = hdr->emax, tshift = hdr->tshift, nt = hdr->nt;
    const double scale = (double)(1u << shift);
    const uint32_t count = min(qcount[16 * sub], qcap);       // a counter nobody reset must not lead past the sub-queue
    queue += (size_t)sub * qcap;
    // A wave serves the entries rank, rank + nwaves, ... of its sub-queue.  It reads them 64 at a time (one entry
    // per lane, one load), so that an item costs one memory round trip -- the vector's bytes and its origin, which
    // depend on the entry alone -- and requests that round trip two items ahead of the one it is searching.
    struct Item {
        int64_t nrow, nout;         // position of the row in `blocks`; vector index the results are written at
        uint32_t xv, part;          // the lane's row byte; the lane's share of the element sum
    };
    auto fetch = [=](uint32_t entry) {
        Item it;
        it.nrow = (int64_t)entry;                               // wave-uniform
        it.nrow = it.nrow < N ? it.nrow : N - 1;
        it.nout = it.nrow;
        if (perm) {                                             // `blocks` is the sorted copy: origin inside the window
            it.nout = (it.nrow & ~(int64_t)(SORT_WIN - 1)) + (int64_t)perm[it.nrow];
            it.nout = it.nout < N ? it.nout : N - 1;
        }
        const uint8_t *xr = blocks + (size_t)it.nrow * d;
        it.xv = lane < d ? (uint32_t)xr[lane] : 0u;
        it.part = it.xv;
        for (int e = lane + WAVE; e < d; e += WAVE) it.part += xr[e];
        return it;
    };
    const uint32_t mine = count > rank ? (count - rank + nwaves - 1) / nwaves : 0u;     // entries this wave serves
    for (uint32_t c0 = 0; c0 < mine; c0 += WAVE) {
        const uint32_t in_chunk = mine - c0 < (uint32_t)WAVE ? mine - c0 : (uint32_t)WAVE;
        const uint32_t qidx = rank + (c0 + (uint32_t)lane) * nwaves;
        const uint32_t entries = queue[(uint32_t)lane < in_chunk ? qidx : rank];            // (rank < count here)
        Item it_a = fetch(__builtin_amdgcn_readlane(entries, 0));
        Item it_b = fetch(__builtin_amdgcn_readlane(entries, in_chunk > 1 ? 1 : 0));
        for (uint32_t j = 0; j < in_chunk; ++j) {
            const Item cur = it_a;
            it_a = it_b;
            it_b = fetch(__builtin_amdgcn_readlane(entries, j + 2 < in_chunk ? (int)(j + 2) : 0));
            const int64_t nrow = cur.nrow, n = cur.nout;
            const uint32_t xv0 = cur.xv;
            uint32_t part = cur.part;
            const uint8_t *xrow = blocks + (size_t)nrow * d;
            // element sum of the vector in the fixed-point units of the codebook rows
    #pragma unroll
            for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, WAVE);
            const uint32_t sx = part << shift;
            double best = __builtin_huge_val();
            int bidx = 0x7fffffff;
            // one pass: 256 consecutive positions of the sorted codebook from p0 on (lane's positions p0 + 64 q +
            // lane, q = 0 .. 3; those outside [0, K) do not count), codeword bias[p], exact distance in
            // sequential fp64.  Loads are issued eight elements x four codewords at a time; the additions
            // stay in element order, one chain per codeword.
            auto pass = [&](int p0) {
                int kk[4], pp[4];                                   // codeword index; (clamped) sorted position
                bool on[4];
    #pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int pq = p0 + q * WAVE + lane;
                    on[q] = pq >= 0 && pq < K;
                    pp[q] = pq < 0 ? 0 : (pq < K ? pq : K - 1);       // clamped: loads stay in bounds
                    kk[q] = (int)(bias[pp[q]] & 0x7fffffffu);
                }
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                for (int ebase = 0; ebase < d; ebase += WAVE) {
                    const int ecount = d - ebase < WAVE ? d - ebase : WAVE;
                    uint32_t xv = xv0;
                    if (ebase > 0) xv = lane < ecount ? (uint32_t)xrow[ebase + lane] : 0u;
                    int e0 = 0;
                    for (; e0 + 8 <= ecount; e0 += 8) {
                        double c[8][4];
    #pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const size_t off = (size_t)(ebase + e0 + j) * K;
    #pragma unroll
                            for (int q = 0; q < 4; ++q) c[j][q] = s_cbT[off + pp[q]];
                        }
    #pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const double xe = (double)__builtin_amdgcn_readlane(xv, e0 + j);
    #pragma unroll
                            for (int q = 0; q < 4; ++q) acc[q] += fabs(xe - c[j][q]);
                        }
                    }
                    for (; e0 < ecount; ++e0) {
                        const size_t off = (size_t)(ebase + e0) * K;
                        const double xe = (double)__builtin_amdgcn_readlane(xv, e0);
    #pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q] += fabs(xe - (s_cbT[off + pp[q]]));
                    }
                }
                double cd = __builtin_huge_val();
                int ci = 0x7fffffff;
    #pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (on[q] && (acc[q] < cd || (acc[q] == cd && kk[q] < ci))) {
                        cd = acc[q];
                        ci = kk[q];
                    }
                }
                wave_argmin(cd, ci);                                // lexicographic (distance, codeword index)
                if (cd < best || (cd == best && ci < bidx)) {
                    best = cd;
                    bidx = ci;
                }
            };
            constexpr int PASS = 4 * WAVE;
            int lo_done = (int)postab[sx >> tshift] - PASS / 2;     // first pass: the 256 positions around the vector's sum
            if (lo_done > K - PASS) lo_done = K - PASS;
            if (lo_done < 0) lo_done = 0;
            int hi_done = lo_done + PASS;
            pass(lo_done);
            for (;;) {
                // positions that can still hold a codeword no further away than `best`
                const double t = ceil(best * scale) + (double)emax + 1.0;
                const uint32_t thr = t < 4.0e9 ? (uint32_t)t : 0xffffffffu;
                const uint32_t lo_sum = sx > thr ? sx - thr : 0u;
                unsigned long long ih = (((unsigned long long)sx + thr) >> tshift) + 1ull;
                if (ih > nt + 1ull) ih = nt + 1ull;
                const int klo = (int)postab[lo_sum >> tshift];
                const int khi = (int)postab[ih];
                if (klo < lo_done) {
                    lo_done -= PASS;
                    pass(lo_done);
                } else if (khi > hi_done) {
                    pass(hi_done);
                    hi_done += PASS;
                } else {
                    break;
                }
            }
            if (lane == 0) {
                labels[n] = bidx;
                if (min_dist) min_dist[n] = best;
            }
            if (sums) {
                for (int e = lane; e < d; e += WAVE) {
                    const unsigned v = xrow[e];
                    if (v) atomicAdd(&sums[(size_t)bidx * d + e], (unsigned long long)v);
                }
                if (lane == 0) atomicAdd(&counts[bidx], 1ull);
            }
        }
    }
}

// The fix-up of a queued vector for sorted codebooks beyond the LDS, without reading the fp64 codebook for more
// than a handful of rows (measuring every searched row in fp64 reads 8 d bytes per row and is bound by the L2).  The
// workgroup (four waves) repeats the filter's search for this one vector with lanes over codewords -- the
// fixed-point SAD of every sorted position whose element sum lies within (smallest SAD so far + ambiguity
// window) of the vector's, 1024 positions per round, read from the transposed fixed-point rows (coalesced,
// d bytes per row), the SADs kept in LDS -- which yields the exact smallest SAD and with it the complete list
// of rows that can be the fp64 argmin (SAD within the window of the smallest: |T_k - S_k| <= Emax for every
// row).  Only those rows are measured in fp64: the terms |x_e - c_e| in parallel, their sum in element order.
// The lowest codeword index wins ties.  A list that overflows (more than FIX_LIST rows within the window:
// constructed inputs) makes the workgroup measure the whole searched range.
constexpr int FIX_LIST = 128;
__global__ __launch_bounds__(256) void k_fixup_shortlist(const uint8_t *__restrict__ blocks,
                                                         const uint16_t *__restrict__ perm, int64_t N, int d, int dp2,
                                                         const double *__restrict__ cbrows, const uint32_t *__restrict__ c16T,
                                                         int K, const uint32_t *__restrict__ queue,
                                                         const uint32_t *__restrict__ qcount, uint32_t qcap,
                                                         const PrepHeader *__restrict__ hdr,
                                                         const uint32_t *__restrict__ bias,
                                                         const uint32_t *__restrict__ postab,
                                                         int32_t *__restrict__ labels, double *__restrict__ min_dist,
                                                         unsigned long long *__restrict__ sums,
                                                         unsigned long long *__restrict__ counts, Gate gate) {
    __shared__ uint32_t s_sad[PREP_SORT_CAP];               // SAD of every searched position
    __shared__ uint32_t s_xw[MAX_D_FAST / 2];               // the vector in the fixed point of the codebook rows
    __shared__ double s_xd[MAX_D_FAST];                     // the vector as doubles
    __shared__ double s_term[4 * MAX_D_FAST];               // per wave: the terms of one distance
    __shared__ uint32_t s_min[2][4];
    __shared__ uint32_t s_list[FIX_LIST];
    __shared__ uint32_t s_nlist;
    __shared__ double s_cd[4];
    __shared__ int s_ci[4];
    if (gate_closed(gate)) return;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
    const uint32_t sub = blockIdx.x & (NQ - 1), rank = blockIdx.x / NQ, nwg = gridDim.x / NQ;
    const uint32_t count = min(qcount[16 * sub], qcap);       // a counter nobody reset must not lead past the sub-queue
    if (rank >= count) return;                                  // (workgroup-uniform)
    const uint32_t shift = hdr->shift, emax = hdr->emax, tshift = hdr->tshift, nt = hdr->nt;
    const uint32_t window = emax >= 0x3fffffffu ? 0xffffffffu : 2u * emax;
    queue += (size_t)sub * qcap;
    constexpr int PASS = 4 * WAVE, ROUND = 4 * PASS;
    int parity = 0;
    for (uint32_t i = rank; i < count; i += nwg) {
        int64_t nrow = (int64_t)queue[i];
        nrow = nrow < N ? nrow : N - 1;
        int64_t n = nrow;                                       // vector index the results are written at
        if (perm) {                                             // `blocks` is the sorted copy: origin inside the window
            n = (nrow & ~(int64_t)(SORT_WIN - 1)) + (int64_t)perm[nrow];
            n = n < N ? n : N - 1;
        }
        const uint8_t *xrow = blocks + (size_t)nrow * d;
        __syncthreads();                                        // the previous vector's LDS state is no longer read
        for (int p = tid; p < dp2; p += 256) {
            const uint32_t b0 = 2 * p < d ? xrow[2 * p] : 0u, b1 = 2 * p + 1 < d ? xrow[2 * p + 1] : 0u;
            s_xw[p] = (b0 | (b1 << 16)) << shift;
        }
        for (int e = tid; e < d; e += 256) s_xd[e] = (double)xrow[e];
        if (tid == 0) s_nlist = 0u;
        __syncthreads();
        uint32_t sx = 0;                                        // element sum in the units of the codebook rows
        for (int p = 0; p < dp2; ++p) sx = __builtin_amdgcn_sad_u16(s_xw[p], 0u, sx);
        sx = __builtin_amdgcn_readfirstlane(sx);
        uint32_t smin = 0xffffffffu;                            // smallest SAD so far (workgroup-uniform)
        // one round: wave w takes the 256 positions from p0w on (four per lane; those outside [0, K) do not count)
        auto round = [&](int p0w) {
            int pc[4];                                          // (clamped) positions: loads stay in bounds
            bool on[4];
            uint32_t sad[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pos = p0w + q * WAVE + lane;
                on[q] = pos >= 0 && pos < K;
                pc[q] = pos < 0 ? 0 : (pos < K ? pos : K - 1);
            }
            if (p0w < K && p0w + PASS > 0) {                    // (wave-uniform) a chunk wholly outside is skipped
                int p = 0;
                for (; p + 8 <= dp2; p += 8) {                  // 32 loads in flight, then the SADs
                    uint32_t c[8][4];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) c[j][q] = c16T[(size_t)(p + j) * K + pc[q]];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t x = s_xw[p + j];
#pragma unroll
                        for (int q = 0; q < 4; ++q) sad[q] = __builtin_amdgcn_sad_u16(c[j][q], x, sad[q]);
                    }
                }
                for (; p < dp2; ++p) {
                    const uint32_t x = s_xw[p];
#pragma unroll
                    for (int q = 0; q < 4; ++q) sad[q] = __builtin_amdgcn_sad_u16(c16T[(size_t)p * K + pc[q]], x, sad[q]);
                }
            }
            uint32_t wmin = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (on[q]) {
                    s_sad[pc[q]] = sad[q];
                    wmin = sad[q] < wmin ? sad[q] : wmin;
                }
            wmin = wave_reduce_u32<false>(wmin);
            if (lane == 0) s_min[parity][wv] = wmin;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < 4; ++w) smin = min(smin, s_min[parity][w]);
            parity ^= 1;                                        // the other buffer next time: no second barrier
        };
        int lo_done = (int)postab[sx >> tshift] - ROUND / 2;     // first round: the 1024 positions around the vector's sum
        if (lo_done > K - ROUND) lo_done = K - ROUND;
        if (lo_done < 0) lo_done = 0;
        int hi_done = lo_done + ROUND;
        round(lo_done + wv * PASS);
        uint32_t lim;
        for (;;) {
            // positions that can still hold a row within the window of the smallest SAD
            lim = smin + window < smin ? 0xffffffffu : smin + window;
            const uint32_t lo_sum = sx > lim ? sx - lim : 0u;
            unsigned long long ih = (((unsigned long long)sx + lim) >> tshift) + 1ull;
            if (ih > nt + 1ull) ih = nt + 1ull;
            const int klo = (int)postab[lo_sum >> tshift];
            const int khi = (int)postab[ih];
            const bool need_lo = klo < lo_done, need_hi = khi > hi_done;
            if (!need_lo && !need_hi) break;
            int p0w;
            if (need_lo && need_hi) {                           // two chunks on each side
                p0w = wv < 2 ? lo_done - (wv + 1) * PASS : hi_done + (wv - 2) * PASS;
                lo_done -= 2 * PASS;
                hi_done += 2 * PASS;
            } else if (need_lo) {
                p0w = lo_done - (wv + 1) * PASS;
                lo_done -= ROUND;
            } else {
                p0w = hi_done + wv * PASS;
                hi_done += ROUND;
            }
            round(p0w);
        }
        const int a0 = lo_done > 0 ? lo_done : 0, a1 = hi_done < K ? hi_done : K;       // searched positions
        // the rows within the window of the smallest SAD
        for (int pos = a0 + tid; pos < a1; pos += 256)
            if (s_sad[pos] <= lim) {
                const uint32_t slot = atomicAdd(&s_nlist, 1u);
                if (slot < (uint32_t)FIX_LIST) s_list[slot] = (uint32_t)pos;
            }
        __syncthreads();
        const uint32_t nl = s_nlist;
        const bool whole = nl > (uint32_t)FIX_LIST;             // overflow: every searched row is measured
        const uint32_t ncand = whole ? (uint32_t)(a1 - a0) : nl;
        double best = __builtin_huge_val();
        int bidx = 0x7fffffff;
        double *term = s_term + wv * MAX_D_FAST;
        for (uint32_t c = wv; c < ncand; c += 4) {              // (wave-uniform) one row per wave and trip
            const int pos = whole ? a0 + (int)c : (int)s_list[c];
            const int kk = (int)(bias[pos] & 0x7fffffffu);
            const double *crow = cbrows + (size_t)kk * d;
            // the terms |x_e - c_e| in parallel (lanes over elements, one coalesced read of the row), then their
            // sum in element order from LDS: the additions of the sequential loop without its round trips
            for (int e = lane; e < d; e += WAVE) term[e] = fabs(s_xd[e] - crow[e]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            double acc = 0.0;
            int e = 0;
            for (; e + 8 <= d; e += 8) {
                double t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = term[e + j];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += t[j];
            }
            for (; e < d; ++e) acc += term[e];                  // the same value in every lane
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (acc < best || (acc == best && kk < bidx)) {
                best = acc;
                bidx = kk;
            }
        }
        if (lane == 0) {
            s_cd[wv] = best;
            s_ci[wv] = bidx;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const double od = s_cd[w];
            const int oi = s_ci[w];
            if (od < best || (od == best && oi < bidx)) {
                best = od;
                bidx = oi;
            }
        }
        if (tid == 0) {
            labels[n] = bidx;
            if (min_dist) min_dist[n] = best;
        }
        if (sums && wv == 0) {
            for (int e = lane; e < d; e += WAVE) {
                const unsigned v = xrow[e];
                if (v) atomicAdd(&sums[(size_t)bidx * d + e], (unsigned long long)v);
            }
            if (lane == 0) atomicAdd(&counts[bidx], 1ull);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// numpy-order sum of an fp64 vector (the np.mean of VQ_Full.py:90 before its division)
// ------------------------------------------------------------------------------------------------
// numpy's pairwise routine on one run of n <= 128 elements.
__device__ __forceinline__ double np_leaf_sum(const double *__restrict__ a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
#pragma unroll 4
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

// One 512-thread workgroup per full 8192-element chunk.  A leaf is numpy's run of 128 elements:
// eight interleaved partial sums r[j] = a[j] + a[8+j] + ... (16 terms, in order), combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)).  Eight lanes own one leaf (lane j owns r[j]); the combine is
// the xor-1,2,4 shuffle tree.  The recursive halving above the leaves is a balanced binary tree
// over adjacent leaves: the xor tree over the 64 leaf sums.  (All 512 threads of the workgroup call.)
__device__ __forceinline__ void npsum_chunk(const double *__restrict__ v, int64_t c, double *__restrict__ chunk_sums,
                                            double *s_leaf) {
    const int tid = threadIdx.x;
    const int j = tid & 7, leaf = tid >> 3;                    // 64 leaves x 8 lanes
    const double *a = v + c * NP_CHUNK + (int64_t)leaf * 128 + j;
    double t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = a[8 * i];
    double r = t[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) r += t[i];
    r += __shfl_xor(r, 1, WAVE);
    r += __shfl_xor(r, 2, WAVE);
    r += __shfl_xor(r, 4, WAVE);
    if (j == 0) s_leaf[leaf] = r;
    __syncthreads();
    if (tid < WAVE) {
        double u = s_leaf[tid];
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) u += __shfl_xor(u, off, WAVE);
        if (tid == 0) chunk_sums[c] = u;
    }
    __syncthreads();
}

__global__ __launch_bounds__(512) void k_npsum_chunks(const double *__restrict__ v, int64_t nfull,
                                                      double *__restrict__ chunk_sums) {
    __shared__ double s_leaf[64];
    for (int64_t c = blockIdx.x; c < nfull; c += gridDim.x) npsum_chunk(v, c, chunk_sums, s_leaf);
}

// total + s[0] + s[1] + ... + s[cnt - 1], added strictly left to right by the calling thread (s in LDS).  The values
// are fetched eight at a time ahead of the additions that use them: one LDS round trip per addition otherwise.
__device__ __forceinline__ double add_left_to_right(double total, const double *s, int cnt) {
    int j = 0;
    for (; j + 8 <= cnt; j += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = s[j + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) total += v[k];
    }
    for (; j < cnt; ++j) total += s[j];
    return total;
}

// One workgroup of BLOCK threads: the trailing partial chunk (m = N mod 8192 elements, numpy's uneven recursive
// halving reproduced with an explicit stack), then the left-to-right sum over all chunk sums, returned in
// thread 0.
template <int BLOCK>
__device__ __forceinline__ double npsum_final_block(const double *__restrict__ v, int64_t N, int64_t nfull,
                                                    double *chunk_sums) {
    static_assert(BLOCK >= 160, "one thread per leaf of the partial chunk");
    __shared__ int s_off[160], s_len[160];
    __shared__ double s_leaf[160];
    __shared__ int s_nleaf;
    __shared__ double s_buf[BLOCK];
    __shared__ int s_sn[20], s_sstate[20];
    __shared__ double s_sleft[20];
    const int m = (int)(N - nfull * NP_CHUNK);
    const double *tail = v + nfull * NP_CHUNK;
    if (m > 0) {
        if (threadIdx.x == 0) {
            // leaves of the recursion, left to right
            int sp = 0, cnt = 0;
            s_sn[sp] = m; s_sstate[sp] = 0; ++sp;               // here s_sstate holds the run's offset
            while (sp > 0) {
                --sp;
                int n = s_sn[sp], off = s_sstate[sp];
                if (n <= 128) {
                    s_off[cnt] = off; s_len[cnt] = n; ++cnt;
                } else {
                    int n2 = n / 2;
                    n2 -= n2 % 8;
                    s_sn[sp] = n - n2; s_sstate[sp] = off + n2; ++sp;   // right half: popped second
                    s_sn[sp] = n2; s_sstate[sp] = off; ++sp;            // left half: popped first
                }
            }
            s_nleaf = cnt;
        }
        __syncthreads();
        if ((int)threadIdx.x < s_nleaf) s_leaf[threadIdx.x] = np_leaf_sum(tail + s_off[threadIdx.x], s_len[threadIdx.x]);
        __syncthreads();
        if (threadIdx.x == 0) {
            // evaluate pairwise(m) = pairwise(left) + pairwise(right) over the leaf sums
            int sp = 0, next = 0;
            double ret = 0.0;
            s_sn[sp] = m; s_sstate[sp] = 0; ++sp;
            while (sp > 0) {
                const int f = sp - 1;
                const int n = s_sn[f];
                if (n <= 128) {
                    ret = s_leaf[next++];
                    --sp;
                    continue;
                }
                int n2 = n / 2;
                n2 -= n2 % 8;
                if (s_sstate[f] == 0) {
                    s_sstate[f] = 1;
                    s_sn[sp] = n2; s_sstate[sp] = 0; ++sp;
                } else if (s_sstate[f] == 1) {
                    s_sleft[f] = ret;
                    s_sstate[f] = 2;
                    s_sn[sp] = n - n2; s_sstate[sp] = 0; ++sp;
                } else {
                    ret = s_sleft[f] + ret;
                    --sp;
                }
            }
            chunk_sums[nfull] = ret;
            s_leaf[0] = ret;                                    // (kept for the sum below)
        }
        __syncthreads();
    }
    // left-to-right sum of the chunk sums, staged through LDS BLOCK at a time
    const int64_t nchunks = nfull + (m > 0 ? 1 : 0);
    const double *cs = chunk_sums;
    double total = 0.0;
    for (int64_t lo = 0; lo < nchunks; lo += BLOCK) {
        const int64_t i = lo + threadIdx.x;
        s_buf[threadIdx.x] = i < nfull ? cs[i] : (i < nchunks ? s_leaf[0] : 0.0);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int cnt = (int)(nchunks - lo < BLOCK ? nchunks - lo : BLOCK);
            total = add_left_to_right(total, s_buf, cnt);
        }
        __syncthreads();
    }
    return total;
}

__global__ __launch_bounds__(1024) void k_npsum_final(const double *__restrict__ v, int64_t N, int64_t nfull,
                                                      double *__restrict__ chunk_sums, double *__restrict__ out) {
    const double total = npsum_final_block<1024>(v, N, nfull, chunk_sums);
    if (threadIdx.x == 0) out[0] = total;
}

// The tail of an LBG iteration in two launches (everything of VQ_Full.py:81-98 that follows the assignment when
// the search accumulated into the replica tables).  k_lbg_fold: every workgroup folds its share of the replica
// tables into sums / counts -- leaving the replicas zeroed for the next iteration -- and sums its 8192-element
// chunks of the winning distances.  k_lbg_finish_t (one workgroup): the chunk sums in numpy's order, the centroid
// update in place and the stop rule.
// perm non-null: the distances of the full chunks lie in the order of the window-sorted vectors (a chunk is a sort
// window); the workgroup puts a chunk back into the caller's order in LDS (64 KB) and sums it from there.
static_assert(SORT_WIN == NP_CHUNK, "a sort window is one summation chunk");
__device__ __forceinline__ void npsum_chunk_sorted(const double *__restrict__ v, const uint16_t *__restrict__ perm, int64_t c,
                                                   double *__restrict__ chunk_sums, double *s_leaf, double *s_d) {
    const int tid = threadIdx.x;
    const double *src = v + c * NP_CHUNK;
    const uint16_t *pp = perm + c * NP_CHUNK;
#pragma unroll 4
    for (int i = tid; i < (int)NP_CHUNK; i += 512) s_d[pp[i] & (NP_CHUNK - 1)] = src[i];
    __syncthreads();
    const int j = tid & 7, leaf = tid >> 3;                    // 64 leaves x 8 lanes, as in npsum_chunk
    const double *a = s_d + leaf * 128 + j;
    double r = a[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) r += a[8 * i];
    r += __shfl_xor(r, 1, WAVE);
    r += __shfl_xor(r, 2, WAVE);
    r += __shfl_xor(r, 4, WAVE);
    if (j == 0) s_leaf[leaf] = r;
    __syncthreads();
    if (tid < WAVE) {
        double u = s_leaf[tid];
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) u += __shfl_xor(u, off, WAVE);
        if (tid == 0) chunk_sums[c] = u;
    }
    __syncthreads();
}

__global__ __launch_bounds__(512) void k_lbg_fold(const double *__restrict__ min_dist, const uint16_t *__restrict__ perm,
                                                  int64_t nfull,
                                                  double *__restrict__ chunk_sums, unsigned long long *__restrict__ rep,
                                                  int K, int d, unsigned long long *__restrict__ sums,
                                                  unsigned long long *__restrict__ counts, uint32_t *__restrict__ qcount,
                                                  Gate g) {
    extern __shared__ __align__(16) double s_dist[];            // perm non-null: one chunk of distances
    __shared__ double s_leaf[64];
    if (gate_closed(g)) return;
    const int tid = threadIdx.x;
    const int tabn = K * (d + 1);
    if (blockIdx.x == 0 && tid < 8) qcount[16 * tid + 4] = 0u;   // the tile counters of the resident search workgroups
    for (int i = blockIdx.x * 512 + tid; i < tabn; i += gridDim.x * 512) {
        unsigned long long t = 0;
#pragma unroll 8
        for (int r = 0; r < NREP; ++r) {
            t += rep[(size_t)r * tabn + i];
            rep[(size_t)r * tabn + i] = 0ull;
        }
        const int kk = i / (d + 1), e = i - kk * (d + 1);
        if (e < d) sums[(size_t)kk * d + e] = t;
        else counts[kk] = t;
    }
    if (perm) {
        for (int64_t c = blockIdx.x; c < nfull; c += gridDim.x) npsum_chunk_sorted(min_dist, perm, c, chunk_sums, s_leaf, s_dist);
    } else {
        for (int64_t c = blockIdx.x; c < nfull; c += gridDim.x) npsum_chunk(min_dist, c, chunk_sums, s_leaf);
    }
}

// TPC > 0: the same workgroup then prepares the workspace for the updated codebook (prepare_small_body), which the
// next iteration of the stage would otherwise do in a launch of its own.
struct PrepArgs {
    int dp2;
    uint32_t clim;
    PrepHeader *hdr;
    uint32_t *c16, *bias, *postab;
    double *cbT;
};
template <int TPC>
__global__ __launch_bounds__(1024) void k_lbg_finish_t(const double *__restrict__ min_dist, int64_t N, int64_t nfull,
                                                       double *__restrict__ chunk_sums,
                                                       const unsigned long long *__restrict__ sums,
                                                       const unsigned long long *__restrict__ counts, double *cb, int K,
                                                       int d, double n_total, double threshold, LbgState *state, Gate g,
                                                       const PrepArgs pa) {
    extern __shared__ __align__(16) unsigned long long s_prepkeys[];
    if (gate_closed(g)) return;
    const double total = npsum_final_block<1024>(min_dist, N, nfull, chunk_sums);
    for (int i = threadIdx.x; i < K * d; i += 1024) {
        const unsigned long long c = counts[i / d];
        if (c) cb[i] = (double)sums[i] / (double)c;
    }
    if (threadIdx.x == 0) {
        state->dist_sum = total;
        apply_stop_rule(state, total, n_total, threshold, g.iter);
    }
    if constexpr (TPC > 0) {
        __syncthreads();                                       // the updated codebook is visible to the whole workgroup
        prepare_small_body<TPC>(cb, K, d, pa.dp2, pa.clim, pa.hdr, pa.c16, pa.bias, pa.cbT, pa.postab, s_prepkeys);
    }
}

// The tail of an LBG iteration whose vectors are sharded over several ranks (one workgroup): the ranks' per-chunk
// distance sums -- rank r's count[r] values at chunks_all[r * stride] -- added left to right in rank order (what one
// process computes on the concatenated vectors when every shard but the last is a whole number of chunks), then the
// centroid update in place from the sums / counts (already reduced, or one record per rank) and the stop rule
// (VQ_Full.py:81-98); TPC > 0: then the codebook preparation of the next iteration, as in k_lbg_finish_t.
constexpr int MAX_RANKS = 64;
struct RankChunks {
    int world;
    int64_t stride;
    int count[MAX_RANKS];
};
template <int TPC>
__global__ __launch_bounds__(1024) void k_lbg_finish_ranks_t(const double *__restrict__ chunks_all, const RankChunks rc,
                                                             const unsigned long long *__restrict__ sums,
                                                             const unsigned long long *__restrict__ counts, int records,
                                                             int64_t record_stride, double *cb, int K, int d, double n_total,
                                                             double threshold, LbgState *state, Gate g, PrepArgs pa) {
    extern __shared__ __align__(16) unsigned long long s_prepkeys[];
    __shared__ double s_buf[1024];
    if (gate_closed(g)) return;
    double total = 0.0;
    for (int r = 0; r < rc.world; ++r) {
        const double *cs = chunks_all + (size_t)r * rc.stride;
        for (int lo = 0; lo < rc.count[r]; lo += 1024) {
            const int i = lo + (int)threadIdx.x;
            s_buf[threadIdx.x] = i < rc.count[r] ? cs[i] : 0.0;
            __syncthreads();
            if (threadIdx.x == 0) {
                const int cnt = rc.count[r] - lo < 1024 ? rc.count[r] - lo : 1024;
                total = add_left_to_right(total, s_buf, cnt);
            }
            __syncthreads();
        }
    }
    // the ranks' integer sums and counts (records > 1: as an all-gather left them, record_stride words apart)
    for (int i = threadIdx.x; i < K * d; i += 1024) {
        unsigned long long c = 0, sm = 0;
        for (int r = 0; r < records; ++r) {
            c += counts[(size_t)r * record_stride + i / d];
            sm += sums[(size_t)r * record_stride + i];
        }
        if (c) cb[i] = (double)sm / (double)c;
    }
    if (threadIdx.x == 0) {
        state->dist_sum = total;
        apply_stop_rule(state, total, n_total, threshold, g.iter);
    }
    if constexpr (TPC > 0) {
        __syncthreads();                                       // the updated codebook is visible to the whole workgroup
        prepare_small_body<TPC>(cb, K, d, pa.dp2, pa.clim, pa.hdr, pa.c16, pa.bias, pa.cbT, pa.postab, s_prepkeys);
    }
}

// ------------------------------------------------------------------------------------------------
// nearest codeword, EXACT mode: every distance in sequential fp64, one vector per lane
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assign_exact(const uint8_t *__restrict__ blocks, int64_t N, int d,
                                                      const double *__restrict__ cb, int K,
                                                      int32_t *__restrict__ labels, double *__restrict__ min_dist,
                                                      Gate gate) {
    if (gate_closed(gate)) return;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const uint8_t *x = blocks + (size_t)n * d;
        double best = __builtin_huge_val();
        int bidx = 0;
        for (int k = 0; k < K; ++k) {
            const double *c = cb + (size_t)k * d;
            double acc = 0.0;
            for (int e = 0; e < d; ++e) acc += fabs((double)x[e] - c[e]);
            if (acc < best) {
                best = acc;
                bidx = k;
            }
        }
        if (labels) labels[n] = bidx;
        if (min_dist) min_dist[n] = best;
    }
}

// Per-cluster sums / counts for given labels, with coalesced atomics: a wave takes V = max(1, 64/d)
// vectors per trip, lane (v, e) adds byte e of vector v to sums[label_v * d + e], so that one
// wave-instruction adds to runs of d consecutive 64-bit words.  Serves vq_accum_labels and the
// codebooks whose [K][d+1] table does not fit the LDS.
__global__ __launch_bounds__(256) void k_accum_rows(const uint8_t *__restrict__ blocks,
                                                    const uint16_t *__restrict__ perm, int64_t N, int d,
                                                    const int32_t *__restrict__ labels, int K,
                                                    unsigned long long *__restrict__ sums,
                                                    unsigned long long *__restrict__ counts, Gate gate) {
    if (gate_closed(gate)) return;
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int V = d >= WAVE ? 1 : WAVE / d;
    const int v = d >= WAVE ? 0 : lane / d;
    const int e0 = d >= WAVE ? lane : lane - v * d;
    const bool lane_on = d >= WAVE || v < V;
    for (int64_t n0 = wave * V; n0 < N; n0 += nwaves * V) {
        const int64_t n = n0 + v;
        if (!lane_on || n >= N) continue;
        int64_t m = n;                                       // index of row n's vector in the caller's order
        if (perm) {
            m = (n & ~(int64_t)(SORT_WIN - 1)) + (int64_t)perm[n];
            m = m < N ? m : N - 1;
        }
        const int k = labels[m];
        if ((unsigned)k >= (unsigned)K) continue;
        const uint8_t *x = blocks + (size_t)n * d;
        for (int e = e0; e < d; e += WAVE) {
            unsigned b = x[e];
            if (b) atomicAdd(&sums[(size_t)k * d + e], (unsigned long long)b);
        }
        if (e0 == 0) atomicAdd(&counts[k], 1ull);
    }
}

// Per-cluster sums / counts for given labels through a per-workgroup LDS table [K][d+1] of 32-bit
// words (ds_add_u32), flushed with 64-bit global atomics: integer, so the totals do not depend on
// the order.  The vectors are taken in the caller's order, where neighbouring lanes mostly hold
// different labels (few same-address LDS conflicts).  With min_dist non-null the kernel also
// evaluates sum_e |x_e - cb[label][e]| in sequential fp64 (the value VQ_Full.py:89 gathers) for
// every vector; CB_LDS stages the fp64 codebook in LDS behind the table.  VEC16: 16-byte rows.
template <bool VEC16, bool CB_LDS>
__global__ __launch_bounds__(1024) void k_accum_dist(const uint8_t *__restrict__ blocks,
                                                    const uint16_t *__restrict__ perm, int64_t N, int d,
                                                    const int32_t *__restrict__ labels,
                                                    const double *__restrict__ cb, int K,
                                                    unsigned long long *__restrict__ sums,
                                                    unsigned long long *__restrict__ counts,
                                                    double *__restrict__ min_dist, int reps, int64_t sorted_end, Gate gate) {
    // sorted_end (with perm): labels and min_dist of the rows below it are indexed by the row's own (sorted) position,
    // not through the origin map -- what the search of a training pass leaves
    extern __shared__ __align__(16) uint32_t s_acc[];
    if (gate_closed(gate)) return;
    const int tid = threadIdx.x;
    // `reps` copies of the table (a power of two; more than one for small K): lane l adds to copy l mod reps, so that
    // the lanes of a wave that hold the same label -- most of them when there are only a few labels -- do not all
    // queue up on one LDS address
    // (copies an odd number of words apart: lanes with one label then add into different banks; codebook rows padded
    // to d + 1 doubles: lanes read the same element of different rows, and the odd stride spreads those over the banks)
    const int tabn = K * (d + 1);
    const int tabs = tabn | 1;
    const int taball = reps * tabs;
    const int cbs = d + 1;
    const double *s_cb = reinterpret_cast<const double *>(s_acc + ((taball + 3) & ~3));
    for (int i = tid; i < taball; i += blockDim.x) s_acc[i] = 0u;
    if (CB_LDS) {
        double *dst = reinterpret_cast<double *>(s_acc + ((taball + 3) & ~3));
        for (int i = tid; i < K * d; i += blockDim.x) {
            const int kk = i / d;
            dst[kk * cbs + (i - kk * d)] = cb[i];
        }
    }
    __syncthreads();
    uint32_t *my_tab = s_acc + (size_t)(tid & (reps - 1)) * tabs;
    auto flush = [&]() {
        for (int i = tid; i < tabn; i += blockDim.x) {
            uint32_t v = 0;
            for (int r = 0; r < reps; ++r) {
                v += s_acc[r * tabs + i];
                s_acc[r * tabs + i] = 0u;
            }
            if (v) {
                const int kk = i / (d + 1), e = i - kk * (d + 1);
                if (e < d) atomicAdd(&sums[(size_t)kk * d + e], (unsigned long long)v);
                else atomicAdd(&counts[kk], (unsigned long long)v);
            }
        }
    };
    // UV vectors per thread and trip: their labels and rows are requested together, so that one
    // memory round trip serves UV vectors
    constexpr int UV = 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t trips = 0;
    for (int64_t n0 = (int64_t)blockIdx.x * blockDim.x; n0 < N; n0 += UV * stride) {
        int64_t n[UV], m[UV];                                    // row read; vector index in the caller's order
        int k[UV];
        uint4 v[UV];
#pragma unroll
        for (int u = 0; u < UV; ++u) {
            n[u] = n0 + u * stride + tid;
            const int64_t nc = n[u] < N ? n[u] : N - 1;           // clamped: loads stay in bounds
            m[u] = nc;
            if (perm && nc >= sorted_end) {                       // `blocks` is the sorted copy
                m[u] = (nc & ~(int64_t)(SORT_WIN - 1)) + (int64_t)perm[nc];
                m[u] = m[u] < N ? m[u] : N - 1;
            }
            if (VEC16) v[u] = reinterpret_cast<const uint4 *>(blocks)[nc];
        }
#pragma unroll
        for (int u = 0; u < UV; ++u) k[u] = labels[m[u]];
#pragma unroll
        for (int u = 0; u < UV; ++u) {
            if (n[u] >= N || (unsigned)k[u] >= (unsigned)K) continue;
            uint32_t *t = my_tab + (size_t)k[u] * (d + 1);
            const double *crow = CB_LDS ? s_cb + (size_t)k[u] * cbs : cb + (size_t)k[u] * d;
            double acc = 0.0;
            if (VEC16) {
                const uint32_t w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t b = (w4[e >> 2] >> (8 * (e & 3))) & 0xffu;
                    atomicAdd(&t[e], b);
                    if (min_dist) acc += fabs((double)b - crow[e]);
                }
            } else {
                const uint8_t *x = blocks + (size_t)n[u] * d;
                for (int e = 0; e < d; ++e) {
                    const uint32_t b = x[e];
                    atomicAdd(&t[e], b);
                    if (min_dist) acc += fabs((double)b - crow[e]);
                }
            }
            atomicAdd(&t[d], 1u);
            if (min_dist) min_dist[m[u]] = acc;
        }
        // 32-bit LDS sums grow by at most 255 * UV * blockDim per trip: flush long before they can wrap
        if ((++trips & 0x3fffu) == 0) {
            __syncthreads();
            flush();
            __syncthreads();
        }
    }
    __syncthreads();
    flush();
}

// ------------------------------------------------------------------------------------------------
// decode (VQ_Full.py:157-169)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cb_to_u8(const double *__restrict__ cb, int64_t n, uint8_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = cast_u8(cb[i]);
}

// The truncated codebook of the decode kernels in LDS: converted from the fp64 codebook (cb non-null: small codebooks),
// or copied from the uint8 table k_cb_to_u8 left in global memory (cb null: a workgroup then reads K * d bytes
// instead of K * d doubles -- with many workgroups per launch the conversions were most of the time of a decode of
// 1 M RGB 4x4 blocks).  cb8g is 16-byte aligned.
__device__ __forceinline__ void stage_cb8(uint8_t *s_cb8, const double *cb, const uint8_t *cb8g, uint32_t n) {
    if (cb) {
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) s_cb8[i] = cast_u8(cb[i]);
    } else {
        const uint32_t n16 = n / 16;
        for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x)
            reinterpret_cast<uint4 *>(s_cb8)[i] = reinterpret_cast<const uint4 *>(cb8g)[i];
        for (uint32_t i = n16 * 16 + threadIdx.x; i < n; i += blockDim.x) s_cb8[i] = cb8g[i];
    }
}

// Byte-wise decode for block row segments that are not a whole number of aligned words: one thread per
// output byte of an image row.  blockIdx.y strides over image rows, so all index arithmetic is 32-bit.
// With LDS_CB each workgroup truncates the fp64 codebook into LDS once (K*d bytes) and gathers from
// there; otherwise it gathers from the uint8 copy that k_cb_to_u8 left in global memory.
template <bool LDS_CB>
__global__ __launch_bounds__(256) void k_decode_u8(const int32_t *__restrict__ labels, int nr, int nc,
                                                   const double *__restrict__ cb, const uint8_t *__restrict__ cb8g,
                                                   int K, int bh, int bw, int C, uint8_t *__restrict__ img) {
    extern __shared__ __align__(16) uint8_t s_cb8[];
    const uint32_t seg = (uint32_t)(bw * C);
    const uint32_t d = (uint32_t)bh * seg;
    const uint32_t rowbytes = (uint32_t)nc * seg;
    const uint32_t rows = (uint32_t)nr * (uint32_t)bh;
    if (LDS_CB) {
        const uint32_t n = (uint32_t)K * d;
        stage_cb8(s_cb8, cb, cb8g, n);
        __syncthreads();
    }
    const uint8_t *cb8 = LDS_CB ? s_cb8 : cb8g;
    for (uint32_t y = blockIdx.y; y < rows; y += gridDim.y) {
        const uint32_t r = y / (uint32_t)bh, i = y - r * (uint32_t)bh;
        const int32_t *lrow = labels + (size_t)r * nc;
        uint8_t *orow = img + (size_t)y * rowbytes;
        for (uint32_t xb = blockIdx.x * blockDim.x + threadIdx.x; xb < rowbytes; xb += gridDim.x * blockDim.x) {
            const uint32_t q = xb / seg, rem = xb - q * seg;
            orow[xb] = cb8[(size_t)min((uint32_t)lrow[q], (uint32_t)K - 1u) * d + i * seg + rem];
        }
    }
}

// Decode for any block shape (segments that are not whole words, image rows at any alignment): a workgroup takes a
// strip of up to TB consecutive blocks of one block row.  Phase 1: a thread per block copies its codeword's bytes
// (truncated codebook in LDS when it fits) into bh LDS rows at the block's place -- each row shifted by the byte
// alignment of its destination in the image, so that aligned words of the image are aligned words of LDS.  Phase 2
// writes every row out: leading bytes up to the first aligned address, whole words, trailing bytes.
// rs: bytes per LDS row, a multiple of four, at least TB * seg + 4.
template <bool LDS_CB>
__global__ __launch_bounds__(256) void k_decode_strip(const int32_t *__restrict__ labels, int nr, int nc,
                                                      const double *__restrict__ cb, const uint8_t *__restrict__ cb8g, int K,
                                                      int bh, int bw, int C, uint8_t *__restrict__ img, int TB, int rs) {
    extern __shared__ __align__(16) uint8_t s_cb8[];             // [truncated codebook, rounded up to 16 bytes |] bh rows
    const int seg = bw * C, d = bh * seg;
    const uint32_t n = (uint32_t)K * (uint32_t)d;
    uint8_t *s_rows = s_cb8 + (LDS_CB ? ((n + 15u) & ~15u) : 0u);
    if (LDS_CB) {
        stage_cb8(s_cb8, cb, cb8g, n);
        __syncthreads();
    }
    const uint8_t *cb8 = LDS_CB ? s_cb8 : cb8g;
    const size_t rowbytes = (size_t)nc * seg;
    const int tiles_per_row = (nc + TB - 1) / TB;
    const int64_t ntiles = (int64_t)nr * tiles_per_row;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {    // (workgroup-uniform)
        const int r = (int)(t / tiles_per_row);
        const int q0 = (int)(t - (int64_t)r * tiles_per_row) * TB;
        const int nb = nc - q0 < TB ? nc - q0 : TB;
        uint8_t *dst0 = img + (size_t)r * bh * rowbytes + (size_t)q0 * seg;     // the strip in the block row's first pixel row
        for (int qq = threadIdx.x; qq < nb; qq += 256) {
            const uint32_t k = min((uint32_t)labels[(size_t)r * nc + q0 + qq], (uint32_t)K - 1u);    // stays in bounds
            const uint8_t *crow = cb8 + (size_t)k * d;
            int e = 0;
            for (int i = 0; i < bh; ++i) {
                const uint32_t a = (uint32_t)(reinterpret_cast<uintptr_t>(dst0 + (size_t)i * rowbytes) & 3u);
                uint8_t *row = s_rows + (size_t)i * rs + a + (size_t)qq * seg;
                for (int rem = 0; rem < seg; ++rem) row[rem] = crow[e++];
            }
        }
        __syncthreads();
        const int nbytes = nb * seg;
        for (int i = 0; i < bh; ++i) {
            uint8_t *dst = dst0 + (size_t)i * rowbytes;
            const uint32_t a = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u);
            const uint8_t *srow = s_rows + (size_t)i * rs + a;    // byte k of the strip's row
            int head = (int)((4u - a) & 3u);
            if (head > nbytes) head = nbytes;
            const int nwords = (nbytes - head) >> 2;
            const int tail0 = head + 4 * nwords;
            if ((int)threadIdx.x < head) dst[threadIdx.x] = srow[threadIdx.x];
            const uint32_t *sw = reinterpret_cast<const uint32_t *>(srow + head);        // (4-byte aligned: a + head = 0 mod 4)
            uint32_t *dw = reinterpret_cast<uint32_t *>(dst + head);
            for (int w = threadIdx.x; w < nwords; w += 256) dw[w] = sw[w];
            if ((int)threadIdx.x < nbytes - tail0) dst[tail0 + threadIdx.x] = srow[tail0 + threadIdx.x];
        }
        __syncthreads();
    }
}

// Decode for block row segments that are whole aligned words.  One thread per G consecutive output bytes
// of an image row (G = 16: four words gathered from the truncated codebook, one 16-byte store; G = 4: one
// word); blockIdx.y strides over BLOCK rows: a thread fetches the labels its G bytes need once and writes
// all bh pixel rows of that block row, so every label is read once.
template <int G, bool LDS_CB>
__global__ __launch_bounds__(256) void k_decode_blockrows_u8(const int32_t *__restrict__ labels, int nr, int nc,
                                                             const double *__restrict__ cb,
                                                             const uint8_t *__restrict__ cb8g, int K, int bh, int bw,
                                                             int C, uint8_t *__restrict__ img) {
    extern __shared__ __align__(16) uint8_t s_cb8[];
    static_assert(G == 16 || G == 4, "word-wide groups only");
    constexpr int NW = G / 4;
    const uint32_t seg = (uint32_t)(bw * C);
    const uint32_t d = (uint32_t)bh * seg;
    const uint32_t rowbytes = (uint32_t)nc * seg;
    const uint32_t groups_per_row = rowbytes / G;
    if (LDS_CB) {
        const uint32_t n = (uint32_t)K * d;
        stage_cb8(s_cb8, cb, cb8g, n);
        __syncthreads();
    }
    const uint8_t *cb8 = LDS_CB ? s_cb8 : cb8g;
    for (uint32_t r = blockIdx.y; r < (uint32_t)nr; r += gridDim.y) {
        const int32_t *lrow = labels + (size_t)r * nc;
        for (uint32_t xg = blockIdx.x * blockDim.x + threadIdx.x; xg < groups_per_row; xg += gridDim.x * blockDim.x) {
            const uint32_t xb0 = xg * G;
            uint32_t base[NW];                                  // byte offset of word w's codeword row + column
            uint32_t q = xb0 / seg, rem = xb0 - q * seg;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                base[w] = min((uint32_t)lrow[q], (uint32_t)K - 1u) * d + rem;      // a label outside [0, K) stays in bounds
                rem += 4;
                if (rem >= seg) {
                    rem = 0;
                    ++q;
                }
            }
            for (uint32_t i = 0; i < (uint32_t)bh; ++i) {
                uint32_t words[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) words[w] = *reinterpret_cast<const uint32_t *>(cb8 + base[w] + i * seg);
                uint8_t *o = img + (size_t)(r * (uint32_t)bh + i) * rowbytes + xb0;
                if (G == 16) {
                    *reinterpret_cast<uint4 *>(o) = make_uint4(words[0], words[1 % NW], words[2 % NW], words[3 % NW]);
                } else {
                    *reinterpret_cast<uint32_t *>(o) = words[0];
                }
            }
        }
    }
}

// Decode of 16-byte codewords that are 4 pixel rows of 4 bytes (4x4 blocks of one-byte pixels, 2x2 of two-byte
// ones ...): a thread takes four horizontally adjacent blocks -- one 16-byte load of their labels, one 16-byte
// read of each codeword from the truncated codebook (LDS when it fits) -- and writes the four pixel rows with
// one 16-byte store each, so a wave's stores are 1 KB runs of an image row.  Two items per thread and trip,
// label loads first.
template <bool LDS_CB>
__global__ __launch_bounds__(1024) void k_decode_quads(const int32_t *__restrict__ labels, int nr, int nc,
                                                       const double *__restrict__ cb, const uint8_t *__restrict__ cb8g,
                                                       int K, uint8_t *__restrict__ img) {
    extern __shared__ __align__(16) uint8_t s_cb8[];
    const uint32_t qpr = (uint32_t)nc / 4;                       // quads per block row
    const size_t rowbytes = (size_t)nc * 4;
    const int64_t total = (int64_t)nr * qpr;
    const uint4 *lab4 = reinterpret_cast<const uint4 *>(labels);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 la = t < total ? lab4[t] : make_uint4(0, 0, 0, 0);
    uint4 lb = t + stride < total ? lab4[t + stride] : make_uint4(0, 0, 0, 0);
    if (LDS_CB) {                                                // truncate the fp64 codebook into LDS, a word per trip
        for (int i = threadIdx.x; i < K * 4; i += blockDim.x) {
            const double *c = cb + (size_t)i * 4;
            reinterpret_cast<uint32_t *>(s_cb8)[i] = (uint32_t)cast_u8(c[0]) | (uint32_t)cast_u8(c[1]) << 8 |
                                                     (uint32_t)cast_u8(c[2]) << 16 | (uint32_t)cast_u8(c[3]) << 24;
        }
        __syncthreads();
    }
    const uint4 *cw = reinterpret_cast<const uint4 *>(LDS_CB ? s_cb8 : cb8g);
    auto emit = [&](int64_t item, const uint4 &l) {
        const uint32_t r = (uint32_t)(item / qpr), qx = (uint32_t)(item - (int64_t)r * qpr);
        // (labels are the caller's: one outside [0, K) must not lead out of the codebook)
        const uint32_t km = (uint32_t)K - 1u;
        const uint4 c0 = cw[min(l.x, km)], c1 = cw[min(l.y, km)], c2 = cw[min(l.z, km)], c3 = cw[min(l.w, km)];
        uint8_t *o = img + (size_t)r * 4 * rowbytes + (size_t)qx * 16;
        *reinterpret_cast<uint4 *>(o) = make_uint4(c0.x, c1.x, c2.x, c3.x);
        *reinterpret_cast<uint4 *>(o + rowbytes) = make_uint4(c0.y, c1.y, c2.y, c3.y);
        *reinterpret_cast<uint4 *>(o + 2 * rowbytes) = make_uint4(c0.z, c1.z, c2.z, c3.z);
        *reinterpret_cast<uint4 *>(o + 3 * rowbytes) = make_uint4(c0.w, c1.w, c2.w, c3.w);
    };
    while (t < total) {
        const int64_t t2 = t + 2 * stride;
        const uint4 na = t2 < total ? lab4[t2] : make_uint4(0, 0, 0, 0);
        const uint4 nb = t2 + stride < total ? lab4[t2 + stride] : make_uint4(0, 0, 0, 0);
        emit(t, la);
        if (t + stride < total) emit(t + stride, lb);
        la = na;
        lb = nb;
        t = t2;
    }
}

// ------------------------------------------------------------------------------------------------
// label bit-packing (VQ_Full.py:134-147) and its inverse
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_labels(const int32_t *__restrict__ labels, int64_t N, int bits,
                                                     uint8_t *__restrict__ out, int64_t nbytes) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nbytes; j += (int64_t)gridDim.x * blockDim.x) {
        uint32_t byte = 0;
        int64_t g = j * 8;
        int64_t n = g / bits;
        int pos = (int)(g - n * bits);                 // bits of label n already consumed
        uint32_t lab = n < N ? (uint32_t)labels[n] : 0u;
        for (int b = 0; b < 8; ++b) {
            uint32_t bit = (lab >> (bits - 1 - pos)) & 1u;
            byte |= bit << (7 - b);
            if (++pos == bits) {
                pos = 0;
                ++n;
                lab = n < N ? (uint32_t)labels[n] : 0u;
            }
        }
        out[j] = (uint8_t)byte;
    }
}

__global__ __launch_bounds__(256) void k_unpack_labels(const uint8_t *__restrict__ packed, int64_t N, int bits,
                                                       int32_t *__restrict__ labels) {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        uint32_t v = 0;
        int64_t g = n * bits;
        for (int b = 0; b < bits; ++b, ++g) {
            uint32_t bit = (packed[g >> 3] >> (7 - (int)(g & 7))) & 1u;
            v = (v << 1) | bit;
        }
        labels[n] = (int32_t)v;
    }
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch helpers
// ------------------------------------------------------------------------------------------------
constexpr int DP2_LIST[] = {2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96};

int dp2_bucket(int d) {
    int need = (d + 1) / 2;
    for (int v : DP2_LIST)
        if (v >= need) return v;
    return -1;
}

constexpr int r_for(int dp2) { return dp2 <= 8 ? VQ_R8 : (dp2 <= 16 ? 4 : VQ_RBIG); }

int ceil_log2(int64_t v) {
    int b = 0;
    while ((1LL << b) < v) ++b;
    return b;
}

int grid_for(int64_t work_items, int block, int max_blocks) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

// Vectors per lane: the scalar-load variant serves d <= 32, the LDS variant everything.
constexpr int r_lds_for(int dp2) { return dp2 <= 32 ? 4 : 2; }
// Pruned search by scalar loads: rows beyond 32 bytes take one vector per lane -- 60 instead of 91 vector registers at
// 48 bytes, tiles of half the length (a launch of a few rounds of workgroups ends with less of the chip idle) and a
// narrower range of element sums per wave (1 M vectors, K = 256: 48-byte rows 119 -> 111 us, 64-byte rows 139 -> 121 us).
#ifndef VQ_RP_LONG
#define VQ_RP_LONG 1
#endif
constexpr int r_prune_for(int dp2) { return dp2 <= 16 ? VQ_RP : VQ_RP_LONG; }

// Events to record around the next search-kernel launch of the calling thread (vq_profile_next_search; one-shot).
thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;

template <int DP2, int KEYMODE, bool LDS_CW, bool PRUNE = false, bool ACC = false>
int launch_assign_fast_t(const AssignArgs &a, const uint32_t *c16, const uint32_t *bias, int block, int grid,
                         size_t lds, hipStream_t st) {
    constexpr int R = LDS_CW ? r_lds_for(DP2) : (PRUNE ? r_prune_for(DP2) : r_for(DP2));
    auto kern = k_assign_fast<DP2, R, KEYMODE, LDS_CW, PRUNE, ACC>;
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    const hipEvent_t ev0 = g_prof_start, ev1 = g_prof_stop;
    g_prof_start = g_prof_stop = nullptr;
#ifdef VQ_DEBUG_LDS
    if (PRUNE && !LDS_CW && !ACC && getenv("VQ_DEBUG_LDS")) {
        lds = (size_t)atoi(getenv("VQ_DEBUG_LDS"));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
#endif
    if (ev0) HIP_TRY(hipEventRecord(ev0, st));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, st, a, c16, bias, a.cb);
    LAUNCH_CHECK("k_assign_fast");
    if (ev1) HIP_TRY(hipEventRecord(ev1, st));
    return VQ_OK;
}

template <int DP2>
int launch_assign_fast_k(const AssignArgs &a, const uint32_t *c16, const uint32_t *bias, bool packed, bool lds_cw,
                         bool prune, int block, int grid, size_t lds, hipStream_t st) {
    if constexpr (DP2 <= 16) {
        if (!lds_cw && prune && a.acc_rep) {
            if (packed) return launch_assign_fast_t<DP2, KEY_PACKED, false, true, true>(a, c16, bias, block, grid, lds, st);
            return launch_assign_fast_t<DP2, KEY_SPLIT, false, true, true>(a, c16, bias, block, grid, lds, st);
        }
        if (!lds_cw && !prune) {
            if (packed) return launch_assign_fast_t<DP2, KEY_PACKED, false>(a, c16, bias, block, grid, lds, st);
            return launch_assign_fast_t<DP2, KEY_SPLIT, false>(a, c16, bias, block, grid, lds, st);
        }
    }
    if constexpr (DP2 <= 32) {
        // scalar-load pruned search: rows up to 64 bytes (two vectors per lane beyond 32 bytes)
        if (!lds_cw && prune) {
            if (packed) return launch_assign_fast_t<DP2, KEY_PACKED, false, true>(a, c16, bias, block, grid, lds, st);
            return launch_assign_fast_t<DP2, KEY_SPLIT, false, true>(a, c16, bias, block, grid, lds, st);
        }
        if (packed) {
            if (prune) return launch_assign_fast_t<DP2, KEY_PACKED, true, true>(a, c16, bias, block, grid, lds, st);
            return launch_assign_fast_t<DP2, KEY_PACKED, true>(a, c16, bias, block, grid, lds, st);
        }
    }
    if (prune) return launch_assign_fast_t<DP2, KEY_SPLIT, true, true>(a, c16, bias, block, grid, lds, st);
    return launch_assign_fast_t<DP2, KEY_SPLIT, true>(a, c16, bias, block, grid, lds, st);
}

// The scalar-load variant is used while the fixed-point codebook (K * dp2 words) stays within
// about twice the scalar cache and a row fits the scalar registers comfortably.
bool use_lds_codewords(int K, int dp2) { return dp2 > 16 || (size_t)K * dp2 * 4 > 32 * 1024; }

struct DeviceInfo {
    int cus;
    size_t lds_per_block;
};

// Properties of the current device.  They never change while the process lives, so each device is
// queried once; the table is written under a mutex and only ever read afterwards.
constexpr int MAX_DEVICES = 64;
std::mutex g_di_mutex;
std::atomic<int> g_di_ready[MAX_DEVICES];
DeviceInfo g_di[MAX_DEVICES];

int device_info(DeviceInfo &di) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const bool cached = dev >= 0 && dev < MAX_DEVICES;
    if (cached && g_di_ready[dev].load(std::memory_order_acquire)) {
        di = g_di[dev];
        return VQ_OK;
    }
    int cus = 0, lds = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    HIP_TRY(hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    di.cus = cus;
    di.lds_per_block = (size_t)lds;
    if (cached) {
        std::lock_guard<std::mutex> lock(g_di_mutex);
        g_di[dev] = di;
        g_di_ready[dev].store(1, std::memory_order_release);
    }
    return VQ_OK;
}

// Device address of `scratch` on the current device, looked up once per device (with that device
// current) and kept for the life of the process.
std::mutex g_scratch_mutex;
std::atomic<int> g_scratch_ready[MAX_DEVICES];
void *g_scratch_ptr[MAX_DEVICES];

int static_scratch(void **out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return fail(VQ_E_UNSUPPORTED, "device %d is beyond the %d this library tracks", dev, MAX_DEVICES);
    if (!g_scratch_ready[dev].load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ready[dev].load(std::memory_order_relaxed)) {
            void *ptr = nullptr;
            HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
            if (!ptr) return fail(VQ_E_HIP, "hipGetSymbolAddress returned a null address for the static scratch");
            g_scratch_ptr[dev] = ptr;
            g_scratch_ready[dev].store(1, std::memory_order_release);
        }
    }
    *out = g_scratch_ptr[dev];
    return VQ_OK;
}

// Key layout of the filter.  Packed keys hold (S << kbits) + index in 31 bits, so the SAD sum S of
// 2*dp2 terms, each at most clim, must stay below 2^(31 - kbits); clim is the fixed-point limit the
// prepare kernel scales codebook and pixels into.  Packed is used when that still leaves at least
// 4 fractional bits for 8-bit data; otherwise value and index are tracked separately (clim 65535).
struct KeyLayout {
    bool packed;
    uint32_t clim;
};

KeyLayout key_layout(int K, int dp2) {
    const int kbits = ceil_log2(K);
    KeyLayout kl{false, 65535u};
    if (dp2 <= 32 && kbits < 31) {
        uint64_t lim = ((1ull << (31 - kbits)) - 1) / (2ull * dp2);
        if (lim > 65535) lim = 65535;
        if (lim >= 255ull * 16) {
            kl.packed = true;
            kl.clim = (uint32_t)lim;
        }
    }
    return kl;
}

// blocks_optional: the call has a window-sorted copy that serves this (K, d), so the block matrix itself is
// not needed
int check_common(const void *blocks, int64_t N, int d, const void *cb, int K, bool blocks_optional = false) {
    if ((!blocks && N != 0 && !blocks_optional) || !cb) return fail(VQ_E_INVALID, "null blocks or codebook pointer");
    if (N < 0 || d <= 0 || K <= 0) return fail(VQ_E_INVALID, "need N >= 0, d > 0, K > 0 (got N=%lld d=%d K=%d)", (long long)N, d, K);
    if (K > MAX_K) return fail(VQ_E_UNSUPPORTED, "K=%d exceeds %d", K, MAX_K);
    return VQ_OK;
}

// What the last vq_lbg_iterate of the calling thread left in a workspace: replica tables already zeroed for the
// next iteration.  Every entry point that takes a workspace comes through check_ws, which forgets it.
struct TailLeft {
    const void *ws;
    const void *state;
    int iteration, K, d;
    int64_t N;
    bool prepared;                   // the workspace is already prepared for the codebook that call left behind
};
thread_local TailLeft g_tail_left = {nullptr, nullptr, 0, 0, 0, 0, false};

int check_ws(const void *ws, size_t ws_bytes, int K, int d, int64_t N) {
    g_tail_left = TailLeft{nullptr, nullptr, 0, 0, 0, 0, false};
    size_t need = ws_layout(K, d, N).total;
    if (!ws || ws_bytes < need) return fail(VQ_E_WORKSPACE, "workspace of %zu bytes needed, %zu given", need, ws_bytes);
    return VQ_OK;
}

bool fast_supported(int d) { return d <= MAX_D_FAST && dp2_bucket(d) > 0; }

// Fills the workspace for a codebook: 16-bit fixed-point rows, key biases, transposed fp64 copy.
int run_prepare(const double *cb, int K, int d, void *ws, hipStream_t st) {
    const int dp2 = dp2_bucket(d);
    WsLayout L = ws_layout(K, d, 0);
    char *w = static_cast<char *>(ws);
    // 2K keys + K hashes + 2K indices; afterwards the first K * 4 bytes hold the sorted element sums
    const size_t hash_lds = (K <= PREP_HASH_CAP && K > 1) ? (size_t)K * 32 : 16;
    static_assert(PREP_SORT_CAP <= PREP_HASH_CAP, "sorted codebooks have a hash table to put their sums in");
    PrepHeader *hdr = reinterpret_cast<PrepHeader *>(w + L.header);
    uint32_t *c16 = reinterpret_cast<uint32_t *>(w + L.c16), *bias = reinterpret_cast<uint32_t *>(w + L.bias);
    double *cbT = reinterpret_cast<double *>(w + L.cbT);
    uint32_t *postab = reinterpret_cast<uint32_t *>(w + L.postab);
    const uint32_t clim = key_layout(K, dp2).clim;
    if (d <= 16 && K <= 512) {                              // short rows, small codebooks: one workgroup, one launch
        if (K <= 256)
            hipLaunchKernelGGL(k_prepare_small<4>, dim3(1), dim3(1024), hash_lds, st, cb, K, d, dp2, clim, hdr, c16, bias, cbT, postab);
        else
            hipLaunchKernelGGL(k_prepare_small<2>, dim3(1), dim3(1024), hash_lds, st, cb, K, d, dp2, clim, hdr, c16, bias, cbT, postab);
        LAUNCH_CHECK("k_prepare_small");
        return VQ_OK;
    }
    DeviceInfo di;
    int rc = device_info(di);
    if (rc) return rc;
    const bool do_sort = K <= PREP_SORT_CAP;
    uint32_t *c16u = reinterpret_cast<uint32_t *>(w + L.c16u);
    unsigned long long *hash = reinterpret_cast<unsigned long long *>(w + L.prep_tmp);
    uint32_t *sc = reinterpret_cast<uint32_t *>(hash + K), *scs = sc + K;
    HIP_TRY(hipMemsetAsync(hdr, 0, sizeof(PrepHeader), st));
    const int64_t n = (int64_t)K * d;
    hipLaunchKernelGGL(k_prep_max, dim3(grid_for(n, 256 * 8, di.cus * 4)), dim3(256), 0, st, cb, n, hdr);
    LAUNCH_CHECK("k_prep_max");
    hipLaunchKernelGGL(k_prep_rows, dim3(grid_for(K, 4, di.cus * 8)), dim3(256), 0, st, cb, K, d, dp2, clim, hdr,
                       do_sort ? c16u : c16, sc, hash, do_sort ? nullptr : cbT, do_sort ? nullptr : bias);
    LAUNCH_CHECK("k_prep_rows");
    if (do_sort) {
        const size_t lds = (size_t)K * 8 + (size_t)((K + 3) & ~3) * 4;
        if (lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_prep_rank), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
        hipLaunchKernelGGL(k_prep_rank, dim3((K + PREP_RANK_ROWS - 1) / PREP_RANK_ROWS), dim3(256), lds, st, cb, K, d, dp2, sc,
                           hash, c16u, c16, bias, cbT, scs,
                           wants_c16T(K, d) ? reinterpret_cast<uint32_t *>(w + L.c16T) : nullptr);
        LAUNCH_CHECK("k_prep_rank");
    }
    hipLaunchKernelGGL(k_prep_finish, dim3(1), dim3(1024), 0, st, K, d, dp2, clim, do_sort ? 1 : 0, hdr, scs, c16, bias, postab);
    LAUNCH_CHECK("k_prep_finish");
    return VQ_OK;
}

// Layout of the buffer vq_blocks_sort fills: the sorted rows, the window-local origins, then a header
// that says what the buffer holds.
struct SortedLayout {
    size_t rows;       // N * d bytes
    size_t perm;       // N uint16
    size_t hdr;        // SortedHeader
    size_t total;
};

SortedLayout sorted_layout(int d, int64_t N) {
    SortedLayout L;
    L.rows = 0;
    L.perm = align_up((size_t)N * d, 256);
    L.hdr = L.perm + align_up((size_t)N * 2, 256);
    L.total = L.hdr + 256;
    return L;
}

// Host-side record of the buffers vq_blocks_sort has filled (keyed by device address): the search
// refuses a `sorted` argument that was never filled, or was filled for another N or d, instead of
// reading past its end.  Guarded by a mutex; an entry is replaced when the same address is sorted again.
struct SortedInfo {
    int64_t N;
    int d;
    size_t bytes;
};
std::mutex g_sorted_mutex;
std::unordered_map<uintptr_t, SortedInfo> g_sorted_known;

constexpr size_t SORTED_KNOWN_CAP = 4096;                    // records kept; beyond that the table starts over

void sorted_register(const void *sorted, int64_t N, int d, size_t bytes) {
    std::lock_guard<std::mutex> lock(g_sorted_mutex);
    // a process that sorts ever new buffers and never calls vq_sorted_forget must not grow the table without bound:
    // dropping records only makes a later search ask for a fresh sort (VQ_E_INVALID), never trust a stale one
    if (g_sorted_known.size() >= SORTED_KNOWN_CAP) g_sorted_known.clear();
    g_sorted_known[reinterpret_cast<uintptr_t>(sorted)] = SortedInfo{N, d, bytes};
}

int sorted_check(const void *sorted, size_t sorted_bytes, int64_t N, int d) {
    const size_t need = sorted_layout(d, N).total;
    if (sorted_bytes < need)
        return fail(VQ_E_INVALID, "sorted buffer of %zu bytes given, N=%lld d=%d needs %zu", sorted_bytes, (long long)N, d, need);
    std::lock_guard<std::mutex> lock(g_sorted_mutex);
    auto it = g_sorted_known.find(reinterpret_cast<uintptr_t>(sorted));
    if (it == g_sorted_known.end())
        return fail(VQ_E_INVALID, "sorted buffer %p was not filled by vq_blocks_sort", sorted);
    if (it->second.N != N || it->second.d != d)
        return fail(VQ_E_INVALID, "sorted buffer %p holds N=%lld d=%d, the call has N=%lld d=%d", sorted,
                    (long long)it->second.N, it->second.d, (long long)N, d);
    return VQ_OK;
}

// Whether a pruned search takes its codewords through scalar loads: the shapes of the scalar-load kernels,
// and rows of 34 .. 64 bytes (RGB 4x4 blocks are 48; two vectors per lane) while the fixed-point codebook
// fits the scalar cache budget.  Everything else streams codeword tiles through LDS.
bool prune_by_scalar_loads(int K, int dp2) {
    if (!use_lds_codewords(K, dp2)) return true;
    return VQ_PRUNE_WIDE != 0 && dp2 <= 32 && (size_t)K * dp2 * 4 <= 32 * 1024;
}

// The pruned search needs the codebook sorted by element sum (K <= PREP_SORT_CAP) and pays from the
// codebook size on at which skipping codewords can cover the bookkeeping.
bool prune_supported(int K, int d) {
    const int dp2 = dp2_bucket(d);
    return dp2 > 0 && K >= 32 && K <= PREP_SORT_CAP;
}

// FAST mode on a prepared workspace: the filter kernel, then the fp64 fix-up of the vectors it
// queued.  Writes labels and, when min_dist is non-null, the winning fp64 distances.
// LDS bytes of the fused accumulation (table + per-wave staging) for a 256-thread workgroup.
size_t fused_acc_lds(int K, int d, int waves = 4, int reps = 1) {
    const int dp2 = dp2_bucket(d);
    return ((size_t)reps * (K + 1) * (2 * dp2 + 2) + 3) / 4 * 16 + (size_t)waves * (WAVE * 2 * dp2 + WAVE * 4);
}
// The fused pass with the fp64 codebook in LDS as well: workgroups of VQ_ACC_WAVES waves, two of them per compute unit.
#ifndef VQ_ACC_WAVES
#define VQ_ACC_WAVES 12
#endif
size_t fused_acc_cb_lds(int K, int d, int reps = 1) {
    return fused_acc_lds(K, d, VQ_ACC_WAVES, reps) + (size_t)K * (d + 1) * 8;
}

// Whether run_fast can accumulate sums / counts itself (pruned kernel, table within a quarter of the LDS).
bool fused_acc_supported(const void *sorted, int K, int d) {
    return sorted != nullptr && prune_supported(K, d) && dp2_bucket(d) <= 16 && !use_lds_codewords(K, dp2_bucket(d)) &&
           fused_acc_lds(K, d) <= 40 * 1024;
}

// How the search kernels read a vector (AssignArgs::vec4): 0 = byte by byte; 1 = word by word, the row ending anywhere
// in the kernel's padded length; 2 = rows of exactly the padded length in 16-byte pieces (rows and base address
// multiples of 16).
int row_load_mode(const void *rows, int d, int dp2) {
    const uintptr_t addr = reinterpret_cast<uintptr_t>(rows);
    if ((d & 3) != 0 || (addr & 3) != 0) return 0;
    return (d == 2 * dp2 && (d & 15) == 0 && (addr & 15) == 0) ? 2 : 1;
}

// sums / counts non-null (only with fused_acc_supported): cluster sums are accumulated too.
int run_fast(const uint8_t *blocks, const void *sorted, size_t sorted_bytes, int64_t N, int d, const double *cb, int K,
             int32_t *labels, double *min_dist, unsigned long long *sums, unsigned long long *counts, void *ws,
             hipStream_t st, int phases = 3, bool dist_sorted = false) {
    const int dp2 = dp2_bucket(d);
    if (sorted != nullptr && prune_supported(K, d)) {          // before any work is enqueued
        int src = sorted_check(sorted, sorted_bytes, N, d);
        if (src) return src;
    }
    WsLayout L = ws_layout(K, d, 0);
    char *w = static_cast<char *>(ws);
    PrepHeader *hdr = reinterpret_cast<PrepHeader *>(w + L.header);
    uint32_t *c16 = reinterpret_cast<uint32_t *>(w + L.c16);
    uint32_t *bias = reinterpret_cast<uint32_t *>(w + L.bias);
    double *cbT = reinterpret_cast<double *>(w + L.cbT);
    WsLayout LN = ws_layout(K, d, N);
    uint32_t *qcount = reinterpret_cast<uint32_t *>(w + LN.qcount);
    uint32_t *queue = reinterpret_cast<uint32_t *>(w + LN.queue);
    const bool do_filter = (phases & 1) != 0, do_fixup = (phases & 2) != 0;
    if (do_filter && !g_ws_zeroed) HIP_TRY(hipMemsetAsync(qcount, 0, (size_t)NQ * 64, st));
    const uint32_t qcap = (uint32_t)queue_cap(N);

    DeviceInfo di;
    int rc = device_info(di);
    if (rc) return rc;

    const int kbits = ceil_log2(K);
    const bool packed = key_layout(K, dp2).packed;
    const bool prune = sorted != nullptr && prune_supported(K, d);
    const bool lds_cw = prune ? !prune_by_scalar_loads(K, dp2) : use_lds_codewords(K, dp2);
    const int R = lds_cw ? r_lds_for(dp2) : (prune ? r_prune_for(dp2) : r_for(dp2));
    // The LDS-tiled pruned search with no more workgroups than the chip holds at once takes workgroups of two waves:
    // four small workgroups per compute unit even out better than two large ones.
    int block = 256;
    const bool acc_cb_lds = sums != nullptr && min_dist != nullptr && fused_acc_cb_lds(K, d) <= 80 * 1024;
    if (acc_cb_lds) block = 64 * VQ_ACC_WAVES;
#ifdef VQ_SMALL_WG
    if (prune && lds_cw && (N + 256 * R - 1) / (256 * R) <= (int64_t)di.cus * VQ_SMALL_WG) block = 128;
#endif
    if (sums && !fused_acc_supported(sorted, K, d)) return fail(VQ_E_HIP, "internal: fused accumulation is not available here");
    // LDS-staged codewords: two tiles of rows and biases, plus the words of the workgroup reductions
    // copies of the accumulation table: up to the number of vectors one atomic instruction adds, within the LDS budget
    int tab_reps = 1;
    if (sums) {
        const int vs = WAVE / dp2;
        while (2 * tab_reps <= vs && 2 * tab_reps <= 8 &&
               (acc_cb_lds ? fused_acc_cb_lds(K, d, 2 * tab_reps) <= 80 * 1024 : fused_acc_lds(K, d, 4, 2 * tab_reps) <= 40 * 1024))
            tab_reps *= 2;
    }
    const size_t lds = lds_cw ? (size_t)2 * tile_k_for(dp2) * (dp2 + 1) * 4 + 128
                              : (sums ? (acc_cb_lds ? fused_acc_cb_lds(K, d, tab_reps) : fused_acc_lds(K, d, 4, tab_reps)) : 0);
    const int64_t tile = (int64_t)block * R;
    const int64_t ntiles = (N + tile - 1) / tile;
    // One tile per workgroup: the dispatcher balances the chip and the last, partly filled round of
    // workgroups runs at full clock (measured faster than any persistent grid).
    if (ntiles > ((int64_t)1 << 30)) return fail(VQ_E_UNSUPPORTED, "N=%lld needs more than 2^30 workgroups", (long long)N);
    int grid = (int)ntiles;
    if (prune && grid >= 8) grid = (grid + 7) & ~7;               // a multiple of the XCD count (see the kernel's tile order)
    if (grid < 1) grid = 1;

    AssignArgs a;
    a.blocks = blocks; a.N = N; a.d = d; a.cb = cb; a.cbT = cbT; a.K = K; a.kbits = kbits; a.hdr = hdr;
    a.labels = labels; a.min_dist = min_dist;
    a.queue = queue; a.qcount = qcount; a.qcap = qcap;
    a.acc_rep = nullptr;
    if (sums) {
        a.acc_rep = reinterpret_cast<unsigned long long *>(w + L.acc_rep);
        if (!g_ws_zeroed) HIP_TRY(hipMemsetAsync(a.acc_rep, 0, (size_t)NREP * K * (d + 1) * 8, st));
    }
    a.ntiles = ntiles;
    a.vec4 = row_load_mode(blocks, d, dp2);
    a.rows = nullptr; a.perm = nullptr; a.shdr = nullptr;
    a.dist_sorted_end = (prune && dist_sorted) ? N / NP_CHUNK * NP_CHUNK : 0;
    a.cb_lds = acc_cb_lds ? 1 : 0;
    a.tab_reps = tab_reps;
    // Every round ends with a reduction over the wave and two table look-ups: larger steps mean fewer rounds and a few
    // rows searched for nothing.  Measured best: 32 rows first and per side for codebooks up to 512 rows, 64 per side
    // beyond (K = 1024, d = 16: 345 -> 323 us for 4.19 M vectors).
    a.first_rows = VQ_G0;
    a.round_rows = K > 512 ? 2 * VQ_GP : VQ_GP;
    a.persist_tiles = 0;
#ifndef VQ_NO_PERSIST
    if (acc_cb_lds && ntiles >= (int64_t)4 * di.cus) {         // (several rounds of workgroups: two resident ones per compute unit)
        const int64_t wave_tiles = (N + (int64_t)WAVE * R - 1) / ((int64_t)WAVE * R);
        // an XCD's share in whole sort windows, so that the tiles of a window stay on one XCD
        const int64_t per_win = SORT_WIN / (WAVE * R);
        a.persist_tiles = (int)(((wave_tiles + 7) / 8 + per_win - 1) / per_win * per_win);
        grid = 2 * di.cus;
    }
#endif
    a.heavy_first = (prune && grid <= 16 * di.cus) ? (grid <= 4 * di.cus ? 2 : 1) : 0;
    a.gate = g_gate;
    a.postab = reinterpret_cast<const uint32_t *>(w + L.postab);
    if (prune) {
        SortedLayout SL = sorted_layout(d, N);
        const char *sp = static_cast<const char *>(sorted);
        a.rows = reinterpret_cast<const uint8_t *>(sp + SL.rows);
        a.perm = reinterpret_cast<const uint16_t *>(sp + SL.perm);
        a.shdr = reinterpret_cast<const SortedHeader *>(sp + SL.hdr);
        a.vec4 = row_load_mode(a.rows, d, dp2);
        // a wave's vectors must come from one sort window; the LDS-tiled search needs that of a whole workgroup
        if (SORT_WIN % (lds_cw ? tile : (int64_t)WAVE * R) != 0)
            return fail(VQ_E_HIP, "internal: sort window %d is not a multiple of the tile %lld", SORT_WIN, (long long)tile);
    }

    if (do_filter)
    switch (dp2) {
#ifndef VQ_FEW_SHAPES                                       // (experimental builds compile three row lengths only)
        case 2:  rc = launch_assign_fast_k<2>(a, c16, bias, packed, lds_cw, prune, block, grid, lds, st); break;
        case 4:  rc = launch_assign_fast_k<4>(a, c16, bias, packed, lds_cw, prune, block, grid, lds, st); break;
        case 6:  rc = launch_assign_fast_k<6>(a, c16, bias, packed, lds_cw, prune, block, grid, lds, st); break;
        case 12: rc = launch_assign_fast_k<12>(a, c16, bias, packed, lds_cw, prune, block, grid, lds, st); break;
        case 16: rc = launch_assign_fast_k<16>(a, c16, bias, packed, lds_cw, prune, block, grid, lds, st); break;
        case 32: rc = launch_assign_fast_k<32>(a, c16, bias, packed, lds_cw, prune, block, grid, lds, st); break;
        case 48: rc = launch_assign_fast_k<48>(a, c16, bias, packed, lds_cw, prune, block, grid, lds, st); break;
        case 64: rc = launch_assign_fast_k<64>(a, c16, bias, packed, lds_cw, prune, block, grid, lds, st); break;
#endif
        case 8:  rc = launch_assign_fast_k<8>(a, c16, bias, packed, lds_cw, prune, block, grid, lds, st); break;
        case 24: rc = launch_assign_fast_k<24>(a, c16, bias, packed, lds_cw, prune, block, grid, lds, st); break;
        case 96: rc = launch_assign_fast_k<96>(a, c16, bias, packed, lds_cw, prune, block, grid, lds, st); break;
        default: return fail(VQ_E_UNSUPPORTED, "no fast kernel for d=%d", d);
    }
    if (rc) return rc;
    if (!do_fixup) return VQ_OK;
    if (resolves_in_kernel(dp2, lds_cw, prune)) {              // the filter left nothing queued
        if (sums) {
            hipLaunchKernelGGL(k_reduce_replicas, dim3((K * (d + 1) + 255) / 256), dim3(256), 0, st, a.acc_rep, K, d, sums,
                               counts, g_gate);
            LAUNCH_CHECK("k_reduce_replicas");
        }
        return VQ_OK;
    }
    // the fix-up reads the rows the queue entries point at: the sorted copy after a pruned search
    const uint8_t *fx_rows = prune ? a.rows : blocks;
    const uint16_t *fx_perm = prune ? a.perm : nullptr;
    // fix-up: a fixed grid strides over the sub-queues (their lengths are only known on the device);
    // small codebooks are staged in LDS, one 1024-thread workgroup per CU
    const size_t cb_bytes = (size_t)K * d * sizeof(double);
    // Codebooks that the prepare kernels sorted by element sum (their transposed fp64 copy is in sorted order too) take
    // the fix-up with the element-sum bound; larger ones are scanned whole.
    if (K <= PREP_SORT_CAP) {
        const uint32_t *postab = reinterpret_cast<const uint32_t *>(w + L.postab);
        const size_t fx_lds = cb_bytes + (size_t)(POSTAB_WORDS + K) * 4;     // codebook, position table, position map
        if (cb_bytes <= 128 * 1024) {
            if (fx_lds > 48 * 1024)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fixup_pruned),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)fx_lds));
            hipLaunchKernelGGL(k_fixup_pruned, dim3((di.cus + 3) / 4 * 4), dim3(1024), fx_lds, st, fx_rows, fx_perm, N, d,
                               cbT, K, queue, qcount, qcap, hdr, bias, postab, labels, min_dist, sums, counts, g_gate);
        } else {
            hipLaunchKernelGGL(k_fixup_shortlist, dim3((di.cus * 8 + NQ - 1) / NQ * NQ), dim3(256), 0, st, fx_rows, fx_perm, N, d,
                               dp2, cb, reinterpret_cast<const uint32_t *>(w + L.c16T), K, queue, qcount, qcap, hdr, bias, postab,
                               labels, min_dist, sums, counts, g_gate);
        }
        LAUNCH_CHECK("k_fixup_pruned");
    } else if (cb_bytes <= 64 * 1024) {
        if (cb_bytes > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fixup<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)cb_bytes));
        hipLaunchKernelGGL(k_fixup<true>, dim3((di.cus + 3) / 4 * 4), dim3(1024), cb_bytes, st, fx_rows, fx_perm, N, d, cbT,
                           K, queue, qcount, qcap, labels, min_dist, sums, counts, g_gate);
    } else {
        hipLaunchKernelGGL(k_fixup<false>, dim3((di.cus * 4 + 15) / 16 * 16), dim3(256), 0, st, fx_rows, fx_perm, N, d, cbT, K,
                           queue, qcount, qcap, labels, min_dist, sums, counts, g_gate);
    }
    LAUNCH_CHECK("k_fixup");
    if (sums) {
        hipLaunchKernelGGL(k_reduce_replicas, dim3((K * (d + 1) + 255) / 256), dim3(256), 0, st, a.acc_rep, K, d, sums,
                           counts, g_gate);
        LAUNCH_CHECK("k_reduce_replicas");
    }
    return VQ_OK;
}

// Whether the [K][d+1] table of k_accum_dist fits the LDS of one workgroup.
bool accum_table_fits(int K, int d, const DeviceInfo &di) { return (size_t)K * (d + 1) * 4 + 32 <= di.lds_per_block; }

// Per-cluster sums / counts for a finished labelling, plus (want_dist) the fp64 distance of every
// vector to its codeword: the LDS-table kernel when the table fits, else coalesced global atomics.
// When the table does not fit, distances are NOT computed here (the caller asks the search for them).
int run_accum(const uint8_t *blocks, const uint16_t *perm, int64_t N, int d, const int32_t *labels, const double *cb,
              int K, unsigned long long *sums, unsigned long long *counts, double *min_dist, hipStream_t st,
              int64_t sorted_end = 0) {
    DeviceInfo di;
    int rc = device_info(di);
    if (rc) return rc;
    if (accum_table_fits(K, d, di)) {
        // small codebooks: several copies of the table (see the kernel), as many as keep it within 32 KB, up to 32
        // (16.7 M vectors of 16 bytes, K = 16: 407 us with one copy, 220 with 4, 105 -- the time of the memory traffic --
        // with 16 copies and padded codebook rows)
        int reps = 1;
        while (reps < 32 && reps * K < 256 && (size_t)(2 * reps) * K * (d + 1) * 4 <= 32 * 1024) reps *= 2;
        const size_t tab = ((size_t)reps * ((size_t)K * (d + 1) | 1) + 3) / 4 * 16;
        const size_t cbb = (size_t)K * (d + 1) * 8;
        const bool cb_lds = min_dist != nullptr && tab + cbb <= 64 * 1024;
        const size_t lds = tab + (cb_lds ? cbb : 0);
        int wg_per_cu = (int)((160 * 1024) / (lds > 0 ? lds : 1));
        if (wg_per_cu > 4) wg_per_cu = 4;
        if (wg_per_cu < 1) wg_per_cu = 1;
        // large tables allow two workgroups per compute unit: with 1024 threads each they still fill its wave slots (the
        // kernel waits on chains of dependent loads: label, then codebook row)
#ifndef VQ_ACC_BLOCK_BIG
#define VQ_ACC_BLOCK_BIG 1024
#endif
        const int ACC_BLOCK = wg_per_cu <= 2 ? VQ_ACC_BLOCK_BIG : 512;
        const int grid = grid_for(N, ACC_BLOCK, di.cus * wg_per_cu);
        const bool vec16 = d == 16 && (reinterpret_cast<uintptr_t>(blocks) & 15) == 0;
#define VQ_LAUNCH_ACCUM(V, C)                                                                                          \
    do {                                                                                                               \
        auto kern = k_accum_dist<V, C>;                                                                                \
        if (lds > 48 * 1024)                                                                                           \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(ACC_BLOCK), lds, st, blocks, perm, N, d, labels, cb, K, sums, counts, min_dist, reps, sorted_end, g_gate); \
    } while (0)
        if (vec16 && cb_lds) VQ_LAUNCH_ACCUM(true, true);
        else if (vec16) VQ_LAUNCH_ACCUM(true, false);
        else if (cb_lds) VQ_LAUNCH_ACCUM(false, true);
        else VQ_LAUNCH_ACCUM(false, false);
#undef VQ_LAUNCH_ACCUM
        LAUNCH_CHECK("k_accum_dist");
        return VQ_OK;
    }
    if (sorted_end != 0) return fail(VQ_E_HIP, "internal: sorted-order labels need the LDS-table accumulation");
    const int V = d >= WAVE ? 1 : WAVE / d;
    hipLaunchKernelGGL(k_accum_rows, dim3(grid_for((N + V - 1) / V, 4, 16384)), dim3(256), 0, st, blocks, perm, N, d, labels,
                       K, sums, counts, g_gate);
    LAUNCH_CHECK("k_accum_rows");
    return VQ_OK;
}

// numpy-order sum of v[0..N) into out[0]; chunk_sums holds ceil(N / 8192) (+1) doubles of scratch.
int run_numpy_sum(const double *v, int64_t N, double *chunk_sums, double *out, hipStream_t st) {
    const int64_t nfull = N / NP_CHUNK;
    if (nfull > 0) {
        DeviceInfo di;
        int rc = device_info(di);
        if (rc) return rc;
        int grid = grid_for(nfull, 1, di.cus * 8);
        hipLaunchKernelGGL(k_npsum_chunks, dim3(grid), dim3(512), 0, st, v, nfull, chunk_sums);
        LAUNCH_CHECK("k_npsum_chunks");
    }
    hipLaunchKernelGGL(k_npsum_final, dim3(1), dim3(1024), 0, st, v, N, nfull, chunk_sums, out);
    LAUNCH_CHECK("k_npsum_final");
    return VQ_OK;
}

// The same for distances whose whole chunks lie in the order of the window-sorted vectors: every chunk is put back into
// the caller's order in LDS before it is summed (perm: the origin map of the sorted copy).
__global__ __launch_bounds__(512) void k_npsum_chunks_sorted(const double *__restrict__ v, const uint16_t *__restrict__ perm,
                                                             int64_t nfull, double *__restrict__ chunk_sums, Gate g) {
    extern __shared__ __align__(16) double s_dist[];
    __shared__ double s_leaf[64];
    if (gate_closed(g)) return;
    for (int64_t c = blockIdx.x; c < nfull; c += gridDim.x) npsum_chunk_sorted(v, perm, c, chunk_sums, s_leaf, s_dist);
}
int run_numpy_sum_sorted(const double *v, const uint16_t *perm, int64_t N, double *chunk_sums, double *out, hipStream_t st) {
    const int64_t nfull = N / NP_CHUNK;
    if (nfull > 0) {
        DeviceInfo di;
        int rc = device_info(di);
        if (rc) return rc;
        const size_t lds = (size_t)NP_CHUNK * sizeof(double);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_npsum_chunks_sorted),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_npsum_chunks_sorted, dim3(grid_for(nfull, 1, di.cus * 2)), dim3(512), lds, st, v, perm, nfull,
                           chunk_sums, g_gate);
        LAUNCH_CHECK("k_npsum_chunks_sorted");
    }
    hipLaunchKernelGGL(k_npsum_final, dim3(1), dim3(1024), 0, st, v, N, nfull, chunk_sums, out);
    LAUNCH_CHECK("k_npsum_final");
    return VQ_OK;
}

// k_window_sort_regs reads an image by direct addressing only: 32-bit byte offsets, rows and block segments of
// whole aligned words, and at least one whole block row and column (its idle lanes read the first block).
bool sort_in_registers_ok(const SrcBlocks &) { return true; }
bool sort_in_registers_ok(const SrcImage &s) {
    return s.fast32 != 0 && s.words_ok != 0 && (s.seg & 3u) == 0 && s.H >= s.bh && s.W >= s.bw;
}

// Fills `sorted` (rows, origins, header) from a vector source.  row_words_ok: rows are whole words the
// source can deliver aligned; rows of one to four such words are staged in LDS and leave in order.
template <class SRC>
int run_sort(const SRC &src, bool row_words_ok, int64_t N, int d, void *sorted, size_t sorted_bytes, hipStream_t st,
             bool in_place = false) {
    const bool words = row_words_ok && d <= 16;
    if (d > MAX_D_FAST) return fail(VQ_E_UNSUPPORTED, "d=%d is beyond the fast kernels", d);
    if (N > 0xfffffff0LL) return fail(VQ_E_UNSUPPORTED, "N=%lld exceeds 2^32 per call", (long long)N);
    if (!sorted) return fail(VQ_E_INVALID, "null pointer");
    SortedLayout SL = sorted_layout(d, N);
    if (sorted_bytes < SL.total) return fail(VQ_E_WORKSPACE, "sorted buffer of %zu bytes needed, %zu given", SL.total, sorted_bytes);
    if (reinterpret_cast<uintptr_t>(sorted) & 15) return fail(VQ_E_INVALID, "sorted buffer must be 16-byte aligned");
    char *sp = static_cast<char *>(sorted);
    uint8_t *rows = reinterpret_cast<uint8_t *>(sp + SL.rows);
    uint16_t *perm = reinterpret_cast<uint16_t *>(sp + SL.perm);
    SortedHeader *shdr = reinterpret_cast<SortedHeader *>(sp + SL.hdr);
    int bshift = 0;                                        // element sums 0 .. 255 * d into SORT_BUCKETS buckets
    while (((255u * (uint32_t)d) >> bshift) >= (uint32_t)SORT_BUCKETS) ++bshift;
    const int grid = (int)((N + SORT_WIN - 1) / SORT_WIN);
    DeviceInfo di;
    int rc = device_info(di);
    if (rc) return rc;
    const size_t lds_need = (size_t)SORT_WIN * d + (size_t)SORT_WIN * 2 + (SORT_BUCKETS + 16) * 4 + 256;
    if (words && lds_need <= di.lds_per_block) {
        const size_t lds = (size_t)SORT_WIN * d + (size_t)SORT_WIN * 2;
        uint32_t *r32 = reinterpret_cast<uint32_t *>(rows);
#define VQ_LAUNCH_SORT(DW)                                                                                          \
    do {                                                                                                            \
        auto kern = k_window_sort_lds<DW, SRC>;                                                                     \
        if (lds > 48 * 1024)                                                                                        \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                       \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                     \
        hipLaunchKernelGGL(kern, dim3(grid < di.cus ? grid : di.cus), dim3(1024), lds, st, src, N, bshift, r32, perm, shdr); \
    } while (0)
        switch (d / 4) {
            case 1: VQ_LAUNCH_SORT(1); break;
            case 2: VQ_LAUNCH_SORT(2); break;
            case 3: VQ_LAUNCH_SORT(3); break;
            default: VQ_LAUNCH_SORT(4); break;
        }
#undef VQ_LAUNCH_SORT
    } else if (row_words_ok && (d == 48 || d == 32) && sort_in_registers_ok(src) &&
               (size_t)SORT_CHUNK * d + (size_t)SORT_WIN * 4 + (SORT_BUCKETS + 64) * 4 + 256 <= di.lds_per_block) {
        const size_t lds = (size_t)SORT_CHUNK * d;
        uint32_t *r32 = reinterpret_cast<uint32_t *>(rows);
#define VQ_LAUNCH_SORT_REGS(DW)                                                                                     \
    do {                                                                                                            \
        auto kern = k_window_sort_regs<DW, SRC>;                                                                    \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, src, N, bshift, r32, perm, shdr);                  \
    } while (0)
        if (d == 48) VQ_LAUNCH_SORT_REGS(12);
        else VQ_LAUNCH_SORT_REGS(8);
#undef VQ_LAUNCH_SORT_REGS
    } else {
        // (the two kernels above have read a window completely before they write it; this one has not)
        if (in_place) return fail(VQ_E_HIP, "internal: the generic window sort cannot run in place");
        // few windows: the rows are copied by a second launch with several workgroups per window
        int parts = 1;
        if (grid * 4 <= di.cus) parts = di.cus / grid < 16 ? di.cus / grid : 16;   // (two parts per window: measured no gain)
        static_assert(255 * MAX_D_FAST < 65536, "element sums travel in 16-bit entries");
        if (parts > 1) {
            if (row_words_ok)
                hipLaunchKernelGGL((k_sort_sums<SRC, true>), dim3(grid * parts), dim3(1024), 0, st, src, N, d, parts, perm);
            else
                hipLaunchKernelGGL((k_sort_sums<SRC, false>), dim3(grid * parts), dim3(1024), 0, st, src, N, d, parts, perm);
            LAUNCH_CHECK("k_sort_sums");
        }
        if (row_words_ok)
            hipLaunchKernelGGL((k_window_sort<SRC, true>), dim3(grid), dim3(1024), 0, st, src, N, d, bshift, rows, perm, shdr,
                               parts == 1 ? 1 : 0, parts == 1 ? 0 : 1);
        else
            hipLaunchKernelGGL((k_window_sort<SRC, false>), dim3(grid), dim3(1024), 0, st, src, N, d, bshift, rows, perm, shdr,
                               parts == 1 ? 1 : 0, parts == 1 ? 0 : 1);
        if (parts > 1) {
            LAUNCH_CHECK("k_window_sort");
            if (row_words_ok)
                hipLaunchKernelGGL((k_sort_gather<SRC, true>), dim3(grid * parts), dim3(1024), 0, st, src, N, d, parts, rows, perm);
            else
                hipLaunchKernelGGL((k_sort_gather<SRC, false>), dim3(grid * parts), dim3(1024), 0, st, src, N, d, parts, rows, perm);
        }
    }
    LAUNCH_CHECK("k_window_sort");
    sorted_register(sorted, N, d, sorted_bytes);
    return VQ_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int vq_abi_version(void) { return ABI_VERSION; }

#ifdef VQ_TIMELINE
// Diagnostic builds only: copies the wave timeline of the last k_assign_fast launch to the host (synchronises).
int vq_debug_timeline(uint32_t *h_out, size_t words) {
    if (words > (size_t)(8 << 18)) words = (size_t)(8 << 18);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_timeline), words * 4, 0, hipMemcpyDeviceToHost));
    return VQ_OK;
}
#endif

const char *vq_last_error(void) { return g_err; }

size_t vq_workspace_bytes(int K, int d, int64_t N) {
    if (K <= 0 || d <= 0 || N < 0) return 0;
    return ws_layout(K, d, N).total;
}

int vq_blocks_u8(const uint8_t *img, int H, int W, int C, int bh, int bw, int pad_mode, uint8_t *blocks,
                 void *stream) {
    if (!img || !blocks) return fail(VQ_E_INVALID, "null image or blocks pointer");
    if (H <= 0 || W <= 0 || C <= 0 || bh <= 0 || bw <= 0) return fail(VQ_E_INVALID, "non-positive image or block size");
    if (pad_mode != VQ_PAD_EDGE && pad_mode != VQ_PAD_ZERO) return fail(VQ_E_INVALID, "unknown pad_mode %d", pad_mode);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int nr = (H + bh - 1) / bh, nc = (W + bw - 1) / bw;
    int64_t d = (int64_t)bh * bw * C;
    int64_t total = (int64_t)nr * nc * d;
    const uintptr_t optr = reinterpret_cast<uintptr_t>(blocks);
    const int words_ok = 1;                                 // (unaligned word loads: see load_u32_any)
    const bool seg4 = (bw * C) % 4 == 0;
    const int pz = pad_mode == VQ_PAD_ZERO;
    if (C == 1 && bh == 4 && bw == 4 && H % 4 == 0 && W % 16 == 0 && (optr & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(img) & 15) == 0 && (int64_t)nr * nc < (1ll << 31)) {
        const int64_t quads = (int64_t)nr * nc / 4;
        hipLaunchKernelGGL(k_blocks_quads, dim3(grid_for(quads, 256, 16384)), dim3(256), 0, st, img, W, blocks, nc, quads);
    } else if (seg4 && d % 16 == 0 && d <= 192 && nc >= 64 && (optr & 15) == 0) {     // (narrow images: too few blocks per strip)
        const int64_t tiles = (int64_t)nr * ((nc + 255) / 256);
        const size_t lds = (size_t)256 * (size_t)d;
        if (lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_blocks_rows), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
        hipLaunchKernelGGL(k_blocks_rows, dim3(grid_for(tiles, 1, 1 << 20)), dim3(256), lds, st, img, H, W, C, bh, bw, pz, blocks,
                           nr, nc);
    } else if (nc >= 64 && d <= 768) {
        // strips of TB blocks: TB * d bytes of LDS (at most 48 KB), TB a multiple of 256 up to 4096
        int TB = (int)((48 * 1024) / d) / 256 * 256;
        if (TB > 4096) TB = 4096;
        if (TB < 64) TB = 64;
        const int rs = (TB * (bw * C) + 3) / 4 * 4;
        const int64_t tiles = (int64_t)nr * ((nc + TB - 1) / TB);
        const int word_stores = (d % 4 == 0 && (optr & 3) == 0) ? 1 : 0;
        hipLaunchKernelGGL(k_blocks_strip, dim3(grid_for(tiles, 1, 1 << 20)), dim3(256), (size_t)bh * rs, st, img, H, W, C, bh, bw,
                           pz, blocks, nr, nc, TB, rs, word_stores);
    } else if (seg4 && total % 16 == 0 && (optr & 15) == 0) {
        int64_t groups = total / 16;
        hipLaunchKernelGGL(k_blocks_u8<16>, dim3(grid_for(groups, 256, 16384)), dim3(256), 0, st, img, H, W, C, bh, bw, pz,
                           blocks, nc, groups, words_ok);
    } else if (seg4 && (optr & 3) == 0) {
        int64_t groups = total / 4;
        hipLaunchKernelGGL(k_blocks_u8<4>, dim3(grid_for(groups, 256, 16384)), dim3(256), 0, st, img, H, W, C, bh, bw, pz,
                           blocks, nc, groups, words_ok);
    } else {
        hipLaunchKernelGGL(k_blocks_u8<1>, dim3(grid_for(total, 256, 16384)), dim3(256), 0, st, img, H, W, C, bh, bw, pz,
                           blocks, nc, total, 0);
    }
    LAUNCH_CHECK("k_blocks_u8");
    return VQ_OK;
}

int vq_colsum_u8(const uint8_t *blocks, int64_t N, int d, uint64_t *sums, void *stream) {
    if ((!blocks && N != 0) || !sums) return fail(VQ_E_INVALID, "null pointer");
    if (N < 0 || d <= 0) return fail(VQ_E_INVALID, "need N >= 0 and d > 0");
    if (d > 8192) return fail(VQ_E_UNSUPPORTED, "d=%d exceeds 8192", d);
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(sums, 0, sizeof(uint64_t) * (size_t)d, st));
    if (N == 0) return VQ_OK;
    int64_t threads = (int64_t)2048 * 256;
    int64_t rows_per_pass = threads / d;
    if (rows_per_pass < 1) rows_per_pass = 1;
    if (rows_per_pass > N) rows_per_pass = N;
    int grid = grid_for(rows_per_pass * d, 256, 1 << 20);
    hipLaunchKernelGGL(k_colsum_u8, dim3(grid), dim3(256), sizeof(unsigned long long) * (size_t)d, st, blocks, N, d,
                       reinterpret_cast<unsigned long long *>(sums), rows_per_pass);
    LAUNCH_CHECK("k_colsum_u8");
    return VQ_OK;
}

int vq_encode_l1(const uint8_t *blocks, int64_t N, int d, const double *cb, int K, int32_t *labels, double *min_dist,
                 int mode, void *ws, size_t ws_bytes, void *stream) {
    int rc = check_common(blocks, N, d, cb, K);
    if (rc) return rc;
    if (mode != VQ_MODE_EXACT && mode != VQ_MODE_FAST) return fail(VQ_E_INVALID, "unknown mode %d", mode);
    if (N == 0) return VQ_OK;
    if (!labels) return fail(VQ_E_INVALID, "null labels pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (mode == VQ_MODE_FAST && fast_supported(d)) {
        if (N > 0xfffffff0LL) return fail(VQ_E_UNSUPPORTED, "N=%lld exceeds 2^32 per call", (long long)N);
        rc = check_ws(ws, ws_bytes, K, d, N);
        if (rc) return rc;
        rc = run_prepare(cb, K, d, ws, st);
        if (rc) return rc;
        return run_fast(blocks, nullptr, 0, N, d, cb, K, labels, min_dist, nullptr, nullptr, ws, st);
    }
    int grid = grid_for(N, 256, 16384);
    hipLaunchKernelGGL(k_assign_exact, dim3(grid), dim3(256), 0, st, blocks, N, d, cb, K, labels, min_dist, g_gate);
    LAUNCH_CHECK("k_assign_exact");
    return VQ_OK;
}

size_t vq_static_scratch_bytes(void) {
    void *p = nullptr;
    (void)static_scratch(&p);                                  // look the current device's array up now (not under a capture later)
    return (size_t)VQ_STATIC_SCRATCH_BYTES;
}

int vq_encode_l1_static(const uint8_t *blocks, int64_t N, int d, const double *cb, int K, int32_t *labels,
                        double *min_dist, int mode, void *stream) {
    int rc = check_common(blocks, N, d, cb, K);
    if (rc) return rc;
    if (mode != VQ_MODE_EXACT && mode != VQ_MODE_FAST) return fail(VQ_E_INVALID, "unknown mode %d", mode);
    if (N == 0) return VQ_OK;
    if (!labels) return fail(VQ_E_INVALID, "null labels pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (mode == VQ_MODE_FAST && fast_supported(d)) {
        if (N > 0xfffffff0LL) return fail(VQ_E_UNSUPPORTED, "N=%lld exceeds 2^32 per call", (long long)N);
        const size_t need = ws_layout(K, d, N).total;
        if (need > (size_t)VQ_STATIC_SCRATCH_BYTES)
            return fail(VQ_E_UNSUPPORTED,
                        "K=%d d=%d N=%lld needs %zu bytes of scratch, the static array holds %zu: use vq_encode_l1 "
                        "with a caller-owned workspace, or split the vectors into smaller calls",
                        K, d, (long long)N, need, (size_t)VQ_STATIC_SCRATCH_BYTES);
        void *ws = nullptr;
        rc = static_scratch(&ws);
        if (rc) return rc;
        rc = run_prepare(cb, K, d, ws, st);
        if (rc) return rc;
        return run_fast(blocks, nullptr, 0, N, d, cb, K, labels, min_dist, nullptr, nullptr, ws, st);
    }
    // the all-fp64 search keeps everything in registers: no scratch at all
    int grid = grid_for(N, 256, 16384);
    hipLaunchKernelGGL(k_assign_exact, dim3(grid), dim3(256), 0, st, blocks, N, d, cb, K, labels, min_dist, g_gate);
    LAUNCH_CHECK("k_assign_exact");
    return VQ_OK;
}

int vq_codebook_prepare(const double *cb, int K, int d, void *ws, size_t ws_bytes, void *stream) {
    if (!cb) return fail(VQ_E_INVALID, "null codebook pointer");
    if (K <= 0 || d <= 0) return fail(VQ_E_INVALID, "need K > 0 and d > 0");
    if (K > MAX_K) return fail(VQ_E_UNSUPPORTED, "K=%d exceeds %d", K, MAX_K);
    if (!fast_supported(d)) return fail(VQ_E_UNSUPPORTED, "d=%d is beyond the prepared (fast) kernels", d);
    int rc = check_ws(ws, ws_bytes, K, d, 0);
    if (rc) return rc;
    return run_prepare(cb, K, d, ws, static_cast<hipStream_t>(stream));
}

size_t vq_sorted_bytes(int d, int64_t N) {
    if (d <= 0 || N <= 0) return 0;
    return sorted_layout(d, N).total;
}

int vq_prune_supported(int K, int d) { return (K > 0 && d > 0 && prune_supported(K, d)) ? 1 : 0; }

int vq_prune_resolves_in_filter(int K, int d) {
    if (K <= 0 || d <= 0 || !prune_supported(K, d)) return 0;
    const int dp2 = dp2_bucket(d);
    return resolves_in_kernel(dp2, !prune_by_scalar_loads(K, dp2), true) ? 1 : 0;
}

int vq_blocks_sort(const uint8_t *blocks, int64_t N, int d, void *sorted, size_t sorted_bytes, void *stream) {
    if (N < 0 || d <= 0) return fail(VQ_E_INVALID, "need N >= 0 and d > 0");
    if (N == 0) return VQ_OK;
    if (!blocks) return fail(VQ_E_INVALID, "null pointer");
    const uintptr_t bp = reinterpret_cast<uintptr_t>(blocks);
    const bool words = (d % 4 == 0) && (bp & 3) == 0 && (d != 16 || (bp & 15) == 0) && (d != 8 || (bp & 7) == 0);
    return run_sort(SrcBlocks{blocks, d}, words, N, d, sorted, sorted_bytes, static_cast<hipStream_t>(stream));
}

int vq_blocks_sort_img(const uint8_t *img, int H, int W, int C, int bh, int bw, int pad_mode, void *sorted,
                       size_t sorted_bytes, void *stream) {
    if (!img) return fail(VQ_E_INVALID, "null image pointer");
    if (H <= 0 || W <= 0 || C <= 0 || bh <= 0 || bw <= 0) return fail(VQ_E_INVALID, "non-positive image or block size");
    if (pad_mode != VQ_PAD_EDGE && pad_mode != VQ_PAD_ZERO) return fail(VQ_E_INVALID, "unknown pad_mode %d", pad_mode);
    const int64_t nr = (H + bh - 1) / bh, nc = (W + bw - 1) / bw;
    const int64_t d64 = (int64_t)bh * bw * C;
    if (d64 > MAX_D_FAST) return fail(VQ_E_UNSUPPORTED, "d=%lld is beyond the fast kernels", (long long)d64);
    const int d = (int)d64;
    SrcImage src;
    src.img = img; src.H = H; src.W = W; src.C = C; src.bh = bh; src.bw = bw; src.pad_zero = pad_mode == VQ_PAD_ZERO;
    src.nc = (uint32_t)nc; src.seg = (uint32_t)(bw * C); src.d = (uint32_t)d;
    src.words_ok = 1;                                       // (unaligned word loads: see load_u32_any)
    src.quads_ok = C == 1 && bh == 4 && bw == 4 && H % 4 == 0 && W % 16 == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0;
    src.fast32 = (unsigned long long)H * (unsigned long long)W * (unsigned long long)C < (1ull << 32) ? 1 : 0;
    // Images whose rows are not whole aligned words (W * C % 4 != 0: most RGB photographs) are read with unaligned
    // loads, which the fused kernels pay for dearly (4090 x 4090 RGB, 4x4 blocks: 128 us against 34 us for 4096 x 4096).
    // Where the block-matrix sort can run in place -- rows of up to 16, 32 or 48 bytes: a window is read completely
    // before it is written -- such an image is extracted into the rows region of `sorted` (a streaming pass that
    // takes the unaligned loads in its stride) and sorted there: 68 us for the same image.
    const bool unaligned = ((int64_t)W * C) % 4 != 0 || (reinterpret_cast<uintptr_t>(img) & 3) != 0;
    const int64_t N = nr * nc;
    if (unaligned && d % 4 == 0 && (d <= 16 || d == 32 || d == 48) && N > 0) {
        const SortedLayout SL = sorted_layout(d, N);
        if (!sorted) return fail(VQ_E_INVALID, "null pointer");
        if (sorted_bytes < SL.total) return fail(VQ_E_WORKSPACE, "sorted buffer of %zu bytes needed, %zu given", SL.total, sorted_bytes);
        if (reinterpret_cast<uintptr_t>(sorted) & 15) return fail(VQ_E_INVALID, "sorted buffer must be 16-byte aligned");
        uint8_t *rows = reinterpret_cast<uint8_t *>(static_cast<char *>(sorted) + SL.rows);
        int rc = vq_blocks_u8(img, H, W, C, bh, bw, pad_mode, rows, stream);
        if (rc) return rc;
        return run_sort(SrcBlocks{rows, d}, true, N, d, sorted, sorted_bytes, static_cast<hipStream_t>(stream), true);
    }
    return run_sort(src, d % 4 == 0, N, d, sorted, sorted_bytes, static_cast<hipStream_t>(stream));
}

int vq_encode_l1_img(const uint8_t *img, int H, int W, int C, int bh, int bw, int pad_mode, const double *cb, int K,
                     int32_t *labels, double *min_dist, void *scratch, size_t scratch_bytes, void *ws, size_t ws_bytes,
                     void *stream) {
    if (!img || !cb) return fail(VQ_E_INVALID, "null image or codebook pointer");
    if (H <= 0 || W <= 0 || C <= 0 || bh <= 0 || bw <= 0) return fail(VQ_E_INVALID, "non-positive image or block size");
    if (K <= 0) return fail(VQ_E_INVALID, "need K > 0");
    if (K > MAX_K) return fail(VQ_E_UNSUPPORTED, "K=%d exceeds %d", K, MAX_K);
    if (pad_mode != VQ_PAD_EDGE && pad_mode != VQ_PAD_ZERO) return fail(VQ_E_INVALID, "unknown pad_mode %d", pad_mode);
    if (!labels || !scratch) return fail(VQ_E_INVALID, "null labels or scratch pointer");
    const int64_t N = (int64_t)((H + bh - 1) / bh) * ((W + bw - 1) / bw);
    const int64_t d64 = (int64_t)bh * bw * C;
    if (d64 > (1 << 24)) return fail(VQ_E_UNSUPPORTED, "block too large");
    const int d = (int)d64;
    if (prune_supported(K, d)) {
        // image -> window-sorted rows in one pass, then the pruned search on them: the block matrix is never built
        int rc = check_ws(ws, ws_bytes, K, d, N);
        if (rc) return rc;
        rc = vq_blocks_sort_img(img, H, W, C, bh, bw, pad_mode, scratch, scratch_bytes, stream);
        if (rc) return rc;
        rc = run_prepare(cb, K, d, ws, static_cast<hipStream_t>(stream));
        if (rc) return rc;
        return run_fast(nullptr, scratch, scratch_bytes, N, d, cb, K, labels, min_dist, nullptr, nullptr, ws,
                        static_cast<hipStream_t>(stream));
    }
    // shapes the pruned search does not serve: the scratch takes the plain block matrix
    if (scratch_bytes < (size_t)N * d) return fail(VQ_E_WORKSPACE, "scratch of %zu bytes needed, %zu given", (size_t)N * d, scratch_bytes);
    int rc = vq_blocks_u8(img, H, W, C, bh, bw, pad_mode, static_cast<uint8_t *>(scratch), stream);
    if (rc) return rc;
    return vq_encode_l1(static_cast<const uint8_t *>(scratch), N, d, cb, K, labels, min_dist, VQ_MODE_FAST, ws, ws_bytes, stream);
}

int vq_encode_l1_prepared_phases(const uint8_t *blocks, const void *sorted, size_t sorted_bytes, int64_t N, int d,
                                 const double *cb, int K, int32_t *labels, double *min_dist, int phases, void *ws,
                                 size_t ws_bytes, void *stream) {
    int rc = check_common(blocks, N, d, cb, K, sorted != nullptr && K > 0 && d > 0 && prune_supported(K, d));
    if (rc) return rc;
    if (!fast_supported(d)) return fail(VQ_E_UNSUPPORTED, "d=%d is beyond the prepared (fast) kernels", d);
    if (N > 0xfffffff0LL) return fail(VQ_E_UNSUPPORTED, "N=%lld exceeds 2^32 per call", (long long)N);
    if (phases < 1 || phases > 3) return fail(VQ_E_INVALID, "phases must be 1 (filter), 2 (fix-up) or 3 (both)");
    rc = check_ws(ws, ws_bytes, K, d, N);
    if (rc) return rc;
    if (N == 0) return VQ_OK;
    if (!labels) return fail(VQ_E_INVALID, "null labels pointer");
    return run_fast(blocks, sorted, sorted_bytes, N, d, cb, K, labels, min_dist, nullptr, nullptr, ws,
                    static_cast<hipStream_t>(stream), phases);
}

int vq_encode_l1_prepared(const uint8_t *blocks, const void *sorted, size_t sorted_bytes, int64_t N, int d,
                          const double *cb, int K, int32_t *labels, double *min_dist, void *ws, size_t ws_bytes,
                          void *stream) {
    return vq_encode_l1_prepared_phases(blocks, sorted, sorted_bytes, N, d, cb, K, labels, min_dist, 3, ws, ws_bytes, stream);
}

size_t vq_workspace_counters_offset(int K, int d, int64_t N) {
    if (K <= 0 || d <= 0 || N < 0) return 0;
    return ws_layout(K, d, N).qcount;
}

int vq_accum_labels(const uint8_t *blocks, int64_t N, int d, const int32_t *labels, int K, uint64_t *sums,
                    uint64_t *counts, void *stream) {
    if (N < 0 || d <= 0 || K <= 0) return fail(VQ_E_INVALID, "need N >= 0, d > 0, K > 0");
    if (N == 0) return VQ_OK;
    if (!blocks || !labels || !sums || !counts) return fail(VQ_E_INVALID, "null pointer");
    return run_accum(blocks, nullptr, N, d, labels, nullptr, K, reinterpret_cast<unsigned long long *>(sums),
                     reinterpret_cast<unsigned long long *>(counts), nullptr, static_cast<hipStream_t>(stream));
}

int vq_numpy_sum_f64(const double *v, int64_t N, double *out, double *chunk_sums, void *stream) {
    if (N < 0) return fail(VQ_E_INVALID, "need N >= 0");
    if (!out || !chunk_sums || (!v && N != 0)) return fail(VQ_E_INVALID, "null pointer");
    return run_numpy_sum(v, N, chunk_sums, out, static_cast<hipStream_t>(stream));
}

}  // extern "C"
namespace {
int lbg_accum_impl(const uint8_t *blocks, const void *sorted, size_t sorted_bytes, int64_t N, int d, const double *cb,
                   int K, uint64_t *sums, uint64_t *counts, int32_t *labels, double *min_dist, double *dist_sum,
                   double *chunk_sums, int mode, void *ws, size_t ws_bytes, void *stream, bool scratch_outputs);
}
extern "C" {

int vq_lbg_accum(const uint8_t *blocks, const void *sorted, size_t sorted_bytes, int64_t N, int d, const double *cb,
                 int K, uint64_t *sums, uint64_t *counts, int32_t *labels, double *min_dist, double *dist_sum,
                 double *chunk_sums, int mode, void *ws, size_t ws_bytes, void *stream) {
    return lbg_accum_impl(blocks, sorted, sorted_bytes, N, d, cb, K, sums, counts, labels, min_dist, dist_sum, chunk_sums, mode,
                          ws, ws_bytes, stream, false);
}

}  // extern "C"

namespace {
// scratch_outputs: labels and min_dist are scratch of a training iteration (nobody reads them afterwards): where the
// search settles its near-ties itself and the accumulation runs through the LDS table, both stay in the order of the
// sorted vectors -- full-width stores and loads instead of a scatter and a gather through the origin map, which across
// the eight L2s cost several times their bytes in HBM traffic.
int lbg_accum_impl(const uint8_t *blocks, const void *sorted, size_t sorted_bytes, int64_t N, int d, const double *cb,
                   int K, uint64_t *sums, uint64_t *counts, int32_t *labels, double *min_dist, double *dist_sum,
                   double *chunk_sums, int mode, void *ws, size_t ws_bytes, void *stream, bool scratch_outputs) {
    const bool via_sorted = sorted != nullptr && mode == VQ_MODE_FAST && K > 0 && d > 0 && prune_supported(K, d);
    int rc = check_common(blocks, N, d, cb, K, via_sorted);
    if (rc) return rc;
    if (!sums || !counts || !dist_sum) return fail(VQ_E_INVALID, "null sums, counts or dist_sum pointer");
    if (N != 0 && (!labels || !min_dist)) return fail(VQ_E_INVALID, "null labels or min_dist pointer");
    if (mode != VQ_MODE_EXACT && mode != VQ_MODE_FAST) return fail(VQ_E_INVALID, "unknown mode %d", mode);
    if (N > 0xfffffff0LL) return fail(VQ_E_UNSUPPORTED, "N=%lld exceeds 2^32 per call", (long long)N);
    rc = check_ws(ws, ws_bytes, K, d, N);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned long long *s64 = reinterpret_cast<unsigned long long *>(sums);
    unsigned long long *c64 = reinterpret_cast<unsigned long long *>(counts);
    bool sorted_io = false;
    const uint16_t *sorted_perm = nullptr;
    if (N > 0) {
        DeviceInfo di;
        rc = device_info(di);
        if (rc) return rc;
        // the accumulation pass also evaluates the distances when its table fits the LDS
        const bool dist_in_accum = accum_table_fits(K, d, di);
        if (mode == VQ_MODE_FAST && fast_supported(d)) {
            rc = run_prepare(cb, K, d, ws, st);
            if (rc) return rc;
            if (fused_acc_supported(sorted, K, d)) {
                rc = run_fast(blocks, sorted, sorted_bytes, N, d, cb, K, labels, min_dist, s64, c64, ws, st);
                if (rc) return rc;
            } else {
                const int dp2 = dp2_bucket(d);
                sorted_io = scratch_outputs && via_sorted && dist_in_accum &&
                            resolves_in_kernel(dp2, !prune_by_scalar_loads(K, dp2), true);
                rc = run_fast(blocks, sorted, sorted_bytes, N, d, cb, K, labels, dist_in_accum ? nullptr : min_dist,
                              nullptr, nullptr, ws, st, 3, sorted_io);
                if (rc) return rc;
                // the accumulation pass takes the rows from the sorted copy when the search did
                const uint8_t *acc_rows = blocks;
                const uint16_t *acc_perm = nullptr;
                if (via_sorted) {
                    SortedLayout SL = sorted_layout(d, N);
                    acc_rows = reinterpret_cast<const uint8_t *>(static_cast<const char *>(sorted) + SL.rows);
                    acc_perm = reinterpret_cast<const uint16_t *>(static_cast<const char *>(sorted) + SL.perm);
                }
                rc = run_accum(acc_rows, acc_perm, N, d, labels, cb, K, s64, c64, dist_in_accum ? min_dist : nullptr, st,
                               sorted_io ? N / NP_CHUNK * NP_CHUNK : 0);
                if (rc) return rc;
                sorted_perm = acc_perm;
            }
        } else {
            int grid = grid_for(N, 256, 16384);
            hipLaunchKernelGGL(k_assign_exact, dim3(grid), dim3(256), 0, st, blocks, N, d, cb, K, labels, min_dist, g_gate);
            LAUNCH_CHECK("k_assign_exact");
            rc = run_accum(blocks, nullptr, N, d, labels, cb, K, s64, c64, nullptr, st);
            if (rc) return rc;
        }
    }
    double *cs = chunk_sums ? chunk_sums
                            : reinterpret_cast<double *>(static_cast<char *>(ws) + ws_layout(K, d, N).chunks);
    if (sorted_io) return run_numpy_sum_sorted(min_dist, sorted_perm, N, cs, dist_sum, st);
    return run_numpy_sum(min_dist, N, cs, dist_sum, st);
}
}  // namespace

extern "C" {

int vq_lbg_pass_fused(int K, int d) {
    static const char dummy = 0;                               // (any non-null `sorted`: only the shape decides)
    return (K > 0 && d > 0 && fast_supported(d) && fused_acc_supported(&dummy, K, d)) ? 1 : 0;
}

size_t vq_lbg_state_bytes(int max_iterations) {
    if (max_iterations < 1) return 0;
    return align_up(offsetof(LbgState, rec) + (size_t)max_iterations * sizeof(LbgRecord), 256);
}

int vq_lbg_stage_begin(void *state, size_t state_bytes, int max_iterations, void *stream) {
    if (!state) return fail(VQ_E_INVALID, "null state pointer");
    if (max_iterations < 1) return fail(VQ_E_INVALID, "need max_iterations >= 1");
    if (state_bytes < vq_lbg_state_bytes(max_iterations))
        return fail(VQ_E_WORKSPACE, "state of %zu bytes needed, %zu given", vq_lbg_state_bytes(max_iterations), state_bytes);
    hipLaunchKernelGGL(k_lbg_stage_begin, dim3((max_iterations + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<LbgState *>(state), max_iterations);
    LAUNCH_CHECK("k_lbg_stage_begin");
    return VQ_OK;
}

}  // extern "C"

namespace {
// One LBG iteration up to the per-rank results (finish = false: sums / counts of this call's vectors, their per-chunk
// distance sums in chunk_out and the left-to-right total of those in total_out) or to the end (finish = true: centroid
// update in place and stop rule as well, for one process holding all n_total vectors).
int lbg_iterate_impl(const uint8_t *blocks, const void *sorted, size_t sorted_bytes, int64_t N, int d, double *cb, int K,
                     int64_t n_total, double threshold, int iteration, uint64_t *sums, uint64_t *counts, int32_t *labels,
                     double *min_dist, int mode, void *state, size_t state_bytes, void *ws, size_t ws_bytes, void *stream,
                     double *chunk_out, double *total_out, bool finish) {
    if (!state) return fail(VQ_E_INVALID, "null state pointer");
    if (iteration < 0 || state_bytes < vq_lbg_state_bytes(iteration + 1))
        return fail(VQ_E_INVALID, "iteration %d is outside the state buffer (%zu bytes)", iteration, state_bytes);
    if (finish && (n_total < 1 || N > n_total)) return fail(VQ_E_INVALID, "need n_total >= max(N, 1)");
    if (!sums || !counts || !cb) return fail(VQ_E_INVALID, "null sums, counts or codebook pointer");
    if (!finish && !chunk_out) return fail(VQ_E_INVALID, "null chunk_sums pointer");
    if (K <= 0 || d <= 0) return fail(VQ_E_INVALID, "need K > 0 and d > 0");
    hipStream_t st = static_cast<hipStream_t>(stream);
    LbgState *S = static_cast<LbgState *>(state);
    struct GateScope {                                        // the launches below carry this iteration's gate
        GateScope(const int32_t *p, int it) { g_gate = Gate{p, it}; }
        ~GateScope() { g_gate = Gate{nullptr, 0}; }
    } scope(&S->stopped_after, iteration);
    // what the previous call on this thread left behind is read before check_ws forgets it
    const bool chained = g_tail_left.ws == ws && g_tail_left.state == state && g_tail_left.iteration + 1 == iteration &&
                         g_tail_left.K == K && g_tail_left.d == d && g_tail_left.N == N;
    const bool prepared = chained && g_tail_left.prepared;
    int rc = check_ws(ws, ws_bytes, K, d, N);
    if (rc) return rc;
    // The fused form: the search accumulates into the replica tables, k_lbg_fold and k_lbg_finish_t do the rest.
    // The fold leaves the replica tables zeroed, so only the first iteration of a run of consecutive calls made by
    // one host thread (same workspace, same state, same N, iteration numbers counting up, no other call on the
    // workspace in between) zeroes them.
    const bool fused = N > 0 && mode == VQ_MODE_FAST && fast_supported(d) && fused_acc_supported(sorted, K, d) &&
                       min_dist != nullptr;
    if (fused) {
        rc = check_common(blocks, N, d, cb, K, true);
        if (rc) return rc;
        if (N > 0xfffffff0LL) return fail(VQ_E_UNSUPPORTED, "N=%lld exceeds 2^32 per call", (long long)N);
        const WsLayout L0 = ws_layout(K, d, 0), LN = ws_layout(K, d, N);
        char *w = static_cast<char *>(ws);
        struct ZeroScope {
            ~ZeroScope() { g_ws_zeroed = false; }
        } zscope;
        if (!chained) {
            HIP_TRY(hipMemsetAsync(w + L0.acc_rep, 0, LN.cbT - LN.acc_rep, st));
            HIP_TRY(hipMemsetAsync(w + LN.qcount, 0, (size_t)NQ * 64, st));
        }
        g_ws_zeroed = true;
        if (!prepared) {                                       // (else the previous iteration's last kernel has done it)
            rc = run_prepare(cb, K, d, ws, st);
            if (rc) return rc;
        }
        // the winning distances of whole chunks are stored in the order of the sorted vectors (full-width stores) and
        // put back into the caller's order by the fold kernel, in LDS; labels are written only when asked for
        rc = run_fast(blocks, sorted, sorted_bytes, N, d, cb, K, labels, min_dist, reinterpret_cast<unsigned long long *>(sums),
                      reinterpret_cast<unsigned long long *>(counts), ws, st, 1, true);
        if (rc) return rc;
        const int64_t nfull = N / NP_CHUNK;
        const int tabn = K * (d + 1);
        int64_t grid = nfull > (tabn + 511) / 512 ? nfull : (tabn + 511) / 512;
        if (grid > 4096) grid = 4096;
        double *chunk_sums = chunk_out ? chunk_out : reinterpret_cast<double *>(w + LN.chunks);
        unsigned long long *s64 = reinterpret_cast<unsigned long long *>(sums);
        unsigned long long *c64 = reinterpret_cast<unsigned long long *>(counts);
        const uint16_t *perm = reinterpret_cast<const uint16_t *>(static_cast<const char *>(sorted) + sorted_layout(d, N).perm);
        const size_t fold_lds = (size_t)NP_CHUNK * sizeof(double);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lbg_fold), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)fold_lds));
        hipLaunchKernelGGL(k_lbg_fold, dim3((unsigned)grid), dim3(512), fold_lds, st, min_dist, perm, nfull, chunk_sums,
                           reinterpret_cast<unsigned long long *>(w + L0.acc_rep), K, d, s64, c64,
                           reinterpret_cast<uint32_t *>(w + LN.qcount), g_gate);
        LAUNCH_CHECK("k_lbg_fold");
        bool prep_next = false;
        if (finish) {
            // shapes of the one-workgroup prepare kernel: its work for the NEXT iteration rides on this launch
            prep_next = d <= 16 && K <= 512;
            PrepArgs pa;
            pa.dp2 = dp2_bucket(d);
            pa.clim = key_layout(K, pa.dp2).clim;
            pa.hdr = reinterpret_cast<PrepHeader *>(w + L0.header);
            pa.c16 = reinterpret_cast<uint32_t *>(w + L0.c16);
            pa.bias = reinterpret_cast<uint32_t *>(w + L0.bias);
            pa.postab = reinterpret_cast<uint32_t *>(w + L0.postab);
            pa.cbT = reinterpret_cast<double *>(w + L0.cbT);
            const size_t hash_lds = (K <= PREP_HASH_CAP && K > 1) ? (size_t)K * 32 : 16;
            if (!prep_next)
                hipLaunchKernelGGL(k_lbg_finish_t<0>, dim3(1), dim3(1024), 0, st, min_dist, N, nfull, chunk_sums, s64, c64, cb,
                                   K, d, (double)n_total, threshold, S, g_gate, pa);
            else if (K <= 256)
                hipLaunchKernelGGL(k_lbg_finish_t<4>, dim3(1), dim3(1024), hash_lds, st, min_dist, N, nfull, chunk_sums, s64,
                                   c64, cb, K, d, (double)n_total, threshold, S, g_gate, pa);
            else
                hipLaunchKernelGGL(k_lbg_finish_t<2>, dim3(1), dim3(1024), hash_lds, st, min_dist, N, nfull, chunk_sums, s64,
                                   c64, cb, K, d, (double)n_total, threshold, S, g_gate, pa);
            LAUNCH_CHECK("k_lbg_finish");
        } else if (total_out || N % NP_CHUNK != 0) {
            // the trailing partial chunk and the total of this call's chunk sums
            hipLaunchKernelGGL(k_npsum_final, dim3(1), dim3(1024), 0, st, min_dist, N, nfull, chunk_sums,
                               total_out ? total_out : &S->dist_sum);
            LAUNCH_CHECK("k_npsum_final");
        }
        g_tail_left = TailLeft{ws, state, iteration, K, d, N, prep_next};
        return VQ_OK;
    }
    // one launch zeroes the caller's sums and counts, the queue counters and the replica tables of the workspace
    // (regions that are not 16-byte multiples fall back to plain fills)
    struct ZeroScope {
        ~ZeroScope() { g_ws_zeroed = false; }
    } zscope;
    {
        const WsLayout L0 = ws_layout(K, d, 0), LN = ws_layout(K, d, N);
        char *w = static_cast<char *>(ws);
        const size_t rep_bytes = LN.cbT - LN.acc_rep;              // 0 when this shape has no replica tables
        void *ptrs[4] = {sums, counts, w + LN.qcount, w + L0.acc_rep};
        size_t bytes[4] = {sizeof(uint64_t) * (size_t)K * d, sizeof(uint64_t) * (size_t)K, (size_t)NQ * 64, rep_bytes};
        bool ok = true;
        for (int r = 0; r < 4; ++r) ok = ok && (reinterpret_cast<uintptr_t>(ptrs[r]) % 16 == 0) && (bytes[r] % 16 == 0);
        if (ok) {
            ZeroRegions z;
            size_t most = 0;
            for (int r = 0; r < 4; ++r) {
                z.p[r] = static_cast<uint4 *>(ptrs[r]);
                z.n16[r] = bytes[r] / 16;
                most = z.n16[r] > most ? z.n16[r] : most;
            }
            hipLaunchKernelGGL(k_zero_regions, dim3(grid_for((int64_t)most, 256, 1024)), dim3(256), 0, st, z);
            LAUNCH_CHECK("k_zero_regions");
            g_ws_zeroed = true;
        } else {
            HIP_TRY(hipMemsetAsync(sums, 0, bytes[0], st));
            HIP_TRY(hipMemsetAsync(counts, 0, bytes[1], st));
        }
    }
    rc = lbg_accum_impl(blocks, sorted, sorted_bytes, N, d, cb, K, sums, counts, labels, min_dist,
                        (finish || !total_out) ? &S->dist_sum : total_out, chunk_out, mode, ws, ws_bytes, stream, true);
    if (rc) return rc;
    if (finish) {
        hipLaunchKernelGGL(k_finalize_rule, dim3(grid_for((int64_t)K * d, 256, 4096)), dim3(256), 0, st,
                           reinterpret_cast<const unsigned long long *>(sums),
                           reinterpret_cast<const unsigned long long *>(counts), cb, K, d, (double)n_total, threshold, S, g_gate);
        LAUNCH_CHECK("k_finalize_rule");
    }
    return VQ_OK;
}
}  // namespace

extern "C" {

int vq_lbg_iterate(const uint8_t *blocks, const void *sorted, size_t sorted_bytes, int64_t N, int d, double *cb, int K,
                   int64_t n_total, double threshold, int iteration, uint64_t *sums, uint64_t *counts, int32_t *labels,
                   double *min_dist, int mode, void *state, size_t state_bytes, void *ws, size_t ws_bytes, void *stream) {
    return lbg_iterate_impl(blocks, sorted, sorted_bytes, N, d, cb, K, n_total, threshold, iteration, sums, counts, labels,
                            min_dist, mode, state, state_bytes, ws, ws_bytes, stream, nullptr, nullptr, true);
}

int vq_lbg_iterate_local(const uint8_t *blocks, const void *sorted, size_t sorted_bytes, int64_t N, int d, const double *cb,
                         int K, int iteration, uint64_t *sums, uint64_t *counts, double *chunk_sums, double *local_total,
                         int32_t *labels, double *min_dist, int mode, void *state, size_t state_bytes, void *ws,
                         size_t ws_bytes, void *stream) {
    return lbg_iterate_impl(blocks, sorted, sorted_bytes, N, d, const_cast<double *>(cb), K, 0, 0.0, iteration, sums, counts,
                            labels, min_dist, mode, state, state_bytes, ws, ws_bytes, stream, chunk_sums, local_total, false);
}

int vq_lbg_iterate_finish(const uint64_t *sums, const uint64_t *counts, int records, int64_t record_stride,
                          const double *chunk_sums_all, int world, int64_t chunk_stride, const int *h_chunk_counts, double *cb,
                          int K, int d, int64_t n_total, double threshold, int iteration, void *state, size_t state_bytes,
                          void *ws, size_t ws_bytes, void *stream) {
    if (!state) return fail(VQ_E_INVALID, "null state pointer");
    if (iteration < 0 || state_bytes < vq_lbg_state_bytes(iteration + 1))
        return fail(VQ_E_INVALID, "iteration %d is outside the state buffer (%zu bytes)", iteration, state_bytes);
    if (!sums || !counts || !cb || !chunk_sums_all || !h_chunk_counts) return fail(VQ_E_INVALID, "null pointer");
    if (K <= 0 || d <= 0 || n_total < 1) return fail(VQ_E_INVALID, "need K > 0, d > 0, n_total >= 1");
    if (world < 1 || world > MAX_RANKS) return fail(VQ_E_UNSUPPORTED, "world=%d is outside 1 .. %d", world, MAX_RANKS);
    if (records < 1 || records > MAX_RANKS) return fail(VQ_E_INVALID, "records=%d is outside 1 .. %d", records, MAX_RANKS);
    if (records > 1 && record_stride < (int64_t)K * d) return fail(VQ_E_INVALID, "record_stride is shorter than a record");
    if (chunk_stride < 1) return fail(VQ_E_INVALID, "need chunk_stride >= 1");
    RankChunks rcs;
    rcs.world = world;
    rcs.stride = chunk_stride;
    for (int r = 0; r < MAX_RANKS; ++r) rcs.count[r] = 0;
    for (int r = 0; r < world; ++r) {
        if (h_chunk_counts[r] < 0 || h_chunk_counts[r] > chunk_stride)
            return fail(VQ_E_INVALID, "rank %d has %d chunk sums, the rows are %lld apart", r, h_chunk_counts[r],
                        (long long)chunk_stride);
        rcs.count[r] = h_chunk_counts[r];
    }
    LbgState *S = static_cast<LbgState *>(state);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // With the workspace of this thread's vq_lbg_iterate_local of the same iteration, the one-workgroup codebook
    // preparation of the next iteration rides on this launch (the shapes it serves), and the next
    // vq_lbg_iterate_local skips its own.
    const TailLeft left = g_tail_left;                         // (check_ws forgets it)
    const bool same = ws != nullptr && left.ws == ws && left.state == state && left.iteration == iteration && left.K == K &&
                      left.d == d;
    PrepArgs pa = {};
    bool prep_next = false;
    if (ws) {
        int rc = check_ws(ws, ws_bytes, K, d, 0);
        if (rc) return rc;
        prep_next = same && d <= 16 && K <= 512;
        const WsLayout L0 = ws_layout(K, d, 0);
        char *w = static_cast<char *>(ws);
        pa.dp2 = dp2_bucket(d);
        pa.clim = key_layout(K, pa.dp2).clim;
        pa.hdr = reinterpret_cast<PrepHeader *>(w + L0.header);
        pa.c16 = reinterpret_cast<uint32_t *>(w + L0.c16);
        pa.bias = reinterpret_cast<uint32_t *>(w + L0.bias);
        pa.postab = reinterpret_cast<uint32_t *>(w + L0.postab);
        pa.cbT = reinterpret_cast<double *>(w + L0.cbT);
    }
    const unsigned long long *s64 = reinterpret_cast<const unsigned long long *>(sums);
    const unsigned long long *c64 = reinterpret_cast<const unsigned long long *>(counts);
    const Gate gate{&S->stopped_after, iteration};
    const size_t hash_lds = (K <= PREP_HASH_CAP && K > 1) ? (size_t)K * 32 : 16;
    if (!prep_next)
        hipLaunchKernelGGL(k_lbg_finish_ranks_t<0>, dim3(1), dim3(1024), 0, st, chunk_sums_all, rcs, s64, c64, records,
                           record_stride, cb, K, d, (double)n_total, threshold, S, gate, pa);
    else if (K <= 256)
        hipLaunchKernelGGL(k_lbg_finish_ranks_t<4>, dim3(1), dim3(1024), hash_lds, st, chunk_sums_all, rcs, s64, c64, records,
                           record_stride, cb, K, d, (double)n_total, threshold, S, gate, pa);
    else
        hipLaunchKernelGGL(k_lbg_finish_ranks_t<2>, dim3(1), dim3(1024), hash_lds, st, chunk_sums_all, rcs, s64, c64, records,
                           record_stride, cb, K, d, (double)n_total, threshold, S, gate, pa);
    LAUNCH_CHECK("k_lbg_finish_ranks");
    if (same) {
        g_tail_left = left;
        g_tail_left.prepared = prep_next;
    }
    return VQ_OK;
}

void vq_profile_next_search(void *ev_start, void *ev_stop) {
    g_prof_start = static_cast<hipEvent_t>(ev_start);
    g_prof_stop = static_cast<hipEvent_t>(ev_stop);
}

int vq_sorted_forget(const void *sorted) {
    std::lock_guard<std::mutex> lock(g_sorted_mutex);
    return g_sorted_known.erase(reinterpret_cast<uintptr_t>(sorted)) ? 1 : 0;
}

int vq_lbg_finalize(const uint64_t *sums, const uint64_t *counts, const double *cb_old, int K, int d, double *cb_new,
                    void *stream) {
    if (!sums || !counts || !cb_old || !cb_new) return fail(VQ_E_INVALID, "null pointer");
    if (K <= 0 || d <= 0) return fail(VQ_E_INVALID, "need K > 0 and d > 0");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int grid = grid_for((int64_t)K * d, 256, 4096);
    hipLaunchKernelGGL(k_finalize, dim3(grid), dim3(256), 0, st, reinterpret_cast<const unsigned long long *>(sums),
                       reinterpret_cast<const unsigned long long *>(counts), cb_old, K, d, cb_new);
    LAUNCH_CHECK("k_finalize");
    return VQ_OK;
}

int vq_lbg_split(const double *cb, int K, int d, double eps, double *cb2, void *stream) {
    if (!cb || !cb2) return fail(VQ_E_INVALID, "null pointer");
    if (K <= 0 || d <= 0) return fail(VQ_E_INVALID, "need K > 0 and d > 0");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int64_t n = (int64_t)K * d;
    int grid = grid_for(n, 256, 4096);
    hipLaunchKernelGGL(k_split, dim3(grid), dim3(256), 0, st, cb, n, 1 + eps, 1 - eps, cb2);
    LAUNCH_CHECK("k_split");
    return VQ_OK;
}

int vq_decode_u8(const int32_t *labels, int nr, int nc, const double *cb, int K, int bh, int bw, int C, uint8_t *img,
                 void *ws, size_t ws_bytes, void *stream) {
    if (!labels || !cb || !img) return fail(VQ_E_INVALID, "null pointer");
    if (nr < 0 || nc < 0 || K <= 0 || bh <= 0 || bw <= 0 || C <= 0) return fail(VQ_E_INVALID, "bad shape");
    int64_t d64 = (int64_t)bh * bw * C;
    if (d64 > (1 << 24)) return fail(VQ_E_UNSUPPORTED, "block too large");
    int d = (int)d64;
    int rc = check_ws(ws, ws_bytes, K, d, 0);
    if (rc) return rc;
    if (nr == 0 || nc == 0) return VQ_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    WsLayout L = ws_layout(K, d, 0);
    uint8_t *cb8 = reinterpret_cast<uint8_t *>(static_cast<char *>(ws) + L.cb_u8);
    const int64_t n = (int64_t)K * d;
    const int64_t rowbytes = (int64_t)nc * bw * C;
    if (rowbytes > 0x7fffffffLL || (int64_t)nr * bh > 0x7fffffffLL) return fail(VQ_E_UNSUPPORTED, "image too large");
    const bool lds_cb = n <= 64 * 1024;
    const size_t lds = lds_cb ? (size_t)((n + 15) / 16 * 16) : 0;
    DeviceInfo di;
    int rc2 = device_info(di);
    if (rc2) return rc2;
    if (d == 16 && bh == 4 && nc % 4 == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(labels) & 15) == 0) {
        const int64_t quads = (int64_t)nr * (nc / 4);
        const int grid = grid_for((quads + 1) / 2, 1024, di.cus * 2);
        if (lds_cb) {
            if (lds > 48 * 1024)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_decode_quads<true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_decode_quads<true>), dim3(grid), dim3(1024), lds, st, labels, nr, nc, cb, cb8, K, img);
        } else {
            hipLaunchKernelGGL(k_cb_to_u8, dim3(grid_for(n, 256, 4096)), dim3(256), 0, st, cb, n, cb8);
            LAUNCH_CHECK("k_cb_to_u8");
            hipLaunchKernelGGL((k_decode_quads<false>), dim3(grid), dim3(1024), 0, st, labels, nr, nc, cb, cb8, K, img);
        }
        LAUNCH_CHECK("k_decode_quads");
        return VQ_OK;
    }
    // the general kernels run many workgroups: beyond a few thousand entries the codebook is truncated once, by its
    // own launch, and the workgroups stage the uint8 table
    const bool convert_once = !lds_cb || (n > 4096 && (reinterpret_cast<uintptr_t>(cb8) & 15) == 0);
    if (convert_once) {
        hipLaunchKernelGGL(k_cb_to_u8, dim3(grid_for(n, 256, 4096)), dim3(256), 0, st, cb, n, cb8);
        LAUNCH_CHECK("k_cb_to_u8");
        if (lds_cb) cb = nullptr;                              // (the kernels' signal to stage from cb8)
    }
    const uintptr_t iptr = reinterpret_cast<uintptr_t>(img);
    const bool seg4 = (bw * C) % 4 == 0;
    const int G = (seg4 && rowbytes % 16 == 0 && (iptr & 15) == 0) ? 16 : ((seg4 && (iptr & 3) == 0) ? 4 : 1);
    if (G == 1 && nc >= 64 && d <= 768) {
        // (the byte-wise kernel otherwise: one thread per output byte)
        int TB = (int)((48 * 1024) / d) / 256 * 256;
        if (TB > 4096) TB = 4096;
        if (TB < 64) TB = 64;
        const int rs = (TB * (bw * C) + 4 + 3) / 4 * 4;
        const size_t lds_s = (size_t)bh * rs + (lds_cb ? lds : 0);
        const int64_t tiles = (int64_t)nr * ((nc + TB - 1) / TB);
        const int grid_s = grid_for(tiles, 1, di.cus * 8);
        if (lds_cb) {
            if (lds_s > 48 * 1024)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_decode_strip<true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
            hipLaunchKernelGGL((k_decode_strip<true>), dim3(grid_s), dim3(256), lds_s, st, labels, nr, nc, cb, cb8, K, bh, bw, C,
                               img, TB, rs);
        } else {
            if (lds_s > 48 * 1024)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_decode_strip<false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
            hipLaunchKernelGGL((k_decode_strip<false>), dim3(grid_s), dim3(256), lds_s, st, labels, nr, nc, cb, cb8, K, bh, bw, C,
                               img, TB, rs);
        }
        LAUNCH_CHECK("k_decode_strip");
        return VQ_OK;
    }
    // a fixed number of workgroups (each stages the codebook once) laid out as (x over a row, y over rows)
    const int64_t groups_per_row = rowbytes / G;
    int gx = grid_for(groups_per_row, 256, 64);
    int64_t gy64 = ((int64_t)di.cus * 8 + gx - 1) / gx;
    if (gy64 > (int64_t)nr * bh) gy64 = (int64_t)nr * bh;
    if (gy64 > 65535) gy64 = 65535;
    dim3 grid(gx, (unsigned)gy64);
    // word-wide groups: one thread writes all bh pixel rows of its block row (labels fetched once)
    if (G != 1) {
        int64_t gyb = ((int64_t)di.cus * 8 + gx - 1) / gx;
        if (gyb > nr) gyb = nr;
        if (gyb > 65535) gyb = 65535;
        dim3 gridb(gx, (unsigned)gyb);
#define VQ_LAUNCH_DECODE_B(GG)                                                                                      \
    do {                                                                                                            \
        if (lds_cb) {                                                                                               \
            if (lds > 48 * 1024)                                                                                    \
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_decode_blockrows_u8<GG, true>),        \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                 \
            hipLaunchKernelGGL((k_decode_blockrows_u8<GG, true>), gridb, dim3(256), lds, st, labels, nr, nc, cb, cb8, K, \
                               bh, bw, C, img);                                                                     \
        } else {                                                                                                    \
            hipLaunchKernelGGL((k_decode_blockrows_u8<GG, false>), gridb, dim3(256), 0, st, labels, nr, nc, cb, cb8, K,  \
                               bh, bw, C, img);                                                                     \
        }                                                                                                           \
    } while (0)
        if (G == 16) VQ_LAUNCH_DECODE_B(16);
        else VQ_LAUNCH_DECODE_B(4);
#undef VQ_LAUNCH_DECODE_B
        LAUNCH_CHECK("k_decode_blockrows_u8");
        return VQ_OK;
    }
    if (lds_cb) {
        if (lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_decode_u8<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_decode_u8<true>), grid, dim3(256), lds, st, labels, nr, nc, cb, cb8, K, bh, bw, C, img);
    } else {
        hipLaunchKernelGGL((k_decode_u8<false>), grid, dim3(256), 0, st, labels, nr, nc, cb, cb8, K, bh, bw, C, img);
    }
    LAUNCH_CHECK("k_decode_u8");
    return VQ_OK;
}

size_t vq_packed_bytes(int64_t N, int K) {
    if (N <= 0 || K <= 1) return 0;
    int bits = ceil_log2(K);
    return (size_t)((N * bits + 7) / 8);
}

int vq_pack_labels(const int32_t *labels, int64_t N, int K, uint8_t *out, void *stream) {
    if (N < 0 || K <= 0) return fail(VQ_E_INVALID, "need N >= 0 and K > 0");
    size_t nbytes = vq_packed_bytes(N, K);
    if (nbytes == 0) return VQ_OK;
    if (!labels || !out) return fail(VQ_E_INVALID, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int bits = ceil_log2(K);
    hipLaunchKernelGGL(k_pack_labels, dim3(grid_for((int64_t)nbytes, 256, 8192)), dim3(256), 0, st, labels, N, bits, out,
                       (int64_t)nbytes);
    LAUNCH_CHECK("k_pack_labels");
    return VQ_OK;
}

int vq_unpack_labels(const uint8_t *packed, int64_t N, int K, int32_t *labels, void *stream) {
    if (N < 0 || K <= 0) return fail(VQ_E_INVALID, "need N >= 0 and K > 0");
    if (N == 0) return VQ_OK;
    if (!labels) return fail(VQ_E_INVALID, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int bits = ceil_log2(K);
    if (bits == 0) {
        HIP_TRY(hipMemsetAsync(labels, 0, sizeof(int32_t) * (size_t)N, st));
        return VQ_OK;
    }
    if (!packed) return fail(VQ_E_INVALID, "null pointer");
    hipLaunchKernelGGL(k_unpack_labels, dim3(grid_for(N, 256, 8192)), dim3(256), 0, st, packed, N, bits, labels);
    LAUNCH_CHECK("k_unpack_labels");
    return VQ_OK;
}

}  // extern "C"
