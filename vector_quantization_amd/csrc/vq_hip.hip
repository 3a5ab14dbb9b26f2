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
//   2. Re-check.  Otherwise the whole wave evaluates that vector against every codeword in
//      sequential fp64 (lanes <-> codewords, coalesced reads of a transposed fp64 codebook) and
//      reduces (distance, index) lexicographically, which is numpy's first-minimum rule.
//   A codeword whose row is bit-identical to an earlier row can never win (ties go to the lower
//   index), so the filter gives it a key that never competes; duplicated codewords (the twin that
//   the split of an all-zero centroid produces) therefore do not force re-checks.
// EXACT mode evaluates every distance in sequential fp64 and is the on-device cross-check.
//
// No entry point allocates memory; scratch comes in through the workspace argument.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "vq_hip.h"

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

constexpr int ABI_VERSION = 1;
constexpr int WAVE = 64;
constexpr int MAX_GRID = 4096;          // upper bound on workgroups of the assignment kernels
constexpr int MAX_D_FAST = 192;         // largest vector dimension of the SAD filter kernels
constexpr int MAX_K = 1 << 20;

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Workspace layout (all offsets multiples of 256 bytes).
struct WsLayout {
    size_t header;     // PrepHeader
    size_t c16;        // K * dp2 uint32: codebook as packed pairs of 16-bit fixed point
    size_t bias;       // K uint32: key bias (index, plus the never-wins bit for duplicate rows)
    size_t cbT;        // d * K doubles: transposed fp64 codebook
    size_t cb_u8;      // K * d bytes: codebook truncated to uint8 (decode)
    size_t partials;   // MAX_GRID doubles: per-workgroup distance sums
    size_t total;
};

int dp2_bucket(int d);   // forward

WsLayout ws_layout(int K, int d) {
    WsLayout L;
    int dp2 = dp2_bucket(d);
    if (dp2 <= 0) dp2 = (d + 1) / 2;
    size_t off = 0;
    L.header = off;   off += 256;
    L.c16 = off;      off += align_up((size_t)K * dp2 * 4, 256);
    L.bias = off;     off += align_up((size_t)K * 4, 256);
    L.cbT = off;      off += align_up((size_t)K * d * 8, 256);
    L.cb_u8 = off;    off += align_up((size_t)K * d, 256);
    L.partials = off; off += align_up((size_t)MAX_GRID * 8, 256);
    L.total = off;
    return L;
}

// Header that the prepare kernel writes and the assignment kernel reads.
struct PrepHeader {
    uint32_t shift;     // s: fixed-point scale is 2^s
    uint32_t emax;      // integer bound, strictly greater than every E_k (units of 2^-s)
    uint32_t cmax;      // ceil of the largest codebook entry, saturated
    uint32_t pad[61];
};
static_assert(sizeof(PrepHeader) == 256, "header is one 256-byte slot");

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
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

// Whole-wave exact search for ONE vector (xrow: d bytes, wave-uniform address).  Lane l evaluates
// codewords l, l+64, ... in sequential fp64 against the transposed codebook cbT[e*K + k].
__device__ __forceinline__ void exact_search_wave(const uint8_t *__restrict__ xrow, int d,
                                                  const double *__restrict__ cbT, int K, int lane,
                                                  int &out_idx, double &out_dist) {
    double best = __builtin_huge_val();
    int bidx = 0x7fffffff;
    for (int k0 = 0; k0 < K; k0 += 4 * WAVE) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int e = 0; e < d; ++e) {
            double xv = (double)xrow[e];
            const double *col = cbT + (size_t)e * K + k0 + lane;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int k = k0 + j * WAVE + lane;
                double c = (k < K) ? col[j * WAVE] : 0.0;
                acc[j] += fabs(xv - c);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int k = k0 + j * WAVE + lane;
            if (k < K && acc[j] < best) {
                best = acc[j];
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
// One thread per output byte group: G = 4 consecutive bytes when a block row segment (bw*C bytes)
// is a multiple of 4, else 1.  Consecutive threads write consecutive bytes of the block matrix.
template <int G>
__global__ __launch_bounds__(256) void k_blocks_u8(const uint8_t *__restrict__ img, int H, int W, int C, int bh,
                                                   int bw, int pad_zero, uint8_t *__restrict__ blocks, int nc,
                                                   int64_t total_groups) {
    const int seg = bw * C;            // bytes of one block row segment
    const int d = bh * seg;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total_groups;
         g += (int64_t)gridDim.x * blockDim.x) {
        int64_t o = g * G;             // first output byte
        int64_t n = o / d;
        int e = (int)(o - n * d);
        int r = (int)(n / nc), q = (int)(n - (int64_t)r * nc);
        int i = e / seg, rem = e - i * seg;
        int y = r * bh + i;
        int yc = y < H ? y : H - 1;
        uint8_t out[G];
#pragma unroll
        for (int t = 0; t < G; ++t) {
            int j = (rem + t) / C, ch = (rem + t) - j * C;
            int x = q * bw + j;
            int xc = x < W ? x : W - 1;
            uint8_t v = img[((size_t)yc * W + xc) * C + ch];
            if (pad_zero && (y >= H || x >= W)) v = 0;
            out[t] = v;
        }
        if (G == 4) {
            uint32_t w = (uint32_t)out[0] | ((uint32_t)out[1 % G] << 8) | ((uint32_t)out[2 % G] << 16) |
                         ((uint32_t)out[3 % G] << 24);
            *reinterpret_cast<uint32_t *>(blocks + o) = w;
        } else {
            blocks[o] = out[0];
        }
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

// ------------------------------------------------------------------------------------------------
// codebook preparation for the SAD filter (one workgroup)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_prepare(const double *__restrict__ cb, int K, int d, int dp2,
                                                  PrepHeader *__restrict__ hdr, uint32_t *__restrict__ c16,
                                                  uint32_t *__restrict__ bias, double *__restrict__ cbT) {
    __shared__ uint32_t s_cmax;
    __shared__ uint32_t s_emax;
    if (threadIdx.x == 0) {
        s_cmax = 0;
        s_emax = 0;
    }
    __syncthreads();
    // phase A: largest entry (ceil, saturated; non-finite counts as huge)
    const int64_t n = (int64_t)K * d;
    uint32_t lmax = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        double c = cb[i];
        uint32_t m;
        if (!(c == c) || c > 1.0e9 || c < -1.0e9) m = 0x7fffffffu;
        else if (c <= 0.0) m = 0;
        else m = (uint32_t)ceil(c);
        lmax = m > lmax ? m : lmax;
    }
    atomicMax(&s_cmax, lmax);
    __syncthreads();
    const uint32_t cmax = s_cmax;
    int s = 8;
    while (s > 0 && ((uint64_t)cmax << s) > 65535ull) --s;
    const double scale = (double)(1u << s);
    // phase B: one thread per codeword
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const double *row = cb + (size_t)k * d;
        double err = 0.0;
        for (int p = 0; p < dp2; ++p) {
            uint32_t pair = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int e = 2 * p + h;
                if (e < d) {
                    double c = row[e];
                    double t = c * scale;
                    double q = rint(t);
                    if (!(q >= 0.0)) q = 0.0;          // also catches NaN
                    if (q > 65535.0) q = 65535.0;
                    double de = fabs(t - q);
                    if (!(de == de)) de = 4.0e9;
                    err += de;
                    pair |= ((uint32_t)q) << (16 * h);
                    cbT[(size_t)e * K + k] = c;
                }
            }
            c16[(size_t)k * dp2 + p] = pair;
        }
        // strict integer upper bound of err (plus slack for the fp64 rounding of err itself)
        uint32_t ek = (err < 2.0e9) ? (uint32_t)err + 2u : 0x7fffffffu;
        atomicMax(&s_emax, ek);
        // duplicate of an earlier row?  (bitwise comparison: identical rows give identical distances)
        uint32_t dup = 0;
        const unsigned long long *rowb = reinterpret_cast<const unsigned long long *>(row);
        for (int j = 0; j < k && !dup; ++j) {
            const unsigned long long *rj = reinterpret_cast<const unsigned long long *>(cb + (size_t)j * d);
            if (rj[0] != rowb[0]) continue;
            int e = 1;
            while (e < d && rj[e] == rowb[e]) ++e;
            if (e == d) dup = 1;
        }
        bias[k] = (uint32_t)k | (dup << 31);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        hdr->shift = (uint32_t)s;
        hdr->emax = s_emax;
        hdr->cmax = cmax;
    }
}

// ------------------------------------------------------------------------------------------------
// nearest codeword, FAST mode (VQ_Full.py:77-78 / :124-125), optionally fused with the
// per-cluster accumulation of VQ_Full.py:81-90
// ------------------------------------------------------------------------------------------------
constexpr int ACC_NONE = 0;     // labels (+ distances) only
constexpr int ACC_LDS = 1;      // cluster sums / counts through an LDS table, flushed with 64-bit atomics
constexpr int ACC_GLOBAL = 2;   // cluster sums / counts straight to global 64-bit atomics

constexpr int KEY_PACKED = 0;   // key = (S << kbits) + (k | dupbit): needs S << kbits < 2^31
constexpr int KEY_SPLIT = 1;    // value and index tracked separately

struct AssignArgs {
    const uint8_t *blocks;
    int64_t N;
    int d;
    const double *cb;           // (K, d) row-major fp64
    int K;
    int kbits;
    const PrepHeader *hdr;
    const uint32_t *c16;
    const uint32_t *bias;
    const double *cbT;
    int32_t *labels;            // nullable
    double *min_dist;           // nullable
    unsigned long long *sums;   // ACC_*: (K, d)
    unsigned long long *counts; // ACC_*: (K)
    double *partials;           // ACC_*: per-workgroup distance sums
    int acc_mode;
    int vec4;                   // rows may be read as aligned 32-bit words (d % 4 == 0, base 4-byte aligned)
    int64_t ntiles;
};

template <int DP2, int R, int KEYMODE>
__global__ __launch_bounds__(512) void k_assign_fast(const AssignArgs a, const uint32_t *__restrict__ c16,
                                                     const uint32_t *__restrict__ bias) {
    extern __shared__ __align__(16) uint32_t s_tab[];   // ACC_LDS: K * (d + 1) sums|count; then 16 doubles for the reduction
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int BLOCK = blockDim.x;
    const int d = a.d, K = a.K;
    const int acc_mode = a.acc_mode;
    const int tabn = K * (d + 1);
    const bool want_dist = (a.min_dist != nullptr) || (acc_mode != ACC_NONE);

    if (acc_mode == ACC_LDS) {
        for (int i = tid; i < tabn; i += BLOCK) s_tab[i] = 0u;
        __syncthreads();
    }
    const uint32_t s = a.hdr->shift;
    const uint32_t emax = a.hdr->emax;
    const uint32_t window = emax >= 0x3fffffffu ? 0xffffffffu : 2u * emax;   // ambiguous iff S2 - S1 <= window
    const uint32_t kbits = (uint32_t)a.kbits;
    const uint32_t kmask = (1u << kbits) - 1u;
    // The shift count lives in a vector register: v_lshl_add_u32 can then take the per-codeword
    // bias as its single scalar operand (one instruction instead of shift + add).
    uint32_t kbits_v = kbits;
    asm volatile("" : "+v"(kbits_v));
    double dsum = 0.0;
    int tiles_done = 0;

    for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int64_t base = tile * (int64_t)(BLOCK * R);
        // ---- load R vectors per lane and convert to packed 16-bit fixed point ----
        uint32_t x[R][DP2];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int64_t n = base + (int64_t)r * BLOCK + tid;
            if (n >= a.N) n = a.N - 1;                 // clamp: loads stay in bounds, result is discarded
            const uint8_t *row = a.blocks + (size_t)n * d;
            // Out-of-row positions (d below the template's padded size) are loaded from a clamped
            // index and then zeroed, so that no load sits behind a branch.
            if (a.vec4) {
                const uint32_t *row4 = reinterpret_cast<const uint32_t *>(row);
                const int nw = d >> 2;
#pragma unroll
                for (int p = 0; p < DP2; p += 2) {
                    const int wi = p >> 1;
                    uint32_t w = row4[wi < nw ? wi : 0];
                    w = wi < nw ? w : 0u;
                    x[r][p] = ((w & 0xffu) | ((w & 0xff00u) << 8)) << s;
                    x[r][p + 1] = (((w >> 16) & 0xffu) | ((w >> 8) & 0xff0000u)) << s;
                }
            } else {
#pragma unroll
                for (int p = 0; p < DP2; ++p) {
                    const int e0 = 2 * p, e1 = 2 * p + 1;
                    uint32_t b0 = row[e0 < d ? e0 : 0];
                    uint32_t b1 = row[e1 < d ? e1 : 0];
                    b0 = e0 < d ? b0 : 0u;
                    b1 = e1 < d ? b1 : 0u;
                    x[r][p] = (b0 | (b1 << 16)) << s;
                }
            }
        }
        // ---- filter: two smallest keys per vector ----
        uint32_t m1[R], m2[R], bi[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            m1[r] = 0xffffffffu;
            m2[r] = 0xffffffffu;
            bi[r] = 0;
        }
        // Codewords are fetched by scalar loads (wave-uniform addresses) in batches of U, so that
        // one scalar-memory wait is paid per U * R * (DP2 + 3) vector instructions.
        constexpr int U = DP2 <= 8 ? 4 : (DP2 <= 16 ? 2 : 1);
        auto step = [&](const uint32_t *cw, uint32_t b, int k) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                uint32_t acc = 0;
#pragma unroll
                for (int p = 0; p < DP2; ++p) acc = __builtin_amdgcn_sad_u16(x[r][p], cw[p], acc);
                if (KEYMODE == KEY_PACKED) {
                    uint32_t key = (acc << kbits_v) + b;
                    m2[r] = umed3(key, m1[r], m2[r]);
                    m1[r] = min(key, m1[r]);
                } else {
                    uint32_t v = acc + ((b >> 31) << 30);      // duplicate rows never compete
                    m2[r] = umed3(v, m1[r], m2[r]);
                    bi[r] = (v < m1[r]) ? (uint32_t)k : bi[r];
                    m1[r] = min(v, m1[r]);
                }
            }
        };
        int k = 0;
        for (; k + U <= K; k += U) {
            uint32_t cw[U][DP2], b[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
#pragma unroll
                for (int p = 0; p < DP2; ++p) cw[j][p] = c16[(size_t)(k + j) * DP2 + p];
                b[j] = bias[k + j];
            }
#pragma unroll
            for (int j = 0; j < U; ++j) step(cw[j], b[j], k + j);
        }
        for (; k < K; ++k) {
            uint32_t cw[DP2];
#pragma unroll
            for (int p = 0; p < DP2; ++p) cw[p] = c16[(size_t)k * DP2 + p];
            step(cw, bias[k], k);
        }
        // ---- resolve, measure, write, accumulate ----
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t n = base + (int64_t)r * BLOCK + tid;
            const bool valid = n < a.N;
            uint32_t s1, s2;
            int idx;
            if (KEYMODE == KEY_PACKED) {
                idx = (int)(m1[r] & kmask);
                s1 = m1[r] >> kbits;
                s2 = m2[r] >> kbits;
            } else {
                idx = (int)bi[r];
                s1 = m1[r];
                s2 = m2[r];
            }
            bool amb = valid && (s2 - s1 <= window);
            double dist = 0.0;
            bool have_dist = false;
            unsigned long long mask = __ballot(amb);
            while (mask) {
                int src = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                int64_t nsrc = base + (int64_t)r * BLOCK + (tid - lane + src);   // wave-uniform
                int ri;
                double rd;
                exact_search_wave(a.blocks + (size_t)nsrc * d, d, a.cbT, K, lane, ri, rd);
                if (lane == src) {
                    idx = ri;
                    dist = rd;
                    have_dist = true;
                }
            }
            if (valid && want_dist && !have_dist) {
                const double *crow = a.cb + (size_t)idx * d;
                double acc = 0.0;
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
                }
                dist = acc;
            }
            if (valid) {
                if (a.labels) a.labels[n] = idx;
                if (a.min_dist) a.min_dist[n] = dist;
                if (acc_mode != ACC_NONE) dsum += dist;
                if (acc_mode == ACC_LDS) {
                    uint32_t *t = s_tab + (size_t)idx * (d + 1);
#pragma unroll
                    for (int p = 0; p < DP2; ++p) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            int e = 2 * p + h;
                            if (e < d) atomicAdd(&t[e], (x[r][p] >> (s + 16 * h)) & 0xffu);
                        }
                    }
                    atomicAdd(&t[d], 1u);
                } else if (acc_mode == ACC_GLOBAL) {
                    unsigned long long *t = a.sums + (size_t)idx * d;
#pragma unroll
                    for (int p = 0; p < DP2; ++p) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            int e = 2 * p + h;
                            if (e < d) atomicAdd(&t[e], (unsigned long long)((x[r][p] >> (s + 16 * h)) & 0xffu));
                        }
                    }
                    atomicAdd(&a.counts[idx], 1ull);
                }
            }
        }
        // 32-bit LDS sums hold 255 * (vectors per workgroup): flush long before they can wrap
        if (acc_mode == ACC_LDS && (++tiles_done & 1023) == 0) {
            __syncthreads();
            for (int i = tid; i < tabn; i += BLOCK) {
                uint32_t v = s_tab[i];
                if (v) {
                    int k = i / (d + 1), e = i - k * (d + 1);
                    if (e < d) atomicAdd(&a.sums[(size_t)k * d + e], (unsigned long long)v);
                    else atomicAdd(&a.counts[k], (unsigned long long)v);
                    s_tab[i] = 0u;
                }
            }
            __syncthreads();
        }
    }

    if (acc_mode != ACC_NONE) {
        __syncthreads();
        if (acc_mode == ACC_LDS) {
            for (int i = tid; i < tabn; i += BLOCK) {
                uint32_t v = s_tab[i];
                if (v) {
                    int k = i / (d + 1), e = i - k * (d + 1);
                    if (e < d) atomicAdd(&a.sums[(size_t)k * d + e], (unsigned long long)v);
                    else atomicAdd(&a.counts[k], (unsigned long long)v);
                }
            }
            __syncthreads();
        }
        // fixed-order reduction of the distance sums: xor tree inside a wave, then waves in order
        double *s_red = reinterpret_cast<double *>(s_tab);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) dsum += __shfl_xor(dsum, off, WAVE);
        if (lane == 0) s_red[tid >> 6] = dsum;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < (BLOCK >> 6); ++w) t += s_red[w];
            a.partials[blockIdx.x] = t;
        }
    }
}

// Sum of the per-workgroup distance partials in index order (deterministic).
__global__ void k_sum_partials(const double *__restrict__ partials, int n, double *__restrict__ out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < n; ++i) t += partials[i];
        out[0] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// nearest codeword, EXACT mode: every distance in sequential fp64, one vector per lane
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assign_exact(const uint8_t *__restrict__ blocks, int64_t N, int d,
                                                      const double *__restrict__ cb, int K,
                                                      int32_t *__restrict__ labels, double *__restrict__ min_dist,
                                                      unsigned long long *__restrict__ sums,
                                                      unsigned long long *__restrict__ counts,
                                                      double *__restrict__ partials) {
    __shared__ double s_red[4];
    double dsum = 0.0;
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
        if (sums) {
            dsum += best;
            for (int e = 0; e < d; ++e) {
                unsigned v = x[e];
                if (v) atomicAdd(&sums[(size_t)bidx * d + e], (unsigned long long)v);
            }
            atomicAdd(&counts[bidx], 1ull);
        }
    }
    if (partials) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) dsum += __shfl_xor(dsum, off, WAVE);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = dsum;
        __syncthreads();
        if (threadIdx.x == 0) partials[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
}

// Per-cluster sums / counts for given labels: straight 64-bit atomics (used by EXACT mode and by
// vq_accum_labels; not on the fast path).
__global__ __launch_bounds__(256) void k_accum_labels(const uint8_t *__restrict__ blocks, int64_t N, int d,
                                                      const int32_t *__restrict__ labels, int K,
                                                      unsigned long long *__restrict__ sums,
                                                      unsigned long long *__restrict__ counts) {
    const int64_t total = N * (int64_t)(d + 1);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t n = i / (d + 1);
        int e = (int)(i - n * (d + 1));
        int k = labels[n];
        if ((unsigned)k >= (unsigned)K) continue;
        if (e < d) {
            unsigned v = blocks[(size_t)n * d + e];
            if (v) atomicAdd(&sums[(size_t)k * d + e], (unsigned long long)v);
        } else {
            atomicAdd(&counts[k], 1ull);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// decode (VQ_Full.py:157-169)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cb_to_u8(const double *__restrict__ cb, int64_t n, uint8_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = cast_u8(cb[i]);
}

// One thread per G output bytes of the image (G = 4 when a block row segment is a multiple of 4
// bytes, else 1); consecutive threads write consecutive bytes of an image row.
template <int G>
__global__ __launch_bounds__(256) void k_decode_u8(const int32_t *__restrict__ labels, int nr, int nc,
                                                   const uint8_t *__restrict__ cb8, int bh, int bw, int C,
                                                   uint8_t *__restrict__ img, int64_t total_groups) {
    const int seg = bw * C;
    const int d = bh * seg;
    const int64_t rowbytes = (int64_t)nc * seg;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total_groups;
         g += (int64_t)gridDim.x * blockDim.x) {
        int64_t o = g * G;
        int64_t y = o / rowbytes;
        int64_t xb = o - y * rowbytes;
        int r = (int)(y / bh), i = (int)(y - (int64_t)r * bh);
        int q = (int)(xb / seg), rem = (int)(xb - (int64_t)q * seg);
        int k = labels[(size_t)r * nc + q];
        const uint8_t *src = cb8 + (size_t)k * d + (size_t)i * seg + rem;
        if (G == 4) *reinterpret_cast<uint32_t *>(img + o) = *reinterpret_cast<const uint32_t *>(src);
        else img[o] = src[0];
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

constexpr int r_for(int dp2) { return dp2 <= 8 ? 4 : (dp2 <= 16 ? 4 : (dp2 <= 32 ? 2 : 1)); }

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

template <int DP2, int KEYMODE>
int launch_assign_fast_t(const AssignArgs &a, int block, int grid, size_t lds, hipStream_t st) {
    constexpr int R = r_for(DP2);
    auto kern = k_assign_fast<DP2, R, KEYMODE>;
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, st, a, a.c16, a.bias);
    LAUNCH_CHECK("k_assign_fast");
    return VQ_OK;
}

template <int DP2>
int launch_assign_fast_k(const AssignArgs &a, bool packed, int block, int grid, size_t lds, hipStream_t st) {
    if constexpr (DP2 <= 32) {
        if (packed) return launch_assign_fast_t<DP2, KEY_PACKED>(a, block, grid, lds, st);
    }
    return launch_assign_fast_t<DP2, KEY_SPLIT>(a, block, grid, lds, st);
}

struct DeviceInfo {
    int cus;
    size_t lds_per_block;
};

int device_info(DeviceInfo &di) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    int cus = 0, lds = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    HIP_TRY(hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    di.cus = cus;
    di.lds_per_block = (size_t)lds;
    return VQ_OK;
}

int check_common(const void *blocks, int64_t N, int d, const void *cb, int K) {
    if ((!blocks && N != 0) || !cb) return fail(VQ_E_INVALID, "null blocks or codebook pointer");
    if (N < 0 || d <= 0 || K <= 0) return fail(VQ_E_INVALID, "need N >= 0, d > 0, K > 0 (got N=%lld d=%d K=%d)", (long long)N, d, K);
    if (K > MAX_K) return fail(VQ_E_UNSUPPORTED, "K=%d exceeds %d", K, MAX_K);
    return VQ_OK;
}

// Runs the prepare kernel and the fused fast assignment.  acc_mode ACC_NONE -> encode only.
int run_fast(const uint8_t *blocks, int64_t N, int d, const double *cb, int K, int32_t *labels, double *min_dist,
             unsigned long long *sums, unsigned long long *counts, double *dist_sum, bool accumulate, void *ws,
             hipStream_t st) {
    const int dp2 = dp2_bucket(d);
    WsLayout L = ws_layout(K, d);
    char *w = static_cast<char *>(ws);
    PrepHeader *hdr = reinterpret_cast<PrepHeader *>(w + L.header);
    uint32_t *c16 = reinterpret_cast<uint32_t *>(w + L.c16);
    uint32_t *bias = reinterpret_cast<uint32_t *>(w + L.bias);
    double *cbT = reinterpret_cast<double *>(w + L.cbT);
    double *partials = reinterpret_cast<double *>(w + L.partials);

    hipLaunchKernelGGL(k_prepare, dim3(1), dim3(1024), 0, st, cb, K, d, dp2, hdr, c16, bias, cbT);
    LAUNCH_CHECK("k_prepare");

    DeviceInfo di;
    int rc = device_info(di);
    if (rc) return rc;

    const int kbits = ceil_log2(K);
    // packed keys need (max SAD sum) << kbits to stay below 2^31
    const uint64_t smax = (uint64_t)dp2 * 2ull * 65535ull;
    const bool packed = dp2 <= 32 && ((smax << kbits) < (1ull << 31));

    int acc_mode = ACC_NONE;
    size_t lds = 0;
    int block = 256;
    if (accumulate) {
        size_t tab = (size_t)K * (d + 1) * 4;
        size_t tab_lds = tab < 128 ? 128 : tab;
        if (tab_lds <= 40 * 1024) { acc_mode = ACC_LDS; lds = tab_lds; block = 256; }
        else if (tab_lds <= 80 * 1024) { acc_mode = ACC_LDS; lds = tab_lds; block = 512; }
        else if (tab_lds <= di.lds_per_block) { acc_mode = ACC_LDS; lds = tab_lds; block = 512; }
        else { acc_mode = ACC_GLOBAL; lds = 128; }
    }
    const int R = r_for(dp2);
    const int64_t tile = (int64_t)block * R;
    const int64_t ntiles = (N + tile - 1) / tile;
    int wg_per_cu = block == 256 ? 8 : 4;
    if (lds > 0) {
        int by_lds = (int)((160 * 1024) / lds);
        if (by_lds < 1) by_lds = 1;
        if (by_lds < wg_per_cu) wg_per_cu = by_lds;
    }
    int64_t max_grid = (int64_t)di.cus * wg_per_cu;
    if (max_grid > MAX_GRID) max_grid = MAX_GRID;
    int grid = (int)(ntiles < max_grid ? ntiles : max_grid);
    if (grid < 1) grid = 1;

    AssignArgs a;
    a.blocks = blocks; a.N = N; a.d = d; a.cb = cb; a.K = K; a.kbits = kbits;
    a.hdr = hdr; a.c16 = c16; a.bias = bias; a.cbT = cbT;
    a.labels = labels; a.min_dist = min_dist; a.sums = sums; a.counts = counts; a.partials = partials;
    a.acc_mode = acc_mode; a.ntiles = ntiles;
    a.vec4 = ((d & 3) == 0) && ((reinterpret_cast<uintptr_t>(blocks) & 3) == 0);

    switch (dp2) {
        case 2:  rc = launch_assign_fast_k<2>(a, packed, block, grid, lds, st); break;
        case 4:  rc = launch_assign_fast_k<4>(a, packed, block, grid, lds, st); break;
        case 6:  rc = launch_assign_fast_k<6>(a, packed, block, grid, lds, st); break;
        case 8:  rc = launch_assign_fast_k<8>(a, packed, block, grid, lds, st); break;
        case 12: rc = launch_assign_fast_k<12>(a, packed, block, grid, lds, st); break;
        case 16: rc = launch_assign_fast_k<16>(a, packed, block, grid, lds, st); break;
        case 24: rc = launch_assign_fast_k<24>(a, packed, block, grid, lds, st); break;
        case 32: rc = launch_assign_fast_k<32>(a, packed, block, grid, lds, st); break;
        case 48: rc = launch_assign_fast_k<48>(a, packed, block, grid, lds, st); break;
        case 64: rc = launch_assign_fast_k<64>(a, packed, block, grid, lds, st); break;
        case 96: rc = launch_assign_fast_k<96>(a, packed, block, grid, lds, st); break;
        default: return fail(VQ_E_UNSUPPORTED, "no fast kernel for d=%d", d);
    }
    if (rc) return rc;
    if (accumulate) {
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, st, partials, grid, dist_sum);
        LAUNCH_CHECK("k_sum_partials");
    }
    return VQ_OK;
}

bool fast_supported(int d) { return d <= MAX_D_FAST && dp2_bucket(d) > 0; }

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int vq_abi_version(void) { return ABI_VERSION; }

const char *vq_last_error(void) { return g_err; }

size_t vq_workspace_bytes(int K, int d) {
    if (K <= 0 || d <= 0) return 0;
    return ws_layout(K, d).total;
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
    bool g4 = ((bw * C) % 4 == 0) && ((reinterpret_cast<uintptr_t>(blocks) & 3) == 0);
    int64_t groups = g4 ? total / 4 : total;
    int grid = grid_for(groups, 256, 8192);
    if (g4) hipLaunchKernelGGL(k_blocks_u8<4>, dim3(grid), dim3(256), 0, st, img, H, W, C, bh, bw, pad_mode == VQ_PAD_ZERO, blocks, nc, groups);
    else hipLaunchKernelGGL(k_blocks_u8<1>, dim3(grid), dim3(256), 0, st, img, H, W, C, bh, bw, pad_mode == VQ_PAD_ZERO, blocks, nc, groups);
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
    if (!labels && !min_dist) return fail(VQ_E_INVALID, "both labels and min_dist are null");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (mode == VQ_MODE_FAST && fast_supported(d)) {
        if (!ws || ws_bytes < vq_workspace_bytes(K, d))
            return fail(VQ_E_WORKSPACE, "workspace of %zu bytes needed, %zu given", vq_workspace_bytes(K, d), ws_bytes);
        return run_fast(blocks, N, d, cb, K, labels, min_dist, nullptr, nullptr, nullptr, false, ws, st);
    }
    int grid = grid_for(N, 256, 16384);
    hipLaunchKernelGGL(k_assign_exact, dim3(grid), dim3(256), 0, st, blocks, N, d, cb, K, labels, min_dist,
                       (unsigned long long *)nullptr, (unsigned long long *)nullptr, (double *)nullptr);
    LAUNCH_CHECK("k_assign_exact");
    return VQ_OK;
}

int vq_accum_labels(const uint8_t *blocks, int64_t N, int d, const int32_t *labels, int K, uint64_t *sums,
                    uint64_t *counts, void *stream) {
    if (N < 0 || d <= 0 || K <= 0) return fail(VQ_E_INVALID, "need N >= 0, d > 0, K > 0");
    if (N == 0) return VQ_OK;
    if (!blocks || !labels || !sums || !counts) return fail(VQ_E_INVALID, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int grid = grid_for(N * (int64_t)(d + 1), 256, 16384);
    hipLaunchKernelGGL(k_accum_labels, dim3(grid), dim3(256), 0, st, blocks, N, d, labels, K,
                       reinterpret_cast<unsigned long long *>(sums), reinterpret_cast<unsigned long long *>(counts));
    LAUNCH_CHECK("k_accum_labels");
    return VQ_OK;
}

int vq_lbg_accum(const uint8_t *blocks, int64_t N, int d, const double *cb, int K, uint64_t *sums, uint64_t *counts,
                 double *dist_sum, int32_t *labels, int mode, void *ws, size_t ws_bytes, void *stream) {
    int rc = check_common(blocks, N, d, cb, K);
    if (rc) return rc;
    if (!sums || !counts || !dist_sum) return fail(VQ_E_INVALID, "null sums, counts or dist_sum pointer");
    if (mode != VQ_MODE_EXACT && mode != VQ_MODE_FAST) return fail(VQ_E_INVALID, "unknown mode %d", mode);
    if (!ws || ws_bytes < vq_workspace_bytes(K, d))
        return fail(VQ_E_WORKSPACE, "workspace of %zu bytes needed, %zu given", vq_workspace_bytes(K, d), ws_bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (N == 0) {
        HIP_TRY(hipMemsetAsync(dist_sum, 0, sizeof(double), st));
        return VQ_OK;
    }
    if (mode == VQ_MODE_FAST && fast_supported(d))
        return run_fast(blocks, N, d, cb, K, labels, nullptr, reinterpret_cast<unsigned long long *>(sums),
                        reinterpret_cast<unsigned long long *>(counts), dist_sum, true, ws, st);
    // EXACT mode (also serves dimensions beyond the SAD filter kernels)
    double *partials = reinterpret_cast<double *>(static_cast<char *>(ws) + ws_layout(K, d).partials);
    int grid = grid_for(N, 256, MAX_GRID);
    hipLaunchKernelGGL(k_assign_exact, dim3(grid), dim3(256), 0, st, blocks, N, d, cb, K, labels, (double *)nullptr,
                       reinterpret_cast<unsigned long long *>(sums), reinterpret_cast<unsigned long long *>(counts),
                       partials);
    LAUNCH_CHECK("k_assign_exact");
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, st, partials, grid, dist_sum);
    LAUNCH_CHECK("k_sum_partials");
    return VQ_OK;
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
    if (!ws || ws_bytes < vq_workspace_bytes(K, d))
        return fail(VQ_E_WORKSPACE, "workspace of %zu bytes needed, %zu given", vq_workspace_bytes(K, d), ws_bytes);
    if (nr == 0 || nc == 0) return VQ_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    WsLayout L = ws_layout(K, d);
    uint8_t *cb8 = reinterpret_cast<uint8_t *>(static_cast<char *>(ws) + L.cb_u8);
    int64_t n = (int64_t)K * d;
    hipLaunchKernelGGL(k_cb_to_u8, dim3(grid_for(n, 256, 4096)), dim3(256), 0, st, cb, n, cb8);
    LAUNCH_CHECK("k_cb_to_u8");
    int64_t total = (int64_t)nr * nc * d;
    bool g4 = ((bw * C) % 4 == 0) && ((reinterpret_cast<uintptr_t>(img) & 3) == 0);
    int64_t groups = g4 ? total / 4 : total;
    int grid = grid_for(groups, 256, 8192);
    if (g4) hipLaunchKernelGGL(k_decode_u8<4>, dim3(grid), dim3(256), 0, st, labels, nr, nc, cb8, bh, bw, C, img, groups);
    else hipLaunchKernelGGL(k_decode_u8<1>, dim3(grid), dim3(256), 0, st, labels, nr, nc, cb8, bh, bw, C, img, groups);
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
