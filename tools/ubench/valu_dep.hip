// Second VALU microbenchmark for gfx950: how the issue rate depends on the
// dependency pattern between consecutive instructions of one wave, and what
// shader clock the chip holds while doing it.
//
// Build: hipcc -O3 --offload-arch=gfx950 valu_dep.hip -o valu_dep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// BODY is an asm string using v accumulators %0..%7, vector operand %8, scalar operand %9.
#define DEFINE_KERNEL(NAME, BODY)                                                          \
__global__ __launch_bounds__(256) void NAME(uint32_t* out, unsigned long long* clk, int iters, uint32_t sc_in) { \
    uint32_t a0, a1, a2, a3, a4, a5, a6, a7;                                                \
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;                                   \
    a0 = tid; a1 = tid + 1; a2 = tid + 2; a3 = tid + 3; a4 = tid + 4; a5 = tid + 5; a6 = tid + 6; a7 = tid + 7; \
    uint32_t x = tid * 2654435761u;                                                         \
    uint32_t sc = __builtin_amdgcn_readfirstlane(sc_in);                                    \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                               \
    for (int i = 0; i < iters; ++i) {                                                       \
        asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                     : "v"(x), "s"(sc));                                                    \
    }                                                                                       \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                               \
    if (tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }                                   \
    uint32_t r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                     \
    if (r == 0x12345678u) out[tid] = r;                                                     \
}

#define S(n) "v_sad_u16 %" #n ", %8, %9, %" #n "\n\t"
#define REP4(x) x x x x
#define REP8(x) x x x x x x x x

// 64 instructions per loop trip in every kernel.
// A: one fully dependent chain
DEFINE_KERNEL(k_chain1, REP8(REP8(S(0))))
// B: two chains, alternating
DEFINE_KERNEL(k_chain2, REP8(REP4(S(0) S(1))))
// C: eight chains, round robin
DEFINE_KERNEL(k_chain8, REP8(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)))
// D: eight chains, each advanced by 8 dependent steps in a row (the shape of a d=16 SAD loop)
DEFINE_KERNEL(k_chain8_runs, REP8(S(0)) REP8(S(1)) REP8(S(2)) REP8(S(3)) REP8(S(4)) REP8(S(5)) REP8(S(6)) REP8(S(7)))
// E: four chains, runs of 2
DEFINE_KERNEL(k_chain4_runs2, REP8(S(0) S(0) S(1) S(1) S(2) S(2) S(3) S(3)))
// F: the candidate inner loop for one codeword and four vectors:
//    8 SAD (dependent) + lshl_add key + med3 + min per vector = 11 instructions, x4 = 44; padded to 64 with 20 more SADs
#define VEC(acc, m1, m2) \
    "v_sad_u16 %" #acc ", %8, %9, 0\n\t" \
    "v_sad_u16 %" #acc ", %8, %9, %" #acc "\n\t" "v_sad_u16 %" #acc ", %8, %9, %" #acc "\n\t" \
    "v_sad_u16 %" #acc ", %8, %9, %" #acc "\n\t" "v_sad_u16 %" #acc ", %8, %9, %" #acc "\n\t" \
    "v_sad_u16 %" #acc ", %8, %9, %" #acc "\n\t" "v_sad_u16 %" #acc ", %8, %9, %" #acc "\n\t" \
    "v_sad_u16 %" #acc ", %8, %9, %" #acc "\n\t" \
    "v_lshl_add_u32 %" #acc ", %" #acc ", 8, %9\n\t" \
    "v_med3_u32 %" #m2 ", %" #acc ", %" #m1 ", %" #m2 "\n\t" \
    "v_min_u32 %" #m1 ", %" #acc ", %" #m1 "\n\t"
DEFINE_KERNEL(k_loop_shape, VEC(0, 1, 2) VEC(3, 4, 5) VEC(0, 6, 7) VEC(3, 1, 2) REP4(S(0) S(3) S(0) S(3) S(0)))
// G: fp32 pair shape, separate chains: sub then add-abs on 8 chains
#define FP(n) "v_sub_f32 %" #n ", %8, %9\n\t" "v_add_f32 %" #n ", %" #n ", |%" #n "|\n\t"
DEFINE_KERNEL(k_fp_pair, REP4(FP(0) FP(1) FP(2) FP(3) FP(4) FP(5) FP(6) FP(7)))
// H: instruction that does not read its destination: v_sub only
#define SUB(n) "v_sub_f32 %" #n ", %8, %9\n\t"
DEFINE_KERNEL(k_sub_only, REP8(SUB(0) SUB(1) SUB(2) SUB(3) SUB(4) SUB(5) SUB(6) SUB(7)))
// I: add reading only its own register
#define ADDSELF(n) "v_add_f32 %" #n ", %" #n ", |%" #n "|\n\t"
DEFINE_KERNEL(k_addself, REP8(ADDSELF(0) ADDSELF(1) ADDSELF(2) ADDSELF(3) ADDSELF(4) ADDSELF(5) ADDSELF(6) ADDSELF(7)))
// J: sad with both sources from the accumulator's own register (one distinct VGPR read)
#define SADSELF(n) "v_sad_u16 %" #n ", %" #n ", %9, %" #n "\n\t"
DEFINE_KERNEL(k_sadself, REP8(SADSELF(0) SADSELF(1) SADSELF(2) SADSELF(3) SADSELF(4) SADSELF(5) SADSELF(6) SADSELF(7)))
// K: v_cndmask with vcc written once before
DEFINE_KERNEL(k_cnd, "v_cmp_lt_u32 vcc, %0, %8\n\t" REP8("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\tv_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc\n\t"))

typedef void (*kern_t)(uint32_t*, unsigned long long*, int, uint32_t);
struct Form { const char* name; kern_t k; };

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    double clk_ghz = p.clockRate / 1e6;
    printf("device %s CUs %d nominal clock %.3f GHz\n", p.gcnArchName, cus, clk_ghz);
    std::vector<Form> forms = {
        {"sad 1 chain", k_chain1}, {"sad 2 chains", k_chain2}, {"sad 8 chains rr", k_chain8},
        {"sad 8 chains runs of 8", k_chain8_runs}, {"sad 4 chains runs of 2", k_chain4_runs2},
        {"loop shape (44+20)", k_loop_shape}, {"fp32 sub+add|.| 8ch", k_fp_pair},
        {"v_sub only", k_sub_only}, {"v_add self", k_addself}, {"sad self", k_sadself},
        {"cmp + 64 cndmask", k_cnd},
    };
    uint32_t* out;
    unsigned long long* clk;
    CHECK(hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 8));
    CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 8192;
    printf("%-26s %6s %10s %16s %12s %10s\n", "form", "w/SIMD", "ms", "winst/nomclk/SIMD", "shader GHz", "cyc/inst");
    for (auto& f : forms) {
        for (int wps : {1, 2, 4, 8}) {
            int grid = cus * wps;
            f.k<<<grid, 256>>>(out, clk, 64, 3);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            unsigned long long h[2] = {0, 0};
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                f.k<<<grid, 256>>>(out, clk, iters, 3);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) { best = ms; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost)); }
            }
            double winst = (double)iters * 64;
            double cycles_nom = best * 1e-3 * clk_ghz * 1e9;
            double shader_ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;   // memrealtime ticks at 100 MHz
            double cyc_per_inst = (double)h[0] / winst / wps;                       // shader cycles per wave-inst per SIMD slot
            printf("%-26s %6d %10.4f %16.4f %12.3f %10.3f\n", f.name, wps, best, winst * wps / cycles_nom, shader_ghz, cyc_per_inst);
        }
    }
    return 0;
}
