// VALU issue-rate microbenchmark for gfx950.
//
// Measures how many wave-instructions per cycle per SIMD the chip sustains for
// the candidate inner-loop instructions of the L1 nearest-codeword search:
// integer SAD forms, fp32 / packed fp32 add and max, fp64 add, and the
// min/med3/shift-add bookkeeping instructions. Each kernel runs 8 independent
// dependency chains per lane so that latency does not bound the result.
//
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
// Run:   ./valu_rate            (prints one line per instruction form)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int CHAINS = 8;
constexpr int UNROLL = 8;   // instructions per chain per loop trip

// One instruction form = one asm template operating on accumulator %0,
// vector operand %1 and scalar operand %2.
#define DEFINE_KERNEL(NAME, ASMSTR, ACC_T, ACC_CONSTRAINT)                               \
__global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters, uint32_t sc_in) {  \
    ACC_T acc[CHAINS];                                                                    \
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;                                 \
    for (int c = 0; c < CHAINS; ++c) acc[c] = (ACC_T)(tid * 7u + c);                      \
    ACC_T x = (ACC_T)(tid * 2654435761u);                                                 \
    uint32_t sc = __builtin_amdgcn_readfirstlane(sc_in);                                  \
    for (int i = 0; i < iters; ++i) {                                                     \
        _Pragma("unroll")                                                                 \
        for (int u = 0; u < UNROLL; ++u) {                                                \
            _Pragma("unroll")                                                             \
            for (int c = 0; c < CHAINS; ++c) {                                            \
                asm volatile(ASMSTR : "+" ACC_CONSTRAINT(acc[c]) : "v"(x), "s"(sc));      \
            }                                                                             \
        }                                                                                 \
    }                                                                                     \
    ACC_T r = 0;                                                                          \
    for (int c = 0; c < CHAINS; ++c) r += acc[c];                                         \
    if (r == (ACC_T)0x12345678) out[tid] = (uint32_t)r;                                   \
}

DEFINE_KERNEL(k_sad_u16_s,   "v_sad_u16 %0, %1, %2, %0",        uint32_t, "v")
DEFINE_KERNEL(k_sad_u16_v,   "v_sad_u16 %0, %1, %1, %0",        uint32_t, "v")
DEFINE_KERNEL(k_sad_u8_s,    "v_sad_u8 %0, %1, %2, %0",         uint32_t, "v")
DEFINE_KERNEL(k_sad_u32_s,   "v_sad_u32 %0, %1, %2, %0",        uint32_t, "v")
DEFINE_KERNEL(k_add_f32_s,   "v_add_f32 %0, %0, %2",            uint32_t, "v")
DEFINE_KERNEL(k_subabs_f32,  "v_sub_f32 %0, %1, %2\n\tv_add_f32 %0, %0, |%0|", uint32_t, "v")
DEFINE_KERNEL(k_min_u32,     "v_min_u32 %0, %0, %1",            uint32_t, "v")
DEFINE_KERNEL(k_med3_u32,    "v_med3_u32 %0, %0, %1, %2",       uint32_t, "v")
DEFINE_KERNEL(k_lshl_add,    "v_lshl_add_u32 %0, %0, 8, %2",    uint32_t, "v")
DEFINE_KERNEL(k_dot2_u16,    "v_dot2_u32_u16 %0, %1, %2, %0",   uint32_t, "v")
DEFINE_KERNEL(k_pk_max_u16,  "v_pk_max_u16 %0, %0, %2",         uint32_t, "v")
DEFINE_KERNEL(k_pk_add_f32,  "v_pk_add_f32 %0, %0, %1",         uint64_t, "v")
DEFINE_KERNEL(k_pk_fma_f32,  "v_pk_fma_f32 %0, %1, %1, %0",     uint64_t, "v")
DEFINE_KERNEL(k_fma_f32,     "v_fma_f32 %0, %1, %2, %0",        uint32_t, "v")
DEFINE_KERNEL(k_add_f64,     "v_add_f64 %0, %0, |%1|",          uint64_t, "v")
DEFINE_KERNEL(k_fma_f64,     "v_fma_f64 %0, %1, %1, %0",        uint64_t, "v")
DEFINE_KERNEL(k_msad_u8,     "v_msad_u8 %0, %1, %2, %0",        uint32_t, "v")
DEFINE_KERNEL(k_cndmask,     "v_cndmask_b32 %0, %0, %1, vcc",   uint32_t, "v")
DEFINE_KERNEL(k_cmp_lt,      "v_cmp_lt_u32 vcc, %0, %1",        uint32_t, "v")

typedef void (*kern_t)(uint32_t*, int, uint32_t);
struct Form { const char* name; kern_t k; int insts_per_slot; int elems_per_inst; };

int main(int argc, char** argv) {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, dev));
    int cus = p.multiProcessorCount;
    double clk_ghz = p.clockRate / 1e6;
    printf("device %s  CUs %d  clock %.3f GHz  LDS/block %zu\n", p.gcnArchName, cus, clk_ghz, p.sharedMemPerBlock);

    std::vector<Form> forms = {
        {"v_sad_u16 (sgpr)", k_sad_u16_s, 1, 2},
        {"v_sad_u16 (vgpr)", k_sad_u16_v, 1, 2},
        {"v_sad_u8 (sgpr)", k_sad_u8_s, 1, 4},
        {"v_msad_u8 (sgpr)", k_msad_u8, 1, 4},
        {"v_sad_u32 (sgpr)", k_sad_u32_s, 1, 1},
        {"v_add_f32 (sgpr)", k_add_f32_s, 1, 1},
        {"v_sub+v_add|.| f32", k_subabs_f32, 2, 1},
        {"v_fma_f32", k_fma_f32, 1, 1},
        {"v_pk_add_f32", k_pk_add_f32, 1, 2},
        {"v_pk_fma_f32", k_pk_fma_f32, 1, 2},
        {"v_add_f64 |.|", k_add_f64, 1, 1},
        {"v_fma_f64", k_fma_f64, 1, 1},
        {"v_min_u32", k_min_u32, 1, 1},
        {"v_med3_u32", k_med3_u32, 1, 1},
        {"v_lshl_add_u32", k_lshl_add, 1, 1},
        {"v_dot2_u32_u16", k_dot2_u16, 1, 2},
        {"v_pk_max_u16", k_pk_max_u16, 1, 2},
        {"v_cndmask_b32", k_cndmask, 1, 1},
        {"v_cmp_lt_u32", k_cmp_lt, 1, 1},
    };

    uint32_t* out;
    CHECK(hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 4096;
    int wps_list[] = {1, 2, 4, 8};   // waves per SIMD (= 256-thread blocks per CU)
    printf("%-22s %5s %12s %14s %14s\n", "form", "w/SIMD", "ms", "winst/clk/SIMD", "Tlane-inst/s");
    for (auto& f : forms) {
        for (int wps : wps_list) {
            int grid = cus * wps;
            f.k<<<grid, 256>>>(out, 64, 3);   // warm
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                f.k<<<grid, 256>>>(out, iters, 3);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            double winst = (double)iters * UNROLL * CHAINS * f.insts_per_slot;   // per wave
            double waves_per_simd = wps;
            double cycles = best * 1e-3 * clk_ghz * 1e9;
            double ipc_simd = winst * waves_per_simd / cycles;
            double lane_rate = winst * 64.0 * (double)grid * 4.0 / (best * 1e-3) / 1e12;
            printf("%-22s %5d %12.4f %14.4f %14.3f\n", f.name, wps, best, ipc_simd, lane_rate);
        }
    }
    CHECK(hipFree(out));
    return 0;
}
