// How many waves a SIMD of this GPU holds at once as a function of the scalar registers a kernel declares: every wave
// notes the wave slot the hardware gave it (HW_ID bits 0..3) and then waits until the whole launch is resident or a
// time limit passes.  Prints, per kernel variant, the number of distinct slots seen.
// Build: hipcc -O2 --offload-arch=gfx950 tools/ubench/occupancy_probe.hip -o build/occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int S>
__global__ __launch_bounds__(256) void k_probe(unsigned *out, unsigned long long wait_ticks) {
    if (S == 32) asm volatile("; use s31" ::: "s31");
    if (S == 48) asm volatile("; use s47" ::: "s47");
    if (S == 64) asm volatile("; use s63" ::: "s63");
    if (S == 80) asm volatile("; use s79" ::: "s79");
    if (S == 88) asm volatile("; use s87" ::: "s87");
    if (S == 96) asm volatile("; use s95" ::: "s95");
    if (S == 100) asm volatile("; use s99" ::: "s99");
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait_ticks) __builtin_amdgcn_s_sleep(32);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = hw;
}

template <int S>
int run(unsigned *d_out, int cus) {
    const int wgs = cus * 12, waves = wgs * 4;                 // more than any occupancy can hold at once
    CHECK(hipMemset(d_out, 0xff, waves * sizeof(unsigned)));
    hipLaunchKernelGGL(k_probe<S>, dim3(wgs), dim3(256), 0, 0, d_out, 3000ull);      // 30 us at 100 MHz
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> h(waves);
    CHECK(hipMemcpy(h.data(), d_out, waves * sizeof(unsigned), hipMemcpyDeviceToHost));
    std::set<unsigned> slots;
    for (unsigned v : h) slots.insert(v & 15u);
    hipFuncAttributes fa;
    CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_probe<S>)));
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(k_probe<S>), 256, 0));
    printf("scalar registers forced to %3d: %zu wave slots per SIMD seen (numRegs %d; the runtime predicts %d workgroups of 4 waves per CU)\n",
           S, slots.size(), fa.numRegs, occ);
    return 0;
}

int main() {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    unsigned *d_out = nullptr;
    CHECK(hipMalloc(&d_out, (size_t)cus * 12 * 4 * sizeof(unsigned)));
    int rc = run<0>(d_out, cus) || run<32>(d_out, cus) || run<48>(d_out, cus) || run<64>(d_out, cus) || run<80>(d_out, cus) ||
             run<88>(d_out, cus) || run<96>(d_out, cus) || run<100>(d_out, cus);
    CHECK(hipFree(d_out));
    return rc;
}
