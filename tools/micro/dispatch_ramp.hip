// Diagnostic micro-benchmark (not product code): how long after the first workgroup of a launch
// does the last one start executing, as a function of grid size, block size and LDS per block?
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/dispatch_ramp tools/micro/dispatch_ramp.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__global__ void k(unsigned long long* out, int spin) {
    extern __shared__ char smem[];
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t;
    // stay resident for a while so that late blocks do not reuse a finished block's CU
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 1 && smem[0] == 77) out[blockIdx.x] = 0;
}
int main() {
    unsigned long long* d;
    hipMalloc(&d, 4096 * 8);
    std::vector<unsigned long long> h(4096);
    const int grids[] = {62, 125, 249, 256, 512};
    const int blocks[] = {64, 256, 512, 1024};
    const int ldss[] = {0, 4096, 65536};
    for (int lds : ldss)
        for (int b : blocks)
            for (int g : grids) {
                if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                double med[9];
                for (int rep = 0; rep < 9; ++rep) {
                    hipLaunchKernelGGL(k, dim3(g), dim3(b), lds, 0, d, 40);
                    hipDeviceSynchronize();
                    hipMemcpy(h.data(), d, g * 8, hipMemcpyDeviceToHost);
                    unsigned long long lo = ~0ull, hi = 0;
                    for (int i = 0; i < g; ++i) { lo = std::min(lo, h[i]); hi = std::max(hi, h[i]); }
                    med[rep] = (hi - lo) * 0.01;
                }
                std::sort(med, med + 9);
                std::vector<unsigned long long> s(h.begin(), h.begin() + g);
                std::sort(s.begin(), s.end());
                printf("lds %6d block %4d grid %4d: last-first %.2f us (median of 9; min %.2f), half of the blocks by %.2f us\n",
                       lds, b, g, med[4], med[0], (s[g / 2] - s[0]) * 0.01);
            }
    // start time by blockIdx % 8 (= XCD, round robin): is the ramp a skew between XCDs?
    for (int g : {125, 249}) {
        double acc[8] = {0}, cnt[8] = {0};
        for (int rep = 0; rep < 50; ++rep) {
            hipLaunchKernelGGL(k, dim3(g), dim3(512), 4096, 0, d, 40);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, g * 8, hipMemcpyDeviceToHost);
            unsigned long long lo = ~0ull;
            for (int i = 0; i < g; ++i) lo = std::min(lo, h[i]);
            for (int i = 0; i < g; ++i) { acc[i % 8] += (h[i] - lo) * 0.01; cnt[i % 8] += 1; }
        }
        printf("grid %d, mean start by blockIdx %% 8:", g);
        for (int x = 0; x < 8; ++x) printf(" %.2f", acc[x] / cnt[x]);
        printf(" us\n");
        double byq[4] = {0}, cq[4] = {0};
        for (int rep = 0; rep < 50; ++rep) {
            hipLaunchKernelGGL(k, dim3(g), dim3(512), 4096, 0, d, 40);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, g * 8, hipMemcpyDeviceToHost);
            unsigned long long lo = ~0ull;
            for (int i = 0; i < g; ++i) lo = std::min(lo, h[i]);
            for (int i = 0; i < g; ++i) { byq[i * 4 / g] += (h[i] - lo) * 0.01; cq[i * 4 / g] += 1; }
        }
        printf("grid %d, mean start by quarter of the block index:", g);
        for (int x = 0; x < 4; ++x) printf(" %.2f", byq[x] / cq[x]);
        printf(" us\n");
    }
    return 0;
}
