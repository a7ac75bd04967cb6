// checks the DPP scan helpers of dc_kernels.hip.h against serial sums
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../bpl-next_amd/csrc/dc_kernels.hip.h"
__global__ void k(const double* in, double* pre, double* suf, double* rowsum) {
    const int lane = threadIdx.x;
    const double v = in[lane];
    pre[lane] = dc::wave_prefix_dpp_f64(v);
    suf[lane] = dc::wave_suffix_dpp_f64(v, lane);
    double r[1] = {v};
    dc::row_sum_f64(r);
    rowsum[lane] = r[0];
}
int main() {
    double h[64], *d, *p, *s, *r, hp[64], hs[64], hr[64];
    for (int i = 0; i < 64; ++i) h[i] = (i * 37 % 11) - 5.0 + 0.25 * i;
    hipMalloc(&d, 512); hipMalloc(&p, 512); hipMalloc(&s, 512); hipMalloc(&r, 512);
    hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, p, s, r);
    hipMemcpy(hp, p, 512, hipMemcpyDeviceToHost); hipMemcpy(hs, s, 512, hipMemcpyDeviceToHost);
    hipMemcpy(hr, r, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        double a = 0, b = 0, c = 0;
        for (int j = 0; j <= i; ++j) a += h[j];
        for (int j = i; j < 64; ++j) b += h[j];
        for (int j = (i & ~15); j < (i & ~15) + 16; ++j) c += h[j];
        if (a != hp[i] || b != hs[i] || c != hr[i]) { if (bad < 8) printf("lane %d prefix %g/%g suffix %g/%g row %g/%g\n", i, hp[i], a, hs[i], b, hr[i], c); ++bad; }
    }
    printf("dpp scans: %d bad lanes\n", bad);
    return bad != 0;
}
