// Micro-benchmark behind dc_predict.hip.h's grid kernel: what one "step" of the inner loop
// (two pmf operands by DPP row broadcast + fma + v_exp_f32, then one v_mfma_f32_16x16x4_f32) costs on
// a SIMD, pipe by pipe, as a function of the waves resident per SIMD.  Prints cycles per step per SIMD.
//   hipcc -O3 --offload-arch=gfx950 grid_pipes.hip -o grid_pipes && ./grid_pipes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int G> __device__ __forceinline__ float pmf_exponent(float E, float Q, float f, float c) {
    float t;
    asm("v_sub_f32_dpp %0, %1, %2 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %3, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
        : "=&v"(t) : "v"(Q), "v"(c), "v"(E), "v"(f), "n"(G));
    return t;
}
// MODE 0: full step | 1: no MFMA (operands summed) | 2: MFMA only | 3: DPP + MFMA, no exp | 4: exp only
// | 5: full step, plain fma instead of DPP
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    float Eh = in[lane], Qh = in[64 + lane], Ea = in[128 + lane], Qa = in[192 + lane];
    const float fx = (float)(lane & 15), cx = in[256 + (lane & 15)];
    f32x4 acc = {0, 0, 0, 0};
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_nop 1" : "+v"(Eh), "+v"(Qh), "+v"(Ea), "+v"(Qa));
#define STEP(g)                                                                                   \
    {                                                                                             \
        float ta, tb;                                                                             \
        if (MODE == 5) { ta = fmaf(Eh, fx, Qh - cx); tb = fmaf(Ea, fx, Qa - cx); Eh += 1e-9f; Ea += 1e-9f; } \
        else if (MODE == 2) { ta = Eh; tb = Ea; }                                                 \
        else if (MODE == 4) { ta = Eh + s; tb = Ea + s; }                                         \
        else { ta = pmf_exponent<g>(Eh, Qh, fx, cx); tb = pmf_exponent<g>(Ea, Qa, fx, cx); }      \
        float a = ta, b = tb;                                                                     \
        if (MODE == 0 || MODE == 1 || MODE == 4 || MODE == 5) { a = __builtin_amdgcn_exp2f(ta); b = __builtin_amdgcn_exp2f(tb); } \
        if (MODE == 1 || MODE == 4) s += a + b;                                                   \
        else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);                      \
    }
        STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
        STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + s;
}
template <int MODE> static void run(const char* name, const float* in, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps : {1, 2, 4, 5, 8}) {   // waves per SIMD: 256 CUs x 4 SIMDs x wps waves
        const int blocks = 256 * wps;   // 4 waves per block -> one per SIMD
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double steps_per_simd = (double)iters * 16 * wps;
        printf("%-34s waves/SIMD %d: %8.1f ns per step per SIMD  (%.1f cycles at 2.4 GHz)\n", name, wps,
               ms * 1e6 / steps_per_simd, ms * 1e6 / steps_per_simd * 2.4);
    }
}
int main() {
    float *in, *out;
    hipMalloc(&in, 4096); hipMalloc(&out, 256 * 8 * 256 * 4);
    std::vector<float> h(1024, 0.001f);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    run<0>("full step (dpp+fma+exp x2, mfma)", in, out);
    run<1>("no mfma (dpp+fma+exp x2, adds)", in, out);
    run<2>("mfma only", in, out);
    run<3>("dpp+fma x2, mfma, no exp", in, out);
    run<4>("exp x2 + adds only", in, out);
    run<5>("full step, plain fma (no dpp)", in, out);
    return 0;
}
