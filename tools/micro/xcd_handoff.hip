// Where do workgroups land, and what does a producer -> consumer hand-off between two workgroups
// cost (a) through agent-scope (sc1) stores / loads / atomics, (b) through the XCD's own L2 when
// both workgroups sit on the SAME XCD (plain write-through store, L2 atomics below agent scope,
// L1 invalidate + plain load)?  Bounded spins everywhere.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ void where(unsigned* xcc, unsigned* cu) {
    if (threadIdx.x == 0) {
        unsigned x, h;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
        xcc[blockIdx.x] = x;
        cu[blockIdx.x] = h;
    }
}

constexpr unsigned LIMIT = 1u << 20;

// ping-pong: producer block writes a payload of 64 doubles (value = round) then raises flag[0] = round;
// consumer waits, checks the payload, raises flag[1] = round.  MODE 0: agent scope, MODE 1: L2 scope.
template <int MODE>
__global__ void pingpong(int prod, int cons, int rounds, double* payload, unsigned* flag, unsigned long long* out) {
    const int b = blockIdx.x;
    if (b != prod && b != cons) return;
    const int lane = threadIdx.x;  // 64 threads
    unsigned long long t0 = 0, bad = 0, timeout = 0;
    if (b == prod) {
        t0 = __builtin_amdgcn_s_memrealtime();
        for (int r = 1; r <= rounds && !timeout; ++r) {
            if (MODE == 0) {
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(payload + lane), (unsigned long long)__double_as_longlong((double)r),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (MODE == 2) {   // the payload is ACCUMULATED: no-return float64 atomics, 8 per lane on 8 lines
                for (int j = 0; j < 8; ++j) atomicAdd(payload + j * 64 + lane, 1.0);
            } else if (MODE == 3) {   // no-return int64 atomics
                for (int j = 0; j < 8; ++j)
                    (void)__hip_atomic_fetch_add(reinterpret_cast<long long*>(payload) + j * 64 + lane, 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                payload[lane] = (double)r;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                if (MODE != 1) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                unsigned spins = 0;
                for (;;) {
                    unsigned v = MODE != 1 ? __hip_atomic_load(flag + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                           : __hip_atomic_fetch_add(flag + 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (v >= (unsigned)r) break;
                    if (++spins > LIMIT) { timeout = 1; break; }
                }
            }
            timeout = __shfl((int)timeout, 0, 64);
        }
        if (lane == 0) {
            out[0] = __builtin_amdgcn_s_memrealtime() - t0;
            out[2] = timeout;
        }
    } else {
        for (int r = 1; r <= rounds && !timeout; ++r) {
            if (lane == 0) {
                unsigned spins = 0;
                for (;;) {
                    unsigned v = MODE != 1 ? __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                           : __hip_atomic_fetch_add(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (v >= (unsigned)r) break;
                    if (++spins > LIMIT) { timeout = 1; break; }
                }
            }
            timeout = __shfl((int)timeout, 0, 64);
            double v;
            if (MODE == 0) {
                v = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(payload + lane),
                                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            } else if (MODE == 2 || MODE == 3) {   // all eight accumulated words must show this round's add
                v = (double)r;
                for (int j = 0; j < 8; ++j) {
                    const unsigned long long w = __hip_atomic_load(reinterpret_cast<unsigned long long*>(payload + j * 64 + lane),
                                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const double got = MODE == 2 ? __longlong_as_double((long long)w) : (double)(long long)w;
                    if (got != (double)r) v = -1.0;
                }
            } else {
                asm volatile("buffer_inv sc1" ::: "memory");   // drop this CU's L1 lines
                v = *reinterpret_cast<volatile double*>(payload + lane);
            }
            if (v != (double)r) ++bad;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                if (MODE != 1) __hip_atomic_fetch_add(flag + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_fetch_add(flag + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        unsigned long long tot = bad;
        for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d, 64);
        if (lane == 0) { out[1] = tot; out[3] = timeout; }
    }
}

int main() {
    const int NB = 256;
    unsigned *dx, *dc;
    hipMalloc(&dx, NB * 4); hipMalloc(&dc, NB * 4);
    hipLaunchKernelGGL(where, dim3(NB), dim3(256), 0, 0, dx, dc);
    std::vector<unsigned> x(NB), c(NB);
    hipMemcpy(x.data(), dx, NB * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, NB * 4, hipMemcpyDeviceToHost);
    int rr = 0;
    for (int i = 0; i < NB; ++i) rr += (x[i] & 0xF) == (unsigned)(i % 8);
    printf("XCC_ID == blockIdx %% 8 for %d of %d workgroups; first 16:", rr, NB);
    for (int i = 0; i < 16; ++i) printf(" %u", x[i] & 0xF);
    printf("\n");
    double* payload; unsigned* flag; unsigned long long* out;
    hipMalloc(&payload, 8192); hipMalloc(&flag, 4096); hipMalloc(&out, 64);
    const int rounds = 20000;
    const int pairs[3][2] = {{0, 8}, {0, 1}, {0, 16}};
    const char* names[4] = {"agent scope (sc1) stores ", "L2 scope (same XCD only) ", "float64 no-return atomics", "int64 no-return atomics  "};
    for (int mode = 0; mode < 4; ++mode)
        for (auto& pr : pairs) {
            hipMemset(payload, 0, 8192); hipMemset(flag, 0, 4096); hipMemset(out, 0, 64);
            if (mode == 0) hipLaunchKernelGGL(pingpong<0>, dim3(64), dim3(64), 0, 0, pr[0], pr[1], rounds, payload, flag, out);
            else if (mode == 2) hipLaunchKernelGGL(pingpong<2>, dim3(64), dim3(64), 0, 0, pr[0], pr[1], rounds, payload, flag, out);
            else if (mode == 3) hipLaunchKernelGGL(pingpong<3>, dim3(64), dim3(64), 0, 0, pr[0], pr[1], rounds, payload, flag, out);
            else hipLaunchKernelGGL(pingpong<1>, dim3(64), dim3(64), 0, 0, pr[0], pr[1], rounds, payload, flag, out);
            hipDeviceSynchronize();
            unsigned long long h[4];
            hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
            printf("%s  blocks %d <-> %d (XCD %u, %u): round trip %.3f us, wrong payload reads %llu, timeouts %llu %llu\n",
                   names[mode], pr[0], pr[1], x[pr[0]] & 0xF, x[pr[1]] & 0xF,
                   h[0] * 0.01 / rounds, h[1], h[2], h[3]);
        }
    return 0;
}
