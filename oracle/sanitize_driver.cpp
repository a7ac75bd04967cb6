// TEST INFRASTRUCTURE (NOT product code): the CPU-side sanitizer run of SURVEY.md section 5 row 2.
// Built by `make -C oracle asan` with -fsanitize=address,undefined together with nuts_harness.cpp
// (which compiles the PRODUCT's header-only NUTS driver and threefry, bpl-next_amd/csrc/nuts.hpp and
// threefry.hpp -- the one place product C++ can be sanitised: GPU AddressSanitizer is not available
// on the pool), the C oracle and the CPU port, and run on seeded inputs that reach every code path
// the CPU tests reach: both models, covariates, weights, ragged pairs, warm-up with adaptation, a
// fixed step size, thinning, the threefry helpers, the CPU port with several threads.
// Exit code 0 and no sanitizer report = pass (tests/test_sanitizers.py).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" {
int dco_latent_dim(int model, int T, int K);
int dco_potential_grad(int model, int64_t n, int T, int K, const uint16_t* h, const uint16_t* a,
                       const uint8_t* x, const uint8_t* y, const double* w, const double* xs,
                       const double* z, double* U, double* grad, double* aux, int nthreads);
int harness_nuts_dc(int model, int64_t n, int T, int K, const uint16_t* h, const uint16_t* a,
                    const uint8_t* x, const uint8_t* y, const double* w, const double* xs, int warm,
                    int samp, int depth, int thin, const double* z0, uint32_t khi, uint32_t klo,
                    double* draws, double* stats, double* summary, double step_size);
int harness_nuts_gauss(int D, const double* sd, int warm, int samp, int depth, int thin,
                       const double* z0, uint32_t khi, uint32_t klo, double* draws, double* stats,
                       double* summary, double step_size);
int harness_schedule(int num_steps, int* out, int cap);
void harness_ckpt_idxs(int n, int* idx_min, int* idx_max);
void harness_threefry_block(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t* out);
void harness_normal(uint32_t khi, uint32_t klo, int n, double* out);
void harness_uniform(uint32_t khi, uint32_t klo, int n, float lo, float hi, double* out);
struct port_t;
port_t* dcp_create(int model, int64_t n, int T, int K, const uint16_t* h, const uint16_t* a,
                   const uint8_t* x, const uint8_t* y, const float* w, const double* xs, int nthreads);
void dcp_destroy(port_t* p);
int dcp_latent_dim(const port_t* p);
int dcp_eval(port_t* p, const double* z, double* U_out, double* grad, double* aux);
double dcp_eval_many(port_t* p, const double* zs, int n_z, int count, double* U_last, double* grad_last);
}

namespace {
uint64_t state = 0x9E3779B97F4A7C15ull;
uint32_t rnd() {  // splitmix64: seeded, no libc state
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 16);
}
double unif(double lo, double hi) { return lo + (hi - lo) * (rnd() / 4294967296.0); }
int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("CHECK failed: %s (line %d)\n", #c, __LINE__); ++fails; } } while (0)

void league(int n, int T, int K, bool weighted, int model) {
    std::vector<uint16_t> h(n), a(n);
    std::vector<uint8_t> x(n), y(n);
    std::vector<double> w(n), xs((size_t)T * K);
    std::vector<float> wf(n);
    for (int i = 0; i < n; ++i) {
        h[i] = (uint16_t)(rnd() % T);
        a[i] = (uint16_t)((h[i] + 1 + rnd() % (T - 1)) % T);
        x[i] = (uint8_t)(rnd() % 5);
        y[i] = (uint8_t)(rnd() % 4);
        wf[i] = (float)unif(0.05, 1.0);
        w[i] = wf[i];
    }
    if (n > 3) { x[3] = 255; y[3] = 255; }  // extreme scorelines are legal
    for (auto& v : xs) v = unif(-1.5, 1.5);
    const int Kk = model == 1 ? K : 0;
    const int D = dco_latent_dim(model, T, Kk);
    std::vector<double> z(D), g(D), g2(D), aux(8), aux2(8);
    for (auto& v : z) v = unif(-0.5, 0.5);
    double U = 0, U2 = 0;
    const double* wp = weighted && model == 1 ? w.data() : nullptr;
    CHECK(dco_potential_grad(model, n, T, Kk, h.data(), a.data(), x.data(), y.data(), wp, Kk ? xs.data() : nullptr,
                             z.data(), &U, g.data(), aux.data(), 3) == 0);
    CHECK(std::isfinite(U));
    // the CPU port (pair-sorted, float32 tables) against it, 1 and 3 threads
    for (int nt : {1, 3}) {
        port_t* p = dcp_create(model, n, T, Kk, h.data(), a.data(), x.data(), y.data(),
                               weighted && model == 1 ? wf.data() : nullptr, Kk ? xs.data() : nullptr, nt);
        CHECK(p != nullptr);
        if (!p) continue;
        CHECK(dcp_latent_dim(p) == D);
        CHECK(dcp_eval(p, z.data(), &U2, g2.data(), aux2.data()) == 0);
        CHECK(std::fabs(U2 - U) <= 1e-5 * std::fabs(U) + 1e-6);
        std::vector<double> zs(4 * (size_t)D);
        for (auto& v : zs) v = unif(-0.3, 0.3);
        (void)dcp_eval_many(p, zs.data(), 4, 9, &U2, g2.data());
        CHECK(std::isfinite(U2));
        dcp_destroy(p);
    }
    // NUTS through the product's driver: adaptation on, then a fixed step with thinning and a start point
    const int warm = 40, samp = 30;
    std::vector<double> draws((size_t)samp * D), stats((size_t)samp * 4), summ(4 + D);
    CHECK(harness_nuts_dc(model, n, T, Kk, h.data(), a.data(), x.data(), y.data(), wp, Kk ? xs.data() : nullptr, warm,
                          samp, 6, 1, nullptr, 0, 42, draws.data(), stats.data(), summ.data(), 1.0) == 0);
    for (double v : draws) CHECK(std::isfinite(v));
    std::vector<double> z0(D, 0.05);
    CHECK(harness_nuts_dc(model, n, T, Kk, h.data(), a.data(), x.data(), y.data(), wp, Kk ? xs.data() : nullptr, 0, 12,
                          4, 3, z0.data(), 7, 9, draws.data(), stats.data(), summ.data(), 0.01) == 0);
}
}  // namespace

int main() {
    league(380, 20, 0, false, 0);
    league(777, 37, 0, false, 0);
    league(380, 20, 3, false, 1);
    league(600, 12, 2, true, 1);
    league(1, 2, 0, false, 0);
    {   // sampler invariants on a Gaussian, dimension 1 and 50, depth limit 1 and 10
        for (int D : {1, 50}) {
            std::vector<double> sd(D), draws((size_t)60 * D), stats(60 * 4), summ(4 + D);
            for (int i = 0; i < D; ++i) sd[i] = 0.5 + 0.1 * i;
            for (int depth : {1, 10})
                CHECK(harness_nuts_gauss(D, sd.data(), 80, 60, depth, 1, nullptr, 1, 2, draws.data(), stats.data(),
                                         summ.data(), 1.0) == 0);
        }
    }
    {   // helpers
        int win[64];
        for (int n : {0, 1, 19, 20, 150, 500, 1000}) CHECK(harness_schedule(n, win, 32) >= 0);
        int lo, hi;
        for (int n = 0; n < 1024; ++n) { harness_ckpt_idxs(n, &lo, &hi); CHECK(hi >= lo - 1 && hi < 11); }
        uint32_t out[2];
        harness_threefry_block(0, 0, 0, 0, out);
        CHECK(out[0] == 0x6b200159u && out[1] == 0x99ba4efeu);  // Random123 known answer
        std::vector<double> v(1001);
        harness_normal(0, 42, 1001, v.data());
        harness_uniform(3, 4, 1001, -2.f, 2.f, v.data());
        for (double u : v) CHECK(u >= -2.0 && u < 2.0);
    }
    std::printf(fails ? "sanitize_driver: %d checks failed\n" : "sanitize_driver: ok\n", fails);
    return fails ? 1 : 0;
}
