// threefry.hpp -- Threefry-2x32-20 counter PRNG with jax.random's (jax 0.4.24,
// jax_threefry_partitionable=False) key plumbing: PRNGKey, split, random_bits, uniform,
// normal, bernoulli.  Needed so that chains are keyed the way the reference keys them
// (`jax.random.PRNGKey(random_state)` at bpl/dixon_coles.py:107, then numpyro's
// random.split calls).  jax itself is not in the reference
// tree; the block function is pinned by the Random123 known-answer vectors and the
// bit layout by jax's published `random_bits(PRNGKey(1701), 32, (3,))` test values
// (tests/test_threefry.py).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace tf {

struct Key {
    uint32_t hi, lo;
};

inline Key prng_key(uint64_t seed) { return {(uint32_t)(seed >> 32), (uint32_t)seed}; }

inline uint32_t rotl(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }

inline void block(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t* o0,
                  uint32_t* o1) {
    static const int R0[4] = {13, 15, 26, 6}, R1[4] = {17, 29, 16, 24};
    const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
    uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
    for (int g = 0; g < 5; ++g) {
        const int* R = (g & 1) ? R1 : R0;
        for (int i = 0; i < 4; ++i) {
            x0 += x1;
            x1 = rotl(x1, R[i]);
            x1 ^= x0;
        }
        x0 += ks[(g + 1) % 3];
        x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
    }
    *o0 = x0;
    *o1 = x1;
}

// jax threefry_2x32(key, count): count (padded to even length) is cut into two halves
// that form the (c0, c1) inputs; outputs are the two halves concatenated.
inline void hash_counts(Key k, const uint32_t* counts, int n, uint32_t* out) {
    const int m = (n + 1) / 2;  // pairs
    std::vector<uint32_t> o0(m), o1(m);
    for (int i = 0; i < m; ++i) {
        const uint32_t c0 = counts[i];
        const uint32_t c1 = (m + i) < n ? counts[m + i] : 0u;  // zero padding when n odd
        block(k.hi, k.lo, c0, c1, &o0[i], &o1[i]);
    }
    for (int i = 0; i < n; ++i) out[i] = i < m ? o0[i] : o1[i - m];
}

// jax.random.bits(key, (n,), uint32)
inline void random_bits(Key k, int n, uint32_t* out) {
    std::vector<uint32_t> counts(n);
    for (int i = 0; i < n; ++i) counts[i] = (uint32_t)i;
    hash_counts(k, counts.data(), n, out);
}

// jax.random.split(key, num)
inline void split(Key k, int num, Key* out) {
    std::vector<uint32_t> flat(2 * (size_t)num);
    random_bits(k, 2 * num, flat.data());
    for (int i = 0; i < num; ++i) out[i] = {flat[2 * i], flat[2 * i + 1]};
}
inline void split2(Key k, Key* a, Key* b) {
    Key o[2];
    split(k, 2, o);
    *a = o[0];
    *b = o[1];
}
inline void split3(Key k, Key* a, Key* b, Key* c) {
    Key o[3];
    split(k, 3, o);
    *a = o[0];
    *b = o[1];
    *c = o[2];
}

// float32 in [0,1): mantissa trick of jax.random.uniform
inline float bits_to_unit_f32(uint32_t b) {
    const uint32_t u = (b >> 9) | 0x3F800000u;
    float f;
    std::memcpy(&f, &u, 4);
    return f - 1.0f;
}

// jax.random.uniform(key, (n,), float32, minval, maxval), widened to double
inline void uniform(Key k, int n, float minval, float maxval, double* out) {
    std::vector<uint32_t> bits(n);
    random_bits(k, n, bits.data());
    for (int i = 0; i < n; ++i) {
        volatile float scaled = bits_to_unit_f32(bits[i]) * (maxval - minval);
        const float v = scaled + minval;
        out[i] = (double)(v > minval ? v : minval);
    }
}

// erfinv: Giles' single-precision polynomial (the form XLA uses for f32), polished by two
// Newton steps on erf() so the result is the float64 inverse of the float32 argument.
inline double erfinv(double x) {
    if (x <= -1.0) return -INFINITY;
    if (x >= 1.0) return INFINITY;
    double w = -std::log((1.0 - x) * (1.0 + x));
    double p;
    if (w < 5.0) {
        w -= 2.5;
        p = 2.81022636e-08;
        p = 3.43273939e-07 + p * w;
        p = -3.5233877e-06 + p * w;
        p = -4.39150654e-06 + p * w;
        p = 0.00021858087 + p * w;
        p = -0.00125372503 + p * w;
        p = -0.00417768164 + p * w;
        p = 0.246640727 + p * w;
        p = 1.50140941 + p * w;
    } else {
        w = std::sqrt(w) - 3.0;
        p = -0.000200214257;
        p = 0.000100950558 + p * w;
        p = 0.00134934322 + p * w;
        p = -0.00367342844 + p * w;
        p = 0.00573950773 + p * w;
        p = -0.0076224613 + p * w;
        p = 0.00943887047 + p * w;
        p = 1.00167406 + p * w;
        p = 2.83297682 + p * w;
    }
    double y = p * x;
    for (int it = 0; it < 2; ++it) {
        const double err = std::erf(y) - x;
        y -= err / (1.1283791670955126 * std::exp(-y * y));
    }
    return y;
}

// jax.random.normal(key, (n,), float32): sqrt(2) * erfinv(uniform(nextafter(-1,0), 1))
inline void normal(Key k, int n, double* out) {
    const float lo = std::nextafterf(-1.0f, 0.0f);
    uniform(k, n, lo, 1.0f, out);
    for (int i = 0; i < n; ++i) out[i] = 1.4142135623730951 * erfinv(out[i]);
}

// jax.random.bernoulli(key, p) for a scalar: uniform(key, ()) < p
inline bool bernoulli(Key k, double p) {
    double u;
    uniform(k, 1, 0.0f, 1.0f, &u);
    return u < p;
}

}  // namespace tf
