// Host-side (x86-64) field arithmetic on 64-bit limbs for the MSM's serial tail: the Horner
// combination of the per-window sums is ~250 dependent point doublings, far too serial for the GPU
// (one lane needs ~10^4 cycles per doubling) and 2-3x slower on 32-bit limbs than with the CPU's
// 64x64->128 multiplier.  `H64<P>` re-reads a field's constants as u64 limbs; Fe<H64<P>> has the same
// bytes as Fe<P> (little-endian), so points cross between the two views with memcpy, and the curve
// formulas of zk_curve.h are reused unchanged through overloads.
#pragma once
#include <string.h>

#include "zk_curve.h"

namespace zk {

template <class Q>
struct H64 {
    static constexpr int N = Q::N / 2;  // u64 limbs
    static constexpr int BITS = Q::BITS;
    static constexpr uint64_t INV = Q::INV64;
    static constexpr const uint64_t (&P)[Q::N / 2] = Q::P64;
    static constexpr const uint64_t (&R)[Q::N / 2] = Q::R64;
    static constexpr const uint64_t (&R2)[Q::N / 2] = Q::R2_64;
    using Base = Q;
};

template <class Q>
struct alignas(16) Fe<H64<Q>> {
    uint64_t v[Q::N / 2];
};

template <class Q>
inline void fe_zero(Fe<H64<Q>>& r) {
    for (int i = 0; i < Q::N / 2; i++) r.v[i] = 0;
}
template <class Q>
inline void fe_one(Fe<H64<Q>>& r) {
    for (int i = 0; i < Q::N / 2; i++) r.v[i] = Q::R64[i];
}
template <class Q>
inline bool fe_is_zero(const Fe<H64<Q>>& a) {
    uint64_t o = 0;
    for (int i = 0; i < Q::N / 2; i++) o |= a.v[i];
    return o == 0;
}
template <class Q>
inline bool fe_eq(const Fe<H64<Q>>& a, const Fe<H64<Q>>& b) {
    uint64_t o = 0;
    for (int i = 0; i < Q::N / 2; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
template <class Q>
inline void h64_reduce_once(uint64_t* t) {
    constexpr int N = Q::N / 2;
    uint64_t d[N];
    unsigned __int128 br = 0;
    for (int i = 0; i < N; i++) {
        unsigned __int128 x = (unsigned __int128)t[i] - Q::P64[i] - (uint64_t)br;
        d[i] = (uint64_t)x;
        br = (x >> 64) & 1;
    }
    if (!br)
        for (int i = 0; i < N; i++) t[i] = d[i];
}
template <class Q>
inline void fe_add(Fe<H64<Q>>& r, const Fe<H64<Q>>& a, const Fe<H64<Q>>& b) {
    constexpr int N = Q::N / 2;
    uint64_t t[N];
    unsigned __int128 c = 0;
    for (int i = 0; i < N; i++) {
        c += (unsigned __int128)a.v[i] + b.v[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    h64_reduce_once<Q>(t);  // a + b < 2p < 2^(64N): no carry out of the top limb
    for (int i = 0; i < N; i++) r.v[i] = t[i];
}
template <class Q>
inline void fe_sub(Fe<H64<Q>>& r, const Fe<H64<Q>>& a, const Fe<H64<Q>>& b) {
    constexpr int N = Q::N / 2;
    uint64_t t[N];
    unsigned __int128 br = 0;
    for (int i = 0; i < N; i++) {
        unsigned __int128 x = (unsigned __int128)a.v[i] - b.v[i] - (uint64_t)br;
        t[i] = (uint64_t)x;
        br = (x >> 64) & 1;
    }
    if (br) {
        unsigned __int128 c = 0;
        for (int i = 0; i < N; i++) {
            c += (unsigned __int128)t[i] + Q::P64[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    for (int i = 0; i < N; i++) r.v[i] = t[i];
}
// CIOS Montgomery product on 64-bit limbs (top bit of every modulus is clear: no extra carry word)
template <class Q>
inline void fe_mul(Fe<H64<Q>>& r, const Fe<H64<Q>>& a, const Fe<H64<Q>>& b) {
    constexpr int N = Q::N / 2;
    uint64_t t[N + 1];
    for (int i = 0; i <= N; i++) t[i] = 0;
    for (int i = 0; i < N; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < N; j++) {
            c += (unsigned __int128)a.v[j] * b.v[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        const uint64_t tn = t[N] + (uint64_t)c;
        const uint64_t m = t[0] * Q::INV64;
        c = (unsigned __int128)m * Q::P64[0] + t[0];
        c >>= 64;
        for (int j = 1; j < N; j++) {
            c += (unsigned __int128)m * Q::P64[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += tn;
        t[N - 1] = (uint64_t)c;
        t[N] = (uint64_t)(c >> 64);
    }
    h64_reduce_once<Q>(t);
    for (int i = 0; i < N; i++) r.v[i] = t[i];
}
template <class Q>
inline void fe_inv(Fe<H64<Q>>& r, const Fe<H64<Q>>& a) {
    Fe<Q> x;
    memcpy(&x, &a, sizeof x);
    fe_inv(x, x);
    memcpy(&r, &x, sizeof x);
}

// the curve seen through 64-bit limbs
template <class C>
struct Host64Curve {
    using Fq = H64<typename C::Fq>;
    using Fr = typename C::Fr;
    static constexpr int EXT = C::EXT;
};
template <class C>
using HostXYZZ = XYZZ<Host64Curve<C>>;

template <class C>
inline void to_host(HostXYZZ<C>& r, const XYZZ<C>& p) {
    static_assert(sizeof(HostXYZZ<C>) == sizeof(XYZZ<C>), "same bytes");
    memcpy(&r, &p, sizeof r);
}
template <class C>
inline void from_host(XYZZ<C>& r, const HostXYZZ<C>& p) {
    memcpy(&r, &p, sizeof r);
}

}  // namespace zk
