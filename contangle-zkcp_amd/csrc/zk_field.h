// Montgomery prime-field arithmetic on 32-bit limbs for gfx950 (and the host).
//
// Memory layout is bit-identical to ark-ff 0.3 `Fp256/Fp384` and pasta_curves 0.4
// `Fp/Fq` (little-endian u64 limbs, Montgomery form, R = 2^(64*limbs64)): 8 (or 12)
// little-endian u32 words are the same bytes.  The reference reaches these types only
// through `Groth16::<Bls12_381>::prove` (lib/src/zk/verifiable_encryption.rs:92 and
// siblings); the arithmetic itself is upstream.
//
// CDNA4 has no 64x64 multiplier: products are built from 32x32->64 MADs
// (`v_mad_u64_u32`), so the natural limb is 32 bits.  All loops are fully unrolled
// over compile-time limb counts so that elements live in VGPRs and modulus words fold
// to literals (for the Pasta primes five of the eight words are 0 or 1 and vanish).
#pragma once
#include <stdint.h>

#include "zk_params.h"
#include "zk_mul_asm.h"  // gfx950 device path of fe_mul (generated)

#if defined(__HIPCC__)
#define ZK_HD __host__ __device__ inline __attribute__((always_inline))
#define ZK_UNROLL _Pragma("unroll")
#else
#define ZK_HD inline __attribute__((always_inline))
#define ZK_UNROLL
#endif

namespace zk {

template <class P>
struct alignas(16) Fe {
    uint32_t v[P::N];
};

template <class P>
ZK_HD void fe_zero(Fe<P>& r) {
    ZK_UNROLL
    for (int i = 0; i < P::N; i++) r.v[i] = 0;
}
template <class P>
ZK_HD void fe_one(Fe<P>& r) {
    ZK_UNROLL
    for (int i = 0; i < P::N; i++) r.v[i] = P::R[i];
}
template <class P>
ZK_HD bool fe_is_zero(const Fe<P>& a) {
    uint32_t o = 0;
    ZK_UNROLL
    for (int i = 0; i < P::N; i++) o |= a.v[i];
    return o == 0;
}
template <class P>
ZK_HD bool fe_eq(const Fe<P>& a, const Fe<P>& b) {
    uint32_t o = 0;
    ZK_UNROLL
    for (int i = 0; i < P::N; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// t -= p if t >= p  (t < 2p on entry)
template <class P>
ZK_HD void fe_reduce_once(uint32_t* t) {
    uint32_t d[P::N];
    uint64_t br = 0;
    ZK_UNROLL
    for (int i = 0; i < P::N; i++) {
        uint64_t x = (uint64_t)t[i] - P::P[i] - br;
        d[i] = (uint32_t)x;
        br = (x >> 32) & 1;
    }
    ZK_UNROLL
    for (int i = 0; i < P::N; i++) t[i] = br ? t[i] : d[i];
}

template <class P>
ZK_HD void fe_add(Fe<P>& r, const Fe<P>& a, const Fe<P>& b) {
    uint32_t t[P::N];
    uint64_t c = 0;
    ZK_UNROLL
    for (int i = 0; i < P::N; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        t[i] = (uint32_t)c;
        c >>= 32;
    }
    // every supported modulus leaves a spare top bit: a + b < 2p < 2^(32N), no carry out
    fe_reduce_once<P>(t);
    ZK_UNROLL
    for (int i = 0; i < P::N; i++) r.v[i] = t[i];
}

template <class P>
ZK_HD void fe_sub(Fe<P>& r, const Fe<P>& a, const Fe<P>& b) {
    uint32_t t[P::N];
    uint64_t br = 0;
    ZK_UNROLL
    for (int i = 0; i < P::N; i++) {
        uint64_t x = (uint64_t)a.v[i] - b.v[i] - br;
        t[i] = (uint32_t)x;
        br = (x >> 32) & 1;
    }
    const uint32_t mask = (uint32_t)0 - (uint32_t)br;
    uint64_t c = 0;
    ZK_UNROLL
    for (int i = 0; i < P::N; i++) {
        c += (uint64_t)t[i] + (P::P[i] & mask);
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
}

template <class P>
ZK_HD void fe_neg(Fe<P>& r, const Fe<P>& a) {
    Fe<P> z;
    fe_zero(z);
    fe_sub(r, z, a);
}
template <class P>
ZK_HD void fe_dbl(Fe<P>& r, const Fe<P>& a) {
    fe_add(r, a, a);
}

// Montgomery product r = a*b*R^-1 mod p (CIOS).  Every supported modulus has its top
// bit clear, so the running value stays below 2p < 2^(32N) and the (N+2)-th word of
// textbook CIOS is never needed.
template <class P>
ZK_HD void fe_mul_portable(Fe<P>& r, const Fe<P>& a, const Fe<P>& b) {
    constexpr int N = P::N;
    uint32_t t[N];
    ZK_UNROLL
    for (int i = 0; i < N; i++) t[i] = 0;
    ZK_UNROLL
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
        const uint32_t bi = b.v[i];
        ZK_UNROLL
        for (int j = 0; j < N; j++) {
            uint64_t x = (uint64_t)a.v[j] * bi + t[j] + c;
            t[j] = (uint32_t)x;
            c = x >> 32;
        }
        const uint32_t tn = (uint32_t)c;
        const uint32_t m = t[0] * P::INV;
        uint64_t x = (uint64_t)m * P::P[0] + t[0];
        c = x >> 32;
        ZK_UNROLL
        for (int j = 1; j < N; j++) {
            x = (uint64_t)m * P::P[j] + t[j] + c;
            t[j - 1] = (uint32_t)x;
            c = x >> 32;
        }
        t[N - 1] = (uint32_t)((uint64_t)tn + c);
    }
    fe_reduce_once<P>(t);
    ZK_UNROLL
    for (int i = 0; i < N; i++) r.v[i] = t[i];
}

template <class P>
ZK_HD void fe_mul(Fe<P>& r, const Fe<P>& a, const Fe<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    // gfx950: hand-scheduled product scanning, 2 VALU instructions per partial product (zk_mul_asm.h)
    uint32_t t[P::N];
    fe_mul_asm<P>(t, a.v, b.v);
    fe_reduce_once<P>(t);
    ZK_UNROLL
    for (int i = 0; i < P::N; i++) r.v[i] = t[i];
#else
    fe_mul_portable(r, a, b);
#endif
}

template <class P>
ZK_HD void fe_sqr(Fe<P>& r, const Fe<P>& a) {
    fe_mul(r, a, a);
}

// canonical <-> Montgomery (ark-ff from_repr / into_repr; pasta from_repr / to_repr)
template <class P>
ZK_HD void fe_to_mont(Fe<P>& r, const Fe<P>& a) {
    Fe<P> r2;
    ZK_UNROLL
    for (int i = 0; i < P::N; i++) r2.v[i] = P::R2[i];
    fe_mul(r, a, r2);
}
template <class P>
ZK_HD void fe_from_mont(Fe<P>& r, const Fe<P>& a) {
    Fe<P> one;
    fe_zero(one);
    one.v[0] = 1;
    fe_mul(r, a, one);
}

// r = a^e for a 64-bit exponent (square-and-multiply, LSB first)
template <class P>
ZK_HD void fe_pow_u64(Fe<P>& r, const Fe<P>& a, uint64_t e) {
    Fe<P> acc, base = a;
    fe_one(acc);
    while (e) {
        if (e & 1) fe_mul(acc, acc, base);
        fe_sqr(base, base);
        e >>= 1;
    }
    r = acc;
}

// Fermat inverse a^(p-2) (host-side use: one-off normalisations); 0 -> 0
template <class P>
ZK_HD void fe_inv(Fe<P>& r, const Fe<P>& a) {
    uint32_t e[P::N];
    uint64_t br = 2;  // e = p - 2 with borrow propagation (the Pasta primes have low word 1)
    for (int i = 0; i < P::N; i++) {
        uint64_t x = (uint64_t)P::P[i] - br;
        e[i] = (uint32_t)x;
        br = (x >> 32) & 1;
    }
    Fe<P> acc, base = a;
    fe_one(acc);
    for (int i = 0; i < 32 * P::N; i++) {
        if ((e[i / 32] >> (i % 32)) & 1) fe_mul(acc, acc, base);
        fe_sqr(base, base);
    }
    r = acc;
}

// ------------------------------------------------------------------------------------------
// Fq2 = Fq[u] / (u^2 + 1): the coordinate field of the G2 twists of BN254 and BLS12-381
// (ark-ff 0.3 Fp2 with NONRESIDUE = -1).  Same free-function names as Fe<P>, so the curve
// formulas in zk_curve.h are written once for both.
// ------------------------------------------------------------------------------------------
template <class P>
struct Fe2 {
    Fe<P> c0, c1;
};
// Base-field product as a real call (operands and result by value, i.e. in VGPRs): an Fq2 point addition
// holds 30+ base multiplications; inlining every one of them makes the G2 kernels enormous (the BLS12-381
// G2 code object took ten minutes to compile and spilled) for no gain -- the multiply is ~600 instructions.
#if defined(__HIPCC__)
#define ZK_HD_CALL __host__ __device__ __attribute__((noinline))
#else
#define ZK_HD_CALL __attribute__((noinline))
#endif
// (Round 1 kept this callee on the portable product for the 12-limb field after a wrong result was attributed to the
// generated inline-asm product running inside a non-inlined function whose operands arrive through the stack.  That
// attribution did not hold up: tools/asm_callee_check.hip runs exactly that configuration -- and the 8-limb fields -- with all
// lanes active, under a static divergent EXEC mask and inside a data-dependent loop, 3 x 65536 chained products per field,
// and every result equals the portable product and the host's (profiles/r02_asm_callee_check.txt); the ISA of the callee
// shows the 2 wait states between every v_mad_u64_u32 carry-out and its v_addc_co_u32 reader.  The asm product is used here
// again; the BLS12-381 G2 parity tests on saturated limbs (zk_msm_opts.limb_bits = 32) cover it.)
template <class P>
ZK_HD_CALL Fe<P> fe_mul_call(Fe<P> a, Fe<P> b) {
    Fe<P> r;
    fe_mul(r, a, b);
    return r;
}
template <class P>
ZK_HD void fe_zero(Fe2<P>& r) {
    fe_zero(r.c0);
    fe_zero(r.c1);
}
template <class P>
ZK_HD void fe_one(Fe2<P>& r) {
    fe_one(r.c0);
    fe_zero(r.c1);
}
template <class P>
ZK_HD bool fe_is_zero(const Fe2<P>& a) {
    return fe_is_zero(a.c0) && fe_is_zero(a.c1);
}
template <class P>
ZK_HD bool fe_eq(const Fe2<P>& a, const Fe2<P>& b) {
    return fe_eq(a.c0, b.c0) && fe_eq(a.c1, b.c1);
}
template <class P>
ZK_HD void fe_add(Fe2<P>& r, const Fe2<P>& a, const Fe2<P>& b) {
    fe_add(r.c0, a.c0, b.c0);
    fe_add(r.c1, a.c1, b.c1);
}
template <class P>
ZK_HD void fe_sub(Fe2<P>& r, const Fe2<P>& a, const Fe2<P>& b) {
    fe_sub(r.c0, a.c0, b.c0);
    fe_sub(r.c1, a.c1, b.c1);
}
template <class P>
ZK_HD void fe_neg(Fe2<P>& r, const Fe2<P>& a) {
    fe_neg(r.c0, a.c0);
    fe_neg(r.c1, a.c1);
}
template <class P>
ZK_HD void fe_dbl(Fe2<P>& r, const Fe2<P>& a) {
    fe_dbl(r.c0, a.c0);
    fe_dbl(r.c1, a.c1);
}
// Karatsuba: 3 base multiplications
template <class P>
ZK_HD void fe_mul(Fe2<P>& r, const Fe2<P>& a, const Fe2<P>& b) {
    Fe<P> s, t;
    const Fe<P> v0 = fe_mul_call(a.c0, b.c0);
    const Fe<P> v1 = fe_mul_call(a.c1, b.c1);
    fe_add(s, a.c0, a.c1);
    fe_add(t, b.c0, b.c1);
    s = fe_mul_call(s, t);
    fe_sub(s, s, v0);
    fe_sub(r.c1, s, v1);
    fe_sub(r.c0, v0, v1);
}
// (a0 + a1 u)^2 = (a0 + a1)(a0 - a1) + 2 a0 a1 u: 2 base multiplications
template <class P>
ZK_HD void fe_sqr(Fe2<P>& r, const Fe2<P>& a) {
    Fe<P> s, d;
    fe_add(s, a.c0, a.c1);
    fe_sub(d, a.c0, a.c1);
    const Fe<P> m = fe_mul_call(a.c0, a.c1);
    r.c0 = fe_mul_call(s, d);
    fe_dbl(r.c1, m);
}
// 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2)
template <class P>
ZK_HD void fe_inv(Fe2<P>& r, const Fe2<P>& a) {
    Fe<P> n, t;
    fe_sqr(n, a.c0);
    fe_sqr(t, a.c1);
    fe_add(n, n, t);
    fe_inv(n, n);
    fe_mul(r.c0, a.c0, n);
    fe_neg(t, a.c1);
    fe_mul(r.c1, t, n);
}
// r = sel ? a : r  (branch-free limb select)
template <class P>
ZK_HD void fe_cmov(Fe<P>& r, const Fe<P>& a, bool sel) {
    ZK_UNROLL
    for (int i = 0; i < P::N; i++) r.v[i] = sel ? a.v[i] : r.v[i];
}
template <class P>
ZK_HD void fe_cmov(Fe2<P>& r, const Fe2<P>& a, bool sel) {
    fe_cmov(r.c0, a.c0, sel);
    fe_cmov(r.c1, a.c1, sel);
}
// load from little-endian u32 words (c0 words then c1 words for Fe2)
template <class P>
ZK_HD void fe_from_words(Fe<P>& r, const uint32_t* w) {
    ZK_UNROLL
    for (int i = 0; i < P::N; i++) r.v[i] = w[i];
}
template <class P>
ZK_HD void fe_from_words(Fe2<P>& r, const uint32_t* w) {
    fe_from_words(r.c0, w);
    fe_from_words(r.c1, w + P::N);
}

}  // namespace zk
