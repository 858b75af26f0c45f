// The NTT pass on lazy 9 x 29-bit limbs (zk_field29.h): the same mixed-radix plan, tiles, addressing and fused coset /
// scaling / zero-extension steps as ntt_pass_kernel (zk_ntt_kernels.h), with the butterflies of a tile done in the
// unsaturated form the MSM buckets use.  The pass kernel is VALU-issue-bound (rocprofv3: 93 % of the issue slots at
// 2^23), so the currency is instructions per butterfly:
//     saturated 32-bit limbs   product ~290 (every MAD drags a carry add), add / sub ~25-30 each (carry chains + conditional -p)
//     lazy 29-bit limbs        product ~205 (one MAD per partial product), add 9, sub 18, a carry step (~27) every third stage
// Representation inside a tile.  A tile element is an integer V = x (mod p) with V < VB p, in 9 limbs of which the lower
// eight are <= LB.  HBM holds plain 8 x 32-bit words as before: elements enter as V < 2p (the caller's canonical values,
// or the < 2p values an earlier pass stored) and leave canonical from the last pass.  The data never changes Montgomery
// domain: twiddles, coset powers and the scale factor are kept as t R' mod p (R' = 2^261), so mont29(V, t R') = V t.
// Decimation in TIME inside the tile (bit-reversed placement on load, natural order on store) because its sum path grows
// additively:
//     t = mont29(w, tw)                     strict limbs, < 2p        (needs LB(w) < 2^31.6, VB(w) <= 128)
//     o0 = u + t                            LB + 2^29,   VB + 2
//     o1 = u - t + 4p  (limb-wise bias)     LB + 2^29.72, VB + 4
// so VB <= 2 + 4 * 10 for the deepest tile (2^10 points), and three stages of limb growth fit a u32 (2^29 + 3 * 2^29.72
// < 2^32; the multiplied operand has seen at most two: 2^31.1 < 2^31.6): one parallel carry step (fe29_norm) after every
// third stage.  Decimation in frequency would double VB on the sum path at every stage.
#pragma once
#include "zk_ntt_kernels.h"
#include "zk_field29.h"

namespace zk {

// x <- x * g^i with the power tables in R' form (packed words): two lazy products
template <class F>
__device__ __forceinline__ void mul_pow29(Fe29<F>& x, const PowTables<F>& t, uint64_t i) {
    Fe<F> a = t.lo[i & 1023], b = t.hi[i >> 10];
    Fe29<F> la, lb;
    fe29_unpack(la, a);
    fe29_unpack(lb, b);
    fe29_mul(x, x, la);
    fe29_mul(x, x, lb);
}

// table entries are t R' mod p for the R-form entries t R mod p: R' / R = 2^5
template <class F>
__global__ void __launch_bounds__(256) table_to_r29_kernel(Fe<F>* __restrict__ tbl, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fe<F> x = tbl[i];
    constexpr int SH = F29<F>::W * F29<F>::L - 32 * F::N;
    for (int k = 0; k < SH; k++) fe_dbl(x, x);
    tbl[i] = x;
}

// The butterflies inside a tile only ever need the 2^(IL-1) powers omega^(m n / 2^IL), IL = min(log n, 10) (stage lg uses
// m = k << (IL - 1 - lg)): kept as a compact table of UNPACKED limbs (12 words = 48 B per entry, 24 KB: cache-resident)
// behind the n/2-entry table, which only the inter-pass twiddles read.  Saves the 18 shift / mask pairs of an unpack in
// every butterfly and turns a 128 MB-strided gather into reads of one small table.
constexpr int NTT_INNER_LOG = 10;
struct alignas(16) InnerTw {
    uint32_t v[12];
};
template <class F>
__global__ void __launch_bounds__(256) inner_table_kernel(const Fe<F>* __restrict__ tw, InnerTw* __restrict__ out, uint32_t count, int shift) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= count) return;
    const Fe<F> raw = tw[(uint64_t)m << shift];
    Fe29<F> t;
    fe29_unpack(t, raw);
    InnerTw o;
    for (int l = 0; l < 12; l++) o.v[l] = l < F29<F>::L ? t.v[l] : 0u;
    out[m] = o;
}

template <class F>
__global__ void __launch_bounds__(1024) ntt_pass29_kernel(const Fe<F>* __restrict__ in, Fe<F>* __restrict__ out, const Fe<F>* __restrict__ tw,
                                                          const InnerTw* __restrict__ inner, NttPass A, Fe<F> scale, PowTables<F> pre,
                                                          PowTables<F> post) {
    ZK_DYN_SHARED(uint32_t, lds);
    using K = F29<F>;
    constexpr int NL = K::L;
    const uint32_t R = 1u << A.log_r, T = 1u << A.log_t, RT = R * T;
    const int log_np = A.logn - A.log_m;
    const int log_s = log_np - A.log_r;
    const uint64_t half = (A.logn > 0) ? (1ull << (A.logn - 1)) : 1ull;
    const uint32_t tid = threadIdx.x, nth = blockDim.x;

    uint64_t base, stride_j, stride_t;
    uint64_t nprime0 = 0, out_fixed = 0;
    if (!A.last) {
        const uint64_t tiles_per_a = (1ull << log_s) >> A.log_t;
        const uint64_t a = blockIdx.x / tiles_per_a;
        nprime0 = (blockIdx.x % tiles_per_a) << A.log_t;
        base = (a << log_np) + nprime0;
        stride_j = 1ull << log_s;
        stride_t = 1;
    } else if (A.nd == 1) {
        base = 0;
        stride_j = 1;
        stride_t = 0;
    } else {
        const int log_rest = A.log_m - A.rd[0];
        const uint64_t rest = blockIdx.x & ((1ull << log_rest) - 1);
        const uint64_t k1_0 = ((uint64_t)blockIdx.x >> log_rest) << A.log_t;
        base = ((k1_0 << log_rest) + rest) << A.log_r;
        stride_j = 1;
        stride_t = 1ull << (log_rest + A.log_r);
        uint64_t rr = rest, acc = 0;
        int logm = A.log_m;
        for (int p = A.nd - 2; p >= 1; p--) {
            logm -= A.rd[p];
            acc += (rr & ((1ull << A.rd[p]) - 1)) << logm;
            rr >>= A.rd[p];
        }
        out_fixed = k1_0 + acc;
    }
    auto pos = [&](uint32_t j, uint32_t t) -> uint32_t { return A.last ? (t << A.log_r) + j : (j << A.log_t) + t; };

    // Zero-extended input (halo2 coeff_to_extended): only radix-axis indices j < J = 2^(in_log - log_s) are non-zero.  In
    // bit-reversed placement they land on the slots that are multiples of 2^z (z = log_r - log2 J), and the first z
    // stages pair each of them with zeros only: after those stages every slot of an aligned group of 2^z holds the
    // group's one value (u + 0 and u - 0 + 4p are the same residue).  So the value is written to its whole group on load
    // and the butterflies start at stage z: 3 of the 23 stages of a 2^20 -> 2^23 extension never run.
    int z = 0;
    if (A.in_log > 0 && A.log_m == 0 && A.in_log >= log_s && A.in_log - log_s < A.log_r) z = A.log_r - (A.in_log - log_s);
    // ---- load the tile: radix-axis index j goes to slot bitrev(j) (decimation in time)
    for (uint32_t e = tid; e < RT; e += nth) {
        uint32_t j, t;
        if (A.last) {
            j = e & (R - 1);
            t = e >> A.log_r;
        } else {
            t = e & (T - 1);
            j = e >> A.log_t;
        }
        const uint64_t gi = base + j * stride_j + t * stride_t;
        Fe29<F> x;
        const bool padded = A.in_log > 0 && (gi >> A.in_log) != 0;
        if (padded) {
            if (z > 0) continue;                      // its slot is filled by the group's one non-zero element
            fe29_zero(x);
        } else {
            const Fe<F> raw = in[gi];
            fe29_unpack(x, raw);                      // strict limbs, V < 2p
            if (A.pre) mul_pow29(x, pre, gi);
        }
        const uint32_t slot = bitrev32(j, A.log_r);
        for (uint32_t c = 0; c < (1u << z); c++) {
            const uint32_t p = pos(slot + c, t);
            ZK_UNROLL
            for (int l = 0; l < NL; l++) lds[l * RT + p] = x.v[l];
        }
    }
    __syncthreads();

    // ---- radix-2 DIT butterflies over the R axis
    const uint32_t nbf = RT >> 1;
    for (int lg = z; lg < A.log_r; lg++) {
        const uint32_t g = 1u << lg;
        const bool carry = (lg % 3) == 2;
        for (uint32_t b = tid; b < nbf; b += nth) {
            uint32_t q, t;
            if (A.last) {
                q = b & ((R >> 1) - 1);
                t = b >> (A.log_r - 1);
            } else {
                t = b & (T - 1);
                q = b >> A.log_t;
            }
            const uint32_t j = ((q >> lg) << (lg + 1)) | (q & (g - 1));
            const uint32_t p0 = pos(j, t), p1 = pos(j + g, t);
            Fe29<F> u, w, s, d;
            ZK_UNROLL
            for (int l = 0; l < NL; l++) {
                u.v[l] = lds[l * RT + p0];
                w.v[l] = lds[l * RT + p1];
            }
            if (lg > 0) {
                const int il = A.logn < NTT_INNER_LOG ? A.logn : NTT_INNER_LOG;
                const InnerTw traw = inner[(q & (g - 1)) << (il - 1 - lg)];
                Fe29<F> tv;
                ZK_UNROLL
                for (int l = 0; l < NL; l++) tv.v[l] = traw.v[l];
                fe29_mul(w, w, tv);                   // strict, < 2p
            }
            fe29_add(s, u, w);
            fe29_sub(d, u, w, K::BIAS4K1);
            if (carry) {
                fe29_norm(s, s);
                fe29_norm(d, d);
            }
            ZK_UNROLL
            for (int l = 0; l < NL; l++) {
                lds[l * RT + p0] = s.v[l];
                lds[l * RT + p1] = d.v[l];
            }
        }
        __syncthreads();
    }

    // ---- store (natural order along the radix axis): one lazy product brings every element back under 2p with strict
    // limbs -- the inter-pass twiddle (omega^0 R' for the elements that need none), or the scale factor / R' itself on the
    // last pass, which then subtracts p once and is canonical
    Fe29<F> sc;
    fe29_unpack(sc, scale);
    for (uint32_t e = tid; e < RT; e += nth) {
        const uint32_t t = e & (T - 1);
        const uint32_t k = e >> A.log_t;
        const uint32_t p = pos(k, t);
        Fe29<F> x;
        ZK_UNROLL
        for (int l = 0; l < NL; l++) x.v[l] = lds[l * RT + p];
        Fe<F> r;
        if (!A.last) {
            const uint64_t ex = ((uint64_t)k * (nprime0 + t)) << A.log_m;
            Fe<F> wraw;
            if (ex >= half) {
                wraw = tw[ex - half];
                fe_neg(wraw, wraw);                   // omega^(n/2) = -1
            } else {
                wraw = tw[ex];
            }
            Fe29<F> wv;
            fe29_unpack(wv, wraw);
            fe29_mul(x, x, wv);
            fe29_pack(r, x);                          // < 2p < 2^256: the next pass unpacks it as it is
            out[base + (uint64_t)k * stride_j + t] = r;
        } else {
            const uint64_t go = out_fixed + t + ((uint64_t)k << A.log_m);
            fe29_mul(x, x, sc);
            if (A.post) mul_pow29(x, post, go);
            fe29_pack(r, x);
            fe_reduce_once<F>(r.v);
            out[ntt_out_index(A, go)] = r;
        }
    }
}

}  // namespace zk
