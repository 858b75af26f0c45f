// HIP kernels for the MSM / NTT hot path (gfx950).  Replaces, behind the C ABI of
// include/zkcp_amd.h, the upstream CPU routines the reference reaches through
// `Groth16::<Bls12_381>::prove` (lib/src/zk/verifiable_encryption.rs:92, encryption.rs:76,
// sample_entries.rs:86, property.rs:133):
//   ark-ec 0.3   msm/variable_base.rs   VariableBaseMSM::multi_scalar_mul      
//   ark-poly 0.3 domain/radix2/fft.rs   Radix2EvaluationDomain::*fft_in_place  
//   halo2_proofs 0.2 arithmetic.rs      best_multiexp / best_fft               
// Design notes (data layout, roofline per kernel) are in DESIGN.md.
// This header: the NTT / pointwise kernels (scalar fields); the MSM kernels are in zk_msm_kernels.h.
#pragma once
#include "zk_rt.h"
// (zk_rt.h first: it brings in the HIP runtime or the test emulator)
#include "zk_curve.h"

namespace zk {

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bitrev32(uint32_t x, int bits) {
    return bits == 0 ? 0u : (__brev(x) >> (32 - bits));
}

// acc = prod_{k : bit k of e set} tbl[k]   (tbl[k] = base^(2^k), Montgomery)
template <class F>
__device__ __forceinline__ void pow_from_table(Fe<F>& acc, const Fe<F>* __restrict__ tbl, uint64_t e, int nbits) {
    fe_one(acc);
    for (int k = 0; k < nbits; k++) {
        if ((e >> k) & 1) {
            Fe<F> t = tbl[k];
            fe_mul(acc, acc, t);
        }
    }
}

// out[i] = base^i, i < count   (twiddle table: base = omega, count = n/2)
template <class F>
__global__ void pow_table_kernel(Fe<F>* __restrict__ out, const Fe<F>* __restrict__ tbl, uint64_t count, int nbits) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fe<F> acc;
    pow_from_table(acc, tbl, i, nbits);
    out[i] = acc;
}

// g^i on the fly from two small tables (1024 + n/1024 entries, L2 resident): lo[i & 1023] * hi[i >> 10]
template <class F>
struct PowTables {
    const Fe<F>* lo;   // g^j,          j < min(n, 1024)
    const Fe<F>* hi;   // (g^1024)^j,   j < max(1, n / 1024)
};
template <class F>
__device__ __forceinline__ void mul_pow(Fe<F>& x, const PowTables<F>& t, uint64_t i) {
    Fe<F> a = t.lo[i & 1023], b = t.hi[i >> 10];
    fe_mul(x, x, a);
    fe_mul(x, x, b);
}

// a[i] *= g^i   (ark-poly 0.3 Radix2EvaluationDomain::distribute_powers; halo2 coset shift by ZETA powers)
template <class F>
__global__ void __launch_bounds__(256) coset_mul_kernel(Fe<F>* __restrict__ a, PowTables<F> t, uint64_t count) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        Fe<F> x = a[i];
        mul_pow(x, t, i);
        a[i] = x;
    }
}

// Pointwise vector kernels: the glue of ark-groth16 0.3 r1cs_to_qap.rs `witness_map` between its seven NTTs
//  and ark-ff `into_repr` batches (a7).
enum VecOp : int {
    VEC_MUL = 0,        // a[i] *= b[i]          mul_polynomials_in_evaluation_domain
    VEC_SUB = 1,        // a[i] -= b[i]          ab -= c
    VEC_ADD = 2,        // a[i] += b[i]
    VEC_SCALE = 3,      // a[i] *= s             divide_by_vanishing_poly_on_coset_in_place (s = 1/Z_H(g))
    VEC_FROM_MONT = 4,  // a[i] = into_repr(a[i])
    VEC_TO_MONT = 5,    // a[i] = from_repr(a[i])
    VEC_QAP = 6         // a[i] = (a[i]*b[i] - c[i]) * s     the three steps above fused (one pass over HBM)
};
template <class F>
__global__ void __launch_bounds__(256) vec_op_kernel(Fe<F>* __restrict__ a, const Fe<F>* __restrict__ b, const Fe<F>* __restrict__ c,
                                                     uint64_t n, int op, Fe<F> s) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Fe<F> x = a[i];
        if (op == VEC_MUL) {
            Fe<F> y = b[i];
            fe_mul(x, x, y);
        } else if (op == VEC_SUB) {
            Fe<F> y = b[i];
            fe_sub(x, x, y);
        } else if (op == VEC_ADD) {
            Fe<F> y = b[i];
            fe_add(x, x, y);
        } else if (op == VEC_SCALE) {
            fe_mul(x, x, s);
        } else if (op == VEC_FROM_MONT) {
            fe_from_mont(x, x);
        } else if (op == VEC_TO_MONT) {
            fe_to_mont(x, x);
        } else {
            Fe<F> y = b[i], z = c[i];
            fe_mul(x, x, y);
            fe_sub(x, x, z);
            fe_mul(x, x, s);
        }
        a[i] = x;
    }
}

// a[i] *= t[i mod m], m a power of two <= 16: halo2_proofs 0.2 EvaluationDomain::divide_by_vanishing_poly -- on the extended
// coset X^n - 1 takes only 2^(extended_k - k) distinct values, `t_evaluations` holds their inverses (poly/domain.rs)
template <class F>
struct PeriodicTable {
    Fe<F> v[16];
};
template <class F>
__global__ void __launch_bounds__(256) scale_periodic_kernel(Fe<F>* __restrict__ a, uint64_t n, PeriodicTable<F> t, uint32_t m) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Fe<F> x = a[i], w;
        const uint32_t j = (uint32_t)i & (m - 1);
        ZK_UNROLL
        for (int l = 0; l < F::N; l++) {
            uint32_t v = 0;
            ZK_UNROLL
            for (uint32_t q = 0; q < 16; q++) v = (j == q) ? t.v[q].v[l] : v;   // the table is a kernel argument (SGPRs): select, do not index
            w.v[l] = v;
        }
        fe_mul(x, x, w);
        a[i] = x;
    }
}

// ------------------------------------------------------------------------------------------
// NTT: mixed-radix decimation-in-frequency, one kernel per pass, natural order in and out.
//
// n = 2^L is split into P digits of widths r_1..r_P.  Input index n = (n_1 | n_2 | .. | n_P)
// (n_1 most significant); after pass p the buffer holds, at position (k_1..k_p | n_{p+1}..n_P),
//   Y_p = DFT over digit p of Y_{p-1}, times the inter-pass twiddle omega^(M_p * k_p * n'),
// with M_p = 2^(r_1+..+r_{p-1}) and n' the value of the remaining digits.  The last pass stores
// to the digit-reversed (= natural) address k = k_1 + M_2 k_2 + .. + M_P k_P.
//
// One workgroup owns an R x T tile (R = 2^r_p points of the radix axis, T adjacent columns so
// that every global access is a run of T*32 B); the R-point DFTs run as radix-2 butterflies in
// LDS (limb-major "SoA" so unit-stride lanes hit distinct banks), twiddles come from one table
// tw[i] = omega^i, i < n/2, shared by the inner butterflies (stride n/R) and the inter-pass step.
// ------------------------------------------------------------------------------------------
struct NttPass {
    int logn;   // L
    int log_m;  // log2 M_p
    int log_r;  // r_p
    int log_t;  // log2 T
    int last;   // final pass (S_p == 1): digit-reversed store, optional scaling
    int scale;  // multiply outputs by `scale` (n^-1 for inverse transforms)
    int nd;     // P
    int rd[4];  // r_1..r_P
    int pre;    // first pass multiplies input element i by g_pre^i on load   (ark coset_fft = distribute_powers ; fft)
    int post;   // last pass multiplies output element k by g_post^k on store (ark coset_ifft = ifft ; distribute_powers)
    int in_log; // first pass: only the first 2^in_log input elements are read, the rest count as zero (halo2 coeff_to_extended); 0 = all
    int out_parts_log;  // last pass: result k is stored at (k mod P) * (n / P) + k / P, P = 2^out_parts_log (ZK_NTT_OUT_SUBCOSETS): the
                        // transform's P sub-cosets one after another; 0 = natural order
};
// where result `go` of the last pass lands
__device__ __forceinline__ uint64_t ntt_out_index(const NttPass& A, uint64_t go) {
    if (A.out_parts_log == 0) return go;
    return ((go & ((1ull << A.out_parts_log) - 1)) << (A.logn - A.out_parts_log)) | (go >> A.out_parts_log);
}

template <class F>
__device__ __forceinline__ void tw_get(Fe<F>& w, const Fe<F>* __restrict__ tw, uint64_t e, uint64_t half) {
    // omega^e for e < n, from the half table: omega^(n/2) = -1
    if (e >= half) {
        Fe<F> t = tw[e - half];
        fe_neg(w, t);
    } else {
        w = tw[e];
    }
}

template <class F>
__global__ void __launch_bounds__(1024) ntt_pass_kernel(const Fe<F>* __restrict__ in, Fe<F>* __restrict__ out, const Fe<F>* __restrict__ tw,
                                NttPass A, Fe<F> scale, PowTables<F> pre, PowTables<F> post) {
    ZK_DYN_SHARED(uint32_t, lds);
    constexpr int NL = F::N;
    const uint32_t R = 1u << A.log_r, T = 1u << A.log_t, RT = R * T;
    const int log_np = A.logn - A.log_m;
    const int log_s = log_np - A.log_r;
    const uint64_t half = (A.logn > 0) ? (1ull << (A.logn - 1)) : 1ull;
    const uint32_t tid = threadIdx.x, nth = blockDim.x;

    uint64_t base, stride_j, stride_t;
    uint64_t nprime0 = 0, out_fixed = 0;  // last pass: k_1 base + reversed middle digits
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
        // reverse the middle digits k_2..k_{P-1} of `rest` into their natural weights M_p
        uint64_t rr = rest, acc = 0;
        int logm = A.log_m;
        for (int p = A.nd - 2; p >= 1; p--) {
            logm -= A.rd[p];
            acc += (rr & ((1ull << A.rd[p]) - 1)) << logm;
            rr >>= A.rd[p];
        }
        out_fixed = k1_0 + acc;
    }
    // LDS position of tile element (j, t): the fast axis follows the contiguous global axis
    auto pos = [&](uint32_t j, uint32_t t) -> uint32_t { return A.last ? (t << A.log_r) + j : (j << A.log_t) + t; };

    // ---- load tile
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
        Fe<F> x;
        if (A.in_log > 0 && (gi >> A.in_log) != 0) {
            fe_zero(x);   // zero padding is implied, never stored or read
        } else {
            x = in[gi];
            if (A.pre) mul_pow(x, pre, gi);
        }
        const uint32_t p = pos(j, t);
        ZK_UNROLL
        for (int l = 0; l < NL; l++) lds[l * RT + p] = x.v[l];
    }
    __syncthreads();

    // ---- radix-2 DIF butterflies over the R axis (output in bit-reversed LDS order)
    const uint32_t nbf = RT >> 1;
    for (int lg = A.log_r - 1; lg >= 0; lg--) {
        const uint32_t g = 1u << lg;
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
            Fe<F> u, w, s, d;
            ZK_UNROLL
            for (int l = 0; l < NL; l++) {
                u.v[l] = lds[l * RT + p0];
                w.v[l] = lds[l * RT + p1];
            }
            fe_add(s, u, w);
            fe_sub(d, u, w);
            if (lg > 0) {
                const uint64_t e = (uint64_t)(q & (g - 1)) << (A.logn - 1 - lg);
                Fe<F> twv = tw[e];
                fe_mul(d, d, twv);
            }
            ZK_UNROLL
            for (int l = 0; l < NL; l++) {
                lds[l * RT + p0] = s.v[l];
                lds[l * RT + p1] = d.v[l];
            }
        }
        __syncthreads();
    }

    // ---- store: inter-pass twiddle (non-last) or digit-reversed natural address (last)
    for (uint32_t e = tid; e < RT; e += nth) {
        const uint32_t t = e & (T - 1);
        const uint32_t k = e >> A.log_t;
        const uint32_t p = pos(bitrev32(k, A.log_r), t);
        Fe<F> x;
        ZK_UNROLL
        for (int l = 0; l < NL; l++) x.v[l] = lds[l * RT + p];
        if (!A.last) {
            const uint64_t ex = ((uint64_t)k * (nprime0 + t)) << A.log_m;
            if (ex != 0) {
                Fe<F> w;
                tw_get(w, tw, ex, half);
                fe_mul(x, x, w);
            }
            out[base + (uint64_t)k * stride_j + t] = x;
        } else {
            const uint64_t go = out_fixed + t + ((uint64_t)k << A.log_m);
            if (A.scale) fe_mul(x, x, scale);
            if (A.post) mul_pow(x, post, go);
            out[ntt_out_index(A, go)] = x;
        }
    }
}

}  // namespace zk
