// Field-side kernels of the halo2 prover steps beyond commit / FFT (halo2_proofs 0.2):
//   batch_invert_kernel            arithmetic.rs BatchInvert (Montgomery's trick, K elements per lane share one inversion; 0 stays 0)
//   scan_block / scan_totals / scan_apply   the grand products Z of the permutation and lookup arguments
//                                  (plonk/permutation/prover.rs, plonk/lookup/prover.rs): z[i] = first * prod_{j<i} f[j]
//   perm_factors_kernel            numerator and denominator of one chunk of permutation columns, row by row
//   lookup_factors_kernel          (A + beta)(S + gamma) and (A' + beta)(S' + gamma)
//   inner_product_kernel           compute_inner_product (the value_l / value_r of an IPA round)
//   vec_fold_kernel                a[i] += c a[i + half]  (the p' and b folds of an IPA round)
//   poly_eval_kernel               eval_polynomial: p(x) for a resident coefficient vector
//   kate_block / kate_totals / kate_apply   kate_division: (a - a(x)) / (X - x), the multiopen argument's per-point division
//   expr_eval_kernel               the quotient numerator: a stack program over extended-domain columns with rotations
// All HBM-streaming with a handful of Montgomery products per element; the scans are three launches (block products,
// scan of the block totals by one workgroup, apply).
#pragma once
#include "zk_rt.h"
#include "zk_field.h"
#include "zk_field29.h"
#include "zk_ntt_kernels.h"

namespace zk {

constexpr uint32_t INV_K = 16;     // elements per lane and inversion
constexpr uint32_t SCAN_K = 16;    // elements per lane of a scan workgroup (256 lanes: 4096 elements per workgroup)
constexpr uint32_t SCAN_WG = 256;

// a[i] <- 1 / a[i] (0 -> 0).  Lane t owns a[t*K .. (t+1)*K): prefix products, ONE Fermat inversion, back-substitution.
template <class F>
__global__ void __launch_bounds__(64) batch_invert_kernel(Fe<F>* __restrict__ a, uint64_t n) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo = t * INV_K;
    if (lo >= n) return;
    const uint32_t cnt = n - lo < INV_K ? (uint32_t)(n - lo) : INV_K;
    Fe<F> pre[INV_K], run;
    fe_one(run);
    for (uint32_t k = 0; k < cnt; k++) {
        pre[k] = run;
        Fe<F> x = a[lo + k];
        if (!fe_is_zero(x)) fe_mul(run, run, x);      // a zero contributes a factor 1 and stays zero
    }
    Fe<F> inv;
    fe_inv(inv, run);
    for (int k = (int)cnt - 1; k >= 0; k--) {
        Fe<F> x = a[lo + k];
        if (fe_is_zero(x)) continue;
        Fe<F> r;
        fe_mul(r, inv, pre[k]);
        fe_mul(inv, inv, x);
        a[lo + k] = r;
    }
}

// ---- exclusive multiplicative scan ----
// phase 1: workgroup b covers elements [b*4096, +4096): out[i] = product of the elements before i INSIDE the workgroup's
// range; block_tot[b] = product of the whole range
template <class F>
__global__ void __launch_bounds__(SCAN_WG) scan_block_kernel(const Fe<F>* __restrict__ in, Fe<F>* __restrict__ out, Fe<F>* __restrict__ block_tot,
                                                            uint64_t n) {
    __shared__ Fe<F> part[SCAN_WG];
    const uint32_t tid = threadIdx.x;
    const uint64_t lo = ((uint64_t)blockIdx.x * SCAN_WG + tid) * SCAN_K;
    Fe<F> v[SCAN_K], run;
    fe_one(run);
    for (uint32_t k = 0; k < SCAN_K; k++) {
        if (lo + k < n) {
            v[k] = in[lo + k];
            fe_mul(run, run, v[k]);
        } else {
            fe_one(v[k]);
        }
    }
    part[tid] = run;
    __syncthreads();
    Fe<F> acc = run;
    for (uint32_t d = 1; d < SCAN_WG; d <<= 1) {     // inclusive Hillis-Steele over the lane products
        Fe<F> o;
        const bool on = tid >= d;
        if (on) o = part[tid - d];
        __syncthreads();
        if (on) {
            fe_mul(acc, acc, o);
            part[tid] = acc;
        }
        __syncthreads();
    }
    Fe<F> pre;   // exclusive prefix of this lane
    if (tid == 0)
        fe_one(pre);
    else
        pre = part[tid - 1];
    if (tid == SCAN_WG - 1) block_tot[blockIdx.x] = acc;
    for (uint32_t k = 0; k < SCAN_K; k++) {
        if (lo + k < n) out[lo + k] = pre;
        fe_mul(pre, pre, v[k]);
    }
}
// phase 2 (one workgroup): block_tot[b] <- first * product of block_tot[0 .. b); total_out = first * product of all
template <class F>
__global__ void __launch_bounds__(SCAN_WG) scan_totals_kernel(Fe<F>* __restrict__ block_tot, uint32_t nblocks, Fe<F> first, Fe<F>* __restrict__ total_out) {
    __shared__ Fe<F> part[SCAN_WG];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (nblocks + SCAN_WG - 1) / SCAN_WG;
    const uint32_t lo = tid * per < nblocks ? tid * per : nblocks, hi = lo + per < nblocks ? lo + per : nblocks;
    Fe<F> run;
    fe_one(run);
    for (uint32_t j = lo; j < hi; j++) {
        Fe<F> x = block_tot[j];
        fe_mul(run, run, x);
    }
    part[tid] = run;
    __syncthreads();
    Fe<F> acc = run;
    for (uint32_t d = 1; d < SCAN_WG; d <<= 1) {
        Fe<F> o;
        const bool on = tid >= d;
        if (on) o = part[tid - d];
        __syncthreads();
        if (on) {
            fe_mul(acc, acc, o);
            part[tid] = acc;
        }
        __syncthreads();
    }
    Fe<F> pre = first;
    if (tid > 0) {
        Fe<F> o = part[tid - 1];
        fe_mul(pre, pre, o);
    }
    for (uint32_t j = lo; j < hi; j++) {
        Fe<F> x = block_tot[j];
        block_tot[j] = pre;
        fe_mul(pre, pre, x);
    }
    if (tid == SCAN_WG - 1) *total_out = pre;   // (the last lane's range ends at nblocks)
}
// phase 3: out[i] *= block_tot[block of i]
template <class F>
__global__ void __launch_bounds__(SCAN_WG) scan_apply_kernel(Fe<F>* __restrict__ out, const Fe<F>* __restrict__ block_tot, uint64_t n) {
    const Fe<F> off = block_tot[blockIdx.x];
    const uint64_t lo = ((uint64_t)blockIdx.x * SCAN_WG + threadIdx.x) * SCAN_K;
    for (uint32_t k = 0; k < SCAN_K; k++) {
        if (lo + k < n) {
            Fe<F> x = out[lo + k];
            fe_mul(x, x, off);
            out[lo + k] = x;
        }
    }
}

// ---- permutation argument, one chunk of <= 8 columns ----
template <class F>
struct PermChunk {
    const Fe<F>* col[8];
    const Fe<F>* sigma[8];
    Fe<F> dcoef[8];      // beta * delta^(index of the column in the whole argument)
    Fe<F> beta, gamma;
    uint32_t ncols;
};
// num[i] = prod_c (v_c[i] + dcoef_c omega^i + gamma) ; den[i] = prod_c (v_c[i] + beta sigma_c[i] + gamma)
template <class F>
__global__ void __launch_bounds__(256) perm_factors_kernel(PermChunk<F> ch, PowTables<F> wpow, Fe<F>* __restrict__ num, Fe<F>* __restrict__ den,
                                                           uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Fe<F> w, a, b;
        fe_one(w);
        mul_pow(w, wpow, i);      // omega^i
        fe_one(a);
        fe_one(b);
        for (uint32_t c = 0; c < ch.ncols; c++) {
            Fe<F> v = ch.col[c][i], s = ch.sigma[c][i], t, u;
            fe_mul(t, ch.dcoef[c], w);
            fe_add(t, t, v);
            fe_add(t, t, ch.gamma);
            fe_mul(a, a, t);
            fe_mul(u, ch.beta, s);
            fe_add(u, u, v);
            fe_add(u, u, ch.gamma);
            fe_mul(b, b, u);
        }
        num[i] = a;
        den[i] = b;
    }
}
// num[i] = (A + beta)(S + gamma), den[i] = (A' + beta)(S' + gamma)
template <class F>
__global__ void __launch_bounds__(256) lookup_factors_kernel(const Fe<F>* __restrict__ A, const Fe<F>* __restrict__ S, const Fe<F>* __restrict__ Ap,
                                                             const Fe<F>* __restrict__ Sp, Fe<F> beta, Fe<F> gamma, Fe<F>* __restrict__ num,
                                                             Fe<F>* __restrict__ den, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Fe<F> a = A[i], s = S[i], ap = Ap[i], sp = Sp[i];
        fe_add(a, a, beta);
        fe_add(s, s, gamma);
        fe_mul(a, a, s);
        fe_add(ap, ap, beta);
        fe_add(sp, sp, gamma);
        fe_mul(ap, ap, sp);
        num[i] = a;
        den[i] = ap;
    }
}

// ---- IPA scalar side ----
// partial[b] = sum over the workgroup's grid-stride share of a[i] b[i]
template <class F>
__global__ void __launch_bounds__(256) inner_product_kernel(const Fe<F>* __restrict__ a, const Fe<F>* __restrict__ b, uint64_t n, Fe<F>* __restrict__ partial) {
    __shared__ Fe<F> part[256];
    Fe<F> acc;
    fe_zero(acc);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Fe<F> x = a[i], y = b[i];
        fe_mul(x, x, y);
        fe_add(acc, acc, x);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) {
            Fe<F> o = part[threadIdx.x + d];
            fe_add(acc, acc, o);
            part[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
// a[i] += c * a[i + half], i < half
// a[i] = a[i] * s + b[i]: one Horner step of the multiopen combination (poly/multiopen/prover.rs: the polynomials queried at the
// same point set are folded with powers of x_1, the per-set quotients with x_4) on resident coefficient vectors
template <class F>
__global__ void __launch_bounds__(256) vec_muladd_kernel(Fe<F>* out, const Fe<F>* a, const Fe<F>* __restrict__ b, uint64_t n, Fe<F> s) {
    // out[i] = a[i] s + b[i]; out == a is the in-place Horner step (every lane reads its element before it writes it)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Fe<F> x = a[i];
        const Fe<F> y = b[i];
        fe_mul(x, x, s);
        fe_add(x, x, y);
        out[i] = x;
    }
}

// lad[k] = x^(2^k), k < 10 ; lad[32 + k] = (x^1024)^(2^k), k < 22: the ladders pow_table_kernel builds the power tables of a fresh
// challenge from (one lane: 32 dependent squarings)
template <class F>
__global__ void __launch_bounds__(64) pow_ladder_kernel(Fe<F> x, Fe<F>* __restrict__ lad) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Fe<F> w = x;
    for (int k = 0; k < 10; k++) {
        lad[k] = w;
        fe_sqr(w, w);
    }
    for (int k = 0; k < 22; k++) {
        lad[32 + k] = w;
        fe_sqr(w, w);
    }
}

// out[j] = sum_i s^(count - 1 - i) src_i[j], src_i = first + i * stride (stride in elements, may be negative): the whole Horner
// fold of `count` resident polynomials in ONE pass -- every polynomial is read once and the result written once, where count - 1
// muladd launches re-read and re-write the accumulator each time (3 x the traffic).  The multiopen argument folds every point
// set's polynomials this way (powers of x_1), the vanishing argument the pieces of h (powers of x^n, last piece first).
template <class F>
__global__ void __launch_bounds__(256) vec_fold_many_kernel(Fe<F>* __restrict__ out, const Fe<F>* first, int64_t stride, uint32_t count, uint64_t n,
                                                            Fe<F> s) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        const Fe<F>* p = first + j;
        Fe<F> acc = *p;
        Fe<F> nxt = acc;
        if (count > 1) nxt = p[stride];
        for (uint32_t i = 1; i < count; i++) {
            const Fe<F> y = nxt;
            if (i + 1 < count) nxt = p[(int64_t)(i + 1) * stride];      // the next load overlaps the product
            fe_mul(acc, acc, s);
            fe_add(acc, acc, y);
        }
        out[j] = acc;
    }
}

// one IPA round's three folds in one launch (poly/commitment/prover.rs: p'_i += u^-1 p'_(i+half), b_i += u b_(i+half), and for the
// fold-free form the weights W[idx] *= u where idx has bit `half` set): the later rounds are bound by their dependent launches
template <class F>
__global__ void __launch_bounds__(256) ipa_fold_round_kernel(Fe<F>* __restrict__ p, Fe<F>* __restrict__ b, Fe<F>* __restrict__ W, uint64_t half,
                                                             uint64_t m0, Fe<F> u_inv, Fe<F> u) {
    const uint64_t work = half > m0 ? half : m0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < work; i += (uint64_t)gridDim.x * blockDim.x) {
        if (i < half) {
            Fe<F> x = p[i], y = p[i + half];
            fe_mul(y, y, u_inv);
            fe_add(x, x, y);
            p[i] = x;
            x = b[i];
            y = b[i + half];
            fe_mul(y, y, u);
            fe_add(x, x, y);
            b[i] = x;
        }
        if (W != nullptr && i < m0 && (i & half)) {
            Fe<F> x = W[i];
            fe_mul(x, x, u);
            W[i] = x;
        }
    }
}

// out[i] = x^i (poly/commitment/prover.rs builds `b`, the powers of x_3, this way before the argument's rounds)
template <class F>
__global__ void __launch_bounds__(256) vec_powers_kernel(Fe<F>* __restrict__ out, uint64_t n, PowTables<F> pw) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Fe<F> a = pw.lo[i & 1023];
        const Fe<F> b = pw.hi[i >> 10];
        fe_mul(a, a, b);
        out[i] = a;
    }
}

// p(x) = sum_i c_i x^i (halo2 arithmetic.rs eval_polynomial: the evaluations at x, omega x, ... that create_proof writes to the
// transcript).  Lane t owns coefficients [t K, (t + 1) K): Horner inside the chunk, one multiplication by x^(t K) from the
// power tables of x, then the block's tree sum; the host adds the per-block partial sums.  grid.y polynomials (stride apart) share x.
constexpr uint32_t EVAL_K = 16;
template <class F>
__global__ void __launch_bounds__(256) poly_eval_kernel(const Fe<F>* __restrict__ c, uint64_t n, uint64_t stride, Fe<F> x, PowTables<F> pw,
                                                        Fe<F>* __restrict__ partial) {
    __shared__ Fe<F> part[256];
    c += (uint64_t)blockIdx.y * stride;          // grid.y = polynomial (all evaluated at the same x)
    partial += (uint64_t)blockIdx.y * gridDim.x;
    Fe<F> acc;
    fe_zero(acc);
    const uint64_t lanes = (n + EVAL_K - 1) / EVAL_K;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < lanes; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t lo = t * EVAL_K;
        const uint32_t cnt = n - lo < EVAL_K ? (uint32_t)(n - lo) : EVAL_K;
        Fe<F> h = c[lo + cnt - 1];
        for (int k = (int)cnt - 2; k >= 0; k--) {
            fe_mul(h, h, x);
            const Fe<F> ck = c[lo + k];
            fe_add(h, h, ck);
        }
        mul_pow(h, pw, lo);
        fe_add(acc, acc, h);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) {
            const Fe<F> o = part[threadIdx.x + d];
            fe_add(acc, acc, o);
            part[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// ---- kate_division (halo2_proofs 0.2 arithmetic.rs; poly/multiopen/prover.rs divides every point set's folded polynomial by
// (X - x) for each of its points): q = (a - a(x)) / (X - x), i.e. q[j] = sum_{i > j} a[i] x^(i - j - 1) and q[n - 1] = 0 (upstream
// returns n - 1 coefficients and the multiopen prover resizes to n).  With A(j) = sum_{i >= j} a[i] x^(i - j) the output is
// q[j] = A(j + 1): a suffix Horner recurrence, run as a three-phase scan like the grand products --
//   kate_block_kernel    workgroup = 4096 coefficients: per-lane Horner of 16, LDS suffix scan over the lanes with the weights
//                        x^16, x^32, ..., then q inside the block (as if nothing followed it) and the block's total A_block
//   kate_totals_kernel   one workgroup: block_tot[b] <- C_b = A(end of block b) from the totals, weight x^4096
//   kate_apply_kernel    q[j] += x^(end_b - j - 1) C_b
// ~3 products per coefficient; a == q allowed (every lane reads its 16 coefficients before it writes them).
constexpr uint32_t KATE_K = 16, KATE_WG = 256, KATE_LOG_WG = 8;
template <class F>
struct KatePows {
    Fe<F> xk[KATE_LOG_WG + 1];   // x^(KATE_K 2^s), s = 0 .. 8: the lane weights of the scan; xk[8] = x^4096 is the block weight
};
template <class F>
__global__ void __launch_bounds__(KATE_WG) kate_block_kernel(const Fe<F>* a, Fe<F>* q, Fe<F>* __restrict__ block_tot, uint64_t n, Fe<F> x,
                                                             KatePows<F> pw) {
    __shared__ Fe<F> part[KATE_WG];
    const uint32_t tid = threadIdx.x;
    const uint64_t lo = ((uint64_t)blockIdx.x * KATE_WG + tid) * KATE_K;
    Fe<F> v[KATE_K], h;
    fe_zero(h);
    for (int k = (int)KATE_K - 1; k >= 0; k--) {
        if (lo + k < n)
            v[k] = a[lo + k];
        else
            fe_zero(v[k]);
        fe_mul(h, h, x);
        fe_add(h, h, v[k]);
    }
    part[tid] = h;
    __syncthreads();
    Fe<F> acc = h;
    for (uint32_t s = 0; s < KATE_LOG_WG; s++) {      // inclusive suffix scan: acc_t = sum_{t' >= t} h_t' x^(16 (t' - t))
        const uint32_t d = 1u << s;
        const bool on = tid + d < KATE_WG;
        Fe<F> o;
        if (on) o = part[tid + d];
        __syncthreads();
        if (on) {
            fe_mul(o, o, pw.xk[s]);
            fe_add(acc, acc, o);
            part[tid] = acc;
        }
        __syncthreads();
    }
    if (tid == 0) block_tot[blockIdx.x] = acc;
    Fe<F> r;
    if (tid + 1 < KATE_WG)
        r = part[tid + 1];
    else
        fe_zero(r);
    for (int k = (int)KATE_K - 1; k >= 0; k--) {
        if (lo + k < n) q[lo + k] = r;
        fe_mul(r, r, x);
        fe_add(r, r, v[k]);
    }
}
// one workgroup: block_tot[b] <- sum_{b' > b} T_b' X^(b' - b - 1), X = x^4096
template <class F>
__global__ void __launch_bounds__(KATE_WG) kate_totals_kernel(Fe<F>* __restrict__ block_tot, uint32_t nblocks, Fe<F> X) {
    __shared__ Fe<F> part[KATE_WG];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (nblocks + KATE_WG - 1) / KATE_WG;
    const uint32_t lo = tid * per < nblocks ? tid * per : nblocks, hi = lo + per < nblocks ? lo + per : nblocks;
    Fe<F> h;
    fe_zero(h);
    for (uint32_t j = hi; j-- > lo;) {
        const Fe<F> t = block_tot[j];
        fe_mul(h, h, X);
        fe_add(h, h, t);
    }
    Fe<F> m;                   // X^per: the weight between neighbouring lanes (every lane spans `per` blocks, missing ones count as zero)
    fe_one(m);
    for (uint32_t k = 0; k < per; k++) fe_mul(m, m, X);
    part[tid] = h;
    __syncthreads();
    Fe<F> acc = h;
    for (uint32_t d = 1; d < KATE_WG; d <<= 1) {
        const bool on = tid + d < KATE_WG;
        Fe<F> o;
        if (on) o = part[tid + d];
        __syncthreads();
        if (on) {
            fe_mul(o, o, m);
            fe_add(acc, acc, o);
            part[tid] = acc;
        }
        __syncthreads();
        fe_sqr(m, m);
    }
    Fe<F> r;
    if (tid + 1 < KATE_WG)
        r = part[tid + 1];
    else
        fe_zero(r);
    for (uint32_t j = hi; j-- > lo;) {
        const Fe<F> t = block_tot[j];
        block_tot[j] = r;
        fe_mul(r, r, X);
        fe_add(r, r, t);
    }
}
template <class F>
__global__ void __launch_bounds__(KATE_WG) kate_apply_kernel(Fe<F>* __restrict__ q, const Fe<F>* __restrict__ carry, uint64_t n, Fe<F> x,
                                                             PowTables<F> pw) {
    const uint32_t tid = threadIdx.x;
    const uint64_t lo = ((uint64_t)blockIdx.x * KATE_WG + tid) * KATE_K;
    if (lo >= n) return;
    Fe<F> m = carry[blockIdx.x];
    if (fe_is_zero(m)) return;                                   // (the last block; wave-uniform)
    mul_pow(m, pw, (uint64_t)(KATE_WG - 1 - tid) * KATE_K);      // C_b x^(end_b - lo - 16)
    for (int k = (int)KATE_K - 1; k >= 0; k--) {
        if (lo + k < n) {
            Fe<F> y = q[lo + k];
            fe_add(y, y, m);
            q[lo + k] = y;
        }
        fe_mul(m, m, x);
    }
}

template <class F>
__global__ void __launch_bounds__(256) vec_fold_kernel(Fe<F>* __restrict__ a, uint64_t half, Fe<F> c) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (uint64_t)gridDim.x * blockDim.x) {
        Fe<F> x = a[i], y = a[i + half];
        fe_mul(y, y, c);
        fe_add(x, x, y);
        a[i] = x;
    }
}

// ---- IPA without folding the generators ----
// Folding G' costs one 255-bit scalar multiplication per surviving point and round (~260 point operations each, n of them
// over the argument); an MSM costs ~16 bucket additions per point.  So the rounds are run over the ORIGINAL generators G0
// (the resident SRS) instead: after the challenges u_1 .. u_r the folded generator is G'_i = sum_t W[i + t cur] G0[i + t cur]
// (cur = n / 2^r, W[idx] = product of the u_a whose fold put idx in an upper half), hence
//     L = <p'_hi, G'_lo> = MSM(G0, S_L),  S_L[idx] = p'[i + cur/2] W[idx] for i = idx mod cur < cur/2, else 0
//     R = <p'_lo, G'_hi> = MSM(G0, S_R),  S_R[idx] = p'[i - cur/2] W[idx] for i >= cur/2,              else 0
// two n-point MSMs with half of the scalars zero per round, one batched call; W is updated in place after each challenge.
template <class F>
__global__ void __launch_bounds__(256) ipa_virtual_scalars_kernel(const Fe<F>* __restrict__ p, const Fe<F>* __restrict__ W, Fe<F>* __restrict__ SL,
                                                                  Fe<F>* __restrict__ SR, uint64_t m0, uint64_t cur) {
    const uint64_t half = cur >> 1;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < m0; idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = idx & (cur - 1);
        Fe<F> w = W[idx], z, x;
        fe_zero(z);
        if (i < half) {
            x = p[i + half];
            fe_mul(x, x, w);
            SL[idx] = x;
            SR[idx] = z;
        } else {
            x = p[i - half];
            fe_mul(x, x, w);
            SL[idx] = z;
            SR[idx] = x;
        }
    }
}
// W[idx] *= u where idx has `bit` set (the upper half of the fold just made, bit = cur / 2)
template <class F>
__global__ void __launch_bounds__(256) ipa_update_weights_kernel(Fe<F>* __restrict__ W, uint64_t m0, uint64_t bit, Fe<F> u) {
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < m0; idx += (uint64_t)gridDim.x * blockDim.x) {
        if (idx & bit) {
            Fe<F> x = W[idx];
            fe_mul(x, x, u);
            W[idx] = x;
        }
    }
}

// ---- quotient numerator: a stack program evaluated at every row of the extended domain ----
// (halo2_proofs 0.2 plonk/prover.rs: every gate polynomial, multiplied into the running sum by y, is an Expression over
// advice / fixed / instance columns with rotations, evaluated on the extended coset; a rotation by r rows is a shift of
// r * 2^(extended_k - k) positions there)
struct ExprOp {
    uint8_t op;      // 0 col(arg = column, rot)  1 const(arg)  2 add  3 sub  4 mul  5 neg  6 scale(arg = constant)
    uint8_t pad;
    int16_t rot;
    uint32_t arg;
};
constexpr uint32_t EXPR_MAX_OPS = 512, EXPR_MAX_COLS = 64, EXPR_MAX_CONSTS = 32, EXPR_STACK = 8, EXPR_WG = 128;

// LDS: per-lane stack below the top (which stays in registers), limb-major (bank-conflict-free): stack[(slot * N + limb) * EXPR_WG + lane]
// The program is read as one aligned 64-bit word per op (op | rot << 16 | arg << 32) and decoded with shifts: every lane
// runs the same program, so the compiler turns these reads into scalar loads, and a scalar load of a field at a 2-byte
// offset inside the struct is not something to rely on.  Operand indices are clamped against the table sizes as well:
// the host validates the program, the kernel still never indexes past its tables.
template <class F>
__global__ void __launch_bounds__(EXPR_WG) expr_eval_kernel(const uint64_t* __restrict__ prog, uint32_t n_ops, const Fe<F>* const* __restrict__ cols,
                                                            uint32_t n_cols, const Fe<F>* __restrict__ consts, uint32_t n_consts, uint32_t log_n,
                                                            uint32_t rot_scale, Fe<F>* __restrict__ out) {
    __shared__ uint32_t stack[EXPR_STACK * F::N * EXPR_WG];
    const uint32_t lane = threadIdx.x;
    const uint64_t n = 1ull << log_n, mask = n - 1;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + lane; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        // the top of the stack lives in registers (`tos`); slots 0 .. sp - 2 in LDS.  A binary operation then costs one LDS
        // read instead of two reads and a write, a unary one none: the adds and subtractions of a gate expression (~25
        // instructions each) were spending twice that on moving operands.  (Measured: no change in kernel time -- the 81 saturated
        // products per row set it, two dependent VALU instructions per partial product; the column values would have to arrive in
        // the lazy-limb radix for the cheaper product to apply.)
        uint32_t sp = 0;
        Fe<F> tos;
        fe_zero(tos);
        for (uint32_t k = 0; k < n_ops; k++) {
            const uint64_t w = prog[k];
            const uint32_t op = (uint32_t)(w & 0xff), arg = (uint32_t)(w >> 32);
            const int32_t rot = (int32_t)(int16_t)(uint16_t)(w >> 16);
            if (op <= 1) {
                if (sp >= EXPR_STACK) break;
                if (sp >= 1) {
                    ZK_UNROLL
                    for (int l = 0; l < F::N; l++) stack[((sp - 1) * F::N + l) * EXPR_WG + lane] = tos.v[l];
                }
                if (op == 0) {
                    const uint64_t j = (i + (uint64_t)((int64_t)rot * (int64_t)rot_scale)) & mask;
                    tos = cols[arg < n_cols ? arg : 0][j];
                } else {
                    tos = consts[arg < n_consts ? arg : 0];
                }
                sp++;
            } else if (op == 5 || op == 6) {
                if (sp < 1) break;
                if (op == 5) {
                    fe_neg(tos, tos);
                } else {
                    const Fe<F> y = consts[arg < n_consts ? arg : 0];
                    fe_mul(tos, tos, y);
                }
            } else {
                if (sp < 2) break;
                Fe<F> x;
                ZK_UNROLL
                for (int l = 0; l < F::N; l++) x.v[l] = stack[((sp - 2) * F::N + l) * EXPR_WG + lane];
                if (op == 2)
                    fe_add(x, x, tos);
                else if (op == 3)
                    fe_sub(x, x, tos);
                else
                    fe_mul(x, x, tos);
                tos = x;
                sp--;
            }
        }
        out[i] = tos;
    }
}

// ---- the same evaluator on lazy 29-bit limbs (zk_field29.h): zk_expr_eval_lazy_device ----
// The saturated form above spends ~290 instructions per product plus carry chains in every addition, and every operand goes
// through LDS as 8 words.  Here the columns arrive as x R' mod p (R' = 2^261: the extended-coset transforms write that form when
// asked, ZK_NTT_OUT_R29), are unpacked to 9 limbs on load, and products are the one-MAD-per-partial-product form (~205
// instructions), additions 9 limb adds, subtractions 9 biased limb subtracts.  The bound discipline of zk_field29.h is
// enforced by the HOST, which walks the program once (expr_compile29 in zk_poly.inl) tracking a limb bound and a value bound
// per stack slot, picks the bias table of every subtraction / negation, and inserts the carry steps (EXPR29_NORM) and value
// contractions (EXPR29_REFRESH: a product by R' mod p) that keep every product's operands inside fe29_mul's precondition.
// The kernel only executes: it never decides anything about bounds.
//   ops: 0 push slot(arg = slot)  1 const  2 add  3 sub(arg = bias id)  4 mul  5 neg(arg = bias id)  6 scale(arg = const)  7 norm
//        8 refresh  9 load(arg = column | slot << 16, rot): slot <- column[row + rot]
//   bias ids: 0 BIAS4K1  1 BIAS4K2  2 BIAS8K2  3 BIAS8K3  4 BIAS16K2
// Column values go through EXPR29_SLOTS register slots per lane: a gate expression reads the same (column, rotation) several
// times (a^5 is five reads of a when the AST is walked as upstream does) and every read is 32 B per row from L2 / HBM -- the
// bench's program has 99 reads of 37 distinct cells; two slots with farthest-next-use eviction (the host knows the whole
// program) leave 69, four would leave 51.  The host also hoists every load a few operations ahead of its first use.
// One wave per workgroup (the stacks are per lane: no barrier anywhere); the LDS stack is sized by the program's depth, so
// shallow programs get more resident waves to hide the column loads behind.
constexpr uint32_t EXPR29_WG = 64, EXPR29_SLOTS = 2;   // (4 slots: see the note at the slot array below)
constexpr uint32_t EXPR_JIT_SLOTS = 4, EXPR_JIT_SLOTS_MAX = 8;   // column slots of the specialised kernel (zk_poly.inl: expr_jit_source)
enum : uint32_t { EXPR29_NORM = 7, EXPR29_REFRESH = 8, EXPR29_LOAD = 9 };

template <class F>
__device__ __forceinline__ void fe29_sub_by_id(Fe29<F>& r, const Fe29<F>& a, const Fe29<F>& b, uint32_t id) {
    using K = F29<F>;
    if (id == 0)
        fe29_sub(r, a, b, K::BIAS4K1);
    else if (id == 1)
        fe29_sub(r, a, b, K::BIAS4K2);
    else if (id == 2)
        fe29_sub(r, a, b, K::BIAS8K2);
    else if (id == 3)
        fe29_sub(r, a, b, K::BIAS8K3);
    else
        fe29_sub(r, a, b, K::BIAS16K2);
}

template <class F>
__global__ void __launch_bounds__(EXPR29_WG) expr_eval29_kernel(const uint64_t* __restrict__ prog, uint32_t n_ops, const Fe<F>* const* __restrict__ cols,
                                                                uint32_t n_cols, const Fe<F>* __restrict__ consts, uint32_t n_consts, uint32_t log_n,
                                                                uint32_t rot_scale, uint32_t depth, Fe<F>* __restrict__ out) {
    ZK_DYN_SHARED(uint32_t, stack);   // depth slots: stack[(slot * L + limb) * EXPR29_WG + lane]
    using K = F29<F>;
    constexpr int L = K::L;
    const uint32_t lane = threadIdx.x;
    const uint64_t n = 1ull << log_n, mask = n - 1;
    // a wave-uniform row loop (the whole wave takes rows base .. base + 63): the program words are then scalar loads and every
    // branch on an opcode is a scalar branch; lanes past the end (only when n < 64) compute on wrapped rows and store nothing
    for (uint64_t base = (uint64_t)blockIdx.x * EXPR29_WG; base < n; base += (uint64_t)gridDim.x * EXPR29_WG) {
        const uint64_t i = base + lane;
        uint32_t sp = 0;
        Fe29<F> tos;
        fe29_zero(tos);
        // the column-value slots.  TWO of them: with four (named registers c0 .. c3 selected by an if-chain on the wave-uniform slot
        // number) the gfx950 build returned wrong rows as soon as slots 2 / 3 were in use -- in the ISA a load into slot 3 also
        // overwrote slot 2 -- while one and two slots, and the CPU build of the same source with four, were right
        // (profiles/r03_b_expr29_slots.txt: the matrix of slot counts x load hoisting on the GPU).
        uint32_t cs[EXPR29_SLOTS * F::N];
        ZK_UNROLL
        for (uint32_t q = 0; q < EXPR29_SLOTS * F::N; q++) cs[q] = 0;
        for (uint32_t k = 0; k < n_ops; k++) {
            const uint64_t w = prog[ZK_UNIFORM32(k)];          // (k is wave-uniform: a scalar load)
            const uint32_t op = (uint32_t)(w & 0xff), arg = (uint32_t)(w >> 32);
            const int32_t rot = (int32_t)(int16_t)(uint16_t)(w >> 16);
            if (op == EXPR29_LOAD) {
                const uint32_t col = arg & 0xffffu, slot = (arg >> 16) & (EXPR29_SLOTS - 1);
                const uint64_t j = (i + (uint64_t)((int64_t)rot * (int64_t)rot_scale)) & mask;
                const Fe<F> raw = cols[col < n_cols ? col : 0][j];
                ZK_UNROLL
                for (int l = 0; l < F::N; l++) cs[slot * F::N + l] = raw.v[l];
            } else if (op <= 1) {
                if (sp > depth) break;
                if (sp >= 1) {
                    ZK_UNROLL
                    for (int l = 0; l < L; l++) stack[((sp - 1) * L + l) * EXPR29_WG + lane] = tos.v[l];
                }
                Fe<F> raw;
                if (op == 0) {
                    const uint32_t slot = arg & (EXPR29_SLOTS - 1);
                    ZK_UNROLL
                    for (int l = 0; l < F::N; l++) raw.v[l] = cs[slot * F::N + l];
                } else {
                    raw = consts[arg < n_consts ? arg : 0];
                }
                fe29_unpack(tos, raw);
                sp++;
            } else if (op == 5) {
                Fe29<F> z;
                fe29_zero(z);
                fe29_sub_by_id<F>(tos, z, tos, arg);
            } else if (op == 6) {
                Fe29<F> c;
                fe29_unpack(c, consts[arg < n_consts ? arg : 0]);
                fe29_mul(tos, tos, c);
            } else if (op == EXPR29_NORM) {
                fe29_norm(tos, tos);
            } else if (op == EXPR29_REFRESH) {
                Fe29<F> one;
                fe29_one(one);
                fe29_mul(tos, tos, one);
            } else {
                if (sp < 2) break;
                Fe29<F> x;
                ZK_UNROLL
                for (int l = 0; l < L; l++) x.v[l] = stack[((sp - 2) * L + l) * EXPR29_WG + lane];
                if (op == 2)
                    fe29_add(tos, x, tos);
                else if (op == 3)
                    fe29_sub_by_id<F>(tos, x, tos, arg);
                else
                    fe29_mul(tos, x, tos);
                sp--;
            }
        }
        Fe<F> r;
        fe29_to_std(r, tos);      // x R' -> x R, canonical (the host left tos normalised enough for this product)
        if (i < n) out[i] = r;
    }
}

}  // namespace zk
