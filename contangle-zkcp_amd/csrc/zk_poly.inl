// Launch sequences of the halo2 prover steps beyond commit / FFT (zk_poly_kernels.h).  Included by zk_ntt.inl, once per
// scalar field.  Scratch (numerators / denominators, block totals, partial sums) belongs to the caller's stream.
#pragma once
#include <map>
#include <utility>
#include "zk_poly_kernels.h"
namespace zk {

template <class F>
int batch_invert_run(Fe<F>* a, uint64_t n, hipStream_t st) {
    if (n == 0) return ZK_OK;
    const uint64_t lanes = (n + INV_K - 1) / INV_K;
    ZK_LAUNCH((batch_invert_kernel<F>), (unsigned)((lanes + 63) / 64), 64, 0, st, a, n);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}

// out[i] = first * prod_{j < i} in[j]  (in == out allowed); *total_dev (one element of scratch) = first * prod of all
template <class F>
int prefix_product_run(DeviceCtx& dc, const Fe<F>* in, Fe<F>* out, uint64_t n, const Fe<F>& first, Fe<F>** total_dev, hipStream_t st) {
    StreamScratch* ss = nullptr;
    ZK_TRY(stream_scratch(dc, st, &ss));
    const uint64_t per_wg = (uint64_t)SCAN_WG * SCAN_K;
    const uint64_t nblocks = n ? (n + per_wg - 1) / per_wg : 1;
    if (nblocks > (1u << 24)) return ZK_ERR_UNSUPPORTED;
    ZK_TRY(ws_get(ss->poly_tot, (nblocks + 1) * sizeof(Fe<F>)));
    Fe<F>* tot = (Fe<F>*)ss->poly_tot.p;
    if (n) ZK_LAUNCH((scan_block_kernel<F>), (unsigned)nblocks, SCAN_WG, 0, st, in, out, tot, n);
    else HIP_TRY(hipMemcpyAsync(tot, F::R, sizeof(Fe<F>), hipMemcpyHostToDevice, st));
    ZK_LAUNCH((scan_totals_kernel<F>), 1, SCAN_WG, 0, st, tot, (uint32_t)nblocks, first, tot + nblocks);
    if (n) ZK_LAUNCH((scan_apply_kernel<F>), (unsigned)nblocks, SCAN_WG, 0, st, out, (const Fe<F>*)tot, n);
    HIP_TRY(hipGetLastError());
    if (total_dev) *total_dev = tot + nblocks;
    return ZK_OK;
}

// f = num / den elementwise (den is inverted in place), then z = exclusive product scan of f
template <class F>
int ratio_scan_run(DeviceCtx& dc, Fe<F>* num, Fe<F>* den, uint64_t n, const Fe<F>& first, Fe<F>* z_out, void* total_host, hipStream_t st) {
    ZK_TRY(batch_invert_run<F>(den, n, st));
    Fe<F> one;
    fe_one(one);
    ZK_TRY(vec_op_run<F>(num, den, nullptr, n, VEC_MUL, one, st));
    Fe<F>* total = nullptr;
    ZK_TRY(prefix_product_run<F>(dc, num, z_out, n, first, &total, st));
    if (total_host) {
        HIP_TRY(hipMemcpyAsync(total_host, total, sizeof(Fe<F>), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return ZK_OK;
}

template <class F>
int perm_product_run(DeviceCtx& dc, int field, uint32_t ncols, const void* const* cols, const void* const* sigmas, uint32_t first_col,
                     const Fe<F>& beta, const Fe<F>& gamma, const Fe<F>& delta, uint32_t k, const Fe<F>& omega, const Fe<F>& first, Fe<F>* z_out,
                     void* total_host, hipStream_t st) {
    if (ncols == 0 || ncols > 8 || k > 30) return ZK_ERR_INVALID_ARG;
    const uint64_t n = 1ull << k;
    PermChunk<F> ch;
    memset(&ch, 0, sizeof ch);
    ch.ncols = ncols;
    ch.beta = beta;
    ch.gamma = gamma;
    Fe<F> dp;
    fe_one(dp);
    for (uint32_t c = 0; c < first_col; c++) fe_mul(dp, dp, delta);
    for (uint32_t c = 0; c < ncols; c++) {
        if (!cols[c] || !sigmas[c]) return ZK_ERR_INVALID_ARG;
        ch.col[c] = (const Fe<F>*)cols[c];
        ch.sigma[c] = (const Fe<F>*)sigmas[c];
        fe_mul(ch.dcoef[c], dp, beta);
        fe_mul(dp, dp, delta);
    }
    PowTables<F> wpow;
    ZK_TRY(pow_tables<F>(dc, omega, k, field, st, &wpow));
    StreamScratch* ss = nullptr;
    ZK_TRY(stream_scratch(dc, st, &ss));
    ZK_TRY(ws_get(ss->poly_a, n * sizeof(Fe<F>)));
    ZK_TRY(ws_get(ss->poly_b, n * sizeof(Fe<F>)));
    Fe<F>*num = (Fe<F>*)ss->poly_a.p, *den = (Fe<F>*)ss->poly_b.p;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ZK_LAUNCH((perm_factors_kernel<F>), (unsigned)blocks, 256, 0, st, ch, wpow, num, den, n);
    return ratio_scan_run<F>(dc, num, den, n, first, z_out, total_host, st);
}

template <class F>
int lookup_product_run(DeviceCtx& dc, const Fe<F>* A, const Fe<F>* S, const Fe<F>* Ap, const Fe<F>* Sp, const Fe<F>& beta, const Fe<F>& gamma,
                       uint64_t n, const Fe<F>& first, Fe<F>* z_out, void* total_host, hipStream_t st) {
    StreamScratch* ss = nullptr;
    ZK_TRY(stream_scratch(dc, st, &ss));
    ZK_TRY(ws_get(ss->poly_a, (n ? n : 1) * sizeof(Fe<F>)));
    ZK_TRY(ws_get(ss->poly_b, (n ? n : 1) * sizeof(Fe<F>)));
    Fe<F>*num = (Fe<F>*)ss->poly_a.p, *den = (Fe<F>*)ss->poly_b.p;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (n) ZK_LAUNCH((lookup_factors_kernel<F>), (unsigned)blocks, 256, 0, st, A, S, Ap, Sp, beta, gamma, num, den, n);
    return ratio_scan_run<F>(dc, num, den, n, first, z_out, total_host, st);
}

// <a, b>: per-workgroup partial sums, added on the host (synchronises the stream)
template <class F>
int inner_product_run(DeviceCtx& dc, const Fe<F>* a, const Fe<F>* b, uint64_t n, void* out_host, hipStream_t st) {
    Fe<F> acc;
    fe_zero(acc);
    if (n) {
        StreamScratch* ss = nullptr;
        ZK_TRY(stream_scratch(dc, st, &ss));
        uint64_t blocks = (n + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        ZK_TRY(ws_get(ss->poly_tot, blocks * sizeof(Fe<F>)));
        ZK_LAUNCH((inner_product_kernel<F>), (unsigned)blocks, 256, 0, st, a, b, n, (Fe<F>*)ss->poly_tot.p);
        HIP_TRY(hipGetLastError());
        std::vector<Fe<F>> part(blocks);
        HIP_TRY(hipMemcpyAsync(part.data(), ss->poly_tot.p, blocks * sizeof(Fe<F>), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (auto& p : part) fe_add(acc, acc, p);
    }
    host_store(out_host, acc);
    return ZK_OK;
}

// One IPA round's scalar side without a host round trip per value (zk_ipa_round_device): the fold-free round's MSM scalars
// and both inner products <p'_hi, b_lo>, <p'_lo, b_hi> are enqueued, their per-workgroup partial sums copied to pinned host
// memory behind an event; ipa_round_end_run adds them up after the round's MSMs have been collected.
template <class F>
int ipa_round_begin_run(DeviceCtx& dc, const Fe<F>* p, const Fe<F>* b, const Fe<F>* W, Fe<F>* SL, Fe<F>* SR, uint64_t m0, uint64_t cur,
                        hipStream_t st) {
    ZK_TRY(ipa_virtual_scalars_run<F>(p, W, SL, SR, m0, cur, st));
    const uint64_t half = cur / 2;
    StreamScratch* ss = nullptr;
    ZK_TRY(stream_scratch(dc, st, &ss));
    uint64_t blocks = (half + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    const size_t bytes = 2 * blocks * sizeof(Fe<F>);
    ZK_TRY(ws_get(ss->poly_tot, bytes));
    if (ss->pinned_cap < bytes) {
        if (ss->pinned) hipHostFree(ss->pinned);
        ss->pinned = nullptr;
        ss->pinned_cap = 0;
        HIP_TRY(hipHostMalloc(&ss->pinned, 2 * 1024 * sizeof(Fe<F>), 0));
        ss->pinned_cap = 2 * 1024 * sizeof(Fe<F>);
    }
    if (!ss->pinned_ev) HIP_TRY(hipEventCreate(&ss->pinned_ev));
    Fe<F>* tot = (Fe<F>*)ss->poly_tot.p;
    if (half) {
        ZK_LAUNCH((inner_product_kernel<F>), (unsigned)blocks, 256, 0, st, p + half, b, half, tot);
        ZK_LAUNCH((inner_product_kernel<F>), (unsigned)blocks, 256, 0, st, p, b + half, half, tot + blocks);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(ss->pinned, tot, bytes, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipEventRecord(ss->pinned_ev, st));
    ss->ip_blocks = half ? (uint32_t)blocks : 0;
    return ZK_OK;
}
template <class F>
int ipa_round_end_run(DeviceCtx& dc, hipStream_t st, void* vl_host, void* vr_host) {
    StreamScratch* ss = nullptr;
    ZK_TRY(stream_scratch(dc, st, &ss));
    if (!ss->pinned_ev) return ZK_ERR_INVALID_ARG;
    HIP_TRY(hipEventSynchronize(ss->pinned_ev));
    const Fe<F>* part = (const Fe<F>*)ss->pinned;
    for (int k = 0; k < 2; k++) {
        Fe<F> acc;
        fe_zero(acc);
        for (uint32_t i = 0; i < ss->ip_blocks; i++) fe_add(acc, acc, part[(size_t)k * ss->ip_blocks + i]);
        host_store(k == 0 ? vl_host : vr_host, acc);
    }
    return ZK_OK;
}

// the power tables of x (x^j for j < 1024 and x^(1024 j), j < 2^(logn - 10)) built in the stream's scratch (poly_b) for one call: an
// evaluation / division point is a fresh transcript challenge, caching it beside the NTT twiddle tables would only evict those
template <class F>
int scratch_pow_tables(StreamScratch& ss, const Fe<F>& x, uint32_t logn, hipStream_t st, PowTables<F>* out) {
    if (logn < 1) logn = 1;
    const uint64_t nlo = logn >= 10 ? 1024 : (1ull << logn), nhi = logn > 10 ? (1ull << (logn - 10)) : 1;
    ZK_TRY(ws_get(ss.poly_b, sizeof(Fe<F>) * (nlo + nhi + 64)));
    Fe<F>* tbl = (Fe<F>*)ss.poly_b.p;
    Fe<F>* d_lad = tbl + nlo + nhi;
    ZK_LAUNCH((pow_ladder_kernel<F>), 1, 64, 0, st, x, d_lad);   // x^(2^k) and (x^1024)^(2^k), on the device: no host staging, no stall
    ZK_LAUNCH((pow_table_kernel<F>), (unsigned)((nlo + 255) / 256), 256, 0, st, tbl, (const Fe<F>*)d_lad, nlo, 10);
    ZK_LAUNCH((pow_table_kernel<F>), (unsigned)((nhi + 255) / 256), 256, 0, st, tbl + nlo, (const Fe<F>*)(d_lad + 32), nhi, 22);
    HIP_TRY(hipGetLastError());
    out->lo = tbl;
    out->hi = tbl + nlo;
    return ZK_OK;
}

template <class F>
int vec_fold_many_run(Fe<F>* out, const Fe<F>* first, int64_t stride, uint32_t count, uint64_t n, const Fe<F>& s, hipStream_t st) {
    if (n == 0 || count == 0) return ZK_OK;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    ZK_LAUNCH((vec_fold_many_kernel<F>), (unsigned)blocks, 256, 0, st, out, first, stride, count, n, s);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}
template <class F>
int ipa_fold_round_run(Fe<F>* p, Fe<F>* b, Fe<F>* W, uint64_t half, uint64_t m0, const Fe<F>& u, hipStream_t st) {
    if (half == 0) return ZK_OK;
    if (W && (m0 == 0 || (m0 & (m0 - 1)) || half >= m0 || (half & (half - 1)))) return ZK_ERR_INVALID_ARG;
    Fe<F> ui;
    fe_inv(ui, u);
    const uint64_t work = W && m0 > half ? m0 : half;
    uint64_t blocks = (work + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ZK_LAUNCH((ipa_fold_round_kernel<F>), (unsigned)blocks, 256, 0, st, p, b, W, half, W ? m0 : 0, ui, u);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}
template <class F>
int vec_powers_run(DeviceCtx& dc, Fe<F>* out, uint64_t n, const Fe<F>& x, hipStream_t st) {
    if (n == 0) return ZK_OK;
    if (n > (1ull << 30)) return ZK_ERR_UNSUPPORTED;
    uint32_t logn = 0;
    while ((1ull << logn) < n) logn++;
    StreamScratch* ss = nullptr;
    ZK_TRY(stream_scratch(dc, st, &ss));
    PowTables<F> pw;
    ZK_TRY(scratch_pow_tables<F>(*ss, x, logn, st, &pw));
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ZK_LAUNCH((vec_powers_kernel<F>), (unsigned)blocks, 256, 0, st, out, n, pw);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}

// q = (a - a(x)) / (X - x), n coefficients in, n out (q[n - 1] = 0); a == q allowed  (zk_poly_kernels.h, kate_*)
template <class F>
int kate_division_run(DeviceCtx& dc, const Fe<F>* a, Fe<F>* q, uint64_t n, const Fe<F>& x, hipStream_t st) {
    if (n == 0) return ZK_OK;
    if (n > (1ull << 30)) return ZK_ERR_UNSUPPORTED;
    StreamScratch* ss = nullptr;
    ZK_TRY(stream_scratch(dc, st, &ss));
    const uint64_t per_wg = (uint64_t)KATE_WG * KATE_K;
    const uint32_t nblocks = (uint32_t)((n + per_wg - 1) / per_wg);
    ZK_TRY(ws_get(ss->poly_tot, (size_t)nblocks * sizeof(Fe<F>)));
    Fe<F>* tot = (Fe<F>*)ss->poly_tot.p;
    KatePows<F> kp;
    Fe<F> w = x;
    for (uint32_t k = 1; k < KATE_K; k <<= 1) fe_sqr(w, w);        // x^16
    for (uint32_t s = 0; s <= KATE_LOG_WG; s++) {
        kp.xk[s] = w;
        fe_sqr(w, w);
    }
    ZK_LAUNCH((kate_block_kernel<F>), nblocks, KATE_WG, 0, st, a, q, tot, n, x, kp);
    if (nblocks > 1) {
        PowTables<F> pw;
        ZK_TRY(scratch_pow_tables<F>(*ss, x, 12, st, &pw));        // exponents below 4096
        ZK_LAUNCH((kate_totals_kernel<F>), 1, KATE_WG, 0, st, tot, nblocks, kp.xk[KATE_LOG_WG]);
        ZK_LAUNCH((kate_apply_kernel<F>), nblocks, KATE_WG, 0, st, q, (const Fe<F>*)tot, n, x, pw);
    }
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}

// p_q(x) for `count` resident coefficient vectors of n elements (stride apart), all at the same x (Montgomery form): one
// launch, one copy.  Synchronises the stream (the results are host values).
template <class F>
int poly_eval_run(DeviceCtx& dc, const Fe<F>* c, uint64_t n, uint32_t count, uint64_t stride, const Fe<F>& x, int field, void* out_host,
                  hipStream_t st) {
    if (count == 0) return ZK_OK;
    std::vector<Fe<F>> acc(count);
    for (auto& a : acc) fe_zero(a);
    if (n) {
        if (n > (1ull << 30) || count > 65535) return ZK_ERR_UNSUPPORTED;
        uint32_t logn = 0;
        while ((1ull << logn) < n) logn++;
        StreamScratch* ss = nullptr;
        ZK_TRY(stream_scratch(dc, st, &ss));
        (void)field;
        PowTables<F> pw;
        ZK_TRY(scratch_pow_tables<F>(*ss, x, logn, st, &pw));
        uint64_t blocks = ((n + EVAL_K - 1) / EVAL_K + 255) / 256;
        if (blocks > 256) blocks = 256;
        ZK_TRY(ws_get(ss->poly_tot, blocks * count * sizeof(Fe<F>)));
#if defined(ZK_EMU)
        for (uint32_t q = 0; q < count; q++)     // the test emulator launches one-dimensional grids
            ZK_LAUNCH((poly_eval_kernel<F>), (unsigned)blocks, 256, 0, st, c + (uint64_t)q * stride, n, (uint64_t)0, x, pw,
                      (Fe<F>*)ss->poly_tot.p + (uint64_t)q * blocks);
#else
        hipLaunchKernelGGL((poly_eval_kernel<F>), dim3((unsigned)blocks, count), dim3(256), 0, st, c, n, stride, x, pw, (Fe<F>*)ss->poly_tot.p);
#endif
        HIP_TRY(hipGetLastError());
        std::vector<Fe<F>> part(blocks * count);
        HIP_TRY(hipMemcpyAsync(part.data(), ss->poly_tot.p, blocks * count * sizeof(Fe<F>), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (uint32_t q = 0; q < count; q++)
            for (uint64_t b = 0; b < blocks; b++) fe_add(acc[q], acc[q], part[(uint64_t)q * blocks + b]);
    }
    for (uint32_t q = 0; q < count; q++) host_store((unsigned char*)out_host + (size_t)q * sizeof(Fe<F>), acc[q]);
    return ZK_OK;
}

template <class F>
int vec_muladd_run(Fe<F>* out, const Fe<F>* a, const Fe<F>* b, uint64_t n, const Fe<F>& s, hipStream_t st) {
    if (n == 0) return ZK_OK;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    ZK_LAUNCH((vec_muladd_kernel<F>), (unsigned)blocks, 256, 0, st, out, a, b, n, s);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}

template <class F>
int vec_fold_run(Fe<F>* a, uint64_t half, const Fe<F>& c, hipStream_t st) {
    if (half == 0) return ZK_OK;
    uint64_t blocks = (half + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ZK_LAUNCH((vec_fold_kernel<F>), (unsigned)blocks, 256, 0, st, a, half, c);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}

template <class F>
int ipa_virtual_scalars_run(const Fe<F>* p, const Fe<F>* W, Fe<F>* SL, Fe<F>* SR, uint64_t m0, uint64_t cur, hipStream_t st) {
    if (m0 == 0 || cur < 2 || cur > m0 || (cur & (cur - 1)) != 0 || (m0 & (m0 - 1)) != 0) return ZK_ERR_INVALID_ARG;
    uint64_t blocks = (m0 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ZK_LAUNCH((ipa_virtual_scalars_kernel<F>), (unsigned)blocks, 256, 0, st, p, W, SL, SR, m0, cur);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}
template <class F>
int ipa_update_weights_run(Fe<F>* W, uint64_t m0, uint64_t bit, const Fe<F>& u, hipStream_t st) {
    if (m0 == 0 || bit == 0 || bit >= m0 || (bit & (bit - 1)) != 0) return ZK_ERR_INVALID_ARG;
    uint64_t blocks = (m0 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ZK_LAUNCH((ipa_update_weights_kernel<F>), (unsigned)blocks, 256, 0, st, W, m0, bit, u);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}

// program / column table / constants are copied to the stream's scratch, then one grid-stride launch
template <class F>
int expr_eval_run(DeviceCtx& dc, const zk_expr_op* prog, uint32_t n_ops, const void* const* cols, uint32_t n_cols, const Fe<F>* consts,
                  uint32_t n_consts, uint32_t log_n, uint32_t rot_scale, Fe<F>* out, hipStream_t st) {
    if (n_ops == 0 || n_ops > EXPR_MAX_OPS || n_cols > EXPR_MAX_COLS || n_consts > EXPR_MAX_CONSTS || log_n > 30) return ZK_ERR_INVALID_ARG;
    static_assert(sizeof(zk_expr_op) == sizeof(ExprOp), "ABI struct = kernel struct");
    // validate the program on the host: stack depth, operand indices -- a kernel must never index past its tables
    int depth = 0;
    for (uint32_t k = 0; k < n_ops; k++) {
        const zk_expr_op& o = prog[k];
        if (o.op > 6) return ZK_ERR_INVALID_ARG;
        if (o.op == 0 && (o.arg >= n_cols || !cols[o.arg])) return ZK_ERR_INVALID_ARG;
        if ((o.op == 1 || o.op == 6) && o.arg >= n_consts) return ZK_ERR_INVALID_ARG;
        if (o.op <= 1) depth++;
        else if (o.op == 5 || o.op == 6) { if (depth < 1) return ZK_ERR_INVALID_ARG; }
        else { if (depth < 2) return ZK_ERR_INVALID_ARG; depth--; }
        if (depth > (int)EXPR_STACK) return ZK_ERR_UNSUPPORTED;
    }
    if (depth != 1) return ZK_ERR_INVALID_ARG;
    StreamScratch* ss = nullptr;
    ZK_TRY(stream_scratch(dc, st, &ss));
    const size_t pb = sizeof(ExprOp) * EXPR_MAX_OPS, cb = sizeof(void*) * EXPR_MAX_COLS, kb = sizeof(Fe<F>) * EXPR_MAX_CONSTS;
    ZK_TRY(ws_get(ss->poly_tot, pb + cb + kb + 64));
    unsigned char* base = (unsigned char*)ss->poly_tot.p;
    // one aligned 64-bit word per op for the kernel: op | rot << 16 | arg << 32
    std::vector<uint64_t> words(n_ops);
    for (uint32_t k = 0; k < n_ops; k++)
        words[k] = (uint64_t)prog[k].op | ((uint64_t)(uint16_t)prog[k].rot << 16) | ((uint64_t)prog[k].arg << 32);
    HIP_TRY(hipMemcpyAsync(base, words.data(), sizeof(uint64_t) * n_ops, hipMemcpyHostToDevice, st));
    if (n_cols) HIP_TRY(hipMemcpyAsync(base + pb, cols, sizeof(void*) * n_cols, hipMemcpyHostToDevice, st));
    if (n_consts) HIP_TRY(hipMemcpyAsync(base + pb + cb, consts, sizeof(Fe<F>) * n_consts, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));   // the sources are the caller's host memory
    const uint64_t n = 1ull << log_n;
    uint64_t blocks = (n + EXPR_WG - 1) / EXPR_WG;
    if (blocks > 8192) blocks = 8192;
    ZK_LAUNCH((expr_eval_kernel<F>), (unsigned)blocks, EXPR_WG, 0, st, (const uint64_t*)base, n_ops, (const Fe<F>* const*)(base + pb), n_cols,
              (const Fe<F>*)(base + pb + cb), n_consts, log_n, rot_scale, out);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}

// ---- zk_expr_eval_lazy_device: the host side of expr_eval29_kernel ----
// Walks the caller's program once with a (limb bound, value bound) pair per stack slot -- the discipline of zk_field29.h -- and
// emits the kernel's program: the same operations plus, where a bound would be violated, a carry step (EXPR29_NORM) or a
// contraction (EXPR29_REFRESH) on the top of the stack, and the bias table id of every subtraction / negation.
//   loads            strict limbs, value < 2p (columns and constants are stored canonical; 2 leaves room for a lazily reduced producer)
//   product a b      needs 9 eff(a) eff(b) + 9 2^58 + 2^36 < 2^64 with eff = max(limb bound, top limb of the value bound);
//                    result strict, value < vb(a) vb(b) / ratio + 1, ratio = a lower bound of R' / p
//   a + b            bounds add (limbs must stay below 2^32)
//   a - b + BIAS     needs lb(b) <= k (2^29 - 1) and vb(b) + 2 <= M for the table (M p, k); limbs + (k + 1) 2^29, value + M
//   a slot pushed to LDS is always normalised (<= 2^29 + 6) with value <= 32 p, so only the top of the stack ever needs fixing
template <class F>
int expr_compile29(const zk_expr_op* prog, uint32_t n_ops, uint32_t n_cols, const void* const* cols, uint32_t n_consts, std::vector<uint64_t>& words,
                   uint32_t& depth_out, uint32_t nslots = EXPR29_SLOTS) {
    using K = F29<F>;
    const double W29 = (double)(1u << K::W), STRICT = W29 - 1, NP = W29 + 6, U32 = 4294967296.0, U64 = 18446744073709551616.0;
    const double ptop1 = (double)K::P[K::L - 1] + 1;
    const double ratio = W29 / ptop1;                          // R' / p >= 2^29 / (top limb of p + 1)
    struct B {
        double lb, vb;
    };
    auto eff = [&](const B& b) { return b.lb > b.vb * ptop1 ? b.lb : b.vb * ptop1; };
    auto mul_ok = [&](const B& a, const B& b) { return 9.0 * eff(a) * eff(b) + 9.0 * W29 * W29 + 68719476736.0 < U64 * 0.999; };
    std::vector<B> st;
    words.clear();
    auto emit = [&](uint32_t op, int rot, uint32_t arg) { words.push_back((uint64_t)op | ((uint64_t)(uint16_t)rot << 16) | ((uint64_t)arg << 32)); };
    // column-value slots: next_use[k] = the next COL op reading the same (column, rotation), the eviction key (farthest first)
    const uint32_t NEVER = 0xffffffffu;
    std::vector<uint32_t> next_use(n_ops, NEVER);
    {
        std::map<std::pair<uint32_t, int>, uint32_t> last;
        for (uint32_t k = n_ops; k-- > 0;)
            if (prog[k].op == 0) {
                const auto key = std::make_pair(prog[k].arg, (int)prog[k].rot);
                auto it = last.find(key);
                if (it != last.end()) next_use[k] = it->second;
                last[key] = k;
            }
    }
    struct Slot {
        bool used = false;
        uint32_t col = 0;
        int rot = 0;
        uint32_t next = 0;
    } slots[EXPR_JIT_SLOTS_MAX];
    if (nslots < 1 || nslots > EXPR_JIT_SLOTS_MAX) return ZK_ERR_INVALID_ARG;
    const uint32_t hoist = 12;
    auto norm_top = [&]() {
        emit(EXPR29_NORM, 0, 0);
        st.back().lb = NP;
    };
    auto refresh_top = [&]() {
        if (st.back().lb > NP) norm_top();
        emit(EXPR29_REFRESH, 0, 0);
        st.back().vb = st.back().vb / ratio + 1;
        st.back().lb = STRICT;
    };
    const B ONE_B{STRICT, 1}, LOAD_B{STRICT, 2};
    struct Bias {
        double M, k;
    };
    const Bias biases[5] = {{4, 1}, {4, 2}, {8, 2}, {8, 3}, {16, 2}};     // ids = the kernel's fe29_sub_by_id
    auto pick_bias = [&](const B& t) -> int {
        int best = -1;
        for (int id = 0; id < 5; id++)
            if (t.lb <= biases[id].k * STRICT && t.vb + 2 <= biases[id].M) {
                if (best < 0 || biases[id].M < biases[best].M || (biases[id].M == biases[best].M && biases[id].k < biases[best].k)) best = id;
            }
        return best;
    };
    auto bias_for_top = [&]() -> int {       // make the top of the stack subtractable, return its table
        int id = pick_bias(st.back());
        if (id < 0 && st.back().lb > NP) {
            norm_top();
            id = pick_bias(st.back());
        }
        if (id < 0) {
            refresh_top();
            id = pick_bias(st.back());
        }
        return id;
    };
    uint32_t depth = 0;
    for (uint32_t k = 0; k < n_ops; k++) {
        const zk_expr_op& o = prog[k];
        if (o.op > 6) return ZK_ERR_INVALID_ARG;
        if (o.op == 0 && (o.arg >= n_cols || !cols[o.arg])) return ZK_ERR_INVALID_ARG;
        if ((o.op == 1 || o.op == 6) && o.arg >= n_consts) return ZK_ERR_INVALID_ARG;
        if (o.op <= 1) {
            if (!st.empty()) {                 // the current top goes to LDS: normalised, value <= 32 p
                if (st.back().vb > 32) refresh_top();
                if (st.back().lb > NP) norm_top();
            }
            if (o.op == 0) {                   // the cell from its slot; a miss loads it first (hoisted below)
                int sl = -1;
                for (uint32_t q = 0; q < nslots; q++)
                    if (slots[q].used && slots[q].col == o.arg && slots[q].rot == (int)o.rot) sl = (int)q;
                if (sl < 0) {
                    for (uint32_t q = 0; q < nslots && sl < 0; q++)
                        if (!slots[q].used) sl = (int)q;
                    if (sl < 0) {
                        sl = 0;
                        for (uint32_t q = 1; q < nslots; q++)
                            if (slots[q].next > slots[sl].next) sl = (int)q;
                    }
                    emit(EXPR29_LOAD, o.rot, o.arg | ((uint32_t)sl << 16));
                    slots[sl].used = true;
                    slots[sl].col = o.arg;
                    slots[sl].rot = o.rot;
                }
                slots[sl].next = next_use[k];
                emit(0, 0, (uint32_t)sl);
            } else {
                emit(1, 0, o.arg);
            }
            st.push_back(LOAD_B);
            if (st.size() > EXPR_STACK) return ZK_ERR_UNSUPPORTED;
            if (st.size() - 1 > depth) depth = (uint32_t)st.size() - 1;
        } else if (o.op == 5) {                // neg: 0 - t + BIAS
            if (st.empty()) return ZK_ERR_INVALID_ARG;
            const int id = bias_for_top();
            if (id < 0) return ZK_ERR_UNSUPPORTED;
            emit(5, 0, (uint32_t)id);
            st.back() = B{(biases[id].k + 1) * W29, biases[id].M};
        } else if (o.op == 6) {                // scale by a constant
            if (st.empty()) return ZK_ERR_INVALID_ARG;
            if (!mul_ok(st.back(), ONE_B) && st.back().lb > NP) norm_top();
            if (!mul_ok(st.back(), ONE_B)) refresh_top();
            if (!mul_ok(st.back(), ONE_B)) return ZK_ERR_UNSUPPORTED;
            emit(6, 0, o.arg);
            st.back() = B{STRICT, st.back().vb * 2 / ratio + 1};
        } else {
            if (st.size() < 2) return ZK_ERR_INVALID_ARG;
            B x = st[st.size() - 2];
            if (o.op == 2) {
                if (x.lb + st.back().lb >= U32 * 0.99) norm_top();
                if (x.vb + st.back().vb > 256) refresh_top();
                emit(2, 0, 0);
                x = B{x.lb + st.back().lb, x.vb + st.back().vb};
            } else if (o.op == 3) {
                const int id = bias_for_top();
                if (id < 0) return ZK_ERR_UNSUPPORTED;
                emit(3, 0, (uint32_t)id);
                x = B{x.lb + (biases[id].k + 1) * W29, x.vb + biases[id].M};
                if (x.lb >= U32 * 0.99) return ZK_ERR_UNSUPPORTED;   // (cannot happen: x was pushed normalised)
            } else {
                if (!mul_ok(x, st.back()) && st.back().lb > NP) norm_top();
                if (!mul_ok(x, st.back())) refresh_top();
                if (!mul_ok(x, st.back())) return ZK_ERR_UNSUPPORTED;
                emit(4, 0, 0);
                x = B{STRICT, x.vb * st.back().vb / ratio + 1};
            }
            st.pop_back();
            st.back() = x;
        }
    }
    if (st.size() != 1) return ZK_ERR_INVALID_ARG;
    if (st.back().vb > 64) refresh_top();      // fe29_to_std: norm, product by R mod p, canonical (value < 20 p after the product)
    if (words.size() > 2 * EXPR_MAX_OPS) return ZK_ERR_UNSUPPORTED;
    // hoist every load up to 12 operations ahead of its first use (never across another operation on the same slot): a load has
    // no effect on the stack, and issued early its latency is spent under the products in between
    for (size_t q = 0; q < words.size(); q++) {
        const uint64_t ld = words[q];
        if ((ld & 0xff) != EXPR29_LOAD) continue;
        const uint32_t slot = (uint32_t)(ld >> 48);
        size_t at = q;
        while (at > 0 && q - at < hoist) {
            const uint64_t prev = words[at - 1];
            const uint32_t pop = (uint32_t)(prev & 0xff);
            if ((pop == EXPR29_LOAD && (uint32_t)(prev >> 48) == slot) || (pop == 0 && (uint32_t)(prev >> 32) == slot)) break;
            words[at] = prev;
            at--;
        }
        words[at] = ld;
    }
    depth_out = depth;
    return ZK_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// The quotient numerator SPECIALISED PER PROGRAM.  expr_eval29_kernel interprets the bound-annotated program: 38 500 VALU
// instructions per row for the bench's 268-op / 81-product program, of which the products are 17 000 -- the rest is the stack
// machine (every push spills the top of the stack to LDS and unpacks a cell: ~110 instructions, 122 pushes per row).  A gate
// expression is fixed per proving key, so the same annotated program -- same operations, same carry steps, same bias tables: the
// bound walk of expr_compile29 stays the one authority -- is written out as straight-line HIP with the stack resolved at
// generation time (every intermediate a named value, no LDS), compiled once with hiprtc and cached: ~20 000 instructions per
// row.  Not available in the CPU test emulator (the interpreter runs there); any failure to build falls back to the interpreter.
#if !defined(ZK_EMU) && defined(ZK_FIELD)
}  // namespace zk
#include <hip/hiprtc.h>
#include <dlfcn.h>
namespace zk {
#define ZK_STR2_(x) #x
#define ZK_STR_(x) ZK_STR2_(x)
struct ExprJitKernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    bool failed = false;
};

template <class F>
std::string expr_jit_source(const std::vector<uint64_t>& words, uint32_t nslots, int waves) {
    static const char* const bias_names[5] = {"BIAS4K1", "BIAS4K2", "BIAS8K2", "BIAS8K3", "BIAS16K2"};   // ids of fe29_sub_by_id
    std::string s;
    s.reserve(words.size() * 96 + 1024);
    char buf[256];
    s += "#include \"zk_field29.h\"\nusing namespace zk;\nusing F = " ZK_STR_(ZK_FIELD) ";\nusing K = F29<F>;\n";
    snprintf(buf, sizeof buf, "extern \"C\" __global__ void __launch_bounds__(64, %d) zk_expr_jit(const Fe<F>* const* __restrict__ cols, "
                              "const Fe<F>* __restrict__ consts, unsigned int log_n, unsigned int rot_scale, Fe<F>* __restrict__ out) {\n", waves);
    s += buf;
    s += "  const unsigned long long n = 1ull << log_n, mask = n - 1;\n"
         "  for (unsigned long long i = (unsigned long long)blockIdx.x * 64 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 64) {\n"
         "    Fe<F> ";
    for (uint32_t q = 0; q < nslots; q++) {
        snprintf(buf, sizeof buf, "%ss%u", q ? ", " : "", q);
        s += buf;
    }
    s += ";\n";
    std::vector<int> st;
    int nt = 0;
    for (uint64_t w : words) {
        const uint32_t op = (uint32_t)(w & 0xff), arg = (uint32_t)(w >> 32);
        const int rot = (int)(int16_t)(uint16_t)(w >> 16);
        if (op == EXPR29_LOAD) {
            snprintf(buf, sizeof buf, "    s%u = cols[%u][(i + (unsigned long long)((long long)%d * (long long)rot_scale)) & mask];\n", (arg >> 16) % nslots, arg & 0xffffu, rot);
            s += buf;
        } else if (op == 0 || op == 1) {
            if (op == 0)
                snprintf(buf, sizeof buf, "    Fe29<F> t%d; fe29_unpack(t%d, s%u);\n", nt, nt, arg % nslots);
            else
                snprintf(buf, sizeof buf, "    Fe29<F> t%d; fe29_unpack(t%d, consts[%u]);\n", nt, nt, arg);
            s += buf;
            st.push_back(nt++);
        } else if (op == 5) {
            if (st.empty()) return std::string();
            snprintf(buf, sizeof buf, "    Fe29<F> t%d, z%d; fe29_zero(z%d); fe29_sub(t%d, z%d, t%d, K::%s);\n", nt, nt, nt, nt, nt, st.back(), bias_names[arg < 5 ? arg : 4]);
            s += buf;
            st.back() = nt++;
        } else if (op == 6) {
            if (st.empty()) return std::string();
            snprintf(buf, sizeof buf, "    Fe29<F> t%d, c%d; fe29_unpack(c%d, consts[%u]); fe29_mul(t%d, t%d, c%d);\n", nt, nt, nt, arg, nt, st.back(), nt);
            s += buf;
            st.back() = nt++;
        } else if (op == EXPR29_NORM) {
            if (st.empty()) return std::string();
            snprintf(buf, sizeof buf, "    fe29_norm(t%d, t%d);\n", st.back(), st.back());
            s += buf;
        } else if (op == EXPR29_REFRESH) {
            if (st.empty()) return std::string();
            snprintf(buf, sizeof buf, "    { Fe29<F> o; fe29_one(o); fe29_mul(t%d, t%d, o); }\n", st.back(), st.back());
            s += buf;
        } else if (op >= 2 && op <= 4) {
            if (st.size() < 2) return std::string();
            const int b = st.back(), a = st[st.size() - 2];
            if (op == 2)
                snprintf(buf, sizeof buf, "    Fe29<F> t%d; fe29_add(t%d, t%d, t%d);\n", nt, nt, a, b);
            else if (op == 3)
                snprintf(buf, sizeof buf, "    Fe29<F> t%d; fe29_sub(t%d, t%d, t%d, K::%s);\n", nt, nt, a, b, bias_names[arg < 5 ? arg : 4]);
            else
                snprintf(buf, sizeof buf, "    Fe29<F> t%d; fe29_mul(t%d, t%d, t%d);\n", nt, nt, a, b);
            s += buf;
            st.pop_back();
            st.back() = nt++;
        } else {
            return std::string();
        }
    }
    if (st.size() != 1) return std::string();
    snprintf(buf, sizeof buf, "    Fe<F> r; fe29_to_std(r, t%d); out[i] = r;\n  }\n}\n", st.back());
    s += buf;
    return s;
}

// where the headers the generated source includes live: next to the library (<dir of libzkcp_amd.so>/csrc), or ZKCP_AMD_CSRC
inline std::string expr_jit_include_dir() {
    if (const char* e = getenv("ZKCP_AMD_CSRC")) return e;
    Dl_info info;
    if (dladdr((const void*)&expr_jit_include_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t k = p.find_last_of('/');
        return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/csrc";
    }
    return "csrc";
}

// the compiled kernel of (source, device), built on first use; nullptr when it cannot be built (the caller interprets instead)
template <class F>
ExprJitKernel* expr_jit_get(DeviceCtx& dc, const std::string& src) {
    static std::map<std::pair<int, std::string>, ExprJitKernel> cache;      // one cache per field unit
    static std::mutex cache_mu;                                             // (dc.mu is per device: two devices' callers may meet here)
    std::lock_guard<std::mutex> lk(cache_mu);
    auto key = std::make_pair(dc.device, src);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second.failed ? nullptr : &it->second;
    ExprJitKernel& k = cache[key];
    k.failed = true;
    const bool verbose = getenv("ZK_EXPR_STATS") != nullptr;
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "zk_expr_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return nullptr;
    const std::string inc = "-I" + expr_jit_include_dir();
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffreestanding", inc.c_str()};
    const hiprtcResult rc = hiprtcCompileProgram(prog, 5, opts);
    if (rc != HIPRTC_SUCCESS) {
        if (verbose) {
            size_t ls = 0;
            hiprtcGetProgramLogSize(prog, &ls);
            std::vector<char> log(ls + 1, 0);
            if (ls) hiprtcGetProgramLog(prog, log.data());
            fprintf(stderr, "expr jit: compilation failed (%s): %.1500s\n", hiprtcGetErrorString(rc), log.data());
        }
        hiprtcDestroyProgram(&prog);
        return nullptr;
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    const bool got = cs && hiprtcGetCode(prog, code.data()) == HIPRTC_SUCCESS;
    hiprtcDestroyProgram(&prog);
    if (!got || hipModuleLoadData(&k.mod, code.data()) != hipSuccess) return nullptr;
    if (hipModuleGetFunction(&k.fn, k.mod, "zk_expr_jit") != hipSuccess) return nullptr;
    if (verbose) fprintf(stderr, "expr jit: %zu bytes of source -> %zu bytes of code object\n", src.size(), cs);
    k.failed = false;
    return &k;
}
#endif

// the source of the specialised kernel for a program, without a device (zk_expr_specialised_source: diagnostics, the CPU test tier)
template <class F>
int expr_source_run(const zk_expr_op* prog, uint32_t n_ops, uint32_t n_cols, uint32_t n_consts, std::string& out) {
#if !defined(ZK_EMU) && defined(ZK_FIELD)
    if (n_ops == 0 || n_ops > EXPR_MAX_OPS || n_cols > EXPR_MAX_COLS || n_consts > EXPR_MAX_CONSTS) return ZK_ERR_INVALID_ARG;
    std::vector<const void*> cols(n_cols ? n_cols : 1, (const void*)&out);      // (the walk only asks that a column be present)
    std::vector<uint64_t> words;
    uint32_t depth = 0;
    ZK_TRY(expr_compile29<F>(prog, n_ops, n_cols, cols.data(), n_consts, words, depth, EXPR_JIT_SLOTS));
    out = expr_jit_source<F>(words, EXPR_JIT_SLOTS, 2);
    return out.empty() ? ZK_ERR_INVALID_ARG : ZK_OK;
#else
    (void)prog, (void)n_ops, (void)n_cols, (void)n_consts, (void)out;
    return ZK_ERR_UNSUPPORTED;
#endif
}

// columns: x R' mod p (R' = 2^261), canonical words; constants: standard Montgomery on the host (converted here); out: standard
template <class F>
int expr_eval_lazy_run(DeviceCtx& dc, const zk_expr_op* prog, uint32_t n_ops, const void* const* cols, uint32_t n_cols, const Fe<F>* consts,
                       uint32_t n_consts, uint32_t log_n, uint32_t rot_scale, Fe<F>* out, hipStream_t st) {
    if (n_ops == 0 || n_ops > EXPR_MAX_OPS || n_cols > EXPR_MAX_COLS || n_consts > EXPR_MAX_CONSTS || log_n > 30) return ZK_ERR_INVALID_ARG;
    std::vector<uint64_t> words;
    uint32_t depth = 0;
    // specialised kernel (g.expr_jit: 0 = for evaluations of 2^16 rows and more, 1 = always, 2 = never): the same annotated program with
    // more column slots (they are named values there, not a select network)
    bool want_jit = false;
#if !defined(ZK_EMU) && defined(ZK_FIELD)
    want_jit = g.expr_jit == 1 || (g.expr_jit == 0 && log_n >= 16);
#endif
    uint32_t nslots = want_jit ? EXPR_JIT_SLOTS : EXPR29_SLOTS;
    if (want_jit)
        if (const char* e = getenv("ZK_EXPR_JIT_SLOTS")) nslots = atoi(e) >= 1 && atoi(e) <= (int)EXPR_JIT_SLOTS_MAX ? (uint32_t)atoi(e) : nslots;   // (tuning)
    ZK_TRY(expr_compile29<F>(prog, n_ops, n_cols, cols, n_consts, words, depth, nslots));
    if (getenv("ZK_EXPR_STATS")) {     // (diagnostic) the executed program by opcode: what the bound walk added to the caller's ops
        uint32_t cnt[16] = {0};
        for (uint64_t w : words) cnt[w & 15]++;
        fprintf(stderr, "expr29: %u caller ops -> %zu executed: push col %u const %u add %u sub %u mul %u neg %u scale %u norm %u refresh %u load %u, depth %u\n",
                n_ops, words.size(), cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5], cnt[6], cnt[EXPR29_NORM], cnt[EXPR29_REFRESH], cnt[EXPR29_LOAD], depth);
    }
    std::vector<Fe<F>> c29(n_consts ? n_consts : 1);
    constexpr int SH = F29<F>::W * F29<F>::L - 32 * F::N;      // x R -> x R': times 2^5
    for (uint32_t i = 0; i < n_consts; i++) {
        c29[i] = consts[i];
        for (int k = 0; k < SH; k++) fe_dbl(c29[i], c29[i]);
    }
    StreamScratch* ss = nullptr;
    ZK_TRY(stream_scratch(dc, st, &ss));
    const size_t pb = sizeof(uint64_t) * 2 * EXPR_MAX_OPS, cb = sizeof(void*) * EXPR_MAX_COLS, kb = sizeof(Fe<F>) * EXPR_MAX_CONSTS;
    ZK_TRY(ws_get(ss->poly_tot, pb + cb + kb + 64));
    unsigned char* base = (unsigned char*)ss->poly_tot.p;
    HIP_TRY(hipMemcpyAsync(base, words.data(), sizeof(uint64_t) * words.size(), hipMemcpyHostToDevice, st));
    if (n_cols) HIP_TRY(hipMemcpyAsync(base + pb, cols, sizeof(void*) * n_cols, hipMemcpyHostToDevice, st));
    if (n_consts) HIP_TRY(hipMemcpyAsync(base + pb + cb, c29.data(), sizeof(Fe<F>) * n_consts, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));   // the sources are host temporaries
    const uint64_t n = 1ull << log_n;
#if !defined(ZK_EMU) && defined(ZK_FIELD)
    if (want_jit) {
        int waves = 2;
        if (const char* e = getenv("ZK_EXPR_JIT_WAVES")) waves = atoi(e) >= 1 && atoi(e) <= 8 ? atoi(e) : 2;
        const std::string src = expr_jit_source<F>(words, nslots, waves);
        ExprJitKernel* jk = src.empty() ? nullptr : expr_jit_get<F>(dc, src);
        if (jk) {
            const void* a_cols = base + pb;
            const void* a_consts = base + pb + cb;
            unsigned a_log = log_n, a_rs = rot_scale;
            void* a_out = out;
            void* args[] = {&a_cols, &a_consts, &a_log, &a_rs, &a_out};
            uint64_t jb = (n + 63) / 64;
            const uint64_t cap = (uint64_t)(dc.num_cus > 0 ? dc.num_cus : 256) * 4 * (uint64_t)waves * 4;
            if (jb > cap) jb = cap;
            HIP_TRY(hipModuleLaunchKernel(jk->fn, (unsigned)jb, 1, 1, 64, 1, 1, 0, st, args, nullptr));
            return ZK_OK;
        }
        // could not be built: the interpreter needs the program for ITS slot count
        ZK_TRY(expr_compile29<F>(prog, n_ops, n_cols, cols, n_consts, words, depth, EXPR29_SLOTS));
        HIP_TRY(hipMemcpyAsync(base, words.data(), sizeof(uint64_t) * words.size(), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
#endif
    uint64_t blocks = (n + EXPR29_WG - 1) / EXPR29_WG;
    if (blocks > 16384) blocks = 16384;
    const size_t shmem = (size_t)(depth ? depth : 1) * F29<F>::L * EXPR29_WG * sizeof(uint32_t);
    ZK_LAUNCH((expr_eval29_kernel<F>), (unsigned)blocks, EXPR29_WG, shmem, st, (const uint64_t*)base, (uint32_t)words.size(),
              (const Fe<F>* const*)(base + pb), n_cols, (const Fe<F>*)(base + pb + cb), n_consts, log_n, rot_scale, depth, out);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
