// NTT launch sequence (plan -> passes).  Included by zk_ntt_inst.cc, once per scalar field.
#pragma once
#include "zk_internal.h"
#include "zk_ntt_decl.h"
#include "zk_ntt_kernels.h"
#include "zk_ntt29_kernels.h"
#include "zk_r1cs_kernels.h"
namespace zk {
// ------------------------------------------------------------------ NTT
struct NttPlan {
    int nd;
    int rd[4];
    int log_t[4];
};

// radix split: <= 10 bits per pass (R*T*32 B of LDS: 1024 x 2 = 64 KiB, 512 x 4 = 64 KiB)
inline NttPlan ntt_plan(uint32_t logn) {
    NttPlan p;
    memset(&p, 0, sizeof p);
    int max_r = 10;
    if (g.ntt_opts.max_log_radix >= 1 && g.ntt_opts.max_log_radix <= 10) max_r = g.ntt_opts.max_log_radix;
    int nd = (int)((logn + max_r - 1) / max_r);
    if (nd < 1) nd = 1;
    p.nd = nd;
    int rem = (int)logn;
    for (int i = 0; i < nd; i++) {
        int r = (rem + (nd - i) - 1) / (nd - i);
        p.rd[i] = r;
        rem -= r;
    }
    int want_t = 2;
    if (g.ntt_opts.log_tile_plus1 >= 1 && g.ntt_opts.log_tile_plus1 <= 5) want_t = g.ntt_opts.log_tile_plus1 - 1;
    int log_m = 0;
    for (int i = 0; i < nd; i++) {
        int lt = want_t;
        if (p.rd[i] + lt > 11) lt = 11 - p.rd[i];  // <= 2048 elements = 64 KiB per tile
        if (i < nd - 1) {
            int log_s = (int)logn - log_m - p.rd[i];
            if (lt > log_s) lt = log_s;
        } else {
            if (nd == 1) lt = 0;
            else if (lt > p.rd[0]) lt = p.rd[0];
        }
        if (lt < 0) lt = 0;
        p.log_t[i] = lt;
        log_m += p.rd[i];
    }
    return p;
}

template <class F>
int tw_table(DeviceCtx& dc, const Fe<F>& omega, uint32_t logn, int field, hipStream_t st, const Fe<F>** out, bool r29 = false) {
    TwKey key;
    memset(&key, 0, sizeof key);
    key.field = field | (r29 ? 0x200 : 0);   // the lazy-limb pass reads omega^i R' mod p, the saturated one omega^i R mod p
    key.logn = logn;
    memcpy(key.omega, omega.v, sizeof(uint32_t) * F::N);
    auto it = dc.tw.find(key);
    if (it != dc.tw.end()) {
        it->second.stamp = ++dc.tw_stamp;
        *out = (const Fe<F>*)it->second.dev;
        return ZK_OK;
    }
    // evict least-recently-used tables beyond 8 entries / 2 GiB
    while (dc.tw.size() >= 8 || dc.tw_bytes > (2ull << 30)) {
        auto victim = dc.tw.begin();
        for (auto i2 = dc.tw.begin(); i2 != dc.tw.end(); ++i2)
            if (i2->second.stamp < victim->second.stamp) victim = i2;
        HIP_TRY(hipStreamSynchronize(st));
        hipFree(victim->second.dev);
        dc.tw_bytes -= victim->second.bytes;
        dc.tw.erase(victim);
    }
    const uint64_t count = logn > 0 ? (1ull << (logn - 1)) : 1;
    const int nbits = logn > 0 ? (int)logn - 1 : 0;
    // tbl[k] = omega^(2^k)
    std::vector<Fe<F>> tbl((size_t)(nbits > 0 ? nbits : 1));
    Fe<F> w = omega;
    for (int k = 0; k < nbits; k++) {
        tbl[k] = w;
        fe_sqr(w, w);
    }
    ZK_TRY(ws_get(dc.pow_tbl, sizeof(Fe<F>) * 64));
    HIP_TRY(hipMemcpyAsync(dc.pow_tbl.p, tbl.data(), sizeof(Fe<F>) * tbl.size(), hipMemcpyHostToDevice, st));
    void* dev = nullptr;
    const uint32_t n_inner = logn > 0 ? 1u << ((logn < (uint32_t)NTT_INNER_LOG ? logn : (uint32_t)NTT_INNER_LOG) - 1) : 1u;
    HIP_TRY(hipMalloc(&dev, sizeof(Fe<F>) * count + (r29 ? sizeof(InnerTw) * n_inner : 0)));
    const unsigned blk = 256;
    ZK_LAUNCH((pow_table_kernel<F>), (unsigned)((count + blk - 1) / blk), blk, 0, st, (Fe<F>*)dev, (const Fe<F>*)dc.pow_tbl.p,
              count, nbits);
    if (r29) {
        ZK_LAUNCH((table_to_r29_kernel<F>), (unsigned)((count + blk - 1) / blk), blk, 0, st, (Fe<F>*)dev, count);
        // the compact unpacked table of the in-tile twiddles, behind the big one
        const int shift = logn > (uint32_t)NTT_INNER_LOG ? (int)logn - NTT_INNER_LOG : 0;
        ZK_LAUNCH((inner_table_kernel<F>), (n_inner + blk - 1) / blk, blk, 0, st, (const Fe<F>*)dev, (InnerTw*)((Fe<F>*)dev + count), n_inner, shift);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));  // pow_tbl is reused by the next table build
    TwEntry e{dev, sizeof(Fe<F>) * count + (r29 ? sizeof(InnerTw) * n_inner : 0), ++dc.tw_stamp};
    dc.tw[key] = e;
    dc.tw_bytes += e.bytes;
    *out = (const Fe<F>*)dev;
    return ZK_OK;
}

// device tables lo[j] = g^j (j < min(n,1024)), hi[j] = g^(1024 j) (j < max(1, n/1024)) for on-the-fly coset powers; cached
template <class F>
int pow_tables(DeviceCtx& dc, const Fe<F>& gshift, uint32_t logn, int field, hipStream_t st, PowTables<F>* out, bool r29 = false) {
    TwKey key;
    memset(&key, 0, sizeof key);
    key.field = field | (r29 ? 0x300 : 0x100);   // separate key spaces from the twiddle tables; 0x300: entries in R' form
    key.logn = logn;
    memcpy(key.omega, gshift.v, sizeof(uint32_t) * F::N);
    auto it = dc.tw.find(key);
    const uint64_t nlo = logn >= 10 ? 1024 : (1ull << logn), nhi = logn > 10 ? (1ull << (logn - 10)) : 1;
    if (it == dc.tw.end()) {
        while (dc.tw.size() >= 16 || dc.tw_bytes > (2ull << 30)) {
            auto victim = dc.tw.begin();
            for (auto i2 = dc.tw.begin(); i2 != dc.tw.end(); ++i2)
                if (i2->second.stamp < victim->second.stamp) victim = i2;
            HIP_TRY(hipStreamSynchronize(st));
            hipFree(victim->second.dev);
            dc.tw_bytes -= victim->second.bytes;
            dc.tw.erase(victim);
        }
        // 2^k power ladders of g (10 entries) and of g^1024 (logn - 10 entries), then two table kernels
        std::vector<Fe<F>> lad(64);
        Fe<F> w = gshift;
        for (int k = 0; k < 10; k++) {
            lad[k] = w;
            fe_sqr(w, w);
        }
        for (int k = 0; k < 22; k++) {   // w = g^1024 here
            lad[32 + k] = w;
            fe_sqr(w, w);
        }
        void* dev = nullptr;
        const size_t bytes = sizeof(Fe<F>) * (nlo + nhi + 64);
        HIP_TRY(hipMalloc(&dev, bytes));
        Fe<F>* d_lad = (Fe<F>*)dev + nlo + nhi;
        HIP_TRY(hipMemcpyAsync(d_lad, lad.data(), sizeof(Fe<F>) * 64, hipMemcpyHostToDevice, st));
        ZK_LAUNCH((pow_table_kernel<F>), (unsigned)((nlo + 255) / 256), 256, 0, st, (Fe<F>*)dev, (const Fe<F>*)d_lad, nlo, 10);
        ZK_LAUNCH((pow_table_kernel<F>), (unsigned)((nhi + 255) / 256), 256, 0, st, (Fe<F>*)dev + nlo, (const Fe<F>*)(d_lad + 32), nhi, 22);
        if (r29) ZK_LAUNCH((table_to_r29_kernel<F>), (unsigned)((nlo + nhi + 255) / 256), 256, 0, st, (Fe<F>*)dev, nlo + nhi);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));   // `lad` is a host temporary
        TwEntry e{dev, bytes, ++dc.tw_stamp};
        dc.tw[key] = e;
        dc.tw_bytes += bytes;
        it = dc.tw.find(key);
    }
    it->second.stamp = ++dc.tw_stamp;
    out->lo = (const Fe<F>*)it->second.dev;
    out->hi = (const Fe<F>*)it->second.dev + nlo;
    return ZK_OK;
}

// size-2^logn DFT of `a` with root omega; optionally fused with a[i] *= g_pre^i before and a[k] *= g_post^k after
template <class F>
int ntt_run(DeviceCtx& dc, int field, Fe<F>* a, uint32_t logn, const Fe<F>& omega, int scale_flag, hipStream_t st, const Fe<F>* g_pre,
            const Fe<F>* g_post, uint32_t in_log, const Fe<F>* src0) {
    // src0 (optional): the input is read from there by the first pass and left untouched; the result lands in `a` (out of place)
    if (!src0) src0 = a;
    if (logn > (uint32_t)F::TWO_ADICITY || logn > 30) return ZK_ERR_INVALID_ARG;   // before anything is sized from logn
    if ((uint32_t)((scale_flag >> 4) & 15) > logn || (scale_flag & ~0xf3)) return ZK_ERR_INVALID_ARG;
    if (logn == 0) return ZK_OK;  // size-1 transform is the identity, n^-1 = 1 and g^0 = 1
    // lazy 29-bit limbs inside the tiles (zk_ntt29_kernels.h) unless zk_ntt_opts asks for the saturated words
    const bool lazy = g.ntt_opts.limb_bits != 32;
    PowTables<F> tpre{nullptr, nullptr}, tpost{nullptr, nullptr};
    if (g_pre) ZK_TRY(pow_tables<F>(dc, *g_pre, logn, field, st, &tpre, lazy));
    if (g_post) ZK_TRY(pow_tables<F>(dc, *g_post, logn, field, st, &tpost, lazy));
    const Fe<F>* tw = nullptr;
    ZK_TRY(tw_table<F>(dc, omega, logn, field, st, &tw, lazy));
    // scale_flag: bits 4..7 (ZK_NTT_OUT_SUBCOSETS(log P)) = the last pass stores result k at (k mod P) (n / P) + k / P ;
    // bit 0 = multiply by n^-1 ; bit 1 (ZK_NTT_OUT_R29) = leave the results as x R' mod p (R' = 2^261, the lazy-limb
    // radix zk_expr_eval_lazy_device reads) instead of x R mod p: the last pass multiplies by 2^5 more
    Fe<F> scale;
    fe_one(scale);
    if (scale_flag & 1) {
        Fe<F> nn;
        fe_zero(nn);
        nn.v[logn / 32] = 1u << (logn % 32);
        fe_to_mont(nn, nn);
        fe_inv(scale, nn);
    }
    {
        constexpr int SH = F29<F>::W * F29<F>::L - 32 * F::N;
        if (scale_flag & 2)
            for (int k = 0; k < SH; k++) fe_dbl(scale, scale);
        if (lazy)   // the last pass always multiplies by `scale`: n^-1 (or 1) in R' form
            for (int k = 0; k < SH; k++) fe_dbl(scale, scale);
    }
    NttPlan plan = ntt_plan(logn);
    Fe<F>* tmp = nullptr;
    if (plan.nd > 1) {
        StreamScratch* ss = nullptr;
        ZK_TRY(stream_scratch(dc, st, &ss));   // the ping-pong buffer belongs to the caller's stream
        ZK_TRY(ws_get(ss->ntt_tmp, sizeof(Fe<F>) << logn));
        tmp = (Fe<F>*)ss->ntt_tmp.p;
    }
    int log_m = 0;
    for (int p = 0; p < plan.nd; p++) {
        NttPass A;
        memset(&A, 0, sizeof A);
        A.logn = (int)logn;
        A.log_m = log_m;
        A.log_r = plan.rd[p];
        A.log_t = plan.log_t[p];
        A.last = (p == plan.nd - 1);
        A.scale = (A.last && (scale_flag & 3)) ? 1 : 0;
        A.out_parts_log = A.last ? ((scale_flag >> 4) & 15) : 0;
        A.nd = plan.nd;
        for (int i = 0; i < plan.nd; i++) A.rd[i] = plan.rd[i];
        A.pre = (p == 0 && g_pre) ? 1 : 0;
        A.in_log = (p == 0 && in_log > 0 && in_log < logn) ? (int)in_log : 0;
        A.post = (A.last && g_post) ? 1 : 0;
        const Fe<F>* src;
        Fe<F>* dst;
        if (plan.nd == 1) {
            src = src0;
            dst = a;
        } else if (p == 0) {
            src = src0;
            dst = tmp;
        } else if (A.last) {
            src = tmp;
            dst = a;
        } else {
            src = tmp;
            dst = tmp;
        }
        const uint64_t tiles = (1ull << logn) >> (A.log_r + A.log_t);
        const uint32_t rt = 1u << (A.log_r + A.log_t);
        unsigned max_blk = 512;  // 8 waves per tile: measured best (tools/tune_ntt.py)
        {
            const int v = g.ntt_opts.block;
            if (v == 64 || v == 128 || v == 256 || v == 512 || v == 1024) max_blk = (unsigned)v;
        }
        unsigned blk = rt / 2 < 64 ? 64 : (rt / 2 > max_blk ? max_blk : rt / 2);
        const size_t shmem = (size_t)rt * (lazy ? sizeof(Fe29<F>) : sizeof(Fe<F>));
        if (shmem > 48 * 1024) {
            if (lazy)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass29_kernel<F>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            else
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_kernel<F>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (g.ntt_profile) {   // bracket the launch with events on its own stream (zk_ntt_profile_read sums them)
            while (dc.ntt_ev_pool.size() < dc.ntt_ev_used + 2) {
                hipEvent_t e;
                HIP_TRY(hipEventCreate(&e));
                dc.ntt_ev_pool.push_back(e);
            }
            e0 = dc.ntt_ev_pool[dc.ntt_ev_used++];
            e1 = dc.ntt_ev_pool[dc.ntt_ev_used++];
            HIP_TRY(hipEventRecord(e0, st));
        }
        if (lazy)
            ZK_LAUNCH((ntt_pass29_kernel<F>), (unsigned)tiles, blk, shmem, st, src, dst, tw, (const InnerTw*)(tw + (logn > 0 ? (1ull << (logn - 1)) : 1)),
                      A, scale, tpre, tpost);
        else
            ZK_LAUNCH((ntt_pass_kernel<F>), (unsigned)tiles, blk, shmem, st, src, dst, tw, A, scale, tpre, tpost);
        if (e1) HIP_TRY(hipEventRecord(e1, st));
        HIP_TRY(hipGetLastError());
        log_m += plan.rd[p];
    }
    if (g.ntt_profile) {
        dc.ntt_transforms++;
        const double n_in = (in_log > 0 && in_log < logn) ? (double)(1ull << in_log) : (double)(1ull << logn);
        dc.ntt_alg_bytes += sizeof(Fe<F>) * (n_in + (double)(1ull << logn));
    }
    return ZK_OK;
}

template <class F>
int coset_run(DeviceCtx& dc, int field, Fe<F>* a, uint32_t logn, const Fe<F>& gshift, hipStream_t st) {
    if (logn > 30) return ZK_ERR_INVALID_ARG;
    const uint64_t count = 1ull << logn;
    PowTables<F> t;
    ZK_TRY(pow_tables<F>(dc, gshift, logn, field, st, &t));
    uint64_t blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    ZK_LAUNCH((coset_mul_kernel<F>), (unsigned)blocks, 256, 0, st, a, t, count);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}

template <class F>
int vec_op_run(Fe<F>* a, const Fe<F>* b, const Fe<F>* c, uint64_t n, int op, const Fe<F>& s, hipStream_t st) {
    if (n == 0) return ZK_OK;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;  // grid-stride: 16 workgroups per CU
    ZK_LAUNCH((vec_op_kernel<F>), (unsigned)blocks, 256, 0, st, a, b, c, n, op, s);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}

template <class F>
int scale_periodic_run(Fe<F>* a, uint64_t n, const Fe<F>* table_host, uint32_t m, hipStream_t st) {
    if (n == 0) return ZK_OK;
    if (m == 0 || m > 16 || (m & (m - 1)) != 0) return ZK_ERR_INVALID_ARG;
    PeriodicTable<F> t;
    memset(&t, 0, sizeof t);
    for (uint32_t i = 0; i < m; i++) t.v[i] = table_host[i];
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ZK_LAUNCH((scale_periodic_kernel<F>), (unsigned)blocks, 256, 0, st, a, n, t, m);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}

// out[0 .. n_rows) = M z, out[n_rows .. out_len) = 0   (CSR matrix resident on the device, see zk_r1cs_kernels.h)
template <class F>
int r1cs_matvec_run(const R1csMatrix& m, const Fe<F>* z, Fe<F>* out, uint64_t out_len, hipStream_t st) {
    if (out_len < m.n_rows) return ZK_ERR_INVALID_ARG;
    if (out_len == 0) return ZK_OK;
    uint64_t blocks = (out_len + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    ZK_LAUNCH((r1cs_matvec_kernel<F>), (unsigned)blocks, 256, 0, st, (const uint64_t*)m.row_ptr, (const uint32_t*)m.col, (const Fe<F>*)m.val, z,
              out, m.n_rows, out_len);
    if (m.n_long)
        ZK_LAUNCH((r1cs_matvec_long_kernel<F>), (unsigned)m.n_long, 256, 0, st, (const uint64_t*)m.row_ptr, (const uint32_t*)m.col,
                  (const Fe<F>*)m.val, z, out, (const uint64_t*)m.long_rows);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}

// ark-groth16 0.3 r1cs_to_qap.rs  R1CStoQAP::witness_map, from the point where a, b, c hold the evaluations
// <A_i,z>, <B_i,z>, <C_i,z> on the size-m domain: seven NTTs and the pointwise glue, all in HBM.
//   ifft(a); ifft(b); coset_fft(a); coset_fft(b); ifft(c); coset_fft(c);
//   ab = a.b - c;  ab *= 1/Z_H(g);  coset_ifft(ab)            -> `a` holds h (m coefficients, Montgomery)
template <class F>
int witness_map_run(DeviceCtx& dc, int field, Fe<F>* a, Fe<F>* b, Fe<F>* c, uint32_t logm, hipStream_t st) {
    if (logm > (uint32_t)F::TWO_ADICITY || logm > 30) return ZK_ERR_INVALID_ARG;
    const uint64_t m = 1ull << logm;
    Fe<F> w, winv, gen, ginv, zinv, one;
    for (int i = 0; i < F::N; i++) {
        w.v[i] = F::ROOT[i];
        gen.v[i] = F::GEN[i];
    }
    for (uint32_t i = logm; i < (uint32_t)F::TWO_ADICITY; i++) fe_sqr(w, w);
    fe_inv(winv, w);
    fe_inv(ginv, gen);
    // Z_H(g) = g^m - 1 on the coset gH
    Fe<F> gm = gen;
    for (uint32_t i = 0; i < logm; i++) fe_sqr(gm, gm);
    fe_one(one);
    fe_sub(gm, gm, one);
    fe_inv(zinv, gm);
    Fe<F>* vs[3] = {a, b, c};
    for (int k = 0; k < 3; k++) {
        ZK_TRY(ntt_run<F>(dc, field, vs[k], logm, winv, 1, st, nullptr, nullptr));   // ifft_in_place
        ZK_TRY(ntt_run<F>(dc, field, vs[k], logm, w, 0, st, &gen, nullptr));          // coset_fft = distribute_powers(g) ; fft (fused)
    }
    ZK_TRY(vec_op_run<F>(a, b, c, m, VEC_QAP, zinv, st));
    ZK_TRY(ntt_run<F>(dc, field, a, logm, winv, 1, st, nullptr, &ginv));             // coset_ifft = ifft ; distribute_powers(g^-1) (fused)
    return ZK_OK;
}
}  // namespace zk
#include "zk_poly.inl"
