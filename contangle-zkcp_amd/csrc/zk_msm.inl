// MSM launch sequence and host tail.  Included by zk_msm_inst.cc, once per curve.
#pragma once
#include "zk_internal.h"
namespace zk {
// ------------------------------------------------------------------ MSM
inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class C>
int msm_run(const BasesEntry& be, const Fe<typename C::Fr>* d_scalars, uint64_t n, int mont, const zk_msm_opts* opts,
            void* out_jac, hipStream_t st) {
    using Fq = typename C::Fq;
    Jacobian<C> result;
    memset(&g.prof, 0, sizeof g.prof);
    XYZZ<C> total;
    xyzz_set_inf(total);
    const int c = msm_pick_c(n, opts ? opts->window_bits : 0);
    const int nwin = msm_windows<C>(c);
    int w0 = 0, w1 = nwin;
    if (opts && !(opts->window_begin == 0 && opts->window_end == 0)) {
        w0 = opts->window_begin;
        w1 = opts->window_end;
        if (w0 < 0 || w1 > nwin || w0 > w1) return ZK_ERR_INVALID_ARG;
    }
    g.prof.window_bits = c;
    g.prof.windows_total = nwin;
    g.prof.windows_done = w1 - w0;
    if (n > 0 && w1 > w0) {
        if (n >= (1ull << 31)) return ZK_ERR_UNSUPPORTED;
        MsmShape sh;
        sh.n = (uint32_t)n;
        sh.c = c;
        sh.w0 = w0;
        sh.nw = w1 - w0;
        sh.nbk = 1u << (c - 1);
        sh.mont = mont;
        const uint32_t nbuckets = (uint32_t)sh.nw * sh.nbk;
        // counts | offs | cursor
        ZK_TRY(ws_get(g.msm_counts, (size_t)nbuckets * 4 * 3));
        uint32_t* counts = (uint32_t*)g.msm_counts.p;
        uint32_t* offs = counts + nbuckets;
        uint32_t* cursor = offs + nbuckets;
        ZK_TRY(ws_get(g.msm_sorted, (size_t)n * sh.nw * 4));
        ZK_TRY(ws_get(g.msm_buckets, (size_t)nbuckets * sizeof(XYZZ<C>)));
        uint32_t L = 8;
        if (const char* e = getenv("ZK_MSM_SLICE")) {
            int v = atoi(e);
            if (v >= 1 && v <= 1024) L = (uint32_t)v;
        }
        if (L > sh.nbk) L = sh.nbk;
        const uint32_t spw = (sh.nbk + L - 1) / L;
        const uint32_t nslices = spw * (uint32_t)sh.nw;
        ZK_TRY(ws_get(g.msm_part_a, (size_t)nslices * sizeof(XYZZ<C>)));
        ZK_TRY(ws_get(g.msm_part_b, ((size_t)nslices / 256 + (size_t)sh.nw + 8) * sizeof(XYZZ<C>)));
        if (!g.have_events) {
            for (auto& e : g.ev) HIP_TRY(hipEventCreate(&e));
            g.have_events = true;
        }
        const unsigned blk = 256;
        HIP_TRY(hipEventRecord(g.ev[0], st));
        HIP_TRY(hipMemsetAsync(counts, 0, (size_t)nbuckets * 4 * 3, st));
        ZK_LAUNCH((msm_hist_kernel<C>), (unsigned)((n + blk - 1) / blk), blk, 0, st, d_scalars, sh, counts);
        HIP_TRY(hipEventRecord(g.ev[1], st));
        ZK_LAUNCH((msm_scan_kernel<void>), 1, 1024, 0, st, (const uint32_t*)counts, offs, nbuckets);
        HIP_TRY(hipEventRecord(g.ev[2], st));
        ZK_LAUNCH((msm_scatter_kernel<C>), (unsigned)((n + blk - 1) / blk), blk, 0, st, d_scalars, sh, (const uint32_t*)offs,
                  cursor, (uint32_t*)g.msm_sorted.p);
        HIP_TRY(hipEventRecord(g.ev[3], st));
        ZK_LAUNCH((msm_accumulate_kernel<C>), (nbuckets + 63) / 64, 64, 0, st, (const Affine<C>*)be.dev,
                  (const uint32_t*)g.msm_sorted.p, (const uint32_t*)offs, (const uint32_t*)counts, (XYZZ<C>*)g.msm_buckets.p,
                  nbuckets);
        HIP_TRY(hipEventRecord(g.ev[4], st));
        ZK_LAUNCH((msm_reduce_kernel<C>), (nslices + 63) / 64, 64, 0, st, (const XYZZ<C>*)g.msm_buckets.p,
                  (XYZZ<C>*)g.msm_part_a.p, sh.nbk, L, spw, nslices);
        // tree-sum the slices of each window until <= 8 remain
        XYZZ<C>* cur = (XYZZ<C>*)g.msm_part_a.p;
        XYZZ<C>* nxt = (XYZZ<C>*)g.msm_part_b.p;
        uint32_t per = spw;
        while (per > 8) {
            const uint32_t E = per >= 1024 ? 4 : 1;
            const uint32_t chunk = 256 * E;
            const uint32_t per_out = (per + chunk - 1) / chunk;
            ZK_LAUNCH((msm_sum_kernel<C>), (unsigned)sh.nw * per_out, 256, 0, st, (const XYZZ<C>*)cur, nxt, per, per_out, E);
            per = per_out;
            XYZZ<C>* t = cur;
            cur = nxt;
            nxt = t;
        }
        HIP_TRY(hipEventRecord(g.ev[5], st));
        HIP_TRY(hipGetLastError());
        std::vector<XYZZ<C>> host((size_t)sh.nw * per);
        HIP_TRY(hipMemcpyAsync(host.data(), cur, host.size() * sizeof(XYZZ<C>), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const double t0 = now_ms();
        // Horner over this call's windows, high to low, then the shift by 2^(c*w0)
        for (int w = sh.nw - 1; w >= 0; w--) {
            for (int k = 0; k < c; k++) xyzz_dbl(total);
            for (uint32_t i = 0; i < per; i++) xyzz_add(total, host[(size_t)w * per + i]);
        }
        for (int k = 0; k < c * w0; k++) xyzz_dbl(total);
        g.prof.host_tail_ms = (float)(now_ms() - t0);
        hipEventElapsedTime(&g.prof.digits_hist_ms, g.ev[0], g.ev[1]);
        hipEventElapsedTime(&g.prof.scan_ms, g.ev[1], g.ev[2]);
        hipEventElapsedTime(&g.prof.scatter_ms, g.ev[2], g.ev[3]);
        hipEventElapsedTime(&g.prof.accumulate_ms, g.ev[3], g.ev[4]);
        hipEventElapsedTime(&g.prof.reduce_ms, g.ev[4], g.ev[5]);
        hipEventElapsedTime(&g.prof.total_ms, g.ev[0], g.ev[5]);
        g.prof.total_ms += g.prof.host_tail_ms;
    }
    xyzz_to_jacobian(result, total);
    memcpy(out_jac, &result, 3 * sizeof(uint32_t) * Fq::N);
    return ZK_OK;
}


template <class C>
int fixed_base_run(const Fe<typename C::Fr>* d_scalars, uint64_t n, Affine<C>* d_out, hipStream_t st) {
    ZK_LAUNCH((fixed_base_mul_kernel<C>), (unsigned)((n + 63) / 64), 64, 0, st, d_scalars, d_out, (uint32_t)n);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}
}  // namespace zk
