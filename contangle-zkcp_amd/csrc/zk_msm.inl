// MSM launch sequence and host tail.  Included by zk_msm_inst.cc, once per curve.
#pragma once
#include "zk_internal.h"
#include "zk_host64.h"
#include "zk_msm_kernels.h"
namespace zk {
// ------------------------------------------------------------------ MSM
inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// partial sums leave the device in the kernels' view CK of the curve (C itself, or its F29 view) and are brought back
// to the caller's limb form here
template <class C>
inline void partial_to_std(XYZZ<C>& r, const XYZZ<C>& p) {
    r = p;
}
template <class C>
inline void partial_to_std(XYZZ<C>& r, const XYZZ<C29<C>>& p) {
    xyzz29_to_std<C>(r, p);
}

// C: the curve of the ABI call.  CK: the view the kernels compute in.  bases: device array of Affine<CK>.
template <class C, class CK>
int msm_run_impl(const Affine<CK>* bases, const Fe<typename C::Fr>* d_scalars, uint64_t n, int mont, const zk_msm_opts* opts,
                 void* out_jac, hipStream_t st) {
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
    g.prof.limb_bits = CK::EXT == 29 ? 29 : 32;
    if (n > 0 && w1 > w0) {
        if (n >= (1ull << 31)) return ZK_ERR_UNSUPPORTED;
        MsmShape sh;
        sh.n = (uint32_t)n;
        sh.n_pad = (uint32_t)((n + 7) & ~7ull);
        sh.c = c;
        sh.w0 = w0;
        sh.nw = w1 - w0;
        sh.nbk = 1u << (c - 1);
        sh.rb = sh.nbk < 2048u ? sh.nbk : 2048u;
        sh.nranges = sh.nbk / sh.rb;
        sh.mont = mont;
        sh.idx_mask = 0x7fffffffu;
        if (const char* e = getenv("ZK_MSM_DEBUG_MASK")) sh.idx_mask = (uint32_t)strtoul(e, nullptr, 0);  // profiling only
        {   // oversize threshold: 2x the mean bucket length + 64 (uniform 2^20 / c=16: mean 32, max ~70 -> none)
            const uint64_t mean = n / sh.nbk;
            sh.big_thresh = (uint32_t)(2 * mean + 64);
            if (const char* e = getenv("ZK_MSM_BIG")) {
                int v = atoi(e);
                if (v >= 1) sh.big_thresh = (uint32_t)v;
            }
        }
        const uint32_t nbuckets = (uint32_t)sh.nw * sh.nbk;
        const uint32_t nwg = (uint32_t)sh.nw * sh.nranges;
        // counts | offs | order | wg_total
        ZK_TRY(ws_get(g.msm_counts, ((size_t)nbuckets * 3 + nwg) * 4));
        uint32_t* counts = (uint32_t*)g.msm_counts.p;
        uint32_t* offs = counts + nbuckets;
        uint32_t* order = offs + nbuckets;
        uint32_t* wg_total = order + nbuckets;
        ZK_TRY(ws_get(g.msm_digits, (size_t)sh.n_pad * sh.nw * 2));
        uint16_t* digits = (uint16_t*)g.msm_digits.p;
        ZK_TRY(ws_get(g.msm_sorted, (size_t)n * sh.nw * 4));
        ZK_TRY(ws_get(g.msm_buckets, (size_t)nbuckets * sizeof(XYZZ<CK>)));
        uint32_t L = 8;
        if (const char* e = getenv("ZK_MSM_SLICE")) {
            int v = atoi(e);
            if (v >= 1 && v <= 1024) L = (uint32_t)v;
        }
        if (L > sh.nbk) L = sh.nbk;
        const uint32_t spw = (sh.nbk + L - 1) / L;
        const uint32_t nslices = spw * (uint32_t)sh.nw;
        ZK_TRY(ws_get(g.msm_part_a, (size_t)nslices * sizeof(XYZZ<CK>)));
        ZK_TRY(ws_get(g.msm_part_b, ((size_t)nslices / 128 + (size_t)sh.nw + 8) * sizeof(XYZZ<CK>)));
        if (!g.have_events) {
            for (auto& e : g.ev) HIP_TRY(hipEventCreate(&e));
            g.have_events = true;
        }
        const unsigned blk = 256;
        const unsigned sblk = sh.n_pad >= 8192 ? 1024 : 256;  // lanes of the per-range sort workgroups (power of two)
        HIP_TRY(hipEventRecord(g.ev[0], st));
        ZK_LAUNCH((msm_digits_kernel<C>), (unsigned)((sh.n_pad + blk - 1) / blk), blk, 0, st, d_scalars, sh, digits);
        HIP_TRY(hipEventRecord(g.ev[1], st));
        ZK_LAUNCH((msm_hist_kernel<void>), nwg, sblk, (size_t)(sh.rb + 1) * 4, st, (const uint16_t*)digits, sh, counts, wg_total);
        HIP_TRY(hipEventRecord(g.ev[2], st));
        ZK_LAUNCH((msm_scatter_kernel<void>), nwg, sblk, (size_t)(sh.rb + 1024 + 258) * 4, st, (const uint16_t*)digits, sh,
                  (const uint32_t*)counts, (const uint32_t*)wg_total, offs, order, (uint32_t*)g.msm_sorted.p);
        HIP_TRY(hipEventRecord(g.ev[3], st));
        // persistent accumulate: 4 waves per SIMD pull 64-bucket tasks, largest first; oversized buckets go to the
        // cooperative segment kernels (fixed grids over device-side lists, no host round trip)
        // every oversized bucket has > big_thresh entries and yields ceil(cnt / MSM_SEG) segments
        const size_t max_seg = (size_t)n * sh.nw / MSM_SEG + (size_t)n * sh.nw / sh.big_thresh + 2;
        ZK_TRY(ws_get(g.msm_queue, sizeof(MsmQueue) + max_seg * sizeof(MsmSeg) + max_seg * 8));
        MsmQueue* q = (MsmQueue*)g.msm_queue.p;
        MsmSeg* seg_list = (MsmSeg*)(q + 1);
        uint32_t* big_list = (uint32_t*)(seg_list + max_seg);
        ZK_TRY(ws_get(g.msm_seg_out, max_seg * sizeof(XYZZ<CK>)));
        HIP_TRY(hipMemsetAsync(q, 0, sizeof(MsmQueue), st));
        const uint32_t ntasks = (nwg * ((sh.rb + 63) / 64) * 64 + MSM_BATCH - 1) / MSM_BATCH;  // batches in the queue
        // resident waves per SIMD: the F29 kernel holds 138 VGPRs (3 fit), the 32-bit one 119 (4 fit); tools/tune_msm.py
        unsigned waves_per_simd = CK::EXT == 29 ? 3 : 4;
        if (const char* e = getenv("ZK_MSM_WAVES")) {
            int v = atoi(e);
            if (v >= 1 && v <= 8) waves_per_simd = (unsigned)v;
        }
        unsigned acc_grid = (g.num_cus > 0 ? (unsigned)g.num_cus : 256u) * 4u * waves_per_simd;
        if (acc_grid > ntasks) acc_grid = ntasks;
        ZK_LAUNCH((msm_accumulate_kernel<CK>), acc_grid, 64, 0, st, bases, (const uint32_t*)g.msm_sorted.p,
                  (const uint32_t*)offs, (const uint32_t*)counts, (const uint32_t*)order, (XYZZ<CK>*)g.msm_buckets.p, sh, q, seg_list,
                  big_list);
        const unsigned big_grid = max_seg < 4096 ? (unsigned)max_seg : 4096u;
        ZK_LAUNCH((msm_accumulate_big_kernel<CK>), big_grid, 64, 0, st, bases, (const uint32_t*)g.msm_sorted.p,
                  (const MsmQueue*)q, (const MsmSeg*)seg_list, (XYZZ<CK>*)g.msm_seg_out.p);
        ZK_LAUNCH((msm_combine_big_kernel<CK>), big_grid < 64 ? big_grid : 64u, tree_lanes<CK>(), 0, st, (const MsmQueue*)q, (const uint32_t*)big_list,
                  (const uint32_t*)counts, (const XYZZ<CK>*)g.msm_seg_out.p, (XYZZ<CK>*)g.msm_buckets.p);
        HIP_TRY(hipEventRecord(g.ev[4], st));
        ZK_LAUNCH((msm_reduce_kernel<CK>), (nslices + 63) / 64, 64, 0, st, (const XYZZ<CK>*)g.msm_buckets.p,
                  (XYZZ<CK>*)g.msm_part_a.p, sh.nbk, L, spw, nslices);
        // tree-sum the slices of each window until <= 8 remain
        XYZZ<CK>* cur = (XYZZ<CK>*)g.msm_part_a.p;
        XYZZ<CK>* nxt = (XYZZ<CK>*)g.msm_part_b.p;
        uint32_t per = spw;
        while (per > 8) {
            const uint32_t E = per >= 1024 ? 4 : 1;
            const uint32_t chunk = tree_lanes<CK>() * E;
            const uint32_t per_out = (per + chunk - 1) / chunk;
            ZK_LAUNCH((msm_sum_kernel<CK>), (unsigned)sh.nw * per_out, tree_lanes<CK>(), 0, st, (const XYZZ<CK>*)cur, nxt, per, per_out, E);
            per = per_out;
            XYZZ<CK>* t = cur;
            cur = nxt;
            nxt = t;
        }
        HIP_TRY(hipEventRecord(g.ev[5], st));
        HIP_TRY(hipGetLastError());
        std::vector<XYZZ<CK>> host((size_t)sh.nw * per);
        HIP_TRY(hipMemcpyAsync(host.data(), cur, host.size() * sizeof(XYZZ<CK>), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const double t0 = now_ms();
        // Horner over this call's windows, high to low, then the shift by 2^(c*w0) -- on 64-bit host limbs
        HostXYZZ<C> htotal, hp;
        to_host<C>(htotal, total);
        for (int w = sh.nw - 1; w >= 0; w--) {
            for (int k = 0; k < c; k++) xyzz_dbl(htotal);
            for (uint32_t i = 0; i < per; i++) {
                XYZZ<C> ps;
                partial_to_std<C>(ps, host[(size_t)w * per + i]);
                to_host<C>(hp, ps);
                xyzz_add(htotal, hp);
            }
        }
        for (int k = 0; k < c * w0; k++) xyzz_dbl(htotal);
        from_host<C>(total, htotal);
        g.prof.host_tail_ms = (float)(now_ms() - t0);
        hipEventElapsedTime(&g.prof.digits_ms, g.ev[0], g.ev[1]);
        hipEventElapsedTime(&g.prof.hist_ms, g.ev[1], g.ev[2]);
        hipEventElapsedTime(&g.prof.scatter_ms, g.ev[2], g.ev[3]);
        hipEventElapsedTime(&g.prof.accumulate_ms, g.ev[3], g.ev[4]);
        hipEventElapsedTime(&g.prof.reduce_ms, g.ev[4], g.ev[5]);
        hipEventElapsedTime(&g.prof.total_ms, g.ev[0], g.ev[5]);
        g.prof.total_ms += g.prof.host_tail_ms;
    }
    xyzz_to_jacobian(result, total);
    memcpy(out_jac, &result, 3 * sizeof(uint32_t) * coord_words<C>());
    return ZK_OK;
}


// the 8-word G1 curves (Pallas, Vesta, BN254 G1) run their bucket arithmetic in the F29 view
template <class C>
constexpr bool has_f29() {
    return C::EXT == 1 && C::Fq::N == 8;
}
inline bool f29_enabled() {
    const char* e = getenv("ZK_MSM_F29");   // "0" forces the saturated 32-bit path (A/B measurements, tests)
    return !(e && e[0] == '0');
}

template <class C>
int bases_prepare_run(BasesEntry& be) {
    be.dev29 = nullptr;
    if constexpr (has_f29<C>()) {
        if (be.n == 0) return ZK_OK;
        void* d = nullptr;
        HIP_TRY(hipMalloc(&d, sizeof(Affine<C29<C>>) * be.n));
        ZK_LAUNCH((bases_to29_kernel<C>), (unsigned)((be.n + 255) / 256), 256, 0, (hipStream_t)0, (const Affine<C>*)be.dev,
                  (Affine<C29<C>>*)d, be.n);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize((hipStream_t)0) != hipSuccess) {
            hipFree(d);
            return ZK_ERR_HIP;
        }
        be.dev29 = d;
    }
    return ZK_OK;
}

template <class C>
int msm_run(const BasesEntry& be, const Fe<typename C::Fr>* d_scalars, uint64_t n, int mont, const zk_msm_opts* opts,
            void* out_jac, hipStream_t st) {
    if constexpr (has_f29<C>()) {
        if (be.dev29 && f29_enabled())
            return msm_run_impl<C, C29<C>>((const Affine<C29<C>>*)be.dev29, d_scalars, n, mont, opts, out_jac, st);
    }
    return msm_run_impl<C, C>((const Affine<C>*)be.dev, d_scalars, n, mont, opts, out_jac, st);
}

template <class C>
int fixed_base_run(const Fe<typename C::Fr>* d_scalars, uint64_t n, Affine<C>* d_out, hipStream_t st) {
    ZK_LAUNCH((fixed_base_mul_kernel<C>), (unsigned)((n + 63) / 64), 64, 0, st, d_scalars, d_out, (uint32_t)n);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}
}  // namespace zk
