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
        {   // oversize threshold: 2x the mean bucket length + 64 (uniform 2^20 / c=16: mean 32, max ~70 -> none)
            const uint64_t mean = n / sh.nbk;
            sh.big_thresh = (uint32_t)(2 * mean + 64);
            if (const char* e = getenv("ZK_MSM_BIG")) {
                int v = atoi(e);
                if (v >= 1) sh.big_thresh = (uint32_t)v;
            }
        }
        const int nw_all = sh.nw;
        // Bucket splitting: with fewer than ~4 work items per resident lane the accumulate kernel ends in a long drain (every
        // lane finishing one partly summed bucket) -- 2.7 buckets per lane for a full 2^20 MSM, less than one for the window
        // share of a rank of a sharded MSM.  Cut every bucket into 2^split_log pieces (one extra point addition per piece).
        // Measured at 2^20 / c = 16 (tools/shard_model.py, accumulate ms for split 0/1/2/3):
        //   16 windows 1.49/1.32/1.36/1.59   8 windows 0.71/0.65/0.70/0.84   4 windows 0.55/0.42/0.39/0.49   2: 0.44/0.33/0.25/0.28
        // i.e. halves while there are fewer than 8 buckets per lane, quarters below one bucket per lane, pieces never
        // shorter than 8 entries on average.
        {
            const uint64_t lanes = (uint64_t)(g.num_cus > 0 ? g.num_cus : 256) * 4 * 3 * 64;
            const uint64_t items = (uint64_t)nw_all * sh.nbk;
            int sl = 0;
            if (items < 8 * lanes && (n >> 1) / sh.nbk >= 8) sl = 1;
            if (items < lanes && (n >> 2) / sh.nbk >= 8) sl = 2;
            if (const char* e = getenv("ZK_MSM_SPLIT")) {
                int v = atoi(e);
                if (v >= 0 && v <= 4) sl = v;
            }
            sh.split_log = sl;
        }
        // Window groups (experimental, default 1): the windows of the call can be cut into G groups issued alternately on two
        // HIP streams.  Measured on MI355X at 2^20 (tools/tune_msm.py): G = 2 / 4 is 1.5x / 2.7x SLOWER -- two persistent
        // accumulate kernels simply share the SIMDs, every group pays its own queue-drain tail, and the later groups' sort
        // kernels crawl beside the resident accumulate waves.  Kept because the grouped path is what a multi-MSM batch
        // (several commitments in flight) would use; ZK_MSM_GROUPS selects it.
        int G = 1;
        if (const char* e = getenv("ZK_MSM_GROUPS")) {
            int v = atoi(e);
            if (v >= 1 && v <= MSM_MAX_GROUPS) G = v;
        }
        if (G > nw_all) G = nw_all;
        const uint32_t nbuckets = (uint32_t)nw_all * sh.nbk;
        const uint32_t nwg_all = (uint32_t)nw_all * sh.nranges;
        // counts | offs | order | wg_total
        ZK_TRY(ws_get(g.msm_counts, ((size_t)nbuckets * 3 + nwg_all) * 4));
        uint32_t* counts = (uint32_t*)g.msm_counts.p;
        uint32_t* offs = counts + nbuckets;
        uint32_t* order = offs + nbuckets;
        uint32_t* wg_total = order + nbuckets;
        ZK_TRY(ws_get(g.msm_digits, (size_t)sh.n_pad * nw_all * 2));
        uint16_t* digits = (uint16_t*)g.msm_digits.p;
        ZK_TRY(ws_get(g.msm_sorted, (size_t)n * nw_all * 4));
        ZK_TRY(ws_get(g.msm_buckets, (size_t)nbuckets * sizeof(XYZZ<CK>)));
        if (sh.split_log > 0) ZK_TRY(ws_get(g.msm_subacc, ((size_t)nbuckets << sh.split_log) * sizeof(XYZZ<CK>)));
        // slice length of the bucket reduction: keep ~64K lanes busy whatever the window share of this call (a rank of a
        // window-sharded MSM owns few windows; shorter slices shorten the dependent chain, which is all this phase costs)
        uint32_t L = (uint32_t)(((uint64_t)nw_all * sh.nbk) >> 16);
        if (L < 1) L = 1;
        if (L > 8) L = 8;
        if (const char* e = getenv("ZK_MSM_SLICE")) {
            int v = atoi(e);
            if (v >= 1 && v <= 1024) L = (uint32_t)v;
        }
        if (L > sh.nbk) L = sh.nbk;
        const uint32_t spw = (sh.nbk + L - 1) / L;                  // slices per window
        const uint32_t pbw = spw / 128 + 2;                          // second ping-pong buffer, points per window
        ZK_TRY(ws_get(g.msm_part_a, (size_t)spw * nw_all * sizeof(XYZZ<CK>)));
        ZK_TRY(ws_get(g.msm_part_b, (size_t)pbw * nw_all * sizeof(XYZZ<CK>)));
        // oversized-bucket lists: a bucket above big_thresh yields ceil(cnt / MSM_SEG) segments
        const size_t max_seg_w = (size_t)n / MSM_SEG + (size_t)n / sh.big_thresh + 2;   // per window
        const size_t qstride_w = ((sizeof(MsmQueue) + max_seg_w * (sizeof(MsmSeg) + 8) + 15) / 16 + 1) * 16;
        ZK_TRY(ws_get(g.msm_queue, qstride_w * nw_all + sizeof(MsmQueue) * MSM_MAX_GROUPS));
        ZK_TRY(ws_get(g.msm_seg_out, max_seg_w * nw_all * sizeof(XYZZ<CK>)));
        if (!g.have_events) {
            for (auto& e : g.ev) HIP_TRY(hipEventCreate(&e));
            HIP_TRY(hipStreamCreateWithFlags(&g.aux_stream, hipStreamNonBlocking));
            g.have_events = true;
        }
        hipStream_t streams[2] = {st, G > 1 ? g.aux_stream : st};
        // resident waves per SIMD: the F29 kernel holds 138 VGPRs (3 fit), the 32-bit one 119 (4 fit); tools/tune_msm.py
        unsigned waves_per_simd = CK::EXT == 29 ? 3 : 4;
        if (const char* e = getenv("ZK_MSM_WAVES")) {
            int v = atoi(e);
            if (v >= 1 && v <= 8) waves_per_simd = (unsigned)v;
        }
        const unsigned blk = 256;
        hipEvent_t* EV = g.ev;   // [0] begin, [1] digits done, [2] aux done, [3] end; then 5 per group
        HIP_TRY(hipEventRecord(EV[0], st));
        ZK_LAUNCH((msm_digits_kernel<C>), (unsigned)((sh.n_pad + blk - 1) / blk), blk, 0, st, d_scalars, sh, digits);
        HIP_TRY(hipEventRecord(EV[1], st));
        if (G > 1) HIP_TRY(hipStreamWaitEvent(g.aux_stream, EV[1], 0));

        // tree levels of the per-window sums are the same for every group
        uint32_t per_final = spw;
        while (per_final > 8) {
            const uint32_t E = per_final >= 1024 ? 4 : 1;
            per_final = (per_final + tree_lanes<CK>() * E - 1) / (tree_lanes<CK>() * E);
        }
        std::vector<XYZZ<CK>> host((size_t)nw_all * per_final);

        for (int gi = 0; gi < G; gi++) {
            hipStream_t S = streams[gi & 1];
            hipEvent_t* ev = EV + 4 + 5 * gi;
            const int lw0 = nw_all * gi / G, lw1 = nw_all * (gi + 1) / G;
            MsmShape sg = sh;
            sg.w0 = w0 + lw0;
            sg.nw = lw1 - lw0;
            const uint32_t nb_g = (uint32_t)sg.nw * sh.nbk;
            const uint32_t nwg = (uint32_t)sg.nw * sh.nranges;
            uint32_t* counts_g = counts + (size_t)lw0 * sh.nbk;
            uint32_t* offs_g = offs + (size_t)lw0 * sh.nbk;
            uint32_t* order_g = order + (size_t)lw0 * sh.nbk;
            uint32_t* wg_total_g = wg_total + (size_t)lw0 * sh.nranges;
            const uint16_t* digits_g = digits + (size_t)lw0 * sh.n_pad;
            uint32_t* sorted_g = (uint32_t*)g.msm_sorted.p + (size_t)lw0 * n;
            XYZZ<CK>* buckets_g = (XYZZ<CK>*)g.msm_buckets.p + (size_t)lw0 * sh.nbk;
            XYZZ<CK>* acc_out_g = sh.split_log > 0 ? (XYZZ<CK>*)g.msm_subacc.p + (((size_t)lw0 * sh.nbk) << sh.split_log) : buckets_g;
            char* qbase = (char*)g.msm_queue.p + qstride_w * lw0;
            MsmQueue* q = (MsmQueue*)qbase;
            const size_t max_seg = max_seg_w * sg.nw;
            MsmSeg* seg_list = (MsmSeg*)(q + 1);
            uint32_t* big_list = (uint32_t*)(seg_list + max_seg);
            XYZZ<CK>* seg_out = (XYZZ<CK>*)g.msm_seg_out.p + max_seg_w * lw0;
            XYZZ<CK>* cur = (XYZZ<CK>*)g.msm_part_a.p + (size_t)spw * lw0;
            XYZZ<CK>* nxt = (XYZZ<CK>*)g.msm_part_b.p + (size_t)pbw * lw0;
            // the first group's sort runs alone on the chip: 16 waves per workgroup; later groups must fit beside the
            // resident accumulate waves: 4 waves per workgroup
            const unsigned sblk = (gi == 0 && sh.n_pad >= 8192) ? 1024 : 256;
            HIP_TRY(hipEventRecord(ev[0], S));
            ZK_LAUNCH((msm_hist_kernel<void>), nwg, sblk, (size_t)(sh.rb + 1) * 4, S, digits_g, sg, counts_g, wg_total_g);
            HIP_TRY(hipEventRecord(ev[1], S));
            ZK_LAUNCH((msm_scatter_kernel<void>), nwg, sblk, (size_t)(sh.rb + 1024 + 258) * 4, S, digits_g, sg,
                      (const uint32_t*)counts_g, (const uint32_t*)wg_total_g, offs_g, order_g, sorted_g);
            HIP_TRY(hipEventRecord(ev[2], S));
            // persistent accumulate: lanes stream buckets, largest first; oversized buckets go to the cooperative segment
            // kernels (fixed grids over device-side lists, no host round trip)
            HIP_TRY(hipMemsetAsync(q, 0, sizeof(MsmQueue), S));
            const uint32_t ntasks = (((nwg * ((sh.rb + 63) / 64) * 64) << sh.split_log) + MSM_BATCH - 1) / MSM_BATCH;  // batches in the queue
            unsigned acc_grid = (g.num_cus > 0 ? (unsigned)g.num_cus : 256u) * 4u * waves_per_simd;
            if (acc_grid > ntasks) acc_grid = ntasks;
            ZK_LAUNCH((msm_accumulate_kernel<CK>), acc_grid, 64, 0, S, bases, (const uint32_t*)sorted_g, (const uint32_t*)offs_g,
                      (const uint32_t*)counts_g, (const uint32_t*)order_g, acc_out_g, sg, q, seg_list, big_list);
            if (sh.split_log > 0)
                ZK_LAUNCH((msm_combine_sub_kernel<CK>), (nb_g + 63) / 64, 64, 0, S, (const XYZZ<CK>*)acc_out_g, buckets_g, nb_g, sh.split_log);
            const unsigned big_grid = max_seg < 4096 ? (unsigned)max_seg : 4096u;
            ZK_LAUNCH((msm_accumulate_big_kernel<CK>), big_grid, 64, 0, S, bases, (const uint32_t*)sorted_g, (const MsmQueue*)q,
                      (const MsmSeg*)seg_list, seg_out);
            ZK_LAUNCH((msm_combine_big_kernel<CK>), big_grid < 64 ? big_grid : 64u, tree_lanes<CK>(), 0, S, (const MsmQueue*)q,
                      (const uint32_t*)big_list, (const uint32_t*)counts_g, (const XYZZ<CK>*)seg_out, buckets_g);
            HIP_TRY(hipEventRecord(ev[3], S));
            const uint32_t nslices = spw * (uint32_t)sg.nw;
            ZK_LAUNCH((msm_reduce_kernel<CK>), (nslices + 63) / 64, 64, 0, S, (const XYZZ<CK>*)buckets_g, cur, sh.nbk, L, spw, nslices);
            // tree-sum the slices of each window until <= 8 remain
            uint32_t per = spw;
            while (per > 8) {
                const uint32_t E = per >= 1024 ? 4 : 1;
                const uint32_t chunk = tree_lanes<CK>() * E;
                const uint32_t per_out = (per + chunk - 1) / chunk;
                ZK_LAUNCH((msm_sum_kernel<CK>), (unsigned)sg.nw * per_out, tree_lanes<CK>(), 0, S, (const XYZZ<CK>*)cur, nxt, per, per_out, E);
                per = per_out;
                XYZZ<CK>* t = cur;
                cur = nxt;
                nxt = t;
            }
            HIP_TRY(hipEventRecord(ev[4], S));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(host.data() + (size_t)lw0 * per_final, cur, (size_t)sg.nw * per_final * sizeof(XYZZ<CK>),
                                   hipMemcpyDeviceToHost, S));
        }
        if (G > 1) {
            HIP_TRY(hipEventRecord(EV[2], g.aux_stream));
            HIP_TRY(hipStreamWaitEvent(st, EV[2], 0));
        }
        HIP_TRY(hipEventRecord(EV[3], st));
        HIP_TRY(hipStreamSynchronize(st));
        if (G > 1) HIP_TRY(hipStreamSynchronize(g.aux_stream));
        const uint32_t per = per_final;
        const double t0 = now_ms();
        // Horner over this call's windows, high to low, then the shift by 2^(c*w0) -- on 64-bit host limbs
        HostXYZZ<C> htotal, hp;
        to_host<C>(htotal, total);
        for (int w = nw_all - 1; w >= 0; w--) {
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
        // phase times are summed over the groups (they overlap in wall time when G > 1); total_ms is wall time
        hipEventElapsedTime(&g.prof.digits_ms, EV[0], EV[1]);
        for (int gi = 0; gi < G; gi++) {
            hipEvent_t* ev = EV + 4 + 5 * gi;
            float t;
            hipEventElapsedTime(&t, ev[0], ev[1]);
            g.prof.hist_ms += t;
            hipEventElapsedTime(&t, ev[1], ev[2]);
            g.prof.scatter_ms += t;
            hipEventElapsedTime(&t, ev[2], ev[3]);
            g.prof.accumulate_ms += t;
            hipEventElapsedTime(&t, ev[3], ev[4]);
            g.prof.reduce_ms += t;
        }
        hipEventElapsedTime(&g.prof.total_ms, EV[0], EV[3]);
        g.prof.total_ms += g.prof.host_tail_ms;
        g.prof.groups = G;
    }
    xyzz_to_jacobian(result, total);
    memcpy(out_jac, &result, 3 * sizeof(uint32_t) * coord_words<C>());
    return ZK_OK;
}


// the G1 curves (Pallas, Vesta, BN254 G1: 9 x 29-bit limbs; BLS12-381 G1: 14 x 28) run their bucket arithmetic in the
// lazy-limb view
template <class C>
constexpr bool has_f29() {
    return C::EXT == 1;
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
