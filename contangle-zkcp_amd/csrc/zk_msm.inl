// MSM launch sequence and host tail.  Included by zk_msm_inst.cc, once per curve.
#pragma once
#include "zk_internal.h"
#include "zk_msm_decl.h"
#include "zk_host64.h"
#include "zk_msm_kernels.h"
namespace zk {
// ------------------------------------------------------------------ MSM
inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// partial sums leave the device in the kernels' view CK of the curve (C itself, or its F29 view) and are brought back
// to the caller's limb form here
// Host tail of one MSM (per curve): Horner over the job's windows, high to low, then the shift by 2^(c*w0) -- on 64-bit
// host limbs, from the per-window partial sums the job copied to pinned memory.  Runs in zk_msm_collect, i.e. after the
// caller has had the chance to enqueue the next MSM: the GPU never waits for it.
template <class C, class CK>
int msm_finish_impl(MsmJob& job, void* out_jac) {
    const size_t out_bytes = 3 * sizeof(uint32_t) * coord_words<C>();
    double t0 = 0;
    if (!job.empty) {
        HIP_TRY(hipEventSynchronize(job.ev[6]));
        t0 = now_ms();
    }
    for (uint32_t bt = 0; bt < job.batch; bt++) {      // one result per scalar vector of the job
    Jacobian<C> result;
    XYZZ<C> total;
    xyzz_set_inf(total);
    if (!job.empty) {
        const XYZZ<CK>* host = (const XYZZ<CK>*)job.host_partials + (size_t)bt * job.nw * job.per;
        HostXYZZ<C> htotal, hp;
        to_host<C>(htotal, total);
        // ZK_MSM_FLAG_DEVICE_PARTIALS: the reduction's last kernel also wrote every partial in the caller's limb form (and the
        // stages of that conversion); the result is built from those, and each is checked against the host's conversion of
        // the raw partial -- prof.reserved = mismatching partials | first differing stage << 24 | its component << 28
        const size_t nparts = (size_t)job.batch * job.nw * job.per;
        const XYZZ<C>* host_std = job.dev_std ? (const XYZZ<C>*)((const XYZZ<CK>*)job.host_partials + nparts) : nullptr;
        const uint32_t* host_dbg = job.dev_std ? (const uint32_t*)(host_std + nparts) : nullptr;
        auto load = [&](HostXYZZ<C>& r, size_t i) {
            XYZZ<C> ps;
            partial_to_std<C>(ps, host[i]);
            if (host_std) {
                const size_t gi = (size_t)bt * job.nw * job.per + i;
                if (memcmp(&ps, &host_std[gi], sizeof ps) != 0) {
                    if ((job.prof.reserved & 0xffffff) == 0) {
                        constexpr uint32_t DW = partial_dbg_words<CK>();
                        int stage = 5, comp = 0;
                        if constexpr (DW != 0) {
                            std::vector<uint32_t> mine(DW);
                            partial_std_stages<CK>(host[i], mine.data());
                            const uint32_t* dev = host_dbg + gi * DW;
                            using F = typename CK::Fq;
                            constexpr uint32_t L = F29<F>::L, N = F::N, PER = 3 * L + N;
                            for (uint32_t k = 0; k < DW && stage == 5; k++)
                                if (mine[k] != dev[k]) {
                                    comp = (int)(k / PER);
                                    const uint32_t o = k % PER;
                                    stage = o < L ? 1 : o < 2 * L ? 2 : o < 3 * L ? 3 : 4;
                                    fprintf(stderr, "zk: device partial %zu differs: component %d, stage %d (1 norm, 2 product, 3 canonical, 4 packed), word %u\n",
                                            gi, comp, stage, o);
                                    const uint32_t b0 = (uint32_t)comp * PER;
                                    for (uint32_t q = 0; q < PER; q++)
                                        fprintf(stderr, "   [%2u] host %08x dev %08x%s\n", q, mine[b0 + q], dev[b0 + q], mine[b0 + q] != dev[b0 + q] ? "  <--" : "");
                                }
                        }
                        job.prof.reserved |= (stage << 24) | (comp << 28);
                    }
                    job.prof.reserved++;
                }
                ps = host_std[gi];
            }
            to_host<C>(r, ps);
        };
        for (int w = job.nw - 1; w >= 0; w--) {
            // axes form: acc 2^c + (2^lc rows + columns) = (acc 2^(c - lc) + rows) 2^lc + columns -- no extra doublings
            for (int k = 0; k < (job.axes ? job.c - (int)job.log_cols : job.c); k++) xyzz_dbl(htotal);
            if (!job.axes) {
                for (uint32_t i = 0; i < job.per; i++) {
                    load(hp, (size_t)w * job.per + i);
                    xyzz_add(htotal, hp);
                }
                continue;
            }
            // row / column form: per window [row blocks | column blocks] x (W, S); axis total = sum_u W_u + TL sum_u u S_u,
            // window total = 2^lc rows + columns
            const size_t base = (size_t)w * job.per;
            for (int axis = 0; axis < 2; axis++) {
                const uint32_t nb = axis == 0 ? job.row_blocks : job.col_blocks;
                const size_t b0 = base + 2 * (size_t)(axis == 0 ? 0 : job.row_blocks);
                HostXYZZ<C> tot, run, acc;
                load(tot, b0);
                if (nb > 1) {
                    XYZZ<C> inf;
                    xyzz_set_inf(inf);
                    to_host<C>(run, inf);
                    to_host<C>(acc, inf);
                    for (uint32_t u = nb - 1; u >= 1; u--) {
                        load(hp, b0 + 2 * (size_t)u + 1);
                        xyzz_add(run, hp);
                        xyzz_add(acc, run);        // sum_u u S_u
                        load(hp, b0 + 2 * (size_t)u);
                        xyzz_add(tot, hp);
                    }
                    for (uint32_t k = 0; k < job.log_tl; k++) xyzz_dbl(acc);
                    xyzz_add(tot, acc);
                }
                xyzz_add(htotal, tot);
                if (axis == 0)
                    for (uint32_t k = 0; k < job.log_cols; k++) xyzz_dbl(htotal);
            }
        }
        for (int k = 0; k < job.c * job.w0; k++) xyzz_dbl(htotal);
        from_host<C>(total, htotal);
    }
    xyzz_to_jacobian(result, total);
    memcpy((unsigned char*)out_jac + (size_t)bt * out_bytes, &result, out_bytes);
    }
    if (!job.empty) {
        job.prof.host_tail_ms = (float)(now_ms() - t0);
        hipEventElapsedTime(&job.prof.digits_ms, job.ev[0], job.ev[1]);
        hipEventElapsedTime(&job.prof.hist_ms, job.ev[1], job.ev[2]);
        hipEventElapsedTime(&job.prof.scatter_ms, job.ev[2], job.ev[3]);
        hipEventElapsedTime(&job.prof.accumulate_ms, job.ev[3], job.ev[4]);
        job.prof.accumulate_kernel_ms = 0;          // the accumulate kernel of every window group
        for (int gi = 0; gi < job.groups; gi++) {
            float ms = 0;
            hipEventElapsedTime(&ms, job.acc_ev[2 * gi], job.acc_ev[2 * gi + 1]);
            job.prof.accumulate_kernel_ms += ms;
        }
        hipEventElapsedTime(&job.prof.reduce_ms, job.ev[4], job.ev[5]);
        hipEventElapsedTime(&job.prof.total_ms, job.ev[0], job.ev[5]);
        job.prof.total_ms += job.prof.host_tail_ms;
        job.prof.groups = job.groups;
    }
    return ZK_OK;
}

// C: the curve of the ABI call.  CK: the view the kernels compute in.  bases: device array of StoredAffine<CK>.
// Enqueues every kernel of the MSM and the copy of the per-window partial sums on job.stream and returns; no host
// synchronisation (workspace growth aside).
// One window group [w0, w1) of the job: group `gi` of `ng` whose windows start `woff` windows into the job's range -- the whole
// pipeline (sort, accumulate, reduction, copy of the partial sums) on job.stream, over workspaces sized for this group.
template <class C, class CK>
int msm_enqueue_group(MsmJob& job, const StoredAffine<CK>* bases, const Fe<typename C::Fr>* d_scalars, uint64_t n, int mont, const MsmTuning& tu,
                      int c, int nwin, int w0, int w1, int gi, int ng, int woff, int nw_job) {
    hipStream_t st = job.stream;
    {
        if (n >= (1ull << 31)) return ZK_ERR_UNSUPPORTED;
        const int num_cus = job.dc->num_cus > 0 ? job.dc->num_cus : 256;
        MsmShape sh;
        sh.n = (uint32_t)n;
        sh.c = c;
        sh.w0 = w0;
        sh.nwb = w1 - w0;
        sh.batch = job.batch;
        sh.batch_stride = tu.batch_stride;
        sh.nw = sh.nwb * (int)sh.batch;          // the job's windows: those of vector 0, then of vector 1, ...
        sh.nbk = 1u << (c - 1);
        sh.rb = sh.nbk < MSM_RANGE ? sh.nbk : MSM_RANGE;
        sh.nranges = sh.nbk / sh.rb;
        sh.mont = mont;
        sh.pre_n = 0;
        sh.pre_w = 0;
        const bool pre = tu.precomputed;
        const uint64_t n_real = n;
        if (pre) {
            // ONE bucket set over the table [2^(c w)] P_i: the job is an MSM over nwin * n table entries with c-bit "scalars" and a
            // single window; finer ranges keep a region at the size the LDS sort handles (nwin * n / nranges entries)
            if (w0 != 0 || w1 != nwin || n * (uint64_t)nwin >= (1ull << 31)) return ZK_ERR_UNSUPPORTED;
            sh.pre_n = (uint32_t)n;
            sh.pre_w = (uint32_t)nwin;
            n = n * (uint64_t)nwin;
            sh.n = (uint32_t)n;
            sh.w0 = 0;
            sh.nwb = 1;
            sh.nw = (int)sh.batch;
            uint32_t rb = sh.nbk < MSM_RANGE ? sh.nbk : MSM_RANGE;
            while (rb > 8 && sh.nbk / rb < 1024 && (n / (sh.nbk / rb)) > 20000) rb >>= 1;
            sh.rb = rb;
            sh.nranges = sh.nbk / sh.rb;
            job.nw = 1;
        }
        {   // oversize threshold: 2x the mean bucket length + 64 (uniform 2^20 / c=16: mean 32, max ~70 -> none)
            const uint64_t mean = n / sh.nbk;
            sh.big_thresh = tu.big_thresh ? tu.big_thresh : (uint32_t)(2 * mean + 64);
            sh.seg = msm_seg_len(n);
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
            const uint64_t lanes = (uint64_t)num_cus * 4 * 3 * 64;
            const uint64_t items = (uint64_t)nw_all * sh.nbk;
            int sl = 0;
            if (items < 8 * lanes && (n >> 1) / sh.nbk >= 8) sl = 1;
            if (items < lanes && (n >> 2) / sh.nbk >= 8) sl = 2;
            if (pre) {                 // ~nwin * n / nbk entries per bucket: pieces of ~32; halves while a lane would get < 8 buckets
                sl = 0;
                while (sl < 4 && ((n >> (sl + 1)) / sh.nbk) >= 24) sl++;
                if (sl == 0 && items < 8 * lanes && (n >> 1) / sh.nbk >= 8) sl = 1;
            }
            if (tu.split_log >= 0) sl = tu.split_log > 4 ? 4 : tu.split_log;
            sh.split_log = sl;
        }
        const uint32_t nbuckets = (uint32_t)nw_all * sh.nbk;
        const uint32_t nreg = (uint32_t)nw_all * sh.nranges;                  // (window, bucket range) regions
        // scalar blocks of the count / stage kernels: 4096 scalars each at full size; a small input (the later IPA rounds, small
        // keys) would leave all of its digit extraction to a handful of workgroups (72 us for 2^12 scalars in one), so it gets
        // blocks as small as 256 scalars, enough for >= 64 workgroups
        sh.sblk = MSM_SBLK;
        while (sh.sblk > 256 && (n_real + sh.sblk - 1) / sh.sblk < 64) sh.sblk >>= 1;
        // precomputed form: the stage kernel's blocks are the (window, scalar block) pairs: the table must tile into whole blocks
        if (pre && n_real % sh.sblk != 0) return ZK_ERR_UNSUPPORTED;
        const uint32_t nblocks_real = (uint32_t)((n_real + sh.sblk - 1) / sh.sblk);
        const uint32_t nblocks = pre ? nblocks_real * sh.pre_w : nblocks_real;
        if (nreg > 4096 || sh.nranges > (pre ? 1024u : 64u)) return ZK_ERR_UNSUPPORTED;   // LDS tables of those kernels (<= 1024 / 64 ranges per window)
        if ((uint64_t)n * (uint64_t)nw_all >= (1ull << 32)) return ZK_ERR_UNSUPPORTED;   // entry positions are u32 (2^27 points x 16 windows fit)
        // counts | offs | order | wg_total | region_base
        ZK_TRY(ws_get(job.counts, ((size_t)nbuckets * 3 + 2 * (size_t)nreg) * 4));
        uint32_t* counts = (uint32_t*)job.counts.p;
        uint32_t* offs = counts + nbuckets;
        uint32_t* order = offs + nbuckets;
        uint32_t* wg_total = order + nbuckets;
        uint32_t* region_base = wg_total + nreg;
        ZK_TRY(ws_get(job.blockcnt, (size_t)nreg * nblocks * 4));
        uint32_t* blockcnt = (uint32_t*)job.blockcnt.p;
        ZK_TRY(ws_get(job.stage_idx, (size_t)n * nw_all * 4));
        ZK_TRY(ws_get(job.stage_low, (size_t)n * nw_all * 2));
        ZK_TRY(ws_get(job.digits, (size_t)n * nw_all * (pre ? 4 : 2)));   // (the one-bucket-set form: 32-bit codes, up to 2^19 buckets)
        uint16_t* digits = (uint16_t*)job.digits.p;
        uint32_t* stage_idx = (uint32_t*)job.stage_idx.p;
        uint16_t* stage_low = (uint16_t*)job.stage_low.p;
        ZK_TRY(ws_get(job.sorted, (size_t)n * nw_all * 4));
        uint32_t* sorted = (uint32_t*)job.sorted.p;
        ZK_TRY(ws_get(job.buckets, (size_t)nbuckets * sizeof(XYZZ<CK>)));
        XYZZ<CK>* buckets = (XYZZ<CK>*)job.buckets.p;
        XYZZ<CK>* acc_out = buckets;
        if (sh.split_log > 0) {
            ZK_TRY(ws_get(job.subacc, ((size_t)nbuckets << sh.split_log) * sizeof(XYZZ<CK>)));
            acc_out = (XYZZ<CK>*)job.subacc.p;
        }
        // oversized-bucket lists: a bucket above big_thresh yields ceil(cnt / sh.seg) segments
        const size_t max_seg = ((size_t)n / sh.seg + (size_t)n / sh.big_thresh + 2) * nw_all;
        ZK_TRY(ws_get(job.queue, sizeof(MsmQueue) + max_seg * (sizeof(MsmSeg) + 8) + 64));
        ZK_TRY(ws_get(job.seg_out, max_seg * sizeof(XYZZ<CK>)));
        MsmQueue* q = (MsmQueue*)job.queue.p;
        MsmSeg* seg_list = (MsmSeg*)(q + 1);
        uint32_t* big_list = (uint32_t*)(seg_list + max_seg);
        XYZZ<CK>* seg_out = (XYZZ<CK>*)job.seg_out.p;
        XYZZ<CK>* cur = nullptr;   // the per-window partial sums that go to the host
        if (!job.have_events) {
            for (auto& e : job.ev) HIP_TRY(hipEventCreate(&e));
            job.have_events = true;
        }
        // resident waves per SIMD = what the kernel was compiled for (msm_acc_waves); opts.waves_per_simd launches fewer
        unsigned waves_per_simd = (unsigned)msm_acc_waves<CK>();
        if (tu.waves >= 1 && tu.waves <= 8) waves_per_simd = (unsigned)tu.waves;
        hipEvent_t* ev = job.ev;   // [0] begin, [1] counted, [2] staged, [3] sorted, [4] accumulated, [5] reduced, [6] partials on the host
        if (gi == 0) HIP_TRY(hipEventRecord(ev[0], st));
        while (job.acc_ev.size() < 2 * (size_t)ng) {      // one bracket of the accumulate kernel per window group
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            job.acc_ev.push_back(e);
        }
        // ---- sort: partition the digits by (window, bucket range), then counting-sort every region in LDS
        // small problems: fewer lanes, cheaper barriers -- but the stage kernel sets up one lane per range (up to 1024 in the
        // one-bucket-set form: a wide window over few points)
        const unsigned dblk = (sh.sblk >= 4096 || sh.nranges > 256) ? 1024u : 256u;
        if (pre)
            ZK_LAUNCH((msm_digits_pre_kernel<C>), nblocks_real * sh.batch, dblk, (size_t)sh.nranges * 4, st, d_scalars, sh, (uint32_t*)job.digits.p, blockcnt);
        else
            ZK_LAUNCH((msm_digits_kernel<C>), nblocks * sh.batch, dblk, (size_t)sh.nwb * sh.nranges * 4, st, d_scalars, sh, digits, blockcnt);
        auto lanes_for = [](uint32_t items) {   // workgroup size for a scan over `items` values: a power of two in [64, 1024]
            unsigned b = 64;
            while (b < items && b < 1024) b <<= 1;
            return b;
        };
        ZK_LAUNCH((msm_region_scan_kernel<void>), nreg, lanes_for(nblocks), 0, st, blockcnt, nblocks, wg_total);
        ZK_LAUNCH((msm_region_base_kernel<void>), 1, lanes_for(nreg), 0, st, (const uint32_t*)wg_total, nreg, region_base);
        HIP_TRY(hipEventRecord(ev[1], st));
        if (pre)
            ZK_LAUNCH((msm_stage_kernel<uint32_t>), nblocks * (unsigned)nw_all, dblk, (size_t)sh.sblk * 8 + (size_t)(3 * sh.nranges + 1) * 4, st,
                      (const uint32_t*)job.digits.p, sh, (const uint32_t*)blockcnt, (const uint32_t*)wg_total, (const uint32_t*)region_base, nblocks,
                      stage_idx, stage_low);
        else
            ZK_LAUNCH((msm_stage_kernel<uint16_t>), nblocks * (unsigned)nw_all, dblk, (size_t)sh.sblk * 8 + (size_t)(3 * sh.nranges + 1) * 4, st,
                      (const uint16_t*)digits, sh, (const uint32_t*)blockcnt, (const uint32_t*)wg_total, (const uint32_t*)region_base, nblocks,
                      stage_idx, stage_low);
        HIP_TRY(hipEventRecord(ev[2], st));
        // LDS permutation capacity of a sort workgroup: 1.5x the mean region, at most 24576 entries (61 KB of LDS in all,
        // two workgroups per CU); larger regions are sorted in chunks of half that
        uint32_t cap = (uint32_t)((n / sh.nranges) * 3 / 2 + 64);
        if (cap > 24576) cap = 24576;
        cap = (cap + 1) & ~1u;
        // regions up to 4x the mean (large n) are sorted in LDS chunks; anything beyond is a skewed witness's hot region
        const uint64_t cl64 = 4 * (n / sh.nranges) + 4 * (uint64_t)cap;
        const uint32_t chunk_limit = cl64 < 0xffffffffull ? (uint32_t)cl64 : 0xffffffffu;
        // hot regions (skewed witnesses) are histogrammed and scattered by MSM_HOT_SLICES workgroups each, around the sort kernel
        const bool hot_help = nreg <= 1024 && !tu.no_hot_help && n >= 32768;   // a small input's hot region is one sort workgroup's work anyway
        uint32_t* hot_flag = nullptr;
        uint32_t* hot_list = nullptr;
        uint32_t* gcur = nullptr;
        uint32_t hot_slices = (uint32_t)(n >> 16);
        if (hot_slices < 1) hot_slices = 1;
        if (hot_slices > MSM_HOT_SLICES) hot_slices = MSM_HOT_SLICES;
        if (hot_help) {
            ZK_TRY(ws_get(job.hot, ((size_t)nreg + 1 + MSM_HOT_MAX + nbuckets) * 4));
            hot_flag = (uint32_t*)job.hot.p;
            hot_list = hot_flag + nreg;
            gcur = hot_list + 1 + MSM_HOT_MAX;
            HIP_TRY(hipMemsetAsync(counts, 0, (size_t)nbuckets * 4, st));
            ZK_LAUNCH((msm_hot_list_kernel<void>), 1, 1024, 0, st, (const uint32_t*)wg_total, nreg, chunk_limit, hot_flag, hot_list);
            ZK_LAUNCH((msm_hot_kernel<void>), MSM_HOT_MAX * hot_slices, 1024, (size_t)2 * sh.rb * 4, st, (const uint32_t*)stage_idx,
                      (const uint16_t*)stage_low, sh, (const uint32_t*)region_base, (const uint32_t*)wg_total, (const uint32_t*)hot_list, counts,
                      gcur, sorted, 0, hot_slices);
        }
        ZK_LAUNCH((msm_sort_kernel<void>), nreg, n >= 8192 ? 1024u : 256u, (size_t)(4 * sh.rb + 1024 + 258) * 4 + (size_t)cap * 2, st,
                  (const uint32_t*)stage_idx, (const uint16_t*)stage_low, sh, (const uint32_t*)region_base, (const uint32_t*)wg_total, counts,
                  offs, order, sorted, cap, chunk_limit, (const uint32_t*)hot_flag, gcur);
        if (hot_help)
            ZK_LAUNCH((msm_hot_kernel<void>), MSM_HOT_MAX * hot_slices, 1024, (size_t)2 * sh.rb * 4, st, (const uint32_t*)stage_idx,
                      (const uint16_t*)stage_low, sh, (const uint32_t*)region_base, (const uint32_t*)wg_total, (const uint32_t*)hot_list, counts,
                      gcur, sorted, 1, hot_slices);
        HIP_TRY(hipMemsetAsync(q, 0, sizeof(MsmQueue), st));
        HIP_TRY(hipEventRecord(ev[3], st));   // [3] -> [7] brackets the accumulate kernel alone (per group: acc_ev)
        HIP_TRY(hipEventRecord(job.acc_ev[2 * gi], st));
        // ---- persistent accumulate: lanes stream buckets, largest first; oversized buckets go to the cooperative segment
        // kernels (fixed grids over device-side lists, no host round trip)
        const uint32_t ntasks = (((nreg * ((sh.rb + MSM_RANKW - 1) / MSM_RANKW) * MSM_RANKW) << sh.split_log) + MSM_BATCH - 1) / MSM_BATCH;  // batches in the queue
        unsigned acc_grid = (unsigned)num_cus * 4u * waves_per_simd;
        if (acc_grid > ntasks) acc_grid = ntasks;
        ZK_LAUNCH((msm_accumulate_kernel<CK>), acc_grid, 64, 0, st, bases, (const uint32_t*)sorted, (const uint32_t*)offs,
                  (const uint32_t*)counts, (const uint32_t*)order, acc_out, sh, q, seg_list, big_list);
        HIP_TRY(hipEventRecord(ev[7], st));
        HIP_TRY(hipEventRecord(job.acc_ev[2 * gi + 1], st));
        if (sh.split_log > 0)
            ZK_LAUNCH((msm_combine_sub_kernel<CK>), (nbuckets + 63) / 64, 64, 0, st, (const XYZZ<CK>*)acc_out, buckets, nbuckets, sh.split_log);
        const unsigned big_grid = max_seg < 4096 ? (unsigned)max_seg : 4096u;
        ZK_LAUNCH((msm_accumulate_big_kernel<CK>), big_grid, 64, 0, st, bases, (const uint32_t*)sorted, (const MsmQueue*)q,
                  (const MsmSeg*)seg_list, seg_out);
        // (an oversized bucket holds more than big_thresh entries: at most n nw / big_thresh of them exist)
        const size_t max_big = (size_t)n * nw_all / sh.big_thresh + 1;
        ZK_LAUNCH((msm_combine_big_kernel<CK>), (unsigned)(max_big < 2048 ? max_big : 2048), tree_lanes<CK>(), 0, st, (const MsmQueue*)q,
                  (const uint32_t*)big_list, (const uint32_t*)counts, (const XYZZ<CK>*)seg_out, buckets, sh.seg);
        HIP_TRY(hipEventRecord(ev[4], st));
        uint32_t per = 0;
        job.axes = !tu.slice_reduce;
        if (job.axes) {
            // ---- bucket reduction by row / column sums (zk_msm_kernels.h, MsmAxes)
            constexpr uint32_t TL = tree_lanes<CK>();
            MsmAxes A;
            A.nbk = sh.nbk;
            A.nw = (uint32_t)nw_all;
            A.log_cols = (uint32_t)c / 2;                               // ceil((c - 1) / 2)
            A.cols = 1u << A.log_cols;
            A.rows = sh.nbk >> A.log_cols;
            uint32_t K = 2;                                             // buckets per lane: <= ~2 waves per SIMD of lanes, at most 8
            while (K < 8 && 2ull * A.nw * A.nbk / K > 131072) K <<= 1;
            A.k_row = K < A.cols ? K : A.cols;
            A.k_col = K < A.rows ? K : A.rows;
            A.p_row = A.cols / A.k_row;
            A.p_col = A.rows / A.k_col;
            A.row_lanes = A.nw * A.rows * A.p_row;
            A.col_lanes = A.nw * A.cols * A.p_col;
            auto fold_blocks = [&](uint32_t nelem, uint32_t P) {
                const uint32_t tw = P < 16 ? P : 16;
                const uint32_t per_wg = TL / tw;
                return (nelem + per_wg - 1) / per_wg;
            };
            const uint32_t fr = fold_blocks(A.nw * A.rows, A.p_row), fc = fold_blocks(A.nw * A.cols, A.p_col);
            const uint32_t rb = (A.rows + TL - 1) / TL, cb = (A.cols + TL - 1) / TL;
            per = 2 * (rb + cb);
            ZK_TRY(ws_get(job.part_a, ((size_t)A.row_lanes + A.col_lanes) * sizeof(XYZZ<CK>)));
            ZK_TRY(ws_get(job.part_b, ((size_t)A.nw * (A.rows + A.cols) + (size_t)A.nw * per) * sizeof(XYZZ<CK>)));
            XYZZ<CK>* part = (XYZZ<CK>*)job.part_a.p;
            XYZZ<CK>* elem = (XYZZ<CK>*)job.part_b.p;
            cur = elem + (size_t)A.nw * (A.rows + A.cols);
            ZK_LAUNCH((msm_axis_partials_kernel<CK>), (A.row_lanes + A.col_lanes + 63) / 64, 64, 0, st, (const XYZZ<CK>*)buckets, part, A);
            ZK_LAUNCH((msm_axis_fold_kernel<CK>), fr + fc, TL, 0, st, (const XYZZ<CK>*)part, elem, A, fr);
            if (tu.device_partials && ViewBase<CK>::LAZY) {
                // [raw partials | converted partials | conversion stages], copied to the host as one block
                const size_t np = (size_t)A.nw * per;
                const size_t bytes = np * (sizeof(XYZZ<CK>) + sizeof(XYZZ<C>) + 4 * (size_t)partial_dbg_words<CK>());
                ZK_TRY(ws_get(job.part_std, bytes));
                cur = (XYZZ<CK>*)job.part_std.p;
                XYZZ<C>* cur_std = (XYZZ<C>*)(cur + np);
                uint32_t* dbg = (uint32_t*)(cur_std + np);
                ZK_LAUNCH((msm_axis_weighted_kernel<CK, true>), A.nw * (rb + cb), TL, 0, st, (const XYZZ<CK>*)elem, cur, A, rb, cb,
                          (XYZZ<typename ViewBase<CK>::type>*)cur_std, dbg);
                job.dev_std = true;
            } else {
                ZK_LAUNCH((msm_axis_weighted_kernel<CK>), A.nw * (rb + cb), TL, 0, st, (const XYZZ<CK>*)elem, cur, A, rb, cb,
                          (XYZZ<typename ViewBase<CK>::type>*)nullptr, (uint32_t*)nullptr);
            }
            job.row_blocks = rb;
            job.col_blocks = cb;
            job.log_cols = A.log_cols;
            job.log_tl = TL == 256 ? 8u : 7u;
        } else {
            // ---- slice form (ZK_MSM_FLAG_SLICE_REDUCE): running sums + windowed multiplier per slice, then tree-sum the
            // slices of each window until <= 8 remain.  Slice length: 64K lanes = one wave on every SIMD; a rank of a
            // window-sharded MSM owns few windows and gets shorter slices
            uint32_t L = (uint32_t)(((uint64_t)nw_all * sh.nbk) >> 16);
            if (L < 1) L = 1;
            if (L > 8) L = 8;
            if (tu.slice_len >= 1 && tu.slice_len <= 1024) L = tu.slice_len;
            if (L > sh.nbk) L = sh.nbk;
            const uint32_t spw = (sh.nbk + L - 1) / L;                  // slices per window
            const uint32_t pbw = spw / 128 + 2;                          // second ping-pong buffer, points per window
            ZK_TRY(ws_get(job.part_a, (size_t)spw * nw_all * sizeof(XYZZ<CK>)));
            ZK_TRY(ws_get(job.part_b, (size_t)pbw * nw_all * sizeof(XYZZ<CK>)));
            cur = (XYZZ<CK>*)job.part_a.p;
            XYZZ<CK>* nxt = (XYZZ<CK>*)job.part_b.p;
            const uint32_t nslices = spw * (uint32_t)nw_all;
            ZK_LAUNCH((msm_reduce_kernel<CK>), (nslices + 63) / 64, 64, 0, st, (const XYZZ<CK>*)buckets, cur, sh.nbk, L, spw, nslices);
            per = spw;
            while (per > 8) {
                const uint32_t E = per >= 1024 ? 4 : 1;
                const uint32_t chunk = tree_lanes<CK>() * E;
                const uint32_t per_out = (per + chunk - 1) / chunk;
                ZK_LAUNCH((msm_sum_kernel<CK>), (unsigned)nw_all * per_out, tree_lanes<CK>(), 0, st, (const XYZZ<CK>*)cur, nxt, per, per_out, E);
                per = per_out;
                XYZZ<CK>* t = cur;
                cur = nxt;
                nxt = t;
            }
        }
        HIP_TRY(hipEventRecord(ev[5], st));
        HIP_TRY(hipGetLastError());
        const size_t pbytes1 = sizeof(XYZZ<CK>) + (job.dev_std ? sizeof(XYZZ<C>) + 4 * (size_t)partial_dbg_words<CK>() : 0);
        const size_t hbytes = (size_t)nw_all * per * pbytes1;
        const size_t hbytes_job = (ng > 1 ? (size_t)nw_job : (size_t)nw_all) * per * pbytes1;     // (groups: batch == 1, no diagnostic form)
        if (gi == 0 && job.host_cap < hbytes_job) {
            if (job.host_partials) hipHostFree(job.host_partials);
            job.host_partials = nullptr;
            job.host_cap = 0;
            HIP_TRY(hipHostMalloc(&job.host_partials, hbytes_job + 4096, 0));
            job.host_cap = hbytes_job + 4096;
        }
        HIP_TRY(hipMemcpyAsync((unsigned char*)job.host_partials + (size_t)woff * per * pbytes1, cur, hbytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(ev[6], st));
        job.per = per;
        job.empty = false;
        job.alg_bytes += (double)job.batch * (double)n_real * (sizeof(Fe<typename C::Fr>) + sizeof(Affine<C>)) * (double)(w1 - w0) / (double)nwin;
    }
    return ZK_OK;
}


// C: the curve of the ABI call.  CK: the view the kernels compute in.  bases: device array of StoredAffine<CK>.
// Enqueues every kernel of the MSM and the copy of the per-window partial sums on job.stream and returns; no host
// synchronisation (workspace growth aside).
// Window groups (zk_msm_opts.window_group, OFF by default): the windows of a single MSM processed a few at a time, each group
// running the whole pipeline over workspaces sized for a group, all partial sums landing in one host buffer.  The idea (round-2
// shard model: 4 windows of a 2^22-point MSM in 1.82 ms against 8.68 / 4 = 2.17 ms) was that 16 windows keep 268 MB of sorted
// entries + 256 MB of bases live, past the 256 MiB Infinity Cache.  Measured inside one job (profiles/r03_e_window_group_sweep.txt,
// BN254 G1 2^22, ms: all 16 windows 8.20 | 2 groups 8.81 | 4 groups 8.81 | 8 groups 11.35): the accumulate kernels do get faster
// (5.96 -> 5.62 in 4 groups) but every group pays its own sort launches and its own latency-bound bucket reduction, which costs
// more than the cache gives back; with a 0/1-heavy witness it is 2.90 -> 3.95.  So the default is one group; the knob stays.
template <class C, class CK>
int msm_enqueue_impl(MsmJob& job, const StoredAffine<CK>* bases, const Fe<typename C::Fr>* d_scalars, uint64_t n, int mont, const MsmTuning& tu) {
    memset(&job.prof, 0, sizeof job.prof);
    job.dev_std = false;
    job.finish = &msm_finish_impl<C, CK>;
    job.empty = true;
    job.alg_bytes = 0;
    const int c = msm_pick_c(n, tu.window_bits, tu.precomputed);
    const int nwin = msm_windows<C>(c);
    int w0 = 0, w1 = nwin;
    if (!(tu.w0 == 0 && tu.w1 == 0)) {
        w0 = tu.w0;
        w1 = tu.w1;
        if (w0 < 0 || w1 > nwin || w0 > w1) return ZK_ERR_INVALID_ARG;
    }
    job.c = c;
    job.w0 = w0;
    job.nw = w1 - w0;
    job.batch = tu.batch ? tu.batch : 1;
    job.prof.window_bits = c;
    job.prof.windows_total = nwin;
    job.prof.windows_done = w1 - w0;
    job.prof.limb_bits = CK::EXT >= 29 ? 29 : 32;
    job.groups = 1;
    if (n == 0 || w1 <= w0) return ZK_OK;
    int gw = w1 - w0;                                  // windows per group
    if (job.batch == 1 && !tu.precomputed && !tu.device_partials) {
        if (tu.window_group > 0) gw = tu.window_group;
        if (gw < 1) gw = 1;
        if (gw > w1 - w0) gw = w1 - w0;
    }
    const int ng = (w1 - w0 + gw - 1) / gw;
    gw = (w1 - w0 + ng - 1) / ng;                      // even groups
    job.groups = ng;
    for (int gi = 0; gi < ng; gi++) {
        const int a = w0 + gi * gw, b = a + gw < w1 ? a + gw : w1;
        ZK_TRY((msm_enqueue_group<C, CK>(job, bases, d_scalars, n, mont, tu, c, nwin, a, b, gi, ng, a - w0, w1 - w0)));
    }
    return ZK_OK;
}

// every curve runs its bucket arithmetic in the lazy-limb view (Pallas, Vesta, BN254: 9 x 29-bit limbs; BLS12-381: 14 x 28;
// G2 as pairs of those); zk_msm_opts.limb_bits = 32 forces the saturated 32-bit path
template <class C>
constexpr bool has_f29() {
    return C::EXT == 1 || C::EXT == 2;
}
template <class C>
int bases_prepare_run(BasesCopy& bc, uint64_t n) {
    bc.dev29 = nullptr;
    if constexpr (has_f29<C>()) {
        if (n == 0) return ZK_OK;
        void* d = nullptr;
        HIP_TRY(hipMalloc(&d, sizeof(StoredAffine<F29View<C>>) * n));
        ZK_LAUNCH((bases_to29_kernel<C>), (unsigned)((n + 255) / 256), 256, 0, (hipStream_t)0, (const Affine<C>*)bc.dev,
                  (StoredAffine<F29View<C>>*)d, n);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize((hipStream_t)0) != hipSuccess) {
            hipFree(d);
            return ZK_ERR_HIP;
        }
        bc.dev29 = d;
    }
    return ZK_OK;
}

// the table of window multiples for the one-bucket-set form (zk_bases_precompute): nwin * n packed lazy-limb points
template <class C>
int bases_precompute_run(BasesCopy& bc, uint64_t n, int c) {
    if constexpr (has_f29<C>()) {
        const int nwin = msm_windows<C>(c);
        if (n == 0 || n * (uint64_t)nwin >= (1ull << 31)) return ZK_ERR_UNSUPPORTED;
        if (bc.pre) hipFree(bc.pre);
        bc.pre = nullptr;
        void* d = nullptr;
        HIP_TRY(hipMalloc(&d, sizeof(StoredAffine<F29View<C>>) * n * nwin));
        ZK_LAUNCH((bases_precompute_kernel<C>), (unsigned)((n + 63) / 64), 64, 0, (hipStream_t)0, (const Affine<C>*)bc.dev,
                  (StoredAffine<F29View<C>>*)d, (uint32_t)n, (uint32_t)c, (uint32_t)nwin);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize((hipStream_t)0) != hipSuccess) {
            hipFree(d);
            return ZK_ERR_HIP;
        }
        bc.pre = d;
        bc.pre_c = c;
        bc.pre_w = nwin;
        return ZK_OK;
    }
    return ZK_ERR_UNSUPPORTED;
}

template <class C>
int bases_refresh_run(const BasesCopy& bc, uint64_t offset, uint64_t count, hipStream_t st) {
    if constexpr (has_f29<C>()) {
        if (!bc.dev29 || count == 0) return ZK_OK;
        ZK_LAUNCH((bases_to29_kernel<C>), (unsigned)((count + 255) / 256), 256, 0, st, (const Affine<C>*)bc.dev + offset,
                  (StoredAffine<F29View<C>>*)bc.dev29 + offset, count);
        HIP_TRY(hipGetLastError());
    }
    return ZK_OK;
}

// every curve runs its bucket arithmetic in the lazy-limb view unless opts.limb_bits = 32 asks for the saturated words
template <class C>
int msm_enqueue(MsmJob& job, const BasesCopy& bc, const Fe<typename C::Fr>* d_scalars, uint64_t n, int mont, const MsmTuning& tu) {
    if constexpr (has_f29<C>()) {
        if (tu.precomputed) {
            if (!bc.pre || tu.base_offset != 0 || bc.pre_c != msm_pick_c(n, tu.window_bits, true) || bc.pre_w != msm_windows<C>(bc.pre_c))
                return ZK_ERR_INVALID_ARG;
            return msm_enqueue_impl<C, F29View<C>>(job, (const StoredAffine<F29View<C>>*)bc.pre, d_scalars, n, mont, tu);
        }
        if (bc.dev29 && tu.limb_bits != 32)
            return msm_enqueue_impl<C, F29View<C>>(job, (const StoredAffine<F29View<C>>*)bc.dev29 + tu.base_offset, d_scalars, n, mont, tu);
    }
    return msm_enqueue_impl<C, C>(job, (const Affine<C>*)bc.dev + tu.base_offset, d_scalars, n, mont, tu);
}

template <class C>
int fixed_base_run(const Fe<typename C::Fr>* d_scalars, uint64_t n, Affine<C>* d_out, hipStream_t st) {
    ZK_LAUNCH((fixed_base_mul_kernel<C>), (unsigned)((n + 63) / 64), 64, 0, st, d_scalars, d_out, (uint32_t)n);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}
// ark-ec 0.3 FixedBaseMSM (window table + multi_scalar_mul) + batch_normalization_into_affine for one base
template <class C>
int fixed_base_msm_run(DeviceCtx& dc, const Affine<C>& base, const Fe<typename C::Fr>* d_scalars, uint64_t n, int mont, Affine<C>* d_out,
                       hipStream_t st) {
    const uint32_t entries = (uint32_t)fb_windows<C>() * FB_ROW;
    StreamScratch* ss = nullptr;
    ZK_TRY(stream_scratch(dc, st, &ss));   // table and temporaries belong to the caller's stream
    ZK_TRY(ws_get(ss->fb_table, (size_t)entries * sizeof(Affine<C>)));
    ZK_TRY(ws_get(ss->fb_tmp, (size_t)n * sizeof(XYZZ<C>)));
    Affine<C>* table = (Affine<C>*)ss->fb_table.p;
    XYZZ<C>* tmp = (XYZZ<C>*)ss->fb_tmp.p;
    ZK_LAUNCH((fixed_base_table_kernel<C>), (entries + 63) / 64, 64, 0, st, base, table, entries);
    ZK_LAUNCH((fixed_base_msm_kernel<C>), (unsigned)((n + 63) / 64), 64, 0, st, (const Affine<C>*)table, d_scalars, tmp, (uint32_t)n, mont);
    const uint64_t lanes = (n + FB_K - 1) / FB_K;
    ZK_LAUNCH((xyzz_batch_to_affine_kernel<C>), (unsigned)((lanes + 63) / 64), 64, 0, st, (const XYZZ<C>*)tmp, d_out, (uint32_t)n);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}
// ---- GLV decomposition of the fold's shared scalar (host, 64-bit limbs; constants from tools/gen_glv.py) ----
namespace glv {
inline void mul_lo256(uint64_t* r, const uint64_t* a, const uint64_t* b) {   // (a * b) mod 2^256
    uint64_t t[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; i + j < 4; j++) {
            c += (unsigned __int128)a[i] * b[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
    }
    for (int i = 0; i < 4; i++) r[i] = t[i];
}
inline void sub256(uint64_t* r, const uint64_t* a, const uint64_t* b) {
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; i++) {
        unsigned __int128 x = (unsigned __int128)a[i] - b[i] - (uint64_t)br;
        r[i] = (uint64_t)x;
        br = (x >> 64) & 1;
    }
}
inline void neg256(uint64_t* r, const uint64_t* a) {
    const uint64_t z[4] = {0, 0, 0, 0};
    sub256(r, z, a);
}
// c = (k * g) >> 384 for k < 2^256, g < 2^320
inline void mulhi384(uint64_t* c, const uint64_t* k, const uint64_t* g5) {
    uint64_t t[9] = {0};
    for (int i = 0; i < 4; i++) {
        unsigned __int128 cy = 0;
        for (int j = 0; j < 5; j++) {
            cy += (unsigned __int128)k[i] * g5[j] + t[i + j];
            t[i + j] = (uint64_t)cy;
            cy >>= 64;
        }
        t[i + 5] = (uint64_t)cy;
    }
    c[0] = t[6];
    c[1] = t[7];
    c[2] = t[8];
    c[3] = 0;
}
}  // namespace glv

template <class C>
bool glv_decompose(const Fe<typename C::Fr>& u_canonical, FoldScalar& out) {
    if constexpr (Glv<C>::HAS) {
        using G = Glv<C>;
        uint64_t k[4], c1[4], c2[4], t[4], k1[4], k2[4];
        for (int i = 0; i < 4; i++) k[i] = (uint64_t)u_canonical.v[2 * i] | ((uint64_t)u_canonical.v[2 * i + 1] << 32);
        glv::mulhi384(c1, k, G::G1);
        glv::mulhi384(c2, k, G::G2);
        if (G::G1_NEG) glv::neg256(c1, c1);
        if (G::G2_NEG) glv::neg256(c2, c2);
        // k1 = k - c1 a1 - c2 a2 ; k2 = -c1 b1 - c2 b2      (mod 2^256, two's complement; the results are ~129-bit)
        glv::mul_lo256(t, c1, G::A1);
        glv::sub256(k1, k, t);
        glv::mul_lo256(t, c2, G::A2);
        glv::sub256(k1, k1, t);
        glv::mul_lo256(k2, c1, G::B1);
        glv::mul_lo256(t, c2, G::B2);
        {
            unsigned __int128 cy = 0;
            for (int i = 0; i < 4; i++) {
                cy += (unsigned __int128)k2[i] + t[i];
                k2[i] = (uint64_t)cy;
                cy >>= 64;
            }
        }
        glv::neg256(k2, k2);
        out.neg1 = (int)(k1[3] >> 63);
        out.neg2 = (int)(k2[3] >> 63);
        if (out.neg1) glv::neg256(k1, k1);
        if (out.neg2) glv::neg256(k2, k2);
        if (k1[3] | k2[3] | (k1[2] >> 2) | (k2[2] >> 2)) return false;   // not short: fall back to the plain chain
        for (int i = 0; i < 4; i++) {
            out.k1[2 * i] = (uint32_t)k1[i];
            out.k1[2 * i + 1] = (uint32_t)(k1[i] >> 32);
            out.k2[2 * i] = (uint32_t)k2[i];
            out.k2[2 * i + 1] = (uint32_t)(k2[i] >> 32);
        }
        out.glv = 1;
        return true;
    }
    return false;
}

// g[i] <- affine(g[i] + [u] g[i + half]) for i < half (u canonical)
template <class C>
int ipa_fold_bases_run(DeviceCtx& dc, Affine<C>* gens, uint64_t half, const Fe<typename C::Fr>& u_canonical, hipStream_t st) {
    if (half == 0) return ZK_OK;
    if (half >= (1ull << 31)) return ZK_ERR_UNSUPPORTED;
    StreamScratch* ss = nullptr;
    ZK_TRY(stream_scratch(dc, st, &ss));
    ZK_TRY(ws_get(ss->fb_tmp, (size_t)half * sizeof(XYZZ<C>)));
    XYZZ<C>* tmp = (XYZZ<C>*)ss->fb_tmp.p;
    FoldScalar ks;
    memset(&ks, 0, sizeof ks);
    if (!glv_decompose<C>(u_canonical, ks)) {
        memset(&ks, 0, sizeof ks);
        for (int i = 0; i < C::Fr::N && i < 8; i++) ks.k1[i] = u_canonical.v[i];
    }
    ks.top_bit = -1;
    for (int b = 255; b >= 0; b--)
        if (((ks.k1[b >> 5] | ks.k2[b >> 5]) >> (b & 31)) & 1) {
            ks.top_bit = b;
            break;
        }
    ZK_LAUNCH((ipa_fold_bases_kernel<C>), (unsigned)((half + 63) / 64), 64, 0, st, (const Affine<C>*)gens, tmp, (uint32_t)half, ks);
    const uint64_t lanes = (half + FB_K - 1) / FB_K;
    ZK_LAUNCH((xyzz_batch_to_affine_kernel<C>), (unsigned)((lanes + 63) / 64), 64, 0, st, (const XYZZ<C>*)tmp, gens, (uint32_t)half);
    HIP_TRY(hipGetLastError());
    return ZK_OK;
}
// G'[i] = sum_{t < T} W_t G0[t cur + i] for first <= i < first + count <= cur (T = m0 / cur): what r = log2 T literal folds would
// have left in generators [first, first + count) (a rank's share of the survivors, or all of them) (zk_msm_kernels.h, "Several IPA rounds of generator folding at once").  w_dev: the weight vector of
// the fold-free rounds; only W[t cur] is read (the weights do not depend on i).
template <class C>
int ipa_collapse_run(DeviceCtx& dc, const BasesCopy& bc, uint64_t base_n, const Fe<typename C::Fr>* w_dev, uint64_t m0, uint64_t cur,
                     uint64_t first, uint64_t count, Affine<C>* g_out, hipStream_t st) {
    using Fr = typename C::Fr;
    using CK = F29View<C>;
    if (cur == 0 || m0 == 0 || (m0 & (m0 - 1)) || (cur & (cur - 1)) || cur > m0 || m0 > base_n) return ZK_ERR_INVALID_ARG;
    if (m0 >= (1ull << 31) || m0 / cur > 4096 || !bc.dev29) return ZK_ERR_UNSUPPORTED;
    if (first > cur || count > cur - first) return ZK_ERR_INVALID_ARG;
    if (count == 0) return ZK_OK;
    const uint32_t T = (uint32_t)(m0 / cur), m = (uint32_t)cur, i0 = (uint32_t)first, cnt = (uint32_t)count;
    if (T == 1) {
        HIP_TRY(hipMemcpyAsync(g_out, (const Affine<C>*)bc.dev + i0, (size_t)cnt * sizeof(Affine<C>), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        return ZK_OK;
    }
    int c = 2;                           // per output and window: T additions into the buckets + 2 per bucket to reduce them
    {
        uint64_t best = ~0ull;
        for (int cc = 2; cc <= 8; cc++) {
            const uint64_t cost = ((uint64_t)T + (1ull << cc)) * (uint64_t)msm_windows<C>(cc);
            if (cost < best) {
                best = cost;
                c = cc;
            }
        }
    }
    const uint32_t nbk = 1u << (c - 1);
    const uint32_t nwin = (uint32_t)msm_windows<C>(c);
    StreamScratch* ss = nullptr;
    ZK_TRY(stream_scratch(dc, st, &ss));
    const size_t list_words = (size_t)nwin * nbk + 1 + (size_t)nwin * T;
    ZK_TRY(ws_get(ss->poly_a, (size_t)T * sizeof(Fe<Fr>) + list_words * 4 + 64));
    ZK_TRY(ws_get(ss->poly_b, (size_t)nwin * cnt * sizeof(XYZZ<CK>)));
    ZK_TRY(ws_get(ss->fb_tmp, (size_t)cnt * sizeof(XYZZ<C>)));
    Fe<Fr>* d_w = (Fe<Fr>*)ss->poly_a.p;
    uint32_t* d_off = (uint32_t*)(d_w + T);
    uint32_t* d_ent = d_off + (size_t)nwin * nbk + 1;
    XYZZ<CK>* part = (XYZZ<CK>*)ss->poly_b.p;
    XYZZ<C>* tmp = (XYZZ<C>*)ss->fb_tmp.p;
    ZK_LAUNCH((ipa_gather_weights_kernel<C>), (T + 255) / 256, 256, 0, st, w_dev, (uint64_t)m, T, d_w);
    std::vector<Fe<Fr>> wt(T);
    HIP_TRY(hipMemcpyAsync(wt.data(), d_w, (size_t)T * sizeof(Fe<Fr>), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // signed c-bit digits of every weight (the carry method of msm_digits_kernel), then the bucket lists of every window
    auto bits = [&](const Fe<Fr>& x, int start) -> uint32_t {
        uint64_t v = 0;
        const int idx = start >> 5, sh = start & 31;
        if (idx < Fr::N) v = x.v[idx];
        if (idx + 1 < Fr::N) v |= (uint64_t)x.v[idx + 1] << 32;
        return (uint32_t)(v >> sh) & ((1u << c) - 1);
    };
    std::vector<int32_t> dig((size_t)nwin * T);
    std::vector<uint32_t> off((size_t)nwin * nbk + 1, 0), ent((size_t)nwin * T);
    for (uint32_t t = 0; t < T; t++) {
        uint32_t carry = 0;
        for (uint32_t w = 0; w < nwin; w++) {
            const uint32_t raw = bits(wt[t], (int)(w * c)) + carry;
            const bool neg = raw > nbk;
            const uint32_t mag = neg ? (1u << c) - raw : raw;
            carry = neg ? 1u : 0u;
            dig[(size_t)w * T + t] = neg ? -(int32_t)mag : (int32_t)mag;
            if (mag) off[(size_t)w * nbk + (mag - 1) + 1]++;
        }
    }
    for (size_t k = 1; k < off.size(); k++) off[k] += off[k - 1];
    {
        std::vector<uint32_t> pos(off.begin(), off.end() - 1);
        for (uint32_t w = 0; w < nwin; w++)
            for (uint32_t t = 0; t < T; t++) {
                const int32_t d = dig[(size_t)w * T + t];
                if (d == 0) continue;
                const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
                ent[pos[(size_t)w * nbk + (mag - 1)]++] = t | (d < 0 ? 0x80000000u : 0u);
            }
    }
    HIP_TRY(hipMemcpyAsync(d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_ent, ent.data(), ent.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));   // the host vectors go out of scope below
    const uint64_t lanes = (uint64_t)nwin * cnt;
    ZK_LAUNCH((ipa_collapse_window_kernel<CK>), (unsigned)((lanes + 63) / 64), 64, 0, st, (const StoredAffine<CK>*)bc.dev29,
              (const uint32_t*)d_off, (const uint32_t*)d_ent, part, m, i0, cnt, nbk, nwin);
    ZK_LAUNCH((ipa_collapse_horner_kernel<C>), (cnt + 63) / 64, 64, 0, st, (const XYZZ<CK>*)part, tmp, cnt, (uint32_t)c, nwin);
    const uint64_t nl = ((uint64_t)cnt + FB_K - 1) / FB_K;
    ZK_LAUNCH((xyzz_batch_to_affine_kernel<C>), (unsigned)((nl + 63) / 64), 64, 0, st, (const XYZZ<C>*)tmp, g_out, cnt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));   // g_out is complete on return: the caller adopts it as a bases handle next (any stream)
    return ZK_OK;
}
}  // namespace zk
