// C ABI implementation (include/zkcp_amd.h): argument checks, dispatch over curve / field / device, the MSM job pool
// (submit -> collect) and the host-side helpers.  The launch sequences live in zk_msm.inl / zk_ntt.inl (one TU per
// curve / field).  Compiled with `hipcc -x hip --offload-arch=gfx950` into libzkcp_amd.so.  (The CPU test tier
// compiles the same files against tests/emu/emu_hip.h; see zk_rt.h.)
#include "zk_internal.h"
#include "zk_msm_decl.h"
#include "zk_ntt_decl.h"
#include "zkcp_amd_prover.h"

#include <thread>

using namespace zk;

namespace zk {
Ctx g;
int msm_pick_c(uint64_t n, int requested, bool pre) {
    // digits are stored as u16 codes; the precomputed-table form (one bucket set, 32-bit codes) goes up to 2^19 buckets
    const int cmax = pre ? 20 : 16;
    if (requested > 0) return requested < 2 ? 2 : (requested > cmax ? cmax : requested);
    int l = 0;
    while ((1ull << l) < n) l++;
    // one bucket set over the table of window multiples: the windows no longer pay a bucket reduction each, so the window can
    // grow until the ONE reduction (2^(c-1) buckets) costs what the 16 x 2^15 of the plain form do -- 13 windows instead of 16
    // at 2^20 points (tools/r3_pre.sh: measured)
    if (pre && l >= 17) return l >= 20 ? 20 : l;
    // Measured on one MI355X (profiles/r02_g_msm_size_sweep_vesta.txt): below 2^20 points the fixed costs decide -- the bucket
    // reduction's dependent additions, the host Horner (c doublings per window), the sort launches -- not the additions per
    // point, so the rule "2^5 points per bucket" (c = log2 n - 4) of round 1 was 20 - 45 % slow from 2^15 to 2^19.
    if (l >= 16) return 16;
    if (l == 15) return 15;
    if (l == 14) return 10;
    return 8;
}

// the scratch set of a caller stream: found by stream, created on first use, least-recently-used one recycled (after its
// stream has drained) beyond ZK_MAX_STREAM_SCRATCH streams.  Called with dc.mu held.
int stream_scratch(DeviceCtx& dc, hipStream_t st, StreamScratch** out) {
    for (auto& s : dc.scratch)
        if (s->stream == st) {
            s->stamp = ++dc.scratch_stamp;
            *out = s.get();
            return ZK_OK;
        }
    if ((int)dc.scratch.size() < ZK_MAX_STREAM_SCRATCH) {
        dc.scratch.emplace_back(new StreamScratch());
        StreamScratch* s = dc.scratch.back().get();
        s->stream = st;
        s->stamp = ++dc.scratch_stamp;
        *out = s;
        return ZK_OK;
    }
    StreamScratch* victim = dc.scratch[0].get();
    for (auto& s : dc.scratch)
        if (s->stamp < victim->stamp) victim = s.get();
    HIP_TRY(hipStreamSynchronize(victim->stream));   // its last user may still be running
    victim->stream = st;
    victim->stamp = ++dc.scratch_stamp;
    *out = victim;
    return ZK_OK;
}
}  // namespace zk

namespace {
// R1CS matrices of the circuits in use (zk_r1cs_matrix_upload): fixed per circuit, resident on the home device
std::mutex g_mat_mu;
std::map<uint64_t, R1csMatrix> g_mats;
uint64_t g_next_mat = 1;
void free_matrix(R1csMatrix& m) {
    for (void* p : {m.row_ptr, m.col, m.val, m.long_rows})
        if (p) hipFree(p);
    m = R1csMatrix();
}

template <class C>
void jac_to_xyzz(XYZZ<C>& r, const Jacobian<C>& j) {
    if (fe_is_zero(j.z)) {
        xyzz_set_inf(r);
        return;
    }
    r.x = j.x;
    r.y = j.y;
    fe_sqr(r.zz, j.z);
    fe_mul(r.zzz, r.zz, j.z);
}

int point_add_host(zk_curve_t c, const void* ja, const void* jb, void* jout) {
    CURVE_SWITCH(c, {
        Jacobian<C> a, b, r;
        memcpy(&a, ja, 3 * 4 * coord_words<C>());
        memcpy(&b, jb, 3 * 4 * coord_words<C>());
        XYZZ<C> xa, xb;
        jac_to_xyzz(xa, a);
        jac_to_xyzz(xb, b);
        xyzz_add(xa, xb);
        xyzz_to_jacobian(r, xa);
        memcpy(jout, &r, 3 * 4 * coord_words<C>());
    });
    return ZK_OK;
}

MsmTuning tuning_from(const zk_msm_opts* o) {
    MsmTuning t;
    if (!o) return t;
    t.window_bits = o->window_bits;
    t.w0 = o->window_begin;
    t.w1 = o->window_end;
    t.limb_bits = o->limb_bits;
    t.split_log = o->split_log_plus1 > 0 ? o->split_log_plus1 - 1 : -1;
    t.slice_len = o->slice_len > 0 ? (uint32_t)o->slice_len : 0;
    t.big_thresh = o->big_threshold > 0 ? (uint32_t)o->big_threshold : 0;
    t.waves = o->waves_per_simd;
    t.no_hot_help = (o->flags & ZK_MSM_FLAG_NO_HOT_HELP) != 0;
    t.slice_reduce = (o->flags & ZK_MSM_FLAG_SLICE_REDUCE) != 0;
    t.precomputed = (o->flags & ZK_MSM_FLAG_PRECOMPUTED) != 0;
    t.device_partials = (o->flags & ZK_MSM_FLAG_DEVICE_PARTIALS) != 0;
    t.base_offset = o->base_offset > 0 ? (uint64_t)o->base_offset : 0;
    t.window_group = o->window_group > 0 ? o->window_group : 0;
    return t;
}

void free_job(MsmJob& j) {
    for (DevBuf* b : {&j.scalars_in, &j.hot, &j.counts, &j.digits, &j.blockcnt, &j.stage_idx, &j.stage_low, &j.queue, &j.seg_out, &j.subacc, &j.sorted,
                      &j.buckets, &j.part_a, &j.part_b, &j.part_std})
        ws_free(*b);
    if (j.host_partials) hipHostFree(j.host_partials);
    j.host_partials = nullptr;
    j.host_cap = 0;
    if (j.have_events) {
        for (auto& e : j.ev) hipEventDestroy(e);
        j.have_events = false;
    }
    for (auto& e : j.acc_ev) hipEventDestroy(e);
    j.acc_ev.clear();
    j.busy = false;
}

void stop_worker(DeviceCtx& dc);
void free_device(DeviceCtx& dc) {
    stop_worker(dc);
    hipSetDevice(dc.device);
    ws_free(dc.batch_in);
    hipDeviceSynchronize();
    for (auto& kv : dc.tw) hipFree(kv.second.dev);
    dc.tw.clear();
    dc.tw_bytes = 0;
    ws_free(dc.pow_tbl);
    ws_free(dc.scratch_in);
    for (auto& s : dc.scratch) {
        ws_free(s->ntt_tmp);
        ws_free(s->fb_table);
        ws_free(s->fb_tmp);
        ws_free(s->poly_a);
        ws_free(s->poly_b);
        ws_free(s->poly_tot);
        if (s->pinned) hipHostFree(s->pinned);
        if (s->pinned_ev) hipEventDestroy(s->pinned_ev);
        s->pinned = nullptr;
        s->pinned_ev = nullptr;
    }
    dc.scratch.clear();
    for (auto& j : dc.jobs) free_job(j);
    for (auto& e : dc.ntt_ev_pool) hipEventDestroy(e);
    dc.ntt_ev_pool.clear();
    dc.ntt_ev_used = 0;
    for (auto& s : dc.side)
        if (s) {
            hipStreamDestroy(s);
            s = nullptr;
        }
    for (auto& s : dc.submit_streams)
        if (s) {
            hipStreamDestroy(s);
            s = nullptr;
        }
    if (dc.own) hipStreamDestroy(dc.own);
    dc.own = nullptr;
    if (dc.fork_ev) hipEventDestroy(dc.fork_ev);
    dc.fork_ev = nullptr;
    for (auto& e : dc.join_ev)
        if (e) {
            hipEventDestroy(e);
            e = nullptr;
        }
}

// the device that owns a device pointer; the home device when the runtime cannot tell (or under the emulator)
DeviceCtx& device_of(const void* p) {
    if (g.devs.size() > 1 && p) {
#if !defined(ZK_EMU)
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, p) == hipSuccess) {
            for (auto& d : g.devs)
                if (d->device == attr.device) return *d;
        } else {
            (void)hipGetLastError();
        }
#endif
    }
    return *g.devs[0];
}

int ensure_lib_streams(DeviceCtx& dc) {
    if (!dc.side[0]) {
        for (auto& s : dc.side) HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (auto& s : dc.submit_streams) HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&dc.own, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&dc.fork_ev, hipEventDisableTiming));
        for (auto& e : dc.join_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    return ZK_OK;
}

// take a free job slot of the device (dc.mu held).  All busy -> ZK_ERR_BUSY: the caller must collect first.
int job_acquire(DeviceCtx& dc, MsmJob** out) {
    for (auto& j : dc.jobs)
        if (!j.busy) {
            j.dc = &dc;
            j.busy = true;
            *out = &j;
            return ZK_OK;
        }
    return ZK_ERR_BUSY;
}

// where the scalars of a submission are
enum ScalarSrc { SRC_LOCAL = 0, SRC_HOST = 1, SRC_PEER = 2 };

// enqueue one MSM on device dc / stream st (dc.mu held, thread bound to dc.device).  Scalars that are not already in this
// device's memory are staged into a buffer the JOB owns, so MSMs in flight never share a staging area.
int submit_on(DeviceCtx& dc, zk_curve_t c, const BasesEntry& be, const void* scalars, ScalarSrc kind, int peer_device, uint64_t n, int mont,
              const MsmTuning& tu, hipStream_t st, MsmJob** out) {
    MsmJob* job = nullptr;
    ZK_TRY(job_acquire(dc, &job));
    job->stream = st;
    job->curve = (int)c;
    int status = ZK_OK;
    const void* d_scalars = scalars;
    auto stage = [&]() -> int {
        if (kind == SRC_LOCAL || n == 0) return ZK_OK;
        ZK_TRY(ws_get(job->scalars_in, 32 * n));
        d_scalars = job->scalars_in.p;
        if (kind == SRC_HOST) {
            HIP_TRY(hipMemcpyAsync(job->scalars_in.p, scalars, 32 * n, hipMemcpyHostToDevice, st));
        } else {
#if defined(ZK_EMU)
            HIP_TRY(hipMemcpyAsync(job->scalars_in.p, scalars, 32 * n, hipMemcpyDeviceToDevice, st));
#else
            HIP_TRY(hipMemcpyPeerAsync(job->scalars_in.p, dc.device, scalars, peer_device, 32 * n, st));
#endif
        }
        return ZK_OK;
    };
    status = stage();
    (void)peer_device;
    const BasesCopy& bc = be.per_dev[dc.index];
    if (status == ZK_OK) {
        status = ZK_ERR_INVALID_ARG;
        switch (c) {
            case ZK_PALLAS: status = msm_enqueue<Pallas>(*job, bc, (const Fe<Pallas::Fr>*)d_scalars, n, mont, tu); break;
            case ZK_VESTA: status = msm_enqueue<Vesta>(*job, bc, (const Fe<Vesta::Fr>*)d_scalars, n, mont, tu); break;
            case ZK_BN254_G1: status = msm_enqueue<Bn254G1>(*job, bc, (const Fe<Bn254G1::Fr>*)d_scalars, n, mont, tu); break;
            case ZK_BLS12_381_G1: status = msm_enqueue<Bls381G1>(*job, bc, (const Fe<Bls381G1::Fr>*)d_scalars, n, mont, tu); break;
            case ZK_BN254_G2: status = msm_enqueue<Bn254G2>(*job, bc, (const Fe<Bn254G2::Fr>*)d_scalars, n, mont, tu); break;
            case ZK_BLS12_381_G2: status = msm_enqueue<Bls381G2>(*job, bc, (const Fe<Bls381G2::Fr>*)d_scalars, n, mont, tu); break;
            default: break;
        }
    }
    if (status != ZK_OK) {
        job->busy = false;
        return status;
    }
    *out = job;
    return ZK_OK;
}

int collect_job(MsmJob& job, void* out) {
    hipSetDevice(job.dc->device);
    const int status = job.finish(job, out);
    {
        std::lock_guard<std::mutex> lk(g.mu);
        g.prof = job.prof;
        if (!job.empty) {
            g.totals.msms += job.batch;
            g.totals.launches++;
            g.totals.accumulate_kernel_ms += job.prof.accumulate_kernel_ms;
            g.totals.accumulate_ms += job.prof.accumulate_ms;
            g.totals.sort_ms += job.prof.digits_ms + job.prof.hist_ms + job.prof.scatter_ms;
            g.totals.reduce_ms += job.prof.reduce_ms;
            g.totals.host_tail_ms += job.prof.host_tail_ms;
            g.totals.device_ms += job.prof.total_ms - job.prof.host_tail_ms;
            g.totals.algorithmic_bytes += job.alg_bytes;
        }
    }
    std::lock_guard<std::mutex> lk(job.dc->mu);
    job.busy = false;
    return status;
}

int find_bases(uint64_t handle, zk_curve_t c, uint64_t n, const BasesEntry** out, const zk_msm_opts* opts = nullptr) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    auto it = g.bases.find(handle);
    if (it == g.bases.end()) return ZK_ERR_BAD_HANDLE;
    const uint64_t off = opts && opts->base_offset > 0 ? (uint64_t)opts->base_offset : 0;
    if (opts && opts->base_offset < 0) return ZK_ERR_INVALID_ARG;
    if (it->second.curve != (int)c || n > it->second.n || off > it->second.n - n) return ZK_ERR_INVALID_ARG;
    *out = &it->second;   // entries are only erased by zk_bases_free, which the caller must not race with a running MSM
    return ZK_OK;
}

// ---- persistent per-device workers (multi-device processes)
void worker_loop(DeviceCtx* dc) {
    hipSetDevice(dc->device);
    DeviceWorker& w = *dc->worker;
    for (;;) {
        std::packaged_task<int()> task;
        {
            std::unique_lock<std::mutex> lk(w.mu);
            w.cv.wait(lk, [&] { return w.stop || !w.q.empty(); });
            if (w.q.empty()) return;     // stop requested and nothing left
            task = std::move(w.q.front());
            w.q.pop_front();
        }
        task();
    }
}
std::future<int> run_on_device(DeviceCtx& dc, std::function<int()> fn) {
    std::packaged_task<int()> task(std::move(fn));
    std::future<int> fut = task.get_future();
    if (!dc.worker) {      // single-device process: inline
        task();
        return fut;
    }
    {
        std::lock_guard<std::mutex> lk(dc.worker->mu);
        dc.worker->q.push_back(std::move(task));
    }
    dc.worker->cv.notify_one();
    return fut;
}
void stop_worker(DeviceCtx& dc) {
    if (!dc.worker) return;
    {
        std::lock_guard<std::mutex> lk(dc.worker->mu);
        dc.worker->stop = true;
    }
    dc.worker->cv.notify_one();
    if (dc.worker->th.joinable()) dc.worker->th.join();
    dc.worker.reset();
}

size_t jac_bytes(zk_curve_t c) {
    size_t w = 0;
    switch (c) {
        case ZK_PALLAS: w = coord_words<Pallas>(); break;
        case ZK_VESTA: w = coord_words<Vesta>(); break;
        case ZK_BN254_G1: w = coord_words<Bn254G1>(); break;
        case ZK_BLS12_381_G1: w = coord_words<Bls381G1>(); break;
        case ZK_BN254_G2: w = coord_words<Bn254G2>(); break;
        default: w = coord_words<Bls381G2>(); break;
    }
    return (size_t)3 * 4 * w;
}

// One MSM over all devices of the process: device d sums windows [W d / G, W (d + 1) / G) from its own resident copy of
// the bases; the Jacobian partial sums are added on the host (EC addition is not an RCCL reduction; the partial is 96 -
// 288 bytes and is already on the host after the job's tail).  `src` is where the scalars are: a host pointer (src_dc ==
// nullptr; every device copies them over its own PCIe link) or a device pointer owned by src_dc (peers copy over xGMI).
// The shares run on the devices' persistent workers; the calling thread takes device 0's.
int msm_fanout(zk_curve_t c, const BasesEntry& be, const void* src, DeviceCtx* src_dc, hipStream_t src_stream, uint64_t n, int mont,
               const MsmTuning& tu_in, void* out) {
    const int G = (int)g.devs.size();
    int nwin = 0;
    CURVE_SWITCH(c, nwin = msm_windows<C>(msm_pick_c(n, tu_in.window_bits, tu_in.precomputed)));
    const size_t pbytes = jac_bytes(c);
    std::vector<std::vector<unsigned char>> parts(G, std::vector<unsigned char>(pbytes));
    hipEvent_t ready = nullptr;
    if (src_dc) {   // peers must not read the scalars before the caller's stream has produced them
        hipSetDevice(src_dc->device);
        std::lock_guard<std::mutex> lk(src_dc->mu);
        ZK_TRY(ensure_lib_streams(*src_dc));
        ready = src_dc->fork_ev;
        HIP_TRY(hipEventRecord(ready, src_stream));
    }
    auto work = [&](int d) -> int {
        DeviceCtx& dc = *g.devs[d];
        MsmTuning tu = tu_in;
        tu.w0 = nwin * d / G;
        tu.w1 = nwin * (d + 1) / G;
        MsmJob* job = nullptr;
        auto run = [&]() -> int {
            ZK_TRY(bind_device(dc));
            std::lock_guard<std::mutex> lk(dc.mu);
            if (tu.w0 == tu.w1) {   // more devices than windows: this one contributes the identity
                tu.w0 = tu.w1 = 0;
                return submit_on(dc, c, be, nullptr, SRC_LOCAL, 0, 0, mont, tu, nullptr, &job);
            }
            ZK_TRY(ensure_lib_streams(dc));
            if (!src_dc) return submit_on(dc, c, be, src, SRC_HOST, 0, n, mont, tu, dc.own, &job);
            if (&dc == src_dc) return submit_on(dc, c, be, src, SRC_LOCAL, 0, n, mont, tu, src_stream, &job);
            HIP_TRY(hipStreamWaitEvent(dc.own, ready, 0));
            return submit_on(dc, c, be, src, SRC_PEER, src_dc->device, n, mont, tu, dc.own, &job);
        };
        int st_ = run();
        if (st_ == ZK_OK) st_ = collect_job(*job, parts[d].data());
        return st_;
    };
    std::vector<std::future<int>> fut;
    for (int d = 1; d < G; d++) fut.push_back(run_on_device(*g.devs[d], [&, d] { return work(d); }));
    int status = work(0);
    for (auto& f : fut) {
        const int s2 = f.get();
        if (status == ZK_OK) status = s2;
    }
    ZK_TRY(status);
    memcpy(out, parts[0].data(), pbytes);
    for (int d = 1; d < G; d++) ZK_TRY(point_add_host(c, out, parts[d].data(), out));
    hipSetDevice(g.devs[0]->device);
    return ZK_OK;
}

bool whole_msm(const MsmTuning& tu) { return tu.w0 == 0 && tu.w1 == 0; }

int init_devices_locked(int n, const int* ids) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ZK_ERR_NO_DEVICE;
    if (n <= 0 || n > ndev || !ids) return ZK_ERR_INVALID_ARG;
    for (int i = 0; i < n; i++) {
        if (ids[i] < 0 || ids[i] >= ndev) return ZK_ERR_INVALID_ARG;
        for (int j = 0; j < i; j++)
            if (ids[j] == ids[i]) return ZK_ERR_INVALID_ARG;
    }
    if (g.inited) {
        if ((int)g.devs.size() != n) return ZK_ERR_INVALID_ARG;
        for (int i = 0; i < n; i++)
            if (g.devs[i]->device != ids[i]) return ZK_ERR_INVALID_ARG;
        return ZK_OK;
    }
    std::vector<std::unique_ptr<DeviceCtx>> devs;
    for (int i = 0; i < n; i++) {
        if (hipSetDevice(ids[i]) != hipSuccess) return ZK_ERR_NO_DEVICE;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ids[i]) != hipSuccess) return ZK_ERR_NO_DEVICE;
        std::unique_ptr<DeviceCtx> dc(new DeviceCtx());
        dc->device = ids[i];
        dc->index = i;
        dc->num_cus = prop.multiProcessorCount;
        if (i == 0) {
#if defined(ZK_EMU)
            snprintf(g.info, sizeof g.info, "emu %.200s x%d", prop.name, n);
#else
            snprintf(g.info, sizeof g.info, "hip %.60s %.120s cu=%d x%d", prop.gcnArchName, prop.name, prop.multiProcessorCount, n);
#endif
        }
        devs.push_back(std::move(dc));
    }
#if !defined(ZK_EMU)
    for (int i = 0; i < n; i++)   // peers copy scalars to each other over xGMI
        for (int j = 0; j < n; j++)
            if (i != j) {
                hipSetDevice(ids[i]);
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, ids[i], ids[j]) == hipSuccess && can) {
                    if (hipDeviceEnablePeerAccess(ids[j], 0) != hipSuccess) (void)hipGetLastError();   // already enabled is fine
                }
            }
#endif
    hipSetDevice(ids[0]);
    g.devs = std::move(devs);
    if (n > 1)
        for (auto& dc : g.devs) {
            dc->worker.reset(new DeviceWorker());
            dc->worker->th = std::thread(worker_loop, dc.get());
        }
    memset(&g.ntt_opts, 0, sizeof g.ntt_opts);
    memset(&g.totals, 0, sizeof g.totals);
    g.ntt_profile = false;
    g.inited = true;
    return ZK_OK;
}
}  // namespace

// ====================================================================== exported C ABI
extern "C" {

#define API __attribute__((visibility("default")))

API const char* zk_strerror(int s) {
    switch (s) {
        case ZK_OK: return "ok";
        case ZK_ERR_INVALID_ARG: return "invalid argument";
        case ZK_ERR_NOT_INITIALIZED: return "zk_init has not been called";
        case ZK_ERR_NO_DEVICE: return "no usable MI355X / HIP device (there is no CPU fallback)";
        case ZK_ERR_HIP: return "HIP runtime error";
        case ZK_ERR_OOM: return "out of device memory";
        case ZK_ERR_UNSUPPORTED: return "unsupported size or curve";
        case ZK_ERR_BAD_HANDLE: return "unknown bases handle or MSM ticket";
        case ZK_ERR_BUSY: return "too many MSMs in flight on this device: collect one first";
        default: return "unknown status";
    }
}

API int zk_init(int device_id) {
    std::lock_guard<std::mutex> lk(g.mu);
    return init_devices_locked(1, &device_id);
}
API int zk_init_devices(int n_devices, const int* device_ids) {
    std::lock_guard<std::mutex> lk(g.mu);
    return init_devices_locked(n_devices, device_ids);
}
API int zk_device_count(void) {
    std::lock_guard<std::mutex> lk(g.mu);
    return g.inited ? (int)g.devs.size() : 0;
}

API int zk_shutdown(void) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.inited) return ZK_OK;
    for (auto& kv : g.bases)
        for (size_t d = 0; d < kv.second.per_dev.size(); d++) {
            hipSetDevice(g.devs[d]->device);
            hipDeviceSynchronize();
            if (kv.second.per_dev[d].owned) hipFree(kv.second.per_dev[d].dev);
            if (kv.second.per_dev[d].dev29) hipFree(kv.second.per_dev[d].dev29);
            if (kv.second.per_dev[d].pre) hipFree(kv.second.per_dev[d].pre);
        }
    g.bases.clear();
    g.tickets.clear();
    {
        std::lock_guard<std::mutex> lkm(g_mat_mu);
        if (!g.devs.empty()) hipSetDevice(g.devs[0]->device);
        for (auto& kv : g_mats) free_matrix(kv.second);
        g_mats.clear();
    }
    for (auto& dc : g.devs) free_device(*dc);
    g.devs.clear();
    g.inited = false;
    return ZK_OK;
}

API int zk_backend_info(char* buf, uint64_t buflen) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!buf || buflen == 0) return ZK_ERR_INVALID_ARG;
    snprintf(buf, (size_t)buflen, "%s", g.info);
    return ZK_OK;
}

API int zk_field_limbs64(zk_field_t f) {
    FIELD_SWITCH(f, return F::N / 2);
    return ZK_ERR_INVALID_ARG;
}
API int zk_curve_base_limbs64(zk_curve_t c) {
    CURVE_SWITCH(c, return coord_words<C>() / 2);
    return ZK_ERR_INVALID_ARG;
}
API int zk_curve_scalar_field(zk_curve_t c) {
    switch (c) {
        case ZK_PALLAS: return ZK_FQ_PALLAS;
        case ZK_VESTA: return ZK_FP_PALLAS;
        case ZK_BN254_G1: return ZK_FR_BN254;
        case ZK_BLS12_381_G1: return ZK_FR_BLS12_381;
        case ZK_BN254_G2: return ZK_FR_BN254;
        case ZK_BLS12_381_G2: return ZK_FR_BLS12_381;
        default: return ZK_ERR_INVALID_ARG;
    }
}
API int zk_msm_window_bits(zk_curve_t c, uint64_t n, int requested) {
    CURVE_SWITCH(c, (void)sizeof(C); return msm_pick_c(n, requested));
    return ZK_ERR_INVALID_ARG;
}
API int zk_msm_window_count(zk_curve_t c, uint64_t n, int window_bits) {
    CURVE_SWITCH(c, return msm_windows<C>(msm_pick_c(n, window_bits)));
    return ZK_ERR_INVALID_ARG;
}

// bases: one resident copy per device of the process.  `src_dev` >= 0: `src` is a device pointer on that device index
// (adopted there without a copy, copied to the peers); -1: `src` is host memory.
static int bases_install(zk_curve_t c, const void* src, int src_dev, uint64_t n, uint64_t* handle_out) {
    size_t esz = 0;
    CURVE_SWITCH(c, esz = sizeof(Affine<C>));
    BasesEntry be;
    be.curve = (int)c;
    be.n = n;
    be.per_dev.resize(g.devs.size());
    auto cleanup = [&]() {
        for (size_t d = 0; d < be.per_dev.size(); d++) {
            hipSetDevice(g.devs[d]->device);
            if (be.per_dev[d].owned && be.per_dev[d].dev) hipFree(be.per_dev[d].dev);
            if (be.per_dev[d].dev29) hipFree(be.per_dev[d].dev29);
        }
        hipSetDevice(g.devs[0]->device);
    };
    for (size_t d = 0; d < g.devs.size(); d++) {
        DeviceCtx& dc = *g.devs[d];
        BasesCopy& bc = be.per_dev[d];
        int st = bind_device(dc);
        if (st == ZK_OK) {
            if ((int)d == src_dev) {
                bc.dev = const_cast<void*>(src);
                bc.owned = false;
            } else {
                hipError_t e = hipMalloc(&bc.dev, esz * (n ? n : 1));
                if (e != hipSuccess) {
                    bc.dev = nullptr;
                    st = e == hipErrorOutOfMemory ? ZK_ERR_OOM : ZK_ERR_HIP;
                } else {
                    bc.owned = true;
                    if (n) {
                        if (src_dev < 0)
                            e = hipMemcpy(bc.dev, src, esz * n, hipMemcpyHostToDevice);
                        else
#if defined(ZK_EMU)
                            e = hipMemcpy(bc.dev, src, esz * n, hipMemcpyDeviceToDevice);
#else
                            e = hipMemcpyPeer(bc.dev, dc.device, src, g.devs[src_dev]->device, esz * n);
#endif
                        if (e != hipSuccess) st = ZK_ERR_HIP;
                    }
                }
            }
        }
        if (st == ZK_OK) CURVE_SWITCH(c, st = bases_prepare_run<C>(bc, n));
        if (st != ZK_OK) {
            cleanup();
            return st;
        }
    }
    hipSetDevice(g.devs[0]->device);
    const uint64_t h = g.next_handle++;
    g.bases[h] = std::move(be);
    *handle_out = h;
    return ZK_OK;
}

API int zk_bases_upload(zk_curve_t c, const void* host, uint64_t n, uint64_t* handle_out) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!handle_out || (n && !host)) return ZK_ERR_INVALID_ARG;
    return bases_install(c, host, -1, n, handle_out);
}
API int zk_bases_adopt_device(zk_curve_t c, const void* dev, uint64_t n, uint64_t* handle_out) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!handle_out || !dev || !aligned16(dev)) return ZK_ERR_INVALID_ARG;
    return bases_install(c, dev, device_of(dev).index, n, handle_out);
}
API int zk_bases_free(uint64_t handle) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    auto it = g.bases.find(handle);
    if (it == g.bases.end()) return ZK_ERR_BAD_HANDLE;
    for (size_t d = 0; d < it->second.per_dev.size(); d++) {
        BasesCopy& bc = it->second.per_dev[d];
        hipSetDevice(g.devs[d]->device);
        if (bc.owned || bc.dev29) hipDeviceSynchronize();
        if (bc.owned) hipFree(bc.dev);
        if (bc.dev29) hipFree(bc.dev29);
        if (bc.pre) hipFree(bc.pre);
    }
    hipSetDevice(g.devs[0]->device);
    g.bases.erase(it);
    return ZK_OK;
}

API int zk_bases_precompute(uint64_t handle, int window_bits) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    auto it = g.bases.find(handle);
    if (it == g.bases.end()) return ZK_ERR_BAD_HANDLE;
    BasesEntry& be = it->second;
    const zk_curve_t c = (zk_curve_t)be.curve;
    const int cw = msm_pick_c(be.n, window_bits, true);
    for (size_t d = 0; d < be.per_dev.size(); d++) {
        ZK_TRY(bind_device(*g.devs[d]));
        int st = ZK_ERR_INVALID_ARG;
        CURVE_SWITCH(c, st = bases_precompute_run<C>(be.per_dev[d], be.n, cw));
        ZK_TRY(st);
    }
    return ZK_OK;
}
API int zk_bases_refresh(uint64_t handle, uint64_t offset, uint64_t count, void* stream) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    auto it = g.bases.find(handle);
    if (it == g.bases.end()) return ZK_ERR_BAD_HANDLE;
    BasesEntry& be = it->second;
    if (offset > be.n || count > be.n - offset) return ZK_ERR_INVALID_ARG;
    size_t esz = 0;
    const zk_curve_t c = (zk_curve_t)be.curve;
    CURVE_SWITCH(c, esz = sizeof(Affine<C>));
    int src = -1;
    for (size_t d = 0; d < be.per_dev.size(); d++)
        if (!be.per_dev[d].owned) src = (int)d;          // the adopted copy is the one the caller rewrites
    if (src < 0) src = 0;
    for (size_t d = 0; d < be.per_dev.size(); d++) {
        DeviceCtx& dc = *g.devs[d];
        ZK_TRY(bind_device(dc));
        hipStream_t st = (hipStream_t)stream;
        if ((int)d != src) {      // a peer: wait for the caller's stream, copy the range over, convert on the library's stream
            HIP_TRY(hipSetDevice(g.devs[src]->device));
            HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
            HIP_TRY(hipSetDevice(dc.device));
#if defined(ZK_EMU)
            HIP_TRY(hipMemcpy((unsigned char*)be.per_dev[d].dev + offset * esz, (unsigned char*)be.per_dev[src].dev + offset * esz, count * esz, hipMemcpyDeviceToDevice));
#else
            HIP_TRY(hipMemcpyPeer((unsigned char*)be.per_dev[d].dev + offset * esz, dc.device, (unsigned char*)be.per_dev[src].dev + offset * esz,
                                  g.devs[src]->device, count * esz));
#endif
            st = nullptr;
        }
        CURVE_SWITCH(c, ZK_TRY(bases_refresh_run<C>(be.per_dev[d], offset, count, st)));
        if ((int)d != src) HIP_TRY(hipStreamSynchronize(nullptr));
    }
    hipSetDevice(g.devs[0]->device);
    return ZK_OK;
}

API int zk_msm_submit(zk_curve_t c, uint64_t handle, const void* d_scalars, uint64_t n, int mont, const zk_msm_opts* opts, void* stream,
                      uint64_t* ticket_out) {
    if (!ticket_out || (n && (!d_scalars || !aligned16(d_scalars)))) return ZK_ERR_INVALID_ARG;
    const BasesEntry* be = nullptr;
    ZK_TRY(find_bases(handle, c, n, &be, opts));
    DeviceCtx& dc = device_of(d_scalars);
    if (g.devs.size() > 1 && whole_msm(tuning_from(opts)) && n > 0) {
        // one process, several GPUs: one job per device (its window share), enqueued here without waiting; the peers copy the
        // scalars over xGMI into their job's own staging buffer behind an event on the caller's stream
        const int G = (int)g.devs.size();
        int nwin = 0;
        const MsmTuning tu_in = tuning_from(opts);
        CURVE_SWITCH(c, nwin = msm_windows<C>(msm_pick_c(n, tu_in.window_bits, tu_in.precomputed)));
        hipEvent_t ready = nullptr;
        {
            ZK_TRY(bind_device(dc));
            std::lock_guard<std::mutex> lk(dc.mu);
            ZK_TRY(ensure_lib_streams(dc));
            ready = dc.fork_ev;
            HIP_TRY(hipEventRecord(ready, (hipStream_t)stream));
        }
        std::vector<MsmJob*> jobs;
        int status = ZK_OK;
        for (int d = 0; d < G && status == ZK_OK; d++) {
            DeviceCtx& dd = *g.devs[d];
            MsmTuning tu = tu_in;
            tu.w0 = nwin * d / G;
            tu.w1 = nwin * (d + 1) / G;
            MsmJob* job = nullptr;
            status = bind_device(dd);
            if (status != ZK_OK) break;
            std::lock_guard<std::mutex> lk(dd.mu);
            if (tu.w0 == tu.w1) {
                tu.w0 = tu.w1 = 0;
                status = submit_on(dd, c, *be, nullptr, SRC_LOCAL, 0, 0, mont ? 1 : 0, tu, nullptr, &job);
            } else if (&dd == &dc) {
                status = submit_on(dd, c, *be, d_scalars, SRC_LOCAL, 0, n, mont ? 1 : 0, tu, (hipStream_t)stream, &job);
            } else {
                status = ensure_lib_streams(dd);
                if (status == ZK_OK && hipStreamWaitEvent(dd.own, ready, 0) != hipSuccess) status = ZK_ERR_HIP;
                if (status == ZK_OK) status = submit_on(dd, c, *be, d_scalars, SRC_PEER, dc.device, n, mont ? 1 : 0, tu, dd.own, &job);
            }
            if (status == ZK_OK) jobs.push_back(job);
        }
        hipSetDevice(g.devs[0]->device);
        if (status != ZK_OK) {      // release what was taken
            for (MsmJob* j : jobs) {
                std::vector<unsigned char> sink(jac_bytes(c));
                collect_job(*j, sink.data());
            }
            return status;
        }
        std::lock_guard<std::mutex> lk(g.mu);
        const uint64_t t = g.next_ticket++;
        for (MsmJob* j : jobs) j->ticket = t;
        g.tickets[t] = jobs;
        *ticket_out = t;
        return ZK_OK;
    }
    ZK_TRY(bind_device(dc));
    MsmJob* job = nullptr;
    {
        std::lock_guard<std::mutex> lk(dc.mu);
        hipStream_t run_on = (hipStream_t)stream;
        if (opts && (opts->flags & ZK_MSM_FLAG_OWN_STREAM)) {
            // the job runs on a library stream forked behind what `stream` holds now: the latency-bound phases of one MSM (the
            // oversized-bucket path and the bucket reduction: dependent additions at one wave per SIMD on the G2 types) then
            // run beside the next MSM's accumulate kernel instead of in front of it
            ZK_TRY(ensure_lib_streams(dc));
            HIP_TRY(hipEventRecord(dc.fork_ev, (hipStream_t)stream));
            run_on = dc.submit_streams[dc.submit_rr++ % ZK_MAX_JOBS];     // as many streams as job slots: every MSM in flight has its own
            HIP_TRY(hipStreamWaitEvent(run_on, dc.fork_ev, 0));
        }
        ZK_TRY(submit_on(dc, c, *be, d_scalars, SRC_LOCAL, 0, n, mont ? 1 : 0, tuning_from(opts), run_on, &job));
    }
    std::lock_guard<std::mutex> lk(g.mu);
    job->ticket = g.next_ticket++;
    g.tickets[job->ticket] = {job};
    *ticket_out = job->ticket;
    return ZK_OK;
}
API int zk_msm_collect(uint64_t ticket, void* out) {
    if (!out) return ZK_ERR_INVALID_ARG;
    std::vector<MsmJob*> jobs;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        ZK_TRY(require_init());
        auto it = g.tickets.find(ticket);
        if (it == g.tickets.end()) return ZK_ERR_BAD_HANDLE;
        jobs = it->second;
        g.tickets.erase(it);
    }
    if (jobs.size() == 1) return collect_job(*jobs[0], out);
    // a fanned-out submission: one window share per device; the tails run on the devices' workers, the partials are added here
    const zk_curve_t c = (zk_curve_t)jobs[0]->curve;
    const size_t pbytes = jac_bytes(c);
    std::vector<std::vector<unsigned char>> parts(jobs.size(), std::vector<unsigned char>(pbytes));
    std::vector<std::future<int>> fut;
    for (size_t d = 1; d < jobs.size(); d++) fut.push_back(run_on_device(*jobs[d]->dc, [&, d] { return collect_job(*jobs[d], parts[d].data()); }));
    int status = collect_job(*jobs[0], parts[0].data());
    for (auto& f : fut) {
        const int s2 = f.get();
        if (status == ZK_OK) status = s2;
    }
    ZK_TRY(status);
    memcpy(out, parts[0].data(), pbytes);
    for (size_t d = 1; d < jobs.size(); d++) ZK_TRY(point_add_host(c, out, parts[d].data(), out));
    hipSetDevice(g.devs[0]->device);
    return ZK_OK;
}

API int zk_msm_device(zk_curve_t c, uint64_t handle, const void* d_scalars, uint64_t n, int mont, const zk_msm_opts* opts,
                      void* out, void* stream) {
    if (!out || (n && (!d_scalars || !aligned16(d_scalars)))) return ZK_ERR_INVALID_ARG;
    const BasesEntry* be = nullptr;
    ZK_TRY(find_bases(handle, c, n, &be, opts));
    const MsmTuning tu = tuning_from(opts);
    DeviceCtx& dc = device_of(d_scalars);
    if (g.devs.size() > 1 && whole_msm(tu) && n > 0) return msm_fanout(c, *be, d_scalars, &dc, (hipStream_t)stream, n, mont ? 1 : 0, tu, out);
    ZK_TRY(bind_device(dc));
    MsmJob* job = nullptr;
    {
        std::lock_guard<std::mutex> lk(dc.mu);
        ZK_TRY(submit_on(dc, c, *be, d_scalars, SRC_LOCAL, 0, n, mont ? 1 : 0, tu, (hipStream_t)stream, &job));
    }
    return collect_job(*job, out);
}

API int zk_msm(zk_curve_t c, uint64_t handle, const void* scalars_host, uint64_t n, int mont, const zk_msm_opts* opts,
               void* out) {
    if (!out || (n && !scalars_host)) return ZK_ERR_INVALID_ARG;
    const BasesEntry* be = nullptr;
    ZK_TRY(find_bases(handle, c, n, &be, opts));
    const MsmTuning tu = tuning_from(opts);
    if (g.devs.size() > 1 && whole_msm(tu) && n > 0) return msm_fanout(c, *be, scalars_host, nullptr, nullptr, n, mont ? 1 : 0, tu, out);
    DeviceCtx& dc = *g.devs[0];
    ZK_TRY(bind_device(dc));
    MsmJob* job = nullptr;
    {
        std::lock_guard<std::mutex> lk(dc.mu);
        ZK_TRY(submit_on(dc, c, *be, scalars_host, SRC_HOST, 0, n, mont ? 1 : 0, tu, (hipStream_t)0, &job));
    }
    return collect_job(*job, out);
}

// `count` MSMs over one bases entry on ONE device (the scalars are in that device's memory): jobs of up to four vectors,
// alternating over the device's two library streams, forked from and joined back into `stream`
static int batch_on_device(DeviceCtx& dc, zk_curve_t c, const BasesEntry* be, const void* d_scalars, uint64_t n, uint32_t count, uint64_t stride_elems,
                           int mont, const MsmTuning& tu, void* out, void* stream) {
    ZK_TRY(bind_device(dc));
    size_t pbytes = 0;
    CURVE_SWITCH(c, pbytes = (size_t)3 * 4 * coord_words<C>());
    {
        std::lock_guard<std::mutex> lk(dc.mu);
        ZK_TRY(ensure_lib_streams(dc));
        HIP_TRY(hipEventRecord(dc.fork_ev, (hipStream_t)stream));
        for (auto& s : dc.side) HIP_TRY(hipStreamWaitEvent(s, dc.fork_ev, 0));
    }
    // Up to 4 scalar vectors go into ONE job: their windows are simply more windows of the same sort / accumulate / reduce
    // launches (one persistent accumulate launch without a drain per MSM, the latency-bound reduction steps once per job).
    // The bound is the (window, range) region count the sort kernels index: 4096.  Job k runs on side stream k mod 2; at most
    // ZK_MAX_JOBS - 1 in flight, collected in order.
    uint32_t per_job = 1;
    {
        int cw = 0, nwin = 0;
        CURVE_SWITCH(c, cw = msm_pick_c(n, tu.window_bits, tu.precomputed); nwin = msm_windows<C>(cw));
        const int nwj = (tu.w0 == 0 && tu.w1 == 0) ? nwin : tu.w1 - tu.w0;
        const uint32_t nbk = 1u << (cw - 1), nranges = nbk > 512 ? nbk / 512 : 1;
        // (the one-bucket-set form: one window of at most 1024 ranges per vector, whatever c)
        const uint32_t regions = tu.precomputed ? (nranges > 1024 ? nranges : 1024u) : (uint32_t)(nwj > 0 ? nwj : 1) * nranges;
        per_job = 4096 / regions;
        if (per_job > 4) per_job = 4;
        if (per_job < 1) per_job = 1;
        if ((uint64_t)n * (uint64_t)(nwj > 0 ? nwj : 1) * per_job >= (1ull << 32)) per_job = 1;   // entry positions are u32
    }
    struct Flight {
        MsmJob* job;
        uint32_t first, size;
    };
    std::vector<Flight> inflight;
    uint32_t submitted = 0, collected = 0, jobs = 0;
    int status = ZK_OK;
    while (collected < count && status == ZK_OK) {
        while (submitted < count && inflight.size() < (size_t)ZK_MAX_JOBS - 1 && status == ZK_OK) {
            MsmJob* job = nullptr;
            MsmTuning tj = tu;
            tj.batch = count - submitted < per_job ? count - submitted : per_job;
            tj.batch_stride = stride_elems;
            std::lock_guard<std::mutex> lk(dc.mu);
            status = submit_on(dc, c, *be, (const unsigned char*)d_scalars + (size_t)submitted * stride_elems * 32, SRC_LOCAL, 0, n, mont ? 1 : 0,
                               tj, dc.side[jobs & 1], &job);
            if (status == ZK_ERR_BUSY && !inflight.empty()) {   // other callers hold the remaining slots: drain ours first
                status = ZK_OK;
                break;
            }
            if (status == ZK_OK) {
                inflight.push_back({job, submitted, tj.batch});
                submitted += tj.batch;
                jobs++;
            }
        }
        if (status != ZK_OK || inflight.empty()) break;
        status = collect_job(*inflight.front().job, (unsigned char*)out + (size_t)inflight.front().first * pbytes);
        collected += inflight.front().size;
        inflight.erase(inflight.begin());
    }
    for (auto& fl : inflight) {   // error path: release what is still in flight
        std::vector<unsigned char> sink(pbytes * fl.size);
        collect_job(*fl.job, sink.data());
    }
    if (status == ZK_OK && collected < count) status = ZK_ERR_BUSY;
    {
        std::lock_guard<std::mutex> lk(dc.mu);
        for (int s = 0; s < 2; s++) {
            HIP_TRY(hipEventRecord(dc.join_ev[s], dc.side[s]));
            HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, dc.join_ev[s], 0));
        }
    }
    return status;
}

API int zk_msm_batch_device(zk_curve_t c, uint64_t handle, const void* d_scalars, uint64_t n, uint32_t count, uint64_t stride_elems,
                            int mont, const zk_msm_opts* opts, void* out, void* stream) {
    if (count == 0) return ZK_OK;
    if (!out || stride_elems < n || (n && (!d_scalars || !aligned16(d_scalars)))) return ZK_ERR_INVALID_ARG;
    const BasesEntry* be = nullptr;
    ZK_TRY(find_bases(handle, c, n, &be, opts));
    const MsmTuning tu = tuning_from(opts);
    DeviceCtx& owner = device_of(d_scalars);
    const int G = (int)g.devs.size();
    if (G == 1 || !whole_msm(tu) || n == 0) return batch_on_device(owner, c, be, d_scalars, n, count, stride_elems, mont, tu, out, stream);
    const size_t pbytes = jac_bytes(c);
    if (count < (uint32_t)G) {
        // fewer vectors than devices (the L / R pair of an IPA round): every vector window-sharded over ALL devices, one after another
        for (uint32_t v = 0; v < count; v++)
            ZK_TRY(msm_fanout(c, *be, (const unsigned char*)d_scalars + (size_t)v * stride_elems * 32, &owner, (hipStream_t)stream, n, mont ? 1 : 0, tu,
                              (unsigned char*)out + (size_t)v * pbytes));
        return ZK_OK;
    }
    // at least one vector per device (halo2's column commitments): device d takes the WHOLE MSMs of vectors [count d / G, count (d + 1) / G)
    // -- whole MSMs keep their fixed costs once per vector instead of once per vector and device -- and only those vectors
    // travel (over xGMI, into the device's staging buffer, behind an event on the caller's stream)
    hipEvent_t ready = nullptr;
    {
        ZK_TRY(bind_device(owner));
        std::lock_guard<std::mutex> lk(owner.mu);
        ZK_TRY(ensure_lib_streams(owner));
        ready = owner.fork_ev;
        HIP_TRY(hipEventRecord(ready, (hipStream_t)stream));
    }
    auto work = [&](int d) -> int {
        DeviceCtx& dc = *g.devs[d];
        const uint32_t first = (uint32_t)((uint64_t)count * d / G), last = (uint32_t)((uint64_t)count * (d + 1) / G);
        if (last == first) return ZK_OK;
        unsigned char* dst_out = (unsigned char*)out + (size_t)first * pbytes;
        if (&dc == &owner)
            return batch_on_device(dc, c, be, (const unsigned char*)d_scalars + (size_t)first * stride_elems * 32, n, last - first, stride_elems, mont, tu,
                                   dst_out, stream);
        ZK_TRY(bind_device(dc));
        std::lock_guard<std::mutex> stage_lk(dc.batch_mu);      // the staging buffer serves one fanned-out batch at a time
        {
            std::lock_guard<std::mutex> lk(dc.mu);
            ZK_TRY(ensure_lib_streams(dc));
            ZK_TRY(ws_get(dc.batch_in, (size_t)(last - first) * n * 32));
            HIP_TRY(hipStreamWaitEvent(dc.own, ready, 0));
            for (uint32_t v = first; v < last; v++) {
                const void* srcp = (const unsigned char*)d_scalars + (size_t)v * stride_elems * 32;
                void* dstp = (unsigned char*)dc.batch_in.p + (size_t)(v - first) * n * 32;
#if defined(ZK_EMU)
                HIP_TRY(hipMemcpyAsync(dstp, srcp, n * 32, hipMemcpyDeviceToDevice, dc.own));
#else
                HIP_TRY(hipMemcpyPeerAsync(dstp, dc.device, srcp, owner.device, n * 32, dc.own));
#endif
            }
        }
        return batch_on_device(dc, c, be, dc.batch_in.p, n, last - first, n, mont, tu, dst_out, dc.own);
    };
    std::vector<std::future<int>> fut;
    for (int d = 0; d < G; d++)
        if (g.devs[d].get() != &owner) fut.push_back(run_on_device(*g.devs[d], [&, d] { return work(d); }));
    int status = work(owner.index);
    for (auto& f : fut) {
        const int s2 = f.get();
        if (status == ZK_OK) status = s2;
    }
    hipSetDevice(g.devs[0]->device);
    return status;
}

API int zk_msm_last_profile(zk_msm_profile* out) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (!out) return ZK_ERR_INVALID_ARG;
    *out = g.prof;
    return ZK_OK;
}

API int zk_msm_profile_totals(zk_msm_totals* out, int reset) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (!out) return ZK_ERR_INVALID_ARG;
    *out = g.totals;
    if (reset) memset(&g.totals, 0, sizeof g.totals);
    return ZK_OK;
}

API int zk_ntt_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g.mu);
    g.ntt_profile = on != 0;
    return ZK_OK;
}
API int zk_ntt_profile_read(zk_ntt_totals* out) {
    if (!out) return ZK_ERR_INVALID_ARG;
    {
        std::lock_guard<std::mutex> lk0(g.mu);
        ZK_TRY(require_init());
    }
    memset(out, 0, sizeof *out);
    for (auto& dcp : g.devs) {
        DeviceCtx& dc = *dcp;
        ZK_TRY(bind_device(dc));
        std::lock_guard<std::mutex> lk(dc.mu);
        for (size_t i = 0; i + 1 < dc.ntt_ev_used; i += 2) {
            HIP_TRY(hipEventSynchronize(dc.ntt_ev_pool[i + 1]));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, dc.ntt_ev_pool[i], dc.ntt_ev_pool[i + 1]));
            out->kernel_ms += ms;
            out->launches++;
        }
        out->transforms += dc.ntt_transforms;
        out->algorithmic_bytes += dc.ntt_alg_bytes;
        dc.ntt_ev_used = 0;
        dc.ntt_transforms = 0;
        dc.ntt_alg_bytes = 0;
    }
    hipSetDevice(g.devs[0]->device);
    return ZK_OK;
}

API int zk_ntt_configure(const zk_ntt_opts* opts) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (opts)
        g.ntt_opts = *opts;
    else
        memset(&g.ntt_opts, 0, sizeof g.ntt_opts);
    return ZK_OK;
}

// entry points on device buffers: bind the calling thread to the owner of `a`, serialise the enqueue on that device
#define DEVICE_ENTRY(ptr)                       \
    {                                           \
        std::lock_guard<std::mutex> lk0(g.mu);  \
        ZK_TRY(require_init());                 \
    }                                           \
    DeviceCtx& dc = device_of(ptr);             \
    ZK_TRY(bind_device(dc));                    \
    std::lock_guard<std::mutex> lk(dc.mu)

API int zk_ntt_device(zk_field_t f, void* a, uint32_t log_n, const void* omega, int scale, void* stream) {
    if (!a || !omega || !aligned16(a)) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(a);
    FIELD_SWITCH(f, {
        Fe<F> w;
        host_load(w, omega);
        return ntt_run<F>(dc, (int)f, (Fe<F>*)a, log_n, w, scale, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_ntt(zk_field_t f, void* a_host, uint32_t log_n, const void* omega, int scale) {
    if (!a_host || !omega || log_n > 30) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(nullptr);
    const size_t bytes = (size_t)32 << log_n;
    ZK_TRY(ws_get(dc.scratch_in, bytes));
    HIP_TRY(hipMemcpy(dc.scratch_in.p, a_host, bytes, hipMemcpyHostToDevice));
    FIELD_SWITCH(f, {
        Fe<F> w;
        host_load(w, omega);
        ZK_TRY(ntt_run<F>(dc, (int)f, (Fe<F>*)dc.scratch_in.p, log_n, w, scale, (hipStream_t)0));
    });
    HIP_TRY(hipStreamSynchronize((hipStream_t)0));
    HIP_TRY(hipMemcpy(a_host, dc.scratch_in.p, bytes, hipMemcpyDeviceToHost));
    return ZK_OK;
}
API int zk_ntt_coset_device(zk_field_t f, void* a, uint32_t log_n, const void* omega, int scale, const void* g_pre,
                            const void* g_post, void* stream) {
    if (!a || !omega || !aligned16(a)) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(a);
    FIELD_SWITCH(f, {
        Fe<F> w, gp, gq;
        host_load(w, omega);
        if (g_pre) host_load(gp, g_pre);
        if (g_post) host_load(gq, g_post);
        return ntt_run<F>(dc, (int)f, (Fe<F>*)a, log_n, w, scale, (hipStream_t)stream, g_pre ? &gp : nullptr, g_post ? &gq : nullptr);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_ntt_extend_device(zk_field_t f, void* a, uint32_t log_n, uint32_t log_in, const void* omega, int scale, const void* g_pre,
                             const void* g_post, void* stream) {
    if (!a || !omega || !aligned16(a) || log_in > log_n || log_n > 30) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(a);
    FIELD_SWITCH(f, {
        if (log_n > (uint32_t)F::TWO_ADICITY) return ZK_ERR_INVALID_ARG;
        Fe<F> w, gp, gq;
        host_load(w, omega);
        if (g_pre) host_load(gp, g_pre);
        if (g_post) host_load(gq, g_post);
        if (log_n > 0 && log_in == 0) {   // a single coefficient: in_log = 0 means "all" to the pass kernel, so pad explicitly
            HIP_TRY(hipMemsetAsync((Fe<F>*)a + 1, 0, (((size_t)1 << log_n) - 1) * sizeof(Fe<F>), (hipStream_t)stream));
        }
        return ntt_run<F>(dc, (int)f, (Fe<F>*)a, log_n, w, scale, (hipStream_t)stream, g_pre ? &gp : nullptr, g_post ? &gq : nullptr, log_in);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_ntt_oop_device(zk_field_t f, const void* src, void* dst, uint32_t log_n, uint32_t log_in, const void* omega, int scale, const void* g_pre,
                          const void* g_post, void* stream) {
    if (!src || !dst || !omega || !aligned16(src) || !aligned16(dst) || log_in > log_n || log_n > 30) return ZK_ERR_INVALID_ARG;
    if (src == dst) return zk_ntt_extend_device(f, dst, log_n, log_in, omega, scale, g_pre, g_post, stream);
    {   // the buffers must not overlap: the first pass reads src while later tiles of the same pass may already write dst
        const uintptr_t s0 = (uintptr_t)src, s1 = s0 + ((uintptr_t)32 << log_in), d0 = (uintptr_t)dst, d1 = d0 + ((uintptr_t)32 << log_n);
        if (s0 < d1 && d0 < s1) return ZK_ERR_INVALID_ARG;
    }
    DEVICE_ENTRY(dst);
    FIELD_SWITCH(f, {
        if (log_n > (uint32_t)F::TWO_ADICITY) return ZK_ERR_INVALID_ARG;
        Fe<F> w, gp, gq;
        host_load(w, omega);
        if (g_pre) host_load(gp, g_pre);
        if (g_post) host_load(gq, g_post);
        if (log_n > 0 && log_in == 0) {   // a single coefficient (in_log = 0 means "all" to the pass kernel): pad explicitly, then in place
            HIP_TRY(hipMemcpyAsync(dst, src, sizeof(Fe<F>), hipMemcpyDeviceToDevice, (hipStream_t)stream));
            HIP_TRY(hipMemsetAsync((Fe<F>*)dst + 1, 0, (((size_t)1 << log_n) - 1) * sizeof(Fe<F>), (hipStream_t)stream));
            return ntt_run<F>(dc, (int)f, (Fe<F>*)dst, log_n, w, scale, (hipStream_t)stream, g_pre ? &gp : nullptr, g_post ? &gq : nullptr, log_in);
        }
        return ntt_run<F>(dc, (int)f, (Fe<F>*)dst, log_n, w, scale, (hipStream_t)stream, g_pre ? &gp : nullptr, g_post ? &gq : nullptr, log_in,
                          (const Fe<F>*)src);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_coset_mul_device(zk_field_t f, void* a, uint32_t log_n, const void* gm, void* stream) {
    if (!a || !gm || !aligned16(a)) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(a);
    FIELD_SWITCH(f, {
        Fe<F> w;
        host_load(w, gm);
        return coset_run<F>(dc, (int)f, (Fe<F>*)a, log_n, w, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_coset_mul(zk_field_t f, void* a_host, uint32_t log_n, const void* gm) {
    if (!a_host || !gm || log_n > 30) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(nullptr);
    const size_t bytes = (size_t)32 << log_n;
    ZK_TRY(ws_get(dc.scratch_in, bytes));
    HIP_TRY(hipMemcpy(dc.scratch_in.p, a_host, bytes, hipMemcpyHostToDevice));
    FIELD_SWITCH(f, {
        Fe<F> w;
        host_load(w, gm);
        ZK_TRY(coset_run<F>(dc, (int)f, (Fe<F>*)dc.scratch_in.p, log_n, w, (hipStream_t)0));
    });
    HIP_TRY(hipMemcpy(a_host, dc.scratch_in.p, bytes, hipMemcpyDeviceToHost));
    return ZK_OK;
}

API int zk_vec_op_device(zk_field_t f, int op, void* a, const void* b, const void* c, uint64_t n, const void* scalar, void* stream) {
    if (op < 0 || op > 6 || (n && (!a || !aligned16(a)))) return ZK_ERR_INVALID_ARG;
    const bool needs_b = op == 0 || op == 1 || op == 2 || op == 6, needs_c = op == 6, needs_s = op == 3 || op == 6;
    if (n && ((needs_b && (!b || !aligned16(b))) || (needs_c && (!c || !aligned16(c))) || (needs_s && !scalar))) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(a);
    FIELD_SWITCH(f, {
        Fe<F> s;
        fe_one(s);
        if (needs_s) host_load(s, scalar);
        return vec_op_run<F>((Fe<F>*)a, (const Fe<F>*)b, (const Fe<F>*)c, n, op, s, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_vec_scale_periodic_device(zk_field_t f, void* a, uint64_t n, const void* table_host, uint32_t m, void* stream) {
    if ((n && (!a || !aligned16(a))) || !table_host) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(a);
    FIELD_SWITCH(f, return scale_periodic_run<F>((Fe<F>*)a, n, (const Fe<F>*)table_host, m, (hipStream_t)stream));
    return ZK_ERR_INVALID_ARG;
}
API int zk_groth16_witness_map_device(zk_field_t f, void* a, void* b, void* c, uint32_t log_m, void* stream) {
    if (!a || !b || !c || !aligned16(a) || !aligned16(b) || !aligned16(c)) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(a);
    FIELD_SWITCH(f, return witness_map_run<F>(dc, (int)f, (Fe<F>*)a, (Fe<F>*)b, (Fe<F>*)c, log_m, (hipStream_t)stream));
    return ZK_ERR_INVALID_ARG;
}

API int zk_r1cs_matrix_upload(zk_field_t f, const uint64_t* row_ptr, const uint32_t* col, const void* val, uint64_t n_rows, uint64_t n_cols,
                              uint64_t* handle_out) {
    if (!handle_out || !row_ptr || (int)f < 0 || (int)f > (int)ZK_FR_BLS12_381) return ZK_ERR_INVALID_ARG;
    const uint64_t nnz = row_ptr[n_rows];
    if (row_ptr[0] != 0 || (nnz && (!col || !val)) || n_cols >= (1ull << 32)) return ZK_ERR_INVALID_ARG;
    std::vector<uint64_t> long_rows;
    for (uint64_t i = 0; i < n_rows; i++) {
        if (row_ptr[i + 1] < row_ptr[i]) return ZK_ERR_INVALID_ARG;
        if (row_ptr[i + 1] - row_ptr[i] > R1CS_LONG_ROW) long_rows.push_back(i);
    }
    for (uint64_t k = 0; k < nnz; k++)
        if (col[k] >= n_cols) return ZK_ERR_INVALID_ARG;   // a mat-vec must never index past the assignment
    DEVICE_ENTRY(nullptr);
    R1csMatrix m;
    m.field = (int)f;
    m.n_rows = n_rows;
    m.n_cols = n_cols;
    m.nnz = nnz;
    m.n_long = long_rows.size();
    auto put = [&](void** dst, const void* src, size_t bytes) -> int {
        HIP_TRY(hipMalloc(dst, bytes ? bytes : 16));
        if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return ZK_OK;
    };
    int st = put(&m.row_ptr, row_ptr, (n_rows + 1) * 8);
    if (st == ZK_OK) st = put(&m.col, col, nnz * 4);
    if (st == ZK_OK) st = put(&m.val, val, nnz * 32);
    if (st == ZK_OK) st = put(&m.long_rows, long_rows.data(), long_rows.size() * 8);
    if (st != ZK_OK) {
        free_matrix(m);
        return st;
    }
    std::lock_guard<std::mutex> lkm(g_mat_mu);
    const uint64_t h = g_next_mat++;
    g_mats[h] = m;
    *handle_out = h;
    return ZK_OK;
}
API int zk_r1cs_matrix_free(uint64_t handle) {
    DEVICE_ENTRY(nullptr);
    std::lock_guard<std::mutex> lkm(g_mat_mu);
    auto it = g_mats.find(handle);
    if (it == g_mats.end()) return ZK_ERR_BAD_HANDLE;
    hipDeviceSynchronize();
    free_matrix(it->second);
    g_mats.erase(it);
    return ZK_OK;
}
static int find_matrix(uint64_t handle, R1csMatrix* out) {
    std::lock_guard<std::mutex> lkm(g_mat_mu);
    auto it = g_mats.find(handle);
    if (it == g_mats.end()) return ZK_ERR_BAD_HANDLE;
    *out = it->second;
    return ZK_OK;
}
API int zk_r1cs_matvec_device(uint64_t matrix, const void* z, void* out, uint64_t out_len, void* stream) {
    if (!z || !out || !aligned16(z) || !aligned16(out)) return ZK_ERR_INVALID_ARG;
    R1csMatrix m;
    ZK_TRY(find_matrix(matrix, &m));
    DEVICE_ENTRY(nullptr);
    FIELD_SWITCH((zk_field_t)m.field, return r1cs_matvec_run<F>(m, (const Fe<F>*)z, (Fe<F>*)out, out_len, (hipStream_t)stream));
    return ZK_ERR_INVALID_ARG;
}
API int zk_groth16_witness_map_r1cs_device(zk_field_t f, uint64_t ha, uint64_t hb, uint64_t hc, const void* z, uint64_t num_inputs,
                                           uint32_t log_m, void* a, void* b, void* c, void* stream) {
    if (!z || !a || !b || !c || !aligned16(z) || !aligned16(a) || !aligned16(b) || !aligned16(c) || log_m > 30) return ZK_ERR_INVALID_ARG;
    R1csMatrix ma, mb, mc;
    ZK_TRY(find_matrix(ha, &ma));
    ZK_TRY(find_matrix(hb, &mb));
    ZK_TRY(find_matrix(hc, &mc));
    const uint64_t m = 1ull << log_m, nc = ma.n_rows;
    if (ma.field != (int)f || mb.field != (int)f || mc.field != (int)f || mb.n_rows != nc || mc.n_rows != nc || nc + num_inputs > m ||
        num_inputs > ma.n_cols)
        return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(nullptr);
    FIELD_SWITCH(f, {
        hipStream_t st = (hipStream_t)stream;
        ZK_TRY(r1cs_matvec_run<F>(ma, (const Fe<F>*)z, (Fe<F>*)a, m, st));
        ZK_TRY(r1cs_matvec_run<F>(mb, (const Fe<F>*)z, (Fe<F>*)b, m, st));
        ZK_TRY(r1cs_matvec_run<F>(mc, (const Fe<F>*)z, (Fe<F>*)c, m, st));
        // the input-consistency rows: a[num_constraints + j] = z[j]
        if (num_inputs) HIP_TRY(hipMemcpyAsync((Fe<F>*)a + nc, z, num_inputs * sizeof(Fe<F>), hipMemcpyDeviceToDevice, st));
        return witness_map_run<F>(dc, (int)f, (Fe<F>*)a, (Fe<F>*)b, (Fe<F>*)c, log_m, st);
    });
    return ZK_ERR_INVALID_ARG;
}

// ---- halo2 prover steps beyond commit / FFT (include/zkcp_amd_prover.h)
API int zk_batch_invert_device(zk_field_t f, void* a, uint64_t n, void* stream) {
    if (n && (!a || !aligned16(a))) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(a);
    FIELD_SWITCH(f, return batch_invert_run<F>((Fe<F>*)a, n, (hipStream_t)stream));
    return ZK_ERR_INVALID_ARG;
}
API int zk_prefix_product_device(zk_field_t f, const void* in, void* out, uint64_t n, const void* first, void* total_host, void* stream) {
    if (n && (!in || !out || !aligned16(in) || !aligned16(out))) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(out);
    FIELD_SWITCH(f, {
        Fe<F> fst;
        fe_one(fst);
        if (first) host_load(fst, first);
        Fe<F>* total = nullptr;
        ZK_TRY(prefix_product_run<F>(dc, (const Fe<F>*)in, (Fe<F>*)out, n, fst, &total, (hipStream_t)stream));
        if (total_host) {
            HIP_TRY(hipMemcpyAsync(total_host, total, sizeof(Fe<F>), hipMemcpyDeviceToHost, (hipStream_t)stream));
            HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        }
        return ZK_OK;
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_halo2_permutation_product_device(zk_field_t f, uint32_t ncols, const void* const* cols, const void* const* sigmas, uint32_t first_col,
                                            const void* beta, const void* gamma, const void* delta, uint32_t k, const void* z_first, void* z_out,
                                            void* z_last_host, void* stream) {
    if (!cols || !sigmas || !beta || !gamma || !delta || !z_out || !aligned16(z_out) || ncols == 0 || ncols > 8) return ZK_ERR_INVALID_ARG;
    for (uint32_t c = 0; c < ncols; c++)
        if (!cols[c] || !sigmas[c] || !aligned16(cols[c]) || !aligned16(sigmas[c])) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(z_out);
    FIELD_SWITCH(f, {
        if (k > (uint32_t)F::TWO_ADICITY || k > 30) return ZK_ERR_INVALID_ARG;
        Fe<F> b, g_, d, fst, w;
        host_load(b, beta);
        host_load(g_, gamma);
        host_load(d, delta);
        fe_one(fst);
        if (z_first) host_load(fst, z_first);
        for (int i = 0; i < F::N; i++) w.v[i] = F::ROOT[i];
        for (uint32_t i = k; i < (uint32_t)F::TWO_ADICITY; i++) fe_sqr(w, w);
        return perm_product_run<F>(dc, (int)f, ncols, cols, sigmas, first_col, b, g_, d, k, w, fst, (Fe<F>*)z_out, z_last_host, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_halo2_lookup_product_device(zk_field_t f, const void* a, const void* s, const void* ap, const void* sp, const void* beta,
                                       const void* gamma, uint64_t n, void* z_out, void* z_last_host, void* stream) {
    if (!beta || !gamma || (n && (!a || !s || !ap || !sp || !z_out || !aligned16(a) || !aligned16(s) || !aligned16(ap) || !aligned16(sp) || !aligned16(z_out))))
        return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(z_out);
    FIELD_SWITCH(f, {
        Fe<F> b, g_, one;
        host_load(b, beta);
        host_load(g_, gamma);
        fe_one(one);
        return lookup_product_run<F>(dc, (const Fe<F>*)a, (const Fe<F>*)s, (const Fe<F>*)ap, (const Fe<F>*)sp, b, g_, n, one, (Fe<F>*)z_out,
                                     z_last_host, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_inner_product_device(zk_field_t f, const void* a, const void* b, uint64_t n, void* out_host, void* stream) {
    if (!out_host || (n && (!a || !b || !aligned16(a) || !aligned16(b)))) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(a);
    FIELD_SWITCH(f, return inner_product_run<F>(dc, (const Fe<F>*)a, (const Fe<F>*)b, n, out_host, (hipStream_t)stream));
    return ZK_ERR_INVALID_ARG;
}
API int zk_vec_muladd_device(zk_field_t f, void* a, const void* b, uint64_t n, const void* s, void* stream) {
    if (!s || (n && (!a || !b || !aligned16(a) || !aligned16(b)))) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(a);
    FIELD_SWITCH(f, {
        Fe<F> ss;
        host_load(ss, s);
        return vec_muladd_run<F>((Fe<F>*)a, (const Fe<F>*)a, (const Fe<F>*)b, n, ss, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_vec_muladd_to_device(zk_field_t f, void* out, const void* a, const void* b, uint64_t n, const void* s, void* stream) {
    if (!s || (n && (!out || !a || !b || !aligned16(out) || !aligned16(a) || !aligned16(b)))) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(out);
    FIELD_SWITCH(f, {
        Fe<F> ss;
        host_load(ss, s);
        return vec_muladd_run<F>((Fe<F>*)out, (const Fe<F>*)a, (const Fe<F>*)b, n, ss, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_poly_eval_device(zk_field_t f, const void* c, uint64_t n, const void* x, void* out_host, void* stream) {
    return zk_poly_eval_batch_device(f, c, n, 1, n, x, out_host, stream);
}
API int zk_poly_eval_batch_device(zk_field_t f, const void* c, uint64_t n, uint32_t count, uint64_t stride_elems, const void* x, void* out_host,
                                  void* stream) {
    if (!out_host || !x || stride_elems < n || (n && count && (!c || !aligned16(c)))) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(c);
    FIELD_SWITCH(f, {
        Fe<F> xx;
        host_load(xx, x);
        return poly_eval_run<F>(dc, (const Fe<F>*)c, n, count, stride_elems, xx, (int)f, out_host, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_vec_fold_many_device(zk_field_t f, void* out, const void* first, int64_t stride_elems, uint32_t count, uint64_t n, const void* s,
                                void* stream) {
    if (!s || count == 0 || (n && (!out || !first || !aligned16(out) || !aligned16(first)))) return ZK_ERR_INVALID_ARG;
    if (count > 1 && (stride_elems == 0 || (uint64_t)(stride_elems < 0 ? -stride_elems : stride_elems) < n)) return ZK_ERR_INVALID_ARG;
    {   // out may be one of the sources (every lane reads all of its inputs before it writes) or disjoint from all of them
        const intptr_t o = (intptr_t)out, len = (intptr_t)n * 32;
        for (uint32_t i = 0; i < count; i++) {
            const intptr_t a = (intptr_t)first + (intptr_t)i * stride_elems * 32;
            if (a != o && a < o + len && o < a + len) return ZK_ERR_INVALID_ARG;
        }
    }
    DEVICE_ENTRY(out);
    FIELD_SWITCH(f, {
        Fe<F> ss;
        host_load(ss, s);
        return vec_fold_many_run<F>((Fe<F>*)out, (const Fe<F>*)first, stride_elems, count, n, ss, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_ipa_fold_round_device(zk_field_t f, void* p, void* b, void* w, uint64_t half, uint64_t m0, const void* u, void* stream) {
    if (!u || (half && (!p || !b || !aligned16(p) || !aligned16(b))) || (w && !aligned16(w))) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(p);
    FIELD_SWITCH(f, {
        Fe<F> uu;
        host_load(uu, u);
        if (fe_is_zero(uu)) return ZK_ERR_INVALID_ARG;
        return ipa_fold_round_run<F>((Fe<F>*)p, (Fe<F>*)b, (Fe<F>*)w, half, m0, uu, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_vec_powers_device(zk_field_t f, void* out, uint64_t n, const void* x, void* stream) {
    if (!x || (n && (!out || !aligned16(out)))) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(out);
    FIELD_SWITCH(f, {
        Fe<F> xx;
        host_load(xx, x);
        return vec_powers_run<F>(dc, (Fe<F>*)out, n, xx, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_kate_division_device(zk_field_t f, const void* a, void* q, uint64_t n, const void* x, void* stream) {
    if (!x || (n && (!a || !q || !aligned16(a) || !aligned16(q)))) return ZK_ERR_INVALID_ARG;
    if (a != q) {
        const uintptr_t a0 = (uintptr_t)a, q0 = (uintptr_t)q, len = (uintptr_t)n * 32;
        if (a0 < q0 + len && q0 < a0 + len) return ZK_ERR_INVALID_ARG;   // in place or disjoint
    }
    DEVICE_ENTRY(q);
    FIELD_SWITCH(f, {
        Fe<F> xx;
        host_load(xx, x);
        return kate_division_run<F>(dc, (const Fe<F>*)a, (Fe<F>*)q, n, xx, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_vec_fold_device(zk_field_t f, void* a, uint64_t half, const void* c, void* stream) {
    if (!c || (half && (!a || !aligned16(a)))) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(a);
    FIELD_SWITCH(f, {
        Fe<F> cc;
        host_load(cc, c);
        return vec_fold_run<F>((Fe<F>*)a, half, cc, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_ipa_fold_bases_device(zk_curve_t c, void* g_aff, uint64_t half, const void* u_mont, void* stream) {
    if (!u_mont || (half && (!g_aff || !aligned16(g_aff)))) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(g_aff);
    CURVE_SWITCH(c, {
        Fe<typename C::Fr> u;
        host_load(u, u_mont);
        fe_from_mont(u, u);
        return ipa_fold_bases_run<C>(dc, (Affine<C>*)g_aff, half, u, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_ipa_collapse_device(zk_curve_t c, uint64_t handle, const void* w, uint64_t m0, uint64_t cur, void* g_out, void* stream) {
    return zk_ipa_collapse_range_device(c, handle, w, m0, cur, 0, cur, g_out, stream);
}
API int zk_ipa_collapse_range_device(zk_curve_t c, uint64_t handle, const void* w, uint64_t m0, uint64_t cur, uint64_t first, uint64_t count,
                                     void* g_out, void* stream) {
    if (!w || !g_out || !aligned16(w) || !aligned16(g_out)) return ZK_ERR_INVALID_ARG;
    const BasesCopy* bc = nullptr;
    uint64_t base_n = 0;
    DeviceCtx* dcp = nullptr;
    {
        std::lock_guard<std::mutex> lk0(g.mu);
        ZK_TRY(require_init());
        auto it = g.bases.find(handle);
        if (it == g.bases.end()) return ZK_ERR_BAD_HANDLE;
        if (it->second.curve != (int)c) return ZK_ERR_INVALID_ARG;
        dcp = &device_of(g_out);
        bc = &it->second.per_dev[dcp->index];
        base_n = it->second.n;
    }
    DeviceCtx& dc = *dcp;
    ZK_TRY(bind_device(dc));
    std::lock_guard<std::mutex> lk(dc.mu);
    CURVE_SWITCH(c, return ipa_collapse_run<C>(dc, *bc, base_n, (const Fe<typename C::Fr>*)w, m0, cur, first, count, (Affine<C>*)g_out,
                                               (hipStream_t)stream));
    return ZK_ERR_INVALID_ARG;
}
API int zk_ipa_virtual_scalars_device(zk_field_t f, const void* p, const void* w, uint64_t m0, uint64_t cur, void* sl, void* sr, void* stream) {
    if (!p || !w || !sl || !sr || !aligned16(p) || !aligned16(w) || !aligned16(sl) || !aligned16(sr)) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(sl);
    FIELD_SWITCH(f, return ipa_virtual_scalars_run<F>((const Fe<F>*)p, (const Fe<F>*)w, (Fe<F>*)sl, (Fe<F>*)sr, m0, cur, (hipStream_t)stream));
    return ZK_ERR_INVALID_ARG;
}
API int zk_ipa_round_device(zk_curve_t c, uint64_t handle, const void* p, const void* b, const void* w, uint64_t m0, uint64_t cur, void* s_dev,
                            void* lr_out, void* v_out, void* stream) {
    if (!p || !b || !w || !s_dev || !lr_out || !v_out || !aligned16(p) || !aligned16(b) || !aligned16(w) || !aligned16(s_dev))
        return ZK_ERR_INVALID_ARG;
    const int fi = zk_curve_scalar_field(c);
    if (fi < 0) return ZK_ERR_INVALID_ARG;
    const zk_field_t f = (zk_field_t)fi;
    size_t fbytes = 0;
    FIELD_SWITCH(f, fbytes = sizeof(Fe<F>));
    unsigned char* sl = (unsigned char*)s_dev;
    unsigned char* sr = sl + (size_t)m0 * fbytes;
    {
        DEVICE_ENTRY(s_dev);
        int st = ZK_ERR_INVALID_ARG;
        FIELD_SWITCH(f, st = ipa_round_begin_run<F>(dc, (const Fe<F>*)p, (const Fe<F>*)b, (const Fe<F>*)w, (Fe<F>*)sl, (Fe<F>*)sr, m0, cur,
                                                     (hipStream_t)stream));
        ZK_TRY(st);
    }
    // both MSMs over the resident generators (two library streams, forked behind the kernels above); dc.mu is not held here
    ZK_TRY(zk_msm_batch_device(c, handle, s_dev, m0, 2, m0, 1, nullptr, lr_out, stream));
    {
        DEVICE_ENTRY(s_dev);
        int st = ZK_ERR_INVALID_ARG;
        FIELD_SWITCH(f, st = ipa_round_end_run<F>(dc, (hipStream_t)stream, v_out, (unsigned char*)v_out + fbytes));
        return st;
    }
}
API int zk_ipa_update_weights_device(zk_field_t f, void* w, uint64_t m0, uint64_t bit, const void* u, void* stream) {
    if (!w || !u || !aligned16(w)) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(w);
    FIELD_SWITCH(f, {
        Fe<F> uu;
        host_load(uu, u);
        return ipa_update_weights_run<F>((Fe<F>*)w, m0, bit, uu, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_expr_eval_device(zk_field_t f, const zk_expr_op* prog, uint32_t n_ops, const void* const* cols, uint32_t n_cols, const void* consts,
                            uint32_t n_consts, uint32_t log_n, uint32_t rot_scale, void* out, void* stream) {
    if (!prog || !out || !aligned16(out) || (n_cols && !cols) || (n_consts && !consts)) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(out);
    FIELD_SWITCH(f, return expr_eval_run<F>(dc, prog, n_ops, cols, n_cols, (const Fe<F>*)consts, n_consts, log_n, rot_scale, (Fe<F>*)out,
                                            (hipStream_t)stream));
    return ZK_ERR_INVALID_ARG;
}

API int zk_expr_eval_lazy_device(zk_field_t f, const zk_expr_op* prog, uint32_t n_ops, const void* const* cols, uint32_t n_cols, const void* consts,
                                 uint32_t n_consts, uint32_t log_n, uint32_t rot_scale, void* out, void* stream) {
    if (!prog || !out || !aligned16(out) || (n_cols && !cols) || (n_consts && !consts)) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(out);
    FIELD_SWITCH(f, return expr_eval_lazy_run<F>(dc, prog, n_ops, cols, n_cols, (const Fe<F>*)consts, n_consts, log_n, rot_scale, (Fe<F>*)out,
                                                 (hipStream_t)stream));
    return ZK_ERR_INVALID_ARG;
}

API int zk_expr_specialised_source(zk_field_t f, const zk_expr_op* prog, uint32_t n_ops, uint32_t n_cols, uint32_t n_consts, char* out, uint64_t cap,
                                   uint64_t* len_out) {
    if (!prog || !len_out) return ZK_ERR_INVALID_ARG;
    std::string src;
    int st = ZK_ERR_INVALID_ARG;
    FIELD_SWITCH(f, st = expr_source_run<F>(prog, n_ops, n_cols, n_consts, src));
    ZK_TRY(st);
    *len_out = src.size();
    if (out && cap) {
        const size_t k = src.size() < cap - 1 ? src.size() : (size_t)cap - 1;
        memcpy(out, src.data(), k);
        out[k] = 0;
    }
    return ZK_OK;
}

API int zk_expr_configure(int jit_mode) {
    if (jit_mode < 0 || jit_mode > 2) return ZK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(g.mu);
    g.expr_jit = jit_mode;
    return ZK_OK;
}

API int zk_field_modulus(zk_field_t f, void* out) {
    if (!out) return ZK_ERR_INVALID_ARG;
    FIELD_SWITCH(f, memcpy(out, F::P, sizeof(uint32_t) * F::N));
    return ZK_OK;
}
API int zk_field_root_of_unity(zk_field_t f, uint32_t log_n, void* out) {
    if (!out) return ZK_ERR_INVALID_ARG;
    FIELD_SWITCH(f, {
        if (log_n > (uint32_t)F::TWO_ADICITY) return ZK_ERR_INVALID_ARG;
        Fe<F> w;
        for (int i = 0; i < F::N; i++) w.v[i] = F::ROOT[i];
        for (uint32_t i = log_n; i < (uint32_t)F::TWO_ADICITY; i++) fe_sqr(w, w);
        host_store(out, w);
    });
    return ZK_OK;
}
API int zk_field_multiplicative_generator(zk_field_t f, void* out) {
    if (!out) return ZK_ERR_INVALID_ARG;
    FIELD_SWITCH(f, {
        Fe<F> w;
        for (int i = 0; i < F::N; i++) w.v[i] = F::GEN[i];
        host_store(out, w);
    });
    return ZK_OK;
}
API int zk_field_inverse(zk_field_t f, const void* a, void* out) {
    if (!a || !out) return ZK_ERR_INVALID_ARG;
    FIELD_SWITCH(f, {
        Fe<F> x;
        host_load(x, a);
        fe_inv(x, x);
        host_store(out, x);
    });
    return ZK_OK;
}
API int zk_point_add(zk_curve_t c, const void* ja, const void* jb, void* jout) {
    if (!ja || !jb || !jout) return ZK_ERR_INVALID_ARG;
    return point_add_host(c, ja, jb, jout);
}
API int zk_point_to_affine(zk_curve_t c, const void* jac, void* aff) {
    if (!jac || !aff) return ZK_ERR_INVALID_ARG;
    CURVE_SWITCH(c, {
        Jacobian<C> a;
        memcpy(&a, jac, 3 * 4 * coord_words<C>());
        XYZZ<C> x;
        jac_to_xyzz(x, a);
        Affine<C> r;
        xyzz_to_affine(r, x);
        memcpy(aff, &r, 2 * 4 * coord_words<C>());
    });
    return ZK_OK;
}

API int zk_fixed_base_mul_device(zk_curve_t c, const void* d_scalars, uint64_t n, void* d_out, void* stream) {
    if (n == 0) return ZK_OK;
    if (!d_scalars || !d_out || !aligned16(d_scalars) || !aligned16(d_out) || n >= (1ull << 31)) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(d_out);
    CURVE_SWITCH(c, return fixed_base_run<C>((const Fe<typename C::Fr>*)d_scalars, n, (Affine<C>*)d_out, (hipStream_t)stream));
    return ZK_OK;
}

API int zk_fixed_base_msm_device(zk_curve_t c, const void* base_affine_mont, const void* d_scalars, uint64_t n, int scalars_are_montgomery,
                                 void* d_out, void* stream) {
    if (n == 0) return ZK_OK;
    if (!d_scalars || !d_out || !aligned16(d_scalars) || !aligned16(d_out) || n >= (1ull << 31)) return ZK_ERR_INVALID_ARG;
    DEVICE_ENTRY(d_out);
    CURVE_SWITCH(c, {
        Affine<C> base;
        if (base_affine_mont)
            memcpy(&base, base_affine_mont, 2 * 4 * coord_words<C>());
        else
            curve_generator(base);
        return fixed_base_msm_run<C>(dc, base, (const Fe<typename C::Fr>*)d_scalars, n, scalars_are_montgomery ? 1 : 0, (Affine<C>*)d_out,
                                     (hipStream_t)stream);
    });
    return ZK_OK;
}

}  // extern "C"
