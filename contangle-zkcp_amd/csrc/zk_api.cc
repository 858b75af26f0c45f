// C ABI implementation (include/zkcp_amd.h): plans, workspaces, launch sequences and the
// small host tail of the MSM.  Compiled with `hipcc -x hip --offload-arch=gfx950` into
// libzkcp_amd.so.  (The CPU test tier compiles the same file against tests/emu/emu_hip.h;
// see zk_rt.h.)
#include "zkcp_amd.h"

#include <stdio.h>
#include <string.h>

#include <chrono>
#include <map>
#include <mutex>
#include <vector>

#include "zk_kernels.h"

using namespace zk;

namespace {

#define HIP_TRY(expr)                                       \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) {                             \
            g.last_hip = (int)e_;                           \
            return e_ == hipErrorOutOfMemory ? ZK_ERR_OOM : ZK_ERR_HIP; \
        }                                                   \
    } while (0)
#define ZK_TRY(expr)               \
    do {                           \
        int s_ = (expr);           \
        if (s_ != ZK_OK) return s_; \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};
struct BasesEntry {
    int curve;
    void* dev;
    uint64_t n;
    bool owned;
};
struct TwKey {
    int field;
    uint32_t logn;
    uint32_t omega[8];
    bool operator<(const TwKey& o) const {
        if (field != o.field) return field < o.field;
        if (logn != o.logn) return logn < o.logn;
        return memcmp(omega, o.omega, sizeof omega) < 0;
    }
};
struct TwEntry {
    void* dev;
    size_t bytes;
    uint64_t stamp;
};

struct Ctx {
    std::mutex mu;
    bool inited = false;
    int device = -1;
    int last_hip = 0;
    char info[256] = {0};
    std::map<uint64_t, BasesEntry> bases;
    uint64_t next_handle = 1;
    std::map<TwKey, TwEntry> tw;
    uint64_t tw_stamp = 0;
    size_t tw_bytes = 0;
    // workspaces (grow-only, reused across calls)
    DevBuf ntt_tmp, pow_tbl, msm_counts, msm_sorted, msm_buckets, msm_part_a, msm_part_b, scratch_in, scratch_out;
    hipEvent_t ev[8];
    bool have_events = false;
    zk_msm_profile prof;
} g;

int ws_get(DevBuf& b, size_t bytes) {
    if (b.cap >= bytes && b.p) return ZK_OK;
    if (b.p) hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    HIP_TRY(hipMalloc(&b.p, want));
    b.cap = want;
    return ZK_OK;
}
void ws_free(DevBuf& b) {
    if (b.p) hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

template <class F>
void host_load(Fe<F>& r, const void* p) {
    memcpy(r.v, p, sizeof(uint32_t) * F::N);
}
template <class F>
void host_store(void* p, const Fe<F>& r) {
    memcpy(p, r.v, sizeof(uint32_t) * F::N);
}

#define FIELD_SWITCH(f, ...)                                         \
    switch (f) {                                                      \
        case ZK_FP_PALLAS: { using F = PallasFp; __VA_ARGS__; } break;       \
        case ZK_FQ_PALLAS: { using F = PallasFq; __VA_ARGS__; } break;       \
        case ZK_FR_BN254: { using F = Bn254Fr; __VA_ARGS__; } break;         \
        case ZK_FR_BLS12_381: { using F = Bls381Fr; __VA_ARGS__; } break;    \
        default: return ZK_ERR_INVALID_ARG;                           \
    }
#define CURVE_SWITCH(c, ...)                                         \
    switch (c) {                                                      \
        case ZK_PALLAS: { using C = Pallas; __VA_ARGS__; } break;            \
        case ZK_VESTA: { using C = Vesta; __VA_ARGS__; } break;              \
        case ZK_BN254_G1: { using C = Bn254G1; __VA_ARGS__; } break;         \
        case ZK_BLS12_381_G1: { using C = Bls381G1; __VA_ARGS__; } break;    \
        default: return ZK_ERR_INVALID_ARG;                           \
    }

int require_init() { return g.inited ? ZK_OK : ZK_ERR_NOT_INITIALIZED; }
bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// ------------------------------------------------------------------ NTT
struct NttPlan {
    int nd;
    int rd[4];
    int log_t[4];
};

// radix split: <= 10 bits per pass (R*T*32 B of LDS: 1024 x 2 = 64 KiB, 512 x 4 = 64 KiB)
NttPlan ntt_plan(uint32_t logn) {
    NttPlan p;
    memset(&p, 0, sizeof p);
    int max_r = 10;
    if (const char* e = getenv("ZK_NTT_MAX_LOGR")) {
        int v = atoi(e);
        if (v >= 1 && v <= 10) max_r = v;
    }
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
    if (const char* e = getenv("ZK_NTT_LOGT")) {
        int v = atoi(e);
        if (v >= 0 && v <= 4) want_t = v;
    }
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
int tw_table(const Fe<F>& omega, uint32_t logn, int field, hipStream_t st, const Fe<F>** out) {
    TwKey key;
    memset(&key, 0, sizeof key);
    key.field = field;
    key.logn = logn;
    memcpy(key.omega, omega.v, sizeof(uint32_t) * F::N);
    auto it = g.tw.find(key);
    if (it != g.tw.end()) {
        it->second.stamp = ++g.tw_stamp;
        *out = (const Fe<F>*)it->second.dev;
        return ZK_OK;
    }
    // evict least-recently-used tables beyond 8 entries / 2 GiB
    while (g.tw.size() >= 8 || g.tw_bytes > (2ull << 30)) {
        auto victim = g.tw.begin();
        for (auto i2 = g.tw.begin(); i2 != g.tw.end(); ++i2)
            if (i2->second.stamp < victim->second.stamp) victim = i2;
        HIP_TRY(hipStreamSynchronize(st));
        hipFree(victim->second.dev);
        g.tw_bytes -= victim->second.bytes;
        g.tw.erase(victim);
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
    ZK_TRY(ws_get(g.pow_tbl, sizeof(Fe<F>) * 64));
    HIP_TRY(hipMemcpyAsync(g.pow_tbl.p, tbl.data(), sizeof(Fe<F>) * tbl.size(), hipMemcpyHostToDevice, st));
    void* dev = nullptr;
    HIP_TRY(hipMalloc(&dev, sizeof(Fe<F>) * count));
    const unsigned blk = 256;
    ZK_LAUNCH((pow_table_kernel<F>), (unsigned)((count + blk - 1) / blk), blk, 0, st, (Fe<F>*)dev, (const Fe<F>*)g.pow_tbl.p,
              count, nbits);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));  // pow_tbl is reused by the next table build
    TwEntry e{dev, sizeof(Fe<F>) * count, ++g.tw_stamp};
    g.tw[key] = e;
    g.tw_bytes += e.bytes;
    *out = (const Fe<F>*)dev;
    return ZK_OK;
}

template <class F>
int ntt_run(int field, Fe<F>* a, uint32_t logn, const Fe<F>& omega, int scale_flag, hipStream_t st) {
    if (logn == 0) return ZK_OK;  // size-1 transform is the identity, and n^-1 = 1
    if (logn > (uint32_t)F::TWO_ADICITY || logn > 30) return ZK_ERR_INVALID_ARG;
    const Fe<F>* tw = nullptr;
    ZK_TRY(tw_table<F>(omega, logn, field, st, &tw));
    Fe<F> scale;
    fe_one(scale);
    if (scale_flag) {
        Fe<F> nn;
        fe_zero(nn);
        nn.v[logn / 32] = 1u << (logn % 32);
        fe_to_mont(nn, nn);
        fe_inv(scale, nn);
    }
    NttPlan plan = ntt_plan(logn);
    Fe<F>* tmp = nullptr;
    if (plan.nd > 1) {
        ZK_TRY(ws_get(g.ntt_tmp, sizeof(Fe<F>) << logn));
        tmp = (Fe<F>*)g.ntt_tmp.p;
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
        A.scale = A.last ? scale_flag : 0;
        A.nd = plan.nd;
        for (int i = 0; i < plan.nd; i++) A.rd[i] = plan.rd[i];
        const Fe<F>* src;
        Fe<F>* dst;
        if (plan.nd == 1) {
            src = a;
            dst = a;
        } else if (p == 0) {
            src = a;
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
        unsigned blk = rt / 2 < 64 ? 64 : (rt / 2 > 256 ? 256 : rt / 2);
        const size_t shmem = (size_t)rt * sizeof(Fe<F>);
        if (shmem > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_kernel<F>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        ZK_LAUNCH((ntt_pass_kernel<F>), (unsigned)tiles, blk, shmem, st, src, dst, tw, A, scale);
        HIP_TRY(hipGetLastError());
        log_m += plan.rd[p];
    }
    return ZK_OK;
}

template <class F>
int coset_run(Fe<F>* a, uint32_t logn, const Fe<F>& gshift, hipStream_t st) {
    if (logn > 30) return ZK_ERR_INVALID_ARG;
    const uint64_t count = 1ull << logn;
    std::vector<Fe<F>> tbl(logn ? logn : 1);
    Fe<F> w = gshift;
    for (uint32_t k = 0; k < logn; k++) {
        tbl[k] = w;
        fe_sqr(w, w);
    }
    ZK_TRY(ws_get(g.pow_tbl, sizeof(Fe<F>) * 64));
    HIP_TRY(hipMemcpyAsync(g.pow_tbl.p, tbl.data(), sizeof(Fe<F>) * tbl.size(), hipMemcpyHostToDevice, st));
    const unsigned blk = 256;
    ZK_LAUNCH((coset_mul_kernel<F>), (unsigned)((count + blk - 1) / blk), blk, 0, st, a, (const Fe<F>*)g.pow_tbl.p, count,
              (int)logn);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));  // pow_tbl reuse
    return ZK_OK;
}

// ------------------------------------------------------------------ MSM
int msm_pick_c(uint64_t n, int requested) {
    if (requested > 0) return requested < 2 ? 2 : (requested > 20 ? 20 : requested);
    if (const char* e = getenv("ZK_MSM_C")) {
        int v = atoi(e);
        if (v >= 2 && v <= 20) return v;
    }
    int l = 0;
    while ((1ull << l) < n) l++;
    int c = l - 4;  // ~2^5 points per bucket per window
    if (c < 4) c = 4;
    if (c > 16) c = 16;
    return c;
}
template <class C>
int msm_windows(int c) {
    return (C::Fr::BITS + 1 + c - 1) / c;
}

double now_ms() {
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
        ZK_LAUNCH((msm_scan_kernel), 1, 1024, 0, st, (const uint32_t*)counts, offs, nbuckets);
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
        case ZK_ERR_BAD_HANDLE: return "unknown bases handle";
        default: return "unknown status";
    }
}

API int zk_init(int device_id) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.inited) return g.device == device_id ? ZK_OK : ZK_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ZK_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= ndev) return ZK_ERR_INVALID_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return ZK_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return ZK_ERR_NO_DEVICE;
#if defined(ZK_EMU)
    snprintf(g.info, sizeof g.info, "emu %s", prop.name);
#else
    snprintf(g.info, sizeof g.info, "hip %s %s cu=%d", prop.gcnArchName, prop.name, prop.multiProcessorCount);
#endif
    g.device = device_id;
    g.inited = true;
    return ZK_OK;
}

API int zk_shutdown(void) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.inited) return ZK_OK;
    hipDeviceSynchronize();
    for (auto& kv : g.bases)
        if (kv.second.owned) hipFree(kv.second.dev);
    g.bases.clear();
    for (auto& kv : g.tw) hipFree(kv.second.dev);
    g.tw.clear();
    g.tw_bytes = 0;
    for (DevBuf* b : {&g.ntt_tmp, &g.pow_tbl, &g.msm_counts, &g.msm_sorted, &g.msm_buckets, &g.msm_part_a, &g.msm_part_b,
                      &g.scratch_in, &g.scratch_out})
        ws_free(*b);
    if (g.have_events) {
        for (auto& e : g.ev) hipEventDestroy(e);
        g.have_events = false;
    }
    g.inited = false;
    g.device = -1;
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
    CURVE_SWITCH(c, return C::Fq::N / 2);
    return ZK_ERR_INVALID_ARG;
}
API int zk_curve_scalar_field(zk_curve_t c) {
    switch (c) {
        case ZK_PALLAS: return ZK_FQ_PALLAS;
        case ZK_VESTA: return ZK_FP_PALLAS;
        case ZK_BN254_G1: return ZK_FR_BN254;
        case ZK_BLS12_381_G1: return ZK_FR_BLS12_381;
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

API int zk_bases_upload(zk_curve_t c, const void* host, uint64_t n, uint64_t* handle_out) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!handle_out || (n && !host)) return ZK_ERR_INVALID_ARG;
    size_t esz = 0;
    CURVE_SWITCH(c, esz = sizeof(Affine<C>));
    void* dev = nullptr;
    HIP_TRY(hipMalloc(&dev, esz * (n ? n : 1)));
    if (n) HIP_TRY(hipMemcpy(dev, host, esz * n, hipMemcpyHostToDevice));
    const uint64_t h = g.next_handle++;
    g.bases[h] = BasesEntry{(int)c, dev, n, true};
    *handle_out = h;
    return ZK_OK;
}
API int zk_bases_adopt_device(zk_curve_t c, const void* dev, uint64_t n, uint64_t* handle_out) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!handle_out || !dev || !aligned16(dev)) return ZK_ERR_INVALID_ARG;
    CURVE_SWITCH(c, (void)sizeof(C));
    const uint64_t h = g.next_handle++;
    g.bases[h] = BasesEntry{(int)c, const_cast<void*>(dev), n, false};
    *handle_out = h;
    return ZK_OK;
}
API int zk_bases_free(uint64_t handle) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    auto it = g.bases.find(handle);
    if (it == g.bases.end()) return ZK_ERR_BAD_HANDLE;
    if (it->second.owned) {
        hipDeviceSynchronize();
        hipFree(it->second.dev);
    }
    g.bases.erase(it);
    return ZK_OK;
}

static int msm_locked(zk_curve_t c, uint64_t handle, const void* d_scalars, uint64_t n, int mont, const zk_msm_opts* opts,
                      void* out, hipStream_t st) {
    auto it = g.bases.find(handle);
    if (it == g.bases.end()) return ZK_ERR_BAD_HANDLE;
    if (it->second.curve != (int)c || n > it->second.n) return ZK_ERR_INVALID_ARG;
    CURVE_SWITCH(c, return msm_run<C>(it->second, (const Fe<typename C::Fr>*)d_scalars, n, mont, opts, out, st));
    return ZK_ERR_INVALID_ARG;
}

API int zk_msm_device(zk_curve_t c, uint64_t handle, const void* d_scalars, uint64_t n, int mont, const zk_msm_opts* opts,
                      void* out, void* stream) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!out || (n && (!d_scalars || !aligned16(d_scalars)))) return ZK_ERR_INVALID_ARG;
    return msm_locked(c, handle, d_scalars, n, mont, opts, out, (hipStream_t)stream);
}

API int zk_msm(zk_curve_t c, uint64_t handle, const void* scalars_host, uint64_t n, int mont, const zk_msm_opts* opts,
               void* out) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!out || (n && !scalars_host)) return ZK_ERR_INVALID_ARG;
    ZK_TRY(ws_get(g.scratch_in, 32 * (n ? n : 1)));
    if (n) HIP_TRY(hipMemcpy(g.scratch_in.p, scalars_host, 32 * n, hipMemcpyHostToDevice));
    return msm_locked(c, handle, g.scratch_in.p, n, mont, opts, out, (hipStream_t)0);
}

API int zk_msm_last_profile(zk_msm_profile* out) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (!out) return ZK_ERR_INVALID_ARG;
    *out = g.prof;
    return ZK_OK;
}

API int zk_ntt_device(zk_field_t f, void* a, uint32_t log_n, const void* omega, int scale, void* stream) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!a || !omega || !aligned16(a)) return ZK_ERR_INVALID_ARG;
    FIELD_SWITCH(f, {
        Fe<F> w;
        host_load(w, omega);
        return ntt_run<F>((int)f, (Fe<F>*)a, log_n, w, scale, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_ntt(zk_field_t f, void* a_host, uint32_t log_n, const void* omega, int scale) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!a_host || !omega || log_n > 30) return ZK_ERR_INVALID_ARG;
    const size_t bytes = (size_t)32 << log_n;
    ZK_TRY(ws_get(g.scratch_in, bytes));
    HIP_TRY(hipMemcpy(g.scratch_in.p, a_host, bytes, hipMemcpyHostToDevice));
    FIELD_SWITCH(f, {
        Fe<F> w;
        host_load(w, omega);
        ZK_TRY(ntt_run<F>((int)f, (Fe<F>*)g.scratch_in.p, log_n, w, scale, (hipStream_t)0));
    });
    HIP_TRY(hipStreamSynchronize((hipStream_t)0));
    HIP_TRY(hipMemcpy(a_host, g.scratch_in.p, bytes, hipMemcpyDeviceToHost));
    return ZK_OK;
}
API int zk_coset_mul_device(zk_field_t f, void* a, uint32_t log_n, const void* gm, void* stream) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!a || !gm || !aligned16(a)) return ZK_ERR_INVALID_ARG;
    FIELD_SWITCH(f, {
        Fe<F> w;
        host_load(w, gm);
        return coset_run<F>((Fe<F>*)a, log_n, w, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_coset_mul(zk_field_t f, void* a_host, uint32_t log_n, const void* gm) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!a_host || !gm || log_n > 30) return ZK_ERR_INVALID_ARG;
    const size_t bytes = (size_t)32 << log_n;
    ZK_TRY(ws_get(g.scratch_in, bytes));
    HIP_TRY(hipMemcpy(g.scratch_in.p, a_host, bytes, hipMemcpyHostToDevice));
    FIELD_SWITCH(f, {
        Fe<F> w;
        host_load(w, gm);
        ZK_TRY(coset_run<F>((Fe<F>*)g.scratch_in.p, log_n, w, (hipStream_t)0));
    });
    HIP_TRY(hipMemcpy(a_host, g.scratch_in.p, bytes, hipMemcpyDeviceToHost));
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
    CURVE_SWITCH(c, {
        Jacobian<C> a, b, r;
        memcpy(&a, ja, 3 * 4 * C::Fq::N);
        memcpy(&b, jb, 3 * 4 * C::Fq::N);
        XYZZ<C> xa, xb;
        jac_to_xyzz(xa, a);
        jac_to_xyzz(xb, b);
        xyzz_add(xa, xb);
        xyzz_to_jacobian(r, xa);
        memcpy(jout, &r, 3 * 4 * C::Fq::N);
    });
    return ZK_OK;
}
API int zk_point_to_affine(zk_curve_t c, const void* jac, void* aff) {
    if (!jac || !aff) return ZK_ERR_INVALID_ARG;
    CURVE_SWITCH(c, {
        Jacobian<C> a;
        memcpy(&a, jac, 3 * 4 * C::Fq::N);
        XYZZ<C> x;
        jac_to_xyzz(x, a);
        Affine<C> r;
        xyzz_to_affine(r, x);
        memcpy(aff, &r, 2 * 4 * C::Fq::N);
    });
    return ZK_OK;
}

API int zk_fixed_base_mul_device(zk_curve_t c, const void* d_scalars, uint64_t n, void* d_out, void* stream) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (n == 0) return ZK_OK;
    if (!d_scalars || !d_out || !aligned16(d_scalars) || !aligned16(d_out) || n >= (1ull << 31)) return ZK_ERR_INVALID_ARG;
    CURVE_SWITCH(c, {
        ZK_LAUNCH((fixed_base_mul_kernel<C>), (unsigned)((n + 63) / 64), 64, 0, (hipStream_t)stream,
                  (const Fe<typename C::Fr>*)d_scalars, (Affine<C>*)d_out, (uint32_t)n);
        HIP_TRY(hipGetLastError());
    });
    return ZK_OK;
}

}  // extern "C"
