// Internals shared by the translation units of libzkcp_amd.so: the process context (one DeviceCtx per GPU the
// process drives), per-stream scratch, the pool of MSM jobs (deferred results: enqueue -> event -> collect), and the
// per-curve / per-field launch sequences that are explicitly instantiated in zk_msm_inst.cc / zk_ntt_inst.cc (one TU
// per curve / field so the gfx950 code objects build in parallel).
#pragma once
#include "zkcp_amd.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "zk_rt.h"
#include "zk_curve.h"

namespace zk {

struct DeviceCtx;

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
struct BasesCopy {
    void* dev = nullptr;     // affine points as uploaded (8 / 12 x u32 Montgomery words per coordinate)
    void* dev29 = nullptr;   // the same points in the lazy-limb view (zk_curve29.h), built once at upload
    bool owned = false;
    void* pre = nullptr;     // optional (zk_bases_precompute): [2^(c w)] P_i for every window w, lazy-limb view, window-major
    int pre_c = 0, pre_w = 0;
};
struct BasesEntry {
    int curve;
    uint64_t n;
    std::vector<BasesCopy> per_dev;   // one resident copy per device of the process (the SRS is fixed: uploaded once)
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

// resolved MSM plan knobs (zk_msm_opts; 0 / negative = choose automatically)
struct MsmTuning {
    int window_bits = 0, w0 = 0, w1 = 0;
    int split_log = -1;
    uint32_t slice_len = 0;
    uint32_t big_thresh = 0;
    int limb_bits = 0;        // 32 forces the saturated path
    int waves = 0;
    int window_group = 0;     // windows per group of a large single MSM (0 = automatic)
    bool no_hot_help = false;
    bool slice_reduce = false;   // the round-1 bucket reduction (slices + multiplier) instead of row / column sums
    bool device_partials = false;   // ZK_MSM_FLAG_DEVICE_PARTIALS: partial sums converted on the device, checked against the host's conversion
    bool precomputed = false;    // ONE bucket set over the handle's precomputed window multiples (ZK_MSM_FLAG_PRECOMPUTED)
    uint64_t base_offset = 0;
    uint32_t batch = 1;          // internal (zk_msm_batch_device): scalar vectors summed by ONE job
    uint64_t batch_stride = 0;   // elements between them
};

// One MSM in flight: its own device workspaces (so two jobs on two streams never share a buffer), a pinned host buffer
// the per-window partial sums are copied to, the events of its phases and the per-curve routine that finishes it on the
// host (Horner over the windows).  A job is reused only after it has been collected.
struct MsmJob {
    DeviceCtx* dc = nullptr;
    bool busy = false;
    uint64_t ticket = 0;
    hipStream_t stream = nullptr;
    DevBuf scalars_in;                // staging for scalars that arrive from the host or from a peer device
    DevBuf hot, counts, digits, blockcnt, stage_idx, stage_low, queue, seg_out, subacc, sorted, buckets, part_a, part_b, part_std;
    void* host_partials = nullptr;    // pinned
    size_t host_cap = 0;
    hipEvent_t ev[8];                 // [0] begin .. [5] reduced, [6] partials on the host, [7] accumulate kernel done
    bool have_events = false;
    std::vector<hipEvent_t> acc_ev;   // begin / end of the accumulate kernel, one pair per window group
    int groups = 1;
    // what collect needs
    int (*finish)(MsmJob&, void* out_jac) = nullptr;
    int curve = 0, c = 0, w0 = 0, nw = 0;
    uint32_t batch = 1;               // results this job produces (windows [b nw, (b + 1) nw) of the partial sums belong to vector b)
    uint32_t per = 0;                 // partial sums per window
    bool axes = false;                // row / column reduction: per window [row blocks | column blocks] x (weighted, plain)
    uint32_t row_blocks = 0, col_blocks = 0, log_cols = 0, log_tl = 0;
    bool dev_std = false;             // host_partials = [raw | converted on the device | conversion stages] (ZK_MSM_FLAG_DEVICE_PARTIALS)
    bool empty = true;                // nothing was launched (n == 0 or no windows): the result is the identity
    zk_msm_profile prof;
    double alg_bytes = 0;             // n x (scalar + affine point bytes) x share of the windows
};

// scratch that belongs to one caller stream (multi-pass NTT ping-pong buffer, fixed-base table / temporaries): two
// NTTs on two streams run over different buffers
struct StreamScratch {
    hipStream_t stream = nullptr;
    uint64_t stamp = 0;
    DevBuf ntt_tmp, fb_table, fb_tmp;
    DevBuf poly_a, poly_b, poly_tot;   // numerators / denominators / block totals of the grand-product and IPA helpers
    void* pinned = nullptr;            // small pinned host buffer: results that are read back without stalling the stream at once
    size_t pinned_cap = 0;
    hipEvent_t pinned_ev = nullptr;    // recorded after the copies into `pinned`
    uint32_t ip_blocks = 0;            // partial sums per inner product waiting in `pinned` (zk_ipa_round_device)
};

// One persistent host thread per device of a multi-device process (zk_init_devices with n > 1): bound to its device once,
// it runs that device's share of every fanned-out call -- enqueue, wait, the Horner tail of its MSM jobs -- so that the
// shares proceed in parallel without a thread being created per call.
struct DeviceWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::packaged_task<int()>> q;
    bool stop = false;
};

constexpr int ZK_MAX_JOBS = 4;
constexpr int ZK_MAX_STREAM_SCRATCH = 6;

struct DeviceCtx {
    int device = -1;
    int index = 0;            // position in Ctx::devs
    int num_cus = 0;
    std::mutex mu;            // serialises host-side enqueue on this device
    std::map<TwKey, TwEntry> tw;
    uint64_t tw_stamp = 0;
    size_t tw_bytes = 0;
    DevBuf pow_tbl, scratch_in;
    std::vector<std::unique_ptr<StreamScratch>> scratch;
    uint64_t scratch_stamp = 0;
    MsmJob jobs[ZK_MAX_JOBS];
    hipStream_t side[2] = {nullptr, nullptr};   // library-owned streams: batched MSMs alternate between them
    hipStream_t submit_streams[ZK_MAX_JOBS] = {nullptr, nullptr, nullptr, nullptr};   // ZK_MSM_FLAG_OWN_STREAM: one per MSM in flight
    unsigned submit_rr = 0;
    hipStream_t own = nullptr;                  // library-owned stream of a device that is not the caller's (multi-device fan-out)
    std::unique_ptr<DeviceWorker> worker;       // multi-device processes only
    DevBuf batch_in;                            // scalar vectors of a fanned-out batch that live on a peer
    std::mutex batch_mu;                        // ... one fanned-out batch at a time per device
    hipEvent_t fork_ev = nullptr, join_ev[2] = {nullptr, nullptr};
    // NTT pass timing (zk_ntt_profile_enable): event pairs of the launches since the last read
    std::vector<hipEvent_t> ntt_ev_pool;
    size_t ntt_ev_used = 0;
    uint64_t ntt_transforms = 0;
    double ntt_alg_bytes = 0;
};

struct Ctx {
    std::mutex mu;            // lifecycle, handles
    bool inited = false;
    int last_hip = 0;
    char info[256] = {0};
    std::vector<std::unique_ptr<DeviceCtx>> devs;
    std::map<uint64_t, BasesEntry> bases;
    uint64_t next_handle = 1;
    uint64_t next_ticket = 1;
    std::map<uint64_t, std::vector<MsmJob*>> tickets;   // one job, or one per device of a fanned-out submission
    zk_msm_profile prof;      // of the last collected job
    zk_ntt_opts ntt_opts;     // process-wide NTT plan knobs (zk_ntt_configure)
    zk_msm_totals totals;     // sums over the collected jobs (zk_msm_profile_totals)
    bool ntt_profile = false;
    int expr_jit = 0;         // zk_expr_configure: 0 = specialised quotient kernel for large evaluations, 1 = always, 2 = never (interpreter)
};
extern Ctx g;

inline int ws_get(DevBuf& b, size_t bytes) {
    if (b.cap >= bytes && b.p) return ZK_OK;
    if (b.p) hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    HIP_TRY(hipMalloc(&b.p, want));
    b.cap = want;
    return ZK_OK;
}
inline void ws_free(DevBuf& b) {
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
        case ZK_BN254_G2: { using C = Bn254G2; __VA_ARGS__; } break;         \
        case ZK_BLS12_381_G2: { using C = Bls381G2; __VA_ARGS__; } break;    \
        default: return ZK_ERR_INVALID_ARG;                           \
    }

inline int require_init() { return g.inited ? ZK_OK : ZK_ERR_NOT_INITIALIZED; }
inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// HIP's current device is per host thread: every entry point binds the calling thread to the device it is about to use
// (upstream callers arrive on rayon worker threads that have never seen hipSetDevice)
inline int bind_device(DeviceCtx& dc) {
    HIP_TRY(hipSetDevice(dc.device));
    return ZK_OK;
}
int stream_scratch(DeviceCtx& dc, hipStream_t st, StreamScratch** out);   // zk_api.cc

int msm_pick_c(uint64_t n, int requested, bool pre = false);
template <class C>
inline int msm_windows(int c) {
    return (C::Fr::BITS + 1 + c - 1) / c;
}
}  // namespace zk
