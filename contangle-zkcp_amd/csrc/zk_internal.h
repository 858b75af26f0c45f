// Internals shared by the translation units of libzkcp_amd.so: the process context (one GPU per
// process), grow-only device workspaces, and the per-curve / per-field launch sequences that are
// explicitly instantiated in zk_msm_inst.cc / zk_ntt_inst.cc (one TU per curve / field so the
// gfx950 code objects build in parallel).
#pragma once
#include "zkcp_amd.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <map>
#include <mutex>
#include <vector>

#include "zk_rt.h"
#include "zk_curve.h"

namespace zk {


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
    void* dev;     // affine points as uploaded (8 / 12 x u32 Montgomery words per coordinate)
    uint64_t n;
    bool owned;
    void* dev29;   // the same points in the F29 view (zk_curve29.h), built once at upload for the 8-word G1 curves
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
    int num_cus = 0;
    char info[256] = {0};
    std::map<uint64_t, BasesEntry> bases;
    uint64_t next_handle = 1;
    std::map<TwKey, TwEntry> tw;
    uint64_t tw_stamp = 0;
    size_t tw_bytes = 0;
    // workspaces (grow-only, reused across calls)
    DevBuf ntt_tmp, pow_tbl, fb_table, fb_tmp, msm_hot, msm_counts, msm_digits, msm_blockcnt, msm_stage_idx, msm_stage_low, msm_queue, msm_seg_out, msm_subacc, msm_sorted, msm_buckets, msm_part_a, msm_part_b, scratch_in, scratch_out;
    hipEvent_t ev[6];   // MSM phase events
    bool have_events = false;
    zk_msm_profile prof;
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


template <class F>
int ntt_run(int field, Fe<F>* a, uint32_t logn, const Fe<F>& omega, int scale_flag, hipStream_t st, const Fe<F>* g_pre = nullptr,
            const Fe<F>* g_post = nullptr, uint32_t in_log = 0);
template <class F>
int coset_run(int field, Fe<F>* a, uint32_t logn, const Fe<F>& gshift, hipStream_t st);
template <class F>
int vec_op_run(Fe<F>* a, const Fe<F>* b, const Fe<F>* c, uint64_t n, int op, const Fe<F>& s, hipStream_t st);
template <class F>
int witness_map_run(int field, Fe<F>* a, Fe<F>* b, Fe<F>* c, uint32_t logm, hipStream_t st);
template <class C>
int msm_run(const BasesEntry& be, const Fe<typename C::Fr>* d_scalars, uint64_t n, int mont, const zk_msm_opts* opts,
            void* out_jac, hipStream_t st);
template <class C>
int bases_prepare_run(BasesEntry& be);   // build the resident F29 copy where the curve has one
template <class C>
int fixed_base_run(const Fe<typename C::Fr>* d_scalars, uint64_t n, Affine<C>* d_out, hipStream_t st);
template <class C>
int fixed_base_msm_run(const Affine<C>& base, const Fe<typename C::Fr>* d_scalars, uint64_t n, int mont, Affine<C>* d_out, hipStream_t st);
int msm_pick_c(uint64_t n, int requested);
template <class C>
inline int msm_windows(int c) {
    return (C::Fr::BITS + 1 + c - 1) / c;
}
}  // namespace zk
