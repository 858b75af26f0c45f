// C ABI implementation (include/zkcp_amd.h): argument checks, dispatch over curve / field and the
// host-side helpers.  The launch sequences live in zk_msm.inl / zk_ntt.inl (one TU per curve /
// field).  Compiled with `hipcc -x hip --offload-arch=gfx950` into libzkcp_amd.so.  (The CPU test
// tier compiles the same files against tests/emu/emu_hip.h; see zk_rt.h.)
#include "zk_internal.h"

using namespace zk;

namespace zk {
Ctx g;
int msm_pick_c(uint64_t n, int requested) {
    if (requested > 0) return requested < 2 ? 2 : (requested > 16 ? 16 : requested);  // digits are stored as u16 codes
    if (const char* e = getenv("ZK_MSM_C")) {
        int v = atoi(e);
        if (v >= 2 && v <= 16) return v;
    }
    int l = 0;
    while ((1ull << l) < n) l++;
    int c = l - 4;  // ~2^5 points per bucket per window
    if (c < 4) c = 4;
    if (c > 16) c = 16;
    return c;
}
}  // namespace zk

namespace {
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
    g.num_cus = prop.multiProcessorCount;
    g.device = device_id;
    g.inited = true;
    return ZK_OK;
}

API int zk_shutdown(void) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.inited) return ZK_OK;
    hipDeviceSynchronize();
    for (auto& kv : g.bases) {
        if (kv.second.owned) hipFree(kv.second.dev);
        if (kv.second.dev29) hipFree(kv.second.dev29);
    }
    g.bases.clear();
    for (auto& kv : g.tw) hipFree(kv.second.dev);
    g.tw.clear();
    g.tw_bytes = 0;
    for (DevBuf* b : {&g.ntt_tmp, &g.pow_tbl, &g.fb_table, &g.fb_tmp, &g.msm_hot, &g.msm_counts, &g.msm_digits, &g.msm_blockcnt, &g.msm_stage_idx, &g.msm_stage_low, &g.msm_queue, &g.msm_seg_out, &g.msm_subacc, &g.msm_sorted, &g.msm_buckets, &g.msm_part_a, &g.msm_part_b,
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
    BasesEntry be{(int)c, dev, n, true, nullptr};
    CURVE_SWITCH(c, {
        int st = bases_prepare_run<C>(be);
        if (st != ZK_OK) {
            hipFree(dev);
            return st;
        }
    });
    g.bases[h] = be;
    *handle_out = h;
    return ZK_OK;
}
API int zk_bases_adopt_device(zk_curve_t c, const void* dev, uint64_t n, uint64_t* handle_out) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!handle_out || !dev || !aligned16(dev)) return ZK_ERR_INVALID_ARG;
    BasesEntry be{(int)c, const_cast<void*>(dev), n, false, nullptr};
    CURVE_SWITCH(c, ZK_TRY(bases_prepare_run<C>(be)));
    const uint64_t h = g.next_handle++;
    g.bases[h] = be;
    *handle_out = h;
    return ZK_OK;
}
API int zk_bases_free(uint64_t handle) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    auto it = g.bases.find(handle);
    if (it == g.bases.end()) return ZK_ERR_BAD_HANDLE;
    if (it->second.owned || it->second.dev29) hipDeviceSynchronize();
    if (it->second.owned) hipFree(it->second.dev);
    if (it->second.dev29) hipFree(it->second.dev29);
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
API int zk_ntt_coset_device(zk_field_t f, void* a, uint32_t log_n, const void* omega, int scale, const void* g_pre,
                            const void* g_post, void* stream) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!a || !omega || !aligned16(a)) return ZK_ERR_INVALID_ARG;
    FIELD_SWITCH(f, {
        Fe<F> w, gp, gq;
        host_load(w, omega);
        if (g_pre) host_load(gp, g_pre);
        if (g_post) host_load(gq, g_post);
        return ntt_run<F>((int)f, (Fe<F>*)a, log_n, w, scale, (hipStream_t)stream, g_pre ? &gp : nullptr, g_post ? &gq : nullptr);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_ntt_extend_device(zk_field_t f, void* a, uint32_t log_n, uint32_t log_in, const void* omega, int scale, const void* g_pre,
                             const void* g_post, void* stream) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!a || !omega || !aligned16(a) || log_in > log_n) return ZK_ERR_INVALID_ARG;
    FIELD_SWITCH(f, {
        Fe<F> w, gp, gq;
        host_load(w, omega);
        if (g_pre) host_load(gp, g_pre);
        if (g_post) host_load(gq, g_post);
        if (log_n > 0 && log_in == 0) {   // a single coefficient: in_log = 0 means "all" to the pass kernel, so pad explicitly
            HIP_TRY(hipMemsetAsync((Fe<F>*)a + 1, 0, (((size_t)1 << log_n) - 1) * sizeof(Fe<F>), (hipStream_t)stream));
        }
        return ntt_run<F>((int)f, (Fe<F>*)a, log_n, w, scale, (hipStream_t)stream, g_pre ? &gp : nullptr, g_post ? &gq : nullptr, log_in);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_coset_mul_device(zk_field_t f, void* a, uint32_t log_n, const void* gm, void* stream) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!a || !gm || !aligned16(a)) return ZK_ERR_INVALID_ARG;
    FIELD_SWITCH(f, {
        Fe<F> w;
        host_load(w, gm);
        return coset_run<F>((int)f, (Fe<F>*)a, log_n, w, (hipStream_t)stream);
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
        ZK_TRY(coset_run<F>((int)f, (Fe<F>*)g.scratch_in.p, log_n, w, (hipStream_t)0));
    });
    HIP_TRY(hipMemcpy(a_host, g.scratch_in.p, bytes, hipMemcpyDeviceToHost));
    return ZK_OK;
}

API int zk_vec_op_device(zk_field_t f, int op, void* a, const void* b, const void* c, uint64_t n, const void* scalar, void* stream) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (op < 0 || op > 6 || (n && (!a || !aligned16(a)))) return ZK_ERR_INVALID_ARG;
    const bool needs_b = op == 0 || op == 1 || op == 2 || op == 6, needs_c = op == 6, needs_s = op == 3 || op == 6;
    if (n && ((needs_b && (!b || !aligned16(b))) || (needs_c && (!c || !aligned16(c))) || (needs_s && !scalar))) return ZK_ERR_INVALID_ARG;
    FIELD_SWITCH(f, {
        Fe<F> s;
        fe_one(s);
        if (needs_s) host_load(s, scalar);
        return vec_op_run<F>((Fe<F>*)a, (const Fe<F>*)b, (const Fe<F>*)c, n, op, s, (hipStream_t)stream);
    });
    return ZK_ERR_INVALID_ARG;
}
API int zk_groth16_witness_map_device(zk_field_t f, void* a, void* b, void* c, uint32_t log_m, void* stream) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (!a || !b || !c || !aligned16(a) || !aligned16(b) || !aligned16(c)) return ZK_ERR_INVALID_ARG;
    FIELD_SWITCH(f, return witness_map_run<F>((int)f, (Fe<F>*)a, (Fe<F>*)b, (Fe<F>*)c, log_m, (hipStream_t)stream));
    return ZK_ERR_INVALID_ARG;
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
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (n == 0) return ZK_OK;
    if (!d_scalars || !d_out || !aligned16(d_scalars) || !aligned16(d_out) || n >= (1ull << 31)) return ZK_ERR_INVALID_ARG;
    CURVE_SWITCH(c, return fixed_base_run<C>((const Fe<typename C::Fr>*)d_scalars, n, (Affine<C>*)d_out, (hipStream_t)stream));
    return ZK_OK;
}

API int zk_fixed_base_msm_device(zk_curve_t c, const void* base_affine_mont, const void* d_scalars, uint64_t n, int scalars_are_montgomery,
                                 void* d_out, void* stream) {
    std::lock_guard<std::mutex> lk(g.mu);
    ZK_TRY(require_init());
    if (n == 0) return ZK_OK;
    if (!d_scalars || !d_out || !aligned16(d_scalars) || !aligned16(d_out) || n >= (1ull << 31)) return ZK_ERR_INVALID_ARG;
    CURVE_SWITCH(c, {
        Affine<C> base;
        if (base_affine_mont)
            memcpy(&base, base_affine_mont, 2 * 4 * coord_words<C>());
        else
            curve_generator(base);
        return fixed_base_msm_run<C>(base, (const Fe<typename C::Fr>*)d_scalars, n, scalars_are_montgomery ? 1 : 0, (Affine<C>*)d_out,
                                     (hipStream_t)stream);
    });
    return ZK_OK;
}

}  // extern "C"
