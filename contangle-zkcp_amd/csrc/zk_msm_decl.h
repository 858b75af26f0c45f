// Launch sequences of the curve side (MSM, fixed-base), instantiated per curve in zk_msm_inst.cc.
#pragma once
#include "zk_internal.h"
namespace zk {
// enqueue the device work of one MSM on job.stream; the result is produced by job.finish after the job's last event
template <class C>
int msm_enqueue(MsmJob& job, const BasesCopy& bc, const Fe<typename C::Fr>* d_scalars, uint64_t n, int mont, const MsmTuning& tu);
template <class C>
int bases_prepare_run(BasesCopy& bc, uint64_t n);   // build the resident lazy-limb copy
template <class C>
int bases_precompute_run(BasesCopy& bc, uint64_t n, int c);   // window multiples for ZK_MSM_FLAG_PRECOMPUTED
template <class C>
int bases_refresh_run(const BasesCopy& bc, uint64_t offset, uint64_t count, hipStream_t st);   // re-derive it for a rewritten range
template <class C>
int fixed_base_run(const Fe<typename C::Fr>* d_scalars, uint64_t n, Affine<C>* d_out, hipStream_t st);
template <class C>
int fixed_base_msm_run(DeviceCtx& dc, const Affine<C>& base, const Fe<typename C::Fr>* d_scalars, uint64_t n, int mont, Affine<C>* d_out,
                       hipStream_t st);
template <class C>
int ipa_fold_bases_run(DeviceCtx& dc, Affine<C>* g, uint64_t half, const Fe<typename C::Fr>& u_canonical, hipStream_t st);
template <class C>
int ipa_collapse_run(DeviceCtx& dc, const BasesCopy& bc, uint64_t base_n, const Fe<typename C::Fr>* w_dev, uint64_t m0, uint64_t cur, uint64_t first,
                     uint64_t count, Affine<C>* g_out, hipStream_t st);
}  // namespace zk
