// Launch sequences of the scalar-field side (NTT, pointwise, polynomial helpers), instantiated per field in
// zk_ntt_inst.cc.  Kept apart from zk_internal.h so that adding a field-side entry point does not rebuild the
// (slow) curve units.
#pragma once
#include "zk_internal.h"
namespace zk {
template <class F>
int ntt_run(DeviceCtx& dc, int field, Fe<F>* a, uint32_t logn, const Fe<F>& omega, int scale_flag, hipStream_t st,
            const Fe<F>* g_pre = nullptr, const Fe<F>* g_post = nullptr, uint32_t in_log = 0);
template <class F>
int coset_run(DeviceCtx& dc, int field, Fe<F>* a, uint32_t logn, const Fe<F>& gshift, hipStream_t st);
template <class F>
int vec_op_run(Fe<F>* a, const Fe<F>* b, const Fe<F>* c, uint64_t n, int op, const Fe<F>& s, hipStream_t st);
template <class F>
int scale_periodic_run(Fe<F>* a, uint64_t n, const Fe<F>* table_host, uint32_t m, hipStream_t st);
template <class F>
int witness_map_run(DeviceCtx& dc, int field, Fe<F>* a, Fe<F>* b, Fe<F>* c, uint32_t logm, hipStream_t st);
}  // namespace zk
