// Launch sequences of the scalar-field side (NTT, pointwise, polynomial helpers), instantiated per field in
// zk_ntt_inst.cc.  Kept apart from zk_internal.h so that adding a field-side entry point does not rebuild the
// (slow) curve units.
#pragma once
#include <string>
#include "zk_internal.h"
#include "zkcp_amd_prover.h"
namespace zk {
constexpr uint32_t R1CS_LONG_ROW = 64;   // rows with more terms are summed by a whole workgroup (zk_r1cs_kernels.h)
// one R1CS matrix (A, B or C of ark-relations' ConstraintMatrices) in CSR form, resident on the home device
struct R1csMatrix {
    int field = 0;
    uint64_t n_rows = 0, n_cols = 0, nnz = 0, n_long = 0;
    void *row_ptr = nullptr, *col = nullptr, *val = nullptr, *long_rows = nullptr;
};
template <class F>
int ntt_run(DeviceCtx& dc, int field, Fe<F>* a, uint32_t logn, const Fe<F>& omega, int scale_flag, hipStream_t st,
            const Fe<F>* g_pre = nullptr, const Fe<F>* g_post = nullptr, uint32_t in_log = 0, const Fe<F>* src0 = nullptr);
template <class F>
int coset_run(DeviceCtx& dc, int field, Fe<F>* a, uint32_t logn, const Fe<F>& gshift, hipStream_t st);
template <class F>
int vec_op_run(Fe<F>* a, const Fe<F>* b, const Fe<F>* c, uint64_t n, int op, const Fe<F>& s, hipStream_t st);
template <class F>
int scale_periodic_run(Fe<F>* a, uint64_t n, const Fe<F>* table_host, uint32_t m, hipStream_t st);
// halo2 prover steps beyond commit / FFT (zk_poly.inl)
template <class F>
int batch_invert_run(Fe<F>* a, uint64_t n, hipStream_t st);
template <class F>
int prefix_product_run(DeviceCtx& dc, const Fe<F>* in, Fe<F>* out, uint64_t n, const Fe<F>& first, Fe<F>** total_dev, hipStream_t st);
template <class F>
int perm_product_run(DeviceCtx& dc, int field, uint32_t ncols, const void* const* cols, const void* const* sigmas, uint32_t first_col,
                     const Fe<F>& beta, const Fe<F>& gamma, const Fe<F>& delta, uint32_t k, const Fe<F>& omega, const Fe<F>& first, Fe<F>* z_out,
                     void* total_host, hipStream_t st);
template <class F>
int lookup_product_run(DeviceCtx& dc, const Fe<F>* A, const Fe<F>* S, const Fe<F>* Ap, const Fe<F>* Sp, const Fe<F>& beta, const Fe<F>& gamma,
                       uint64_t n, const Fe<F>& first, Fe<F>* z_out, void* total_host, hipStream_t st);
template <class F>
int inner_product_run(DeviceCtx& dc, const Fe<F>* a, const Fe<F>* b, uint64_t n, void* out_host, hipStream_t st);
template <class F>
int vec_muladd_run(Fe<F>* out, const Fe<F>* a, const Fe<F>* b, uint64_t n, const Fe<F>& s, hipStream_t st);
template <class F>
int poly_eval_run(DeviceCtx& dc, const Fe<F>* c, uint64_t n, uint32_t count, uint64_t stride, const Fe<F>& x, int field, void* out_host, hipStream_t st);
template <class F>
int vec_fold_many_run(Fe<F>* out, const Fe<F>* first, int64_t stride, uint32_t count, uint64_t n, const Fe<F>& s, hipStream_t st);
template <class F>
int ipa_fold_round_run(Fe<F>* p, Fe<F>* b, Fe<F>* W, uint64_t half, uint64_t m0, const Fe<F>& u, hipStream_t st);
template <class F>
int vec_powers_run(DeviceCtx& dc, Fe<F>* out, uint64_t n, const Fe<F>& x, hipStream_t st);
template <class F>
int kate_division_run(DeviceCtx& dc, const Fe<F>* a, Fe<F>* q, uint64_t n, const Fe<F>& x, hipStream_t st);
template <class F>
int vec_fold_run(Fe<F>* a, uint64_t half, const Fe<F>& c, hipStream_t st);
template <class F>
int ipa_virtual_scalars_run(const Fe<F>* p, const Fe<F>* W, Fe<F>* SL, Fe<F>* SR, uint64_t m0, uint64_t cur, hipStream_t st);
template <class F>
int ipa_round_begin_run(DeviceCtx& dc, const Fe<F>* p, const Fe<F>* b, const Fe<F>* W, Fe<F>* SL, Fe<F>* SR, uint64_t m0, uint64_t cur, hipStream_t st);
template <class F>
int ipa_round_end_run(DeviceCtx& dc, hipStream_t st, void* vl_host, void* vr_host);
template <class F>
int ipa_update_weights_run(Fe<F>* W, uint64_t m0, uint64_t bit, const Fe<F>& u, hipStream_t st);
template <class F>
int expr_eval_run(DeviceCtx& dc, const zk_expr_op* prog, uint32_t n_ops, const void* const* cols, uint32_t n_cols, const Fe<F>* consts,
                  uint32_t n_consts, uint32_t log_n, uint32_t rot_scale, Fe<F>* out, hipStream_t st);
template <class F>
int expr_source_run(const zk_expr_op* prog, uint32_t n_ops, uint32_t n_cols, uint32_t n_consts, std::string& out);
template <class F>
int expr_eval_lazy_run(DeviceCtx& dc, const zk_expr_op* prog, uint32_t n_ops, const void* const* cols, uint32_t n_cols, const Fe<F>* consts,
                       uint32_t n_consts, uint32_t log_n, uint32_t rot_scale, Fe<F>* out, hipStream_t st);
template <class F>
int r1cs_matvec_run(const R1csMatrix& m, const Fe<F>* z, Fe<F>* out, uint64_t out_len, hipStream_t st);
template <class F>
int witness_map_run(DeviceCtx& dc, int field, Fe<F>* a, Fe<F>* b, Fe<F>* c, uint32_t logm, hipStream_t st);
}  // namespace zk
