// One translation unit per scalar field: hipcc -DZK_FIELD=<PallasFp|PallasFq|Bn254Fr|Bls381Fr>
#include "zk_ntt.inl"
namespace zk {
template int ntt_run<ZK_FIELD>(DeviceCtx&, int, Fe<ZK_FIELD>*, uint32_t, const Fe<ZK_FIELD>&, int, hipStream_t, const Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, uint32_t, const Fe<ZK_FIELD>*);
template int coset_run<ZK_FIELD>(DeviceCtx&, int, Fe<ZK_FIELD>*, uint32_t, const Fe<ZK_FIELD>&, hipStream_t);
template int vec_op_run<ZK_FIELD>(Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, uint64_t, int, const Fe<ZK_FIELD>&, hipStream_t);
template int scale_periodic_run<ZK_FIELD>(Fe<ZK_FIELD>*, uint64_t, const Fe<ZK_FIELD>*, uint32_t, hipStream_t);
template int batch_invert_run<ZK_FIELD>(Fe<ZK_FIELD>*, uint64_t, hipStream_t);
template int prefix_product_run<ZK_FIELD>(DeviceCtx&, const Fe<ZK_FIELD>*, Fe<ZK_FIELD>*, uint64_t, const Fe<ZK_FIELD>&, Fe<ZK_FIELD>**, hipStream_t);
template int perm_product_run<ZK_FIELD>(DeviceCtx&, int, uint32_t, const void* const*, const void* const*, uint32_t, const Fe<ZK_FIELD>&, const Fe<ZK_FIELD>&,
                                        const Fe<ZK_FIELD>&, uint32_t, const Fe<ZK_FIELD>&, const Fe<ZK_FIELD>&, Fe<ZK_FIELD>*, void*, hipStream_t);
template int lookup_product_run<ZK_FIELD>(DeviceCtx&, const Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, const Fe<ZK_FIELD>&,
                                          const Fe<ZK_FIELD>&, uint64_t, const Fe<ZK_FIELD>&, Fe<ZK_FIELD>*, void*, hipStream_t);
template int inner_product_run<ZK_FIELD>(DeviceCtx&, const Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, uint64_t, void*, hipStream_t);
template int vec_muladd_run<ZK_FIELD>(Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, uint64_t, const Fe<ZK_FIELD>&, hipStream_t);
template int poly_eval_run<ZK_FIELD>(DeviceCtx&, const Fe<ZK_FIELD>*, uint64_t, uint32_t, uint64_t, const Fe<ZK_FIELD>&, int, void*, hipStream_t);
template int vec_fold_many_run<ZK_FIELD>(Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, int64_t, uint32_t, uint64_t, const Fe<ZK_FIELD>&, hipStream_t);
template int ipa_fold_round_run<ZK_FIELD>(Fe<ZK_FIELD>*, Fe<ZK_FIELD>*, Fe<ZK_FIELD>*, uint64_t, uint64_t, const Fe<ZK_FIELD>&, hipStream_t);
template int vec_powers_run<ZK_FIELD>(DeviceCtx&, Fe<ZK_FIELD>*, uint64_t, const Fe<ZK_FIELD>&, hipStream_t);
template int kate_division_run<ZK_FIELD>(DeviceCtx&, const Fe<ZK_FIELD>*, Fe<ZK_FIELD>*, uint64_t, const Fe<ZK_FIELD>&, hipStream_t);
template int vec_fold_run<ZK_FIELD>(Fe<ZK_FIELD>*, uint64_t, const Fe<ZK_FIELD>&, hipStream_t);
template int ipa_virtual_scalars_run<ZK_FIELD>(const Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, Fe<ZK_FIELD>*, Fe<ZK_FIELD>*, uint64_t, uint64_t, hipStream_t);
template int ipa_round_begin_run<ZK_FIELD>(DeviceCtx&, const Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, Fe<ZK_FIELD>*, Fe<ZK_FIELD>*, uint64_t, uint64_t, hipStream_t);
template int ipa_round_end_run<ZK_FIELD>(DeviceCtx&, hipStream_t, void*, void*);
template int ipa_update_weights_run<ZK_FIELD>(Fe<ZK_FIELD>*, uint64_t, uint64_t, const Fe<ZK_FIELD>&, hipStream_t);
template int expr_eval_run<ZK_FIELD>(DeviceCtx&, const zk_expr_op*, uint32_t, const void* const*, uint32_t, const Fe<ZK_FIELD>*, uint32_t, uint32_t, uint32_t,
                                     Fe<ZK_FIELD>*, hipStream_t);
template int expr_eval_lazy_run<ZK_FIELD>(DeviceCtx&, const zk_expr_op*, uint32_t, const void* const*, uint32_t, const Fe<ZK_FIELD>*, uint32_t, uint32_t, uint32_t,
                                          Fe<ZK_FIELD>*, hipStream_t);
template int expr_source_run<ZK_FIELD>(const zk_expr_op*, uint32_t, uint32_t, uint32_t, std::string&);
template int r1cs_matvec_run<ZK_FIELD>(const R1csMatrix&, const Fe<ZK_FIELD>*, Fe<ZK_FIELD>*, uint64_t, hipStream_t);
template int witness_map_run<ZK_FIELD>(DeviceCtx&, int, Fe<ZK_FIELD>*, Fe<ZK_FIELD>*, Fe<ZK_FIELD>*, uint32_t, hipStream_t);
}  // namespace zk
