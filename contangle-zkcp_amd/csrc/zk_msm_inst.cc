// One translation unit per curve: hipcc -DZK_CURVE=<Pallas|Vesta|Bn254G1|Bls381G1|Bn254G2|Bls381G2>
#include "zk_msm.inl"
namespace zk {
template int msm_enqueue<ZK_CURVE>(MsmJob&, const BasesCopy&, const Fe<ZK_CURVE::Fr>*, uint64_t, int, const MsmTuning&);
template int bases_prepare_run<ZK_CURVE>(BasesCopy&, uint64_t);
template int bases_refresh_run<ZK_CURVE>(const BasesCopy&, uint64_t, uint64_t, hipStream_t);
template int fixed_base_run<ZK_CURVE>(const Fe<ZK_CURVE::Fr>*, uint64_t, Affine<ZK_CURVE>*, hipStream_t);
template int fixed_base_msm_run<ZK_CURVE>(DeviceCtx&, const Affine<ZK_CURVE>&, const Fe<ZK_CURVE::Fr>*, uint64_t, int, Affine<ZK_CURVE>*, hipStream_t);
template int bases_precompute_run<ZK_CURVE>(BasesCopy&, uint64_t, int);
template int ipa_fold_bases_run<ZK_CURVE>(DeviceCtx&, Affine<ZK_CURVE>*, uint64_t, const Fe<ZK_CURVE::Fr>&, hipStream_t);
template int ipa_collapse_run<ZK_CURVE>(DeviceCtx&, const BasesCopy&, uint64_t, const Fe<ZK_CURVE::Fr>*, uint64_t, uint64_t, uint64_t, uint64_t, Affine<ZK_CURVE>*,
                                        hipStream_t);
}  // namespace zk
