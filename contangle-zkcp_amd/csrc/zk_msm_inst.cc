// One translation unit per curve: hipcc -DZK_CURVE=<Pallas|Vesta|Bn254G1|Bls381G1>
#include "zk_msm.inl"
namespace zk {
template int msm_run<ZK_CURVE>(const BasesEntry&, const Fe<ZK_CURVE::Fr>*, uint64_t, int, const zk_msm_opts*, void*, hipStream_t);
template int bases_prepare_run<ZK_CURVE>(BasesEntry&);
template int fixed_base_run<ZK_CURVE>(const Fe<ZK_CURVE::Fr>*, uint64_t, Affine<ZK_CURVE>*, hipStream_t);
template int fixed_base_msm_run<ZK_CURVE>(const Affine<ZK_CURVE>&, const Fe<ZK_CURVE::Fr>*, uint64_t, int, Affine<ZK_CURVE>*, hipStream_t);
}  // namespace zk
