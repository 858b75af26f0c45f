// One translation unit per scalar field: hipcc -DZK_FIELD=<PallasFp|PallasFq|Bn254Fr|Bls381Fr>
#include "zk_ntt.inl"
namespace zk {
template int ntt_run<ZK_FIELD>(DeviceCtx&, int, Fe<ZK_FIELD>*, uint32_t, const Fe<ZK_FIELD>&, int, hipStream_t, const Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, uint32_t);
template int coset_run<ZK_FIELD>(DeviceCtx&, int, Fe<ZK_FIELD>*, uint32_t, const Fe<ZK_FIELD>&, hipStream_t);
template int vec_op_run<ZK_FIELD>(Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, const Fe<ZK_FIELD>*, uint64_t, int, const Fe<ZK_FIELD>&, hipStream_t);
template int scale_periodic_run<ZK_FIELD>(Fe<ZK_FIELD>*, uint64_t, const Fe<ZK_FIELD>*, uint32_t, hipStream_t);
template int witness_map_run<ZK_FIELD>(DeviceCtx&, int, Fe<ZK_FIELD>*, Fe<ZK_FIELD>*, Fe<ZK_FIELD>*, uint32_t, hipStream_t);
}  // namespace zk
