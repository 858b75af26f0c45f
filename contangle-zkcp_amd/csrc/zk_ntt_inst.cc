// One translation unit per scalar field: hipcc -DZK_FIELD=<PallasFp|PallasFq|Bn254Fr|Bls381Fr>
#include "zk_ntt.inl"
namespace zk {
template int ntt_run<ZK_FIELD>(int, Fe<ZK_FIELD>*, uint32_t, const Fe<ZK_FIELD>&, int, hipStream_t);
template int coset_run<ZK_FIELD>(Fe<ZK_FIELD>*, uint32_t, const Fe<ZK_FIELD>&, hipStream_t);
}  // namespace zk
