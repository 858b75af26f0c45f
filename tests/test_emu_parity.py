"""CPU tier: kernel-logic parity.  The HIP sources of contangle-zkcp_amd/csrc are compiled with
g++ against tests/emu (a HIP-semantics emulator, TEST INFRASTRUCTURE) and driven through the
same C ABI and Python mirror as on the GPU; results are compared with the oracle and the
pure-Python fixtures.  The real parity gate is tests/test_gpu_parity.py (-m gpu)."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import parity_suite as ps  # noqa: E402


@pytest.fixture(scope="module")
def zk():
    spec = importlib.util.spec_from_file_location("zk_build", os.path.join(ROOT, "contangle-zkcp_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    emu = b.build_emu()
    import contangle_zkcp_amd as zk
    zk.load(path=emu)
    zk.init(0)
    assert zk.backend_info().startswith("emu")
    yield zk
    zk.shutdown()
    zk._lib = None


@pytest.fixture
def ntt_plan(zk):
    """zk_ntt_configure for one test; the defaults come back afterwards"""
    yield zk.ntt_configure
    zk.ntt_configure()


def test_ntt_golden(zk):
    ps.check_ntt_golden(zk)


@pytest.mark.parametrize("max_logr,logt", [("10", "2"), ("3", "1"), ("2", "2"), ("4", "0")])
def test_ntt_multipass(zk, ntt_plan, max_logr, logt):
    # small radices force 2-, 3- and 4-pass plans at sizes the emulator handles quickly
    ntt_plan(max_log_radix=int(max_logr), log_tile=int(logt))
    for name, logn in (("PallasFp", 7), ("Bls381Fr", 8), ("PallasFq", 5), ("Bn254Fr", 6), ("PallasFp", 1), ("PallasFp", 2)):
        if (logn + int(max_logr) - 1) // int(max_logr) > 4:
            continue
        ps.check_ntt_vs_oracle(zk, name, logn)


def test_ntt_extend(zk, ntt_plan):
    ps.check_ntt_extend(zk, "PallasFp", 5, 8)        # one pass
    ps.check_ntt_extend(zk, "Bls381Fr", 9, 12)       # two passes
    ps.check_ntt_extend(zk, "PallasFp", 0, 6)        # a single coefficient
    ps.check_ntt_extend(zk, "PallasFq", 7, 7)        # nothing to extend
    ntt_plan(max_log_radix=3)
    ps.check_ntt_extend(zk, "Bn254Fr", 4, 8)         # three passes


def test_ntt_two_pass_default_plan(zk):
    ps.check_ntt_vs_oracle(zk, "PallasFp", 12)


@pytest.mark.parametrize("block,logt", [("64", "2"), ("64", "4"), ("128", "3")])
def test_ntt_butterflies_per_lane(zk, ntt_plan, block, logt):
    # 1, 2 and 8 butterflies per lane and stage
    ntt_plan(log_tile=int(logt), block=int(block))
    ps.check_ntt_vs_oracle(zk, "PallasFp", 12)
    ps.check_ntt_vs_oracle(zk, "Bls381Fr", 11)


def test_msm_golden(zk):
    ps.check_msm_golden(zk)


@pytest.mark.parametrize("cname", ps.CURVES)
def test_msm_edges(zk, cname):
    ps.check_msm_edges(zk, cname)


@pytest.mark.parametrize("cname,n,wb,realistic", [
    ("Pallas", 300, 0, False), ("Vesta", 257, 6, True), ("Bn254G1", 200, 5, False), ("Bls381G1", 150, 7, True),
    ("Pallas", 1024, 8, True), ("Bls381G2", 120, 5, False), ("Bn254G2", 100, 6, True),
    ("Vesta", 3000, 16, True)])      # 64 ranges per window: the unit scalars make one region 20x the mean (wave-aggregated path)
def test_msm_vs_oracle(zk, cname, n, wb, realistic):
    ps.check_msm_vs_oracle(zk, cname, n, wb, realistic)


def test_msm_window_sharding(zk):
    ps.check_msm_window_sharding(zk, "Vesta", 128, 6, 4)
    ps.check_msm_window_sharding(zk, "Bls381G1", 64, 5, 8)


def test_msm_big_buckets(zk):
    ps.check_msm_big_buckets(zk, "Vesta")
    ps.check_msm_big_buckets(zk, "Bls381G1", n=2100, window_bits=4)


def test_vec_ops_and_witness_map(zk):
    ps.check_vec_ops(zk, "Bls381Fr", 300)
    ps.check_vec_ops(zk, "PallasFp", 100)
    ps.check_witness_map(zk, "Bls381Fr", 6)
    ps.check_witness_map(zk, "Bn254Fr", 4)


def test_msm_slice_lengths(zk):
    ps.check_msm_slice_lengths(zk, "Vesta", 300, 8)
    ps.check_msm_slice_lengths(zk, "Bn254G2", 60, 6)


def test_msm_precomputed_table(zk):
    ps.check_msm_precomputed(zk, "Vesta", 512, 8, count=3)
    ps.check_msm_precomputed(zk, "Bls381G1", 256, 6, realistic=True)
    ps.check_msm_precomputed(zk, "Bn254G2", 256, 5)
    ps.check_msm_precomputed(zk, "Pallas", 256, 17, count=2)    # wider than a u16 digit code: the 32-bit codes of this form


def test_msm_window_groups(zk):
    ps.check_msm_window_groups(zk, "Vesta", 500, 6)
    ps.check_msm_window_groups(zk, "Bls381G2", 60, 5, groups=(2, 7))


def test_msm_axis_reduce(zk):
    ps.check_msm_axis_reduce(zk, "Vesta", 700, [2, 3, 4, 5, 8, 11])
    ps.check_msm_axis_reduce(zk, "Bn254G2", 300, [3, 6, 9])
    ps.check_msm_axis_reduce(zk, "Bls381G2", 200, [16], windows=(3, 5))     # 256 columns > 128 lanes: two blocks
    ps.check_msm_axis_reduce(zk, "Pallas", 300, [16, 15], windows=(0, 2))


def test_msm_device_side_partial_conversion(zk):
    ps.check_msm_device_partials(zk, "Bn254G1", n=120, window_bits_list=(0, 5))
    ps.check_msm_device_partials(zk, "Bls381G2", n=40, window_bits_list=(4,))
    ps.check_msm_device_partials(zk, "Vesta", n=90, window_bits_list=(3,))


def test_fixed_base_msm(zk):
    ps.check_fixed_base_msm(zk, "Vesta", 21)
    ps.check_fixed_base_msm(zk, "Bn254G2", 13)


def test_msm_sort_shapes(zk):
    ps.check_msm_sort_shapes(zk, "Vesta", 2100, (5, 13))        # 3 scalar blocks; 1 and 2 ranges per window
    ps.check_msm_sort_shapes(zk, "Bls381G1", 90, (7,))


def test_ntt_fused_coset(zk, ntt_plan):
    for name, logn in (("Bls381Fr", 6), ("PallasFp", 11), ("Bn254Fr", 1), ("PallasFq", 12)):
        ps.check_ntt_fused_coset(zk, name, logn)
    ntt_plan(max_log_radix=3)   # three passes: pre on the first, post on the last
    ps.check_ntt_fused_coset(zk, "Bls381Fr", 8)


def test_msm_bucket_splitting(zk):
    ps.check_msm_split(zk, "Vesta", 400, 5)
    ps.check_msm_split(zk, "Bls381G1", 150, 4, realistic=False)
    ps.check_msm_split(zk, "Pallas", 3000, 6)        # oversized buckets + splitting
    ps.check_msm_split(zk, "Bn254G2", 60, 3)
    ps.check_msm_split(zk, "Vesta", 900, 8)          # 128 buckets per range: the size-rank zones exist
    ps.check_msm_split(zk, "Bn254G1", 2500, 11)      # two ranges of 512 buckets per window


def test_msm_deferred_results(zk):
    ps.check_msm_async(zk, "Vesta", 300, 6)
    ps.check_msm_async(zk, "Bls381G2", 40, 4)


def test_msm_batch(zk):
    ps.check_msm_batch(zk, "Pallas", 260, 5, 6)
    ps.check_msm_batch(zk, "Bn254G1", 100, 2)


def test_halo2_domain(zk):
    ps.check_halo2_domain(zk, "PallasFp", 5)          # degree-9 gates: extended_k = k + 3
    ps.check_halo2_domain(zk, "PallasFq", 4, j=5)     # extended_k = k + 2
    ps.check_halo2_domain(zk, "PallasFp", 3, j=2)     # nothing to extend


def test_quotient_by_parts(zk):
    ps.check_quotient_by_parts(zk, "PallasFp", "Vesta", 3, 4, direct_pieces=2)      # two pieces per sub-coset (the GPU tier: 8, 4, 2, 1 parts)


def test_groth16_prove_end_to_end(zk):
    ps.check_groth16_prove(zk, "Bls381", num_constraints=26, long_rows=(20,))      # domain 32; one row of > 20 terms


def test_halo2_products(zk):
    ps.check_batch_invert_and_scan(zk, "PallasFp", 5000)       # two scan workgroups, a ragged tail
    ps.check_batch_invert_and_scan(zk, "Bls381Fr", 37)
    ps.check_batch_invert_and_scan(zk, "PallasFq", 1)
    ps.check_permutation_and_lookup_products(zk, "PallasFp", 6)


def test_halo2_permute_expression_pair(zk):
    ps.check_permute_expression_pair(zk, "PallasFp")
    ps.check_permute_expression_pair(zk, "Bls381Fr", n=90, usable=83)


def test_halo2_ipa_collapse_edges(zk):
    ps.check_ipa_collapse_edges(zk, "Vesta", 5)


def test_halo2_eval_polynomial(zk):
    ps.check_eval_polynomial(zk, "PallasFp")
    ps.check_eval_polynomial(zk, "Bls381Fr", sizes=(3, 300))


def test_halo2_kate_division(zk):
    ps.check_kate_division(zk, "PallasFp")
    ps.check_kate_division(zk, "Bls381Fr", sizes=(3, 4100))
    ps.check_kate_division_at_size(zk, "PallasFq", 14)


def test_halo2_ipa(zk):
    ps.check_ipa(zk, "Vesta", 4)
    ps.check_ipa(zk, "Pallas", 2)
    ps.check_ipa(zk, "Bn254G1", 2)


def test_halo2_expression(zk):
    ps.check_expression(zk, "PallasFp", 4)


def test_halo2_work_list_shapes_small(zk):
    """the at-size checks of the GPU tier (bench program on sampled rows; IPA in the exponent with a collapse), at emulator sizes"""
    assert ps.check_expression_at_size(zk, "PallasFp", 4, 3, samples=20) >= 20
    assert ps.check_expression_at_size(zk, "PallasFp", 5, 3, samples=20, parts=4) >= 20      # one sub-coset of a 4-way sharded quotient
    ps.check_ipa_at_size(zk, "Vesta", 6, 2, survivors=4)
    ps.check_ipa_at_size(zk, "Pallas", 4, 3, survivors=2)


def test_ntt_saturated_limbs_path(zk, ntt_plan):
    """the 32-bit-word butterflies stay reachable through zk_ntt_opts (A/B measurements) and agree with the oracle too"""
    ntt_plan(limb_bits=32)
    ps.check_ntt_vs_oracle(zk, "PallasFp", 11)
    ps.check_ntt_fused_coset(zk, "Bls381Fr", 6)
    ps.check_ntt_extend(zk, "Bn254Fr", 5, 8)
    ntt_plan(limb_bits=32, max_log_radix=3)
    ps.check_ntt_vs_oracle(zk, "PallasFq", 8)
