//! Raw binding of `include/zkcp_amd.h` + `include/zkcp_amd_prover.h`, and the helpers shared by the ark-ec / ark-poly /
//! ark-groth16 / halo2_proofs shims (rust/patches/).  NOT COMPILED in the build image (no Rust toolchain there); kept in
//! step with the headers by tests/test_rust_shims.py.
#![allow(non_camel_case_types)]
use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};
use std::sync::{Mutex, Once};

// ---- zk_curve_t / zk_field_t / zk_pairing_t / zk_status
pub const ZK_PALLAS: c_int = 0;
pub const ZK_VESTA: c_int = 1;
pub const ZK_BN254_G1: c_int = 2;
pub const ZK_BLS12_381_G1: c_int = 3;
pub const ZK_BN254_G2: c_int = 4;
pub const ZK_BLS12_381_G2: c_int = 5;
pub const ZK_FP_PALLAS: c_int = 0;
pub const ZK_FQ_PALLAS: c_int = 1;
pub const ZK_FR_BN254: c_int = 2;
pub const ZK_FR_BLS12_381: c_int = 3;
pub const ZK_PAIRING_BN254: c_int = 0;
pub const ZK_PAIRING_BLS12_381: c_int = 1;
pub const ZK_OK: c_int = 0;
pub const ZK_ERR_BUSY: c_int = -8;

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct zk_msm_opts {
    pub window_bits: c_int,
    pub window_begin: c_int,
    pub window_end: c_int,
    pub limb_bits: c_int,
    pub split_log_plus1: c_int,
    pub slice_len: c_int,
    pub big_threshold: c_int,
    pub waves_per_simd: c_int,
    pub flags: c_int,
    pub base_offset: c_int,
    pub window_group: c_int,
    pub reserved: c_int,
}
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct zk_ntt_opts {
    pub max_log_radix: c_int,
    pub log_tile_plus1: c_int,
    pub block: c_int,
    pub limb_bits: c_int,
}
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct zk_msm_profile {
    pub digits_ms: f32,
    pub hist_ms: f32,
    pub scatter_ms: f32,
    pub accumulate_ms: f32,
    pub reduce_ms: f32,
    pub host_tail_ms: f32,
    pub total_ms: f32,
    pub window_bits: c_int,
    pub windows_total: c_int,
    pub windows_done: c_int,
    pub groups: c_int,
    pub limb_bits: c_int,
    pub accumulate_kernel_ms: f32,
    pub reserved: c_int,
}
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct zk_msm_totals {
    pub msms: u64,
    pub accumulate_kernel_ms: f64,
    pub accumulate_ms: f64,
    pub sort_ms: f64,
    pub reduce_ms: f64,
    pub host_tail_ms: f64,
    pub device_ms: f64,
    pub algorithmic_bytes: f64,
    pub launches: u64,
}
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct zk_ntt_totals {
    pub transforms: u64,
    pub launches: u64,
    pub kernel_ms: f64,
    pub algorithmic_bytes: f64,
}
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct zk_ark_span {
    pub offset: u64,
    pub count: u64,
}
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct zk_ark_pk_index {
    pub alpha_g1: zk_ark_span,
    pub beta_g2: zk_ark_span,
    pub gamma_g2: zk_ark_span,
    pub delta_g2: zk_ark_span,
    pub gamma_abc_g1: zk_ark_span,
    pub beta_g1: zk_ark_span,
    pub delta_g1: zk_ark_span,
    pub a_query: zk_ark_span,
    pub b_g1_query: zk_ark_span,
    pub b_g2_query: zk_ark_span,
    pub h_query: zk_ark_span,
    pub l_query: zk_ark_span,
    pub total_bytes: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zk_groth16_assembly {
    pub alpha_g1: *const c_void,
    pub beta_g1: *const c_void,
    pub delta_g1: *const c_void,
    pub beta_g2: *const c_void,
    pub delta_g2: *const c_void,
    pub a_query0: *const c_void,
    pub b_g1_query0: *const c_void,
    pub b_g2_query0: *const c_void,
    pub a_acc: *const c_void,
    pub b_g1_acc: *const c_void,
    pub l_acc: *const c_void,
    pub h_acc: *const c_void,
    pub b_g2_acc: *const c_void,
    pub r: *const c_void,
    pub s: *const c_void,
}
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct zk_expr_op {
    pub op: u8,
    pub pad: u8,
    pub rot: i16,
    pub arg: u32,
}

extern "C" {
    // ---- include/zkcp_amd.h
    pub fn zk_init(device_id: c_int) -> c_int;
    pub fn zk_init_devices(n_devices: c_int, device_ids: *const c_int) -> c_int;
    pub fn zk_device_count() -> c_int;
    pub fn zk_shutdown() -> c_int;
    pub fn zk_strerror(status: c_int) -> *const c_char;
    pub fn zk_backend_info(buf: *mut c_char, buflen: u64) -> c_int;
    pub fn zk_field_limbs64(f: c_int) -> c_int;
    pub fn zk_curve_base_limbs64(c: c_int) -> c_int;
    pub fn zk_curve_scalar_field(c: c_int) -> c_int;
    pub fn zk_msm_window_bits(c: c_int, n: u64, requested: c_int) -> c_int;
    pub fn zk_msm_window_count(c: c_int, n: u64, window_bits: c_int) -> c_int;
    pub fn zk_bases_upload(c: c_int, affine_xy_mont_host: *const c_void, n: u64, handle_out: *mut u64) -> c_int;
    pub fn zk_bases_adopt_device(c: c_int, affine_xy_mont_dev: *const c_void, n: u64, handle_out: *mut u64) -> c_int;
    pub fn zk_bases_free(handle: u64) -> c_int;
    pub fn zk_bases_precompute(handle: u64, window_bits: c_int) -> c_int;
    pub fn zk_bases_refresh(handle: u64, offset: u64, count: u64, hip_stream: *mut c_void) -> c_int;
    pub fn zk_msm(c: c_int, bases_handle: u64, scalars_host: *const c_void, n: u64, scalars_are_montgomery: c_int,
                  opts: *const zk_msm_opts, out_jacobian_host: *mut c_void) -> c_int;
    pub fn zk_msm_device(c: c_int, bases_handle: u64, scalars_dev: *const c_void, n: u64, scalars_are_montgomery: c_int,
                         opts: *const zk_msm_opts, out_jacobian_host: *mut c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_msm_last_profile(out: *mut zk_msm_profile) -> c_int;
    pub fn zk_msm_profile_totals(out: *mut zk_msm_totals, reset: c_int) -> c_int;
    pub fn zk_msm_submit(c: c_int, bases_handle: u64, scalars_dev: *const c_void, n: u64, scalars_are_montgomery: c_int,
                         opts: *const zk_msm_opts, hip_stream: *mut c_void, ticket_out: *mut u64) -> c_int;
    pub fn zk_msm_collect(ticket: u64, out_jacobian_host: *mut c_void) -> c_int;
    pub fn zk_msm_batch_device(c: c_int, bases_handle: u64, scalars_dev: *const c_void, n: u64, count: u32, stride_elems: u64,
                               scalars_are_montgomery: c_int, opts: *const zk_msm_opts, out_jacobian_host: *mut c_void,
                               hip_stream: *mut c_void) -> c_int;
    pub fn zk_ntt(f: c_int, a_mont_host: *mut c_void, log_n: u32, omega_mont_host: *const c_void, scale_by_n_inv: c_int) -> c_int;
    pub fn zk_ntt_device(f: c_int, a_mont_dev: *mut c_void, log_n: u32, omega_mont_host: *const c_void, scale_by_n_inv: c_int,
                         hip_stream: *mut c_void) -> c_int;
    pub fn zk_ntt_configure(opts: *const zk_ntt_opts) -> c_int;
    pub fn zk_ntt_profile_enable(on: c_int) -> c_int;
    pub fn zk_ntt_profile_read(out: *mut zk_ntt_totals) -> c_int;
    pub fn zk_ntt_coset_device(f: c_int, a_mont_dev: *mut c_void, log_n: u32, omega_mont_host: *const c_void, scale_by_n_inv: c_int,
                               g_pre_mont_host: *const c_void, g_post_mont_host: *const c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_ntt_extend_device(f: c_int, a_mont_dev: *mut c_void, log_n: u32, log_in: u32, omega_mont_host: *const c_void,
                                scale_by_n_inv: c_int, g_pre_mont_host: *const c_void, g_post_mont_host: *const c_void,
                                hip_stream: *mut c_void) -> c_int;
    pub fn zk_ntt_oop_device(f: c_int, src_mont_dev: *const c_void, dst_mont_dev: *mut c_void, log_n: u32, log_in: u32,
                             omega_mont_host: *const c_void, scale_by_n_inv: c_int, g_pre_mont_host: *const c_void,
                             g_post_mont_host: *const c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_coset_mul(f: c_int, a_mont_host: *mut c_void, log_n: u32, g_mont_host: *const c_void) -> c_int;
    pub fn zk_coset_mul_device(f: c_int, a_mont_dev: *mut c_void, log_n: u32, g_mont_host: *const c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_vec_op_device(f: c_int, op: c_int, a_dev: *mut c_void, b_dev: *const c_void, c_dev: *const c_void, n: u64,
                            scalar_mont_host: *const c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_groth16_witness_map_device(f: c_int, a_dev: *mut c_void, b_dev: *mut c_void, c_dev: *mut c_void, log_m: u32,
                                         hip_stream: *mut c_void) -> c_int;
    pub fn zk_vec_scale_periodic_device(f: c_int, a_dev: *mut c_void, n: u64, table_mont_host: *const c_void, m: u32,
                                        hip_stream: *mut c_void) -> c_int;
    pub fn zk_field_modulus(f: c_int, p_canonical_out: *mut c_void) -> c_int;
    pub fn zk_field_root_of_unity(f: c_int, log_n: u32, omega_mont_out: *mut c_void) -> c_int;
    pub fn zk_field_multiplicative_generator(f: c_int, g_mont_out: *mut c_void) -> c_int;
    pub fn zk_field_inverse(f: c_int, a_mont: *const c_void, out_mont: *mut c_void) -> c_int;
    pub fn zk_point_add(c: c_int, jac_a: *const c_void, jac_b: *const c_void, jac_out: *mut c_void) -> c_int;
    pub fn zk_point_to_affine(c: c_int, jac: *const c_void, affine_out: *mut c_void) -> c_int;
    pub fn zk_fixed_base_mul_device(c: c_int, scalars_canonical_dev: *const c_void, n: u64, affine_out_dev: *mut c_void,
                                    hip_stream: *mut c_void) -> c_int;
    pub fn zk_fixed_base_msm_device(c: c_int, base_affine_mont: *const c_void, scalars_dev: *const c_void, n: u64,
                                    scalars_are_montgomery: c_int, affine_out_dev: *mut c_void, hip_stream: *mut c_void) -> c_int;
    // ---- include/zkcp_amd_prover.h
    pub fn zk_ark_point_size(c: c_int, compressed: c_int) -> c_int;
    pub fn zk_ark_points_encode(c: c_int, affine_mont: *const c_void, n: u64, compressed: c_int, out: *mut u8) -> c_int;
    pub fn zk_ark_points_decode(c: c_int, input: *const u8, n: u64, compressed: c_int, check_on_curve: c_int,
                                affine_mont_out: *mut c_void) -> c_int;
    pub fn zk_ark_scalars_encode(f: c_int, mont: *const c_void, n: u64, out: *mut u8) -> c_int;
    pub fn zk_ark_scalars_decode(f: c_int, input: *const u8, n: u64, mont_out: *mut c_void) -> c_int;
    pub fn zk_ark_proving_key_index(p: c_int, buf: *const u8, len: u64, out: *mut zk_ark_pk_index) -> c_int;
    pub fn zk_bases_upload_ark(c: c_int, uncompressed_points: *const u8, n: u64, handle_out: *mut u64) -> c_int;
    pub fn zk_ark_proof_size(p: c_int) -> c_int;
    pub fn zk_ark_proof_encode(p: c_int, a_g1_affine_mont: *const c_void, b_g2_affine_mont: *const c_void,
                               c_g1_affine_mont: *const c_void, out: *mut u8) -> c_int;
    pub fn zk_ark_proof_decode(p: c_int, input: *const u8, a_g1_affine_mont: *mut c_void, b_g2_affine_mont: *mut c_void,
                               c_g1_affine_mont: *mut c_void) -> c_int;
    pub fn zk_r1cs_matrix_upload(f: c_int, row_ptr_host: *const u64, col_idx_host: *const u32, val_mont_host: *const c_void, n_rows: u64,
                                 n_cols: u64, handle_out: *mut u64) -> c_int;
    pub fn zk_r1cs_matrix_free(handle: u64) -> c_int;
    pub fn zk_r1cs_matvec_device(matrix: u64, z_mont_dev: *const c_void, out_mont_dev: *mut c_void, out_len: u64, hip_stream: *mut c_void) -> c_int;
    pub fn zk_groth16_witness_map_r1cs_device(f: c_int, matrix_a: u64, matrix_b: u64, matrix_c: u64, z_mont_dev: *const c_void,
                                              num_inputs: u64, log_m: u32, a_dev: *mut c_void, b_dev: *mut c_void, c_dev: *mut c_void,
                                              hip_stream: *mut c_void) -> c_int;
    pub fn zk_groth16_assemble_proof(p: c_int, input: *const zk_groth16_assembly, a_g1_affine_out: *mut c_void, b_g2_affine_out: *mut c_void,
                                     c_g1_affine_out: *mut c_void) -> c_int;
    pub fn zk_batch_invert_device(f: c_int, a_dev: *mut c_void, n: u64, hip_stream: *mut c_void) -> c_int;
    pub fn zk_prefix_product_device(f: c_int, in_dev: *const c_void, out_dev: *mut c_void, n: u64, first_mont_host: *const c_void,
                                    total_out_mont_host: *mut c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_halo2_permutation_product_device(f: c_int, ncols: u32, columns_dev: *const *const c_void, sigmas_dev: *const *const c_void,
                                               first_column_index: u32, beta: *const c_void, gamma: *const c_void, delta: *const c_void, k: u32,
                                               z_first: *const c_void, z_out_dev: *mut c_void, z_last_out_host: *mut c_void,
                                               hip_stream: *mut c_void) -> c_int;
    pub fn zk_halo2_lookup_product_device(f: c_int, a_dev: *const c_void, s_dev: *const c_void, a_perm_dev: *const c_void,
                                          s_perm_dev: *const c_void, beta: *const c_void, gamma: *const c_void, n: u64, z_out_dev: *mut c_void,
                                          z_last_out_host: *mut c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_inner_product_device(f: c_int, a_dev: *const c_void, b_dev: *const c_void, n: u64, out_mont_host: *mut c_void,
                                   hip_stream: *mut c_void) -> c_int;
    pub fn zk_poly_eval_device(f: c_int, coeffs_dev: *const c_void, n: u64, x_mont_host: *const c_void, out_mont_host: *mut c_void,
                               hip_stream: *mut c_void) -> c_int;
    pub fn zk_vec_muladd_device(f: c_int, a_dev: *mut c_void, b_dev: *const c_void, n: u64, s_mont_host: *const c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_vec_muladd_to_device(f: c_int, out_dev: *mut c_void, a_dev: *const c_void, b_dev: *const c_void, n: u64, s_mont_host: *const c_void,
                                   hip_stream: *mut c_void) -> c_int;
    pub fn zk_vec_fold_many_device(f: c_int, out_dev: *mut c_void, first_dev: *const c_void, stride_elems: i64, count: u32, n: u64,
                                   s_mont_host: *const c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_ipa_fold_round_device(f: c_int, p_dev: *mut c_void, b_dev: *mut c_void, w_dev: *mut c_void, half: u64, m0: u64,
                                    u_mont_host: *const c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_vec_powers_device(f: c_int, out_dev: *mut c_void, n: u64, x_mont_host: *const c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_kate_division_device(f: c_int, a_dev: *const c_void, q_dev: *mut c_void, n: u64, x_mont_host: *const c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_poly_eval_batch_device(f: c_int, coeffs_dev: *const c_void, n: u64, count: u32, stride_elems: u64, x_mont_host: *const c_void,
                                     out_mont_host: *mut c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_vec_fold_device(f: c_int, a_dev: *mut c_void, half: u64, c_mont_host: *const c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_ipa_fold_bases_device(c: c_int, g_affine_dev: *mut c_void, half: u64, u_mont_host: *const c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_ipa_virtual_scalars_device(f: c_int, p_dev: *const c_void, w_dev: *const c_void, m0: u64, cur: u64, sl_dev: *mut c_void,
                                         sr_dev: *mut c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_ipa_update_weights_device(f: c_int, w_dev: *mut c_void, m0: u64, bit: u64, u_mont_host: *const c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_ipa_collapse_range_device(c: c_int, bases_handle: u64, w_dev: *const c_void, m0: u64, cur: u64, first: u64, count: u64,
                                        g_out_range_dev: *mut c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_ipa_round_device(c: c_int, bases_handle: u64, p_dev: *const c_void, b_dev: *const c_void, w_dev: *const c_void, m0: u64, cur: u64,
                               s_dev: *mut c_void, lr_out_host: *mut c_void, v_out_mont_host: *mut c_void, hip_stream: *mut c_void) -> c_int;
    pub fn zk_ipa_collapse_device(c: c_int, bases_handle: u64, w_dev: *const c_void, m0: u64, cur: u64, g_out_affine_dev: *mut c_void,
                                  hip_stream: *mut c_void) -> c_int;
    pub fn zk_expr_eval_device(f: c_int, program_host: *const zk_expr_op, n_ops: u32, columns_dev: *const *const c_void, n_columns: u32,
                               consts_mont_host: *const c_void, n_consts: u32, log_n_ext: u32, rot_scale: u32, out_dev: *mut c_void,
                               hip_stream: *mut c_void) -> c_int;
    pub fn zk_expr_eval_lazy_device(f: c_int, program_host: *const zk_expr_op, n_ops: u32, columns_dev: *const *const c_void, n_columns: u32,
                               consts_mont_host: *const c_void, n_consts: u32, log_n_ext: u32, rot_scale: u32, out_dev: *mut c_void,
                               hip_stream: *mut c_void) -> c_int;
    /// 0 = the quotient kernel specialised per program (hiprtc) for evaluations of 2^16 rows and more, 1 = always, 2 = never
    pub fn zk_expr_configure(jit_mode: c_int) -> c_int;
    pub fn zk_expr_specialised_source(f: c_int, program_host: *const zk_expr_op, n_ops: u32, n_columns: u32, n_consts: u32, out: *mut c_char, cap: u64,
                                      len_out: *mut u64) -> c_int;
}

// =====================================================================================================================
// helpers shared by the shims
// =====================================================================================================================
#[derive(Debug)]
pub struct ZkError(pub c_int, pub String);
pub fn check(status: c_int, what: &str) -> Result<(), ZkError> {
    if status == ZK_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(zk_strerror(status)) }.to_string_lossy().into_owned();
    Err(ZkError(status, format!("{}: {}", what, msg)))
}

static INIT: Once = Once::new();
/// One process drives every GPU named in ZKCP_AMD_DEVICES ("0,1,2,3"; default "0"): whole MSMs are then split over them by
/// scalar window inside the library (include/zkcp_amd.h, zk_init_devices).
pub fn init_once() {
    INIT.call_once(|| {
        let ids: Vec<c_int> = std::env::var("ZKCP_AMD_DEVICES").unwrap_or_else(|_| "0".into())
            .split(',').filter_map(|s| s.trim().parse().ok()).collect();
        let st = unsafe { zk_init_devices(ids.len() as c_int, ids.as_ptr()) };
        check(st, "zk_init_devices").expect("libzkcp_amd needs an MI355X: there is no CPU fallback");
    });
}

/// Which library curve a short-Weierstrass ark type is, decided from the base-field modulus (low limb) and the extension
/// degree of the coordinate field -- the forks are generic over `G: AffineCurve` and cannot name ark-bls12-381 types
/// (that crate depends on ark-ec, not the other way round).
pub fn curve_id(base_modulus_limb0: u64, base_limbs: usize, ext_degree: usize) -> Option<c_int> {
    match (base_modulus_limb0, base_limbs, ext_degree) {
        (0xb9feffffffffaaab, 6, 1) => Some(ZK_BLS12_381_G1),
        (0xb9feffffffffaaab, 6, 2) => Some(ZK_BLS12_381_G2),
        (0x3c208c16d87cfd47, 4, 1) => Some(ZK_BN254_G1),
        (0x3c208c16d87cfd47, 4, 2) => Some(ZK_BN254_G2),
        _ => None, // BLS12-377, BW6-761, JubJub ...: the upstream CPU body stays
    }
}
pub fn field_id(modulus_limb0: u64, modulus_limb3: u64) -> Option<c_int> {
    match (modulus_limb0, modulus_limb3) {
        (0xffffffff00000001, 0x73eda753299d7d48) => Some(ZK_FR_BLS12_381),
        (0x43e1f593f0000001, 0x30644e72e131a029) => Some(ZK_FR_BN254),
        (0x992d30ed00000001, 0x4000000000000000) => Some(ZK_FP_PALLAS),
        (0x8c46eb2100000001, 0x4000000000000000) => Some(ZK_FQ_PALLAS),
        _ => None,
    }
}

/// SRS residency: a query vector is uploaded the first time an MSM sees it and found again by
/// (address, length, content probe).  The probe -- the serialized first, middle and last point -- guards against an
/// allocator handing the same address to a different vector of the same length.
pub struct SrsCache {
    map: Mutex<HashMap<(usize, usize, c_int), (Vec<u8>, u64)>>,
}
lazy_static::lazy_static! {
    pub static ref SRS: SrsCache = SrsCache { map: Mutex::new(HashMap::new()) };
}
impl SrsCache {
    /// `serialize_uncompressed(i, out)` appends point i in ark-serialize's uncompressed form (what `ProvingKey::
    /// serialize_unchecked` writes): the upload goes through `zk_bases_upload_ark`, so no assumption is made about the
    /// in-memory layout of `GroupAffine` (it is not `repr(C)`).
    pub fn get_or_upload(&self, curve: c_int, addr: usize, n: usize, serialize_uncompressed: &dyn Fn(usize, &mut Vec<u8>)) -> u64 {
        let mut probe = Vec::new();
        if n > 0 {
            for i in [0, n / 2, n - 1] {
                serialize_uncompressed(i, &mut probe);
            }
        }
        let mut map = self.map.lock().unwrap();
        if let Some((p, h)) = map.get(&(addr, n, curve)) {
            if *p == probe {
                return *h;
            }
            unsafe { zk_bases_free(*h) };
        }
        let ps = unsafe { zk_ark_point_size(curve, 0) } as usize;
        let mut buf = Vec::with_capacity(n * ps);
        for i in 0..n {
            serialize_uncompressed(i, &mut buf);
        }
        let mut h = 0u64;
        check(unsafe { zk_bases_upload_ark(curve, buf.as_ptr(), n as u64, &mut h) }, "zk_bases_upload_ark").unwrap();
        map.insert((addr, n, curve), (probe, h));
        h
    }
    pub fn clear(&self) {
        for (_, (_, h)) in self.map.lock().unwrap().drain() {
            unsafe { zk_bases_free(h) };
        }
    }
}

/// Scalars as the library reads them: n x 4 little-endian u64.  ark-ff 0.3 `BigInteger256(pub [u64; 4])` and pasta_curves
/// 0.4 `Fp(pub(crate) [u64; 4])` are single-field structs; their layout is checked once against a known value before any
/// slice is reinterpreted (`layout_is_flat`), otherwise the caller copies limb by limb.
pub fn layout_is_flat<T: Copy>(known: &T, limbs: &[u64; 4]) -> bool {
    std::mem::size_of::<T>() == 32 && std::mem::align_of::<T>() >= 8
        && unsafe { std::slice::from_raw_parts(known as *const T as *const u64, 4) } == limbs
}
pub fn flat_or_copy<'a, T: Copy>(xs: &'a [T], flat: bool, limbs_of: &dyn Fn(&T) -> [u64; 4], scratch: &'a mut Vec<u64>) -> &'a [u64] {
    if flat {
        unsafe { std::slice::from_raw_parts(xs.as_ptr() as *const u64, xs.len() * 4) }
    } else {
        scratch.clear();
        for x in xs {
            scratch.extend_from_slice(&limbs_of(x));
        }
        scratch
    }
}

/// MSM result -> uncompressed ark bytes of the affine point (the caller deserializes unchecked and goes projective).
pub fn jacobian_to_ark_uncompressed(curve: c_int, jac: &[u64]) -> Vec<u8> {
    let l = unsafe { zk_curve_base_limbs64(curve) } as usize;
    let mut aff = vec![0u64; 2 * l];
    check(unsafe { zk_point_to_affine(curve, jac.as_ptr() as _, aff.as_mut_ptr() as _) }, "zk_point_to_affine").unwrap();
    let mut out = vec![0u8; unsafe { zk_ark_point_size(curve, 0) } as usize];
    check(unsafe { zk_ark_points_encode(curve, aff.as_ptr() as _, 1, 0, out.as_mut_ptr()) }, "zk_ark_points_encode").unwrap();
    out
}

// =====================================================================================================================
// device buffers for the *_device entry points (the library takes plain device pointers; the forks own the memory)
// =====================================================================================================================
#[link(name = "amdhip64")]
extern "C" {
    fn hipMalloc(ptr: *mut *mut c_void, bytes: usize) -> c_int;
    fn hipFree(ptr: *mut c_void) -> c_int;
    fn hipMemcpy(dst: *mut c_void, src: *const c_void, bytes: usize, kind: c_int) -> c_int;
    fn hipMemset(dst: *mut c_void, value: c_int, bytes: usize) -> c_int;
}
const HIP_MEMCPY_HOST_TO_DEVICE: c_int = 1;
const HIP_MEMCPY_DEVICE_TO_HOST: c_int = 2;

/// `len` u64 words in HBM (a field element is 4 of them, a G1 affine point 8 or 12), freed on drop
pub struct DeviceBuf {
    ptr: *mut c_void,
    len: usize,
}
unsafe impl Send for DeviceBuf {}
impl DeviceBuf {
    pub fn zeroed(len: usize) -> Self {
        let mut ptr = std::ptr::null_mut();
        let bytes = len.max(1) * 8;
        assert_eq!(unsafe { hipMalloc(&mut ptr, bytes) }, 0, "hipMalloc({})", bytes);
        assert_eq!(unsafe { hipMemset(ptr, 0, bytes) }, 0, "hipMemset");
        DeviceBuf { ptr, len }
    }
    pub fn upload(words: &[u64]) -> Self {
        let b = Self::zeroed(words.len());
        assert_eq!(unsafe { hipMemcpy(b.ptr, words.as_ptr() as _, words.len() * 8, HIP_MEMCPY_HOST_TO_DEVICE) }, 0, "hipMemcpy H2D");
        b
    }
    /// `n` copies of one field element (e.g. the Montgomery form of 1: the initial IPA weights)
    pub fn filled(n: usize, elem: &[u64; 4]) -> Self {
        let host: Vec<u64> = (0..n).flat_map(|_| elem.iter().copied()).collect();
        Self::upload(&host)
    }
    pub fn download(&self, first_word: usize, out: &mut [u64]) {
        assert!(first_word + out.len() <= self.len);
        let src = unsafe { (self.ptr as *const u64).add(first_word) };
        assert_eq!(unsafe { hipMemcpy(out.as_mut_ptr() as _, src as _, out.len() * 8, HIP_MEMCPY_DEVICE_TO_HOST) }, 0, "hipMemcpy D2H");
    }
    pub fn ptr(&self) -> *mut c_void { self.ptr }
    pub fn len(&self) -> usize { self.len }
}
impl Drop for DeviceBuf {
    fn drop(&mut self) {
        unsafe { hipFree(self.ptr) };
    }
}
