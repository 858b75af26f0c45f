// R1CS side of the Groth16 witness map (ark-groth16 0.3 r1cs_to_qap.rs): the sparse products
// <A_i, z>, <B_i, z>, <C_i, z> over Fr as CSR mat-vecs on the device, so that the full assignment is the only thing that
// crosses PCIe per proof (3 x m x 32 B of evaluations otherwise: 3 x 128 MiB at m = 2^22).
//   upstream: evaluate_constraint(terms, assignment) = sum coeff * assignment[index]      (one rayon task per row)
//   here:     r1cs_matvec_kernel       one lane per row (rows of a circuit are short: a few terms)
//             r1cs_matvec_long_kernel  one workgroup per row longer than R1CS_LONG_ROW terms (linear combinations over
//                                      a whole vector -- the reference's public-input packing, circuits-ark/src/encryption.rs:139-152)
// Coefficients of real circuits are mostly +-1: those terms cost an addition, not a Montgomery product.
#pragma once
#include "zk_rt.h"
#include "zk_field.h"

namespace zk {


template <class F>
__device__ __forceinline__ void r1cs_term(Fe<F>& acc, const Fe<F>& coeff, const Fe<F>& zv) {
    Fe<F> one, mone;
    fe_one(one);
    fe_neg(mone, one);
    if (fe_eq(coeff, one)) {
        fe_add(acc, acc, zv);
    } else if (fe_eq(coeff, mone)) {
        fe_sub(acc, acc, zv);
    } else {
        Fe<F> t;
        fe_mul(t, coeff, zv);
        fe_add(acc, acc, t);
    }
}

// out[i] = <row i, z> for i < n_rows (rows longer than R1CS_LONG_ROW are left to the long kernel), 0 for n_rows <= i < out_len
template <class F>
__global__ void __launch_bounds__(256) r1cs_matvec_kernel(const uint64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                                          const Fe<F>* __restrict__ val, const Fe<F>* __restrict__ z, Fe<F>* __restrict__ out,
                                                          uint64_t n_rows, uint64_t out_len) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < out_len; i += (uint64_t)gridDim.x * blockDim.x) {
        Fe<F> acc;
        fe_zero(acc);
        if (i < n_rows) {
            const uint64_t k0 = row_ptr[i], k1 = row_ptr[i + 1];
            if (k1 - k0 > R1CS_LONG_ROW) continue;
            for (uint64_t k = k0; k < k1; k++) {
                Fe<F> c = val[k], zv = z[col[k]];
                r1cs_term(acc, c, zv);
            }
        }
        out[i] = acc;
    }
}

// grid = number of long rows; one workgroup of 256 lanes strides over the row's terms and tree-sums in LDS
template <class F>
__global__ void __launch_bounds__(256) r1cs_matvec_long_kernel(const uint64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                                               const Fe<F>* __restrict__ val, const Fe<F>* __restrict__ z, Fe<F>* __restrict__ out,
                                                               const uint64_t* __restrict__ long_rows) {
    __shared__ Fe<F> part[256];
    const uint64_t i = long_rows[blockIdx.x];
    const uint64_t k0 = row_ptr[i], k1 = row_ptr[i + 1];
    Fe<F> acc;
    fe_zero(acc);
    for (uint64_t k = k0 + threadIdx.x; k < k1; k += blockDim.x) {
        Fe<F> c = val[k], zv = z[col[k]];
        r1cs_term(acc, c, zv);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) {
            Fe<F> o = part[threadIdx.x + d];
            fe_add(acc, acc, o);
            part[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[i] = acc;
}

}  // namespace zk
