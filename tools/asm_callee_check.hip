// Diagnostic: does the generated inline-asm Montgomery product (zk_mul_asm.h) compute the same
// values inside a NON-inlined callee -- 12-limb operands arrive through the stack -- as the portable product, with all
// lanes active and under a divergent EXEC mask?  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude
// -Icontangle-zkcp_amd/csrc tools/asm_callee_check.hip -o tools/asm_callee_check ; prints mismatch counts.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "zk_field.h"
using namespace zk;

template <class P>
__device__ __attribute__((noinline)) Fe<P> mul_call_asm(Fe<P> a, Fe<P> b) {
    Fe<P> r;
    fe_mul(r, a, b);   // device path = fe_mul_asm
    return r;
}
template <class P>
__device__ __attribute__((noinline)) Fe<P> mul_call_portable(Fe<P> a, Fe<P> b) {
    Fe<P> r;
    fe_mul_portable(r, a, b);
    return r;
}
// mode 0: all lanes; 1: odd lanes only; 2: lanes with (i % 3 == 0) inside a data-dependent loop
template <class P>
__global__ void check_kernel(const Fe<P>* a, const Fe<P>* b, Fe<P>* out_asm, Fe<P>* out_ref, Fe<P>* out_inl, int n, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<P> x = a[i], y = b[i];
    Fe<P> r1 = x, r2 = x, r3 = x;
    const bool active = mode == 0 || (mode == 1 && (i & 1)) || (mode == 2 && i % 3 == 0);
    if (active) {
        const int reps = mode == 2 ? 1 + (i % 5) : 3;
        for (int k = 0; k < reps; k++) {
            r1 = mul_call_asm(r1, y);
            r2 = mul_call_portable(r2, y);
            fe_mul(r3, r3, y);
        }
    }
    out_asm[i] = r1;
    out_ref[i] = r2;
    out_inl[i] = r3;
}

template <class P>
int run(const char* name) {
    const int n = 1 << 16;
    std::vector<Fe<P>> a(n), b(n), o1(n), o2(n), o3(n);
    srand(12345);
    for (int i = 0; i < n; i++)
        for (int l = 0; l < P::N; l++) {
            a[i].v[l] = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
            b[i].v[l] = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
        }
    for (int i = 0; i < n; i++) {   // below p: clear the top bits
        a[i].v[P::N - 1] &= P::P[P::N - 1] >> 1;
        b[i].v[P::N - 1] &= P::P[P::N - 1] >> 1;
    }
    Fe<P>*da, *db, *d1, *d2, *d3;
    hipMalloc(&da, n * sizeof(Fe<P>));
    hipMalloc(&db, n * sizeof(Fe<P>));
    hipMalloc(&d1, n * sizeof(Fe<P>));
    hipMalloc(&d2, n * sizeof(Fe<P>));
    hipMalloc(&d3, n * sizeof(Fe<P>));
    hipMemcpy(da, a.data(), n * sizeof(Fe<P>), hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * sizeof(Fe<P>), hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int mode = 0; mode < 3; mode++) {
        hipLaunchKernelGGL(check_kernel<P>, dim3(n / 256), dim3(256), 0, 0, da, db, d1, d2, d3, n, mode);
        hipDeviceSynchronize();
        hipMemcpy(o1.data(), d1, n * sizeof(Fe<P>), hipMemcpyDeviceToHost);
        hipMemcpy(o2.data(), d2, n * sizeof(Fe<P>), hipMemcpyDeviceToHost);
        hipMemcpy(o3.data(), d3, n * sizeof(Fe<P>), hipMemcpyDeviceToHost);
        int bad_callee = 0, bad_inline = 0;
        for (int i = 0; i < n; i++) {
            // host reference of the same chain
            Fe<P> r = a[i];
            const bool active = mode == 0 || (mode == 1 && (i & 1)) || (mode == 2 && i % 3 == 0);
            if (active) {
                const int reps = mode == 2 ? 1 + (i % 5) : 3;
                for (int k = 0; k < reps; k++) fe_mul_portable(r, r, b[i]);
            }
            if (memcmp(&o2[i], &r, sizeof r) != 0) { printf("%s: device portable product differs from host at %d!\n", name, i); return 100000; }
            if (memcmp(&o1[i], &r, sizeof r) != 0) bad_callee++;
            if (memcmp(&o3[i], &r, sizeof r) != 0) bad_inline++;
        }
        printf("%s mode %d: asm-in-callee mismatches %d / %d, asm-inlined mismatches %d / %d\n", name, mode, bad_callee, n, bad_inline, n);
        bad_total += bad_callee + bad_inline;
    }
    return bad_total;
}

int main() {
    int bad = 0;
    bad += run<Bls381Fq>("Bls381Fq (12 limbs, operands through the stack)");
    bad += run<Bn254Fq>("Bn254Fq (8 limbs)");
    bad += run<PallasFp>("PallasFp (8 limbs, sparse modulus)");
    bad += run<Bls381Fr>("Bls381Fr (8 limbs)");
    printf(bad ? "ASM-CALLEE-CHECK: %d mismatches\n" : "ASM-CALLEE-CHECK: all products agree (%d)\n", bad);
    return bad ? 1 : 0;
}
