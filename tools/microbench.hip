// Instruction-rate and field-op microbenchmarks on the MI355X: the measured integer-MAD peak that
// the "integer roofline" of DESIGN.md / bench.py is priced against.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icontangle-zkcp_amd/csrc tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "zk_curve.h"
using namespace zk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;

__global__ void k_mad64(uint32_t* out, uint32_t a, uint32_t b) {
    uint64_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; i++) {
        x0 = (uint64_t)(uint32_t)x0 * a + x0; x1 = (uint64_t)(uint32_t)x1 * a + x1; x2 = (uint64_t)(uint32_t)x2 * b + x2; x3 = (uint64_t)(uint32_t)x3 * b + x3;
        x4 = (uint64_t)(uint32_t)x4 * a + x4; x5 = (uint64_t)(uint32_t)x5 * a + x5; x6 = (uint64_t)(uint32_t)x6 * b + x6; x7 = (uint64_t)(uint32_t)x7 * b + x7;
    }
    uint64_t s = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
__global__ void k_mullo(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; i++) {
        x0 = x0 * a + 1; x1 = x1 * a + 1; x2 = x2 * b + 1; x3 = x3 * b + 1; x4 = x4 * a + 1; x5 = x5 * a + 1; x6 = x6 * b + 1; x7 = x7 * b + 1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_mulhi(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t x0 = threadIdx.x | 0x80000000u, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; i++) {
        x0 = __umulhi(x0, a) | 0x80000000u; x1 = __umulhi(x1, a) | 0x80000000u; x2 = __umulhi(x2, b) | 0x80000000u; x3 = __umulhi(x3, b) | 0x80000000u;
        x4 = __umulhi(x4, a) | 0x80000000u; x5 = __umulhi(x5, a) | 0x80000000u; x6 = __umulhi(x6, b) | 0x80000000u; x7 = __umulhi(x7, b) | 0x80000000u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_mad24(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; i++) {
        x0 = __umul24(x0, a) + x0; x1 = __umul24(x1, a) + x1; x2 = __umul24(x2, b) + x2; x3 = __umul24(x3, b) + x3;
        x4 = __umul24(x4, a) + x4; x5 = __umul24(x5, a) + x5; x6 = __umul24(x6, b) + x6; x7 = __umul24(x7, b) + x7;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_add32(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; i++) {
        x0 = (x0 + a) ^ b; x1 = (x1 + a) ^ b; x2 = (x2 + b) ^ a; x3 = (x3 + b) ^ a; x4 = (x4 + a) ^ b; x5 = (x5 + a) ^ b; x6 = (x6 + b) ^ a; x7 = (x7 + b) ^ a;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_fma64(double* out, double a, double b) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; i++) {
        x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, b, a); x3 = fma(x3, b, a); x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, b, a); x7 = fma(x7, b, a);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
// the 64-bit VALU helpers the lazy-limb product leans on besides the MAD (exact instructions, 8 independent chains)
#define ASM8(INSN)                                                                                                   \
    for (int i = 0; i < ITERS; i++) {                                                                                \
        asm volatile(INSN : "+v"(x0) : "v"(y)); asm volatile(INSN : "+v"(x1) : "v"(y)); asm volatile(INSN : "+v"(x2) : "v"(y)); \
        asm volatile(INSN : "+v"(x3) : "v"(y)); asm volatile(INSN : "+v"(x4) : "v"(y)); asm volatile(INSN : "+v"(x5) : "v"(y)); \
        asm volatile(INSN : "+v"(x6) : "v"(y)); asm volatile(INSN : "+v"(x7) : "v"(y));                              \
    }
__global__ void k_lshladd64(uint32_t* out, uint32_t a, uint32_t b) {
    uint64_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7, y = ((uint64_t)a << 32) | b;
    ASM8("v_lshl_add_u64 %0, %0, 0, %1")
    uint64_t s = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
__global__ void k_lshr64(uint32_t* out, uint32_t a, uint32_t b) {
    uint64_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7, y = ((uint64_t)a << 32) | b;
    ASM8("v_lshrrev_b64 %0, 1, %1")
    uint64_t s = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
__global__ void k_alignbit(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7, y = a ^ b;
    ASM8("v_alignbit_b32 %0, %1, %0, 29")
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_addco(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7, y = a ^ b;
    ASM8("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc")
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}

template <class F, int CH>
__global__ void k_femul(Fe<F>* out, const Fe<F>* in, int iters) {
    Fe<F> x[CH], y = in[threadIdx.x & 7];
    for (int c = 0; c < CH; c++) x[c] = in[(threadIdx.x + c) & 7];
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) fe_mul(x[c], x[c], y);
    }
    Fe<F> s = x[0];
    for (int c = 1; c < CH; c++) fe_add(s, s, x[c]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F>
__global__ void k_feadd(Fe<F>* out, const Fe<F>* in, int iters) {
    Fe<F> x0 = in[threadIdx.x & 7], x1 = in[(threadIdx.x + 1) & 7], y = in[(threadIdx.x + 3) & 7];
    for (int i = 0; i < iters; i++) { fe_add(x0, x0, y); fe_sub(x1, x1, y); }
    fe_add(x0, x0, x1);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0;
}
template <class C>
__global__ void k_madd(XYZZ<C>* out, const Affine<C>* in, int iters) {
    XYZZ<C> acc; xyzz_set_inf(acc);
    for (int i = 0; i < iters; i++) { Affine<C> p = in[(threadIdx.x + i) & 63]; xyzz_add_mixed(acc, p); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// sustained shader clock under the mixed-add instruction stream: shader cycles (s_memtime) per 100 MHz tick (s_memrealtime)
template <class C>
__global__ void k_madd_clock(XYZZ<C>* out, const Affine<C>* in, int iters, unsigned long long* stamps) {
    XYZZ<C> acc; xyzz_set_inf(acc);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) { Affine<C> p = in[(threadIdx.x + i) & 63]; xyzz_add_mixed(acc, p); }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <class L>
float time_ms(L&& launch, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}

template <class F> void fill(std::vector<Fe<F>>& v) { for (size_t i = 0; i < v.size(); i++) { Fe<F> x; fe_zero(x); x.v[0] = 12345 + 7 * i; x.v[3] = 0x9e3779b9u * (i + 1); fe_to_mont(x, x); fe_mul(x, x, x); v[i] = x; } }

template <class F> void bench_field(const char* name, int blocks, int wpb) {
    std::vector<Fe<F>> h(8); fill(h);
    Fe<F>* din; Fe<F>* dout; CK(hipMalloc(&din, sizeof(Fe<F>) * 8)); CK(hipMalloc(&dout, sizeof(Fe<F>) * blocks * wpb * 64));
    CK(hipMemcpy(din, h.data(), sizeof(Fe<F>) * 8, hipMemcpyHostToDevice));
    const int iters = 2048; const double thr = (double)blocks * wpb * 64;
    float t1 = time_ms([&] { hipLaunchKernelGGL((k_femul<F, 1>), dim3(blocks), dim3(wpb * 64), 0, 0, dout, din, iters); });
    float t2 = time_ms([&] { hipLaunchKernelGGL((k_femul<F, 2>), dim3(blocks), dim3(wpb * 64), 0, 0, dout, din, iters); });
    float t3 = time_ms([&] { hipLaunchKernelGGL((k_feadd<F>), dim3(blocks), dim3(wpb * 64), 0, 0, dout, din, iters); });
    printf("%-10s blocks=%d waves/blk=%d  fe_mul x1: %.1f Gmul/s  x2: %.1f Gmul/s   fe_add/sub: %.1f Gop/s\n", name, blocks, wpb,
           thr * iters / t1 / 1e6, thr * iters * 2 / t2 / 1e6, thr * iters * 2 / t3 / 1e6);
    CK(hipFree(din)); CK(hipFree(dout));
}
template <class C> void bench_madd(const char* name, int blocks, int wpb) {
    using F = typename C::Fq;
    std::vector<Affine<C>> h(64);
    // points k*G by repeated host addition
    Affine<C> g; curve_generator(g);
    XYZZ<C> acc; xyzz_set_inf(acc);
    for (int i = 0; i < 64; i++) { xyzz_add_mixed(acc, g); xyzz_to_affine(h[i], acc); }
    Affine<C>* din; XYZZ<C>* dout; CK(hipMalloc(&din, sizeof(Affine<C>) * 64)); CK(hipMalloc(&dout, sizeof(XYZZ<C>) * blocks * wpb * 64));
    CK(hipMemcpy(din, h.data(), sizeof(Affine<C>) * 64, hipMemcpyHostToDevice));
    const int iters = 256; const double thr = (double)blocks * wpb * 64;
    float t = time_ms([&] { hipLaunchKernelGGL((k_madd<C>), dim3(blocks), dim3(wpb * 64), 0, 0, dout, din, iters); });
    printf("%-10s blocks=%d waves/blk=%d  xyzz_add_mixed: %.2f Gadd/s  (%.1f Gmul-equiv/s at 10 mul/add)\n", name, blocks, wpb, thr * iters / t / 1e6, thr * iters * 10 / t / 1e6);
    CK(hipFree(din)); CK(hipFree(dout));
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s %s CUs=%d clock=%d MHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    const int cus = p.multiProcessorCount;
    uint32_t* d; CK(hipMalloc(&d, (size_t)cus * 8 * 1024 * 8));
    for (int wpb : {4, 8, 16}) {   // waves per block; blocks = 2 per CU
        const int blocks = cus * 2; const double lanes = (double)blocks * wpb * 64; const double ops = lanes * ITERS * 8;
        float t;
        t = time_ms([&] { hipLaunchKernelGGL(k_mad64, dim3(blocks), dim3(wpb * 64), 0, 0, d, 0x9e3779b9u, 0x85ebca6bu); });
        printf("waves/CU=%2d  v_mad_u64_u32 : %8.2f Tops/s\n", wpb * 2, ops / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(k_mullo, dim3(blocks), dim3(wpb * 64), 0, 0, d, 0x9e3779b9u, 0x85ebca6bu); });
        printf("waves/CU=%2d  v_mul_lo_u32+add: %6.2f Tops/s\n", wpb * 2, ops / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(k_mulhi, dim3(blocks), dim3(wpb * 64), 0, 0, d, 0x9e3779b9u, 0x85ebca6bu); });
        printf("waves/CU=%2d  v_mul_hi_u32+or: %7.2f Tops/s\n", wpb * 2, ops / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(k_mad24, dim3(blocks), dim3(wpb * 64), 0, 0, d, 0x9e3779u, 0x85ebcau); });
        printf("waves/CU=%2d  v_mad_u32_u24 : %8.2f Tops/s\n", wpb * 2, ops / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(k_add32, dim3(blocks), dim3(wpb * 64), 0, 0, d, 0x9e3779b9u, 0x85ebca6bu); });
        printf("waves/CU=%2d  v_add+v_xor (2 ops): %5.2f Tops/s\n", wpb * 2, 2 * ops / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(k_fma64, dim3(blocks), dim3(wpb * 64), 0, 0, (double*)d, 1.0000001, 0.5); });
        printf("waves/CU=%2d  v_fma_f64     : %8.2f Tops/s\n", wpb * 2, ops / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(k_lshladd64, dim3(blocks), dim3(wpb * 64), 0, 0, d, 0x9e3779b9u, 0x85ebca6bu); });
        printf("waves/CU=%2d  v_lshl_add_u64: %8.2f Tops/s\n", wpb * 2, ops / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(k_lshr64, dim3(blocks), dim3(wpb * 64), 0, 0, d, 0x9e3779b9u, 0x85ebca6bu); });
        printf("waves/CU=%2d  v_lshrrev_b64 : %8.2f Tops/s\n", wpb * 2, ops / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(k_alignbit, dim3(blocks), dim3(wpb * 64), 0, 0, d, 0x9e3779b9u, 0x85ebca6bu); });
        printf("waves/CU=%2d  v_alignbit_b32: %8.2f Tops/s\n", wpb * 2, ops / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL(k_addco, dim3(blocks), dim3(wpb * 64), 0, 0, d, 0x9e3779b9u, 0x85ebca6bu); });
        printf("waves/CU=%2d  v_add_co+v_addc_co (2 ops): %5.2f Tops/s\n", wpb * 2, 2 * ops / t / 1e9);
    }
    for (int wpb : {1, 2, 4}) {
        bench_field<PallasFp>("PallasFp", cus * 4, wpb);
        bench_field<Bls381Fr>("Bls381Fr", cus * 4, wpb);
        bench_field<Bls381Fq>("Bls381Fq", cus * 4, wpb);
    }
    for (int wpb : {1, 2, 4}) { bench_madd<Vesta>("Vesta", cus * 4, wpb); bench_madd<Bls381G1>("Bls381G1", cus * 4, wpb); }
    {   // sustained clock: 4 waves/SIMD of mixed adds for tens of milliseconds
        using C = Vesta; using F = C::Fq;
        std::vector<Affine<C>> h(64);
        Affine<C> g; curve_generator(g);
        XYZZ<C> acc; xyzz_set_inf(acc);
        for (int i = 0; i < 64; i++) { xyzz_add_mixed(acc, g); xyzz_to_affine(h[i], acc); }
        const int blocks = cus * 16;
        Affine<C>* din; XYZZ<C>* dout; unsigned long long* st;
        CK(hipMalloc(&din, sizeof(Affine<C>) * 64)); CK(hipMalloc(&dout, sizeof(XYZZ<C>) * blocks * 64)); CK(hipMalloc(&st, 16 * blocks));
        CK(hipMemcpy(din, h.data(), sizeof(Affine<C>) * 64, hipMemcpyHostToDevice));
        for (int iters : {64, 256, 2048, 8192}) {
            float t = time_ms([&] { hipLaunchKernelGGL((k_madd_clock<C>), dim3(blocks), dim3(64), 0, 0, dout, din, iters, st); }, 3);
            std::vector<unsigned long long> hs(2 * blocks); CK(hipMemcpy(hs.data(), st, 16 * blocks, hipMemcpyDeviceToHost));
            double cyc = 0, tick = 0; for (int b = 0; b < blocks; b++) { cyc += hs[2 * b]; tick += hs[2 * b + 1]; }
            printf("sustained: iters=%5d  %.2f ms  %.2f Gadd/s  shader clock %.0f MHz\n", iters, t, (double)blocks * 64 * iters / t / 1e6, cyc / tick * 100.0);
        }
    }
    return 0;
}
