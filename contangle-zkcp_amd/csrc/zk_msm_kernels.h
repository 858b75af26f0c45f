// HIP kernels for the MSM / NTT hot path (gfx950).  Replaces, behind the C ABI of
// include/zkcp_amd.h, the upstream CPU routines the reference reaches through
// `Groth16::<Bls12_381>::prove` (lib/src/zk/verifiable_encryption.rs:92, encryption.rs:76,
// sample_entries.rs:86, property.rs:133):
//   ark-ec 0.3   msm/variable_base.rs   VariableBaseMSM::multi_scalar_mul      
//   ark-poly 0.3 domain/radix2/fft.rs   Radix2EvaluationDomain::*fft_in_place  
//   halo2_proofs 0.2 arithmetic.rs      best_multiexp / best_fft               
// Design notes (data layout, roofline per kernel) are in DESIGN.md.
// This header: the MSM kernels (curves); the NTT / pointwise kernels are in zk_ntt_kernels.h.
#pragma once
#include "zk_rt.h"
// (zk_rt.h first: it brings in the HIP runtime or the test emulator)
#include "zk_curve.h"
#include "zk_curve29.h"
#include "zk_glv_params.h"

namespace zk {

// ------------------------------------------------------------------------------------------
// MSM: signed-digit Pippenger.
//   1. msm_digits_kernel     scalar -> W signed c-bit digits (u16 codes, window-major) + per block of 4096 scalars the
//                            number of non-zero digits in each (window, bucket range) region
//   2. msm_region_scan_kernel / msm_region_base_kernel   block offsets inside each region, region bases
//   3. msm_stage_kernel      per (block, window): partition the digits by range inside LDS, copy the chunks to their
//                            regions: (point index | sign) and the bucket number inside the range
//   4. msm_sort_kernel       one workgroup per region: LDS histogram of its buckets, LDS scan -> bucket offsets and the
//                            size-ordered bucket list, counting sort inside LDS, one coalesced store of the region
//   5. msm_accumulate_kernel one lane per bucket (largest first), XYZZ mixed adds over its slice
//   6. msm_reduce_kernel     sum_b b*B_b by slices: X_t = W_t + [t*L] S_t
//   7. msm_sum_kernel        per-window tree sums of X_t
//   host: <= a few points per window, Horner over windows (c doublings each).
// Digits are in [-(2^(c-1)-1), 2^(c-1)]; bucket index b-1 (b = |digit| in 1..2^(c-1)) holds the sum of
// (+-)P_i.  No global atomics in the sort of a uniform input (msm_hot_kernel, for the hot regions of skewed witnesses, issues
// one per slice and non-empty bucket): every counter lives in LDS, and both sorting steps permute inside LDS and
// store contiguous chunks.  Every digit is touched four times (digit, stage, histogram, scatter) whatever the number
// of ranges -- the first version had each (window, range) workgroup filter the whole digit row of its window (16 x
// redundant at c = 16) and scatter 4-byte stores across its region.
// ------------------------------------------------------------------------------------------
struct MsmShape {
    uint32_t n;
    int c;            // window bits (<= 16)
    int w0, nw;       // this call accumulates windows [w0, w0 + nw) of the scalar (window-range sharding)
    uint32_t nbk;     // buckets per window = 2^(c-1)
    uint32_t rb;      // buckets per range (power of two, <= 512: a region of the sorted array should fit the LDS of its sort workgroup)
    uint32_t nranges; // ranges per window = nbk / rb
    int mont;         // scalars arrive in Montgomery form (halo2) rather than canonical (ark BigInt)
    uint32_t big_thresh;  // buckets longer than this take the cooperative path
    int split_log;        // every bucket's entry list is cut into 2^split_log pieces summed by different lanes (msm_combine_sub_kernel adds them)
    uint32_t sblk;        // scalars per workgroup of the digit / stage kernels: MSM_SBLK, less for small inputs (>= 64 workgroups)
    uint32_t batch;       // scalar vectors summed over the same bases by this job; nw = batch * nwb windows in all, window-major per vector
    int nwb;              // windows per scalar vector
    uint64_t batch_stride;   // elements between the scalar vectors
    uint32_t seg;         // entries per cooperative segment of an oversized bucket (power of two in [128, MSM_SEG_MAX], from n: msm_seg_len)
    uint32_t pre_n;       // precomputed-table form (ZK_MSM_FLAG_PRECOMPUTED): the real point count; n = pre_w * pre_n table entries,
    uint32_t pre_w;       // ONE bucket set (nwb = 1): entry w * pre_n + i = the digit of scalar i in window w, point [2^(c w)] P_i
};

template <int N>
__device__ __forceinline__ uint32_t word_at(const uint32_t (&s)[N], int idx) {
    uint32_t v = 0;
    ZK_UNROLL
    for (int k = 0; k < N; k++) v = (idx == k) ? s[k] : v;
    return v;
}
template <int N>
__device__ __forceinline__ uint32_t bits_at(const uint32_t (&s)[N], int start, int c) {
    const int idx = start >> 5, off = start & 31;
    uint64_t v = word_at<N>(s, idx);
    v |= (uint64_t)word_at<N>(s, idx + 1) << 32;  // idx+1 == N selects 0
    return (uint32_t)(v >> off) & ((1u << c) - 1);
}

constexpr uint32_t MSM_RANGE = 512;   // buckets per (window, range) region of the sort
constexpr uint32_t MSM_SBLK = 4096;   // scalars per workgroup of the digit / stage kernels (1024 lanes x 4) at full size: MsmShape::sblk

// u16 digit code: two's complement of the signed digit; positive magnitudes reach 2^15 (0x8000),
// negative ones only 2^15 - 1, so the code is unambiguous:  neg <=> code > 0x8000.
__device__ __forceinline__ uint32_t digit_mag(uint16_t code, bool& neg) {
    neg = code > 0x8000u;
    return neg ? 0x10000u - (uint32_t)code : (uint32_t)code;
}
// u32 digit code of the one-bucket-set form (windows of up to 20 bits): magnitude | sign << 31
__device__ __forceinline__ uint32_t digit_mag(uint32_t code, bool& neg) {
    neg = (code >> 31) != 0;
    return code & 0x7fffffffu;
}

// atomicAdd(&ctr[key], 1) for every lane with `valid`, returning the value before the lane's increment -- but the lanes of
// the wave that share the first valid lane's key are served by ONE atomic (ballot + popcount).  For a skewed witness most
// entries of the hot region carry the same bucket (the unit scalars of a Groth16 assignment): same-address LDS atomics
// serialise lane by lane, this does not.  Every lane of the wave must call it (wave-uniform trip counts).
#ifdef ZK_EMU
#define ZK_READLANE(v, l) __shfl((v), (l))
#else
#define ZK_READLANE(v, l) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (l)))   // `l` is wave-uniform here: no LDS round trip
#endif
__device__ __forceinline__ uint32_t wave_agg_inc(uint32_t* ctr, uint32_t key, bool valid) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t vm = __ballot(valid);
    if (vm == 0) return 0;
    const int leader = __ffsll((unsigned long long)vm) - 1;
    const uint32_t hot = ZK_READLANE(key, leader);
    const bool match = valid && key == hot;
    const uint64_t mm = __ballot(match);
    uint32_t base = 0;
    if (lane == (uint32_t)leader) base = atomicAdd(&ctr[hot], (uint32_t)__popcll((unsigned long long)mm));
    base = ZK_READLANE(base, leader);
    if (match) return base + (uint32_t)__popcll((unsigned long long)(mm & ((1ull << lane) - 1)));
    if (valid) return atomicAdd(&ctr[key], 1u);
    return 0;
}

// grid = ceil(n / sh.sblk).  Writes the digit codes window-major (digits[w * n + i]) and counts this block's non-zero
// digits per region: blockcnt[region * nblocks + block], region = window * nranges + range
template <class C>
__global__ void __launch_bounds__(1024) msm_digits_kernel(const Fe<typename C::Fr>* __restrict__ scalars, MsmShape sh,
                                                          uint16_t* __restrict__ digits, uint32_t* __restrict__ blockcnt) {
    using Fr = typename C::Fr;
    ZK_DYN_SHARED(uint32_t, cnt);   // nwb * nranges: the regions of this block's scalar vector
    const uint32_t nreg = (uint32_t)sh.nwb * sh.nranges;
    const uint32_t rb_log = 31u - (uint32_t)__clz(sh.rb);
    const uint32_t nblocks = gridDim.x / sh.batch, blk = blockIdx.x % nblocks, bt = blockIdx.x / nblocks;   // grid = blocks x vectors
    scalars += (uint64_t)bt * sh.batch_stride;
    digits += (uint64_t)bt * sh.nwb * sh.n;
    for (uint32_t j = threadIdx.x; j < nreg; j += blockDim.x) cnt[j] = 0;
    __syncthreads();
    // few ranges per window (small c: the later IPA rounds, small keys): the lanes of a wave mostly hit the SAME counter, and
    // same-address LDS atomics serialise lane by lane (0.12 ms for 2^12 scalars x 32 windows in one workgroup) -- one
    // aggregated atomic per wave and counter instead.  Trip counts are wave-uniform: no lane leaves the loops early.
    const bool agg = sh.nranges <= 8;
    for (uint32_t k = threadIdx.x; k < sh.sblk; k += blockDim.x) {
        const uint32_t i = blk * sh.sblk + k;
        const bool valid = i < sh.n;
        Fe<Fr> x;
        fe_zero(x);
        if (valid) {
            x = scalars[i];
            if (sh.mont) fe_from_mont(x, x);
        }
        uint32_t carry = 0;
        for (int w = 0; w < sh.w0 + sh.nwb; w++) {
            const uint32_t raw = bits_at<Fr::N>(x.v, w * sh.c, sh.c) + carry;
            const bool neg = raw > sh.nbk;
            const uint32_t mag = neg ? (1u << sh.c) - raw : raw;
            carry = neg ? 1u : 0u;
            if (w >= sh.w0) {
                const uint32_t wl = (uint32_t)(w - sh.w0);
                if (valid) digits[(uint64_t)wl * sh.n + i] = (uint16_t)(neg ? 0x10000u - mag : mag);
                const uint32_t key = wl * sh.nranges + (mag ? (mag - 1) >> rb_log : 0u);
                if (agg)
                    wave_agg_inc(cnt, key, mag != 0);
                else if (mag != 0)
                    atomicAdd(&cnt[key], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < nreg; j += blockDim.x) blockcnt[((uint64_t)bt * nreg + j) * nblocks + blk] = cnt[j];
}

// The precomputed-table form of the digit kernel: the same digit array (window-major = the entry order of the table
// [2^(c w)] P_i), but ONE bucket set -- a digit's region is its bucket range alone -- and the "blocks" of the later kernels are
// the (window, scalar block) pairs: block index w * nblocks + blk.  Pass 1 writes the digits, pass 2 counts every window's
// digits per range from the block's own writes.
template <class C>
__global__ void __launch_bounds__(1024) msm_digits_pre_kernel(const Fe<typename C::Fr>* __restrict__ scalars, MsmShape sh,
                                                              uint32_t* __restrict__ digits, uint32_t* __restrict__ blockcnt) {
    using Fr = typename C::Fr;
    ZK_DYN_SHARED(uint32_t, cnt);   // nranges
    const uint32_t R = sh.nranges;
    const uint32_t rb_log = 31u - (uint32_t)__clz(sh.rb);
    const uint32_t nblocks = gridDim.x / sh.batch, blk = blockIdx.x % nblocks, bt = blockIdx.x / nblocks;
    scalars += (uint64_t)bt * sh.batch_stride;
    digits += (uint64_t)bt * sh.n;
    for (uint32_t k = threadIdx.x; k < sh.sblk; k += blockDim.x) {
        const uint32_t i = blk * sh.sblk + k;
        if (i < sh.pre_n) {
            Fe<Fr> x = scalars[i];
            if (sh.mont) fe_from_mont(x, x);
            uint32_t carry = 0;
            for (uint32_t w = 0; w < sh.pre_w; w++) {
                const uint32_t raw = bits_at<Fr::N>(x.v, (int)(w * sh.c), sh.c) + carry;
                const bool neg = raw > sh.nbk;
                const uint32_t mag = neg ? (1u << sh.c) - raw : raw;
                carry = neg ? 1u : 0u;
                digits[(uint64_t)w * sh.pre_n + i] = mag | (neg ? 0x80000000u : 0u);
            }
        }
    }
    __syncthreads();
    const uint32_t nblocks_all = nblocks * sh.pre_w;
    for (uint32_t w = 0; w < sh.pre_w; w++) {
        for (uint32_t j = threadIdx.x; j < R; j += blockDim.x) cnt[j] = 0;
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < sh.sblk; k += blockDim.x) {
            const uint32_t i = blk * sh.sblk + k;
            if (i < sh.pre_n) {
                bool neg;
                const uint32_t mag = digit_mag(digits[(uint64_t)w * sh.pre_n + i], neg);
                if (mag != 0) atomicAdd(&cnt[(mag - 1) >> rb_log], 1u);
            }
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < R; j += blockDim.x)
            blockcnt[((uint64_t)bt * R + j) * nblocks_all + (w * nblocks + blk)] = cnt[j];
        __syncthreads();
    }
}

// grid = regions: exclusive scan of the region's block counts in place; wg_total[region] = their sum
template <class Tag>
__global__ void __launch_bounds__(1024) msm_region_scan_kernel(uint32_t* __restrict__ blockcnt, uint32_t nblocks, uint32_t* __restrict__ wg_total) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry_s;
    const uint32_t tid = threadIdx.x, nth = blockDim.x;
    uint32_t* row = blockcnt + (uint64_t)blockIdx.x * nblocks;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += nth) {
        const uint32_t v = base + tid < nblocks ? row[base + tid] : 0;
        part[tid] = v;
        __syncthreads();
        for (uint32_t d = 1; d < nth; d <<= 1) {
            const uint32_t u = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += u;
            __syncthreads();
        }
        const uint32_t carry = carry_s;
        if (base + tid < nblocks) row[base + tid] = carry + part[tid] - v;
        __syncthreads();
        if (tid == nth - 1) carry_s = carry + part[tid];
        __syncthreads();
    }
    if (tid == 0) wg_total[blockIdx.x] = carry_s;
}

// one workgroup: region_base[r] = sum of wg_total[0 .. r)   (a thousand regions at most)
template <class Tag>
__global__ void __launch_bounds__(1024) msm_region_base_kernel(const uint32_t* __restrict__ wg_total, uint32_t nreg, uint32_t* __restrict__ region_base) {
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x, nth = blockDim.x;
    const uint32_t per = (nreg + nth - 1) / nth;
    const uint32_t lo = tid * per < nreg ? tid * per : nreg, hi = lo + per < nreg ? lo + per : nreg;
    uint32_t sum = 0;
    for (uint32_t j = lo; j < hi; j++) sum += wg_total[j];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < nth; d <<= 1) {
        const uint32_t u = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += u;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (uint32_t j = lo; j < hi; j++) {
        region_base[j] = run;
        run += wg_total[j];
    }
}

// grid = nblocks * nw: workgroup (block b, window w) partitions the <= sh.sblk digits of its block by bucket range INSIDE
// LDS and then copies every range's chunk to its place in the region (window, range): entry = (point index | sign << 31)
// in stage_idx and the bucket number inside the range in stage_low.  The copy is what makes this cheap: a lane-per-digit
// scatter would issue one 4-byte store request per digit to L2 (33 M requests at 2^20 x 16), the chunk copy issues a
// few requests per 64-entry chunk.
// LDS: e_idx[sblk] u32 | starts[R + 1] | gdst[R] | lcur[R] | e_low[sblk] u16 | e_h[sblk] u16      (R = nranges <= 64; <= 1024 in the
// one-bucket-set form).  blockDim.x >= R: one lane per range sets up its chunk.
template <class DT>   // uint16_t codes (a window's digits), uint32_t codes (the one-bucket-set form)
__global__ void __launch_bounds__(1024) msm_stage_kernel(const DT* __restrict__ digits, MsmShape sh, const uint32_t* __restrict__ blockoff,
                                                         const uint32_t* __restrict__ wg_total, const uint32_t* __restrict__ region_base,
                                                         uint32_t nblocks, uint32_t* __restrict__ stage_idx, uint16_t* __restrict__ stage_low) {
    ZK_DYN_SHARED(uint32_t, lds);
    const uint32_t R = sh.nranges;
    uint32_t* e_idx = lds;
    uint32_t* starts = e_idx + sh.sblk;
    uint32_t* gdst = starts + (R + 1);
    uint32_t* lcur = gdst + R;
    uint16_t* e_low = reinterpret_cast<uint16_t*>(lcur + R);
    uint16_t* e_h = e_low + sh.sblk;
    const uint32_t tid = threadIdx.x, nth = blockDim.x;
    const uint32_t blk = blockIdx.x % nblocks, wl = blockIdx.x / nblocks;
    const uint32_t rb_log = 31u - (uint32_t)__clz(sh.rb);
    // chunk sizes of this block (from the scanned block counts), their exclusive scan, their destinations
    uint32_t size = 0;
    if (tid < R) {
        const uint32_t r = wl * R + tid;
        const uint32_t off = blockoff[(uint64_t)r * nblocks + blk];
        const uint32_t next = blk + 1 < nblocks ? blockoff[(uint64_t)r * nblocks + blk + 1] : wg_total[r];
        size = next - off;
        gdst[tid] = region_base[r] + off;
        lcur[tid] = 0;
        starts[tid + 1] = size;
    }
    if (tid == 0) starts[0] = 0;
    __syncthreads();
    for (uint32_t d = 1; d < R; d <<= 1) {   // inclusive Hillis-Steele on starts[1 .. R]
        const uint32_t u = (tid < R && tid >= d) ? starts[tid + 1 - d] : 0;
        __syncthreads();
        if (tid < R) starts[tid + 1] += u;
        __syncthreads();
    }
    const DT* row = digits + (uint64_t)wl * sh.n;
    for (uint32_t k = tid; k < sh.sblk; k += nth) {
        const uint32_t i = blk * sh.sblk + k;
        if (i >= sh.n) break;
        bool neg;
        const uint32_t mag = digit_mag(row[i], neg);
        if (mag != 0) {
            const uint32_t b = mag - 1, h = b >> rb_log;
            const uint32_t pos = starts[h] + atomicAdd(&lcur[h], 1u);
            e_idx[pos] = i | (neg ? 0x80000000u : 0u);
            e_low[pos] = (uint16_t)(b & (sh.rb - 1));
            e_h[pos] = (uint16_t)h;
        }
    }
    __syncthreads();
    const uint32_t total = starts[R];
    for (uint32_t k = tid; k < total; k += nth) {
        const uint32_t h = e_h[k];
        const uint32_t d = gdst[h] + (k - starts[h]);
        stage_idx[d] = e_idx[k];
        stage_low[d] = e_low[k];
    }
}

constexpr uint32_t MSM_HOT_UNROLL = 8;   // entries per lane and iteration in the hot-region loops of the sort kernel

// ---- hot regions (far above the mean: a skewed witness) are histogrammed and scattered by many workgroups ----
constexpr uint32_t MSM_HOT_MAX = 8;       // hot regions served this way (the rest stay with their own sort workgroup)
constexpr uint32_t MSM_HOT_SLICES = 32;   // workgroups per hot region at most (n / 65536 of them for smaller inputs)

// one workgroup: hot_flag[r] = 1 + rank for the first MSM_HOT_MAX regions with more than chunk_limit entries (0 otherwise),
// hot_list[0] = their number, hot_list[1 + k] = the region of rank k.  nreg <= 1024.
template <class Tag>
__global__ void __launch_bounds__(1024) msm_hot_list_kernel(const uint32_t* __restrict__ wg_total, uint32_t nreg, uint32_t chunk_limit,
                                                            uint32_t* __restrict__ hot_flag, uint32_t* __restrict__ hot_list) {
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x, nth = blockDim.x;
    const uint32_t f = (tid < nreg && wg_total[tid] > chunk_limit) ? 1u : 0u;
    part[tid] = f;
    __syncthreads();
    for (uint32_t d = 1; d < nth; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const uint32_t rank = part[tid] - f;
    if (tid < nreg) hot_flag[tid] = (f && rank < MSM_HOT_MAX) ? rank + 1 : 0u;
    if (f && rank < MSM_HOT_MAX) hot_list[1 + rank] = tid;
    if (tid == nth - 1) hot_list[0] = part[tid] < MSM_HOT_MAX ? part[tid] : MSM_HOT_MAX;
}

// grid = MSM_HOT_MAX * slices; workgroup (k, s) takes slice s of the hot region of rank k: an LDS histogram of the
// slice (wave-aggregated), then  phase 0: counts[bucket] += slice count (global atomics, one per non-empty bucket)
//                                phase 1: reserve the slice's run of every bucket at gcur[bucket], scatter through LDS cursors
// LDS: hist[rb] | cur[rb]
template <class Tag>
__global__ void __launch_bounds__(1024) msm_hot_kernel(const uint32_t* __restrict__ stage_idx, const uint16_t* __restrict__ stage_low,
                                                       MsmShape sh, const uint32_t* __restrict__ region_base,
                                                       const uint32_t* __restrict__ wg_total, const uint32_t* __restrict__ hot_list,
                                                       uint32_t* __restrict__ counts, uint32_t* __restrict__ gcur,
                                                       uint32_t* __restrict__ sorted, int phase, uint32_t slices) {
    ZK_DYN_SHARED(uint32_t, lds);
    uint32_t* hist = lds;
    uint32_t* cur = hist + sh.rb;
    const uint32_t tid = threadIdx.x, nth = blockDim.x;
    const uint32_t k = blockIdx.x / slices, sl = blockIdx.x % slices;
    if (k >= hot_list[0]) return;
    const uint32_t r = hot_list[1 + k];
    const uint32_t base = region_base[r], total = wg_total[r];
    const uint32_t len = (total + slices - 1) / slices;
    const uint32_t lo = sl * len < total ? sl * len : total, hi = lo + len < total ? lo + len : total;
    const uint64_t gb0 = (uint64_t)r * sh.rb;     // regions are consecutive runs of rb buckets
    for (uint32_t j = tid; j < sh.rb; j += nth) hist[j] = 0;
    __syncthreads();
    for (uint32_t e0 = lo; e0 < hi; e0 += nth * MSM_HOT_UNROLL) {
        uint32_t lows[MSM_HOT_UNROLL];
        ZK_UNROLL
        for (uint32_t u = 0; u < MSM_HOT_UNROLL; u++) {
            const uint32_t e = e0 + u * nth + tid;
            lows[u] = e < hi ? (uint32_t)stage_low[base + e] : 0xffffffffu;
        }
        ZK_UNROLL
        for (uint32_t u = 0; u < MSM_HOT_UNROLL; u++) wave_agg_inc(hist, lows[u], lows[u] != 0xffffffffu);
    }
    __syncthreads();
    if (phase == 0) {
        for (uint32_t j = tid; j < sh.rb; j += nth)
            if (hist[j]) atomicAdd(&counts[gb0 + j], hist[j]);
        return;
    }
    for (uint32_t j = tid; j < sh.rb; j += nth) cur[j] = hist[j] ? atomicAdd(&gcur[gb0 + j], hist[j]) : 0u;
    __syncthreads();
    for (uint32_t e0 = lo; e0 < hi; e0 += nth * MSM_HOT_UNROLL) {
        uint32_t lows[MSM_HOT_UNROLL], vals[MSM_HOT_UNROLL];
        ZK_UNROLL
        for (uint32_t u = 0; u < MSM_HOT_UNROLL; u++) {
            const uint32_t e = e0 + u * nth + tid;
            lows[u] = e < hi ? (uint32_t)stage_low[base + e] : 0xffffffffu;
            vals[u] = e < hi ? stage_idx[base + e] : 0u;
        }
        ZK_UNROLL
        for (uint32_t u = 0; u < MSM_HOT_UNROLL; u++) {
            const bool valid = lows[u] != 0xffffffffu;
            const uint32_t p = wave_agg_inc(cur, lows[u], valid);
            if (valid) sorted[p] = vals[u];
        }
    }
}

// grid = regions; workgroup (w, h) owns buckets [h*rb, (h+1)*rb) of window w and entries [region_base, + wg_total) of the
// staged and of the sorted array.  LDS: hist[rb] | cur[rb] | part[1024] | bins[258] | coff[rb] | ccur[rb] | perm[cap] (u16).
// A region of at most `cap` entries is sorted inside LDS -- as a permutation of its entry numbers, 2 bytes each -- and
// written out as one coalesced stream (sorted[k] = staged[perm[k]], the gather served by L2); a larger one in chunks (below).
// Scattering 4-byte stores over the region directly costs ~5x the HBM write traffic.
template <class Tag>
__global__ void __launch_bounds__(1024) msm_sort_kernel(const uint32_t* __restrict__ stage_idx, const uint16_t* __restrict__ stage_low,
                                                        MsmShape sh, const uint32_t* __restrict__ region_base,
                                                        const uint32_t* __restrict__ wg_total, uint32_t* __restrict__ counts,
                                                        uint32_t* __restrict__ offs, uint32_t* __restrict__ order,
                                                        uint32_t* __restrict__ sorted, uint32_t cap, uint32_t chunk_limit,
                                                        const uint32_t* __restrict__ hot_flag, uint32_t* __restrict__ gcur) {
    ZK_DYN_SHARED(uint32_t, lds);
    uint32_t* hist = lds;
    uint32_t* cur = hist + sh.rb;
    uint32_t* part = cur + sh.rb;
    uint32_t* bins = part + 1024;
    uint16_t* perm = reinterpret_cast<uint16_t*>(bins + 258 + 2 * sh.rb);
    const uint32_t tid = threadIdx.x, nth = blockDim.x;
    const uint32_t w = blockIdx.x / sh.nranges, h = blockIdx.x % sh.nranges;
    const uint64_t gb0 = (uint64_t)w * sh.nbk + (uint64_t)h * sh.rb;
    const uint32_t base = region_base[blockIdx.x], total = wg_total[blockIdx.x];
    const bool local = total <= cap;   // cap <= 65536: entry numbers fit a u16
    for (uint32_t j = tid; j < sh.rb; j += nth) hist[j] = 0;
    for (uint32_t j = tid; j < 258; j += nth) bins[j] = 0;
    __syncthreads();
    const bool hot = total > chunk_limit;   // far above the mean region: a skewed witness
    const bool helped = hot && hot_flag != nullptr && hot_flag[blockIdx.x] != 0;   // msm_hot_kernel histograms and scatters it
    if (helped) {
        for (uint32_t j = tid; j < sh.rb; j += nth) hist[j] = counts[gb0 + j];
    } else if (hot) {
        // one workgroup, a third of all entries: MSM_HOT_UNROLL loads in flight per lane, or every entry costs a memory latency
        for (uint32_t e0 = 0; e0 < total; e0 += nth * MSM_HOT_UNROLL) {
            uint32_t lows[MSM_HOT_UNROLL];
            ZK_UNROLL
            for (uint32_t u = 0; u < MSM_HOT_UNROLL; u++) {
                const uint32_t e = e0 + u * nth + tid;
                lows[u] = e < total ? (uint32_t)stage_low[base + e] : 0xffffffffu;
            }
            ZK_UNROLL
            for (uint32_t u = 0; u < MSM_HOT_UNROLL; u++) wave_agg_inc(hist, lows[u], lows[u] != 0xffffffffu);
        }
    } else {
        for (uint32_t e = tid; e < total; e += nth) atomicAdd(&hist[stage_low[base + e]], 1u);
    }
    __syncthreads();
    // exclusive scan of this range's counts: per-lane chunk sums, Hillis-Steele over the lanes, then refill
    const uint32_t per = (sh.rb + nth - 1) / nth;
    const uint32_t jlo = tid * per < sh.rb ? tid * per : sh.rb;
    const uint32_t jhi = jlo + per < sh.rb ? jlo + per : sh.rb;
    uint32_t sum = 0;
    for (uint32_t j = jlo; j < jhi; j++) sum += hist[j];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < nth; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;   // position inside the region
    for (uint32_t j = jlo; j < jhi; j++) {
        const uint32_t cnt = hist[j];
        cur[j] = run;
        offs[gb0 + j] = base + run;
        counts[gb0 + j] = cnt;
        run += cnt;
        atomicAdd(&bins[255u - (cnt < 255u ? cnt : 255u)], 1u);  // size classes, largest first
    }
    __syncthreads();
    // order[]: the buckets of this range sorted by descending size class, so that the 64 buckets a wave of the
    // accumulate kernel works on have (nearly) equal lengths.  Exclusive scan of the 256 class counts first.
    const uint32_t bv = tid < 256 ? bins[tid] : 0;
    part[tid] = bv;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint32_t v = (tid >= d && tid < 256) ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    if (tid < 256) bins[tid] = part[tid] - bv;
    __syncthreads();
    for (uint32_t j = jlo; j < jhi; j++) {
        const uint32_t cnt = hist[j];
        const uint32_t r = atomicAdd(&bins[255u - (cnt < 255u ? cnt : 255u)], 1u);
        order[gb0 + r] = (uint32_t)(gb0 + j);
    }
    // counting-sort scatter through the LDS cursors
    if (local) {
        for (uint32_t e = tid; e < total; e += nth) {
            const uint32_t p = atomicAdd(&cur[stage_low[base + e]], 1u);
            perm[p] = (uint16_t)e;
        }
        __syncthreads();
        for (uint32_t k = tid; k < total; k += nth) sorted[base + k] = stage_idx[base + perm[k]];
    } else {
        if (helped) {
            // the slices of this region are scattered by msm_hot_kernel (phase 1) from these bucket cursors
            for (uint32_t j = tid; j < sh.rb; j += nth) gcur[gb0 + j] = base + cur[j];
            return;
        }
        if (hot) {
            // (a skewed witness: the unit scalars of a Groth16 assignment all land in one bucket) this workgroup is the
            // critical path of the launch: wave-aggregated cursors, stores straight to their places
            for (uint32_t e0 = 0; e0 < total; e0 += nth * MSM_HOT_UNROLL) {
                uint32_t lows[MSM_HOT_UNROLL], vals[MSM_HOT_UNROLL];
                ZK_UNROLL
                for (uint32_t u = 0; u < MSM_HOT_UNROLL; u++) {
                    const uint32_t e = e0 + u * nth + tid;
                    lows[u] = e < total ? (uint32_t)stage_low[base + e] : 0xffffffffu;
                    vals[u] = e < total ? stage_idx[base + e] : 0u;
                }
                ZK_UNROLL
                for (uint32_t u = 0; u < MSM_HOT_UNROLL; u++) {
                    const bool valid = lows[u] != 0xffffffffu;
                    const uint32_t p = wave_agg_inc(cur, lows[u], valid);
                    if (valid) sorted[base + p] = vals[u];
                }
            }
            return;
        }
        // A region larger than the permutation buffer because n is large (>= 2^22) is sorted in chunks of cap / 2
        // entries: each chunk is counting-sorted inside LDS (entry number + bucket per sorted position) and every bucket's
        // run of the chunk is appended at the bucket's running position cur[] -- runs of chunk / rb entries instead of
        // single 4-byte stores.
        uint32_t* coff = bins + 258;          // exclusive offsets of the chunk's buckets
        uint32_t* ccur = coff + sh.rb;        // counts, then cursors
        uint16_t* cperm = perm;               // entry number inside the chunk, by sorted position
        const uint32_t ccap = cap >> 1;
        uint16_t* cbkt = perm + ccap;         // bucket of the sorted position
        for (uint32_t c0 = 0; c0 < total; c0 += ccap) {
            const uint32_t clen = total - c0 < ccap ? total - c0 : ccap;
            for (uint32_t j = tid; j < sh.rb; j += nth) ccur[j] = 0;
            __syncthreads();
            for (uint32_t e = tid; e < clen; e += nth) atomicAdd(&ccur[stage_low[base + c0 + e]], 1u);
            __syncthreads();
            uint32_t csum = 0;
            for (uint32_t j = jlo; j < jhi; j++) csum += ccur[j];
            part[tid] = csum;
            __syncthreads();
            for (uint32_t d = 1; d < nth; d <<= 1) {
                const uint32_t v = tid >= d ? part[tid - d] : 0;
                __syncthreads();
                part[tid] += v;
                __syncthreads();
            }
            uint32_t crun = part[tid] - csum;
            for (uint32_t j = jlo; j < jhi; j++) {
                const uint32_t cnt = ccur[j];
                coff[j] = crun;
                ccur[j] = crun;
                crun += cnt;
            }
            __syncthreads();
            for (uint32_t e = tid; e < clen; e += nth) {
                const uint32_t j = stage_low[base + c0 + e];
                const uint32_t p = atomicAdd(&ccur[j], 1u);
                cperm[p] = (uint16_t)e;
                cbkt[p] = (uint16_t)j;
            }
            __syncthreads();
            for (uint32_t p = tid; p < clen; p += nth) {
                const uint32_t j = cbkt[p];
                sorted[base + cur[j] + (p - coff[j])] = stage_idx[base + c0 + cperm[p]];
            }
            __syncthreads();
            for (uint32_t j = jlo; j < jhi; j++) cur[j] += ccur[j] - coff[j];
            __syncthreads();
        }
    }
}

// Bucket accumulation: persistent waves pulling 64-bucket tasks from a queue, largest size class first
// (longest-processing-time order, so the SIMDs finish together).  Task t = (rank r, range g): lane l sums
// bucket order[g*rb + 64 r + l].  Buckets longer than sh.big_thresh (a few times the mean length: skewed
// witnesses, e.g. the 25% of unit scalars of a Groth16 assignment) are not summed by one lane: they are cut
// into segments of sh.seg entries appended to `seg_list` for msm_accumulate_big_kernel.
// Entries per cooperative segment (one wave: len / 64 additions per lane + a 6-level tree).  The worst oversized bucket of a
// Groth16 assignment holds the unit scalars, a quarter of all entries: with 2048-entry segments a 2^20-point MSM cuts it into
// 128 segments -- 128 waves on 1024 SIMDs, each doing 38 dependent additions (1.35 ms per G2 launch in round 2).  The length
// now follows n so that such a bucket yields ~1024 segments: n / 4096, at least 128 (below that the tree dominates).
constexpr uint32_t MSM_SEG_MAX = 2048;
inline uint32_t msm_seg_len(uint64_t n) {
    uint32_t s = 128;
    while (s < MSM_SEG_MAX && (uint64_t)s * 4096 < n) s <<= 1;
    return s;
}

// lanes per workgroup of the LDS tree kernels: 256 XYZZ points must fit the 64 KiB static LDS limit (G2: 384 B each)
template <class C>
constexpr uint32_t tree_lanes() {
    return sizeof(XYZZ<C>) * 256 <= 48 * 1024 ? 256u : 128u;
}

struct MsmQueue {          // device-side control words, zeroed before every MSM
    uint32_t head;         // next task
    uint32_t nseg;         // segments appended
    uint32_t nbig;         // big buckets appended
    uint32_t pad;
};
struct MsmSeg {
    uint32_t bucket, start, len, big_index;
};

// How the bases of a kernel view are kept in HBM.  The saturated views read the points as uploaded.  The lazy-limb views
// keep x R' mod p (a value below 2p) PACKED in the caller's 32-bit words -- 64 B per G1 point (96 B BLS12-381), 128 / 192 B on
// G2, the size of the uploaded point -- and unpack to 29 / 28-bit limbs in registers (18 shift / mask pairs per point, ~1.5 %
// of a mixed addition): every window gathers every base once, so the record size is the kernel's HBM / Infinity-Cache
// traffic (a 64-B record is one aligned fabric request; the 80-B unpacked record straddled two).
template <class CK>
struct StoredBase {
    using type = Affine<CK>;
};
template <class C>
struct StoredBase<C29<C>> {
    using type = Affine<C>;
};
template <class C>
struct StoredBase<C29x2<C>> {
    using type = Affine<C>;
};
template <class CK>
using StoredAffine = typename StoredBase<CK>::type;

template <class CK>
__device__ __forceinline__ void load_base(Affine<CK>& p, const Affine<CK>* __restrict__ bases, uint32_t idx) {
    p = bases[idx];
}
template <class C>
__device__ __forceinline__ void load_base(Affine<C29<C>>& p, const Affine<C>* __restrict__ bases, uint32_t idx) {
    const Affine<C> raw = bases[idx];
    fe29_unpack(p.x, raw.x);
    fe29_unpack(p.y, raw.y);
}
template <class C>
__device__ __forceinline__ void load_base(Affine<C29x2<C>>& p, const Affine<C>* __restrict__ bases, uint32_t idx) {
    const Affine<C> raw = bases[idx];
    fe29_unpack(p.x.c0, raw.x.c0);
    fe29_unpack(p.x.c1, raw.x.c1);
    fe29_unpack(p.y.c0, raw.y.c0);
    fe29_unpack(p.y.c1, raw.y.c1);
}

template <class C>
__device__ __forceinline__ void accumulate_slice(XYZZ<C>& acc, const StoredAffine<C>* __restrict__ bases,
                                                 const uint32_t* __restrict__ sorted, uint32_t start, uint32_t cnt,
                                                 uint32_t stride) {
    for (uint32_t k = 0; k < cnt; k += stride) {
        const uint32_t e = sorted[start + k];
        Affine<C> p;
        load_base(p, bases, e & 0x7fffffffu);
        aff_neg_if(p, (e >> 31) != 0);
        xyzz_add_mixed(acc, p);
    }
}

// Persistent waves; every LANE streams buckets: when a lane finishes its bucket it takes the next one from the
// wave's batch (LDS cursor), and the wave refills its batch from the global queue 64 buckets at a time.  Buckets
// are visited largest size class first (rank-major over the per-range size-sorted lists), so the queue drains
// into the shortest buckets and all SIMDs finish together; lanes never wait for a longer neighbour.
#ifndef ZK_ACC_WAVES_9
#define ZK_ACC_WAVES_9 4
#endif
constexpr uint32_t MSM_BATCH = 64;
// Visiting order: rank-major over the per-range size-sorted bucket lists, MSM_RANKW buckets of a range at a time.  The
// narrower the rank, the closer the order is to globally largest-first (512-bucket ranges: 32 ranks).
constexpr uint32_t MSM_RANKW = 16;

// resident waves per SIMD the accumulate kernel is compiled for: the 9-limb lazy form needs 134 VGPRs unconstrained (3 waves);
// held to 128 (8 dwords of scratch) a fourth wave fits -- issue efficiency grows with the number of resident waves
template <class C>
constexpr int msm_acc_waves() {
    if constexpr (C::EXT == 29) return F29<typename C::Fq>::L <= 9 ? ZK_ACC_WAVES_9 : 2;
    if constexpr (C::EXT == 58) return F29<typename C::Fq>::L <= 9 ? 2 : 1;
    return 4;
}
template <class C>
__global__ void __launch_bounds__(64, msm_acc_waves<C>()) msm_accumulate_kernel(const StoredAffine<C>* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                      const uint32_t* __restrict__ offs, const uint32_t* __restrict__ counts,
                                      const uint32_t* __restrict__ order, XYZZ<C>* __restrict__ buckets, MsmShape sh,
                                      MsmQueue* __restrict__ q, MsmSeg* __restrict__ seg_list, uint32_t* __restrict__ big_list) {
    __shared__ uint32_t s_next, s_end;
    const uint32_t nrt = (uint32_t)sh.nw * sh.nranges;                 // ranges in this call
    const uint32_t per_rank = nrt * MSM_RANKW;                         // bucket positions per size rank
    const uint32_t S = 1u << sh.split_log;                             // pieces per bucket
    const uint32_t total = (((sh.rb + MSM_RANKW - 1) / MSM_RANKW) * per_rank) << sh.split_log;   // visiting positions (ranges padded to whole ranks)
    const uint32_t lane = threadIdx.x;
    if (lane == 0) {
        s_next = 0;
        s_end = 0;
    }
    __syncthreads();
    bool have = false, drained = false;
    uint32_t out_slot = 0, pos = 0, end = 0;
    XYZZ<C> acc;
    xyzz_set_inf(acc);
    for (;;) {
        // ---- lanes without work take the next position of the wave's batch: a (bucket, piece) pair
        bool want_refill = false;
        if (!have && !drained) {
            const uint32_t i = atomicAdd(&s_next, 1u);
            if (i < s_end) {
                const uint32_t ib = i >> sh.split_log, piece = i & (S - 1);
                const uint32_t r = ib / per_rank, rem = ib % per_rank;
                const uint32_t g = rem / MSM_RANKW, slot = r * MSM_RANKW + (rem % MSM_RANKW);
                if (slot < sh.rb) {
                    const uint32_t gb = order[(uint64_t)g * sh.rb + slot];
                    const uint32_t cnt = counts[gb];
                    out_slot = (gb << sh.split_log) + piece;
                    // this lane's share of the bucket's sorted slice
                    const uint32_t plen = (cnt + S - 1) >> sh.split_log;
                    uint32_t lo = piece * plen, hi = lo + plen;
                    if (lo > cnt) lo = cnt;
                    if (hi > cnt) hi = cnt;
                    const uint32_t start = offs[gb];
                    if (cnt > sh.big_thresh) {
                        xyzz_set_inf(acc);
                        buckets[out_slot] = acc;   // the cooperative path below owns this bucket
                        if (piece == 0) {   // (no `continue` here: every lane must reach the wave votes below)
                            const uint32_t ns = (cnt + sh.seg - 1) / sh.seg;
                            const uint32_t bi = atomicAdd(&q->nbig, 1u);
                            const uint32_t s0 = atomicAdd(&q->nseg, ns);
                            big_list[2 * bi] = gb;
                            big_list[2 * bi + 1] = s0;
                            for (uint32_t k = 0; k < ns; k++) {
                                MsmSeg sg;
                                sg.bucket = gb;
                                sg.start = start + k * sh.seg;
                                sg.len = (k + 1 == ns) ? cnt - k * sh.seg : sh.seg;
                                sg.big_index = bi;
                                seg_list[s0 + k] = sg;
                            }
                        }
                    } else if (lo == hi) {
                        xyzz_set_inf(acc);
                        buckets[out_slot] = acc;
                    } else {
                        have = true;
                        pos = start + lo;
                        end = start + hi;
                        xyzz_set_inf(acc);
                    }
                }
            } else {
                want_refill = true;
            }
        }
        // ---- one mixed add for every lane that owns a bucket
        if (have) {
            const uint32_t e = sorted[pos];
            Affine<C> p;
            load_base(p, bases, e & 0x7fffffffu);
            aff_neg_if(p, (e >> 31) != 0);
            xyzz_add_mixed(acc, p);
            if (++pos == end) {
                buckets[out_slot] = acc;
                have = false;
            }
        }
        if (__syncthreads_count(want_refill) != 0) {   // wave-uniform
            if (lane == 0) {
                const uint32_t base = atomicAdd(&q->head, MSM_BATCH);
                s_next = base < total ? base : total;
                s_end = base + MSM_BATCH < total ? base + MSM_BATCH : total;
            }
            __syncthreads();
            if (s_next >= s_end) drained = true;        // the global queue is empty: nothing left to take
        }
        if (__syncthreads_count(have || !drained) == 0) break;  // every lane has stored its last bucket
    }
}

// buckets[gb] = sum of the 2^split_log pieces of bucket gb (only launched when split_log > 0)
template <class C>
__global__ void __launch_bounds__(64) msm_combine_sub_kernel(const XYZZ<C>* __restrict__ sub, XYZZ<C>* __restrict__ buckets, uint32_t nbuckets,
                                                             int split_log) {
    const uint32_t gb = blockIdx.x * blockDim.x + threadIdx.x;
    if (gb >= nbuckets) return;
    XYZZ<C> acc = sub[(uint64_t)gb << split_log];
    for (uint32_t s = 1; s < (1u << split_log); s++) {
        XYZZ<C> p = sub[((uint64_t)gb << split_log) + s];
        xyzz_add(acc, p);
    }
    buckets[gb] = acc;
}

// One wave per segment of an oversized bucket: lane-strided partial sums + LDS tree.
// Fixed grid; waves stride over the device-side segment list.
template <class C>
__global__ void __launch_bounds__(64) msm_accumulate_big_kernel(const StoredAffine<C>* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                                const MsmQueue* __restrict__ q, const MsmSeg* __restrict__ seg_list,
                                                                XYZZ<C>* __restrict__ seg_out) {
    __shared__ XYZZ<C> sh[64];
    const uint32_t tid = threadIdx.x;
    const uint32_t nseg = q->nseg;
    for (uint32_t s = blockIdx.x; s < nseg; s += gridDim.x) {
        const MsmSeg sg = seg_list[s];
        XYZZ<C> acc;
        xyzz_set_inf(acc);
        if (tid < sg.len) accumulate_slice<C>(acc, bases, sorted, sg.start + tid, sg.len - tid, 64);
        sh[tid] = acc;
        __syncthreads();
        for (uint32_t d = 32; d > 0; d >>= 1) {
            if (tid < d) {
                XYZZ<C> b = sh[tid + d];
                xyzz_add(acc, b);
                sh[tid] = acc;
            }
            __syncthreads();
        }
        if (tid == 0) seg_out[s] = acc;
        __syncthreads();
    }
}

// bucket = sum of its segments: one workgroup per oversized bucket (the grid covers every bucket that can be oversized, the
// others leave at once).  Every addition in here is a DEPENDENT one (~36 us each on the BLS12-381 G2 point type at one wave per
// SIMD), so the shape matters more than the count: a lane adds ceil(ns / TL) segments, then the tree runs over the lanes that
// hold something only -- two segments cost one addition, not a 7-level tree -- and a workgroup never takes a second bucket
// while others idle (round 2: 64 workgroups x ~4 buckets x 8 levels = 0.83 ms per G2 launch of a Groth16 witness).
template <class C>
__global__ void __launch_bounds__(256) msm_combine_big_kernel(const MsmQueue* __restrict__ q, const uint32_t* __restrict__ big_list,
                                                              const uint32_t* __restrict__ counts, const XYZZ<C>* __restrict__ seg_out,
                                                              XYZZ<C>* __restrict__ buckets, uint32_t seg) {
    constexpr uint32_t TL = tree_lanes<C>();
    __shared__ XYZZ<C> sh[TL];
    const uint32_t tid = threadIdx.x;
    const uint32_t nbig = q->nbig;
    for (uint32_t b = blockIdx.x; b < nbig; b += gridDim.x) {
        const uint32_t gb = big_list[2 * b], s0 = big_list[2 * b + 1];
        const uint32_t ns = (counts[gb] + seg - 1) / seg;
        const uint32_t live = ns < TL ? ns : TL;              // lanes that hold a partial sum
        XYZZ<C> acc;
        xyzz_set_inf(acc);
        for (uint32_t k = tid; k < ns; k += TL) {
            XYZZ<C> v = seg_out[s0 + k];
            xyzz_add(acc, v);
        }
        sh[tid] = acc;
        __syncthreads();
        uint32_t top = 1;
        while (top < live) top <<= 1;
        for (uint32_t d = top >> 1; d > 0; d >>= 1) {
            if (tid < d && tid + d < live) {
                XYZZ<C> v = sh[tid + d];
                xyzz_add(acc, v);
                sh[tid] = acc;
            }
            __syncthreads();
        }
        if (tid == 0) buckets[gb] = acc;
        __syncthreads();
    }
}

// slice t of window w covers bucket indices [t*L, (t+1)*L) (weights index+1):
//   X_t = sum_l (l+1) B_{tL+l} + [t*L] * sum_l B_{tL+l}
// The kernel runs at one wave per SIMD, so its instruction stream is all that matters -- and written naively (two additions in
// the running-sum loop, a doubling and an addition in the multiplier loop, one more addition at the end) it is 175 KB of code
// against a 64 KB instruction cache: every pass streams from L2 and the kernel slows by 25% whenever the kernel before it
// has just swept L2 (measured: tools/shard_model.py with ZK_MSM_SPLIT).  It is therefore a little state machine with ONE
// addition site and ONE doubling site (xyzz_add_nodbl + xyzz_dbl, ~40 KB) and wave-uniform operand selection:
//   steps 0 .. 2L-1        run += B_l ; wsum += run          (l = L-1 .. 0, the next bucket prefetched under the additions)
//   then                   t2 = 2 run ; t3 = t2 + run
//   3 per two bits of t    acc = 4 acc (two steps) ; acc += run / t2 / t3 by the digit   (MSB first, uniform digit count)
//   log2 L                 acc = 2 acc
//   last                   wsum += acc
template <class C>
__global__ void __launch_bounds__(64) msm_reduce_kernel(const XYZZ<C>* __restrict__ buckets, XYZZ<C>* __restrict__ out, uint32_t nbk, uint32_t L,
                                  uint32_t slices_per_window, uint32_t nslices) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    if (gt >= nslices) return;
    const uint32_t w = gt / slices_per_window, t = gt % slices_per_window;
    XYZZ<C> run, wsum, acc, b;
    xyzz_set_inf(run);
    xyzz_set_inf(wsum);
    xyzz_set_inf(acc);
    xyzz_set_inf(b);
    const XYZZ<C>* src = buckets + (uint64_t)w * nbk + (uint64_t)t * L;
    const uint32_t avail = nbk - t * L < L ? nbk - t * L : L;   // the last slice of a window may be short
    if (avail > 0) b = src[avail - 1];
    // multiplier phase: wsum += [t * L] run.  WB-bit windows over t: for WB = 2, run, 2 run, 3 run are precomputed and two
    // bits cost 3 steps instead of 4 (the G1 point types; the wide G2 ones would spill the two extra points and keep WB = 1);
    // then log2(L) doublings (L is a power of two whenever it is > 1).  Step counts are uniform over the grid.
    constexpr uint32_t WB = sizeof(XYZZ<C>) <= 256 ? 2u : 1u;
    constexpr uint32_t PRE = WB == 2 ? 2u : 0u;
    const uint32_t tmax = slices_per_window - 1;
    const uint32_t log_l = 31u - (uint32_t)__clz(L);
    const bool pow2 = (L & (L - 1)) == 0;
    const uint32_t mult = pow2 ? t : t * L;                       // multiplier scanned by the digit loop
    const uint32_t mmax = pow2 ? tmax : tmax * L;
    const uint32_t mbits = mmax ? 32u - (uint32_t)__clz(mmax) : 0u;
    const uint32_t mdig = (mbits + WB - 1) / WB;
    const uint32_t tail = pow2 ? log_l : 0u;
    const uint32_t s_mul = 2 * L;                                 // first step of the multiplier phase
    const uint32_t nsteps = mdig ? s_mul + PRE + (WB + 1) * mdig + tail + 1 : s_mul;
    XYZZ<C> t2, t3;
    if constexpr (WB == 2) {
        xyzz_set_inf(t2);
        xyzz_set_inf(t3);
    }
#pragma unroll 1
    for (uint32_t s = 0; s < nsteps; s++) {
        XYZZ<C> X, Y;
        bool dbl = false, on = true;
        int kind;   // destination: 0 run, 1 wsum, 2 acc, 3 t2, 4 t3
        if (s < s_mul) {
            const uint32_t l = L - 1 - (s >> 1);
            if ((s & 1) == 0) {
                kind = 0;
                X = run;
                Y = b;
                on = l < avail;
                if (on && l > 0) b = src[l - 1];
            } else {
                kind = 1;
                X = wsum;
                Y = run;
            }
        } else if (WB == 2 && s == s_mul) {             // t2 = 2 run
            kind = 3;
            X = run;
            Y = run;
            dbl = true;
        } else if (WB == 2 && s == s_mul + 1) {         // t3 = t2 + run
            kind = 4;
            if constexpr (WB == 2) X = t2;
            Y = run;
        } else if (s < s_mul + PRE + (WB + 1) * mdig) {
            const uint32_t j = s - s_mul - PRE, dg = mdig - 1 - j / (WB + 1), ph = j % (WB + 1);
            kind = 2;
            X = acc;
            if (ph < WB) {
                Y = acc;
                dbl = true;
            } else {
                const uint32_t v = (mult >> (WB * dg)) & ((1u << WB) - 1);
                on = v != 0;
                Y = run;
                if constexpr (WB == 2) {
                    if (v == 2) Y = t2;
                    if (v == 3) Y = t3;
                }
            }
        } else if (s < nsteps - 1) {         // acc = L * acc
            kind = 2;
            X = acc;
            Y = acc;
            dbl = true;
        } else {
            kind = 1;
            X = wsum;
            Y = acc;
        }
        bool need = dbl;
        if (on && !dbl) need = xyzz_add_nodbl(X, Y);
        if (on && need) xyzz_dbl(X);
        if (kind == 0)
            run = X;
        else if (kind == 1)
            wsum = X;
        else if (kind == 2)
            acc = X;
        else if constexpr (WB == 2) {
            if (kind == 3)
                t2 = X;
            else
                t3 = X;
        }
    }
    out[gt] = wsum;
}

// out[s*per_out + o] = sum of in[s*per_in + o*chunk .. +chunk), chunk = TL*E; one workgroup (TL lanes) per output.
// One addition site for the strided loads and the LDS tree alike (instruction cache, see msm_reduce_kernel); E >= 1.
template <class C>
__global__ void __launch_bounds__(256) msm_sum_kernel(const XYZZ<C>* __restrict__ in, XYZZ<C>* __restrict__ out, uint32_t per_in, uint32_t per_out,
                               uint32_t E) {
    constexpr uint32_t TL = tree_lanes<C>();
    __shared__ XYZZ<C> sh[TL];
    const uint32_t tid = threadIdx.x;
    const uint32_t s = blockIdx.x / per_out, o = blockIdx.x % per_out;
    const uint32_t chunk = TL * E;
    const uint32_t lo = o * chunk;
    XYZZ<C> acc;
    xyzz_set_inf(acc);
    const XYZZ<C>* src = in + (uint64_t)s * per_in;
    const uint32_t lim = per_in < lo + chunk ? per_in : lo + chunk;
    XYZZ<C> nxt;
    xyzz_set_inf(nxt);
    if (lo + tid < lim) nxt = src[lo + tid];
    uint32_t nlev = 0;
    for (uint32_t d = TL / 2; d > 0; d >>= 1) nlev++;
#pragma unroll 1
    for (uint32_t k = 0; k < E + nlev; k++) {
        XYZZ<C> b = nxt;
        bool on;
        if (k < E) {
            const uint32_t i = lo + k * TL + tid;
            on = i < lim;
            if (on && k + 1 < E && i + TL < lim) nxt = src[i + TL];   // prefetch: the load overlaps the addition below
        } else {
            const uint32_t d = TL >> (k - E + 1);
            on = tid < d;
            if (on) b = sh[tid + d];
        }
        if (on) xyzz_add(acc, b);
        if (k + 1 >= E) {
            if (on || k + 1 == E) sh[tid] = acc;
            __syncthreads();
        }
    }
    if (tid == 0) out[(uint64_t)s * per_out + o] = acc;
}

// ------------------------------------------------------------------------------------------
// Bucket reduction by row and column sums (the default; the slice kernels above stay behind ZK_MSM_FLAG_SLICE_REDUCE for A/B).
// The slice form spends 24 of its 40 steps per lane multiplying a slice total by its offset: 2.6 M additions for the 2^19
// buckets of a 2^20-point MSM, where the sum itself needs 2 per bucket.  Here a window's nbk = R x C buckets are a matrix,
// bucket b = hi C + lo (weight b + 1):
//     sum_b (b + 1) B_b  =  C * sum_hi hi * Row_hi  +  sum_lo (lo + 1) * Col_lo          Row_hi = sum_lo B, Col_lo = sum_hi B
// so every bucket is added exactly twice, with no multiplications, and what remains are two weighted sums over R and C
// (<= 256) points per window, which a workgroup does as a suffix scan followed by a tree (sum_k (k + 1) x_k = sum of all
// suffix sums).
//   msm_axis_partials_kernel   lane = K buckets of one row (stride C / K: neighbouring lanes read neighbouring buckets) or of
//                              one column -> a partial sum                                              K - 1 additions
//   msm_axis_fold_kernel       TW lanes per row / column: its partials -> Row_hi / Col_lo               log2 steps in LDS
//   msm_axis_weighted_kernel   workgroup per (window, axis, block of TL elements): W = sum (t + 1) x_t and S = sum x_t
//   host                       axis total = sum_u W_u + TL * sum_u u S_u ; window = 2^lc * rows + columns ; Horner
// Each kernel has ONE addition site (instruction cache: see msm_reduce_kernel).
// ------------------------------------------------------------------------------------------
struct MsmAxes {
    uint32_t nbk;        // buckets per window = rows * cols
    uint32_t nw;         // windows
    uint32_t log_cols;   // C = 2^log_cols, R = nbk / C  (R <= C)
    uint32_t k_row, k_col;   // buckets per lane in the partials kernel (powers of two, k_row <= C, k_col <= R)
    // derived
    uint32_t rows, cols, p_row, p_col;   // partials per row = C / k_row, per column = R / k_col
    uint32_t row_lanes, col_lanes;       // nw * rows * p_row, nw * cols * p_col
};

template <class C>
__global__ void __launch_bounds__(64) msm_axis_partials_kernel(const XYZZ<C>* __restrict__ buckets, XYZZ<C>* __restrict__ part, MsmAxes A) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    if (gt >= A.row_lanes + A.col_lanes) return;
    uint64_t addr;
    uint32_t stride, count, dst;
    if (gt < A.row_lanes) {            // (window, row, k): buckets hi C + k + j p_row
        const uint32_t k = gt % A.p_row, hi = (gt / A.p_row) % A.rows, w = gt / (A.p_row * A.rows);
        addr = (uint64_t)w * A.nbk + (uint64_t)hi * A.cols + k;
        stride = A.p_row;
        count = A.k_row;
        dst = gt;                      // [window][row][k]
    } else {                           // (window, hc, column): buckets (hc k_col + j) C + lo ; neighbouring lanes = neighbouring columns
        const uint32_t gl = gt - A.row_lanes;
        const uint32_t lo = gl % A.cols, hc = (gl / A.cols) % A.p_col, w = gl / (A.cols * A.p_col);
        addr = (uint64_t)w * A.nbk + (uint64_t)hc * A.k_col * A.cols + lo;
        stride = A.cols;
        count = A.k_col;
        dst = A.row_lanes + (w * A.cols + lo) * A.p_col + hc;   // [window][column][hc]
    }
    XYZZ<C> acc = buckets[addr];
    XYZZ<C> nxt;
    xyzz_set_inf(nxt);
    if (count > 1) nxt = buckets[addr + stride];
#pragma unroll 1
    for (uint32_t j = 1; j < count; j++) {
        const XYZZ<C> b = nxt;
        if (j + 1 < count) nxt = buckets[addr + (uint64_t)(j + 1) * stride];   // the load overlaps the addition
        xyzz_add(acc, b);
    }
    part[dst] = acc;
}

// elem[e] = sum of part[e P .. e P + P): the rows (P = p_row) then the columns (P = p_col) of every window.  TW = min(P, 16)
// lanes per element (lane tid = sub * per_wg + element); a lane adds P / TW partials (stride TW), then an LDS tree over the TW lanes.
template <class C>
__global__ void __launch_bounds__(256) msm_axis_fold_kernel(const XYZZ<C>* __restrict__ part, XYZZ<C>* __restrict__ elem, MsmAxes A,
                                                            uint32_t row_blocks) {
    constexpr uint32_t TL = tree_lanes<C>();
    __shared__ XYZZ<C> sh[TL];
    const uint32_t tid = threadIdx.x;
    const bool is_row = blockIdx.x < row_blocks;
    const uint32_t P = is_row ? A.p_row : A.p_col;
    const uint32_t nelem = is_row ? A.nw * A.rows : A.nw * A.cols;
    const uint32_t TW = P < 16 ? P : 16;                  // lanes per element (power of two)
    const uint32_t per_wg = TL / TW;
    // sub-major lane order: the lanes that are still adding at tree level d are tid < d * per_wg, so whole waves drop out of the
    // deeper levels (element-major order keeps one busy lane in every wave until the last level: 4.5 x the wave-additions)
    const uint32_t e = (is_row ? blockIdx.x : blockIdx.x - row_blocks) * per_wg + tid % per_wg;
    const uint32_t sub = tid / per_wg;
    const bool live = e < nelem;
    const XYZZ<C>* src = part + (is_row ? 0u : A.row_lanes) + (uint64_t)e * P;
    const uint32_t serial = P / TW;
    uint32_t nlev = 0;
    for (uint32_t d = TW >> 1; d > 0; d >>= 1) nlev++;
    XYZZ<C> acc, nxt;
    xyzz_set_inf(acc);
    xyzz_set_inf(nxt);
    if (live) nxt = src[sub];
#pragma unroll 1
    for (uint32_t k = 0; k < serial + nlev; k++) {
        XYZZ<C> b = nxt;
        bool on;
        if (k < serial) {
            on = live;
            if (on && k + 1 < serial) nxt = src[sub + (k + 1) * TW];
        } else {
            const uint32_t d = TW >> (k - serial + 1);
            on = sub < d;
            if (on) b = sh[tid + d * per_wg];
        }
        if (on) xyzz_add(acc, b);
        if (k + 1 >= serial) {
            __syncthreads();            // every read of this level is done
            sh[tid] = acc;
            __syncthreads();
        }
    }
    if (live && sub == 0) elem[(is_row ? 0u : A.nw * A.rows) + e] = acc;
}

// One workgroup per (window, axis, block u of TL elements).  x_t = element u TL + t + shift of the axis (shift = 1 for the rows:
// Row_hi has weight hi, so Row_0 drops out and x_t = Row_{t + 1}; 0 for the columns), infinity beyond the axis.
//   out[2 i]     = sum_t (t + 1) x_t         (suffix scan, then a tree over the suffix sums)
//   out[2 i + 1] = sum_t x_t                 (the first suffix sum)
// the partial sums go to the host in the kernel view's limb form: the host converts (partial_to_std) while it adds them up
template <class CB>
ZK_HD void partial_to_std(XYZZ<CB>& r, const XYZZ<CB>& p) {
    r = p;
}
template <class CB>
ZK_HD void partial_to_std(XYZZ<CB>& r, const XYZZ<C29<CB>>& p) {
    xyzz29_to_std<CB>(r, p);
}
template <class CB>
ZK_HD void partial_to_std(XYZZ<CB>& r, const XYZZ<C29x2<CB>>& p) {
    xyzz29_to_std<CB>(r, p);
}
// the curve a kernel view computes for (its caller-side limb form)
template <class CK>
struct ViewBase {
    using type = CK;
    static constexpr bool LAZY = false;
};
template <class C>
struct ViewBase<C29<C>> {
    using type = C;
    static constexpr bool LAZY = true;
};
template <class C>
struct ViewBase<C29x2<C>> {
    using type = C;
    static constexpr bool LAZY = true;
};
// ZK_MSM_FLAG_DEVICE_PARTIALS (diagnostic form): the stages of fe29_to_std of every base-field component of a lazy point, written
// out one by one -- norm | product by FROM29 | canonical | packed words -- so that the host can name the first stage whose device
// result differs from its own (a GPU-only wrong conversion on the BN254 curves)
template <class CK>
constexpr uint32_t partial_dbg_words() {
    if constexpr (ViewBase<CK>::LAZY) {
        using F = typename CK::Fq;
        return (uint32_t)(sizeof(XYZZ<CK>) / sizeof(Fe29<F>)) * (3u * F29<F>::L + (uint32_t)F::N);
    } else {
        return 0;
    }
}
template <class CK>
ZK_HD void partial_std_stages(const XYZZ<CK>& p, uint32_t* dbg) {
    if constexpr (ViewBase<CK>::LAZY) {
        using F = typename CK::Fq;
        constexpr int L = F29<F>::L, N = F::N;
        constexpr int NC = (int)(sizeof(XYZZ<CK>) / sizeof(Fe29<F>));
        const Fe29<F>* comp = reinterpret_cast<const Fe29<F>*>(&p);
        for (int j = 0; j < NC; j++) {
            Fe29<F> t, c;
            Fe<F> w;
            for (int i = 0; i < L; i++) c.v[i] = F29<F>::FROM29[i];
            fe29_norm(t, comp[j]);
            for (int i = 0; i < L; i++) dbg[i] = t.v[i];
            fe29_mul(t, t, c);
            for (int i = 0; i < L; i++) dbg[L + i] = t.v[i];
            fe29_canon(t, t);
            for (int i = 0; i < L; i++) dbg[2 * L + i] = t.v[i];
            fe29_pack(w, t);
            for (int i = 0; i < N; i++) dbg[3 * L + i] = w.v[i];
            dbg += 3 * L + N;
        }
    }
}
template <class C, bool STD = false>
__global__ void __launch_bounds__(256) msm_axis_weighted_kernel(const XYZZ<C>* __restrict__ elem, XYZZ<C>* __restrict__ out, MsmAxes A,
                                                                uint32_t row_blocks, uint32_t col_blocks,
                                                                XYZZ<typename ViewBase<C>::type>* __restrict__ out_std = nullptr,
                                                                uint32_t* __restrict__ dbg = nullptr) {
    constexpr uint32_t TL = tree_lanes<C>();
    __shared__ XYZZ<C> sh[TL];
    const uint32_t tid = threadIdx.x;
    const uint32_t per_w = row_blocks + col_blocks;
    const uint32_t w = blockIdx.x / per_w, bi = blockIdx.x % per_w;
    const bool is_row = bi < row_blocks;
    const uint32_t u = is_row ? bi : bi - row_blocks;
    const uint32_t len = is_row ? A.rows : A.cols;
    const XYZZ<C>* src = elem + (is_row ? (uint64_t)w * A.rows : (uint64_t)A.nw * A.rows + (uint64_t)w * A.cols);
    const uint32_t g = u * TL + tid + (is_row ? 1u : 0u);
    XYZZ<C> acc, total;
    xyzz_set_inf(acc);
    xyzz_set_inf(total);
    if (g < len) acc = src[g];
    uint32_t E = len - u * TL < TL ? len - u * TL : TL;    // elements of this block (a power of two)
    uint32_t logE = 0;
    while ((1u << logE) < E) logE++;
    sh[tid] = acc;
    __syncthreads();
#pragma unroll 1
    for (uint32_t k = 0; k < 2 * logE; k++) {
        uint32_t off;
        bool on;
        if (k < logE) {                 // suffix scan: x_t += x_{t + 2^k}
            off = 1u << k;
            on = tid + off < E;
        } else {                        // tree over the suffix sums
            off = E >> (k - logE + 1);
            on = tid < off;
        }
        if (k == logE) total = acc;     // lane 0: the plain sum
        XYZZ<C> b;
        xyzz_set_inf(b);
        if (on) b = sh[tid + off];
        __syncthreads();
        if (on) xyzz_add(acc, b);
        sh[tid] = acc;
        __syncthreads();
    }
    if (logE == 0) total = acc;
    if (tid == 0) {
        out[2 * (uint64_t)blockIdx.x] = acc;
        out[2 * (uint64_t)blockIdx.x + 1] = total;
        if constexpr (STD) {
            XYZZ<typename ViewBase<C>::type> s0, s1;
            partial_to_std<typename ViewBase<C>::type>(s0, acc);
            partial_to_std<typename ViewBase<C>::type>(s1, total);
            out_std[2 * (uint64_t)blockIdx.x] = s0;
            out_std[2 * (uint64_t)blockIdx.x + 1] = s1;
            if (dbg) {
                partial_std_stages<C>(acc, dbg + (2 * (uint64_t)blockIdx.x) * partial_dbg_words<C>());
                partial_std_stages<C>(total, dbg + (2 * (uint64_t)blockIdx.x + 1) * partial_dbg_words<C>());
            }
        }
    }
}

template <class C>
__device__ __forceinline__ void pack_base(Affine<C>& r, const Affine<C29<C>>& q) {
    fe29_pack(r.x, q.x);
    fe29_pack(r.y, q.y);
}
template <class C>
__device__ __forceinline__ void pack_base(Affine<C>& r, const Affine<C29x2<C>>& q) {
    fe29_pack(r.x.c0, q.x.c0);
    fe29_pack(r.x.c1, q.x.c1);
    fe29_pack(r.y.c0, q.y.c0);
    fe29_pack(r.y.c1, q.y.c1);
}
// bases as uploaded (x R mod p)  ->  what the lazy-limb bucket kernels gather: x R' mod p (R' = 2^261 / 2^392), strict limbs below
// 2p, packed back into 32-bit words (StoredBase above); (0, 0) stays (0, 0)
template <class C>
__global__ void __launch_bounds__(256) bases_to29_kernel(const Affine<C>* __restrict__ in, StoredAffine<F29View<C>>* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<C> p = in[i];
    Affine<F29View<C>> q;
    aff29_from_std(q, p);
    Affine<C> r;
    pack_base(r, q);
    out[i] = r;
}

// table[w n + i] = [2^(c w)] P_i in the stored form of the bucket kernels (packed lazy limbs), w < W: c doublings from the
// previous window's affine point, normalised again (one inversion per entry: a one-time cost per resident key)
template <class C>
__global__ void __launch_bounds__(64) bases_precompute_kernel(const Affine<C>* __restrict__ in, StoredAffine<F29View<C>>* __restrict__ table,
                                                              uint32_t n, uint32_t c, uint32_t W) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<C> p = in[i];
    for (uint32_t w = 0; w < W; w++) {
        if (w > 0 && !aff_is_inf(p)) {
            XYZZ<C> acc;
            xyzz_set_inf(acc);
            xyzz_add_mixed(acc, p);
            for (uint32_t k = 0; k < c; k++) xyzz_dbl(acc);
            xyzz_to_affine(p, acc);
        }
        Affine<F29View<C>> q;
        aff29_from_std(q, p);
        Affine<C> r;
        pack_base(r, q);
        table[(uint64_t)w * n + i] = r;
    }
}


// out[i] = [k_i] G in affine form (k canonical).  Used to build seeded test / bench bases
// (P_i = [k_i]G) and as the building block of fixed-base setup work.
template <class C>
__global__ void __launch_bounds__(64) fixed_base_mul_kernel(const Fe<typename C::Fr>* __restrict__ scalars, Affine<C>* __restrict__ out, uint32_t n) {
    using Fr = typename C::Fr;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<Fr> k = scalars[i];
    Affine<C> g;
    curve_generator(g);
    XYZZ<C> acc;
    xyzz_set_inf(acc);
    for (int bit = 32 * Fr::N - 1; bit >= 0; bit--) {
        xyzz_dbl(acc);
        if ((word_at<Fr::N>(k.v, bit >> 5) >> (bit & 31)) & 1) xyzz_add_mixed(acc, g);
    }
    Affine<C> r;
    xyzz_to_affine(r, acc);
    out[i] = r;
}

// halo2_proofs 0.2 poly/commitment/prover.rs parallel_generator_collapse (one IPA round): tmp[i] = g[i] + [u] g[i + half].
// The challenge u is the same for every lane, so the scalar multiplication has wave-uniform control flow; it runs on lazy
// limbs like the buckets.  On the Pasta curves (j = 0) u is split as k1 + k2 lambda with |k1|, |k2| < 2^129 (host side) and
// [u] P = [k1] P + [k2] phi(P), phi(x, y) = (beta x, y): one joint chain of ~129 doublings instead of 255, ~129 additions.
// xyzz_batch_to_affine_kernel then writes the folded generators back as affine points (batch_normalize upstream).
struct FoldScalar {
    uint32_t k1[8], k2[8];   // magnitudes (k2 = 0, k1 = u without GLV)
    int neg1, neg2;          // use -P / -phi(P)
    int top_bit;             // highest set bit of k1 | k2, -1 when both are zero
    int glv;
};
template <class C>
__global__ void __launch_bounds__(64) ipa_fold_bases_kernel(const Affine<C>* __restrict__ g, XYZZ<C>* __restrict__ tmp, uint32_t half, FoldScalar ks) {
    using CK = F29View<C>;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const Affine<C> hi_std = g[i + half], lo_std = g[i];
    Affine<CK> p1, p2, lo;
    aff29_from_std(p1, hi_std);
    aff29_from_std(lo, lo_std);
    aff_neg_if(p1, ks.neg1 != 0);
    p2 = p1;
    if constexpr (Glv<C>::HAS) {
        if (ks.glv) {
            Affine<C> q = hi_std;       // phi(P) = (beta x, y); the identity (0, 0) stays (0, 0)
            Coord<C> beta;
            fe_from_words(beta, Glv<C>::BETA);
            fe_mul(q.x, q.x, beta);
            aff29_from_std(p2, q);
            aff_neg_if(p2, ks.neg2 != 0);
        }
    }
    XYZZ<CK> acc;
    xyzz_set_inf(acc);
    for (int bit = ks.top_bit; bit >= 0; bit--) {
        xyzz_dbl(acc);
        if ((word_at<8>(ks.k1, bit >> 5) >> (bit & 31)) & 1) xyzz_add_mixed(acc, p1);
        if ((word_at<8>(ks.k2, bit >> 5) >> (bit & 31)) & 1) xyzz_add_mixed(acc, p2);
    }
    xyzz_add_mixed(acc, lo);
    XYZZ<C> r;
    xyzz29_to_std<C>(r, acc);
    tmp[i] = r;
}

// ------------------------------------------------------------------------------------------
// Several IPA rounds of generator folding at once.  After r rounds of the fold-free form (zk_ipa_virtual_scalars_device) the
// generators upstream would hold are
//     G'[i] = sum_{t < T} W_t G0[t m + i]        i < m = m0 / T, T = 2^r, W_t = the product of the challenges that t's bits select
// -- m multi-scalar multiplications over T points each that all use the SAME T scalars.  So the digit decomposition and the
// bucket lists of the T scalars are made once, on the host (T <= 2^12), and the device work is perfectly regular: lane
// (i, window) walks the window's buckets from the top, sums each bucket's points G0[t m + i] (neighbouring lanes read
// neighbouring points), and keeps the two running sums of the bucket method; a second kernel does the Horner over the
// windows of every output.  ~(255 / c) additions per original generator instead of the ~260 point operations per survivor
// of every literal fold, and no latency-bound small rounds in between.
// ------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(256) ipa_gather_weights_kernel(const Fe<typename C::Fr>* __restrict__ w, uint64_t stride, uint32_t T,
                                                                 Fe<typename C::Fr>* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    Fe<typename C::Fr> x = w[(uint64_t)t * stride];
    fe_from_mont(x, x);
    out[t] = x;
}

// list_off[w nbk + b] .. list_off[w nbk + b + 1]: the entries (t | sign << 31) whose digit in window w has magnitude b + 1
template <class CK>
__global__ void __launch_bounds__(64) ipa_collapse_window_kernel(const StoredAffine<CK>* __restrict__ bases, const uint32_t* __restrict__ list_off,
                                                                 const uint32_t* __restrict__ list_ent, XYZZ<CK>* __restrict__ out, uint32_t m,
                                                                 uint32_t i0, uint32_t cnt, uint32_t nbk, uint32_t nwin) {
    // outputs i0 .. i0 + cnt of the m survivors (a rank's share, or all of them); out is [window][cnt]
    const uint64_t gt = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t il = (uint32_t)(gt % cnt), w = (uint32_t)(gt / cnt);
    const uint32_t i = i0 + il;
    if (w >= nwin) return;
    XYZZ<CK> run, acc;
    xyzz_set_inf(run);
    xyzz_set_inf(acc);
    const uint32_t* off = list_off + (uint64_t)w * nbk;
#pragma unroll 1
    for (uint32_t b = nbk; b-- > 0;) {
        XYZZ<CK> bsum;
        xyzz_set_inf(bsum);
        const uint32_t e1 = off[b + 1];
#pragma unroll 1
        for (uint32_t k = off[b]; k < e1; k++) {
            const uint32_t e = list_ent[k];
            Affine<CK> p;
            load_base(p, bases, (e & 0x7fffffffu) * m + i);
            aff_neg_if(p, (e >> 31) != 0);
            xyzz_add_mixed(bsum, p);
        }
#pragma unroll 1
        for (int s = 0; s < 2; s++) {          // run += bsum ; acc += run   (one addition site)
            XYZZ<CK> X = s ? acc : run;
            const XYZZ<CK> Y = s ? run : bsum;
            xyzz_add(X, Y);
            if (s)
                acc = X;
            else
                run = X;
        }
    }
    out[(uint64_t)w * cnt + il] = acc;
}

// out[i] = sum_w 2^(c w) part[w m + i]  (Horner from the top window), as a saturated-limb XYZZ point for the batched normalisation
template <class C>
__global__ void __launch_bounds__(64) ipa_collapse_horner_kernel(const XYZZ<F29View<C>>* __restrict__ part, XYZZ<C>* __restrict__ out, uint32_t m,
                                                                 uint32_t c, uint32_t nwin) {
    using CK = F29View<C>;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    XYZZ<CK> acc = part[(uint64_t)(nwin - 1) * m + i];
#pragma unroll 1
    for (uint32_t w = nwin - 1; w-- > 0;) {
#pragma unroll 1
        for (uint32_t k = 0; k < c; k++) xyzz_dbl(acc);
        const XYZZ<CK> p = part[(uint64_t)w * m + i];
        xyzz_add(acc, p);
    }
    XYZZ<C> r;
    xyzz29_to_std<C>(r, acc);
    out[i] = r;
}

// ------------------------------------------------------------------------------------------
// Fixed-base windowed scalar multiplication: out[i] = [k_i] B for ONE base B -- ark-ec 0.3 `FixedBaseMSM::get_window_table`
// + `FixedBaseMSM::multi_scalar_mul` + `ProjectiveCurve::batch_normalization_into_affine`, the shape of Groth16 key
// generation (ark-groth16 0.3 generate_parameters: every query vector of the proving key is [s_i] g for one random
// generator g; reached from the reference at lib/src/zk/encryption.rs:169).
//   table[w * 255 + d - 1] = [d * 2^(8w)] B   (affine, FB_C = 8-bit unsigned windows: 32 x 255 points, 0.5 MB, L2-resident)
//   fixed_base_msm_kernel   one lane per scalar: <= 32 mixed additions of table entries, no doublings
//   xyzz_batch_to_affine_kernel   K = 8 points per lane share one field inversion (Montgomery's trick)
// ------------------------------------------------------------------------------------------
constexpr int FB_C = 8;
constexpr uint32_t FB_ROW = (1u << FB_C) - 1;   // table entries per window
constexpr int FB_K = 8;                         // points per inversion in the normalisation kernel

template <class C>
constexpr int fb_windows() {
    return (C::Fr::BITS + FB_C - 1) / FB_C;
}

template <class C>
__global__ void __launch_bounds__(64) fixed_base_table_kernel(Affine<C> base, Affine<C>* __restrict__ table, uint32_t entries) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= entries) return;
    const uint32_t w = t / FB_ROW, d = t % FB_ROW + 1;
    XYZZ<C> acc;
    xyzz_set_inf(acc);
    for (int bit = FB_C - 1; bit >= 0; bit--) {   // [d] B
        xyzz_dbl(acc);
        if ((d >> bit) & 1) xyzz_add_mixed(acc, base);
    }
    for (uint32_t k = 0; k < w * FB_C; k++) xyzz_dbl(acc);   // ... * 2^(8w)
    Affine<C> r;
    xyzz_to_affine(r, acc);
    table[t] = r;
}

template <class C>
__global__ void __launch_bounds__(64) fixed_base_msm_kernel(const Affine<C>* __restrict__ table, const Fe<typename C::Fr>* __restrict__ scalars,
                                                            XYZZ<C>* __restrict__ out, uint32_t n, int mont) {
    using Fr = typename C::Fr;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<Fr> k = scalars[i];
    if (mont) fe_from_mont(k, k);
    XYZZ<C> acc;
    xyzz_set_inf(acc);
    for (int w = 0; w < fb_windows<C>(); w++) {
        const uint32_t d = bits_at<Fr::N>(k.v, w * FB_C, FB_C);
        if (d != 0) {
            Affine<C> p = table[(uint32_t)w * FB_ROW + d - 1];
            xyzz_add_mixed(acc, p);
        }
    }
    out[i] = acc;
}

// out[i] = affine(in[i]); identity -> (0, 0).  Lane t owns points [FB_K t, FB_K (t + 1)): prefix products of their ZZZ,
// one inversion, back-substitution.  1/ZZ = ZZ^2 / ZZZ^2 (ZZ^3 = ZZZ^2 is the XYZZ invariant).
template <class C>
__global__ void __launch_bounds__(64) xyzz_batch_to_affine_kernel(const XYZZ<C>* __restrict__ in, Affine<C>* __restrict__ out, uint32_t n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lo = t * FB_K;
    if (lo >= n) return;
    const uint32_t cnt = n - lo < (uint32_t)FB_K ? n - lo : (uint32_t)FB_K;
    Coord<C> pre[FB_K], run, z;
    fe_one(run);
    for (uint32_t k = 0; k < cnt; k++) {
        pre[k] = run;                                  // product of the ZZZ before point k
        z = in[lo + k].zzz;
        if (fe_is_zero(in[lo + k].zz)) fe_one(z);      // the identity contributes a factor 1
        fe_mul(run, run, z);
    }
    Coord<C> inv;
    fe_inv(inv, run);
    for (int k = (int)cnt - 1; k >= 0; k--) {
        const XYZZ<C> p = in[lo + k];
        Affine<C> r;
        if (xyzz_is_inf(p)) {
            fe_zero(r.x);
            fe_zero(r.y);
        } else {
            Coord<C> izzz, t2, izz;
            fe_mul(izzz, inv, pre[k]);                 // 1 / ZZZ_k
            fe_mul(inv, inv, p.zzz);                   // drop ZZZ_k from the running inverse
            fe_mul(t2, izzz, p.zz);                    // ZZ / ZZZ = 1 / z
            fe_sqr(izz, t2);                           // 1 / ZZ
            fe_mul(r.x, p.x, izz);
            fe_mul(r.y, p.y, izzz);
        }
        out[lo + k] = r;
    }
}

}  // namespace zk
