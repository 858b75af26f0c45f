// TEST INFRASTRUCTURE -- a minimal HIP-semantics emulator so that `pytest -m "not gpu"` (no GPU
// in the build container) and CPU sanitizers can execute the kernel sources of
// contangle-zkcp_amd/csrc unchanged: each workgroup runs as a set of cooperative fibers
// (ucontext), `__syncthreads()` is a counting fiber barrier, `__ballot` / `__shfl` are per-wave rendezvous,
// `__shared__` is static storage, atomics are plain read-modify-writes (fibers never run concurrently).
//
// It is NOT a backend: nothing in the product loads tests/emu/libzkcp_emu.so, bench.py and
// __graft_entry__.smoke() never touch it, and the product library fails with ZK_ERR_NO_DEVICE
// when no MI355X is present.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

namespace emu {
struct Dim3 {
    unsigned x, y, z;
};
struct ThreadCtx {
    Dim3 tid, bid;
};
extern ThreadCtx* cur;
extern Dim3 g_block_dim, g_grid_dim;
extern char* g_dyn_smem;
void launch(unsigned grid, unsigned block, size_t shmem, const std::function<void()>& body);
void syncthreads();
int syncthreads_count(int pred);
// wave-level rendezvous: every lane of the calling lane's 64-wide wave (the lanes of the workgroup's last wave if it is
// short) contributes one word; returns the wave's 64 words (valid until the wave's next exchange).  Like the hardware
// cross-lane operations it requires all those lanes to reach the call.
const uint32_t* wave_exchange(uint32_t v);
}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::cur->bid)
#define blockDim (emu::g_block_dim)
#define gridDim (emu::g_grid_dim)
#define __syncthreads() emu::syncthreads()
#define __syncthreads_count(p) emu::syncthreads_count((p) ? 1 : 0)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

#define ZK_DYN_SHARED(type, name) type* name = reinterpret_cast<type*>(emu::g_dyn_smem)
#define ZK_EMU_STRIP_PARENS(...) __VA_ARGS__
#define ZK_UNIFORM32(v) ((uint32_t)(v))
#define ZK_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    emu::launch((unsigned)(grid), (unsigned)(block), (size_t)(shmem), [&]() { (ZK_EMU_STRIP_PARENS kernel)(__VA_ARGS__); })

struct alignas(16) uint4 {
    uint32_t x, y, z, w;
};

template <class T>
static inline T atomicAdd(T* p, T v) {
    T o = *p;
    *p = o + v;
    return o;
}
static inline uint32_t __brev(uint32_t x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    return __builtin_bswap32(x);
}
static inline int __clz(uint32_t x) { return x ? __builtin_clz(x) : 32; }
static inline int __popc(uint32_t x) { return __builtin_popcount(x); }
static inline int __popcll(uint64_t x) { return __builtin_popcountll(x); }
static inline int __ffsll(uint64_t x) { return __builtin_ffsll((long long)x); }
static inline uint64_t __ballot(int pred) {
    const uint32_t* w = emu::wave_exchange(pred ? 1u : 0u);
    uint64_t m = 0;
    for (int i = 0; i < 64; i++) m |= (uint64_t)(w[i] & 1u) << i;
    return m;
}
static inline uint32_t __shfl(uint32_t v, int src_lane) { return emu::wave_exchange(v)[src_lane & 63]; }

// ---- host runtime shims (device memory == host memory) ----
typedef int hipError_t;
typedef void* hipStream_t;
typedef struct emuEvent* hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    int multiProcessorCount;
};
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipStreamSynchronize(hipStream_t st);
enum { hipStreamNonBlocking = 1 };
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t st);
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t e, unsigned flags);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipEventCreate(hipEvent_t* e);
enum { hipEventDisableTiming = 2 };
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
#define hipFuncAttributeMaxDynamicSharedMemorySize 0
template <class K>
static inline hipError_t hipFuncSetAttribute(K, int, int) {
    return hipSuccess;
}
