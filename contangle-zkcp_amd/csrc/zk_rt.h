// Thin runtime seam: the same kernel and launch-sequence sources build (a) with hipcc for
// gfx950 -- the product -- and (b) with g++ against tests/emu/emu_hip.h, a HIP-semantics
// emulator (threads as fibers, LDS as static storage) that exists ONLY so that the CPU test
// tier (`pytest -m "not gpu"`) and CPU sanitizers can exercise kernel indexing and host
// logic without a GPU.  The product library is always build (a); it has no CPU path.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(ZK_EMU)
#include "emu_hip.h"
#else
#include <hip/hip_runtime.h>
#define ZK_DYN_SHARED(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#define ZK_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (shmem), (stream), __VA_ARGS__)
#define ZK_UNIFORM32(v) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(v)))   // a value the whole wave agrees on, as a scalar
#endif
