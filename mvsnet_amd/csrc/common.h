// Shared helpers for the gfx950 kernels of libmvsnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mvsnet_hip.h"

#define MVS_WAVE 64

#define MVS_CHECK_ARG(cond) do { if (!(cond)) return MVS_E_BADARG; } while (0)
#define MVS_LAUNCH_RET() do { hipError_t e__ = hipGetLastError(); return (int)e__; } while (0)

static inline hipStream_t mvs_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int mvs_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Sum over the 64 lanes of a wave (all lanes receive the total).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float relu(float v) { return v > 0.f ? v : 0.f; }

// XCD-aware block remap (MI355X: 8 XCDs with private 4 MiB L2s, workgroups dealt round-robin over
// them, so blocks b and b+8 share an L2).  Gives each XCD a contiguous run of the index space so that
// neighbouring tiles (shared halos, overlapping warp footprints) hit the same L2.  Speed only; the
// identity is used when the count is not a multiple of 8.
__device__ __forceinline__ int xcd_swizzle(int bid, int nb) {
    return (nb & 7) ? bid : (bid & 7) * (nb >> 3) + (bid >> 3);
}
