// Shared helpers for the gfx950 kernels of libmvsnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mvsnet_hip.h"

#define MVS_WAVE 64

#define MVS_CHECK_ARG(cond) do { if (!(cond)) return MVS_E_BADARG; } while (0)
#define MVS_LAUNCH_RET() do { hipError_t e__ = hipGetLastError(); return (int)e__; } while (0)

// test / measurement hooks (mvs_set_test_hook, include/mvsnet_hip.h; defined in regnet.hip).  No getenv anywhere in the library.
#include <atomic>
extern std::atomic<int> mvs_hooks[MVS_HOOK_COUNT];
static inline int mvs_hook(int id) { return mvs_hooks[id].load(std::memory_order_relaxed); }

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

// tf.sigmoid / tf.tanh (convgru.py:101-102,117) on the fast transcendental path (v_exp_f32 + v_rcp_f32, ~2 ulp), tanh in the
// form that does not cancel near 0: t = e^{-2|x|}, (1 - t) / (1 + t) with the sign of x.  Round 3 measured them against libm
// expf + IEEE divides in cell 1 of the recurrent sweep: the same distance from the float64 fixture (1.169e-3 vs 1.172e-3 worst
// probability, plane agreement 0.99988 both) -- the distance is float32 summation order, not these forms.  Round 4: cells 2 / 3
// and the blend kernel use them too (libm expf / tanhf / IEEE division were ~45 % of the small cells' vector instructions).
__device__ __forceinline__ float mvs_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float mvs_tanh_fast(float x) {
    const float t = __expf(-2.0f * fabsf(x));
    return copysignf((1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t), x);
}

// XCD-aware block remap (MI355X: 8 XCDs with private 4 MiB L2s, workgroups dealt round-robin over
// them, so blocks b and b+8 share an L2).  Gives each XCD a contiguous run of the index space so that
// neighbouring tiles (shared halos, overlapping warp footprints) hit the same L2.  Speed only; the
// identity is used when the count is not a multiple of 8.
__device__ __forceinline__ int xcd_swizzle(int bid, int nb) {
    return (nb & 7) ? bid : (bid & 7) * (nb >> 3) + (bid >> 3);
}
