// Shared device/host helpers for the dotaclient PPO hot-path library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DC_WAVE 64
// Developer builds only (python -m dotaclient_amd.build with DC_BUILD_VARIANT=timing DC_BUILD_FLAGS=-DDC_DEV_TIMING=1): the
// phase-cycle instrumentation of the persistent kernels (extra kernel instantiations, a blocking read-back and a line on
// stderr per launch).  The shipped library compiles it out; nothing on the call path reads the environment.
#ifndef DC_DEV_TIMING
#define DC_DEV_TIMING 0
#endif
// 1: clear buffers with hipMemsetAsync / hipMemset2DAsync / hipMemcpyAsync instead of this library's own kernels (fill.hip) -
// the round-2 behaviour, kept as an A/B build for the hipGraph replay fault (profiles/r03/crash_bisect.md)
#ifndef DC_HIP_MEMSET
#define DC_HIP_MEMSET 0
#endif
#ifndef DC_DEV_HOOKMODE
#define DC_DEV_HOOKMODE 0
#endif

namespace dc {

// Records the first failing HIP call; returned as the int result of every C-ABI entry point.
void set_error(const char* what, int code);
int launch_check(const char* what);

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

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
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// max / relu that PROPAGATE NaN (v_maximum3_f32, IEEE-754-2019 maximum; fmaxf = maxNum returns the other operand).  torch.max and
// torch.relu propagate NaN too (policy.py:97-138), and the f16x2 products rely on it: an operand beyond f16's range becomes inf, its
// products NaN, and that NaN has to REACH the loss (the reference's own NaN guard, optimizer.py:667-669) - a relu or a max-pool that
// swallows it would turn an out-of-range input into a finite, wrong step (tests/test_gpu_parity.py::..._at_the_range_edge).
__device__ __forceinline__ float max_nan(float a, float b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ float relu_nan(float a) { return __builtin_elementwise_maximum(a, 0.f); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace dc
