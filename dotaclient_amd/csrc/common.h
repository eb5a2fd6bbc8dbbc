// Shared device/host helpers for the dotaclient PPO hot-path library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DC_WAVE 64

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

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace dc
