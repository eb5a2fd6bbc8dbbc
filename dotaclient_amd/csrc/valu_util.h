// Lane-level building blocks of the register-resident recurrent kernels (rnn_persist_valu.hip, rnn_team.hip):
// packed-f32 FMAs with an op_sel broadcast, DPP / permlane-swap reduce-scatter steps, hardware tanh.
#pragma once
#include "common.h"

namespace dc {
namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int DPP_ROR8 = 0x128, DPP_HALF_MIRROR = 0x141, DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E;
constexpr int DPP_Q0 = 0x00, DPP_Q1 = 0x55, DPP_Q2 = 0xAA, DPP_Q3 = 0xFF;

// acc (+)= w * v[E] (both halves of the pair w times ONE element of v: the broadcast is the instruction's
// op_sel).  Written as asm because hipcc folds the splat for three of the four elements of a b128 LDS read
// and copies the fourth into a pair whose other half may be an in-flight global load (-> s_waitcnt vmcnt(0)
// in the middle of the step).
template <int E>
__device__ __forceinline__ void pk_fma_bcast(f32x2& acc, f32x2 w, f32x2 v) {
    if constexpr (E == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(v));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(w), "v"(v));
}
__device__ __forceinline__ f32x2 pk_mul_bcast0(f32x2 w, f32x2 v) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(w), "v"(v));
    return r;
}
template <int CTRL>
__device__ __forceinline__ float dpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float tanh_hw(float x) {   // v_exp_f32 / v_rcp_f32, ~1 ulp each (see rnn_persist.hip)
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.0f;
}
// lanes 32..63 of a <-> lanes 0..31 of b; the sum is then, in the low half, a(l) + a(l+32) and, in the high
// half, b(l-32) + b(l): low lanes keep the "a" output, high lanes the "b" output.
__device__ __forceinline__ float swap32_sum(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// odd 16-lane rows of a <-> even rows of b: even rows keep a(l) + a(l+16), odd rows b(l-16) + b(l)
__device__ __forceinline__ float swap16_sum(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// forward reduce-scatter: local column (0..15 of the row's 16) kept in register r of row lane l.
// Stage s adds the partner's register c + 2^s to register c, so the partner's upper registers must hold
// this lane's lower columns: register r of lane l = register (r minus its top bit) of that stage's partner.
__device__ __forceinline__ int half_mirror16(int l) { return (l & 8) | (7 - (l & 7)); }
__device__ __forceinline__ int colmap(int l, int r) {
    if (r & 8) l ^= 8;
    if (r & 4) l = half_mirror16(l);
    if (r & 2) l ^= 2;
    if (r & 1) l ^= 1;
    return l;
}

}  // namespace
}  // namespace dc
