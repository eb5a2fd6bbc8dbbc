// MFMA building blocks of the register-resident recurrent kernels (rnn_persist.hip, rnn_team_mfma.hip): the
// v_mfma_f32_4x4x1_16b_f32 product phase with block-broadcast A operands and AGPR-resident weights, hand-issued with
// memory "hooks" between the MFMA pairs; the gate non-linearities on the hardware transcendentals; slot mapping.
#pragma once
#include <utility>
#include "kernels.h"

namespace dc {

template <int H>
struct PersistCfg {
    static constexpr int WAVES = H / 32;
    static constexpr int THREADS = WAVES * 64;
    static constexpr int HLD = H + 4;          // h rows in LDS: +4 floats -> the 4 rows of a b128 read hit disjoint banks
    static constexpr int GLD = 4 * H + 8;      // gate-gradient rows in LDS
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// Gate non-linearities on the hardware transcendentals (v_exp_f32 / v_rcp_f32, ~1 ulp each): the
// cell epilogue sits on the per-step critical path, and libm's range-reduced expf + IEEE division +
// branchy tanhf cost several hundred dependent cycles there.  Absolute error ~1e-7, far inside the
// 1e-4 parity bar (tests/test_gpu_parity.py).
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    // 2*sigmoid(2x) - 1; exp2 overflow to +inf gives rcp(inf) = 0 -> -1, underflow -> +1
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.0f;
}


// ---------------------------------------------------------------------------------------------------
// Hand-issued product phase.
//
// A operand by BROADCAST: all 16 blocks of the 4x4x1 MFMA need the same four rows (sequences), so the
// instruction's block broadcast (cbsz:4 abid:b = "every block takes its A from block b", verified on
// gfx950 by tools/ubench/mfma_bcast.hip) lets ONE register carry 16 different k: lane 4b+i of register
// j holds state[i][16j+b], and the 16 MFMAs abid = 0..15 walk k = 16j .. 16j+15.  The whole A operand
// of a step is then H/16 (forward) or 2H/16 (backward) registers = 2 resp. 4 ds_read_b128 per lane
// (state kept in LDS in that permuted order) instead of one read per 4 k.
// B operand straight from AGPRs ("a" constraint: MFMA A/B operands may be AGPRs on gfx950) - hipcc
// itself copies AGPR-resident weights back to VGPRs with v_accvgpr_read + hazard nops per MFMA.
//
// Issue budget (tools/ubench/mfma4x4.hip): the MFMA issues every 8.55 cycles whatever the register
// files and chain count, and with one wave per SIMD every OTHER instruction placed between two MFMAs
// costs ~5 cycles of its own - nothing hides behind a 2-pass MFMA.  The product phase therefore carries
// the bare minimum: the 2-4 LDS reads, and one memory instruction per hook for next step's operands and
// the previous step's results (32-bit byte offsets from uniform bases, no branches).
// Every statement is asm volatile: issue order = program order, and the compiler emits no LDS/SMEM
// operation of its own between the reads and their waits.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

template <int OFF>
__device__ __forceinline__ void lds_read16(float4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N));
}
// c0 (+)= bcast_b(a) * w0 ; c1 (+)= bcast_b(a) * w1.  ZERO: start from the inline constant 0.
template <bool ZERO, int ABID>
__device__ __forceinline__ void mfma_pair_same(f32x4& c0, f32x4& c1, float a, float w0, float w1) {
    if constexpr (ZERO) {
        asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, 0 cbsz:4 abid:%5\n\tv_mfma_f32_4x4x1_16b_f32 %1, %2, %4, 0 cbsz:4 abid:%5"
                     : "=&v"(c0), "=&v"(c1)
                     : "v"(a), "a"(w0), "a"(w1), "i"(ABID));
    } else {
        asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, %0 cbsz:4 abid:%5\n\tv_mfma_f32_4x4x1_16b_f32 %1, %2, %4, %1 cbsz:4 abid:%5"
                     : "+v"(c0), "+v"(c1)
                     : "v"(a), "a"(w0), "a"(w1), "i"(ABID));
    }
}
// c0 (+)= bcast_b0(a) * w0 ; c1 (+)= bcast_b1(a) * w1   (two consecutive k of one column)
// cbsz:3 = broadcast inside each group of 8 blocks (= each wave half): the halves keep different A
template <bool ZERO, int ABID0, int ABID1>
__device__ __forceinline__ void mfma_pair_seq(f32x4& c0, f32x4& c1, float a, float w0, float w1) {
    if constexpr (ZERO) {
        asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, 0 cbsz:3 abid:%5\n\tv_mfma_f32_4x4x1_16b_f32 %1, %2, %4, 0 cbsz:3 abid:%6"
                     : "=&v"(c0), "=&v"(c1)
                     : "v"(a), "a"(w0), "a"(w1), "i"(ABID0), "i"(ABID1));
    } else {
        asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, %0 cbsz:3 abid:%5\n\tv_mfma_f32_4x4x1_16b_f32 %1, %2, %4, %1 cbsz:3 abid:%6"
                     : "+v"(c0), "+v"(c1)
                     : "v"(a), "a"(w0), "a"(w1), "i"(ABID0), "i"(ABID1));
    }
}
// c0 (+)= bcast_b0(a0) * w0 ; c1 (+)= bcast_b1(a1) * w1   (two different k of ONE column, whole-wave broadcast: cbsz:4)
template <bool ZERO, int ABID0, int ABID1>
__device__ __forceinline__ void mfma_pair_k(f32x4& c0, f32x4& c1, float a0, float a1, float w0, float w1) {
    if constexpr (ZERO) {
        asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %4, 0 cbsz:4 abid:%6\n\tv_mfma_f32_4x4x1_16b_f32 %1, %3, %5, 0 cbsz:4 abid:%7"
                     : "=&v"(c0), "=&v"(c1)
                     : "v"(a0), "v"(a1), "a"(w0), "a"(w1), "i"(ABID0), "i"(ABID1));
    } else {
        asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %4, %0 cbsz:4 abid:%6\n\tv_mfma_f32_4x4x1_16b_f32 %1, %3, %5, %1 cbsz:4 abid:%7"
                     : "+v"(c0), "+v"(c1)
                     : "v"(a0), "v"(a1), "a"(w0), "a"(w1), "i"(ABID0), "i"(ABID1));
    }
}
// MFMA -> VALU read hazard, padded by hand (nothing after an asm is padded by the compiler)
__device__ __forceinline__ void mfma_tail_pad(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// lanes 32..63 of `lo` <-> lanes 0..31 of `hi_` (v_permlane32_swap): afterwards
//   lo  = [lo.low  | hi_.low ]      hi_ = [lo.high | hi_.high]
__device__ __forceinline__ void half_swap(float& lo, float& hi_) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi_), false, false);
    lo = __uint_as_float(r[0]);
    hi_ = __uint_as_float(r[1]);
}

// lanes 16..31 / 48..63 of `lo` <-> lanes 0..15 / 32..47 of `hi_` (v_permlane16_swap: inside each half of the wave, the upper
// sixteen lanes of the first operand change places with the lower sixteen of the second)
__device__ __forceinline__ void row_swap(float& lo, float& hi_) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi_), false, false);
    lo = __uint_as_float(r[0]);
    hi_ = __uint_as_float(r[1]);
}
// 4 x 4 transpose between the four 16-lane rows of a wave and four registers: afterwards row g' holds in a[s] what row s held in
// a[g'] (two v_permlane32_swap for the distance-two pairs, two v_permlane16_swap for the neighbours)
__device__ __forceinline__ void rows_transpose4(float (&a)[4]) {
    half_swap(a[0], a[2]);
    half_swap(a[1], a[3]);
    row_swap(a[0], a[1]);
    row_swap(a[2], a[3]);
}

// the value of lane l ^ 8 (DPP row_ror:8: a rotation by eight inside each 16-lane row swaps its halves)
__device__ __forceinline__ float lane_xor8(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, true));
}
// 4 x 4 transpose between the four 8-lane groups of each 32-lane half of a wave (group g = (lane >> 3) & 3) and four registers:
// afterwards group g' holds in a[s] what group s held in a[g'] (two v_permlane16_swap for the groups sixteen lanes apart, DPP rotations
// + selects for the neighbours)
__device__ __forceinline__ void groups8_transpose4(float (&a)[4], int lane) {
    row_swap(a[0], a[2]);
    row_swap(a[1], a[3]);
    const bool odd = (lane & 8) != 0;
    const float p0 = lane_xor8(a[0]), p1 = lane_xor8(a[1]), p2 = lane_xor8(a[2]), p3 = lane_xor8(a[3]);
    const float n0 = odd ? p1 : a[0], n1 = odd ? a[1] : p0, n2 = odd ? p3 : a[2], n3 = odd ? a[3] : p2;
    a[0] = n0; a[1] = n1; a[2] = n2; a[3] = n3;
}

// register j of the A operand = component (j & 3) of LDS read (j >> 2).  Components are named at the use
// site (a sub-register reference, no instruction): copying them earlier would read the destination of a
// ds_read the compiler does not know is still in flight.
template <int J, int N>
__device__ __forceinline__ const float& a_reg(const float4 (&r)[N]) {
    if constexpr ((J & 3) == 0) return r[J >> 2].x;
    else if constexpr ((J & 3) == 1) return r[J >> 2].y;
    else if constexpr ((J & 3) == 2) return r[J >> 2].z;
    else return r[J >> 2].w;
}

// position of state[seq][k] inside a broadcast-ordered LDS row set: lane 4b+i reads the NJ floats
// [i][b][0..NJ) contiguously; k = 16j + b.  ROW = floats per sequence row (>= 16*NJ, padded).
template <int NJ, int ROW>
__device__ __forceinline__ int bcast_pos(int seq, int k) { return seq * ROW + (k & 15) * NJ + (k >> 4); }

template <int H>
struct FwdProduct {
    static constexpr int NJ = H / 16;        // A registers per step
    static constexpr int HOOKS = H / 2;      // one hook per MFMA quad (2 k x 2 columns)
    template <int K2, class Hook>            // k = 2*K2, 2*K2+1
    static __device__ __forceinline__ void quad(const float4 (&r)[NJ / 4], f32x4 (&acc)[4], const float (&w0)[H],
                                                const float (&w1)[H], Hook& hook) {
        constexpr int k = 2 * K2;
        if constexpr (k == 0) wait_lgkm<NJ / 4 - 1>();
        if constexpr (NJ == 8 && k == 64) wait_lgkm<0>();
        mfma_pair_same<k == 0, k & 15>(acc[0], acc[1], a_reg<(k >> 4)>(r), w0[k], w1[k]);
        mfma_pair_same<k == 0, (k + 1) & 15>(acc[2], acc[3], a_reg<((k + 1) >> 4)>(r), w0[k + 1], w1[k + 1]);
        hook(std::integral_constant<int, K2>{});
    }
    template <class Hook, int... Ks>
    static __device__ __forceinline__ void quads(const float4 (&r)[NJ / 4], f32x4 (&acc)[4], const float (&w0)[H],
                                                 const float (&w1)[H], Hook& hook, std::integer_sequence<int, Ks...>) {
        (quad<Ks>(r, acc, w0, w1, hook), ...);
    }
    // acc: chains [0] col0 even k, [1] col1 even k, [2] col0 odd k, [3] col1 odd k
    template <class Hook>
    static __device__ __forceinline__ void run(f32x4 (&acc)[4], const float (&w0)[H], const float (&w1)[H], uint32_t addr,
                                               Hook& hook) {
        float4 r[NJ / 4];
        lds_read16<0>(r[0], addr);
        if constexpr (NJ == 8) lds_read16<16>(r[1], addr);
        quads(r, acc, w0, w1, hook, std::make_integer_sequence<int, H / 2>{});
        mfma_tail_pad(acc[0], acc[1], acc[2], acc[3]);
    }
};

// One gate column per lane over the WHOLE K (no k split across waves, hence no cross-wave sum): lane 4b + i reads the K / 16
// consecutive floats [i][(K / 16) b ..] of the state image (plain order); MFMA number kk' (abid = kk' & 15 of A register kk' >> 4)
// contracts k = (K / 16) (kk' & 15) + (kk' >> 4) - the weights are loaded in that order.  Four accumulator chains.
template <int K>
struct FwdProductCol {
    static constexpr int NJ = K / 16;        // A registers per step
    static constexpr int NR = NJ / 4;        // ds_read_b128 per step
    static constexpr int HOOKS = K / 4;      // one hook per MFMA quad (4 consecutive k')
    template <int K4, class Hook>
    static __device__ __forceinline__ void quad(const float4 (&r)[NR], f32x4 (&acc)[4], const float (&w)[K], Hook& hook) {
        constexpr int k = 4 * K4;
        if constexpr (k % 64 == 0) wait_lgkm<NR - 1 - k / 64>();
        mfma_pair_k<k == 0, k & 15, (k + 1) & 15>(acc[0], acc[1], a_reg<(k >> 4)>(r), a_reg<((k + 1) >> 4)>(r), w[k], w[k + 1]);
        mfma_pair_k<k == 0, (k + 2) & 15, (k + 3) & 15>(acc[2], acc[3], a_reg<((k + 2) >> 4)>(r), a_reg<((k + 3) >> 4)>(r), w[k + 2], w[k + 3]);
        hook(std::integral_constant<int, K4>{});
    }
    template <class Hook, int... Ks>
    static __device__ __forceinline__ void quads(const float4 (&r)[NR], f32x4 (&acc)[4], const float (&w)[K], Hook& hook,
                                                 std::integer_sequence<int, Ks...>) {
        (quad<Ks>(r, acc, w, hook), ...);
    }
    template <int... Rs>
    static __device__ __forceinline__ void reads(float4 (&r)[NR], uint32_t addr, std::integer_sequence<int, Rs...>) {
        (lds_read16<16 * Rs>(r[Rs], addr), ...);
    }
    template <class Hook>
    static __device__ __forceinline__ void run(f32x4 (&acc)[4], const float (&w)[K], uint32_t addr, Hook& hook) {
        float4 r[NR];
        reads(r, addr, std::make_integer_sequence<int, NR>{});
        quads(r, acc, w, hook, std::make_integer_sequence<int, K / 4>{});
        mfma_tail_pad(acc[0], acc[1], acc[2], acc[3]);
    }
};

// ---------------------------------------------------------------------------------------------------
// The same product with BOTH operands as two f16 planes (x 2^s = h + m, gemm_x3.hip's PREC = 4 arithmetic) on
// v_mfma_f32_4x4x4_16b_f16: one instruction contracts FOUR k at the 4x4x1 f32 instruction's issue rate (8.6 cycles,
// tools/ubench/mfma4x4_f16.hip), and three of them (h h, h m, m h) replace four: 192 instead of 256 per wave and step.
// Lane 4b + i reads, per plane, the sixteen halfs [i][16 b .. 16 b + 15] of the plain-order image as four ds_read_b64;
// k-group g (abid = g & 15 of read g >> 4) contracts k = 16 (g & 15) + 4 (g >> 4) + {0 .. 3}.  Three accumulator chains, one per piece
// product: the two small ones sum apart from the large one.
// ---------------------------------------------------------------------------------------------------
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void lds_read8(f16x4& dst, uint32_t addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <bool ZERO, int ABID>
__device__ __forceinline__ void mfma3_f16(f32x4& c0, f32x4& c1, f32x4& c2, const f16x4& ah, const f16x4& am, const f16x4& wh, const f16x4& wm) {
    if constexpr (ZERO) {
        asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %3, %5, 0 cbsz:4 abid:%7\n\tv_mfma_f32_4x4x4_16b_f16 %1, %3, %6, 0 cbsz:4 abid:%7\n\t"
                     "v_mfma_f32_4x4x4_16b_f16 %2, %4, %5, 0 cbsz:4 abid:%7"
                     : "=&v"(c0), "=&v"(c1), "=&v"(c2)
                     : "v"(ah), "v"(am), "a"(wh), "a"(wm), "i"(ABID));
    } else {
        asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %3, %5, %0 cbsz:4 abid:%7\n\tv_mfma_f32_4x4x4_16b_f16 %1, %3, %6, %1 cbsz:4 abid:%7\n\t"
                     "v_mfma_f32_4x4x4_16b_f16 %2, %4, %5, %2 cbsz:4 abid:%7"
                     : "+v"(c0), "+v"(c1), "+v"(c2)
                     : "v"(ah), "v"(am), "a"(wh), "a"(wm), "i"(ABID));
    }
}
__device__ __forceinline__ void mfma_tail_pad3(f32x4& a, f32x4& b, f32x4& c) {
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b), "+v"(c));
}
template <int K, int PLANE_BYTES>
struct FwdProductColH {
    static constexpr int NG = K / 4;         // k-groups = hook points per step
    static constexpr int NJ = K / 64;        // ds_read_b64 per plane and step
    static constexpr int HOOKS = NG;
    template <int Gi, class Hook>
    static __device__ __forceinline__ void group(const f16x4 (&rh)[NJ], const f16x4 (&rm)[NJ], f32x4 (&acc)[3], const f16x4 (&wh)[NG],
                                                 const f16x4 (&wm)[NG], Hook& hook) {
        if constexpr (Gi % 16 == 0) wait_lgkm<2 * (NJ - 1 - Gi / 16)>();
        mfma3_f16<Gi == 0, Gi & 15>(acc[0], acc[1], acc[2], rh[Gi >> 4], rm[Gi >> 4], wh[Gi], wm[Gi]);
        hook(std::integral_constant<int, Gi>{});
    }
    template <class Hook, int... Gs>
    static __device__ __forceinline__ void groups(const f16x4 (&rh)[NJ], const f16x4 (&rm)[NJ], f32x4 (&acc)[3], const f16x4 (&wh)[NG],
                                                  const f16x4 (&wm)[NG], Hook& hook, std::integer_sequence<int, Gs...>) {
        (group<Gs>(rh, rm, acc, wh, wm, hook), ...);
    }
    template <int... Js>
    static __device__ __forceinline__ void reads(f16x4 (&rh)[NJ], f16x4 (&rm)[NJ], uint32_t addr, std::integer_sequence<int, Js...>) {
        ((lds_read8<8 * Js>(rh[Js], addr), lds_read8<PLANE_BYTES + 8 * Js>(rm[Js], addr)), ...);
    }
    template <class Hook>
    static __device__ __forceinline__ void run(f32x4 (&acc)[3], const f16x4 (&wh)[NG], const f16x4 (&wm)[NG], uint32_t addr, Hook& hook) {
        f16x4 rh[NJ], rm[NJ];
        reads(rh, rm, addr, std::make_integer_sequence<int, NJ>{});
        groups(rh, rm, acc, wh, wm, hook, std::make_integer_sequence<int, NG>{});
        mfma_tail_pad3(acc[0], acc[1], acc[2]);
    }
};

// The two-plane product in TWO phases, for a team member that knows its OWN 64 units' states a hand-off earlier than the other members':
// with the image in member-local k order (own units first) the k-groups of broadcast blocks 0 .. 3 contract own states only.
//   phase A: those 16 k-groups (48 MFMAs), issued while the other members' granules are in flight;
//   phase B: the other 48 (144 MFMAs) after the gather, from a second set of LDS reads.
// Hook numbers: 0 .. 15 in phase A, 16 .. 63 in phase B.
template <int K, int PLANE_BYTES>
struct FwdProductOwnFirst {
    static_assert(K == 256, "four members x 64 units");
    static constexpr int NG = K / 4, NJ = K / 64;
    template <int... Js>
    static __device__ __forceinline__ void reads(f16x4 (&rh)[NJ], f16x4 (&rm)[NJ], uint32_t addr, std::integer_sequence<int, Js...>) {
        ((lds_read8<8 * Js>(rh[Js], addr), lds_read8<PLANE_BYTES + 8 * Js>(rm[Js], addr)), ...);
    }
    template <int A, class Hook>           // phase A position A: read J = A / 4, block A % 4
    static __device__ __forceinline__ void group_a(const f16x4 (&rh)[NJ], const f16x4 (&rm)[NJ], f32x4 (&acc)[3], const f16x4 (&wh)[NG],
                                                   const f16x4 (&wm)[NG], Hook& hook) {
        constexpr int J = A / 4, B = A % 4, g = 16 * J + B;
        if constexpr (A % 4 == 0) wait_lgkm<2 * (NJ - 1 - J)>();
        mfma3_f16<A == 0, B>(acc[0], acc[1], acc[2], rh[J], rm[J], wh[g], wm[g]);
        hook(std::integral_constant<int, A>{});
    }
    template <int Q, class Hook>           // phase B position Q: read J = Q / 12, block 4 + Q % 12
    static __device__ __forceinline__ void group_b(const f16x4 (&rh)[NJ], const f16x4 (&rm)[NJ], f32x4 (&acc)[3], const f16x4 (&wh)[NG],
                                                   const f16x4 (&wm)[NG], Hook& hook) {
        constexpr int J = Q / 12, B = 4 + Q % 12, g = 16 * J + B;
        if constexpr (Q % 12 == 0) wait_lgkm<2 * (NJ - 1 - J)>();
        mfma3_f16<false, B>(acc[0], acc[1], acc[2], rh[J], rm[J], wh[g], wm[g]);
        hook(std::integral_constant<int, 16 + Q>{});
    }
    template <class Hook, int... As>
    static __device__ __forceinline__ void groups_a(const f16x4 (&rh)[NJ], const f16x4 (&rm)[NJ], f32x4 (&acc)[3], const f16x4 (&wh)[NG],
                                                    const f16x4 (&wm)[NG], Hook& hook, std::integer_sequence<int, As...>) {
        (group_a<As>(rh, rm, acc, wh, wm, hook), ...);
    }
    template <class Hook, int... Qs>
    static __device__ __forceinline__ void groups_b(const f16x4 (&rh)[NJ], const f16x4 (&rm)[NJ], f32x4 (&acc)[3], const f16x4 (&wh)[NG],
                                                    const f16x4 (&wm)[NG], Hook& hook, std::integer_sequence<int, Qs...>) {
        (group_b<Qs>(rh, rm, acc, wh, wm, hook), ...);
    }
    template <class Hook>
    static __device__ __forceinline__ void phase_a(f32x4 (&acc)[3], const f16x4 (&wh)[NG], const f16x4 (&wm)[NG], uint32_t addr, Hook& hook) {
        f16x4 rh[NJ], rm[NJ];
        reads(rh, rm, addr, std::make_integer_sequence<int, NJ>{});
        groups_a(rh, rm, acc, wh, wm, hook, std::make_integer_sequence<int, 16>{});
    }
    template <class Hook>
    static __device__ __forceinline__ void phase_b(f32x4 (&acc)[3], const f16x4 (&wh)[NG], const f16x4 (&wm)[NG], uint32_t addr, Hook& hook) {
        f16x4 rh[NJ], rm[NJ];
        reads(rh, rm, addr, std::make_integer_sequence<int, NJ>{});
        groups_b(rh, rm, acc, wh, wm, hook, std::make_integer_sequence<int, 48>{});
        mfma_tail_pad3(acc[0], acc[1], acc[2]);
    }
};

template <int KH>
struct BwdProduct {
    // the two wave halves contract different k ranges, so the broadcast stays inside a half (cbsz:3):
    // lane 4b'+i (b' = 0..7 within the half) of register j holds dgates[i][half*KH + 8j + b']
    static constexpr int NJ = KH / 8;        // A registers per step
    static constexpr int NR = NJ / 4;        // ds_read_b128 per step
    static constexpr int HOOKS = KH / 4;     // one hook per MFMA quad (4 consecutive k)
    template <int K4, class Hook>
    static __device__ __forceinline__ void quad(const float4 (&r)[NR], f32x4 (&acc)[4], const float (&w)[KH], Hook& hook) {
        constexpr int k = 4 * K4;
        if constexpr (k % 32 == 0) wait_lgkm<NR - 1 - k / 32>();
        mfma_pair_seq<k == 0, k & 7, (k + 1) & 7>(acc[0], acc[1], a_reg<(k >> 3)>(r), w[k], w[k + 1]);
        mfma_pair_seq<k == 0, (k + 2) & 7, (k + 3) & 7>(acc[2], acc[3], a_reg<(k >> 3)>(r), w[k + 2], w[k + 3]);
        hook(std::integral_constant<int, K4>{});
    }
    template <class Hook, int... Ks>
    static __device__ __forceinline__ void quads(const float4 (&r)[NR], f32x4 (&acc)[4], const float (&w)[KH], Hook& hook,
                                                 std::integer_sequence<int, Ks...>) {
        (quad<Ks>(r, acc, w, hook), ...);
    }
    template <int... Rs>
    static __device__ __forceinline__ void reads(float4 (&r)[NR], uint32_t addr, std::integer_sequence<int, Rs...>) {
        (lds_read16<16 * Rs>(r[Rs], addr), ...);
    }
    template <class Hook>
    static __device__ __forceinline__ void run(f32x4 (&acc)[4], const float (&w)[KH], uint32_t addr, Hook& hook) {
        float4 r[NR];
        reads(r, addr, std::make_integer_sequence<int, NR>{});
        quads(r, acc, w, hook, std::make_integer_sequence<int, KH / 4>{});
        mfma_tail_pad(acc[0], acc[1], acc[2], acc[3]);
    }
    // position of gate column `col` (0..2*KH) inside a sequence row of the LDS image
    static __device__ __forceinline__ int pos(int col) {
        const int half = col / KH, kk = col - half * KH;
        return half * KH + (kk & 7) * NJ + (kk >> 3);
    }
};

// The four sequence slots of a workgroup.  A slot whose sequence does not exist (ragged last
// workgroup) or is empty duplicates the workgroup's first real sequence: it then computes and stores
// bit-identical values to the same addresses, and no part of the step needs an "is this cell real"
// branch.  Returns false if the workgroup has nothing to do.
__device__ __forceinline__ bool map_slots(const RnnStepArgs& p, int b0, int (&bmap)[4], int& tmax) {
    int first = -1;
    tmax = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = b0 + q;
        const int l = b < p.n_seq ? p.seq_len[b] : 0;
        if (l > 0 && first < 0) first = b;
        tmax = max(tmax, l);
        bmap[q] = l > 0 ? b : -1;
    }
    if (first < 0) return false;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (bmap[q] < 0) bmap[q] = first;
    return true;
}

}  // namespace dc
