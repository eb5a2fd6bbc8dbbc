// LDS tile machinery shared by the fast GEMM (gemm.hip) and the fused embedding kernels
// (embed_fused.hip): direct global -> LDS DMA of 128-byte k-contiguous rows with a source-side XOR
// swizzle, k-major tiles as they are, and the MFMA fragment reads that go with both images.
// See the "Fast path" comment in gemm.hip for the layout rationale.
#pragma once
#include "common.h"

namespace dc {

enum { GEMM_BK = 32 };

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BR, bool KM>
struct FastTile {
    static constexpr int LDS_FLOATS = GEMM_BK * BR;
    static constexpr int NI = BR / 32;     // DMA instructions per wave per tile (each moves 1 KB)

    // per-lane source offsets (floats, relative to the operand at k = k_base) of this wave's NI DMA
    // pieces; constant over the K loop.  EDGE: rows past the logical extent R re-read a valid row (their
    // products only reach output rows/columns the epilogue never stores).
    template <bool EDGE>
    static __device__ __forceinline__ void src_offsets(size_t (&off)[NI], int ld, int r_base, int R, int wave, int lane) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int n = wave * NI + i;   // piece index: LDS floats [n*256, n*256+256)
            if constexpr (!KM) {
                const int row = n * 8 + (lane >> 3), pos = lane & 7;
                int rg = r_base + row;
                if constexpr (EDGE) rg = min(rg, R - 1);
                off[i] = (size_t)rg * ld + 4 * (pos ^ ((row >> 1) & 7));
            } else {
                constexpr int V = BR / 4;              // 16-byte slots per k row
                const int k = n * (64 / V) + lane / V, r4 = lane % V;
                int cg = r_base + 4 * r4;
                if constexpr (EDGE) cg = min(cg, ld - 4);   // stay inside the physical row
                off[i] = (size_t)k * ld + cg;
            }
        }
    }
    // AUX: cache policy of the DMA (2 = non-temporal: an operand that is read once and would otherwise push a re-read one out of the L2)
    template <int AUX = 0>
    static __device__ __forceinline__ void issue(const float* __restrict__ origin, const size_t (&off)[NI],
                                                 float* __restrict__ S, int wave) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(origin + off[i]), (lptr_t)(S + (wave * NI + i) * 256), 16, 0, AUX);
    }
    // the four k values (8j+4q+e, e = 0..3) of row r for this lane
    static __device__ __forceinline__ float4 frag(const float* __restrict__ S, int r, int j, int q) {
        if constexpr (!KM) {
            return *reinterpret_cast<const float4*>(S + r * GEMM_BK + 4 * ((2 * j + q) ^ ((r >> 1) & 7)));
        } else {
            const float* s = S + (8 * j + 4 * q) * BR + r;
            return make_float4(s[0], s[BR], s[2 * BR], s[3 * BR]);
        }
    }
};


// ---------------------------------------------------------------------------------------------------
// fp32-grade products on the bf16 matrix cores (DC_GEMM_X3, the default).
//
// gfx950 has no reduced-precision fast path for f32 inputs (no xf32 / TF32) and its f32-input MFMA runs at the f32
// VECTOR rate, 1/16 of the bf16 MFMA.  But an f32 splits EXACTLY into three bf16 pieces,
//     x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)      (8 + 8 + 8 = 24 mantissa bits,
// round-to-nearest pieces: every residual is exact in f32), every piece product is exact in f32, and
//     a * b = hh + (hm + mh) + (hl + lh + mm) + O(2^-32 |ab|)
// so six v_mfma_f32_32x32x16_bf16 (f32 accumulate) per K = 16 reproduce the f32 product to f32 round-off (measured on
// the network's shapes, tools/ubench/gemm_x3.hip: max error 1.4e-7 .. 5.5e-7 of max |C| against an f64 reference, the
// k-ordered f32 fma chain of the f32 MFMA: 2.1e-7 .. 3.7e-7) for 6 x 32 = 192 matrix-pipe cycles instead of the
// 8 x 64 = 512 of eight v_mfma_f32_32x32x2_f32: a 2.67x higher ceiling (419 TF).  The split is VALU work done on the
// fragments right after their LDS reads (v_cvt_pk_bf16_f32 + subtract, ~5.5 instructions per element), issued
// between the MFMAs.  Build with -DDC_GEMM_X3=0 for the exact-f32 MFMA products (the A/B reference).
// ---------------------------------------------------------------------------------------------------
#ifndef DC_GEMM_X3
#define DC_GEMM_X3 1
#endif

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {   // {bf16(hi), bf16(lo)}, round to nearest even
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
struct Split3 { bf16x8 h, m, l; };
// eight f32 (two fragment reads) -> three bf16x8 MFMA operands
__device__ __forceinline__ Split3 split3(const float4& x0, const float4& x1) {
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        h[i] = cvt_pk_bf16(a, b);
        const float ra = a - __uint_as_float(h[i] << 16), rb = b - __uint_as_float(h[i] & 0xffff0000u);
        m[i] = cvt_pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(m[i] << 16), sb = rb - __uint_as_float(m[i] & 0xffff0000u);
        l[i] = cvt_pk_bf16(sa, sb);
    }
    Split3 s;
    s.h = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
    s.m = __builtin_bit_cast(bf16x8, u32x4{m[0], m[1], m[2], m[3]});
    s.l = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
    return s;
}

// one K step (BK = 32) of a wave's (TM x 32) x (TN x 32) output tile.
// DC_GEMM_X3: two K = 16 sub-steps; lane (i = lane & 31, q = lane >> 5) contracts k = 16s + 4q + e and 16s + 8 + 4q + e
// (e = 0..3) - two ds_read_b128 per operand block, the same k permutation on both operands - split into bf16 pieces, six
// MFMAs per (row tile, column tile), smallest terms first.
// else: 4 fragment groups, each read feeds four v_mfma_f32_32x32x2_f32 per (row tile, column tile)
template <class LA, class LB, int TM, int TN>
__device__ __forceinline__ void mma_kstep(const float* __restrict__ a_s, const float* __restrict__ b_s, int a_row0, int b_row0,
                                          int fr, int fq, f32x16 (&acc)[TM][TN]) {
#if DC_GEMM_X3
#pragma unroll
    for (int s = 0; s < GEMM_BK / 16; ++s) {
        Split3 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            a[i] = split3(LA::frag(a_s, a_row0 + i * 32 + fr, 2 * s, fq), LA::frag(a_s, a_row0 + i * 32 + fr, 2 * s + 1, fq));
#pragma unroll
        for (int i = 0; i < TN; ++i)
            b[i] = split3(LB::frag(b_s, b_row0 + i * 32 + fr, 2 * s, fq), LB::frag(b_s, b_row0 + i * 32 + fr, 2 * s + 1, fq));
        // piece-major order: the TM x TN accumulators take turns, so consecutive MFMAs never chain on one accumulator
#define DC_MMA_P(X, Y)                                                                                            \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                          \
            _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)                                                   \
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].X, b[jn].Y, acc[i][jn], 0, 0, 0);
        DC_MMA_P(l, h) DC_MMA_P(h, l) DC_MMA_P(m, m) DC_MMA_P(m, h) DC_MMA_P(h, m) DC_MMA_P(h, h)
#undef DC_MMA_P
    }
#else
#pragma unroll
    for (int j = 0; j < GEMM_BK / 8; ++j) {
        float4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = LA::frag(a_s, a_row0 + i * 32 + fr, j, fq);
#pragma unroll
        for (int i = 0; i < TN; ++i) bf[i] = LB::frag(b_s, b_row0 + i * 32 + fr, j, fq);
        // element-major order: the TM x TN accumulators take turns, so consecutive MFMAs never chain on the
        // same accumulator (each sees TM*TN - 1 independent issues before its own next update)
#define DC_MMA_E(E)                                                                                              \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                          \
            _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)                                                   \
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].E, bf[jn].E, acc[i][jn], 0, 0, 0);
        DC_MMA_E(x) DC_MMA_E(y) DC_MMA_E(z) DC_MMA_E(w)
#undef DC_MMA_E
    }
#endif
}


// ---------------------------------------------------------------------------------------------------
// A k-contiguous operand that arrives PRE-SPLIT (the weights: split once per pass by split_weight_planes, gemm_x3.hip):
// three bf16 planes [plane][rows][ld], moved global -> LDS by DMA like FastTile.  Image of one K step (32 k) per plane:
// [BR rows][64 bytes], lane-linear (a DMA piece = 16 rows x 64 B), the four 16-byte k chunks of a row XOR-swizzled on the
// SOURCE side with (row >> 2) & 3: the 16 rows of every ds_read_b128 lane group then hit 16 different 16-byte bank groups.
// Its fragments need no VALU work at all.
// ---------------------------------------------------------------------------------------------------
template <int BR, int NPL = 3>     // NPL planes: 3 bf16 pieces, or 2 f16 pieces (mma_kstep_bplanes_h)
struct PlaneTile {
    static constexpr int PLANE_FLOATS = BR * 16;            // BR x 32 bf16
    static constexpr int LDS_FLOATS = NPL * PLANE_FLOATS;
    static constexpr int NI = NPL * BR / 16 / 4;            // DMA pieces per wave per K step (4 waves): 6 at BR = 128, 3 planes
    // per-lane source offsets (bf16 elements, relative to plane 0 at k = k_base) of this wave's pieces
    static __device__ __forceinline__ void src_offsets(size_t (&off)[NI], int ld, long long plane_elems, int r_base, int wave, int lane) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int n = wave * NI + i;                   // piece index over the three planes
            const int pl = n / (BR / 16), pr = n % (BR / 16);
            const int row = pr * 16 + (lane >> 2), slot = lane & 3;
            off[i] = (size_t)pl * plane_elems + (size_t)(r_base + row) * ld + 8 * (slot ^ ((row >> 2) & 3));
        }
    }
    static __device__ __forceinline__ void issue(const uint16_t* __restrict__ origin, const size_t (&off)[NI], float* __restrict__ S, int wave) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(origin + off[i]), (lptr_t)(S + (wave * NI + i) * 256), 16, 0, 0);
    }
    // k = 16 s + 8 g .. + 7 of row r, plane p
    static __device__ __forceinline__ bf16x8 frag(const float* __restrict__ S, int r, int s, int g, int p) {
        const int chunk = (2 * s + g) ^ ((r >> 2) & 3);
        return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(S + p * PLANE_FLOATS) + r * 64 + chunk * 16);
    }
};

// K step with an f32 A image (FastTile: split after the fragment reads) and a pre-split B image (PlaneTile).  A's k
// permutation inside a K = 16 sub-step (lane group q contracts k = 16s + 4q + e and 16s + 8 + 4q + e) differs from the
// natural order of the planes (group g: k = 16s + 8g + e'), so A fragments are read in the planes' order instead: group q
// takes the two float4 at k chunks 4s + 2q and 4s + 2q + 1 (k = 16s + 8q .. + 7).
template <class LA, int BRB, int TM, int TN>
__device__ __forceinline__ void mma_kstep_bplanes(const float* __restrict__ a_s, const float* __restrict__ b_s, int a_row0, int b_row0,
                                                  int fr, int fq, f32x16 (&acc)[TM][TN]) {
    using LB = PlaneTile<BRB>;
#pragma unroll
    for (int s = 0; s < GEMM_BK / 16; ++s) {
        Split3 a[TM];
        bf16x8 bh[TN], bm[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            a[i] = split3(LA::frag(a_s, a_row0 + i * 32 + fr, 2 * s + fq, 0), LA::frag(a_s, a_row0 + i * 32 + fr, 2 * s + fq, 1));
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bh[j] = LB::frag(b_s, b_row0 + j * 32 + fr, s, fq, 0);
            bm[j] = LB::frag(b_s, b_row0 + j * 32 + fr, s, fq, 1);
            bl[j] = LB::frag(b_s, b_row0 + j * 32 + fr, s, fq, 2);
        }
#define DC_MMA_Q(X, Y)                                                                                            \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                          \
            _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)                                                   \
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].X, Y[jn], acc[i][jn], 0, 0, 0);
        DC_MMA_Q(l, bh) DC_MMA_Q(h, bl) DC_MMA_Q(m, bm) DC_MMA_Q(m, bh) DC_MMA_Q(h, bm) DC_MMA_Q(h, bh)
#undef DC_MMA_Q
    }
}

// ---------------------------------------------------------------------------------------------------
// The same products from TWO f16 pieces per operand and THREE MFMAs (DC_DIMS_F16X2; rationale, accuracy and exponent-range
// caveat: gemm_x3.hip, PREC = 4).  x * s = h + m (+ l, dropped) with h = f16(x s), m = f16(x s - h); s: the operand's power-of-two
// pre-scale (1 when the producer already applied it); the caller scales the accumulators back by 1 / (sa sb).
// ---------------------------------------------------------------------------------------------------
// The m*m term.  With x 2^s = h + m + l (round to nearest even: |m| <= 2^-11 |x|, |l| <= 2^-23 |x|) a product is
//     a b = hh + hm + mh + [mm] + (al b + a bl - al bl).
// What the pieces cannot represent is <= 2^-22 |ab|; the m*m term is <= 2^-22 |ab| too, so dropping it takes the bound per term to
// 2^-21 |ab| - below what an f32 fma chain of K >= 8 terms may lose to rounding (K 2^-24) - for 25 % fewer MFMAs on products that run
// at the socket's power cap.  Measured against f64 on the network's shapes the two forms are indistinguishable, and both are at or
// below the six-bf16-MFMA form (tools/ubench/gemm_x3.hip: identical to three digits; tools/gemm_bench.py, profiles/r04/
// v12_gemm_bench_{three,four}_mfma.txt: 2.8e-7 .. 8.1e-7 against 2.7e-7 .. 8.3e-7 of max |C|; the f32 accumulation dominates), the
// dense products are 5-10 % faster and the step 2.2 % (19.5 against 20.0 ms, same box, alternating).  Default: THREE MFMAs (hh, hm, mh);
// -DDC_X2H_KEEP_MM builds the four-MFMA form (A/B).
#ifdef DC_X2H_KEEP_MM
#define DC_X2H_MM(x) x
#else
#define DC_X2H_MM(x)
#endif
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {      // {f16(hi), f16(lo)}, round to nearest even
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, f16x2_t));
}
struct Split2h { f16x8 h, m; };
template <bool SCALE>
__device__ __forceinline__ Split2h split2h(const float4& x0, const float4& x1, float s) {
    float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    if constexpr (SCALE) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] *= s;
    }
    unsigned h[4], m[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = cvt_pk_f16(x[2 * i], x[2 * i + 1]);
        const f16x2_t hv = __builtin_bit_cast(f16x2_t, h[i]);
        m[i] = cvt_pk_f16(x[2 * i] - (float)hv.x, x[2 * i + 1] - (float)hv.y);
    }
    Split2h r;
    r.h = __builtin_bit_cast(f16x8, u32x4{h[0], h[1], h[2], h[3]});
    r.m = __builtin_bit_cast(f16x8, u32x4{m[0], m[1], m[2], m[3]});
    return r;
}

// mma_kstep with f16 pieces: same fragment reads and k permutation, three MFMAs per (row tile, column tile), smallest terms first
template <class LA, class LB, int TM, int TN, bool SCALE_A, bool SCALE_B>
__device__ __forceinline__ void mma_kstep_h(const float* __restrict__ a_s, const float* __restrict__ b_s, int a_row0, int b_row0,
                                            int fr, int fq, f32x16 (&acc)[TM][TN], float sa, float sb) {
#pragma unroll
    for (int s = 0; s < GEMM_BK / 16; ++s) {
        Split2h a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            a[i] = split2h<SCALE_A>(LA::frag(a_s, a_row0 + i * 32 + fr, 2 * s, fq), LA::frag(a_s, a_row0 + i * 32 + fr, 2 * s + 1, fq), sa);
#pragma unroll
        for (int i = 0; i < TN; ++i)
            b[i] = split2h<SCALE_B>(LB::frag(b_s, b_row0 + i * 32 + fr, 2 * s, fq), LB::frag(b_s, b_row0 + i * 32 + fr, 2 * s + 1, fq), sb);
#define DC_MMA_H(X, Y)                                                                                            \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                          \
            _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)                                                   \
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i].X, b[jn].Y, acc[i][jn], 0, 0, 0);
        DC_X2H_MM(DC_MMA_H(m, m)) DC_MMA_H(m, h) DC_MMA_H(h, m) DC_MMA_H(h, h)
#undef DC_MMA_H
    }
}

// mma_kstep_bplanes with f16 pieces: A image f32 (split after the fragment reads, scale sa), B two pre-scaled f16 planes
template <class LA, int BRB, int TM, int TN, bool SCALE_A>
__device__ __forceinline__ void mma_kstep_bplanes_h(const float* __restrict__ a_s, const float* __restrict__ b_s, int a_row0, int b_row0,
                                                    int fr, int fq, f32x16 (&acc)[TM][TN], float sa) {
    using LB = PlaneTile<BRB, 2>;
#pragma unroll
    for (int s = 0; s < GEMM_BK / 16; ++s) {
        Split2h a[TM];
        f16x8 bh[TN], bm[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            a[i] = split2h<SCALE_A>(LA::frag(a_s, a_row0 + i * 32 + fr, 2 * s + fq, 0), LA::frag(a_s, a_row0 + i * 32 + fr, 2 * s + fq, 1), sa);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bh[j] = __builtin_bit_cast(f16x8, LB::frag(b_s, b_row0 + j * 32 + fr, s, fq, 0));
            bm[j] = __builtin_bit_cast(f16x8, LB::frag(b_s, b_row0 + j * 32 + fr, s, fq, 1));
        }
#define DC_MMA_HQ(X, Y)                                                                                           \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                          \
            _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)                                                   \
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i].X, Y[jn], acc[i][jn], 0, 0, 0);
        DC_X2H_MM(DC_MMA_HQ(m, bm)) DC_MMA_HQ(m, bh) DC_MMA_HQ(h, bm) DC_MMA_HQ(h, bh)
#undef DC_MMA_HQ
    }
}

}  // namespace dc
