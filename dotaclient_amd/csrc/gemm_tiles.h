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
    static __device__ __forceinline__ void issue(const float* __restrict__ origin, const size_t (&off)[NI],
                                                 float* __restrict__ S, int wave) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(origin + off[i]), (lptr_t)(S + (wave * NI + i) * 256), 16, 0, 0);
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


// one K step (BK = 32) of a wave's (TM x 32) x (TN x 32) output tile: 4 fragment groups, each read
// feeds four v_mfma_f32_32x32x2_f32 per (row tile, column tile)
template <class LA, class LB, int TM, int TN>
__device__ __forceinline__ void mma_kstep(const float* __restrict__ a_s, const float* __restrict__ b_s, int a_row0, int b_row0,
                                          int fr, int fq, f32x16 (&acc)[TM][TN]) {
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
#ifdef DC_MMA_SETPRIO
        __builtin_amdgcn_s_setprio(1);      // experiment (A/B build): the wave inside its MFMA burst wins issue arbitration -
                                            // measured 2-7 % SLOWER on every MFMA kernel of the step
#endif
        DC_MMA_E(x) DC_MMA_E(y) DC_MMA_E(z) DC_MMA_E(w)
#ifdef DC_MMA_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#undef DC_MMA_E
    }
}

}  // namespace dc
