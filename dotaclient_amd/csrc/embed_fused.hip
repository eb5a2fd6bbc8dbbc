// Fused per-unit embedding MLP: the 12 -> 128 first layer never touches HBM.
//
// Replaces /root/reference/policy.py:100-126 (forward) and the autograd products of
// /root/reference/optimizer.py:672 for
//     basic = relu(x W1^T + b1)          (affine_unit_basic_stats, shared by all unit types, K = 12)
//     emb   = basic W2_t^T + b2_t        (affine_unit_{ah,eh,anh,enh,ath,eth}, K = 128)
// The unfused path (embed.hip + gemm.hip) materialises `basic` (512 B per unit, 335 MB per 64x256
// batch), reads it back in the forward GEMM, twice more in the backward, and writes + re-reads the
// equally large d(basic).  Here the K = 12 layer is recomputed from the 48-byte unit record wherever
// it is needed, ON THE MATRIX CORES: a 32x32 block of `basic` is six v_mfma_f32_32x32x2_f32 whose
// operands are one register per lane each (the records; W1 stays in registers for the whole kernel) -
// +9 % MFMA work instead of 192 VALU FMAs and 48 LDS reads per thread per K step, which used to take
// as long as the 64 MFMAs of the step they fed.
//   embed_fwd_fused : A tile (basic) generated into the swizzled LDS image of the fast GEMM, B tile
//                     (W2_t) by DMA, exact-fp32 MFMA, emb written once, the max-pools of the 1- and
//                     16-unit types taken from the accumulators
//   embed_bwd_dw2   : dW2_t = demb_t^T basic_t as a split-K product whose B operand is generated
//   embed_bwd_dw1   : d(basic) = (demb_t W2_t) * [basic > 0] stays in the accumulators (the mask is
//                     regenerated in the accumulators' own layout); the epilogue folds it straight into
//                     dW1 / db1 (d(basic) is never stored)
// `basic` is evaluated the same way everywhere - six v_mfma_f32_32x32x2_f32 over the feature pairs (an
// exact k-ordered fmaf chain) plus b1 - so the relu mask of the backward is bitwise the forward's.  Requires rows % 128 == 0 (every type block then starts on a tile boundary); other
// batches take the unfused path.
#include <stdio.h>
#include <stdlib.h>
#include <utility>
#include "kernels.h"
#include "gemm_tiles.h"

namespace dc {

enum { EF_OBS = 483, EF_EMB = 128, EF_TILE = 128 };
constexpr float F16_S_W1 = 256.f;      // DC_DIMS_F16X2: power-of-two pre-scale of the first-layer weights (the records take the activations' 2^4)

struct EmbTypes {
    long long row_begin[7];   // type-major row where type t starts (nr * cum[t])
    int tile_begin[7];        // row_begin / 128
    int wg_begin[7];          // embed_bwd_dw2: first workgroup of type t
    int steps_per_wg;         // embed_bwd_dw2: K steps (32 rows) per workgroup
    long long nr_valid;       // env-steps that exist; steps in [nr_valid, nr) are padding up to a multiple of 128: their records re-read
                              // the last valid step (forward results land in pad rows nobody reads, backward sees zero gradients)
    int sparse16;             // backward: the two 16-unit types are handled by embed_bwd_pool16 (embed_sparse.hip);
                              // their dW2 slab ranges [wg_begin[2], wg_begin[4]) are written by that kernel
    // forward only (embed_fwd_fused): first tile of type t.  pool5 = 0: tile_begin.  pool5 = 1: a tile of the five-unit type holds 24 WHOLE
    // env-steps - six per 32-row block (rows 30, 31 of a block are padding) - so that its max-pool can be taken in the epilogue too
    int ftile_begin[7];
    int pool5;
};
enum { EF_P5_STEPS = 24 };

__device__ __forceinline__ int ef_type_of_tile(const EmbTypes& ty, int tile) {
    int t = 0;
#pragma unroll
    for (int i = 1; i < 6; ++i)
        if (tile >= ty.tile_begin[i]) t = i;
    return t;
}
__device__ __forceinline__ int ef_ftype_of_tile(const EmbTypes& ty, int tile) {
    int t = 0;
#pragma unroll
    for (int i = 1; i < 6; ++i)
        if (tile >= ty.ftile_begin[i]) t = i;
    return t;
}
// type-major row of the tile's row `r` (forward tiling): rows of the five-unit type with pool5 - block b = r >> 5 holds the steps
// 24 tile + 6 b .. + 5, row 5 s + u of the block = unit u of its step s; the block's rows 30, 31 repeat its last unit
__device__ __forceinline__ long long ef_frow(const EmbTypes& ty, int t, int tile, int r) {
    if (t == 1 && ty.pool5) {
        const int b = r >> 5, q = min(r & 31, 29);
        return ty.row_begin[1] + 5LL * ((long long)(tile - ty.ftile_begin[1]) * EF_P5_STEPS + 6 * b) + q;
    }
    return ty.row_begin[t] + (long long)(tile - ty.ftile_begin[t]) * EF_TILE + r;
}
__device__ __forceinline__ int ef_units(int t) { return t == 1 ? 5 : ((t == 2 || t == 3) ? 16 : 1); }
__device__ __forceinline__ int ef_cum(int t) { return t == 0 ? 0 : (t == 1 ? 1 : (t == 2 ? 6 : (t == 3 ? 22 : (t == 4 ? 38 : 39)))); }

// 12-feature record of type-major row `local` (relative to its type block; < 2^31: policy.hip check_dims).
// The unit counts are 1, 5 and 16: shift / multiply-high instead of a 64-bit division per record.
__device__ __forceinline__ const float* ef_record(const float* __restrict__ obs, int t, long long local_, long long nr_valid) {
    const unsigned local = (unsigned)local_;
    unsigned n, u;
    if (t == 2 || t == 3) { n = local >> 4; u = local & 15u; }
    else if (t == 1) { n = __umulhi(local, 0xCCCCCCCDu) >> 2; u = local - 5u * n; }
    else { n = local; u = 0u; }
    n = min(n, (unsigned)(nr_valid - 1));        // padding steps
    return obs + (size_t)n * EF_OBS + 3 + (ef_cum(t) + (int)u) * 12;
}

// ---------------------------------------------------------------------------------------------------
// forward: persistent workgroups stride over the 128-row tiles (all 128 output channels, K = 128 in 4
// steps of 32).  Phase stamps (s_memtime, DC_EF_TIMING=1) of the one-tile-per-workgroup version showed the
// K loop at 43 % of a tile's life: 26 % went to a prologue that re-loaded W1 and waited for the unit
// records, 31 % to the epilogue.  Hence the two points below.  What remains (measured per tile and wave, in
// cycles): K loop 26.8 k for 17.9 k of MFMA issue, epilogue 15.6 k.  Experiments: with the emb stores removed
// altogether 20.2 k / 8.5 k; storing only the rows the attention can read (a quarter of the bytes, WRITE_SIZE
// 382 -> 127 MB), skipping the store instructions of unneeded rows, non-temporal stores, or staggering the
// workgroups in time (per CU or across the chip) change NOTHING; an explicit s_waitcnt vmcnt(0) behind the epilogue
// costs only ~500 cycles (the stores are acknowledged quickly) and leaves the K loop at 27 k.  So it is neither
// store bytes, nor store issue, nor store latency in front of the DMA barriers - the mechanism by which the
// presence of the emb stores costs ~7 k cycles in each phase is NOT understood yet; the robust way around it is
// not to materialise emb at all ("next").
//   * W1 / b1 live in registers for the whole kernel; the NEXT tile's records are loaded during the
//     current K loop, and its step-0 operands (generated A, DMA'd B) are produced during the current
//     tile's last K step, in the stage buffer that step does not use - the K steps of consecutive tiles
//     form one continuous pipeline;
//   * the epilogue transposes the accumulators through the other (free) stage buffer and writes emb
//     with 16-byte stores, 256 contiguous bytes per 16 lanes.
// ---------------------------------------------------------------------------------------------------
// BPL: W2 arrives as pre-split bf16 planes W2p [3][6 x 128][128] (split_weight_planes, once per pass) and goes to LDS as such:
// its fragments need no split in the K loop (half of the loop's VALU work)
// F16 (DC_DIMS_F16X2, needs BPL): the 128 x 128 layer from two f16 pieces per operand and three MFMAs (gemm_x3.hip, PREC = 4).  W2p then
// holds [2] f16 planes of W2 * 2^8, and the A operand is generated pre-scaled: basic * s_act = relu(x (W1 s_act)^T + b1 s_act) - exact
// for a power of two - so neither operand needs a multiply in the K loop; emb = acc * inv + b2.
template <bool TIMING, bool BPL, bool F16 = false>   // TIMING (DC_DEV_TIMING build): s_memtime phase sums of wave 0 of every workgroup -> dbg[wg][4]
__global__ __launch_bounds__(256, 2) void embed_fwd_fused_kernel(const float* __restrict__ obs, const float* __restrict__ W1,
                                                                 const float* __restrict__ b1, const float* __restrict__ W2,
                                                                 const uint16_t* __restrict__ W2p,
                                                                 const float* __restrict__ b2, float* __restrict__ emb,
                                                                 float* __restrict__ xcat, uint8_t* __restrict__ amax,
                                                                 EmbTypes ty, int n_tiles, long long* __restrict__ dbg, float s_act, float inv,
                                                                 const uint8_t* __restrict__ umask, const float* __restrict__ Wenv,
                                                                 const float* __restrict__ benv) {
    static_assert(!F16 || BPL, "the f16 pieces need the pre-split weight planes");
    long long tm_k = 0, tm_e = 0, tm_b = 0, tm0 = 0, tm_start = 0;
    if constexpr (TIMING) tm_start = __builtin_amdgcn_s_memtime();
    using LT = FastTile<128, false>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using PT = PlaneTile<128, F16 ? 2 : 3>;
    float* stage = smem;               // 2 x (A [128][32] f32 | B [128][32] f32, or three bf16 / two f16 planes of it)
    constexpr int STAGE_FL = 4096 + (BPL ? PT::LDS_FLOATS : 4096);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fq = lane >> 5;

    // The first layer (K = 12) is itself a matrix product and runs on the matrix cores: per K step of the
    // second layer, wave w produces basic[rows 32w..32w+31][channels 32kt..32kt+31] with six
    // v_mfma_f32_32x32x2_f32 (A = unit records, B = W1 rows, both one register per lane per instruction),
    // adds b1, applies the relu and writes the 16 accumulator values into the swizzled A image of the
    // second product.
    //   A operand of MFMA kk: lane (fr, fq) = x[row 32w + fr][feature 2kk + fq]
    //   B operand          : lane (fr, fq) = W1[channel 32kt + fr][feature 2kk + fq]
    //   D                  : lane (fr, fq), register r = basic[row 32w + 8(r>>2) + 4fq + (r&3)][channel 32kt + fr]
    // F16: the first layer too runs on the 16-bit matrix cores - K = 12 padded to 16 is ONE K step: four v_mfma_f32_32x32x16_f16 on
    // two f16 pieces of the records (x 2^4, split once per tile when they are loaded) and of W1 (x 2^8, as two planes [128][16] in LDS,
    // built once per workgroup) instead of six v_mfma_f32_32x32x2_f32 - 128 matrix-pipe cycles instead of 384 per 32 x 32 block, next
    // to the 1 024 of the block's second-layer products.  (The backward kernels re-evaluate the relu mask with the exact-f32 MFMA:
    // a pre-activation within ~1e-7 of zero may get the other sign there - as between any two implementations of this layer.)
    float wb[F16 ? 1 : 4][F16 ? 1 : 6], b1v[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        if constexpr (!F16) {
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) wb[kt][kk] = W1[(32 * kt + fr) * 12 + 2 * kk + fq];
        }
        b1v[kt] = b1[32 * kt + fr] * (F16 ? s_act : 1.f);
    }
    char* const w1p = reinterpret_cast<char*>(smem + 2 * STAGE_FL);     // F16: [2 planes][128 channels][16 features] f16
    if constexpr (F16) {
        const int ch = tid >> 1, f0 = 8 * (tid & 1);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f0 + e < 12 ? W1[ch * 12 + f0 + e] : 0.f;
        const Split2h sp = split2h<true>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), F16_S_W1);
        *reinterpret_cast<f16x8*>(w1p + ch * 32 + f0 * 2) = sp.h;
        *reinterpret_cast<f16x8*>(w1p + 4096 + ch * 32 + f0 * 2) = sp.m;
    }
    const float ginv = s_act / (s_act * F16_S_W1);     // (x s_act)(W1 s_w1) -> basic * s_act
    int aoff[16];   // LDS float index of D register r inside an A stage (row-major [128][32], 16-byte chunks XOR-swizzled)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * wave + 8 * (r >> 2) + 4 * fq + (r & 3);
        aoff[r] = row * GEMM_BK + 4 * ((fr >> 2) ^ ((row >> 1) & 7)) + (fr & 3);
    }
    size_t offb[LT::NI];
    LT::src_offsets<false>(offb, EF_EMB, 0, 128, wave, lane);
    size_t offp[PT::NI];
    PT::src_offsets(offp, EF_EMB, (long long)6 * EF_EMB * EF_EMB, 0, wave, lane);
    auto issue_b = [&](int t, int kt, float* dst) {      // W2_t, k in [32 kt, 32 kt + 32) -> the B half of a stage
        if constexpr (BPL) PT::issue(W2p + (size_t)t * EF_EMB * EF_EMB + kt * GEMM_BK, offp, dst, wave);
        else LT::issue(W2 + (size_t)t * EF_EMB * EF_EMB + kt * GEMM_BK, offb, dst, wave);
    };

    // the records of a tile in the registers of its consumer: six floats (f32 MFMA: features 2 kk + fq), or - F16 - the eight features
    // 8 fq .. 8 fq + 7 (12 .. 15: zero) already split into two f16 pieces (floats 0..3 = h, 4..7 = m as bit patterns)
    constexpr int XR = F16 ? 8 : 6;
    auto load_x = [&](int tile, float (&x)[XR]) {
        const int tl = min(tile, n_tiles - 1);                      // past the end: a valid tile, never used
        const int tt = ef_ftype_of_tile(ty, tl);
        if constexpr (F16) {
            const float* xp = ef_record(obs, tt, ef_frow(ty, tt, tl, 32 * wave + fr) - ty.row_begin[tt], ty.nr_valid) + 8 * fq;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = xp[e];
#pragma unroll
            for (int e = 4; e < 8; ++e) v[e] = fq ? 0.f : xp[e];      // features 12..15 do not exist
            const Split2h sp = split2h<true>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), s_act);
            const u32x4 hh = __builtin_bit_cast(u32x4, sp.h), mm = __builtin_bit_cast(u32x4, sp.m);
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = __uint_as_float(hh[e]); x[4 + e] = __uint_as_float(mm[e]); }
        } else {
            const float* xp = ef_record(obs, tt, ef_frow(ty, tt, tl, 32 * wave + fr) - ty.row_begin[tt], ty.nr_valid) + fq;
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) x[kk] = xp[2 * kk];
        }
    };
    auto gen_a = [&](auto KT, const float (&x)[XR], float* a_s) {
        constexpr int kt = decltype(KT)::value;
        f32x16 g;
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = 0.f;
        if constexpr (F16) {
            const f16x8 xh = __builtin_bit_cast(f16x8, u32x4{__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])});
            const f16x8 xm = __builtin_bit_cast(f16x8, u32x4{__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7])});
            const f16x8 wh = *reinterpret_cast<const f16x8*>(w1p + (32 * kt + fr) * 32 + fq * 16);
            const f16x8 wm = *reinterpret_cast<const f16x8*>(w1p + 4096 + (32 * kt + fr) * 32 + fq * 16);
            DC_X2H_MM(g = __builtin_amdgcn_mfma_f32_32x32x16_f16(xm, wm, g, 0, 0, 0);)
            g = __builtin_amdgcn_mfma_f32_32x32x16_f16(xm, wh, g, 0, 0, 0);
            g = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wm, g, 0, 0, 0);
            g = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, g, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) a_s[aoff[r]] = relu_nan(fmaf(g[r], ginv, b1v[kt]));      // NaN-propagating: common.h
        } else {
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) g = __builtin_amdgcn_mfma_f32_32x32x2f32(x[kk], wb[kt][kk], g, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) a_s[aoff[r]] = relu_nan(g[r] + b1v[kt]);
        }
    };

    int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    float xa[XR], xn[XR];
    load_x(tile, xa);
    if constexpr (F16) __syncthreads();            // the W1 planes
    {   // step-0 operands of the first tile
        const int t0 = ef_ftype_of_tile(ty, tile);
        issue_b(t0, 0, stage + 4096);
        gen_a(std::integral_constant<int, 0>{}, xa, stage);
    }
    __syncthreads();

    for (; tile < n_tiles; tile += gridDim.x) {
        const int t = ef_ftype_of_tile(ty, tile);
        const bool p5 = F16 && ty.pool5 && t == 1;                         // this tile: 24 whole steps of the five-unit type
        const long long row0 = ef_frow(ty, t, tile, 0);                    // (p5: of block 0 only - see ef_frow)
        const int ntile = tile + gridDim.x;
        const bool more = ntile < n_tiles;
        load_x(ntile, xn);              // lands behind the K loop
        const float b2v[2] = {b2[t * EF_EMB + wn * 64 + fr], b2[t * EF_EMB + wn * 64 + 32 + fr]};   // epilogue operands: likewise
        // Mask-aware emb stores (F16 variant - the others have no register to spare): the emb rows of the two 16-unit types are read by
        // the target-unit attention only, and only for units whose mask byte is set (heads.hip: attn_logits_masked, attn_bwd_q via dtu).
        // The 64 rows of this wave = four env-steps x sixteen units: lane l takes the byte of (step l >> 4, unit l & 15) of each half.
        int mb[2] = {1, 1};
        if constexpr (F16) {
            if (p5) {
                // the five-unit type pooled here (pool5): its rows too are read by the attention only.  Row l < 30 of half i = unit l % 5
                // of step 24 tile + 6 (2 wm + i) + l / 5; rows 30, 31 and steps past the end are never stored
                const long long st0 = (long long)(tile - ty.ftile_begin[1]) * EF_P5_STEPS;
                const long long nsteps = (ty.row_begin[2] - ty.row_begin[1]) / 5;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int l = lane & 31, sl = (l * 13) >> 6;
                    const long long n = st0 + 6 * (2 * wm + i) + sl;
                    mb[i] = (l < 30 && n < nsteps) ? (umask != nullptr ? umask[min(n, ty.nr_valid - 1) * 65 + 22 + 1 + (l - 5 * sl)] : 1) : 0;
                }
            } else if (umask != nullptr && (t == 2 || t == 3)) {
                const long long lr = row0 - ty.row_begin[t] + wm * 64;            // type-local row of this wave's row 0: a multiple of 16
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const long long n = min((lr >> 4) + 2 * i + ((lane >> 4) & 1), ty.nr_valid - 1);
                    mb[i] = umask[n * 65 + 22 + ef_cum(t) + (lane & 15)];
                }
            } else if (umask != nullptr && t != 1) {
                // the three one-unit types (own hero - never targetable, policy.py:255 - and the two towers): their pooled value is taken from
                // the accumulators; the row itself is read by the attention only.  The 64 rows of this wave = 64 env-steps: lane l of half i
                // takes the byte of step 32 i + (l & 31).  (The five-unit type keeps all rows: pool_env_fwd pools it from emb.)
                const long long lr = row0 - ty.row_begin[t] + wm * 64;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const long long n = min(lr + 32 * i + (lane & 31), ty.nr_valid - 1);
                    mb[i] = umask[n * 65 + 22 + ef_cum(t)];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TIMING) tm0 = __builtin_amdgcn_s_memtime();

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            float* cur = stage + (kt & 1) * STAGE_FL;
            float* nxt = stage + ((kt + 1) & 1) * STAGE_FL;
            if (kt < 3) {
                issue_b(t, kt + 1, nxt + 4096);
                if (kt == 0) gen_a(std::integral_constant<int, 1>{}, xa, nxt);
                else if (kt == 1) gen_a(std::integral_constant<int, 2>{}, xa, nxt);
                else gen_a(std::integral_constant<int, 3>{}, xa, nxt);
            } else if (more) {          // the next tile's step 0, into the buffer step 3 does not read
                const int t1 = ef_ftype_of_tile(ty, ntile);
                issue_b(t1, 0, nxt + 4096);
                gen_a(std::integral_constant<int, 0>{}, xn, nxt);
            }
            if constexpr (F16) mma_kstep_bplanes_h<LT, 128, 2, 2, false>(cur, cur + 4096, wm * 64, wn * 64, fr, fq, acc, 1.f);
            else if constexpr (BPL) mma_kstep_bplanes<LT, 128, 2, 2>(cur, cur + 4096, wm * 64, wn * 64, fr, fq, acc);
            else mma_kstep<LT, LT, 2, 2>(cur, cur + 4096, wm * 64, wn * 64, fr, fq, acc);
            __syncthreads();
        }

        // ---- epilogue.  emb = acc + b2; the max-pool over the units of an env-step (policy.py:102-127) is
        // taken straight from the accumulators (no 335 MB re-read of emb by a pooling kernel).  A 32x32
        // accumulator tile holds rows 8*(r>>2) + 4*fq + (r&3): the 16 units of a step (types anh/enh, U = 16,
        // tiles start on step boundaries) are 8 registers of this lane + 8 of lane^32.  "First maximum wins"
        // like torch.max.  Types with one unit (ah, ath) copy through; eh (U = 5, steps straddle tiles) and
        // the env embedding are left to pool_env_fwd's residual pass; eth is never pooled (policy.py:127
        // pools enh twice instead).
        // emb itself goes through LDS (stage buffer 1: step 3 is done with it, the next tile's step 0 sits in
        // buffer 0): each wave transposes its 64x64 block in two 32-row halves of 8 KB and stores 16 bytes
        // per lane, 256 contiguous bytes per 16 lanes.
        if constexpr (TIMING) { const long long x = __builtin_amdgcn_s_memtime(); tm_k += x - tm0; tm0 = x; }
        const long long lrow0 = row0 - ty.row_begin[t];
        float* tr = stage + STAGE_FL + wave * 2048;     // [32][64]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = wn * 64 + j * 32 + fr;
                const float bv = b2v[j];
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v[r] = (F16 ? acc[i][j][r] * inv : acc[i][j][r]) + bv;
                    tr[(8 * (r >> 2) + 4 * fq + (r & 3)) * 64 + j * 32 + fr] = v[r];
                }
                if (xcat == nullptr) continue;
                const long long lr = lrow0 + wm * 64 + i * 32;           // type-local row of this tile's row 0
                if (t == 2 || t == 3) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {                         // rows 16h..16h+15 = one env-step
                        float m = v[8 * h];
                        int am = 4 * fq;
#pragma unroll
                        for (int rr = 1; rr < 8; ++rr) {
                            const float x = v[8 * h + rr];
                            const int u = (rr & 3) + 8 * (rr >> 2) + 4 * fq;
                            am = x > m ? u : am;
                            m = max_nan(m, x);                            // a NaN unit makes the pooled value NaN, like torch.max
                        }
                        // partner lane^32 through v_permlane32_swap (VALU; a ds_bpermute costs an LDS round trip):
                        // swap(x, x) = {[x.low | x.low], [x.high | x.high]} -> the other half's value is r[1 - fq]
                        const auto sm = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                        const auto sa = __builtin_amdgcn_permlane32_swap((unsigned)am, (unsigned)am, false, false);
                        const float pm = __uint_as_float(fq ? sm[0] : sm[1]);
                        const int pam = (int)(fq ? sa[0] : sa[1]);
                        if (pm > m || (pm == m && pam < am)) am = pam;
                        m = max_nan(m, pm);
                        if (fq == h) {
                            const long long n = (lr >> 4) + h;
                            float* xo = xcat + n * 896;
                            xo[(1 + t) * EF_EMB + col] = m;
                            if (t == 3) xo[6 * EF_EMB + col] = m;         // policy.py:127: the "eth" slot holds the enh max
                            amax[(n * 3 + (t - 1)) * EF_EMB + col] = (uint8_t)am;
                        }
                    }
                } else if (t == 0 || t == 4) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        xcat[(lr + (r & 3) + 8 * (r >> 2) + 4 * fq) * 896 + (1 + t) * EF_EMB + col] = v[r];
                    if (F16 && ty.pool5 && t == 0) {
                        // the env embedding (policy.py:97) rides on the own-hero tiles (a row = an env-step): pool_env_fwd's arithmetic
                        const float w0 = Wenv[col * 3 + 0], w1 = Wenv[col * 3 + 1], w2 = Wenv[col * 3 + 2], be = benv[col];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const long long n = lr + (r & 3) + 8 * (r >> 2) + 4 * fq;
                            const float* e = obs + min(n, ty.nr_valid - 1) * EF_OBS;
                            xcat[n * 896 + col] = relu_nan(fmaf(e[2], w2, fmaf(e[1], w1, fmaf(e[0], w0, be))));
                        }
                    }
                }
            }
            if (p5 && xcat != nullptr) {
                // max over the five units of a step, from the block's [32][64] image (this wave wrote it; LDS ops of a wave are in order):
                // lane = column, six steps; "first maximum wins" and NaN propagation like pool_env_fwd / torch.max
                const long long st0 = (long long)(tile - ty.ftile_begin[1]) * EF_P5_STEPS + 6 * (2 * wm + i);
                const long long nsteps = (ty.row_begin[2] - ty.row_begin[1]) / 5;
#pragma unroll
                for (int sl = 0; sl < 6; ++sl) {
                    float m = tr[(5 * sl) * 64 + lane];
                    int am = 0;
#pragma unroll
                    for (int u = 1; u < 5; ++u) {
                        const float x = tr[(5 * sl + u) * 64 + lane];
                        am = x > m ? u : am;
                        m = max_nan(m, x);
                    }
                    const long long n = st0 + sl;
                    if (n < nsteps) {
                        xcat[n * 896 + 2 * EF_EMB + wn * 64 + lane] = m;
                        amax[(n * 3 + 0) * EF_EMB + wn * 64 + lane] = (uint8_t)am;
                    }
                }
            }
            // rows 4*it + lane/16 of the half, 16 bytes at column 4*(lane%16): one wave, LDS ops in order
            float* eo = emb + (size_t)(p5 ? row0 + 30 * (2 * wm + i) : row0 + wm * 64 + i * 32) * EF_EMB + wn * 64;
            const unsigned live = F16 ? (unsigned)__ballot(mb[i] != 0) : 0xffffffffu;      // bit rr: row rr of this half is read by someone
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int rr = 4 * it + (lane >> 4), c4 = lane & 15;
                const float4 q = *reinterpret_cast<const float4*>(tr + rr * 64 + 4 * c4);
                // (non-temporal: the rows are read by later kernels only; measured round 5: this kernel unchanged, pool_env_fwd behind it 63 -> 53 us)
                typedef __attribute__((ext_vector_type(4))) float ef_f32x4;
                if (!F16 || ((live >> rr) & 1u)) __builtin_nontemporal_store(ef_f32x4{q.x, q.y, q.z, q.w}, reinterpret_cast<ef_f32x4*>(eo + (size_t)rr * EF_EMB + 4 * c4));
            }
        }
        if constexpr (TIMING) { const long long x = __builtin_amdgcn_s_memtime(); tm_e += x - tm0; tm0 = x; }
        __syncthreads();            // buffer 1 is free again for step 1 of the next tile
        if constexpr (TIMING) { const long long x = __builtin_amdgcn_s_memtime(); tm_b += x - tm0; }
#pragma unroll
        for (int kk = 0; kk < XR; ++kk) xa[kk] = xn[kk];
    }
    if constexpr (TIMING) {
        if (tid == 0) {
            dbg[blockIdx.x * 4 + 0] = tm_k; dbg[blockIdx.x * 4 + 1] = tm_e; dbg[blockIdx.x * 4 + 2] = tm_b;
            dbg[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime() - tm_start;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward 1: dW2_t = demb_t^T basic_t.  Workgroup = (type, range of 32-row K steps), full 128x128
// output -> slab[workgroup][128][128] (reduced per type by splitk_reduce).  A = demb rows by DMA
// (k-major), B = basic generated k-major: thread (channel c, row half) with W1[c] in registers and the
// unit records read through wave-uniform addresses.
// ---------------------------------------------------------------------------------------------------
template <bool F16>     // F16: A = d(emb) scaled by s_grad at the split, B = basic generated pre-scaled by s_act (through W1, b1), slab = acc * inv
__global__ __launch_bounds__(256) void embed_bwd_dw2_kernel(const float* __restrict__ obs, const float* __restrict__ demb,
                                                            const float* __restrict__ W1, const float* __restrict__ b1,
                                                            float* __restrict__ slab, EmbTypes ty, float s_grad, float s_act, float inv) {
    using LT = FastTile<128, true>;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 x (A [32][128] | B [32][128])
    constexpr int STAGE_FL = 2 * 4096;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fq = lane >> 5;
    const int wg = blockIdx.x;
    int t = 0;
#pragma unroll
    for (int i = 1; i < 6; ++i)
        if (wg >= ty.wg_begin[i]) t = i;
    if (ty.sparse16 && (t == 2 || t == 3)) return;      // slab written by embed_bwd_pool16
    const long long steps_t = (ty.row_begin[t + 1] - ty.row_begin[t]) / GEMM_BK;
    const long long s0 = (long long)(wg - ty.wg_begin[t]) * ty.steps_per_wg;
    const int ns = (int)min((long long)ty.steps_per_wg, steps_t - s0);

    // B operand (basic, k-major [32 rows][128 channels]) on the matrix cores, like the forward: wave w makes
    // channels 32w..32w+31 of the K step's 32 rows with six MFMAs.  B of those = W1 (six registers for the whole
    // kernel); A = the step's unit records, one register per MFMA, loaded two steps ahead straight from HBM/L2
    // (each wave its own copy - 1.5 KB per step).
    // F16: like the forward, the regenerated first layer is one K = 16 step of three f16 MFMAs (records x 2^4 split per step, this wave's
    // 32 W1 rows x 2^8 split once into registers)
    float wb[F16 ? 1 : 6];
    f16x8 w1h, w1m;
    if constexpr (F16) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 8 * fq + e < 12 ? W1[(32 * wave + fr) * 12 + 8 * fq + e] : 0.f;
        const Split2h sp = split2h<true>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), F16_S_W1);
        w1h = sp.h; w1m = sp.m;
    } else {
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) wb[kk] = W1[(32 * wave + fr) * 12 + 2 * kk + fq];
    }
    const float b1v = b1[32 * wave + fr] * (F16 ? s_act : 1.f);
    const float ginv = 1.f / F16_S_W1;                 // (x s_act)(W1 s_w1) -> basic * s_act
    int boff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) boff[r] = (8 * (r >> 2) + 4 * fq + (r & 3)) * 128 + 32 * wave + fr;

    size_t offa[LT::NI];
    LT::src_offsets<false>(offa, EF_EMB, 0, 128, wave, lane);
    const float* ga = demb + (size_t)(ty.row_begin[t] + s0 * GEMM_BK) * EF_EMB;

    constexpr int XR = F16 ? 8 : 6;
    auto load_x = [&](int s, float (&x)[XR]) {
        const float* xp = ef_record(obs, t, (s0 + min(s, ns - 1)) * GEMM_BK + fr, ty.nr_valid) + (F16 ? 8 * fq : fq);   // clamped: always a valid step
        if constexpr (F16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = xp[e];
#pragma unroll
            for (int e = 4; e < 8; ++e) x[e] = fq ? 0.f : xp[e];      // features 12..15 do not exist
        } else {
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) x[kk] = xp[2 * kk];
        }
    };
    auto gen_b = [&](const float (&x)[XR], float* b_s) {
        f32x16 g;
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = 0.f;
        if constexpr (F16) {
            const Split2h sp = split2h<true>(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]), s_act);
            DC_X2H_MM(g = __builtin_amdgcn_mfma_f32_32x32x16_f16(sp.m, w1m, g, 0, 0, 0);)
            g = __builtin_amdgcn_mfma_f32_32x32x16_f16(sp.m, w1h, g, 0, 0, 0);
            g = __builtin_amdgcn_mfma_f32_32x32x16_f16(sp.h, w1m, g, 0, 0, 0);
            g = __builtin_amdgcn_mfma_f32_32x32x16_f16(sp.h, w1h, g, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) b_s[boff[r]] = fmaxf(fmaf(g[r], ginv, b1v), 0.f);
        } else {
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) g = __builtin_amdgcn_mfma_f32_32x32x2f32(x[kk], wb[kk], g, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) b_s[boff[r]] = fmaxf(g[r] + b1v, 0.f);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float x0[XR], x1[XR];
    load_x(0, x0);
    load_x(1, x1);
    LT::issue(ga, offa, smem, wave);
    gen_b(x0, smem + 4096);
    __syncthreads();
    // steps unrolled by two with ping-pong record registers: x1 feeds gen_b(s+1) while x0 receives step s+2
    auto step = [&](int s, float (&xnext)[XR], float (&xfill)[XR]) {
        float* cur = smem + (s & 1) * STAGE_FL;
        // The barrier at the end of a step waits for the A DMA (vmcnt retires in order), hence for every load
        // of the step: the record loads go FIRST, so that they complete behind the 64 MFMAs.
        load_x(s + 2, xfill);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < ns) {
            float* nxt = smem + ((s + 1) & 1) * STAGE_FL;
            LT::issue(ga + (size_t)(s + 1) * GEMM_BK * EF_EMB, offa, nxt, wave);
            gen_b(xnext, nxt + 4096);
        }
        if constexpr (F16) mma_kstep_h<LT, LT, 2, 2, true, false>(cur, cur + 4096, wm * 64, wn * 64, fr, fq, acc, s_grad, 1.f);
        else mma_kstep<LT, LT, 2, 2>(cur, cur + 4096, wm * 64, wn * 64, fr, fq, acc);
        __syncthreads();
    };
    for (int s = 0; s < ns; s += 2) {
        step(s, x1, x0);
        if (s + 1 < ns) step(s + 1, x0, x1);
    }
    float* out = slab + (size_t)wg * EF_EMB * EF_EMB;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* c = out + (size_t)(wm * 64 + i * 32 + 4 * fq) * EF_EMB + wn * 64 + j * 32 + fr;
#pragma unroll
            for (int r = 0; r < 16; ++r) c[(size_t)((r & 3) + 8 * (r >> 2)) * EF_EMB] = F16 ? acc[i][j][r] * inv : acc[i][j][r];
        }
}

// ---------------------------------------------------------------------------------------------------
// backward 2: d(basic) = (demb_t W2_t) * [basic > 0] -> dW1 / db1, d(basic) never stored.
// Persistent workgroups stride over the 128-row tiles; main loop = the NN fast GEMM (A = demb rows by
// DMA, B = W2_t k-major by DMA); the epilogue recomputes basic for the mask and accumulates
// dW1[c][f] += g x[f], db1[c] += g in registers across tiles.  Output: partials[workgroup][13][128]
// (12 features + bias), summed by unit_basic_reduce (embed.hip).
// ---------------------------------------------------------------------------------------------------
template <bool F16>     // F16: A = d(emb) scaled by s_grad, B = W2 (f32 tile) scaled by s_w, both at the split; d(basic) = acc * inv
__global__ __launch_bounds__(256, 2) void embed_bwd_dw1_kernel(const float* __restrict__ obs, const float* __restrict__ demb,
                                                            const float* __restrict__ W1, const float* __restrict__ b1,
                                                            const float* __restrict__ W2, float* __restrict__ partials,
                                                            EmbTypes ty, int n_tiles, float s_grad, float s_w, float inv) {
    using LA = FastTile<128, false>;
    using LB = FastTile<128, true>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;              // [128][12] unit records of the tile
    float* stage = smem + 1536;    // 2 x (A [128][32] | B [32][128])
    constexpr int STAGE_FL = 2 * 4096;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fq = lane >> 5;

    // relu mask: basic is re-evaluated on the matrix cores in the accumulators' own layout (six MFMAs per 32x32
    // tile: A = unit records from xs, B = W1 in twelve registers for the whole kernel) - bitwise the forward's
    float wb[2][6], bias[2], dw[2][12], db[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = wn * 64 + j * 32 + fr;
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) wb[j][kk] = W1[c * 12 + 2 * kk + fq];
#pragma unroll
        for (int f = 0; f < 12; ++f) dw[j][f] = 0.f;
        bias[j] = b1[c];
        db[j] = 0.f;
    }
    size_t offa[LA::NI], offb[LB::NI];
    LA::src_offsets<false>(offa, EF_EMB, 0, 128, wave, lane);
    LB::src_offsets<false>(offb, EF_EMB, 0, 128, wave, lane);

    // sparse16: only the tiles of the four small types; virtual tile v -> the v-th tile outside [tile_begin[2], tile_begin[4])
    const int skip_lo = ty.sparse16 ? ty.tile_begin[2] : n_tiles, skip_n = ty.sparse16 ? ty.tile_begin[4] - ty.tile_begin[2] : 0;
    for (int vt = blockIdx.x; vt < n_tiles - skip_n; vt += gridDim.x) {
        const int tile = vt < skip_lo ? vt : vt + skip_n;
        const int t = ef_type_of_tile(ty, tile);
        const long long row0 = (long long)tile * EF_TILE;
        const float* ga = demb + (size_t)row0 * EF_EMB;
        const float* gb = W2 + (size_t)t * EF_EMB * EF_EMB;
        __syncthreads();   // previous tile's epilogue is done with xs / the stage buffers
        for (int e = tid; e < 1536; e += 256) {
            const int r = e / 12, f = e - r * 12;
            xs[e] = ef_record(obs, t, row0 + r - ty.row_begin[t], ty.nr_valid)[f];
        }
        // d(emb) is read once; W2 of the type by every tile: as ordinary loads the 268 MB stream pushed W2 out of the L2 between two tiles
        // (FETCH_SIZE: 654 MB per launch = d(emb) + records + 64 KB of W2 PER TILE) - hence non-temporal (DC_DW1_NT=0: ordinary, A/B)
#ifndef DC_DW1_NT
#define DC_DW1_NT 2
#endif
        LA::template issue<DC_DW1_NT>(ga, offa, stage, wave);
        LB::issue(gb, offb, stage + 4096, wave);
        __syncthreads();

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            float* cur = stage + (kt & 1) * STAGE_FL;
            if (kt < 3) {
                float* nxt = stage + ((kt + 1) & 1) * STAGE_FL;
                LA::template issue<DC_DW1_NT>(ga + (kt + 1) * GEMM_BK, offa, nxt, wave);
                LB::issue(gb + (size_t)(kt + 1) * GEMM_BK * EF_EMB, offb, nxt + 4096, wave);
            }
            if constexpr (F16) mma_kstep_h<LA, LB, 2, 2, true, true>(cur, cur + 4096, wm * 64, wn * 64, fr, fq, acc, s_grad, s_w);
            else mma_kstep<LA, LB, 2, 2>(cur, cur + 4096, wm * 64, wn * 64, fr, fq, acc);
            __syncthreads();
        }
        // epilogue: rows of this lane = wm*64 + i*32 + 4*fq + (r&3) + 8*(r>>2)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x16 bs[2];
            {
                const float* xr = xs + (wm * 64 + i * 32 + fr) * 12 + fq;
                float xa[6];
#pragma unroll
                for (int kk = 0; kk < 6; ++kk) xa[kk] = xr[2 * kk];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) bs[j][r] = 0.f;
#pragma unroll
                    for (int kk = 0; kk < 6; ++kk) bs[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[kk], wb[j][kk], bs[j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 64 + i * 32 + 4 * fq + (r & 3) + 8 * (r >> 2);
                const float4* xp = reinterpret_cast<const float4*>(xs + rl * 12);
                const float4 xa = xp[0], xb = xp[1], xc = xp[2];
                const float x[12] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w, xc.x, xc.y, xc.z, xc.w};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float g = (bs[j][r] + bias[j]) > 0.f ? (F16 ? acc[i][j][r] * inv : acc[i][j][r]) : 0.f;
#pragma unroll
                    for (int f = 0; f < 12; ++f) dw[j][f] = fmaf(g, x[f], dw[j][f]);
                    db[j] += g;
                }
            }
        }
    }
    // combine the 4 partial sums per column (2 lane halves x 2 row waves) through LDS
    __syncthreads();
    float* red = smem;   // [4][13][128]
    {
        const int slot = wm * 2 + fq;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = wn * 64 + j * 32 + fr;
#pragma unroll
            for (int f = 0; f < 12; ++f) red[(slot * 13 + f) * 128 + c] = dw[j][f];
            red[(slot * 13 + 12) * 128 + c] = db[j];
        }
    }
    __syncthreads();
    float* o = partials + (size_t)blockIdx.x * 1664;
    for (int e = tid; e < 1664; e += 256) o[e] = (red[e] + red[1664 + e]) + (red[2 * 1664 + e] + red[3 * 1664 + e]);
}

// ---------------------------------------------------------------------------------------------------
static const int H_UNITS[6] = {1, 5, 16, 16, 1, 1};
static const int H_CUM[7] = {0, 1, 6, 22, 38, 39, 40};

bool embed_fused_supported(long long nr) { return nr > 0 && nr % 128 == 0; }   // nr: the PADDED step count

enum { SPARSE_WG_PER_TYPE = 128 };   // embed_bwd_pool16: one workgroup per CU over the two 16-unit types

static EmbTypes make_types(long long nr, long long nr_valid, int* total_wg, bool sparse16 = false) {
    EmbTypes ty;
    ty.sparse16 = sparse16 ? 1 : 0;
    ty.nr_valid = nr_valid;
    for (int t = 0; t <= 6; ++t) {
        ty.row_begin[t] = nr * H_CUM[t];
        ty.tile_begin[t] = (int)(ty.row_begin[t] / EF_TILE);
    }
    // dense split-K workgroups: 448 over the types that take the dense path (sparse16: the 8 units of the small types)
    const long long total_steps = nr * (sparse16 ? 8 : 40) / GEMM_BK;
    int spw = (int)((total_steps + 447) / 448);
    if (spw < 4) spw = 4;
    ty.steps_per_wg = spw;
    int wg = 0;
    for (int t = 0; t < 6; ++t) {
        ty.wg_begin[t] = wg;
        if (sparse16 && (t == 2 || t == 3)) { wg += SPARSE_WG_PER_TYPE; continue; }
        const long long st = nr * H_UNITS[t] / GEMM_BK;
        wg += (int)((st + spw - 1) / spw);
    }
    ty.wg_begin[6] = wg;
    *total_wg = wg;
    for (int t = 0; t <= 6; ++t) ty.ftile_begin[t] = ty.tile_begin[t];
    ty.pool5 = 0;
    return ty;
}

template <class K>
static int set_lds(K kernel, size_t bytes, bool* done) {
    if (*done) return 0;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { set_error("embed_fused: hipFuncSetAttribute", (int)e); return (int)e; }
    *done = true;
    return 0;
}

int embed_fwd_fused(const float* obs, const float* W1, const float* b1, const float* W2, const uint16_t* W2p, const float* b2, float* emb,
                    float* xcat, uint8_t* amax, long long nr_valid, long long nr, hipStream_t s, F16x2Scales f16, const uint8_t* unit_mask,
                    const float* Wenv, const float* benv) {
    int nwg;
    EmbTypes ty = make_types(nr, nr_valid, &nwg);
    const bool bpl = W2p != nullptr;
    const bool h = f16.on && bpl;
    int tiles = (int)(nr * 40 / EF_TILE);
    if (Wenv != nullptr) {     // env embedding + the five-unit pool in this kernel too (embed_fwd_pools_all): 24-step tiles for that type
        if (!h || xcat == nullptr) { set_error("embed_fwd_fused: the in-kernel env / five-unit pooling needs the f16x2 variant", 1041); return 1041; }
        ty.pool5 = 1;
        const int t5 = (int)((nr + EF_P5_STEPS - 1) / EF_P5_STEPS), shift = t5 - (ty.tile_begin[2] - ty.tile_begin[1]);
        for (int t = 2; t <= 6; ++t) ty.ftile_begin[t] = ty.tile_begin[t] + shift;
        tiles += shift;
    }
    const size_t lds = (size_t)(2 * (4096 + (bpl ? (h ? PlaneTile<128, 2>::LDS_FLOATS : PlaneTile<128>::LDS_FLOATS) : 4096)) + (h ? 2048 : 0)) * sizeof(float);
    static bool attr = false, attr_p = false, attr_h = false;
    if (h) { if (int e = set_lds(embed_fwd_fused_kernel<false, true, true>, lds, &attr_h)) return e; }
    else if (bpl) { if (int e = set_lds(embed_fwd_fused_kernel<false, true>, lds, &attr_p)) return e; }
    else if (int e = set_lds(embed_fwd_fused_kernel<false, false>, lds, &attr)) return e;
    const int grid = tiles < 512 ? tiles : 512;      // two resident workgroups per CU
    ProfScope prof("embed_fwd_fused", 2.0 * nr * 40 * 128 * (128 + 12), 4.0 * nr * 40 * (12 + 128), s);
    constexpr bool timing = DC_DEV_TIMING != 0;
    if (timing) {   // debugging aid: mean per-tile phase cycles (wave 0 of each workgroup), printed per launch
        static bool attr2 = false;
        if (int e = set_lds(embed_fwd_fused_kernel<true, false>, (size_t)(4 * 4096) * sizeof(float), &attr2)) return e;
        static long long* dbg = nullptr;
        if (!dbg) (void)hipMalloc(&dbg, 512 * 4 * sizeof(long long));
        hipLaunchKernelGGL((embed_fwd_fused_kernel<true, false>), dim3(grid), dim3(256), (size_t)(4 * 4096) * sizeof(float), s, obs, W1, b1, W2,
                           (const uint16_t*)nullptr, b2, emb, xcat, amax, ty, tiles, dbg, 1.f, 1.f, (const uint8_t*)nullptr, (const float*)nullptr, (const float*)nullptr);
        long long h[512 * 4];
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        double ph[4] = {0, 0, 0, 0};
        for (int b = 0; b < grid; ++b)
            for (int i = 0; i < 4; ++i) ph[i] += (double)h[b * 4 + i];
        fprintf(stderr, "embed_fwd timing (cycles per tile, mean over %d workgroups): k-loop %.0f  epilogue %.0f  end barrier %.0f | workgroup life %.0f (%d tiles)\n",
                grid, ph[0] / tiles, ph[1] / tiles, ph[2] / tiles, ph[3] / grid, tiles / grid);
        return launch_check("embed_fwd_fused");
    }
    if (h) hipLaunchKernelGGL((embed_fwd_fused_kernel<false, true, true>), dim3(grid), dim3(256), lds, s, obs, W1, b1, W2, W2p, b2, emb, xcat, amax, ty,
                              tiles, (long long*)nullptr, f16.s_act, 1.f / (f16.s_act * f16.s_w), unit_mask, Wenv, benv);
    else if (bpl) hipLaunchKernelGGL((embed_fwd_fused_kernel<false, true>), dim3(grid), dim3(256), lds, s, obs, W1, b1, W2, W2p, b2, emb, xcat, amax, ty,
                                tiles, (long long*)nullptr, 1.f, 1.f, (const uint8_t*)nullptr, (const float*)nullptr, (const float*)nullptr);
    else hipLaunchKernelGGL((embed_fwd_fused_kernel<false, false>), dim3(grid), dim3(256), lds, s, obs, W1, b1, W2, W2p, b2, emb, xcat, amax, ty,
                            tiles, (long long*)nullptr, 1.f, 1.f, (const uint8_t*)nullptr, (const float*)nullptr, (const float*)nullptr);
    return launch_check("embed_fwd_fused");
}

// dW2 [6][128][128] (overwritten), dW1 [128][12] / db1 [128] (accumulated into, like unit_basic_bwd);
// scratch: >= max(total_wg * 16384, 512 * 1664) floats
// dW2 [6][128][128] (overwritten), dW1 [128][12] / db1 [128] (accumulated into, like unit_basic_bwd).
// sp != nullptr: the two 16-unit types go through embed_bwd_pool16 (their d(emb) rows need not exist in `demb`, and
// their second-layer bias gradients are ACCUMULATED into sp->db2 [6][128]); the four small types stay dense.
// scratch: >= (dense + sparse workgroups) * 16384 + sparse partials floats
int embed_bwd_fused(const float* obs, const float* demb, const float* W1, const float* b1, const float* W2, float* dW2,
                    float* dW1, float* db1, float* scratch, long long scratch_floats, long long nr_valid, long long nr,
                    const EmbSparseIn* sp, hipStream_t s, F16x2Scales f16) {
    int nwg;
    const bool sparse16 = sp != nullptr;
    EmbTypes ty = make_types(nr, nr_valid, &nwg, sparse16);
    const bool small_fused = embed_small_fused(sparse16, f16, sp);
    if (small_fused) {
        // embed_small.hip's 256 workgroups (32 per unit of a small type) around the 2 x 128 of the 16-unit types, in splitk_reduce_grouped's order
        const int wb[7] = {0, 32, 192, 192 + SPARSE_WG_PER_TYPE, 192 + 2 * SPARSE_WG_PER_TYPE, 224 + 2 * SPARSE_WG_PER_TYPE, 256 + 2 * SPARSE_WG_PER_TYPE};
        for (int t = 0; t <= 6; ++t) ty.wg_begin[t] = wb[t];
        nwg = wb[6];
    }
    const long long sp_floats = sparse16 ? 2LL * SPARSE_WG_PER_TYPE * (1664 + 128) : 0;
    if ((long long)nwg * EF_EMB * EF_EMB + sp_floats + (small_fused ? 256LL * (1664 + 128) : 0) > scratch_floats || 512LL * 1664 + sp_floats > scratch_floats) {
        set_error("embed_bwd_fused: scratch too small", 1040);
        return 1040;
    }
    float* part1 = scratch + scratch_floats - sp_floats;                 // sparse dW1/db1 partials, then db2 partials
    float* part2 = part1 + 2LL * SPARSE_WG_PER_TYPE * 1664;
    if (sparse16) {
        // two-f16-piece products: the dense form on the matrix cores with on-chip operands (embed_pool16m.hip); otherwise (bf16x3
        // products: f32's exponent range; A/B flags) the sparse form on the packed-f32 VALU (embed_sparse.hip).  Same partial formats.
        if (f16.on && !sp->valu && !sp->eight_waves) {
            if (int e = embed_bwd_pool16m(obs, sp->dxcat, sp->amax, sp->dtu, sp->q, sp->ldq, W1, b1, W2,
                                          scratch + (size_t)ty.wg_begin[2] * EF_EMB * EF_EMB, part1, part2, sp->prep, nr_valid, SPARSE_WG_PER_TYPE, s, f16,
                                          sp->w2t_planes))
                return e;
        } else if (int e = embed_bwd_pool16(obs, sp->dxcat, sp->amax, sp->dtu, sp->q, sp->ldq, W1, b1, W2,
                                            scratch + (size_t)ty.wg_begin[2] * EF_EMB * EF_EMB, part1, part2, sp->prep, nr_valid, SPARSE_WG_PER_TYPE, s, sp->eight_waves))
            return e;
    }
    if (small_fused) {
        // the four small types in one kernel with on-chip operands (embed_small.hip): d(emb) does not exist for any type on this path
        float* const part_small = scratch + (size_t)nwg * EF_EMB * EF_EMB;
        float* const db2_small = sp->small_db2 ? part_small + 256 * 1664 : nullptr;     // [256][128] column sums of demb
        if (int e = embed_bwd_small(obs, sp->dxcat, sp->amax, sp->dtu, sp->q, sp->ldq, W1, b1, W2, scratch, 2 * SPARSE_WG_PER_TYPE, part_small, db2_small,
                                    nr_valid, s, f16))
            return e;
        if (int e = splitk_reduce_grouped(scratch, dW2, EF_EMB, EF_EMB, 6, ty.wg_begin, s)) return e;
        return embed_tail_reduce(part_small, 256, part1, 2 * SPARSE_WG_PER_TYPE, dW1, db1, part2, SPARSE_WG_PER_TYPE, sp->db2 + 2 * 128, s, db2_small, sp->db2);
    }
    {
        const size_t lds = (size_t)(4 * 4096) * sizeof(float);
        static bool attr = false, attr_h = false;
        if (int e = f16.on ? set_lds(embed_bwd_dw2_kernel<true>, lds, &attr_h) : set_lds(embed_bwd_dw2_kernel<false>, lds, &attr)) return e;
        {
            const double units = sparse16 ? 8 : 40;
            ProfScope prof("embed_bwd_dw2", 2.0 * nr * units * 128 * (128 + 12), 4.0 * nr * units * (12 + 128), s);
            if (f16.on) hipLaunchKernelGGL(embed_bwd_dw2_kernel<true>, dim3(nwg), dim3(256), lds, s, obs, demb, W1, b1, scratch, ty, f16.s_grad, f16.s_act,
                                           1.f / (f16.s_grad * f16.s_act));
            else hipLaunchKernelGGL(embed_bwd_dw2_kernel<false>, dim3(nwg), dim3(256), lds, s, obs, demb, W1, b1, scratch, ty, 1.f, 1.f, 1.f);
        }
        if (int e = launch_check("embed_bwd_dw2")) return e;
        if (int e = splitk_reduce_grouped(scratch, dW2, EF_EMB, EF_EMB, 6, ty.wg_begin, s)) return e;   // one launch, six types
    }
    {
        const size_t lds = (size_t)(1536 + 4 * 4096) * sizeof(float);
        static bool attr = false, attr_h = false;
        if (int e = f16.on ? set_lds(embed_bwd_dw1_kernel<true>, lds, &attr_h) : set_lds(embed_bwd_dw1_kernel<false>, lds, &attr)) return e;
        const int tiles = (int)(nr * 40 / EF_TILE);
        const int dense_tiles = sparse16 ? tiles - (ty.tile_begin[4] - ty.tile_begin[2]) : tiles;
        const int grid = dense_tiles < 512 ? dense_tiles : 512;
        {
            const double units = sparse16 ? 8 : 40;
            ProfScope prof("embed_bwd_dw1", 2.0 * nr * units * 128 * (128 + 24), 4.0 * nr * units * (12 + 128), s);
            if (f16.on) hipLaunchKernelGGL(embed_bwd_dw1_kernel<true>, dim3(grid), dim3(256), lds, s, obs, demb, W1, b1, W2, scratch, ty, tiles, f16.s_grad,
                                           f16.s_w, 1.f / (f16.s_grad * f16.s_w));
            else hipLaunchKernelGGL(embed_bwd_dw1_kernel<false>, dim3(grid), dim3(256), lds, s, obs, demb, W1, b1, W2, scratch, ty, tiles, 1.f, 1.f, 1.f);
        }
        if (int e = launch_check("embed_bwd_dw1")) return e;
        // one launch: dW1/db1 from the dense and the sparse partials, db2 of types 2, 3 from the sparse bias partials
        if (sparse16) return embed_tail_reduce(scratch, grid, part1, 2 * SPARSE_WG_PER_TYPE, dW1, db1, part2, SPARSE_WG_PER_TYPE, sp->db2 + 2 * 128, s);
        return unit_basic_reduce(scratch, grid, dW1, db1, s);
    }
}

}  // namespace dc
