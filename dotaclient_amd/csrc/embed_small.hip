// Backward of the per-unit embedding MLP for the four SMALL unit types (1, 5, 1, 1 units per env-step: 8 of the 40), fused (round 6; the
// default with the two-f16-piece products).
//
// Replaces, for these types, the part of /root/reference/optimizer.py:672 (autograd) that flows through policy.py:100-105,118-127,152:
//     demb[u][c]   = [amax(c) == u] d(xcat)[slot][c] + dtu[u] q[c]        (max-pool routing + the attention keys' rank-one term)
//     dW2[c][k]   += sum_rows demb[row][c] basic[row][k]                  basic = relu(x W1^T + b1)
//     dbasic[row][k] = sum_c demb[row][c] W2[c][k], masked by basic > 0
//     dW1[k][f]   += sum_rows dbasic[row][k] x[row][f],  db1[k] += sum_rows dbasic[row][k]
// Until round 5 this was three launches around a d(emb) buffer in HBM: embed_scatter_bwd wrote the rows (268 MB), embed_bwd_dw2 and
// embed_bwd_dw1 (embed_fused.hip) read them back through the generic DMA tile and split every fragment into f16 pieces in every wave that
// used it - 0.16 / 0.12 of the matrix pipes' ceiling, the two kernels furthest below their roofline, 470 us per pass together.  Here, like
// embed_pool16m.hip for the 16-unit types, nothing of demb / basic / dbasic exists in memory, and every operand element is split ONCE:
//   tile = 64 unit rows of one type; workgroup = 4 waves, one per SIMD and CU (512 registers each: eight waves at 256 spilled ~300), 32
//   workgroups per unit of a type (the type's rows in contiguous ranges)
//   build : each thread forms an 8-row x 4-channel patch of demb from d(xcat), the arg-max bytes, dtu and q (loaded a tile ahead into
//           registers), scales by s_grad, splits into (h, m) and writes BOTH LDS images the two products want: [row][c] (A operand of
//           demb W2: eight channels per 16-byte read) and [c][row] (B operand of basic^T demb: eight rows per read)
//   P1    : basic of the wave's 32 hidden units, one 32 x 32 block per row half, on the f32 matrix core (six v_mfma_f32_32x32x2_f32: the relu mask is the dense kernels' and the
//           oracle's bit for bit, embed_pool16m.hip's header), mask kept as 16 bits, basic x s_act split once -> LDS image [k][row]
//   P3    : dbasic block = demb W2, K = 128 channels: A fragments from LDS, W2's fragments in 64 registers for the whole kernel (split once)
//   fold  : dW1^T / db1 += records^T x masked dbasic as a two-piece product too (K = the block's rows: the D registers as they lie are the
//           B operand), accumulator live for the whole kernel
//   P2    : dW2^T[k][c] += basic^T demb, K = the tile's 64 rows: the wave's 32 hidden units x all 128 channels (four blocks sharing the A
//           fragments), accumulators live for the whole kernel
// Per tile and wave 12 f32 + 108 f16 MFMAs; LDS 143 KB; three barriers.
// Register layouts as in embed_pool16m.hip: A[row fr][K slot 8 fq + j], B[K slot 8 fq + j][col fr], D register r = D[row 8 (r >> 2) + 4 fq + (r & 3)][col fr].
// Outputs in the dense kernels' formats: slab[workgroup][c][k] (splitk_reduce_grouped) and partials[workgroup][13][128] (embed_tail_reduce).
// The second-layer bias gradients d(b2_t) = the column sums of demb are taken here too (db2part; a pass of their own over d(xcat), q and dtu until
// the middle of round 6); only the env-embedding gradient stays with embed_scatter_bwd, which writes no d(emb) at all.
#include <stdio.h>
#include "kernels.h"
#include "gemm_tiles.h"

#ifndef ES_DB2
#define ES_DB2 1      // 0 (A/B build): no column sums of demb in the build phase (db2part then holds zeros)
#endif

namespace dc {
namespace {

enum { ES_THREADS = 256, ES_TILE = 64, ES_OBS = 483, ES_XCAT = 896,
       ES_CM_LD = 72,                      // halfs per channel of the [c][row] / [k][row] images: 64 rows + 8 = 144 B (16 lanes x 16 B: 64 banks)
       ES_RM_LD = 136,                     // halfs per row of the [row][c] image: 128 channels + 8 = 272 B
       ES_CM_PLANE = 128 * ES_CM_LD, ES_RM_PLANE = ES_TILE * ES_RM_LD };
enum : int {   // LDS (bytes)
    ES_DCM = 0,                            // demb [c][row], planes h | m
    ES_DRM = ES_DCM + 2 * ES_CM_PLANE * 2, // demb [row][c]
    ES_H1T = ES_DRM + 2 * ES_RM_PLANE * 2, // basic [k][row]
    ES_XS = ES_H1T + 2 * ES_CM_PLANE * 2,  // unit records [row][12] f32
    ES_W2M = ES_XS + ES_TILE * 12 * 4,     // W2_t's m plane [k][c] (its h plane lives in registers: both would be 64 per lane)
    ES_LDS = ES_W2M + 128 * ES_RM_LD * 2
};
static_assert(ES_LDS <= 160 * 1024, "LDS budget");

struct SmallArgs {
    const float* obs; const float* dxcat; const uint8_t* amax; const float* dtu; const float* q; int ldq;
    const float* W1; const float* b1; const float* W2;
    float* slab;              // [.][128][128]: workgroup g writes block g (g < 192) or g + skip (see embed_bwd_fused)
    int slab_skip;
    float* part;              // [256][13][128]
    float* db2part;           // [256][128] or NULL: the workgroup's column sums of demb = its share of the type's second-layer bias gradient
    long long nr;             // env-steps that exist (padding steps have no gradient: never visited)
    float s_act, s_w, s_grad;
};

__device__ __forceinline__ Split2h es_split8(const float (&v)[8], float s) {
    return split2h<true>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), s);
}

}  // namespace

__global__ __launch_bounds__(ES_THREADS, 1) void embed_small_bwd_kernel(SmallArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* const dcm = reinterpret_cast<_Float16*>(smem + ES_DCM);
    _Float16* const drm = reinterpret_cast<_Float16*>(smem + ES_DRM);
    _Float16* const h1t = reinterpret_cast<_Float16*>(smem + ES_H1T);
    float* const xs = reinterpret_cast<float*>(smem + ES_XS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int W = __builtin_amdgcn_readfirstlane(tid >> 6);      // P1 / P3 / fold: hidden units 32 W .. (both row blocks); P2: dW2^T rows 32 W .., all four channel blocks
    const int fr = lane & 31, fq = lane >> 5;
    const int g = blockIdx.x;
    const int us = g >> 5;                          // unit slot 0 .. 7: ah | eh x 5 | ath | eth
    const int t = us == 0 ? 0 : (us < 6 ? 1 : (us == 6 ? 4 : 5));
    const int U = t == 1 ? 5 : 1;
    const int cum = t == 0 ? 0 : (t == 1 ? 1 : (t == 4 ? 38 : 39));
    const int slot = t == 0 ? 1 : (t == 1 ? 2 : (t == 4 ? 5 : -1));      // xcat slot the type's max-pool feeds (policy.py:118-127; eth: none)
    const int widx = g - 32 * (t == 0 ? 0 : (t == 1 ? 1 : (t == 4 ? 6 : 7)));
    const int rows_t = (int)(p.nr * U);             // (rows of a type < 2^31: policy.hip check_dims)
    int per = (rows_t + 32 * U - 1) / (32 * U);
    per = (per + ES_TILE - 1) / ES_TILE * ES_TILE;
    const int R0 = widx * per, R1 = min(rows_t, R0 + per);
    const int n_tiles = R1 > R0 ? (R1 - R0 + ES_TILE - 1) / ES_TILE : 0;
    const float s_grad = p.s_grad, s_act = p.s_act;

    // ---- weights: W1 rows of this wave's hidden units (first layer's B operand), W2_t's planes for demb W2 (split once: h in registers, m in LDS)
    float w1f[6];
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) w1f[kk] = p.W1[(32 * W + fr) * 12 + 2 * kk + fq];
    const float b1v = p.b1[32 * W + fr];
    f16x8 w2h[8];
    _Float16* const w2m_s = reinterpret_cast<_Float16*>(smem + ES_W2M);
    {
        const float* W2t = p.W2 + (size_t)t * 128 * 128;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = W2t[(size_t)(16 * ks + 8 * fq + e) * 128 + 32 * W + fr];      // B[K slot = channel][col = hidden unit]
            w2h[ks] = es_split8(v, p.s_w).h;
        }
        for (int e = tid; e < 128 * 128; e += ES_THREADS) {       // m plane -> LDS [k][c] (same rounding as the h plane above)
            const int c = e >> 7, k = e & 127;
            const float x = W2t[e] * p.s_w;
            w2m_s[k * ES_RM_LD + c] = (_Float16)(x - (float)(_Float16)x);
        }
    }
    f32x16 acc2[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[cb][r] = 0.f;
    // dW1^T[f][k] (f = 12: db1) x s_act s_grad of the wave's 32 hidden units: the fold is a product on the matrix cores too (13 fmas per
    // element on the vector unit were a quarter of the kernel's time) - A = the records (feature f; f = 12: ones), B = the masked dbasic
    f32x16 accf = {};
    f32x16 accs = {};                                             // ones^T demb of channel block W (every row of D the same): d(b2_t) x s_grad
    const f16x8 ones8 = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    const float inv_sw = 1.f / p.s_w;
    const float xsel = fr < 12 ? 1.f : 0.f, xone = fr == 12 ? 1.f : 0.f;

    // ---- raw inputs of a tile: wave W forms rows 16 W .. 16 W + 15; lane = (row half hf, channel quad cq): eight rows x four channels per
    // thread, per-lane row indices (as scalars the sixteen rows' index and address arithmetic spilled 130 SGPRs).
    // No branch around a load (a conditional load is a basic block of its own with a wait behind it: the rows' loads would go out one after
    // the other): a type without a pooled slot (eth) reads slot 1 and multiplies it away, the one-unit types read the arg-max bytes of type
    // eh and ignore them, rows past the range re-read the last row and are switched off.
    struct Raw { float4 d[8], q[8]; float dt[8]; unsigned a[8]; int u[8]; float x[3]; };
    const int slot_c = slot >= 0 ? slot : 1;
    const float dsel = slot >= 0 ? 1.f : 0.f;
    const bool one_unit = U == 1;
    const int hf = lane >> 5, cq = lane & 31;
    int xr[3], xf[3];                                             // record elements of this thread: [row][12] flattened, 768 per tile
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int e = tid + ES_THREADS * i; xr[i] = e / 12; xf[i] = e - 12 * xr[i]; }
    auto load_raw = [&](int tile, Raw& r) __attribute__((always_inline)) {
        const int Rt = R0 + tile * ES_TILE;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int R = Rt + 16 * W + 8 * hf + e;
            const bool valid = R < R1;
            const unsigned Rc = (unsigned)(valid ? R : R1 - 1);
            const unsigned n = one_unit ? Rc : Rc / 5u, u = one_unit ? 0u : Rc - 5u * n;
            r.d[e] = reinterpret_cast<const float4*>(p.dxcat + (size_t)n * ES_XCAT + slot_c * 128)[cq];
            r.q[e] = reinterpret_cast<const float4*>(p.q + (size_t)n * p.ldq)[cq];
            r.a[e] = reinterpret_cast<const unsigned*>(p.amax + ((size_t)n * 3 + 0) * 128)[cq];
            const float dt = p.dtu[(size_t)n * 40 + cum + u];
            r.dt[e] = valid ? dt : 0.f;
            r.u[e] = valid ? (int)u : 255;                        // (255: a row that does not exist: dtu = 0 above, no routing below)
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned Rc = (unsigned)min(Rt + xr[i], R1 - 1);
            const unsigned n = one_unit ? Rc : Rc / 5u, u = one_unit ? 0u : Rc - 5u * n;
            r.x[i] = p.obs[(size_t)n * ES_OBS + 3 + (cum + u) * 12 + xf[i]];
        }
    };
    // demb of the patch: the arithmetic (per row, into registers: placed between the MFMAs of P2, whose issue slots are free) ...
    struct Built { unsigned ph[8][2], pm[8][2]; };                // per row: channels (4 cq, 4 cq + 1), (4 cq + 2, 4 cq + 3) packed, planes h / m
    auto build_row = [&](const Raw& r, Built& o, int e) __attribute__((always_inline)) {
        const bool live = r.u[e] != 255;
        const float dv[4] = {r.d[e].x, r.d[e].y, r.d[e].z, r.d[e].w}, qv[4] = {r.q[e].x, r.q[e].y, r.q[e].z, r.q[e].w};
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool mj = live && (one_unit || (int)((r.a[e] >> (8 * j)) & 0xffu) == r.u[e]);
            v[j] = fmaf(r.dt[e], qv[j], mj ? dv[j] * dsel : 0.f) * s_grad;
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            o.ph[e][h2] = cvt_pk_f16(v[2 * h2], v[2 * h2 + 1]);
            const f16x2_t hv = __builtin_bit_cast(f16x2_t, o.ph[e][h2]);
            o.pm[e][h2] = cvt_pk_f16(v[2 * h2] - (float)hv.x, v[2 * h2 + 1] - (float)hv.y);
        }
    };
    // ... and the stores into both LDS images (planes h, m) once the tile before is done with them; records -> xs
    auto build_store = [&](const Raw& r, const Built& o) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            uint2* const row = reinterpret_cast<uint2*>(drm + (16 * W + 8 * hf + e) * ES_RM_LD) + cq;
            row[0] = make_uint2(o.ph[e][0], o.ph[e][1]);
            row[ES_RM_PLANE / 4] = make_uint2(o.pm[e][0], o.pm[e][1]);
        }
        // [c][row]: the eight rows of each of the four channels as 16 bytes (channel 4 cq + 2 h2 in the low halves, + 1 in the high halves)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            u32x4 c0h, c1h, c0m, c1m;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c0h[i] = __builtin_amdgcn_perm(o.ph[2 * i + 1][h2], o.ph[2 * i][h2], 0x05040100u);
                c1h[i] = __builtin_amdgcn_perm(o.ph[2 * i + 1][h2], o.ph[2 * i][h2], 0x07060302u);
                c0m[i] = __builtin_amdgcn_perm(o.pm[2 * i + 1][h2], o.pm[2 * i][h2], 0x05040100u);
                c1m[i] = __builtin_amdgcn_perm(o.pm[2 * i + 1][h2], o.pm[2 * i][h2], 0x07060302u);
            }
            // (the 16-byte chunk of eight rows sits at chunk ^ ((c >> 4) & 3) of its channel's row: lanes four channels apart would otherwise put
            // these stores four to a bank group; P2's reads stay conflict-free - the XOR's lane part there is fq ^ (fr >> 4))
            _Float16* const ch = dcm + (4 * cq + 2 * h2) * ES_CM_LD + 8 * ((2 * W + hf) ^ ((cq >> 2) & 3));
            *reinterpret_cast<u32x4*>(ch) = c0h;
            *reinterpret_cast<u32x4*>(ch + ES_CM_LD) = c1h;
            *reinterpret_cast<u32x4*>(ch + ES_CM_PLANE) = c0m;
            *reinterpret_cast<u32x4*>(ch + ES_CM_PLANE + ES_CM_LD) = c1m;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) xs[tid + ES_THREADS * i] = r.x[i];
    };

    Raw raw;
    Built built;
    if (n_tiles > 0) {
        load_raw(0, raw);
#pragma unroll
        for (int e = 0; e < 8; ++e) build_row(raw, built, e);
        build_store(raw, built);
    }
#ifdef ES_TIMING
    long long tacc[6] = {0, 0, 0, 0, 0, 0};
#define ES_T(i) const long long tq##i = __builtin_amdgcn_s_memtime()
#else
#define ES_T(i)
#endif
    for (int tile = 0; tile < n_tiles; ++tile) {
        ES_T(0);
        __syncthreads();                                                           // #1: images and records of this tile are complete
        ES_T(1);
        if (tile + 1 < n_tiles) load_raw(tile + 1, raw);                           // in flight behind the tile's products
        // The two row halves of the tile in lock step: every product below has two independent accumulator chains (one wave per SIMD:
        // nothing else fills the latency of a dependent MFMA), and the W2 fragments serve both.
        {
            // ---- P1: basic of rows 32 b + .., hidden units 32 W + .. -----------------------------------------------------------------------
            unsigned mask[2] = {0u, 0u};
            f32x16 ga[2] = {};
            {
                const float* xp = xs + fr * 12 + fq;
#pragma unroll
                for (int kk = 0; kk < 6; ++kk)
#pragma unroll
                    for (int b = 0; b < 2; ++b) ga[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xp[b * 32 * 12 + 2 * kk], w1f[kk], ga[b], 0, 0, 0);
            }
            // relu, mask bits, scale + split, -> LDS [k][row]: one group of four rows; called from inside P3's K loop (vector work behind its MFMAs)
            auto p1_group = [&](int b, int grp) __attribute__((always_inline)) {
                _Float16* const hk = h1t + (32 * W + fr) * ES_CM_LD + 32 * b + 4 * fq;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = relu_nan(ga[b][4 * grp + i] + b1v);
                    mask[b] |= (v[i] > 0.f ? 1u : 0u) << (4 * grp + i);
                    v[i] *= s_act;
                }
                const unsigned h01 = cvt_pk_f16(v[0], v[1]), h23 = cvt_pk_f16(v[2], v[3]);
                const f16x2_t v01 = __builtin_bit_cast(f16x2_t, h01), v23 = __builtin_bit_cast(f16x2_t, h23);
                const unsigned m01 = cvt_pk_f16(v[0] - (float)v01.x, v[1] - (float)v01.y), m23 = cvt_pk_f16(v[2] - (float)v23.x, v[3] - (float)v23.y);
                *reinterpret_cast<uint2*>(hk + 8 * grp) = make_uint2(h01, h23);                  // rows 32 b + 8 grp + 4 fq + 0 .. 3
                *reinterpret_cast<uint2*>(hk + ES_CM_PLANE + 8 * grp) = make_uint2(m01, m23);
            };
            asm volatile("" ::: "memory");
            // ---- P3: dbasic blocks = demb W2 (K = 128 channels) ----------------------------------------------------------------------------
            f32x16 a3[2] = {};
            {
                const _Float16* const ar = drm + fr * ES_RM_LD + 8 * fq;
                const _Float16* const wm = w2m_s + (32 * W + fr) * ES_RM_LD + 8 * fq;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    f16x8 ah[2], am[2];
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        ah[b] = *reinterpret_cast<const f16x8*>(ar + b * 32 * ES_RM_LD + 16 * ks);
                        am[b] = *reinterpret_cast<const f16x8*>(ar + b * 32 * ES_RM_LD + ES_RM_PLANE + 16 * ks);
                    }
                    const f16x8 w2m = *reinterpret_cast<const f16x8*>(wm + 16 * ks);
#pragma unroll
                    for (int b = 0; b < 2; ++b) a3[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[b], w2h[ks], a3[b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < 2; ++b) a3[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], w2m, a3[b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < 2; ++b) a3[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], w2h[ks], a3[b], 0, 0, 0);
                    p1_group(ks >> 2, ks & 3);      // (an explicit MFMA / VALU issue pattern here, sched_group_barrier, measured slower: 8.4 -> 9.6 k cycles)
                    if (ks & 1) asm volatile("" ::: "memory");       // (fragment reads at most two K steps ahead: left alone the compiler requests all of them)
                }
            }
            // ---- fold: dW1^T[f][k] += records^T x masked dbasic, K = rows; K slot 8 fq + j of step s = row 32 b + 16 s + 8 (j >> 2) + 4 fq + (j & 3),
            // which is D register 8 s + j of this very lane: the B operand is the accumulator as it lies (x 1 / s_w: back to gradient scale) --------
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int sk = 0; sk < 2; ++sk) {
                    float gv[8], xv[8];
                    const float* const xr = xs + (32 * b + 16 * sk + 4 * fq) * 12 + min(fr, 11);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        gv[j] = (mask[b] >> (8 * sk + j)) & 1u ? a3[b][8 * sk + j] : 0.f;
                        xv[j] = fmaf(xr[(8 * (j >> 2) + (j & 3)) * 12], xsel, xone);
                    }
                    const Split2h A = es_split8(xv, s_act), B = es_split8(gv, inv_sw);
                    accf = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.m, B.h, accf, 0, 0, 0);
                    accf = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.h, B.m, accf, 0, 0, 0);
                    accf = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.h, B.h, accf, 0, 0, 0);
                }
        }
        ES_T(2);
        __syncthreads();                                                           // #2: basic [k][row] is complete
        ES_T(3);
        // ---- P2: dW2^T[k][c] += basic^T demb, K = the tile's rows: this wave's 32 hidden units x all 128 channels ------------------------
        {
            const _Float16* const ak = h1t + (32 * W + fr) * ES_CM_LD + 8 * fq;
            const _Float16* const bc = dcm + fr * ES_CM_LD + 8 * (fq ^ (fr >> 4));        // ([c][row] image: chunk swizzle, see build_store)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(ak + 16 * ks);
                const f16x8 am = *reinterpret_cast<const f16x8*>(ak + ES_CM_PLANE + 16 * ks);
                f16x8 bh[4], bm[4];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    bh[cb] = *reinterpret_cast<const f16x8*>(bc + cb * 32 * ES_CM_LD + 16 * (ks ^ (cb & 1)));
                    bm[cb] = *reinterpret_cast<const f16x8*>(bc + cb * 32 * ES_CM_LD + ES_CM_PLANE + 16 * (ks ^ (cb & 1)));
                }
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bh[cb], acc2[cb], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bm[cb], acc2[cb], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[cb], acc2[cb], 0, 0, 0);
#if ES_DB2
                {   // d(b2_t) = the column sums of demb = ones^T demb: wave W takes channel block W (its fragments once more: the block index is
                    // not a compile-time one), two MFMAs with a constant A - as vector adds in the build phase the sums cost 60 us per launch
                    const _Float16* const bw = bc + W * 32 * ES_CM_LD + 16 * (ks ^ (W & 1));
                    const f16x8 bhw = *reinterpret_cast<const f16x8*>(bw), bmw = *reinterpret_cast<const f16x8*>(bw + ES_CM_PLANE);
                    accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones8, bmw, accs, 0, 0, 0);
                    accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones8, bhw, accs, 0, 0, 0);
                }
#endif
                // a quarter of the next tile's demb arithmetic behind these twelve MFMAs (its inputs were requested at the top of the tile)
                build_row(raw, built, 2 * ks); build_row(raw, built, 2 * ks + 1);
                // (issue order for the scheduler: one MFMA, then a few of the vector instructions above, twelve times - left to itself it issues
                // the twelve MFMAs back to back and the ~90 vector instructions after them, one wave per SIMD: nothing overlaps)
#pragma unroll
                for (int i = 0; i < (ES_DB2 ? 14 : 12); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, ES_DB2 ? 7 : 8, 0);
                }
                asm volatile("" ::: "memory");
            }
        }
        ES_T(4);
        __syncthreads();                                                           // #3: everybody is done with this tile's images
        ES_T(5);
        if (tile + 1 < n_tiles) build_store(raw, built);
#ifdef ES_TIMING
        const long long tq6 = __builtin_amdgcn_s_memtime();
        tacc[0] += tq1 - tq0; tacc[1] += tq2 - tq1; tacc[2] += tq3 - tq2; tacc[3] += tq4 - tq3; tacc[4] += tq5 - tq4; tacc[5] += tq6 - tq5;
#endif
    }
#ifdef ES_TIMING
    if (lane == 0 && (g == 0 || g == 100 || g == 230) && n_tiles > 0)
        printf("embed_small wg %3d wave %d tiles %d: barrier1 %.0f  P1+P3+fold (+load issue) %.0f  barrier2 %.0f  P2 %.0f  barrier3 %.0f  build %.0f (cycles per tile)\n", g, W, n_tiles,
               (double)tacc[0] / n_tiles, (double)tacc[1] / n_tiles, (double)tacc[2] / n_tiles, (double)tacc[3] / n_tiles, (double)tacc[4] / n_tiles, (double)tacc[5] / n_tiles);
#endif

    // ---- outputs ----------------------------------------------------------------------------------------------------------------------------
    {   // dW2[c][k]: this wave's four blocks of dW2^T, transposed on the way out (four consecutive k per 16-byte store)
        const float inv2 = 1.f / (s_act * s_grad);
        float* const out = p.slab + (size_t)(g < 192 ? g : g + p.slab_skip) * 128 * 128;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            float* const o = out + (size_t)(32 * cb + fr) * 128 + 32 * W + 4 * fq;
#pragma unroll
            for (int grp = 0; grp < 4; ++grp)
                *reinterpret_cast<float4*>(o + 8 * grp) = make_float4(acc2[cb][4 * grp] * inv2, acc2[cb][4 * grp + 1] * inv2, acc2[cb][4 * grp + 2] * inv2,
                                                                     acc2[cb][4 * grp + 3] * inv2);
        }
    }
    if (p.db2part != nullptr && fq == 0) p.db2part[(size_t)g * 128 + 32 * W + fr] = accs[0] * (1.f / s_grad);   // d(b2_t)[c] of this workgroup's rows
    {   // dW1 / db1 of this wave's hidden units: D register r of the fold = feature 8 (r >> 2) + 4 fq + (r & 3) (12 = the bias)
        float* const o = p.part + (size_t)g * 1664 + 32 * W + fr;
        const float inv_fold = 1.f / (s_act * s_grad);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int f = 8 * (r >> 2) + 4 * fq + (r & 3);
            if (f < 13) o[f * 128] = accf[r] * inv_fold;
        }
    }
}

// slab: the workgroups' dW2 blocks in splitk_reduce_grouped's order - ah: 0 .. 31, eh: 32 .. 191, then `slab_skip` blocks of other
// kernels (the two 16-unit types'), ath: 192 + skip .., eth: 224 + skip ..; part: [256][13][128]
int embed_bwd_small(const float* obs, const float* dxcat, const uint8_t* amax, const float* dtu, const float* q, int ldq, const float* W1,
                    const float* b1, const float* W2, float* slab, int slab_skip, float* part, float* db2part, long long nr, hipStream_t s,
                    const F16x2Scales& f16) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)embed_small_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ES_LDS);
        if (e != hipSuccess) { set_error("embed_bwd_small: hipFuncSetAttribute", (int)e); return (int)e; }
        attr = true;
    }
    SmallArgs a{obs, dxcat, amax, dtu, q, ldq, W1, b1, W2, slab, slab_skip, part, db2part, nr, f16.s_act, f16.s_w, f16.s_grad};
    ProfScope prof("embed_bwd_small", 2.0 * nr * 8 * 128 * (2 * 128 + 24), 4.0 * nr * (8 * 12 + 4 * 128 + 128 + 8), s);
    hipLaunchKernelGGL(embed_small_bwd_kernel, dim3(256), dim3(ES_THREADS), ES_LDS, s, a);
    return launch_check("embed_bwd_small");
}

}  // namespace dc
