// Backward of the per-unit embedding MLP for the two 16-unit types, third form (round 5; the default with the two-f16-piece products):
// the DENSE products on the f16 matrix cores with every operand generated on chip.
//
// Replaces, for these types, the part of /root/reference/optimizer.py:672 (autograd) that flows through policy.py:102-136,152 - like
// embed_sparse.hip, whose header states the algebra.  embed_sparse.hip executes the sparse form (one non-zero of d(emb) per step, type
// and channel: 1/16 of the dense MACs) as per-channel gathers on the packed-f32 VALU and is bound by their latency: sorted channel lists,
// dependent LDS round trips, a barrier per step, 0.20 of the vector peak after three rounds of tuning.  The matrix cores do sixteen
// times the VALU's MACs per cycle, so here the DENSE form runs instead - per env-step and type
//     dW2^T[k][c]  += sum_u basic[u][k] demb[u][c]          K = 16 units                          (kernel 1)
//     dbasic[u][k]  = sum_c demb[u][c] W2[c][k]             K = 128 channels                      (kernel 2)
//     dW1^T[f][k]  += sum_u x[u][f] relu'(.) dbasic[u][k]   K = 16 units, f = 12: ones (db1)      (kernel 2)
// with demb[u][c] = [amax(c) == u] d(xcat)[c] + dtu[u] q[c] - and NOTHING of it ever exists in memory: a lane builds the 8 operand
// elements an MFMA wants from it out of the arg-max bytes and d(xcat) (read where the forward / the loss left them), basic comes out
// of the first-layer MFMA in exactly the register layout the next product takes as its A operand, the relu-masked d(basic) in the
// layout the fold takes as its B operand.  No prepare pass, no sorted lists, no LDS gathers, no barrier inside a loop: a wave only ever
// consumes what it computed itself or what is read-only.
//
// The rank-one attention term dtu[u] q[c] would make the one-hot operands dense, so it stays out of them:
//     dW2^T[k][c]   += s[k] q[c],  s[k] = sum_u dtu[u] basic[u][k]: two K slots (the pair's two steps) of three more MFMAs per column block
//     d(basic)[u][k] += dtu[u] R[k], R = q W2 - a dense product over all steps launched ahead (like embed_sparse.hip's), added in f32
//
// Arithmetic: two f16 pieces per f32 operand, three v_mfma_f32_32x32x16_f16 per product (hh, hm, mh), f32 accumulate - gemm_x3.hip's
// PREC 4 with the same power-of-two pre-scales (activations and records s_act, weights s_w, gradients s_grad); d(xcat) is scaled and
// split ONCE, when a step is staged (one dword per channel: h | m << 16), so an operand element is a select of a staged dword - no
// arithmetic per element.  The first layer (K = 12) alone runs on the f32-input MFMA, like the dense backward kernels' mask evaluation:
// the relu mask is theirs bit for bit, and - like the oracle's torch f32 - a pre-activation must be within ~1e-7 of zero to get another
// sign (the forward's two-f16-piece sequence flips five times as many of the 2.7 x 10^8 pre-activations of a bench pass, each a whole
// term of a dW1 row: one flip measured 3.6e-3 of the largest entry at 1 536 steps).
//
// Two kernels, because the two halves want different splits of the work (measured as ONE kernel with k-quarter waves doing everything:
// 727 us, four waves building the same d(basic) operands):
//   1. embed_pool16m_dw2_kernel: workgroup = one type x a range of steps; wave = (stream, k quarter): a stream takes every other PAIR of
//      steps (a pair's 2 x 16 units are the 32 rows of an MFMA tile), a wave owns the 32 hidden units k = 32 kq .. + 31: its slice of basic
//      (first layer), the relu masks (written out as 16 bits per lane, 128 B per wave and pair), and its 32 x 128 slice of dW2^T - four
//      accumulator tiles for the whole kernel.
//   2. embed_pool16m_dw1_kernel: a wave takes whole pairs for ALL 128 hidden units: the one-hot operand of a K step is built once and
//      meets four W2 column blocks (LDS image) in four independent accumulator chains; the masks come from kernel 1, the fold
//      accumulates dW1^T / db1 in four tiles per wave.
// Register layouts (lane = (fr = lane & 31, fq = lane >> 5)):
//     A operand: A[row fr][K slot 8 fq + j], B operand: B[K slot 8 fq + j][col fr], D: register r = D[row 8 (r >> 2) + 4 fq + (r & 3)][col fr]
// The first layer's D registers hold, for item e of the pair, units sigma(fq, j) = 4 fq + (j & 3) + 8 (j >> 2) in registers 8 e + j: K slot
// 8 fq + j of the unit-contracting products stands for unit sigma(fq, j), and those products take D registers as operands as they are.
// tools/pool16m_sim.py models exactly this index math lane by lane against a dense float64 evaluation.
#include <stdio.h>
#include "kernels.h"
#include "gemm_tiles.h"

namespace dc {
namespace {

enum { PM_THREADS = 512, PM_OBS = 483, PM_XCAT = 896 };

struct PoolMArgs {
    const float* obs; const float* dxcat; const uint8_t* amax; const float* dtu; const float* q; int ldq;
    const float* W1; const float* b1; const float* W2;
    float* slab; float* part1; float* part2;
    long long nr; int wg_per_type; int steps_per_wg;
    float s_act, s_w, s_grad;
    const float* R;           // [2][nr][128]: R = q W2_t of every step and type (a dense product, launched ahead of the kernels)
    uint16_t* mask;           // [2][nr / 2 rounded up][64 lanes][4 k blocks]: relu masks of a pair's first layer, bit r = D register r of the lane
};

__device__ __forceinline__ Split2h split8(const float (&v)[8], float s) {
    return split2h<true>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), s);
}
__device__ __forceinline__ Split2h split8_noscale(const float (&v)[8]) {
    return split2h<false>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), 1.f);
}
// three-term product of two-piece operands, smallest terms first (gemm_tiles.h: the m m term is below what two pieces represent)
__device__ __forceinline__ f32x16 mma3(const Split2h& a, const Split2h& b, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.m, b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.m, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h, acc, 0, 0, 0);
}
// (d0, d1) x s_grad as two staged dwords h | m << 16
__device__ __forceinline__ uint2 stage_pieces(float d0, float d1, float s_grad) {
    const float x0 = d0 * s_grad, x1 = d1 * s_grad;
    const unsigned hh = cvt_pk_f16(x0, x1);
    const f16x2_t hv = __builtin_bit_cast(f16x2_t, hh);
    const unsigned mm = cvt_pk_f16(x0 - (float)hv.x, x1 - (float)hv.y);
    return make_uint2((hh & 0xffffu) | (mm << 16), (hh >> 16) | (mm & 0xffff0000u));
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------------
// Kernel 1: first layer, relu masks, dW2^T (+ the second-layer bias gradient)
// ---------------------------------------------------------------------------------------------------------------------------------------
enum : int {   // LDS (bytes)
    P1_STG = 0,                      // every wave's own staging block: 2 buffers x 2 items x 2.4 KB
    P1_ITEM = 592,                   // dwords per item; 592 = 16 mod 64: the two items of a pair start 16 banks apart
    P1_RED_ACC = 0,                  // at the end, over the staging blocks: another stream's dW2^T tiles [kq 4][64 registers][64 lanes] f32
    P1_RED_B = 65536                 // ... and its bias-gradient sums [2][64] f32
};
// one-hot table behind the staging blocks: entry i < 8 = f16 1.0 at element i, entries 8..15 = zeros
constexpr int p1_tab(int nstream) { return P1_STG + 4 * nstream * (nstream == 2 ? 4 : 2) * P1_ITEM * 4; }      // two staging buffers with two streams, one with three
constexpr int p1_lds(int nstream) { return (65536 + 1024 > p1_tab(nstream) + 256) ? 65536 + 1024 : p1_tab(nstream) + 256; }

// NSTREAM streams of pairs x 4 k quarters = 4 NSTREAM waves; NSTREAM = 3: three waves per SIMD (<= 168 registers)
template <int NSTREAM>
__global__ __launch_bounds__(256 * NSTREAM) void embed_pool16m_dw2_kernel(PoolMArgs p) {
    constexpr int P1_TAB = p1_tab(NSTREAM);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int W = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = W & 3, st = W >> 2;
    const int fr = lane & 31, fq = lane >> 5;
    const int t = 2 + blockIdx.x / p.wg_per_type;          // 2 = allied non-heroes, 3 = enemy non-heroes
    const int wgi = blockIdx.x % p.wg_per_type;
    const long long n0 = (long long)wgi * p.steps_per_wg;
    const long long n1 = min(p.nr, n0 + p.steps_per_wg);
    const int cum = t == 2 ? 6 : 22;                        // first unit of the type inside the 40
    const int slot0 = t == 2 ? 3 : 4;                       // xcat slot the type's max feeds; enh feeds slot 6 as well (policy.py:127)
    const float s_act = p.s_act, s_grad = p.s_grad;

    if (tid < 16 * 8) reinterpret_cast<uint16_t*>(smem + P1_TAB)[tid] = (tid >> 3) < 8 && (tid & 7) == (tid >> 3) ? 0x3C00u : 0u;   // f16 1.0
    // W1 rows of this wave's k block as the first layer's B operand (six v_mfma_f32_32x32x2_f32, K = 12: file header)
    float w1f[6];
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) w1f[kk] = p.W1[(32 * kq + fr) * 12 + 2 * kk + fq];
    const float b1v = p.b1[32 * kq + fr];

    f32x16 acc[4];                                           // dW2^T[k = 32 kq + row][c = 32 cb + col] x s_act s_grad
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; acc[2][r] = 0.f; acc[3][r] = 0.f; }
    __syncthreads();                                         // the table

    // ---- per pair: raw inputs global -> registers (a pair ahead) -> this wave's own LDS block -> operand builds --------------------------
    // Every wave stages what IT needs (the four waves of a stream read the same lines: L1 / L2 serve three of them); the loads of pair i + 1
    // are in flight while pair i computes, so no operand build ever waits for HBM, and no wave ever waits for another.
    // Block of an item (dwords): d(xcat) pieces [128] | q [128] | unit records [16][12] | dtu [16] | per channel and lane group, the
    // one-hot table entry of the arg-max unit: slot (a & 3) + 4 (a >> 3) if the unit is of the group, else >= 8 [2][128 B]
    enum { ST_D = 0, ST_Q = 128, ST_X = 256, ST_DTU = 448, ST_IDX = 464, ST_LIVE = 528, ST_ITEM = P1_ITEM };     // ... | 1 if any dtu != 0
    float* const stg = reinterpret_cast<float*>(smem + P1_STG) + (size_t)W * ((NSTREAM == 2 ? 2 : 1) * 2 * ST_ITEM);
    const long long n_pairs = (n1 - n0 + 1) / 2;
    const int e_row = fr >> 4, u_row = fr & 15;              // as a ROW of the pair's tile this lane is unit u_row of item e_row
    uint16_t* const mask_t = p.mask + ((size_t)(t - 2) * ((p.nr + 1) / 2)) * 256;
    float db2a[2] = {0.f, 0.f};                              // kq == 0: sum over steps of demb's column sums, channels 2 lane, 2 lane + 1
    struct Raw { float2 d, d2, q; float x0, x1, x2, dt; unsigned a; bool valid; };
    auto load_raw = [&](long long pi, int e) {
        Raw r;
        long long n = n0 + 2 * pi + e;
        const bool valid = n < n1;                           // wave-uniform; an absent item (odd range) contributes zeros
        n = valid ? n : n1 - 1;
        const float* dx = p.dxcat + n * PM_XCAT + slot0 * 128 + 2 * lane;
        r.d = *reinterpret_cast<const float2*>(dx);
        r.d2 = t == 3 ? *reinterpret_cast<const float2*>(dx + 2 * 128) : make_float2(0.f, 0.f);
        r.q = *reinterpret_cast<const float2*>(p.q + n * p.ldq + 2 * lane);
        const float* xr = p.obs + n * PM_OBS + 3 + cum * 12 + lane;
        r.x0 = xr[0]; r.x1 = xr[64]; r.x2 = xr[128];
        r.dt = lane < 16 ? p.dtu[n * 40 + cum + lane] : 0.f;
        r.a = lane < 32 ? reinterpret_cast<const unsigned*>(p.amax + (n * 3 + (t - 1)) * 128)[lane] : 0u;
        r.valid = valid;               // applied when the item is STAGED: a select here would wait for the loads a pair too early
        return r;
    };
    auto store_raw = [&](const Raw& r, int buf, int e) {
        float* b = stg + (buf * 2 + e) * ST_ITEM;
        const float d0 = r.valid ? r.d.x + r.d2.x : 0.f, d1 = r.valid ? r.d.y + r.d2.y : 0.f;      // policy.py:127: enh feeds two slots
        const float dt = r.valid ? r.dt : 0.f;
        *reinterpret_cast<uint2*>(b + ST_D + 2 * lane) = stage_pieces(d0, d1, s_grad);
        *reinterpret_cast<float2*>(b + ST_Q + 2 * lane) = r.q;
        b[ST_X + lane] = r.x0; b[ST_X + 64 + lane] = r.x1; b[ST_X + 128 + lane] = r.x2;
        if (lane < 16) b[ST_DTU + lane] = dt;
        const unsigned long long nz = __ballot(dt != 0.f);
        if (lane == 0) reinterpret_cast<int*>(b)[ST_LIVE] = nz != 0ull;
        if (lane < 32) {
            // four channels at once: slot = (a & 3) | ((a >> 1) & 4), group = (a >> 2) & 1; entry = slot | 8 for the OTHER group
            const unsigned jj = (r.a & 0x03030303u) | ((r.a >> 1) & 0x04040404u), g1 = (r.a >> 2) & 0x01010101u;
            reinterpret_cast<unsigned*>(b + ST_IDX)[lane] = jj | (g1 << 3);                          // lane group 0
            reinterpret_cast<unsigned*>(b + ST_IDX)[32 + lane] = jj | ((g1 ^ 0x01010101u) << 3);     // lane group 1
        }
        if (kq == 0) {                                                       // wave-uniform: the bias gradient rides on wave (st, 0)
            float sum = dt;                                                  // row 0 of the wave = lanes 0..15: their sum in each of them
            sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x128, 0xf, 0xf, true));
            sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x124, 0xf, 0xf, true));
            sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x122, 0xf, 0xf, true));
            sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x121, 0xf, 0xf, true));
            const float sumdtu = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sum)));
            db2a[0] += fmaf(r.q.x, sumdtu, d0);
            db2a[1] += fmaf(r.q.y, sumdtu, d1);
        }
    };
    // first layer of the pair staged in buffer b: basic of its 32 rows, this wave's 32 hidden units (unscaled: row 8 (r >> 2) + 4 fq + (r & 3))
    auto first_layer = [&](int b) {
        const float* xp = stg + (b * 2 + e_row) * ST_ITEM + ST_X + u_row * 12 + fq;     // A operand of MFMA kk: x[row][feature 2 kk + fq]
        f32x16 g = {};
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) g = __builtin_amdgcn_mfma_f32_32x32x2f32(xp[2 * kk], w1f[kk], g, 0, 0, 0);
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = relu_nan(g[r] + b1v);
        return o;
    };
    // Software pipeline over the pairs of the stream: while pair i's products run, pair i + 1 is staged (its loads were issued an iteration
    // ago) and goes through the first layer (six dependent f32 MFMAs whose results nobody waits for), and pair i + 2's loads are issued.
    // (PIPE, two streams: 213 registers.  Three streams - three waves per SIMD, <= 168 registers - do without it: the next pair's
    // loads are in flight while the pair computes and are staged at the bottom, into the same block.)
    constexpr bool PIPE = NSTREAM == 2;
    Raw nx0, nx1;
    f32x16 basic_next = {};
    if (st < n_pairs) {
        const Raw r0 = load_raw(st, 0), r1 = load_raw(st, 1);
        store_raw(r0, 0, 0); store_raw(r1, 0, 1);
        if constexpr (PIPE) {
            if (st + NSTREAM < n_pairs) { nx0 = load_raw(st + NSTREAM, 0); nx1 = load_raw(st + NSTREAM, 1); }
            __builtin_amdgcn_wave_barrier();
            basic_next = first_layer(0);
        }
    }
    int buf = 0;
    for (long long pi = st; pi < n_pairs; pi += NSTREAM, buf ^= PIPE ? 1 : 0) {
        const bool more = pi + NSTREAM < n_pairs;            // wave-uniform
        if constexpr (!PIPE) {
            if (more) { nx0 = load_raw(pi + NSTREAM, 0); nx1 = load_raw(pi + NSTREAM, 1); }
            __builtin_amdgcn_wave_barrier();
            basic_next = first_layer(0);
        }
        const f32x16 basic = basic_next;
        {   // its relu masks for kernel 2: bit r = D register r of this lane is past the relu
            unsigned bits = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) bits |= (basic[r] > 0.f ? 1u : 0u) << r;
            mask_t[((size_t)((n0 >> 1) + pi) * 64 + lane) * 4 + kq] = (uint16_t)bits;
        }
        if constexpr (PIPE) {
            if (more) {
                store_raw(nx0, buf ^ 1, 0); store_raw(nx1, buf ^ 1, 1);
                if (pi + 2 * NSTREAM < n_pairs) { nx0 = load_raw(pi + 2 * NSTREAM, 0); nx1 = load_raw(pi + 2 * NSTREAM, 1); }
                __builtin_amdgcn_wave_barrier();
                basic_next = first_layer(buf ^ 1);
            }
        }
        const float* it0 = stg + (buf * 2) * ST_ITEM;        // item e of the pair: it0 + e * ST_ITEM

        // ---- dW2^T += basic^T demb, item by item (K = the item's 16 units) -------------------------------------------------------------
        float s_att[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float* it = it0 + e * ST_ITEM;
            float bv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = basic[8 * e + j];                  // K slot 8 fq + j <-> unit sigma(fq, j): as they lie
            const Split2h A = split8(bv, s_act);
            if constexpr (!PIPE) {      // three streams: one column block at a time (16 live operand registers instead of 48)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const unsigned d2 = reinterpret_cast<const unsigned*>(it + ST_D)[32 * cb + fr];
                    const int idx = reinterpret_cast<const uint8_t*>(it + ST_IDX)[128 * fq + 32 * cb + fr];
                    const f16x8 hot = *reinterpret_cast<const f16x8*>(smem + P1_TAB + idx * 16);
                    const f16x2_t dd = __builtin_bit_cast(f16x2_t, d2);
                    Split2h B;
                    B.h = hot * dd.x;
                    B.m = hot * dd.y;
                    acc[cb] = mma3(A, B, acc[cb]);
                }
            } else {
                // reads first (two dependent LDS round trips: entry index, then the table entry), then the four operands, then the products -
                // spelled out in that order so that the round trips of the four column blocks overlap
                unsigned d2[4];
                int idx[4];
    #pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    d2[cb] = reinterpret_cast<const unsigned*>(it + ST_D)[32 * cb + fr];
                    idx[cb] = reinterpret_cast<const uint8_t*>(it + ST_IDX)[128 * fq + 32 * cb + fr];
                }
                f16x8 hot[4];
    #pragma unroll
                for (int cb = 0; cb < 4; ++cb) hot[cb] = *reinterpret_cast<const f16x8*>(smem + P1_TAB + idx[cb] * 16);
                // one-hot over the lane group's eight K slots: the table entry (f16 1.0 at the unit's slot, or zeros) times the piece - eight
                // packed f16 multiplies by exactly 0 or 1 instead of compares and selects per register
                Split2h B[4];
    #pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const f16x2_t dd = __builtin_bit_cast(f16x2_t, d2[cb]);
                    B[cb].h = hot[cb] * dd.x;
                    B[cb].m = hot[cb] * dd.y;
                }
                // (independent accumulators: term by term across the four blocks, not three dependent MFMAs per block)
    #pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.m, B[cb].h, acc[cb], 0, 0, 0);
    #pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.h, B[cb].m, acc[cb], 0, 0, 0);
    #pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.h, B[cb].h, acc[cb], 0, 0, 0);
            }
            {   // s[k] (x s_act s_grad) of the rank-one attention term (below), over the lane group's eight units ...
                const float4 du0 = *reinterpret_cast<const float4*>(it + ST_DTU + 4 * fq), du1 = *reinterpret_cast<const float4*>(it + ST_DTU + 8 + 4 * fq);
                float sk = du0.x * bv[0];
                sk = fmaf(du0.y, bv[1], sk); sk = fmaf(du0.z, bv[2], sk); sk = fmaf(du0.w, bv[3], sk);
                sk = fmaf(du1.x, bv[4], sk); sk = fmaf(du1.y, bv[5], sk); sk = fmaf(du1.z, bv[6], sk); sk = fmaf(du1.w, bv[7], sk);
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sk), __float_as_uint(sk), false, false);
                s_att[e] = (sk + __uint_as_float(fq ? sw[0] : sw[1])) * (s_grad * s_act);   // ... plus the other group's (lane ^ 32); dtu is a gradient
            }
        }
        if (__builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(it0)[ST_LIVE] | reinterpret_cast<const int*>(it0)[ST_ITEM + ST_LIVE]) != 0) {
            // the rank-one attention term of BOTH items, for a pair with a live target-unit head (about half of them; s = 0 for the
            // other step of such a pair): K slots 0 and 1 of lane group 0 carry A[k][e] = s_e[k], B[e][c] = q_e[c]
            const unsigned sh = cvt_pk_f16(s_att[0], s_att[1]);
            const f16x2_t shv = __builtin_bit_cast(f16x2_t, sh);
            const unsigned sm = cvt_pk_f16(s_att[0] - (float)shv.x, s_att[1] - (float)shv.y);
            Split2h A1;
            A1.h = __builtin_bit_cast(f16x8, u32x4{fq ? 0u : sh, 0u, 0u, 0u});
            A1.m = __builtin_bit_cast(f16x8, u32x4{fq ? 0u : sm, 0u, 0u, 0u});
            if constexpr (!PIPE) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const float q0 = it0[ST_Q + 32 * cb + fr], q1 = it0[ST_ITEM + ST_Q + 32 * cb + fr];
                    const unsigned qh = cvt_pk_f16(q0, q1);
                    const f16x2_t qhv = __builtin_bit_cast(f16x2_t, qh);
                    const unsigned qm = cvt_pk_f16(q0 - (float)qhv.x, q1 - (float)qhv.y);
                    Split2h B1;
                    B1.h = __builtin_bit_cast(f16x8, u32x4{fq ? 0u : qh, 0u, 0u, 0u});
                    B1.m = __builtin_bit_cast(f16x8, u32x4{fq ? 0u : qm, 0u, 0u, 0u});
                    acc[cb] = mma3(A1, B1, acc[cb]);
                }
            } else {
                Split2h B1[4];
    #pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const float q0 = it0[ST_Q + 32 * cb + fr], q1 = it0[ST_ITEM + ST_Q + 32 * cb + fr];      // O(1): the gradient pre-scale went into s
                    const unsigned qh = cvt_pk_f16(q0, q1);
                    const f16x2_t qhv = __builtin_bit_cast(f16x2_t, qh);
                    const unsigned qm = cvt_pk_f16(q0 - (float)qhv.x, q1 - (float)qhv.y);
                    B1[cb].h = __builtin_bit_cast(f16x8, u32x4{fq ? 0u : qh, 0u, 0u, 0u});
                    B1[cb].m = __builtin_bit_cast(f16x8, u32x4{fq ? 0u : qm, 0u, 0u, 0u});
                }
    #pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1.m, B1[cb].h, acc[cb], 0, 0, 0);
    #pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1.h, B1[cb].m, acc[cb], 0, 0, 0);
    #pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1.h, B1[cb].h, acc[cb], 0, 0, 0);
        
            }
        }
        __builtin_amdgcn_wave_barrier();
        if constexpr (!PIPE) {
            if (more) { store_raw(nx0, 0, 0); store_raw(nx1, 0, 1); }
        }
    }

    // ---- results: the streams meet through LDS (streams 1 .. NSTREAM - 1 write one after the other, stream 0 adds them in that order) --------
    __syncthreads();                                         // every wave is done with its staging block
    float* racc = reinterpret_cast<float*>(smem + P1_RED_ACC) + (size_t)kq * 64 * 64 + lane;
    float* rb = reinterpret_cast<float*>(smem + P1_RED_B);
#pragma unroll
    for (int other = 1; other < NSTREAM; ++other) {
        if (st == other) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) racc[(16 * cb + r) * 64] = acc[cb][r];
            if (kq == 0) { rb[lane] = db2a[0]; rb[64 + lane] = db2a[1]; }
        }
        __syncthreads();
        if (st == 0) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] += racc[(16 * cb + r) * 64];
            if (kq == 0) { db2a[0] += rb[lane]; db2a[1] += rb[64 + lane]; }
        }
        __syncthreads();
    }
    if (st == 0) {
        const float inv = 1.f / (s_act * s_grad);
        float* out = p.slab + (size_t)blockIdx.x * 128 * 128;            // [c][k]
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 v;
                v.x = acc[cb][4 * g4 + 0] * inv;
                v.y = acc[cb][4 * g4 + 1] * inv;
                v.z = acc[cb][4 * g4 + 2] * inv;
                v.w = acc[cb][4 * g4 + 3] * inv;
                *reinterpret_cast<float4*>(out + (size_t)(32 * cb + fr) * 128 + 32 * kq + 8 * g4 + 4 * fq) = v;
            }
        if (kq == 0)
            *reinterpret_cast<float2*>(p.part2 + (size_t)blockIdx.x * 128 + 2 * lane) = make_float2(db2a[0], db2a[1]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Kernel 2: d(basic) = demb W2 for all 128 hidden units, through kernel 1's relu masks, folded into dW1^T / db1
// ---------------------------------------------------------------------------------------------------------------------------------------
enum : int {   // LDS (bytes)
    P2_W2P = 0,                      // [piece 2][K step 8][fq 2][k 128][8 channels] f16: the B operand of the d(basic) product (W2 x s_w)
    P2_STG = 65536,                  // every wave's own staging block: 2 items x 2.1 KB
    P2_ITEM = 528,                   // dwords per item; 528 = 16 mod 64
    P2_RED = 0                       // at the end, over the W2 image: the running sum [13][128] f32 of the waves' fold tiles
};
constexpr int p2_lds(int waves) { return P2_STG + waves * 2 * P2_ITEM * 4; }

// bit `pos` of `bits` ? x : 0 - a sign-extended one-bit field (all ones or zero) and an AND: two instructions
__device__ __forceinline__ float keep_if(float x, unsigned bits, int pos) {
    return __uint_as_float(__float_as_uint(x) & (unsigned)__builtin_amdgcn_sbfe((int)bits, pos, 1));
}

// KSPLIT = 1, NSTREAM = 8 (the default): eight waves, all four k blocks each, two waves per SIMD.  KSPLIT = 2, NSTREAM = 6 (-DDC_PM_K2_SPLIT):
// twelve waves, wave = (tile stream, k half) - half the accumulators, three waves per SIMD (four would need <= 128 registers: the compiler
// spills 40) - measured 3 % SLOWER on the same box: the one-hot operand of a K step is then built twice, and occupancy was not the limit.
template <int KSPLIT, int NSTREAM>
__global__ __launch_bounds__(64 * KSPLIT * NSTREAM) void embed_pool16m_dw1_kernel(PoolMArgs p) {
    constexpr int NT = 64 * KSPLIT * NSTREAM, NKB = 4 / KSPLIT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int W = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sw = W / KSPLIT, kb0 = (W % KSPLIT) * NKB;      // tile stream, first k block of this wave
    const int fr = lane & 31, fq = lane >> 5;
    const int t = 2 + blockIdx.x / p.wg_per_type;
    const int wgi = blockIdx.x % p.wg_per_type;
    const long long n0 = (long long)wgi * p.steps_per_wg;
    const long long n1 = min(p.nr, n0 + p.steps_per_wg);
    const int cum = t == 2 ? 6 : 22;
    const int slot0 = t == 2 ? 3 : 4;
    const float s_act = p.s_act, s_grad = p.s_grad;

    // ---- W2 of the type -> LDS as the d(basic) product's B operand: element (c, k) at [piece][c >> 4][(c >> 3) & 1][k][c & 7] ----------
    {
        const float* W2t = p.W2 + (size_t)t * 128 * 128;     // [c][k]
        for (int e = tid; e < 128 * 16; e += NT) {
            const int k = e & 127, oct = e >> 7;             // channels 8 oct .. 8 oct + 7
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = W2t[(size_t)(8 * oct + j) * 128 + k];
            const Split2h sp = split8(v, p.s_w);
            const int off = (((oct >> 1) * 2 + (oct & 1)) * 128 + k) * 16;
            *reinterpret_cast<f16x8*>(smem + P2_W2P + off) = sp.h;
            *reinterpret_cast<f16x8*>(smem + P2_W2P + 32768 + off) = sp.m;
        }
    }
    const float inv_w = 1.f / p.s_w;
    const float rs = s_grad * p.s_w;                         // scale of the d(basic) accumulators
    f32x16 facc[NKB];                                        // dW1^T[f = row][k = 32 (kb0 + i) + col] x s_act s_grad; row 12: db1
#pragma unroll
    for (int i = 0; i < NKB; ++i) facc[i] = f32x16{};
    __syncthreads();                                         // the W2 image

    // Block of an item (dwords): d(xcat) pieces [128] | R [128] | arg-max bytes [128 B] | unit records [16][12] | dtu [16]
    enum { ST_D = 0, ST_R = 128, ST_A = 256, ST_X = 288, ST_DTU = 480, ST_ITEM = P2_ITEM };
    float* const stg = reinterpret_cast<float*>(smem + P2_STG) + (size_t)W * (2 * ST_ITEM);
    const long long n_pairs = (n1 - n0 + 1) / 2;
    const int e_row = fr >> 4, u_row = fr & 15;
    const float* Rt = p.R + (size_t)(t - 2) * p.nr * 128;
    const uint16_t* const mask_t = p.mask + ((size_t)(t - 2) * ((p.nr + 1) / 2)) * 256;
    struct Raw { float2 d, d2, r; float x0, x1, x2, dt; unsigned a; bool valid; };
    auto load_raw = [&](long long pi, int e) {
        Raw r;
        long long n = n0 + 2 * pi + e;
        const bool valid = n < n1;
        n = valid ? n : n1 - 1;
        const float* dx = p.dxcat + n * PM_XCAT + slot0 * 128 + 2 * lane;
        r.d = *reinterpret_cast<const float2*>(dx);
        r.d2 = t == 3 ? *reinterpret_cast<const float2*>(dx + 2 * 128) : make_float2(0.f, 0.f);
        r.r = *reinterpret_cast<const float2*>(Rt + n * 128 + 2 * lane);
        const float* xr = p.obs + n * PM_OBS + 3 + cum * 12 + lane;
        r.x0 = xr[0]; r.x1 = xr[64]; r.x2 = xr[128];
        r.dt = lane < 16 ? p.dtu[n * 40 + cum + lane] : 0.f;
        r.a = lane < 32 ? reinterpret_cast<const unsigned*>(p.amax + (n * 3 + (t - 1)) * 128)[lane] : 0u;
        r.valid = valid;
        return r;
    };
    auto store_raw = [&](const Raw& r, int e) {
        float* b = stg + e * ST_ITEM;
        const float d0 = r.valid ? r.d.x + r.d2.x : 0.f, d1 = r.valid ? r.d.y + r.d2.y : 0.f;
        *reinterpret_cast<uint2*>(b + ST_D + 2 * lane) = stage_pieces(d0, d1, s_grad);
        *reinterpret_cast<float2*>(b + ST_R + 2 * lane) = r.r;
        b[ST_X + lane] = r.x0; b[ST_X + 64 + lane] = r.x1; b[ST_X + 128 + lane] = r.x2;
        if (lane < 16) b[ST_DTU + lane] = r.valid ? r.dt : 0.f;
        if (lane < 32) reinterpret_cast<unsigned*>(b + ST_A)[lane] = r.a;
    };
    if (sw < n_pairs) {
        const Raw r0 = load_raw(sw, 0), r1 = load_raw(sw, 1);
        store_raw(r0, 0); store_raw(r1, 1);
    }
    for (long long pi = sw; pi < n_pairs; pi += NSTREAM) {              // a wave (a wave pair) takes whole pairs
        const bool more = pi + NSTREAM < n_pairs;
        Raw nx0, nx1;
        if (more) { nx0 = load_raw(pi + NSTREAM, 0); nx1 = load_raw(pi + NSTREAM, 1); }      // in flight while this pair computes; staged at the bottom
        __builtin_amdgcn_wave_barrier();
        const float* it0 = stg;
        const float* itr = it0 + e_row * ST_ITEM;

        // ---- d(basic) x s_grad s_w = demb W2: eight K steps of 16 channels, the pair's 32 rows x all 128 hidden units ---------------------
        f32x16 cacc[NKB];
#pragma unroll
        for (int i = 0; i < NKB; ++i) cacc[i] = f32x16{};
        {
            const unsigned* dr = reinterpret_cast<const unsigned*>(itr + ST_D) + 8 * fq;
            const uint8_t* ar = reinterpret_cast<const uint8_t*>(itr + ST_A) + 8 * fq;
            const char* w2l = smem + P2_W2P + (fq * 128 + 32 * kb0 + fr) * 16;
            auto build = [&](int ks) {             // the A operand of K step ks: selects of the staged pieces, re-packed h with h, m with m
                const uint4 d0 = *reinterpret_cast<const uint4*>(dr + 16 * ks), d1 = *reinterpret_cast<const uint4*>(dr + 16 * ks + 4);
                const unsigned d2[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                const uint2 ab = *reinterpret_cast<const uint2*>(ar + 16 * ks);
                unsigned sel[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int a = (int)(((j < 4 ? ab.x : ab.y) >> (8 * (j & 3))) & 0xffu);
                    sel[j] = a == u_row ? d2[j] : 0u;
                }
                u32x4 ah, am;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ah[i] = __builtin_amdgcn_perm(sel[2 * i + 1], sel[2 * i], 0x05040100u);      // the h halves of two channels
                    am[i] = __builtin_amdgcn_perm(sel[2 * i + 1], sel[2 * i], 0x07060302u);      // the m halves
                }
                Split2h o;
                o.h = __builtin_bit_cast(f16x8, ah);
                o.m = __builtin_bit_cast(f16x8, am);
                return o;
            };
            // software pipeline: the operand of K step ks + 1 is built while the 3 NKB MFMAs of K step ks run (NKB independent chains)
            Split2h Aop = build(0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                f16x8 Bh[NKB], Bm[NKB];
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    Bh[kb] = *reinterpret_cast<const f16x8*>(w2l + ks * 4096 + kb * 512);
                    Bm[kb] = *reinterpret_cast<const f16x8*>(w2l + 32768 + ks * 4096 + kb * 512);
                }
                const Split2h cur = Aop;
                if (ks + 1 < 8) Aop = build(ks + 1);
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) cacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.m, Bh[kb], cacc[kb], 0, 0, 0);
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) cacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.h, Bm[kb], cacc[kb], 0, 0, 0);
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) cacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.h, Bh[kb], cacc[kb], 0, 0, 0);
            }
        }
        // ---- + dtu[u] R[k] (the attention term, f32), through the relu (kernel 1's masks), into dW1^T / db1: K step e = item e's 16 units ----
        const uint2 mbits = *reinterpret_cast<const uint2*>(mask_t + ((size_t)((n0 >> 1) + pi) * 64 + lane) * 4);     // this lane's 4 x 16 bits
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float* it = it0 + e * ST_ITEM;
            const float4 du0 = *reinterpret_cast<const float4*>(it + ST_DTU + 4 * fq), du1 = *reinterpret_cast<const float4*>(it + ST_DTU + 8 + 4 * fq);
            const float du[8] = {du0.x, du0.y, du0.z, du0.w, du1.x, du1.y, du1.z, du1.w};
            const float* xr = it + ST_X + (4 * fq) * 12 + min(fr, 11);               // x[unit 4 fq + ..][feature fr]
            float xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xx = xr[((j & 3) + 8 * (j >> 2)) * 12];
                xv[j] = fr < 12 ? xx : (fr == 12 ? 1.f : 0.f);
            }
            const Split2h X = split8(xv, s_act);
#pragma unroll
            for (int i = 0; i < NKB; ++i) {
                const int kb = kb0 + i;
                const float Rk = it[ST_R + 32 * kb + fr] * rs;
                const unsigned mb = (kb < 2 ? mbits.x : mbits.y) >> (16 * (kb & 1));
                float bv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bv[j] = keep_if(fmaf(du[j], Rk, cacc[i][8 * e + j]), mb, 8 * e + j);
                facc[i] = mma3(X, split8(bv, inv_w), facc[i]);                      // (the accumulators carry s_w: taken out in the split)
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (more) { store_raw(nx0, 0); store_raw(nx1, 1); }
    }

    // ---- results: the waves' fold tiles summed in a fixed order through LDS -> part1[wg][13][128] ---------------------------------
    __syncthreads();                                         // every wave is done with the W2 image
    float* red = reinterpret_cast<float*>(smem + P2_RED);   // [13][128]
    for (int e = tid; e < 13 * 128; e += NT) red[e] = 0.f;
    __syncthreads();
    for (int ww = 0; ww < NSTREAM; ++ww) {                    // the KSPLIT waves of a stream own different k blocks: together
        if (sw == ww) {
#pragma unroll
            for (int i = 0; i < NKB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 8 * (r >> 2) + 4 * fq + (r & 3);
                    if (f < 13) red[f * 128 + 32 * (kb0 + i) + fr] += facc[i][r];
                }
        }
        __syncthreads();
    }
    const float inv = 1.f / (s_act * s_grad);
    float* o1 = p.part1 + (size_t)blockIdx.x * 1664;
    for (int e = tid; e < 1664; e += NT) o1[e] = red[e] * inv;
}

// Same outputs as embed_bwd_pool16 (embed_sparse.hip): slab 2 * wg_per_type x [128][128], part1 2 * wg_per_type x [13][128],
// part2 2 * wg_per_type x [128] - per-workgroup partials in the formats the dense path's reducers take.
// scratch: 2 * nr * 128 floats (R) + 2 * ceil(nr / 2) * 64 lanes x 8 bytes (the relu masks) = 384 floats per step
int embed_bwd_pool16m(const float* obs, const float* dxcat, const uint8_t* amax, const float* dtu, const float* q, int ldq,
                      const float* W1, const float* b1, const float* W2, float* slab, float* part1, float* part2, float* scratch,
                      long long nr, int wg_per_type, hipStream_t s, const F16x2Scales& f16, const uint16_t* w2t_planes) {
    PoolMArgs a{obs, dxcat, amax, dtu, q, ldq, W1, b1, W2, slab, part1, part2, nr, wg_per_type,
                (int)(((nr + wg_per_type - 1) / wg_per_type + 1) / 2 * 2),      // even: a pair never straddles two workgroups
                f16.s_act, f16.s_w, f16.s_grad, scratch, reinterpret_cast<uint16_t*>(scratch + 2 * nr * 128)};
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)embed_pool16m_dw2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, p1_lds(2));
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)embed_pool16m_dw2_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, p1_lds(3));
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)embed_pool16m_dw1_kernel<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, p2_lds(8));
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)embed_pool16m_dw1_kernel<2, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, p2_lds(12));
        if (e != hipSuccess) { set_error("embed_bwd_pool16m: hipFuncSetAttribute", (int)e); return (int)e; }
        attr = true;
    }
    // R[n][k] = sum_c q[n][c] W2_t[c][k] of every step and type: the attention term of d(basic) is dtu[u] R[k] (2 x 2 GFLOP, f32)
    // (round 6: as an f16x2 product like every other one - the row-streaming kernel, W2_t^T from policy_backward's plane pre-pass: 21 -> ~10 us each)
    for (int t = 2; t < 4; ++t) {
        float* const R = scratch + (size_t)(t - 2) * nr * 128;
        if (w2t_planes != nullptr) {
            X3Gemm g;
            g.A = q; g.a_mode = X3_ROW; g.lda = ldq;
            g.B = w2t_planes + 3 * (size_t)t * 128 * 128; g.b_mode = X3_PLANES; g.ldb = 128; g.b_plane = 128 * 128;
            g.C = R; g.ldc = 128; g.M = (int)nr; g.N = 128; g.K = 128; g.prec = 4; g.transposed_w = 1;
            g.sa = f16.s_act; g.sb = f16.s_w;
            if (gemm_x3_shape_ok(g.M, g.N, g.K, g.lda, g.ldb, g.a_mode, g.b_mode)) {
                if (int e = gemm_x3(g, s)) return e;
                continue;
            }
        }
        if (int e = gemm_f32(q, W2 + (size_t)t * 128 * 128, R, (int)nr, 128, 128, ldq, 128, 128, 0, 1, nullptr, 0, nullptr, 0, 0, 1, s)) return e;
    }
    // work = the DENSE form (what the reference's autograd computes for these units, what SURVEY.md 8(d) counts, and what executes here):
    // per step and type the two 16 x 128 x 128 products, the first layer (K = 12) and the fold (13 rows); bytes: each kernel reads the
    // step's records, d(xcat) slot(s), arg-max bytes, dtu and q or R once, plus the 256 B of relu masks written and read
    ProfScope prof("embed_bwd_pool16m", 2.0 * 2.0 * nr * (2.0 * 16 * 128 * 128 + 16.0 * 128 * 12 + 16.0 * 128 * 13),
                   2.0 * nr * (2.0 * (768 + 768 + 128 + 64) + 512 + 512 + 512), s);
#ifdef DC_PM_K1_TWO
    hipLaunchKernelGGL(embed_pool16m_dw2_kernel<2>, dim3(2 * wg_per_type), dim3(512), p1_lds(2), s, a);
#else
    hipLaunchKernelGGL(embed_pool16m_dw2_kernel<3>, dim3(2 * wg_per_type), dim3(768), p1_lds(3), s, a);
#endif
    if (int e = launch_check("embed_pool16m_dw2")) return e;
#ifdef DC_PM_K2_SPLIT      // A/B build: twelve waves of half the accumulators, three per SIMD: 666-677 us for both kernels against 645-654 (same box)
    hipLaunchKernelGGL((embed_pool16m_dw1_kernel<2, 6>), dim3(2 * wg_per_type), dim3(768), p2_lds(12), s, a);
#else
    hipLaunchKernelGGL((embed_pool16m_dw1_kernel<1, 8>), dim3(2 * wg_per_type), dim3(512), p2_lds(8), s, a);
#endif
    return launch_check("embed_pool16m_dw1");
}

}  // namespace dc
