// Backward of the per-unit embedding MLP for the two 16-unit types, third form (round 5; the default with the two-f16-piece products):
// the DENSE products on the f16 matrix cores with every operand generated on chip.
//
// Replaces, for these types, the part of /root/reference/optimizer.py:672 (autograd) that flows through policy.py:102-136,152 - like
// embed_sparse.hip, whose header states the algebra.  embed_sparse.hip executes the sparse form (one non-zero of d(emb) per step, type
// and channel: 1/16 of the dense MACs) as per-channel gathers on the packed-f32 VALU and is bound by their latency: sorted channel lists,
// dependent LDS round trips, a barrier per step, 0.20 of the vector peak after three rounds of tuning.  The matrix cores do sixteen
// times the VALU's MACs per cycle, so here the DENSE form runs instead - per env-step and type
//     dW2^T[k][c]  += sum_u basic[u][k] demb[u][c]          K = 16 units
//     dbasic[u][k]  = sum_c demb[u][c] W2[c][k]             K = 128 channels
//     dW1^T[f][k]  += sum_u x[u][f] relu'(.) dbasic[u][k]   K = 16 units, f = 12: ones (db1)
// with demb[u][c] = [amax(c) == u] d(xcat)[c] + dtu[u] q[c] - and NOTHING of it ever exists in memory: a lane builds the 8 operand
// elements an MFMA wants from it out of the arg-max bytes, d(xcat), dtu and the attention query (all read where the forward / the loss
// left them), basic comes out of the first-layer MFMA in exactly the register layout the next product takes as its A operand, the
// relu-masked d(basic) in the layout the fold takes as its B operand.  No prepare pass, no R = q W2 product, no sorted lists, no LDS
// gathers, no barrier inside the loop: a wave only ever consumes what it computed itself or what is read-only.
//
// Arithmetic: two f16 pieces per f32 operand, three v_mfma_f32_32x32x16_f16 per product (hh, hm, mh), f32 accumulate - gemm_x3.hip's
// PREC 4 with the same power-of-two pre-scales (activations and records s_act, weights s_w, gradients s_grad); the first layer is
// the forward's instruction sequence (embed_fused.hip, F16), so the relu mask is the forward's bit for bit.
//
// Work split.  Workgroup = one type x a contiguous range of env-steps, 512 threads.  Wave W = (stream st = W >> 2, k quarter kq = W & 3):
// a stream takes every other PAIR of env-steps (a pair's 2 x 16 units are the 32 rows of an MFMA tile), a wave owns the 32 hidden units
// k = 32 kq .. 32 kq + 31 for everything - its slice of basic, of dW2^T (4 accumulator tiles, the whole kernel), of d(basic) and of the
// dW1 fold.  The four waves of a stream build the same demb operands redundantly (VALU work in the shadow of their own MFMAs); the two
// streams' accumulators meet once, at the end, through LDS.  Register layouts (lane = (fr = lane & 31, fq = lane >> 5)):
//     A operand: A[row fr][K slot 8 fq + j], B operand: B[K slot 8 fq + j][col fr], D: register r = D[row 8 (r >> 2) + 4 fq + (r & 3)][col fr]
// The first layer's D registers hold, for item e of the pair, units sigma(fq, j) = 4 fq + (j & 3) + 8 (j >> 2) in registers 8 e + j: K slot
// 8 fq + j of the unit-contracting products stands for unit sigma(fq, j), and those products take D registers as operands as they are.
// tools/pool16m_sim.py models exactly this index math lane by lane against a dense float64 evaluation.
#include "kernels.h"
#include "gemm_tiles.h"

namespace dc {
namespace {

enum { PM_THREADS = 512, PM_OBS = 483, PM_XCAT = 896 };
enum : int {   // LDS (bytes)
    PM_W2P = 0,                      // [piece 2][K step 8][fq 2][k 128][8 channels] f16: the B operand of the d(basic) product (W2 x s_w)
    PM_RED_ACC = 0,                  // at the end, over it: stream 1's dW2^T tiles [kq 4][64 registers][64 lanes] f32
    PM_RED_F = 65536,                // ... its fold tiles [kq 4][16][64] f32
    PM_RED_B = PM_RED_F + 16384,     // ... its bias-gradient sums [4][64] f32
    PM_LDS = PM_RED_B + 1024
};

struct PoolMArgs {
    const float* obs; const float* dxcat; const uint8_t* amax; const float* dtu; const float* q; int ldq;
    const float* W1; const float* b1; const float* W2;
    float* slab; float* part1; float* part2;
    long long nr; int wg_per_type; int steps_per_wg;
    float s_act, s_w, s_grad;
};

__device__ __forceinline__ int sigma_unit(int fq, int j) { return 4 * fq + (j & 3) + 8 * (j >> 2); }

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

}  // namespace

__global__ __launch_bounds__(PM_THREADS) void embed_bwd_pool16m_kernel(PoolMArgs p) {
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

    // ---- W2 of the type -> LDS as the d(basic) product's B operand: element (c, k) at [piece][c >> 4][(c >> 3) & 1][k][c & 7] ----------
    {
        const float* W2t = p.W2 + (size_t)t * 128 * 128;     // [c][k]
        for (int e = tid; e < 128 * 16; e += PM_THREADS) {
            const int k = e & 127, oct = e >> 7;             // channels 8 oct .. 8 oct + 7
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = W2t[(size_t)(8 * oct + j) * 128 + k];
            const Split2h sp = split8(v, p.s_w);
            const int off = (((oct >> 1) * 2 + (oct & 1)) * 128 + k) * 16;
            *reinterpret_cast<f16x8*>(smem + PM_W2P + off) = sp.h;
            *reinterpret_cast<f16x8*>(smem + PM_W2P + 32768 + off) = sp.m;
        }
    }
    // ---- W1 rows of this wave's k block as the first layer's B operand (the forward's: x 2^8, features 12..15 zero), b1 x s_act ----------
    Split2h w1;
    {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 8 * fq + j < 12 ? p.W1[(32 * kq + fr) * 12 + 8 * fq + j] : 0.f;
        w1 = split8(v, 256.f);
    }
    const float b1s = p.b1[32 * kq + fr] * s_act;
    const float ginv = 1.f / 256.f;                          // (x s_act)(W1 2^8) -> basic x s_act  (embed_fused.hip, F16)
    const float inv_w = 1.f / p.s_w;

    f32x16 acc[4];                                           // dW2^T[k = 32 kq + row][c = 32 cb + col] x s_act s_grad
    f32x16 facc;                                             // dW1^T[f = row][k = 32 kq + col] x s_act s_grad; row 12: db1
#pragma unroll
    for (int r = 0; r < 16; ++r) { facc[r] = 0.f; acc[0][r] = 0.f; acc[1][r] = 0.f; acc[2][r] = 0.f; acc[3][r] = 0.f; }
    float db2a[4] = {0.f, 0.f, 0.f, 0.f};                    // kq == 0: sum over steps of demb's column sums, channel 32 cb + fr
    __syncthreads();                                         // the W2 image

    const long long n_pairs = (n1 - n0 + 1) / 2;
    const int e_row = fr >> 4, u_row = fr & 15;              // as a ROW of the pair's tile this lane is unit u_row of item e_row
    float sink = 0.f, pf = 0.f;
    for (long long pi = st; pi < n_pairs; pi += 2) {
        const long long nA = n0 + 2 * pi, nB = nA + 1;
        const bool validB = nB < n1;                         // wave-uniform
        const long long itB = validB ? nB : nA;
        const long long n_of[2] = {nA, itB};
        const long long n_row = e_row ? itB : nA;
        const bool row_valid = e_row == 0 || validB;

        // pull the NEXT pair's lines towards the caches: one load instruction, a different 128-byte line per lane
        {
            const long long nn = min(n0 + 2 * (pi + 2) + (lane >= 32 ? 1 : 0), n1 - 1);
            const int l5 = lane & 31;
            const float* a = l5 < 4 ? p.dxcat + nn * PM_XCAT + slot0 * 128 + 32 * l5
                           : l5 < 8 ? p.dxcat + nn * PM_XCAT + 6 * 128 + 32 * (l5 - 4)
                           : l5 < 12 ? p.q + nn * p.ldq + 32 * (l5 - 8)
                           : l5 < 19 ? p.obs + nn * PM_OBS + 3 + cum * 12 + 32 * (l5 - 12) - (l5 == 18 ? 5 : 0)
                           : l5 == 19 ? p.dtu + nn * 40 + cum
                           : reinterpret_cast<const float*>(p.amax + (nn * 3 + (t - 1)) * 128);
            sink += pf;                                       // last iteration's: waited for a whole pair later
            pf = l5 < 21 ? *a : 0.f;
        }

        // ---- first layer: basic x s_act of the pair's 32 rows, this wave's 32 hidden units ----------------------------------------
        f32x16 basic;
        {
            const float* xp = p.obs + n_row * PM_OBS + 3 + (cum + u_row) * 12 + 8 * fq;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = xp[e];
#pragma unroll
            for (int e = 4; e < 8; ++e) v[e] = fq ? 0.f : xp[e];                   // features 12..15 do not exist
            const Split2h x = split8(v, s_act);
            f32x16 g;
#pragma unroll
            for (int r = 0; r < 16; ++r) g[r] = 0.f;
            g = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.m, w1.h, g, 0, 0, 0);     // the forward's sequence
            g = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.h, w1.m, g, 0, 0, 0);
            g = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.h, w1.h, g, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) basic[r] = relu_nan(fmaf(g[r], ginv, b1s));
        }

        // ---- dW2^T += basic^T demb, item by item (K = the item's 16 units) -------------------------------------------------------------
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const long long n = n_of[e];
            const bool valid = e == 0 || validB;
            float bv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = basic[8 * e + j];                  // K slot 8 fq + j <-> unit sigma(fq, j): as they lie
            const Split2h A = split8_noscale(bv);
            float du[8];                                                           // dtu of the lane group's eight units
            {
                const float* dp = p.dtu + n * 40 + cum + 4 * fq;
#pragma unroll
                for (int j = 0; j < 8; ++j) du[j] = valid ? dp[(j & 3) + 8 * (j >> 2)] : 0.f;
            }
            float sumdtu = 0.f;
            if (kq == 0) {                                                         // wave-uniform: the bias gradient rides on wave (st, 0)
                const float v = p.dtu[n * 40 + cum + (lane & 15)];
                float sum = v;
                sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x128, 0xf, 0xf, true));
                sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x124, 0xf, 0xf, true));
                sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x122, 0xf, 0xf, true));
                sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x121, 0xf, 0xf, true));
                sumdtu = valid ? sum : 0.f;
            }
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int c = 32 * cb + fr;
                const float* dx = p.dxcat + n * PM_XCAT + c;
                float d = dx[slot0 * 128];
                if (t == 3) d += dx[6 * 128];
                d = valid ? d : 0.f;
                const float qc = p.q[n * p.ldq + c];
                const int a = p.amax[(n * 3 + (t - 1)) * 128 + c];
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaf(du[j], qc, a == sigma_unit(fq, j) ? d : 0.f);
                acc[cb] = mma3(A, split8(v, s_grad), acc[cb]);
                if (kq == 0) db2a[cb] += fmaf(qc, sumdtu, d);
            }
        }

        // ---- d(basic) x s_grad s_w = demb W2: eight K steps of 16 channels, the pair's 32 rows ------------------------------------------
        f32x16 cacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) cacc[r] = 0.f;
        {
            const float du = row_valid ? p.dtu[n_row * 40 + cum + u_row] : 0.f;
            const float* dxr = p.dxcat + n_row * PM_XCAT + slot0 * 128 + 8 * fq;
            const float* qr = p.q + n_row * p.ldq + 8 * fq;
            const uint8_t* ar = p.amax + (n_row * 3 + (t - 1)) * 128 + 8 * fq;
            const char* w2l = smem + PM_W2P + (fq * 128 + 32 * kq + fr) * 16;
#pragma unroll 2
            for (int ks = 0; ks < 8; ++ks) {
                const float4 d0 = *reinterpret_cast<const float4*>(dxr + 16 * ks), d1 = *reinterpret_cast<const float4*>(dxr + 16 * ks + 4);
                float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                if (t == 3) {
                    const float4 e0 = *reinterpret_cast<const float4*>(dxr + 16 * ks + 2 * 128), e1 = *reinterpret_cast<const float4*>(dxr + 16 * ks + 2 * 128 + 4);
                    d[0] += e0.x; d[1] += e0.y; d[2] += e0.z; d[3] += e0.w; d[4] += e1.x; d[5] += e1.y; d[6] += e1.z; d[7] += e1.w;
                }
                const float4 q0 = *reinterpret_cast<const float4*>(qr + 16 * ks), q1 = *reinterpret_cast<const float4*>(qr + 16 * ks + 4);
                const float qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                const uint2 ab = *reinterpret_cast<const uint2*>(ar + 16 * ks);
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int a = (int)(((j < 4 ? ab.x : ab.y) >> (8 * (j & 3))) & 0xffu);
                    v[j] = row_valid ? fmaf(du, qv[j], a == u_row ? d[j] : 0.f) : 0.f;
                }
                Split2h B;
                B.h = *reinterpret_cast<const f16x8*>(w2l + ks * 4096);
                B.m = *reinterpret_cast<const f16x8*>(w2l + 32768 + ks * 4096);
                cacc = mma3(split8(v, s_grad), B, cacc);
            }
        }
        // ---- through the relu, then dW1^T / db1 += x^T d(basic): K step e = item e's 16 units ----------------------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) cacc[r] = basic[r] > 0.f ? cacc[r] * inv_w : 0.f;      // = relu'(.) d(basic) x s_grad
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float bv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = cacc[8 * e + j];
            const float* xr = p.obs + n_of[e] * PM_OBS + 3 + (cum + 4 * fq) * 12 + fr;     // x[unit 4 fq + ..][feature fr]
            float xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = fr < 12 ? xr[((j & 3) + 8 * (j >> 2)) * 12] : (fr == 12 ? 1.f : 0.f);
            facc = mma3(split8(xv, s_act), split8_noscale(bv), facc);
        }
    }

    // ---- results: the two streams meet through LDS (stream 1 writes, stream 0 adds and stores) ---------------------------------------------
    __syncthreads();                                         // every wave is done with the W2 image
    float* racc = reinterpret_cast<float*>(smem + PM_RED_ACC) + (size_t)kq * 64 * 64 + lane;
    float* rf = reinterpret_cast<float*>(smem + PM_RED_F) + (size_t)kq * 16 * 64 + lane;
    float* rb = reinterpret_cast<float*>(smem + PM_RED_B);
    if (st == 1) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) racc[(16 * cb + r) * 64] = acc[cb][r];
#pragma unroll
        for (int r = 0; r < 16; ++r) rf[r * 64] = facc[r];
        if (kq == 0) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) rb[cb * 64 + lane] = db2a[cb];
        }
    }
    __syncthreads();
    if (st == 0) {
        const float inv = 1.f / (s_act * s_grad);
        float* out = p.slab + (size_t)blockIdx.x * 128 * 128;            // [c][k]
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 v;
                v.x = (acc[cb][4 * g4 + 0] + racc[(16 * cb + 4 * g4 + 0) * 64]) * inv;
                v.y = (acc[cb][4 * g4 + 1] + racc[(16 * cb + 4 * g4 + 1) * 64]) * inv;
                v.z = (acc[cb][4 * g4 + 2] + racc[(16 * cb + 4 * g4 + 2) * 64]) * inv;
                v.w = (acc[cb][4 * g4 + 3] + racc[(16 * cb + 4 * g4 + 3) * 64]) * inv;
                *reinterpret_cast<float4*>(out + (size_t)(32 * cb + fr) * 128 + 32 * kq + 8 * g4 + 4 * fq) = v;
            }
        float* o1 = p.part1 + (size_t)blockIdx.x * 1664;                  // [13][128]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 8 * (r >> 2) + 4 * fq + (r & 3);
            if (f < 13) o1[f * 128 + 32 * kq + fr] = (facc[r] + rf[r * 64]) * inv;
        }
        if (kq == 0 && fq == 0) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) p.part2[(size_t)blockIdx.x * 128 + 32 * cb + fr] = db2a[cb] + rb[cb * 64 + lane];
        }
    }
    if (sink + pf == 1.2345e-33f && p.nr < 0) p.part2[0] = sink;             // keeps the cache-warming loads alive; never true
}

// Same outputs as embed_bwd_pool16 (embed_sparse.hip): slab 2 * wg_per_type x [128][128], part1 2 * wg_per_type x [13][128],
// part2 2 * wg_per_type x [128] - per-workgroup partials in the formats the dense path's reducers take.
int embed_bwd_pool16m(const float* obs, const float* dxcat, const uint8_t* amax, const float* dtu, const float* q, int ldq,
                      const float* W1, const float* b1, const float* W2, float* slab, float* part1, float* part2,
                      long long nr, int wg_per_type, hipStream_t s, const F16x2Scales& f16) {
    PoolMArgs a{obs, dxcat, amax, dtu, q, ldq, W1, b1, W2, slab, part1, part2, nr, wg_per_type,
                (int)(((nr + wg_per_type - 1) / wg_per_type + 1) / 2 * 2),      // even: a pair never straddles two workgroups
                f16.s_act, f16.s_w, f16.s_grad};
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)embed_bwd_pool16m_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PM_LDS);
        if (e != hipSuccess) { set_error("embed_bwd_pool16m: hipFuncSetAttribute", (int)e); return (int)e; }
        attr = true;
    }
    // algorithmic work = the sparse form's (embed_sparse.hip counts the same): basic + dW1 fold 2 x 16 x 128 x 12 MACs, the two gathers
    // 2 x 128 x 128 MACs per step and type; what EXECUTES is the dense form, 16 x the gathers' MACs, on the matrix cores
    ProfScope prof("embed_bwd_pool16", 2.0 * 2.0 * nr * (2.0 * 16 * 128 * 12 + 2.0 * 128 * 128),
                   4.0 * 2.0 * nr * (16 * 12 + 3 * 128 + 16 + 32), s);
    hipLaunchKernelGGL(embed_bwd_pool16m_kernel, dim3(2 * wg_per_type), dim3(PM_THREADS), PM_LDS, s, a);
    return launch_check("embed_bwd_pool16m");
}

}  // namespace dc
