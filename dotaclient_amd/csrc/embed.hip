// Per-unit embedding front end of the actor-critic network and its backward.
//
// Replaces /root/reference/policy.py:97-136 (forward) and the autograd products torch derives from it
// at /root/reference/optimizer.py:672.
//
// HBM layout ("type-major"): the 40 units of an env-step are regrouped by unit type so that the
// second embedding layer of each type (its own 128x128 weight, policy.py:58-63) is ONE dense GEMM:
//     basic/emb/demb [ type t ][ n * U_t + u_local ][128]   rows of type t start at row NR*CUM_t
// with U = {1,5,16,16,1,1}, CUM = {0,1,6,22,38,39}.  The flattened observation row is
//     obs[n][483] = env(3) | unit 0..39 x 12 features  (layout.py).
//
// Kernels (all HBM-bound, coalesced along the 128 embedding channels):
//   unit_basic_fwd : basic = relu(W1 x + b1)   (K = 12: VALU, 512 B written per unit)
//   pool_env_fwd   : env embedding + per-type max-pool (+ argmax) -> xcat[n][896]
//                    (keeps the reference's bug: slot 6 is the max over ENEMY NON-HEROES, policy.py:127)
//   embed_scatter_bwd : demb = dtu*q (attention keys, policy.py:152) + max-pool routing of dxcat,
//                    plus the env-embedding weight gradient
//   unit_basic_bwd : dW1/db1 from the (already relu-masked) dbasic
#include "kernels.h"

namespace dc {

__constant__ int c_type_units[6] = {1, 5, 16, 16, 1, 1};
__constant__ int c_type_cum[7] = {0, 1, 6, 22, 38, 39, 40};

enum { OBS_DIM = 483, EMB = 128, XCAT = 896 };

__device__ __forceinline__ int unit_type_of_row(long long row, long long nr, long long* local) {
    // row in [0, 40*nr): which type block does it fall in
    int t = 0;
#pragma unroll
    for (int i = 1; i < 6; ++i)
        if (row >= nr * c_type_cum[i]) t = i;
    *local = row - nr * c_type_cum[t];
    return t;
}

// Walks type-major rows [r0, r1) keeping (type, env-step n, local unit ul) incrementally (one 64-bit
// division per block instead of one per row).
struct RowCursor {
    int t, U, ul;
    long long n;
    long long next_type_row;   // first row of type t+1
    __device__ __forceinline__ void init(long long row, long long nr) {
        long long local;
        t = unit_type_of_row(row, nr, &local);
        U = c_type_units[t];
        n = local / U;
        ul = (int)(local - n * U);
        next_type_row = nr * c_type_cum[t + 1];
    }
    __device__ __forceinline__ void advance(long long new_row, long long nr) {
        if (new_row >= next_type_row) { init(new_row, nr); return; }
        if (++ul == U) { ul = 0; ++n; }
    }
    __device__ __forceinline__ const float* x(const float* obs) const {
        return obs + n * OBS_DIM + 3 + (c_type_cum[t] + ul) * 12;
    }
};

// 256 threads = 2 interleaved row streams x 128 channels; each block owns a contiguous range of rows
__global__ __launch_bounds__(256) void unit_basic_fwd_kernel(const float* __restrict__ obs, const float* __restrict__ W1,
                                                             const float* __restrict__ b1, float* __restrict__ basic,
                                                             long long nr, int rows_per_block) {
    const int c = threadIdx.x & 127;
    const int sub = threadIdx.x >> 7;
    float w[12];
#pragma unroll
    for (int f = 0; f < 12; ++f) w[f] = W1[c * 12 + f];
    const float b = b1[c];
    const long long total = nr * 40;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(total, r0 + rows_per_block);
    // the two halves of the block take the two halves of the range
    const long long mid = r0 + ((r1 - r0 + 1) >> 1);
    long long row = sub == 0 ? r0 : mid;
    const long long end = sub == 0 ? mid : r1;
    if (row >= end) return;
    RowCursor cur;
    cur.init(row, nr);
    for (; row < end;) {
        const float* x = cur.x(obs);
        float acc = b;
#pragma unroll
        for (int f = 0; f < 12; ++f) acc = fmaf(x[f], w[f], acc);
        basic[row * EMB + c] = relu_nan(acc);
        ++row;
        cur.advance(row, nr);
    }
}

// one env-step per 128 threads
__global__ __launch_bounds__(256) void pool_env_fwd_kernel(const float* __restrict__ obs, const float* __restrict__ emb,
                                                           const float* __restrict__ Wenv, const float* __restrict__ benv,
                                                           float* __restrict__ xcat, uint8_t* __restrict__ amax,
                                                           long long nr, long long nrp, int residual) {
    // nrp: rows per unit of the type-major emb blocks (nr padded to a multiple of 128 on the fused path)
    // residual != 0: embed_fwd_fused's epilogue already pooled every type except eh (t = 1)
    const int c = threadIdx.x & 127;
    const int sub = threadIdx.x >> 7;
    const float w0 = Wenv[c * 3 + 0], w1 = Wenv[c * 3 + 1], w2 = Wenv[c * 3 + 2], be = benv[c];
    for (long long n = (long long)blockIdx.x * 2 + sub; n < nr; n += (long long)gridDim.x * 2) {
        const float* e = obs + n * OBS_DIM;
        float* xo = xcat + n * XCAT;
        xo[c] = relu_nan(fmaf(e[2], w2, fmaf(e[1], w1, fmaf(e[0], w0, be))));  // policy.py:97
        if (residual) {
            const float* p = emb + (nrp * c_type_cum[1] + n * 5) * EMB + c;
            float m = p[0];
            int am = 0;
#pragma unroll
            for (int u = 1; u < 5; ++u) {
                const float v = p[(long long)u * EMB];
                am = v > m ? u : am;
                m = max_nan(m, v);                  // a NaN unit makes the pooled value NaN, like torch.max
            }
            amax[(n * 3 + 0) * EMB + c] = (uint8_t)am;
            xo[2 * EMB + c] = m;
            continue;
        }
        float enh_max = 0.f;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int U = c_type_units[t];
            const float* p = emb + (nrp * c_type_cum[t] + n * U) * EMB + c;
            float m = p[0];
            int am = 0;
            for (int u = 1; u < U; ++u) {
                const float v = p[(long long)u * EMB];
                am = v > m ? u : am;            // first maximum wins, like torch.max
                m = max_nan(m, v);              // ... and a NaN unit makes the pooled value NaN, like torch.max
            }
            if (t == 3) enh_max = m;
            if (t >= 1 && t <= 3) amax[(n * 3 + (t - 1)) * EMB + c] = (uint8_t)am;
            if (t < 5) xo[(1 + t) * EMB + c] = m;
        }
        xo[6 * EMB + c] = enh_max;  // policy.py:127: eth_embedding_max is computed from enh_embedding
    }
}

// demb[type-major row][c] = dtu[n][u] * q[n][c]  +  pool routing of dxcat[n][.]
// Also accumulates, per block and then with one atomic per value: dW_env[128][3], db_env[128]
// (relu-masked by the stored env embedding) and the six second-layer bias gradients db2[t][c] (column
// sums of demb, which would otherwise need another 335 MB pass over it).
__global__ __launch_bounds__(256) void embed_scatter_bwd_kernel(
    const float* __restrict__ obs, const float* __restrict__ xcat, const float* __restrict__ dxcat,
    const float* __restrict__ dtu, const float* __restrict__ q, int ldq, const uint8_t* __restrict__ amax,
    float* __restrict__ demb, float* __restrict__ partials, long long nr, long long nrp, int steps_per_block, int skip16) {
    constexpr int kU[6] = {1, 5, 16, 16, 1, 1};
    constexpr int kCum[7] = {0, 1, 6, 22, 38, 39, 40};
    const int c = threadIdx.x & 127;
    const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);   // wave-uniform: dtu loads go scalar
    const long long n0 = (long long)blockIdx.x * steps_per_block;
    const long long n1 = min(nr, n0 + steps_per_block);
    float gw0 = 0.f, gw1 = 0.f, gw2 = 0.f, gb = 0.f;
    float gb2[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long n = n0 + sub; n < n1; n += 2) {
        const float* dx = dxcat + n * XCAT;
        // env embedding backward (policy.py:97)
        const float de = (xcat[n * XCAT + c] > 0.f) ? dx[c] : 0.f;
        const float* e = obs + n * OBS_DIM;
        gw0 = fmaf(de, e[0], gw0); gw1 = fmaf(de, e[1], gw1); gw2 = fmaf(de, e[2], gw2); gb += de;
        const float qc = q[n * ldq + c];
        const float* dt = dtu + n * 40;
        float pool[6];
        pool[0] = dx[1 * EMB + c];
        pool[1] = dx[2 * EMB + c];
        pool[2] = dx[3 * EMB + c];
        pool[3] = dx[4 * EMB + c] + dx[6 * EMB + c];   // enh feeds slots 4 and 6 (policy.py:127)
        pool[4] = dx[5 * EMB + c];
        pool[5] = 0.f;                                 // eth is never pooled
        int am[6] = {0, 0, 0, 0, 0, 0};
        am[1] = amax[(n * 3 + 0) * EMB + c];
        am[2] = amax[(n * 3 + 1) * EMB + c];
        am[3] = amax[(n * 3 + 2) * EMB + c];
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            if (skip16 && (t == 2 || t == 3)) continue;     // handled by embed_bwd_pool16 without materialising d(emb)
            float* p = demb + (nrp * kCum[t] + n * kU[t]) * EMB + c;
            float sum_dt = 0.f;
#pragma unroll
            for (int u = 0; u < kU[t]; ++u) {
                const float d = dt[kCum[t] + u];
                sum_dt += d;
                float g = d * qc;
                if (u == am[t]) g += pool[t];
                if (skip16 < 2) p[u * EMB] = g;        // (2: the small types' backward forms its d(emb) on chip too, embed_small.hip)
            }
            gb2[t] += sum_dt * qc + pool[t];           // column sum of this step's demb rows of type t
        }
    }
    // per-block partials -> scratch[block][10][128] (contended atomics on 1280 addresses from thousands of
    // blocks cost more than the kernel itself); combine the block's two halves through LDS first
    __shared__ float sh[10][128];
    if (sub == 1) {
        sh[0][c] = gw0; sh[1][c] = gw1; sh[2][c] = gw2; sh[3][c] = gb;
#pragma unroll
        for (int t = 0; t < 6; ++t) sh[4 + t][c] = gb2[t];
    }
    __syncthreads();
    if (sub == 0) {
        float* o = partials + (size_t)blockIdx.x * 1280;
        o[0 * 128 + c] = gw0 + sh[0][c]; o[1 * 128 + c] = gw1 + sh[1][c]; o[2 * 128 + c] = gw2 + sh[2][c];
        o[3 * 128 + c] = gb + sh[3][c];
#pragma unroll
        for (int t = 0; t < 6; ++t) o[(4 + t) * 128 + c] = gb2[t] + sh[4 + t][c];
    }
}

// embed_scatter_bwd's env-only form (every type's bias gradient comes from its on-chip backward kernel): the env embedding's backward
// (policy.py:97) alone - per env-step 512 B of d(xcat), 512 B of xcat (the relu mask), 12 B of the observation.  Eight rows per thread in
// flight (the general kernel walks its rows one at a time behind dependent loads: 47 us for these 68 MB).  Partials [block][0..3][128].
__global__ __launch_bounds__(256) void embed_env_bwd_kernel(const float* __restrict__ obs, const float* __restrict__ xcat,
                                                            const float* __restrict__ dxcat, float* __restrict__ partials, long long nr,
                                                            int steps_per_block) {
    const int c = threadIdx.x & 127;
    const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);
    const long long n0 = (long long)blockIdx.x * steps_per_block;
    const long long n1 = min(nr, n0 + steps_per_block);
    float gw0 = 0.f, gw1 = 0.f, gw2 = 0.f, gb = 0.f;
    for (long long n = n0 + sub; n < n1; n += 16) {
        float de[8], e0[8], e1[8], e2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool valid = n + 2 * i < n1;
            const long long m = valid ? n + 2 * i : n1 - 1;
            const float dx = dxcat[m * XCAT + c], x = xcat[m * XCAT + c];
            const float* e = obs + m * OBS_DIM;
            e0[i] = e[0]; e1[i] = e[1]; e2[i] = e[2];
            de[i] = (valid && x > 0.f) ? dx : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { gw0 = fmaf(de[i], e0[i], gw0); gw1 = fmaf(de[i], e1[i], gw1); gw2 = fmaf(de[i], e2[i], gw2); gb += de[i]; }
    }
    __shared__ float sh[4][128];
    if (sub == 1) { sh[0][c] = gw0; sh[1][c] = gw1; sh[2][c] = gw2; sh[3][c] = gb; }
    __syncthreads();
    if (sub == 0) {
        float* o = partials + (size_t)blockIdx.x * 1280;
        o[0 * 128 + c] = gw0 + sh[0][c]; o[1 * 128 + c] = gw1 + sh[1][c]; o[2 * 128 + c] = gw2 + sh[2][c]; o[3 * 128 + c] = gb + sh[3][c];
    }
}

// sum of p[b * stride + idx] over b = b0, b0 + step, .. < n: eight loads in flight (a plain loop is a chain of dependent round trips: the
// scatter reduce took 19 us for 4 MB, the tail reduce 15 us)
__device__ __forceinline__ float strided_sum(const float* __restrict__ p, size_t stride, int idx, int b0, int step, int n) {
    float acc = 0.f;
    int b = b0;
    for (; b + 7 * step < n; b += 8 * step) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = p[(size_t)(b + i * step) * stride + idx];
        acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; b < n; b += step) acc += p[(size_t)b * stride + idx];
    return acc;
}

// stage 2: dWenv[c][f] / dbenv[c] / db2[t][c] += sum over blocks
__global__ __launch_bounds__(256) void embed_scatter_reduce_kernel(const float* __restrict__ partials, int nblk,
                                                                   float* __restrict__ dWenv, float* __restrict__ dbenv,
                                                                   float* __restrict__ db2) {
    const int idx = blockIdx.x * 256 + threadIdx.x;   // 0..1279
    if (idx >= 1280) return;
    // blockIdx.y strides over the partial blocks: 32 atomics per output instead of thousands
    const float acc = strided_sum(partials, 1280, idx, blockIdx.y, gridDim.y, nblk);
    const int k = idx >> 7, c = idx & 127;
    if (k < 3) atomicAdd(&dWenv[c * 3 + k], acc);
    else if (k == 3) atomicAdd(&dbenv[c], acc);
    else atomicAdd(&db2[(k - 4) * EMB + c], acc);
}

// dW1[c][f] += sum_rows dbasic[row][c] * x[row][f];  db1[c] += sum_rows dbasic[row][c]
// thread = (channel c, feature half h): 6 weight accumulators (+ bias on h == 0)
__global__ __launch_bounds__(256) void unit_basic_bwd_kernel(const float* __restrict__ obs,
                                                             const float* __restrict__ dbasic,
                                                             float* __restrict__ partials, long long nr,
                                                             int rows_per_block) {
    const int c = threadIdx.x & 127;
    const int h = threadIdx.x >> 7;
    const long long total = nr * 40;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(total, r0 + rows_per_block);
    if (r0 >= r1) return;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float accb = 0.f;
    RowCursor cur;
    cur.init(r0, nr);
    for (long long row = r0; row < r1;) {
        const float* x = cur.x(obs) + h * 6;
        const float g = dbasic[row * EMB + c];
#pragma unroll
        for (int f = 0; f < 6; ++f) acc[f] = fmaf(g, x[f], acc[f]);
        accb += g;
        ++row;
        cur.advance(row, nr);
    }
    float* o = partials + (size_t)blockIdx.x * 1664;   // [13][128]: 12 features + bias
#pragma unroll
    for (int f = 0; f < 6; ++f) o[(h * 6 + f) * 128 + c] = acc[f];
    if (h == 0) o[12 * 128 + c] = accb;
}

__global__ __launch_bounds__(256) void unit_basic_reduce_kernel(const float* __restrict__ partials, int nblk,
                                                                float* __restrict__ dW1, float* __restrict__ db1) {
    const int idx = blockIdx.x * 256 + threadIdx.x;   // 0..1663
    if (idx >= 1664) return;
    float acc = 0.f;
    for (int b = blockIdx.y; b < nblk; b += gridDim.y) acc += partials[(size_t)b * 1664 + idx];
    const int f = idx >> 7, c = idx & 127;
    if (f < 12) atomicAdd(&dW1[c * 12 + f], acc);
    else atomicAdd(&db1[c], acc);
}

// out[j] += sum_rows X[row][j]   (bias gradients).  blockDim = 256 = TX column lanes x TY row lanes.
template <int TX>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int ld, long long rows, int cols,
                                                     float* __restrict__ out, int rows_per_block) {
    constexpr int TY = 256 / TX;
    __shared__ float sh[256];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = min(rows, r0 + rows_per_block);
    const int j = blockIdx.x * TX + tx;
    float acc = 0.f;
    if (j < cols) {
        long long r = r0 + ty;
        for (; r + 7 * TY < r1; r += 8 * TY) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = X[(r + i * TY) * ld + j];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += v[i];
        }
        for (; r < r1; r += TY) acc += X[r * ld + j];
    }
    if (TY > 1) {
        sh[threadIdx.x] = acc;
        __syncthreads();
        if (ty == 0) {
#pragma unroll
            for (int i = 1; i < TY; ++i) acc += sh[i * TX + tx];
        }
    }
    if (ty == 0 && j < cols) atomicAdd(&out[j], acc);
}

static inline int grid_for(long long items, int per_block, int cap) {
    long long g = (items + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

int unit_basic_fwd(const float* obs, const float* W1, const float* b1, float* basic, long long nr, hipStream_t s) {
    const long long total = nr * 40;
    int rpb = (int)((total + 8191) / 8192);
    if (rpb < 16) rpb = 16;
    hipLaunchKernelGGL(unit_basic_fwd_kernel, dim3((unsigned)((total + rpb - 1) / rpb)), dim3(256), 0, s, obs, W1, b1, basic,
                       nr, rpb);
    return launch_check("unit_basic_fwd");
}

int pool_env_fwd(const float* obs, const float* emb, const float* Wenv, const float* benv, float* xcat, uint8_t* amax,
                 long long nr, long long nrp, int residual, hipStream_t s) {
    ProfScope prof("pool_env_fwd", 0.0, 4.0 * nr * ((residual ? 5 : 40) * 128 + 896), s);
    hipLaunchKernelGGL(pool_env_fwd_kernel, dim3(grid_for(nr, 2, 256 * 16)), dim3(256), 0, s, obs, emb, Wenv, benv, xcat,
                       amax, nr, nrp, residual);
    return launch_check("pool_env_fwd");
}

int embed_scatter_bwd(const float* obs, const float* xcat, const float* dxcat, const float* dtu, const float* q, int ldq,
                      const uint8_t* amax, float* demb, float* dWenv, float* dbenv, float* db2, float* scratch,
                      long long nr, long long nrp, int skip16, hipStream_t s) {
    int spb = (int)((nr + 2047) / 2048);
    if (spb < 4) spb = 4;
    const int nblk = (int)((nr + spb - 1) / spb);        // <= 2048 -> <= 10.5 MB of scratch
    // algorithmic bytes: d(emb) rows written for the units this pass owns (8 of 40 with skip16), xcat / dxcat rows, the
    // attention query, the target-unit gradients
    const double units = skip16 >= 2 ? 0.0 : (skip16 ? 8.0 : 40.0);
    // (env only: the env slots of xcat and d(xcat) and three observation floats per step)
    ProfScope prof("embed_scatter_bwd(+reduce)", skip16 == 3 ? 2.0 * nr * 128 * 4 : 2.0 * nr * units * 128,
                   skip16 == 3 ? 4.0 * nr * (128 * 2 + 3) : 4.0 * nr * (units * 128 + 896 * 2 + 128 + 40), s);
    if (skip16 == 3) hipLaunchKernelGGL(embed_env_bwd_kernel, dim3(nblk), dim3(256), 0, s, obs, xcat, dxcat, scratch, nr, spb);
    else hipLaunchKernelGGL(embed_scatter_bwd_kernel, dim3(nblk), dim3(256), 0, s, obs, xcat, dxcat, dtu, q, ldq, amax, demb,
                            scratch, nr, nrp, spb, skip16);
    // (env only: the partials' rows 0 .. 3 are all there is)
    hipLaunchKernelGGL(embed_scatter_reduce_kernel, dim3(skip16 == 3 ? 2 : 5, 32), dim3(256), 0, s, scratch, nblk, dWenv, dbenv, db2);
    return launch_check("embed_scatter_bwd");
}

int unit_basic_bwd(const float* obs, const float* dbasic, float* dW1, float* db1, float* scratch, long long nr,
                   hipStream_t s) {
    const long long total = nr * 40;
    int rpb = (int)((total + 2047) / 2048);
    if (rpb < 64) rpb = 64;
    const int nblk = (int)((total + rpb - 1) / rpb);     // <= 2048 -> <= 13.6 MB of scratch
    hipLaunchKernelGGL(unit_basic_bwd_kernel, dim3(nblk), dim3(256), 0, s, obs, dbasic, scratch, nr, rpb);
    hipLaunchKernelGGL(unit_basic_reduce_kernel, dim3(7, 32), dim3(256), 0, s, scratch, nblk, dW1, db1);
    return launch_check("unit_basic_bwd");
}

// The tail of the fused embedding backward in one launch: dW1/db1 += the dense kernel's partials [na][13][128] and the
// sparse kernel's [nb][13][128]; db2 of the two 16-unit types += the sparse kernel's [2][n2][128] bias partials.
__global__ __launch_bounds__(256) void embed_tail_reduce_kernel(const float* __restrict__ pa, int na, const float* __restrict__ pb,
                                                                int nb, float* __restrict__ dW1, float* __restrict__ db1,
                                                                const float* __restrict__ p2, int n2, float* __restrict__ db2,
                                                                const float* __restrict__ p3, float* __restrict__ db2_small) {
    if (blockIdx.x >= 8) {                              // bias gradients of the small types (embed_small.hip's column sums): thread = (type of the pair, channel)
        const int pair = blockIdx.x - 8, h = threadIdx.x >> 7, c = threadIdx.x & 127;
        const int t = pair == 0 ? h : 4 + h;
        const int lo = t == 0 ? 0 : (t == 1 ? 32 : (t == 4 ? 192 : 224)), hi = t == 0 ? 32 : (t == 1 ? 192 : (t == 4 ? 224 : 256));
        atomicAdd(&db2_small[t * 128 + c], strided_sum(p3, 128, c, lo + blockIdx.y, gridDim.y, hi));
        return;
    }
    if (blockIdx.x == 7) {                              // bias gradients of types 2, 3: thread = (type, channel)
        const int t = threadIdx.x >> 7, c = threadIdx.x & 127;
        atomicAdd(&db2[t * 128 + c], strided_sum(p2 + (size_t)t * n2 * 128, 128, c, blockIdx.y, gridDim.y, n2));
        return;
    }
    const int idx = blockIdx.x * 256 + threadIdx.x;   // 0..1663
    if (idx >= 1664) return;
    const float acc = strided_sum(pa, 1664, idx, blockIdx.y, gridDim.y, na) + strided_sum(pb, 1664, idx, blockIdx.y, gridDim.y, nb);
    const int f = idx >> 7, c = idx & 127;
    if (f < 12) atomicAdd(&dW1[c * 12 + f], acc);
    else atomicAdd(&db1[c], acc);
}

int embed_tail_reduce(const float* pa, int na, const float* pb, int nb, float* dW1, float* db1, const float* p2, int n2,
                      float* db2, hipStream_t s, const float* p3, float* db2_small) {
    hipLaunchKernelGGL(embed_tail_reduce_kernel, dim3(p3 != nullptr ? 10 : 8, 32), dim3(256), 0, s, pa, na, pb, nb, dW1, db1, p2, n2, db2, p3, db2_small);
    return launch_check("embed_tail_reduce");
}

int unit_basic_reduce(const float* partials, int nblk, float* dW1, float* db1, hipStream_t s) {
    hipLaunchKernelGGL(unit_basic_reduce_kernel, dim3(7, 32), dim3(256), 0, s, partials, nblk, dW1, db1);
    return launch_check("unit_basic_reduce");
}

int colsum(const float* X, int ld, long long rows, int cols, float* out, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return 0;
    int rpb = (int)((rows + 1023) / 1024);
    if (rpb < 64) rpb = 64;
    const unsigned gy = (unsigned)((rows + rpb - 1) / rpb);
    if (cols <= 64) hipLaunchKernelGGL(colsum_kernel<64>, dim3((cols + 63) / 64, gy), dim3(256), 0, s, X, ld, rows, cols, out, rpb);
    else if (cols <= 128) hipLaunchKernelGGL(colsum_kernel<128>, dim3((cols + 127) / 128, gy), dim3(256), 0, s, X, ld, rows, cols, out, rpb);
    else hipLaunchKernelGGL(colsum_kernel<256>, dim3((cols + 255) / 256, gy), dim3(256), 0, s, X, ld, rows, cols, out, rpb);
    return launch_check("colsum");
}

}  // namespace dc
