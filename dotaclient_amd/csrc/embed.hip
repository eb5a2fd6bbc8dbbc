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

// grid-stride over type-major rows; 256 threads = 2 rows x 128 channels per pass
__global__ __launch_bounds__(256) void unit_basic_fwd_kernel(const float* __restrict__ obs, const float* __restrict__ W1,
                                                             const float* __restrict__ b1, float* __restrict__ basic,
                                                             long long nr) {
    const int c = threadIdx.x & 127;
    const int sub = threadIdx.x >> 7;
    float w[12];
#pragma unroll
    for (int f = 0; f < 12; ++f) w[f] = W1[c * 12 + f];
    const float b = b1[c];
    const long long total = nr * 40;
    for (long long row = (long long)blockIdx.x * 2 + sub; row < total; row += (long long)gridDim.x * 2) {
        long long local;
        const int t = unit_type_of_row(row, nr, &local);
        const int U = c_type_units[t];
        const long long n = local / U;
        const int ul = (int)(local - n * U);
        const float* x = obs + n * OBS_DIM + 3 + (c_type_cum[t] + ul) * 12;
        float acc = b;
#pragma unroll
        for (int f = 0; f < 12; ++f) acc = fmaf(x[f], w[f], acc);
        basic[row * EMB + c] = fmaxf(acc, 0.f);
    }
}

// one env-step per 128 threads
__global__ __launch_bounds__(256) void pool_env_fwd_kernel(const float* __restrict__ obs, const float* __restrict__ emb,
                                                           const float* __restrict__ Wenv, const float* __restrict__ benv,
                                                           float* __restrict__ xcat, uint8_t* __restrict__ amax,
                                                           long long nr) {
    const int c = threadIdx.x & 127;
    const int sub = threadIdx.x >> 7;
    const float w0 = Wenv[c * 3 + 0], w1 = Wenv[c * 3 + 1], w2 = Wenv[c * 3 + 2], be = benv[c];
    for (long long n = (long long)blockIdx.x * 2 + sub; n < nr; n += (long long)gridDim.x * 2) {
        const float* e = obs + n * OBS_DIM;
        float* xo = xcat + n * XCAT;
        xo[c] = fmaxf(fmaf(e[2], w2, fmaf(e[1], w1, fmaf(e[0], w0, be))), 0.f);  // policy.py:97
        float enh_max = 0.f;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int U = c_type_units[t];
            const float* p = emb + (nr * c_type_cum[t] + n * U) * EMB + c;
            float m = p[0];
            int am = 0;
            for (int u = 1; u < U; ++u) {
                const float v = p[(long long)u * EMB];
                if (v > m) { m = v; am = u; }   // first maximum wins, like torch.max
            }
            if (t == 3) enh_max = m;
            if (t >= 1 && t <= 3) amax[(n * 3 + (t - 1)) * EMB + c] = (uint8_t)am;
            if (t < 5) xo[(1 + t) * EMB + c] = m;
        }
        xo[6 * EMB + c] = enh_max;  // policy.py:127: eth_embedding_max is computed from enh_embedding
    }
}

// demb[type-major row][c] = dtu[n][u] * q[n][c]  +  pool routing of dxcat[n][.]
// also accumulates dW_env[128][3], db_env[128] (relu-masked by the stored env embedding)
__global__ __launch_bounds__(256) void embed_scatter_bwd_kernel(
    const float* __restrict__ obs, const float* __restrict__ xcat, const float* __restrict__ dxcat,
    const float* __restrict__ dtu, const float* __restrict__ q, int ldq, const uint8_t* __restrict__ amax,
    float* __restrict__ demb, float* __restrict__ dWenv, float* __restrict__ dbenv, long long nr) {
    const int c = threadIdx.x & 127;
    const int sub = threadIdx.x >> 7;
    float gw0 = 0.f, gw1 = 0.f, gw2 = 0.f, gb = 0.f;
    for (long long n = (long long)blockIdx.x * 2 + sub; n < nr; n += (long long)gridDim.x * 2) {
        const float* dx = dxcat + n * XCAT;
        const float qc = q[n * ldq + c];
        const float* dt = dtu + n * 40;
        // env embedding backward (policy.py:97)
        const float de = (xcat[n * XCAT + c] > 0.f) ? dx[c] : 0.f;
        const float* e = obs + n * OBS_DIM;
        gw0 = fmaf(de, e[0], gw0); gw1 = fmaf(de, e[1], gw1); gw2 = fmaf(de, e[2], gw2); gb += de;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int U = c_type_units[t];
            float pool_g;  // gradient arriving at this type's pooled slot(s)
            if (t == 3) pool_g = dx[4 * EMB + c] + dx[6 * EMB + c];   // enh feeds slots 4 and 6
            else if (t == 5) pool_g = 0.f;                             // eth is never pooled
            else pool_g = dx[(1 + t) * EMB + c];
            int am = 0;
            if (t >= 1 && t <= 3) am = amax[(n * 3 + (t - 1)) * EMB + c];
            float* p = demb + (nr * c_type_cum[t] + n * U) * EMB + c;
            for (int u = 0; u < U; ++u) {
                float g = dt[c_type_cum[t] + u] * qc;
                if (u == am) g += pool_g;
                p[(long long)u * EMB] = g;
            }
        }
    }
    atomicAdd(&dWenv[c * 3 + 0], gw0);
    atomicAdd(&dWenv[c * 3 + 1], gw1);
    atomicAdd(&dWenv[c * 3 + 2], gw2);
    atomicAdd(&dbenv[c], gb);
}

// dW1[c][f] += sum_rows dbasic[row][c] * x[row][f];  db1[c] += sum_rows dbasic[row][c]
// thread = (channel c, feature half h): 6 weight accumulators (+ bias on h == 0)
__global__ __launch_bounds__(256) void unit_basic_bwd_kernel(const float* __restrict__ obs,
                                                             const float* __restrict__ dbasic, float* __restrict__ dW1,
                                                             float* __restrict__ db1, long long nr, int rows_per_block) {
    const int c = threadIdx.x & 127;
    const int h = threadIdx.x >> 7;
    const long long total = nr * 40;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(total, r0 + rows_per_block);
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float accb = 0.f;
    for (long long row = r0; row < r1; ++row) {
        long long local;
        const int t = unit_type_of_row(row, nr, &local);
        const int U = c_type_units[t];
        const long long n = local / U;
        const int ul = (int)(local - n * U);
        const float* x = obs + n * OBS_DIM + 3 + (c_type_cum[t] + ul) * 12 + h * 6;
        const float g = dbasic[row * EMB + c];
#pragma unroll
        for (int f = 0; f < 6; ++f) acc[f] = fmaf(g, x[f], acc[f]);
        accb += g;
    }
#pragma unroll
    for (int f = 0; f < 6; ++f) atomicAdd(&dW1[c * 12 + h * 6 + f], acc[f]);
    if (h == 0) atomicAdd(&db1[c], accb);
}

// out[j] += sum_rows X[row][j]   (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int ld, long long rows, int cols,
                                                     float* __restrict__ out, int rows_per_block) {
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = min(rows, r0 + rows_per_block);
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= cols) return;
    float acc = 0.f;
    for (long long r = r0; r < r1; ++r) acc += X[r * ld + j];
    atomicAdd(&out[j], acc);
}

static inline int grid_for(long long items, int per_block, int cap) {
    long long g = (items + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

int unit_basic_fwd(const float* obs, const float* W1, const float* b1, float* basic, long long nr, hipStream_t s) {
    hipLaunchKernelGGL(unit_basic_fwd_kernel, dim3(grid_for(nr * 40, 2, 256 * 32)), dim3(256), 0, s, obs, W1, b1, basic, nr);
    return launch_check("unit_basic_fwd");
}

int pool_env_fwd(const float* obs, const float* emb, const float* Wenv, const float* benv, float* xcat, uint8_t* amax,
                 long long nr, hipStream_t s) {
    hipLaunchKernelGGL(pool_env_fwd_kernel, dim3(grid_for(nr, 2, 256 * 16)), dim3(256), 0, s, obs, emb, Wenv, benv, xcat,
                       amax, nr);
    return launch_check("pool_env_fwd");
}

int embed_scatter_bwd(const float* obs, const float* xcat, const float* dxcat, const float* dtu, const float* q, int ldq,
                      const uint8_t* amax, float* demb, float* dWenv, float* dbenv, long long nr, hipStream_t s) {
    hipLaunchKernelGGL(embed_scatter_bwd_kernel, dim3(grid_for(nr, 2, 256 * 8)), dim3(256), 0, s, obs, xcat, dxcat, dtu, q,
                       ldq, amax, demb, dWenv, dbenv, nr);
    return launch_check("embed_scatter_bwd");
}

int unit_basic_bwd(const float* obs, const float* dbasic, float* dW1, float* db1, long long nr, hipStream_t s) {
    const long long total = nr * 40;
    int rpb = (int)((total + 2047) / 2048);
    if (rpb < 64) rpb = 64;
    hipLaunchKernelGGL(unit_basic_bwd_kernel, dim3((unsigned)((total + rpb - 1) / rpb)), dim3(256), 0, s, obs, dbasic, dW1,
                       db1, nr, rpb);
    return launch_check("unit_basic_bwd");
}

int colsum(const float* X, int ld, long long rows, int cols, float* out, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return 0;
    int rpb = (int)((rows + 511) / 512);
    if (rpb < 32) rpb = 32;
    dim3 grid((cols + 255) / 256, (unsigned)((rows + rpb - 1) / rpb));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, X, ld, rows, cols, out, rpb);
    return launch_check("colsum");
}

}  // namespace dc
