// Feasibility probe: fp32-grade GEMM on the bf16 matrix cores by EXACT 3-way splitting.
//   x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): three 8-bit-mantissa pieces carry
//   the whole 24-bit f32 mantissa, every piece product is exact in f32, and
//   a*b ~= hh + (hm + mh) + (hl + lh + mm)   (the three dropped terms are <= 2^-32 |ab|)
//   accumulated in f32 by v_mfma_f32_32x32x16_bf16: 6 bf16 MFMAs per K16 instead of 8 f32 MFMAs (32x32x2) = 6*32 vs
//   8*64 matrix-pipe cycles: 2.67x the f32-MFMA peak (419 TF) at f32 accuracy.
// C[M][N] = A[M][K] * B[N][K]^T (+ bias).  B (the weights) is split once by a tiny pre-pass; A on the fly.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/gemm_x3.hip -o tools/ubench/gemm_x3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// two f32 -> three packed bf16 pairs (exact: x = h + m + l)
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16(s0, s1);
}

// ---- f16 variant (round 4 probe): x * 2^s = h + m (+ l) with h = f16(x 2^s), m = f16(x 2^s - h): two 11-bit pieces carry 22-23 bits of the
// f32 mantissa; a*b ~= hh + hm + mh (+ mm): 3 or 4 MFMAs per K16 instead of 6.  f16 has 5 exponent bits: operands are pre-scaled by a
// power of two (exact) so that the m pieces stay normal, and the product is scaled back in the epilogue.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
__device__ __forceinline__ uint32_t pk_f16(float a, float b) {
    f16x2 v; v.x = (_Float16)a; v.y = (_Float16)b;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void split2h(float x0, float x1, uint32_t& h, uint32_t& m) {
    h = pk_f16(x0, x1);
    const f16x2 hv = __builtin_bit_cast(f16x2, h);
    m = pk_f16(x0 - (float)hv.x, x1 - (float)hv.y);
}
__global__ void split_planes_f16_kernel(const float* __restrict__ B, uint16_t* __restrict__ P, long long n, float scale) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i >= n) return;
    uint32_t h, m;
    split2h(B[i] * scale, B[i + 1] * scale, h, m);
    *reinterpret_cast<uint32_t*>(P + i) = h;
    *reinterpret_cast<uint32_t*>(P + n + i) = m;
}

// pre-pass: B [N][K] f32 -> planes [3][N][K] bf16
__global__ void split_planes_kernel(const float* __restrict__ B, uint16_t* __restrict__ P, long long n) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i >= n) return;
    uint32_t h, m, l;
    split2(B[i], B[i + 1], h, m, l);
    *reinterpret_cast<uint32_t*>(P + i) = h;
    *reinterpret_cast<uint32_t*>(P + n + i) = m;
    *reinterpret_cast<uint32_t*>(P + 2 * n + i) = l;
}

enum { BM = 128, BN = 128, BK = 32, A_LD = BK + 4 /* floats */, B_LD = BK + 8 /* bf16 */ };
enum { A_STAGE = BM * A_LD * 4, B_PLANE = BN * B_LD * 2, STAGE = A_STAGE + 3 * B_PLANE };

template <int NPROD, bool F16 = false>   // 6 = f32-grade, 3 = hh + hm + mh (16-bit mantissa), 1 = plain bf16; F16: f16 pieces, NPROD 3 or 4
__global__ __launch_bounds__(256) void gemm_x3_kernel(const float* __restrict__ A, const uint16_t* __restrict__ Bp, float* __restrict__ C,
                                                      const float* __restrict__ bias, int M, int N, int K, float sa = 1.f, float inv = 1.f) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                 // 2 x 2 waves, 64 x 64 each
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const long long plane = (long long)N * K;

    // global -> register staging: A 128 x 32 f32 = 1024 float4 (4 per thread); B 3 x 128 x 32 bf16 = 1536 x 16 B (6 per thread)
    float4 ra[4];
    u32x4 rb[6];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, r = idx >> 3, c = (idx & 7) * 4;
            ra[i] = *reinterpret_cast<const float4*>(A + (long long)(m0 + r) * K + k0 + c);
        }
#pragma unroll
        for (int i = 0; i < (F16 ? 4 : 6); ++i) {
            const int idx = tid + 256 * i, p = idx >> 9, r = (idx >> 2) & 127, c = (idx & 3) * 8;
            rb[i] = *reinterpret_cast<const u32x4*>(Bp + p * plane + (long long)(n0 + r) * K + k0 + c);
        }
    };
    auto lstore = [&](int st) {
        char* base = lds + st * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, r = idx >> 3, c = (idx & 7) * 4;
            *reinterpret_cast<float4*>(base + (r * A_LD + c) * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < (F16 ? 4 : 6); ++i) {
            const int idx = tid + 256 * i, p = idx >> 9, r = (idx >> 2) & 127, c = (idx & 3) * 8;
            *reinterpret_cast<u32x4*>(base + A_STAGE + p * B_PLANE + (r * B_LD + c) * 2) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(0);
    lstore(0);
    __syncthreads();
    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        const char* base = lds + st * STAGE;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int ko = kk * 16 + 8 * (lane >> 5);
            bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm * 64 + i * 32 + (lane & 31);
                const float4 x0 = *reinterpret_cast<const float4*>(base + (r * A_LD + ko) * 4);
                const float4 x1 = *reinterpret_cast<const float4*>(base + (r * A_LD + ko + 4) * 4);
                uint32_t h[4], m[4], l[4] = {0, 0, 0, 0};
                if (F16) {
                    split2h(x0.x * sa, x0.y * sa, h[0], m[0]);
                    split2h(x0.z * sa, x0.w * sa, h[1], m[1]);
                    split2h(x1.x * sa, x1.y * sa, h[2], m[2]);
                    split2h(x1.z * sa, x1.w * sa, h[3], m[3]);
                } else {
                split2(x0.x, x0.y, h[0], m[0], l[0]);
                split2(x0.z, x0.w, h[1], m[1], l[1]);
                split2(x1.x, x1.y, h[2], m[2], l[2]);
                split2(x1.z, x1.w, h[3], m[3], l[3]);
                }
                ah[i] = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
                am[i] = __builtin_bit_cast(bf16x8, u32x4{m[0], m[1], m[2], m[3]});
                al[i] = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wn * 64 + j * 32 + (lane & 31);
                const char* bp = base + A_STAGE + (r * B_LD + ko) * 2;
                bh[j] = *reinterpret_cast<const bf16x8*>(bp);
                if (NPROD > 1) bm[j] = *reinterpret_cast<const bf16x8*>(bp + B_PLANE);
                if (NPROD > 3 && !F16) bl[j] = *reinterpret_cast<const bf16x8*>(bp + 2 * B_PLANE);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc[i][j];
                    if (F16) {
                        auto H = [](bf16x8 v) { return __builtin_bit_cast(f16x8, v); };
                        if (NPROD > 3) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(H(am[i]), H(bm[j]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(H(am[i]), H(bh[j]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(H(ah[i]), H(bm[j]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(H(ah[i]), H(bh[j]), c, 0, 0, 0);
                        acc[i][j] = c;
                        continue;
                    }
                    if (NPROD > 3) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[j], c, 0, 0, 0);
                    }
                    if (NPROD > 1) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
        if (kt + 1 < nk) lstore(st ^ 1);
        __syncthreads();
    }
    // epilogue: C layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                C[(long long)row * N + col] = acc[i][j][r] * inv + bv;
            }
        }
}

template <int NPROD, bool F16 = false>
static float run(const float* A, const uint16_t* Bp, float* C, const float* bias, int M, int N, int K, int iters, float sa = 1.f, float inv = 1.f) {
    const size_t lds = 2 * STAGE;
    hipFuncSetAttribute((const void*)gemm_x3_kernel<NPROD, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(N / BN, M / BM);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_x3_kernel<NPROD, F16>), grid, dim3(256), lds, 0, A, Bp, C, bias, M, N, K, sa, inv);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_x3_kernel<NPROD, F16>), grid, dim3(256), lds, 0, A, Bp, C, bias, M, N, K, sa, inv);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main(int argc, char** argv) {
    struct Shape { int M, N, K; const char* what; };
    const Shape shapes[] = {{1024, 256, 256, "accuracy"}, {65536, 1024, 256, "gates 256x256 LSTM-256"}, {65536, 256, 896, "pre_rnn"},
                            {65536, 896, 256, "dxcat"}, {16384, 512, 256, "gates 64x256 LSTM-128"}};
    for (const Shape& s : shapes) {
        const int M = s.M, N = s.N, K = s.K;
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hbias(N);
        uint32_t st = 12345u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f * 2.f - 1.f; };
        for (auto& x : hA) x = rnd() * 1.7f;
        for (auto& x : hB) x = rnd() * 0.06f;
        for (auto& x : hbias) x = rnd();
        float *dA, *dB, *dC, *dbias;
        uint16_t* dP;
        hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, (size_t)M * N * 4); hipMalloc(&dbias, N * 4);
        hipMalloc(&dP, hB.size() * 2 * 3);
        hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice);
        const long long nb = (long long)N * K;
        hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((nb / 2 + 255) / 256)), dim3(256), 0, 0, dB, dP, nb);
        const double flops = 2.0 * M * N * (double)K;
        const float t6 = run<6>(dA, dP, dC, dbias, M, N, K, 10);
        std::vector<float> hC((size_t)M * N);
        hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
        const float t3 = run<3>(dA, dP, dC, dbias, M, N, K, 10);
        std::vector<float> hC3((size_t)M * N);
        hipMemcpy(hC3.data(), dC, hC3.size() * 4, hipMemcpyDeviceToHost);
        const float t1 = run<1>(dA, dP, dC, dbias, M, N, K, 10);
        // f16 pieces: A scaled by 2^6 (values up to +-1.7), B by 2^12 (+-0.06): exact powers of two, undone in the epilogue
        const float SA = 64.f, SB = 4096.f;
        hipLaunchKernelGGL(split_planes_f16_kernel, dim3((unsigned)((nb / 2 + 255) / 256)), dim3(256), 0, 0, dB, dP, nb, SB);
        const float th4 = run<4, true>(dA, dP, dC, dbias, M, N, K, 10, SA, 1.f / (SA * SB));
        std::vector<float> hC4((size_t)M * N);
        hipMemcpy(hC4.data(), dC, hC4.size() * 4, hipMemcpyDeviceToHost);
        const float th3 = run<3, true>(dA, dP, dC, dbias, M, N, K, 10, SA, 1.f / (SA * SB));
        std::vector<float> hCh3((size_t)M * N);
        hipMemcpy(hCh3.data(), dC, hCh3.size() * 4, hipMemcpyDeviceToHost);
        // the same without pre-scaling (m pieces of small operands go subnormal)
        hipLaunchKernelGGL(split_planes_f16_kernel, dim3((unsigned)((nb / 2 + 255) / 256)), dim3(256), 0, 0, dB, dP, nb, 1.f);
        run<4, true>(dA, dP, dC, dbias, M, N, K, 1, 1.f, 1.f);
        std::vector<float> hC4u((size_t)M * N);
        hipMemcpy(hC4u.data(), dC, hC4u.size() * 4, hipMemcpyDeviceToHost);
        double eh4 = 0, eh3 = 0, eh4u = 0;
        // accuracy on a sample of rows against an f64 reference and against a plain f32 fma chain
        double e6 = 0, e3 = 0, ef = 0, ref_max = 0;
        for (int r = 0; r < M; r += (M > 2048 ? 997 : 7))
            for (int c = 0; c < N; ++c) {
                double acc = hbias[c];
                float f = 0.f;
                for (int k = 0; k < K; ++k) { acc += (double)hA[(size_t)r * K + k] * hB[(size_t)c * K + k]; f = fmaf(hA[(size_t)r * K + k], hB[(size_t)c * K + k], f); }
                f += hbias[c];
                e6 = fmax(e6, fabs(hC[(size_t)r * N + c] - acc)); e3 = fmax(e3, fabs(hC3[(size_t)r * N + c] - acc)); ef = fmax(ef, fabs(f - acc));
                eh4 = fmax(eh4, fabs(hC4[(size_t)r * N + c] - acc)); eh3 = fmax(eh3, fabs(hCh3[(size_t)r * N + c] - acc));
                eh4u = fmax(eh4u, fabs(hC4u[(size_t)r * N + c] - acc));
                ref_max = fmax(ref_max, fabs(acc));
            }
        printf("%-26s M=%6d N=%5d K=%4d | x3(6 prod) %8.1f us %6.1f TF err %.2e | x2(3 prod) %8.1f us %6.1f TF err %.2e | bf16 %8.1f us %6.1f TF | f32 fma-chain err %.2e (max|C| %.2f)\n",
               s.what, M, N, K, t6 * 1e3, flops / t6 / 1e9, e6 / ref_max, t3 * 1e3, flops / t3 / 1e9, e3 / ref_max, t1 * 1e3, flops / t1 / 1e9, ef / ref_max, ref_max);
        printf("%-26s   f16 pieces, pre-scaled: 4 prod %8.1f us %6.1f TF err %.2e | 3 prod %8.1f us %6.1f TF err %.2e | 4 prod unscaled err %.2e\n",
               "", th4 * 1e3, flops / th4 / 1e9, eh4 / ref_max, th3 * 1e3, flops / th3 / 1e9, eh3 / ref_max, eh4u / ref_max);
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dbias); hipFree(dP);
    }
    return 0;
}
