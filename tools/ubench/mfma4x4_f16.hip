// Micro-benchmark (round 6): issue rate of v_mfma_f32_4x4x4_16b_f16 against v_mfma_f32_4x4x1_16b_f32 - would W_hh as two f16 planes
// (three instructions per K = 4: hh, hm, mh) beat the four f32 instructions per K = 4 of the H = 256 team kernels' product phase?
// Also the 32x32x16 f16 MFMA back to back with the accumulator in VGPRs (what gemm_x3s.hip's MATRIX phase issues): cycles per instruction.
// hipcc --offload-arch=gfx950 -O3 mfma4x4_f16.hip -o mfma4x4_f16 && ./mfma4x4_f16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

template <int MODE, int NC>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
    f32x4 c[NC];
    f32x16 C[4];
    float b[8];
    f16x4 bh[8];
    f16x8 b8[4];
    const float a = threadIdx.x * 0.001f;
    const f16x4 ah = {(_Float16)a, (_Float16)(a + 1), (_Float16)(a + 2), (_Float16)(a + 3)};
    const f16x4 ah2 = {(_Float16)(a + 2), (_Float16)(a + 1), (_Float16)(a + 2), (_Float16)(a + 3)};
    const f16x8 a8 = {(_Float16)a, (_Float16)(a + 1), (_Float16)(a + 2), (_Float16)(a + 3), (_Float16)a, (_Float16)a, (_Float16)a, (_Float16)a};
    for (int i = 0; i < NC; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) C[i][r] = 0.f;
    for (int i = 0; i < 8; ++i) { b[i] = 1.0f + i + threadIdx.x; bh[i] = f16x4{(_Float16)b[i], (_Float16)1, (_Float16)2, (_Float16)3}; }
    for (int i = 0; i < 4; ++i) b8[i] = f16x8{(_Float16)b[i], (_Float16)1, (_Float16)2, (_Float16)3, (_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1};
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if constexpr (MODE == 0) { asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(c[i % NC]) : "v"(a), "v"(b[i & 7])); }
            if constexpr (MODE == 1) { asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+v"(c[i % NC]) : "v"(ah), "v"(bh[i & 7])); }
            if constexpr (MODE == 2) { asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0 cbsz:4 abid:3" : "+v"(c[i % NC]) : "v"(ah), "v"(bh[i & 7])); }
            if constexpr (MODE == 4) { asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:3" : "+v"(c[i % NC]) : "v"(a), "a"(b[i & 7])); }
            if constexpr (MODE == 5) { asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0 cbsz:4 abid:3" : "+v"(c[i % NC]) : "v"(ah), "a"(bh[i & 7])); }
            if constexpr (MODE == 6 && NC >= 3) { asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %3, %4, %0 cbsz:4 abid:3\n\tv_mfma_f32_4x4x4_16b_f16 %1, %3, %5, %1 cbsz:4 abid:3\n\tv_mfma_f32_4x4x4_16b_f16 %2, %6, %4, %2 cbsz:4 abid:3" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]) : "v"(ah), "a"(bh[i & 7]), "a"(bh[(i + 1) & 7]), "v"(ah2)); }
            if constexpr (MODE == 3) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(C[i & 3]) : "v"(a8), "v"(b8[i & 3])); }
        }
    }
    asm volatile("s_nop 7\n s_nop 7");
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    for (int i = 0; i < 4; ++i) s += C[i][0] + C[i][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int NC>
void run(const char* name, float* out, long long* cyc, int threads, int blocks) {
    const int iters = 200;
    hipLaunchKernelGGL((k<MODE, NC>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<MODE, NC>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-52s threads %3d blocks %4d chains %2d : %6.2f cycles/MFMA (workgroup 0, wave 0)\n", name, threads, blocks, NC, (double)h / (iters * 32.0));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 64 << 20); hipMalloc(&cyc, 64);
    for (int blocks : {1, 256, 512}) {
        run<0, 4>("4x4x1_16b_f32", out, cyc, 256, blocks);
        run<1, 4>("4x4x4_16b_f16", out, cyc, 256, blocks);
        run<2, 4>("4x4x4_16b_f16 cbsz:4 abid:3 (block broadcast)", out, cyc, 256, blocks);
        run<3, 4>("32x32x16_f16, C in VGPRs, 4 chains", out, cyc, 256, blocks);
    }
    run<3, 4>("32x32x16_f16, 512 threads (2 waves per SIMD)", out, cyc, 512, 256);
    run<4, 4>("4x4x1_16b_f32 cbsz:4, B in AGPR", out, cyc, 256, 256);
    run<5, 4>("4x4x4_16b_f16 cbsz:4, B in AGPR", out, cyc, 256, 256);
    run<6, 4>("4x4x4_16b_f16 cbsz:4, B in AGPR, the kernel's triple (per 3 MFMAs)", out, cyc, 256, 256);
    // dependent-issue distance: how many independent chains does the 4x4 instruction need?
    run<2, 3>("4x4x4_16b_f16 cbsz:4, 3 chains", out, cyc, 256, 256);
    run<2, 2>("4x4x4_16b_f16 cbsz:4, 2 chains", out, cyc, 256, 256);
    run<2, 1>("4x4x4_16b_f16 cbsz:4, 1 chain", out, cyc, 256, 256);
    run<0, 2>("4x4x1_16b_f32, 2 chains", out, cyc, 256, 256);
    run<0, 1>("4x4x1_16b_f32, 1 chain", out, cyc, 256, 256);
    return 0;
}
