// Micro-benchmark: issue rate of v_mfma_f32_4x4x1_16b_f32 by operand register file and chain count.
// hipcc --offload-arch=gfx950 -O3 mfma4x4.hip -o mfma4x4 && ./mfma4x4
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MF(ACC, BREG) \
    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+" ACC(c[i & (NC - 1)]) : "v"(a), BREG(b[i & 7]));

template <int MODE, int NC>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
    f32x4 c[NC];
    float b[8];
    float a = threadIdx.x * 0.001f;
    for (int i = 0; i < NC; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) b[i] = 1.0f + i + threadIdx.x;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if constexpr (MODE == 0) { asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(c[i % NC]) : "v"(a), "a"(b[i & 7])); }
            if constexpr (MODE == 1) { asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+a"(c[i % NC]) : "v"(a), "v"(b[i & 7])); }
            if constexpr (MODE == 2) { asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(c[i % NC]) : "v"(a), "v"(b[i & 7])); }
            if constexpr (MODE == 3) { asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+a"(c[i % NC]) : "v"(a), "a"(b[i & 7])); }
            if constexpr (MODE == 4) { asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c[i % NC]) : "v"(a), "v"(b[i & 7])); }
            if constexpr (MODE == 5) { asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[i % NC]) : "v"(a), "a"(b[i & 7])); }
        }
    }
    asm volatile("s_nop 7\n s_nop 7");
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int NC>
void run(const char* name, float* out, long long* cyc, int threads) {
    const int iters = 200;
    hipLaunchKernelGGL((k<MODE, NC>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<MODE, NC>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s threads %3d chains %2d : %6.2f cycles/MFMA\n", name, threads, NC, (double)h / (iters * 32.0));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    for (int th : {64, 256}) {
        if (th == 64) {
            run<0, 4>("4x4x1  acc VGPR, B AGPR", out, cyc, 64); run<0, 8>("4x4x1  acc VGPR, B AGPR", out, cyc, 64);
            run<1, 4>("4x4x1  acc AGPR, B VGPR", out, cyc, 64); run<1, 8>("4x4x1  acc AGPR, B VGPR", out, cyc, 64);
            run<2, 4>("4x4x1  acc VGPR, B VGPR", out, cyc, 64); run<2, 8>("4x4x1  acc VGPR, B VGPR", out, cyc, 64);
            run<3, 4>("4x4x1  acc AGPR, B AGPR", out, cyc, 64); run<3, 8>("4x4x1  acc AGPR, B AGPR", out, cyc, 64);
            run<2, 1>("4x4x1  acc VGPR, B VGPR", out, cyc, 64); run<2, 2>("4x4x1  acc VGPR, B VGPR", out, cyc, 64);
            run<4, 4>("16x16x4 acc AGPR, B VGPR", out, cyc, 64); run<5, 4>("16x16x4 acc VGPR, B AGPR", out, cyc, 64);
        } else {
            run<0, 4>("4x4x1  acc VGPR, B AGPR", out, cyc, 256); run<1, 4>("4x4x1  acc AGPR, B VGPR", out, cyc, 256);
            run<2, 4>("4x4x1  acc VGPR, B VGPR", out, cyc, 256); run<4, 4>("16x16x4 acc AGPR, B VGPR", out, cyc, 256);
        }
    }
    return 0;
}
