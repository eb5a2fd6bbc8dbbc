// Micro-benchmark: issue cost of v_pk_fma_f32 (op_sel broadcast, as in rnn_persist_valu.hip), v_fma_f32,
// v_add_f32_dpp and v_exp_f32 at one and two waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 pkfma.hip -o pkfma && ./pkfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int MODE, int NC>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    f32x2 c[NC];
    f32x2 w[8];
    f32x2 h = {threadIdx.x * 0.001f, 0.5f};
    for (int i = 0; i < NC; ++i) c[i] = f32x2{0.f, 0.f};
    for (int i = 0; i < 8; ++i) w[i] = f32x2{1.0f + i + threadIdx.x, 0.25f * i};
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if constexpr (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(c[i % NC]) : "v"(w[i & 7]), "v"(h));
            if constexpr (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c[i % NC].x) : "v"(w[i & 7].x), "v"(h.x));
            if constexpr (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c[i % NC]) : "v"(w[i & 7]), "v"(h));
            if constexpr (MODE == 3) asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(c[i % NC].x) : "v"(w[i & 7].x));
            if constexpr (MODE == 4) asm volatile("v_exp_f32 %0, %1" : "=v"(c[i % NC].x) : "v"(w[i & 7].x));
            if constexpr (MODE == 5) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(c[i % NC]) : "v"(w[i & 7]), "v"(h));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NC; ++i) s += c[i].x + c[i].y;
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
    printf("%-34s threads %3d chains %2d : %6.2f memtime ticks/instr/wave\n", name, threads, NC, (double)h / (iters * 64.0));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    for (int th : {64, 256, 512}) {
        run<0, 8>("v_pk_fma_f32 op_sel bcast", out, cyc, th);
        run<2, 8>("v_pk_fma_f32 plain", out, cyc, th);
        run<0, 2>("v_pk_fma_f32 op_sel bcast", out, cyc, th);
        run<1, 8>("v_fma_f32", out, cyc, th);
        run<5, 8>("v_pk_mul_f32", out, cyc, th);
        run<3, 8>("v_add_f32_dpp row_ror", out, cyc, th);
        run<3, 1>("v_add_f32_dpp row_ror (dependent)", out, cyc, th);
        run<4, 8>("v_exp_f32", out, cyc, th);
    }
    // s_memtime runs at a fixed 100 MHz on gfx950: also print the ratio to a known-cost instruction
    return 0;
}
