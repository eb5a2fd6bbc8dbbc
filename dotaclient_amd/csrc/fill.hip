// Zero-fill / small device-to-device copy as plain kernels of this library.
//
// Every buffer the hot path clears (the team kernels' exchange ring and ticket counters, the gradient bucket, the loss
// statistics, padding rows) is cleared by a kernel launched like every other kernel of the step, not by hipMemsetAsync /
// hipMemset2DAsync / hipMemcpyAsync: an epoch captured into a hipGraph then consists of kernel nodes only, whose order is the
// capture order.  (Round 3: with the runtime's memset nodes in the captured epoch, replays of the H = 256 team kernels
// faulted intermittently - profiles/r03/crash_bisect.md.)
#include "kernels.h"

namespace dc {
namespace {

__global__ void __launch_bounds__(256) zero_kernel(uint4* __restrict__ p16, size_t n16, unsigned char* __restrict__ tail, int ntail) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p16[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

__global__ void __launch_bounds__(256) zero2d_kernel(float* __restrict__ p, long long ld, int width, long long rows) {
    const long long total = rows * width;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) p[(i / width) * ld + (i % width)] = 0.f;
}

__global__ void __launch_bounds__(256) copy_f32_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

inline int blocks_for(size_t items) {
    size_t b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

int zero_async(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return 0;
#if DC_HIP_MEMSET
    if (hipMemsetAsync(p, 0, bytes, s) != hipSuccess) return launch_check("zero_async (hipMemsetAsync)");
    return 0;
#else
    unsigned char* b = static_cast<unsigned char*>(p);
    // leading bytes up to 16-byte alignment are handled as a "tail" of their own launch (never happens for the library's buffers)
    const size_t mis = (16 - (reinterpret_cast<uintptr_t>(b) & 15)) & 15;
    if (mis) {
        const int lead = (int)(mis < bytes ? mis : bytes);
        hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(256), 0, s, nullptr, (size_t)0, b, lead);
        b += lead; bytes -= lead;
        if (bytes == 0) return launch_check("zero_async");
    }
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(zero_kernel, dim3(blocks_for(n16)), dim3(256), 0, s, reinterpret_cast<uint4*>(b), n16, b + n16 * 16, (int)(bytes - n16 * 16));
    return launch_check("zero_async");
#endif
}

int zero2d_f32_async(float* p, long long ld, int width, long long rows, hipStream_t s) {
    if (width <= 0 || rows <= 0) return 0;
    if (ld == width) return zero_async(p, (size_t)rows * width * sizeof(float), s);
#if DC_HIP_MEMSET
    if (hipMemset2DAsync(p, (size_t)ld * sizeof(float), 0, (size_t)width * sizeof(float), (size_t)rows, s) != hipSuccess)
        return launch_check("zero2d_f32_async (hipMemset2DAsync)");
    return 0;
#else
    hipLaunchKernelGGL(zero2d_kernel, dim3(blocks_for((size_t)rows * width)), dim3(256), 0, s, p, ld, width, rows);
    return launch_check("zero2d_f32_async");
#endif
}

int copy_f32_async(float* dst, const float* src, long long n, hipStream_t s) {
    if (n <= 0) return 0;
#if DC_HIP_MEMSET
    if (hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return launch_check("copy_f32_async (hipMemcpyAsync)");
    return 0;
#else
    hipLaunchKernelGGL(copy_f32_kernel, dim3(blocks_for((size_t)n)), dim3(256), 0, s, dst, src, n);
    return launch_check("copy_f32_async");
#endif
}

}  // namespace dc
