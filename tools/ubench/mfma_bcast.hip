// Semantics probe: v_mfma_f32_4x4x1_16b_f32 with cbsz:4 abid:b - is block b's A broadcast to all 16 blocks?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int CBSZ, int ABID>
__global__ void k(float* out) {
    const int l = threadIdx.x;
    float a = (float)l, b = 1.0f;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, CBSZ, ABID, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
template <int CBSZ, int ABID> int check(float* d) {
    hipLaunchKernelGGL((k<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, d);
    float h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    // expected: blocks are grouped by 2^CBSZ; every block takes A from block (group base + ABID)
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        const int blk = l / 4, grp = blk >> CBSZ << CBSZ;
        if (h[l * 4 + i] != (float)(4 * (grp + ABID) + i)) ++bad;
    }
    printf("cbsz %d abid %2d: lane0 regs = %g %g %g %g ; lane 37 regs = %g %g %g %g ; mismatches vs (4*(group+abid)+i) = %d\n", CBSZ, ABID, h[0], h[1], h[2], h[3],
           h[37 * 4], h[37 * 4 + 1], h[37 * 4 + 2], h[37 * 4 + 3], bad);
    return bad;
}
int main() { float* d; hipMalloc(&d, 4096); check<4, 0>(d); check<4, 5>(d); check<4, 15>(d); check<3, 0>(d); check<3, 5>(d); check<3, 7>(d); check<2, 3>(d); return 0; }
