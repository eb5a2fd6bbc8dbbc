// GEMM for the network's dense products on the bf16 matrix cores, f32-grade by exact 3-way splitting (PREC = 6), or
// plain bf16 operands with f32 accumulate (PREC = 1, the bf16 path of BASELINE.json configs[4]).
//
// Replaces the arithmetic torch's nn.Linear forward/backward does for /root/reference/policy.py:54-75,138-155
// (affine_pre_rnn, the recurrent cell's input projection, the head projections) and their autograd products.
//
// Why.  gfx950 has no reduced-precision fast path for f32 inputs, and its f32-input MFMA runs at 1/16 of the bf16 rate.
// An f32 value splits exactly into three bf16 pieces (x = h + m + l, gemm_tiles.h), so
//     a b = hh + (hm + mh) + (hl + lh + mm) + O(2^-32 |ab|):
// six v_mfma_f32_32x32x16_bf16 per K = 16 give the f32 product to f32 round-off at 6 x 32 matrix-pipe cycles instead of
// the 8 x 64 of the f32 MFMA.  The first version of this idea (mma_kstep in gemm_tiles.h, still used by the fused
// embedding kernels) split the fragments AFTER their LDS reads, i.e. every element twice per workgroup and inside the
// MFMA loop: the fwd / dX products went 118 -> 150 TF and stopped there, VALU-bound.  Here:
//   * SPLIT ON LOAD: an operand tile goes global -> registers -> split once -> LDS as three bf16 planes; the MFMA loop
//     only reads ready fragments (one ds_read_b128 per plane and 32 x 16 block);
//   * WEIGHTS ARRIVE PRE-SPLIT: a small pre-pass per epoch writes every weight matrix as bf16 planes, in the orientation
//     its consumer contracts over (W for the forward, W^T for dX), so the weight operand needs no VALU work at all;
//   * k-major operands (both sides of the weight-gradient products, contraction over the env-steps) are packed in
//     pairs along k while they are split and stored kpair-major, so the stores are 16-byte and the fragment reads
//     conflict-free ds_read_b32;
//   * the epilogue goes through LDS and stores 16 bytes per lane (a K = 256 product spends as long in a 4-byte-per-lane
//     epilogue as in its main loop).
// Tile 128 x 128 x 16, 256 threads (2 x 2 waves of 64 x 64), two stages of 36 KB: two workgroups per CU.
//
// PREC = 4 (round 4, DC_DIMS_F16X2): TWO f16 pieces instead of three bf16 ones.  x 2^s = h + m + l with h = f16(x 2^s), m = f16(x 2^s - h):
// 11 + 11 significand bits plus the sign of m carry 23 of the 24 bits of an f32, |l| <= 2^-23 |x|, and
//     a b = hh + hm + mh + (mm + hl + lh + ...),   dropped: <= 2^-21 |ab| (mm <= 2^-22, the l terms <= 2^-22: gemm_tiles.h, DC_X2H_MM)
// is THREE v_mfma_f32_32x32x16_f16 per K = 16 instead of six bf16 ones (measured against f64 on the network's shapes,
// tools/ubench/gemm_x3.hip: 1.2e-7 .. 4.9e-7 of max |C| - the six-bf16 form 1.4e-7 .. 5.5e-7, the f32 fma chain 2.1e-7 .. 3.7e-7), two
// planes through the LDS instead of three, 3.5 instead of 5.5 VALU instructions per element split.  These products run at the
// chip's power limit (profiles/r04/gemm_power_evidence.json), so fewer MFMAs and fewer LDS bytes per flop is what buys time.
// f16 has five exponent bits: every operand is pre-scaled by a power of two (exact; `sa`, `sb`: activations 2^4, weights 2^8,
// gradients 2^(ceil(log2 rows) + 2)) so that the m pieces stay normal, and the accumulators are scaled back in the epilogue.
// An operand entry beyond 65504 / scale becomes inf -> NaN downstream -> the optimizer's own NaN guard (status word) stops the
// update: the host then re-runs with the bf16 pieces (Engine: DC_DIMS_F16X2 cleared), whose exponent range is f32's.
#include <cstdio>
#include "kernels.h"
#include "gemm_tiles.h"

namespace dc {
namespace {

enum { XB = 128, XK = 16 };
enum { RM_ROW_BYTES = 48, RM_PLANE = XB * RM_ROW_BYTES /* 6144 */, KM_PLANE = 8 * XB * 4 /* 4096 */ };
enum { OPER_BYTES = 3 * RM_PLANE, STAGE_BYTES = 2 * OPER_BYTES, X3_LDS = 2 * STAGE_BYTES /* 73728 */ };
enum { EP_LD = 68 };   // floats per row of a wave's 64 x 64 epilogue image (4 x 64 x 68 x 4 = 69632 <= X3_LDS)
// PREC = 4 needs two planes per operand, not three: 48 KB of stage buffers per workgroup instead of 72 - room for a THIRD workgroup per CU
// (150 / 164 VGPRs: three waves per SIMD fit) if the epilogue goes through a half-size image (32 rows at a time, 34 KB).  A/B switch:
// -DDC_X3_OCC=2 keeps two workgroups per CU for every precision.
#ifndef DC_X3_OCC
#define DC_X3_OCC 3
#endif
template <int PREC> struct X3Cfg {
    static constexpr bool kThree = PREC != 6 && DC_X3_OCC == 3;      // two-plane (f16x2) and one-plane (bf16 mode) products
    static constexpr int kOcc = kThree ? 3 : 2;
    static constexpr int kOper = (kThree ? 2 : 3) * RM_PLANE, kStage = 2 * kOper, kLds = 2 * kStage;   // 49152 / 73728
    static constexpr int kImgRows = kThree ? 32 : 64;
    // PREC = 1 has ONE piece per operand: the second LDS plane of the two-plane budget holds the NEXT sixteen k instead, so a step
    // between two barriers is K = 32 - eight MFMAs per wave instead of four (x3_kh below)
    static constexpr int kKHmax = (PREC == 1 && kThree) ? 2 : 1;
};
// ... except with an f32 k-major operand: two staged steps of two sub-steps of f32 rows do not fit the 168 registers of three workgroups
// per CU, and at two per CU the K = 32 form was slower than K = 16 at three (configs[4]'s dW with f32 x: 912 -> 1 400 us)
template <int PREC, int A_MODE, int B_MODE>
constexpr int x3_kh() { return (A_MODE == X3_KMAJ || B_MODE == X3_KMAJ) ? 1 : X3Cfg<PREC>::kKHmax; }

struct X3Args {
    const void* A; const void* B; const void* B2;
    float* C; float* C2; const float* bias; const float* aux; float* slab; float* a_colsum;
    long long a_plane, b_plane, b2_plane;   // plane strides (elements) of pre-split operands
    int M, N, K, lda, ldb, ldb2, ldc, ldc2, ldaux, n_split, k_per_split, relu, accumulate, nbias;
    float sa, sb, inv;      // PREC = 4: power-of-two pre-scales of the A / B operands and 1 / (sa sb)
    int c_bf16, aux_bf16;   // bf16 storage (configs[4]): C written as bf16 [M][ldc]; aux read as bf16 [M][ldaux]
    long long* dbg;         // developer builds (DC_DEV_TIMING): phase clocks of workgroup 0, thread 0
};

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2v;
__device__ __forceinline__ unsigned pk_f16(float a, float b) {      // {f16(b), f16(a)}, round to nearest even (v_cvt_pk_f16_f32)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, f16x2));
}

// ---- operand loaders: global -> registers (issued one K step ahead) -> split -> LDS planes ---------------------------
// MODE X3_ROW   : f32 [rows][ld], k contiguous           -> planes [row][16 k] bf16, 48-byte rows
// MODE X3_KMAJ  : f32 [k][ld], rows contiguous           -> planes [8 kpairs][128 rows] u32
// MODE X3_PLANES: bf16 [NP][rows][ld] pre-split weights  -> planes [row][16 k] bf16 (no arithmetic); NP = 1 is also an ACTIVATION
//                 stored as bf16 [rows][ld] (bf16 storage of configs[4]: X3Gemm::a_bf16)
// MODE X3_KMAJ16: bf16 [k][ld], rows contiguous          -> planes [8 kpairs][128 rows] u32 (PREC = 1; the pairs along k are packed
//                 with two v_perm - no conversion)
// KH: sub-steps of sixteen k per staged step (X3Cfg::kKH); sub-step h of piece p lives in LDS plane h * NP + p
template <int MODE, int NP, bool F16 = false, int KH = 1>
struct X3Loader {
    struct Regs { float4 v[2 * KH]; u32x4 w[NP * KH]; uint2 h[2 * KH]; };   // one staged K step of this thread (only the members its MODE uses are live)
    // Which tile row a thread stages (row-major modes).  A ds_write_b64 / _b128 is serviced in contiguous groups of 16 / 8 lanes = four
    // row slots, over 32 banks; with 48-byte rows four CONSECUTIVE rows put rows 0 and 3 on the same banks (two cycles per group: the
    // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.32 of these kernels, all of it from the stores); rows r, r + 2, r + 4, r + 6 tile the 32
    // banks exactly.  Slot p -> row: groups of four slots take the even / the odd rows of a block of eight.
    static __device__ __forceinline__ int lrow(int p) { const int m = p >> 2; return 8 * (m >> 1) + (m & 1) + 2 * (p & 3); }
    const char* src[2];
    long long step;        // bytes per K step
    long long sub;         // bytes between the sub-steps of a step
    long long plane;       // bytes between planes (X3_PLANES)
    float scale;           // F16: power-of-two pre-scale of this operand

    __device__ __forceinline__ void init(const void* P, int ld, long long plane_elems, int r_base, int R, int k0, int tid, float sc = 1.f) {
        scale = sc;
        if constexpr (MODE == X3_ROW) {
            const int kc = (tid & 3) * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = min(r_base + lrow(tid >> 2) + 64 * i, R - 1);
                src[i] = reinterpret_cast<const char*>(static_cast<const float*>(P) + (size_t)row * ld + k0 + kc);
            }
            sub = XK * 4;
        } else if constexpr (MODE == X3_KMAJ) {
            const int kp = tid >> 5;
            const int c = min(r_base + (tid & 31) * 4, ld - 4);   // stay inside the physical row (columns past R are never stored)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                src[i] = reinterpret_cast<const char*>(static_cast<const float*>(P) + (size_t)(k0 + 2 * kp + i) * ld + c);
            sub = (long long)XK * ld * 4;
        } else if constexpr (MODE == X3_KMAJ16) {
            const int kp = tid >> 5;
            const int c = min(r_base + (tid & 31) * 4, ld - 4);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                src[i] = reinterpret_cast<const char*>(static_cast<const uint16_t*>(P) + (size_t)(k0 + 2 * kp + i) * ld + c);
            sub = (long long)XK * ld * 2;
        } else {
            const int row = min(r_base + lrow(tid >> 1), R - 1);
            src[0] = reinterpret_cast<const char*>(static_cast<const uint16_t*>(P) + (size_t)row * ld + k0 + (tid & 1) * 8);
            src[1] = nullptr;
            sub = XK * 2;
            plane = plane_elems * 2;
        }
        step = KH * sub;
    }
    // nsub (workgroup-uniform): sub-steps of this step that exist (the last step of a K that is an odd multiple of 16 has one): the
    // others are staged as zeros
    __device__ __forceinline__ void load(Regs& r, int nsub = KH) {
#pragma unroll
        for (int h = 0; h < KH; ++h) {
            const bool on = KH == 1 || h < nsub;
            if constexpr (MODE == X3_PLANES) {
#pragma unroll
                for (int p = 0; p < NP; ++p) r.w[h * NP + p] = on ? *reinterpret_cast<const u32x4*>(src[0] + p * plane + h * sub) : u32x4{0u, 0u, 0u, 0u};
            } else if constexpr (MODE == X3_KMAJ16) {
                r.h[2 * h] = on ? *reinterpret_cast<const uint2*>(src[0] + h * sub) : make_uint2(0u, 0u);
                r.h[2 * h + 1] = on ? *reinterpret_cast<const uint2*>(src[1] + h * sub) : make_uint2(0u, 0u);
            } else {
                r.v[2 * h] = on ? *reinterpret_cast<const float4*>(src[0] + h * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
                r.v[2 * h + 1] = on ? *reinterpret_cast<const float4*>(src[1] + h * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        src[0] += step;
        if constexpr (MODE != X3_PLANES) src[1] += step;
    }
    // split2: two f32 -> packed bf16 pairs of the three pieces (low half = first argument); F16: scaled, two f16 pieces
    static __device__ __forceinline__ void split2(float a, float b, unsigned (&o)[3], float sc = 1.f) {
        if constexpr (F16) {
            a *= sc; b *= sc;
            o[0] = pk_f16(a, b);
            const f16x2 hv = __builtin_bit_cast(f16x2, o[0]);
            o[1] = pk_f16(a - (float)hv.x, b - (float)hv.y);
            return;
        }
        o[0] = cvt_pk_bf16(a, b);
        if constexpr (NP > 1) {
            const float ra = a - __uint_as_float(o[0] << 16), rb = b - __uint_as_float(o[0] & 0xffff0000u);
            o[1] = cvt_pk_bf16(ra, rb);
            const float sa = ra - __uint_as_float(o[1] << 16), sb = rb - __uint_as_float(o[1] & 0xffff0000u);
            o[2] = cvt_pk_bf16(sa, sb);
        }
    }
    __device__ __forceinline__ void store(const Regs& r, char* __restrict__ S, int tid) const {
        const float sc = scale;
#pragma unroll
        for (int hh = 0; hh < KH; ++hh) {
        if constexpr (MODE == X3_ROW) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 v = r.v[2 * hh + i];
                unsigned lo[3], hi[3];
                split2(v.x, v.y, lo, sc);
                split2(v.z, v.w, hi, sc);
                char* d = S + (lrow(tid >> 2) + 64 * i) * RM_ROW_BYTES + (tid & 3) * 8;
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(d + (hh * NP + p) * RM_PLANE) = make_uint2(lo[p], hi[p]);
            }
        } else if constexpr (MODE == X3_KMAJ) {
            // v[0] = k even, v[1] = k odd, four consecutive rows each: pack the pair along k
            const float4 v0 = r.v[2 * hh], v1 = r.v[2 * hh + 1];
            unsigned q[4][3];
            split2(v0.x, v1.x, q[0], sc); split2(v0.y, v1.y, q[1], sc); split2(v0.z, v1.z, q[2], sc); split2(v0.w, v1.w, q[3], sc);
            char* d = S + ((tid >> 5) * XB + (tid & 31) * 4) * 4;
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(d + (hh * NP + p) * KM_PLANE) = u32x4{q[0][p], q[1][p], q[2][p], q[3][p]};
        } else if constexpr (MODE == X3_KMAJ16) {
            // h[0] = k even, h[1] = k odd, four consecutive rows (bf16) each: {even, odd} of a row share a dword
            static_assert(MODE != X3_KMAJ16 || NP == 1, "bf16 storage goes with PREC = 1");
            const uint2 e = r.h[2 * hh], o = r.h[2 * hh + 1];
            char* d = S + ((tid >> 5) * XB + (tid & 31) * 4) * 4 + hh * KM_PLANE;
            *reinterpret_cast<u32x4*>(d) = u32x4{__builtin_amdgcn_perm(o.x, e.x, 0x05040100u), __builtin_amdgcn_perm(o.x, e.x, 0x07060302u),
                                                  __builtin_amdgcn_perm(o.y, e.y, 0x05040100u), __builtin_amdgcn_perm(o.y, e.y, 0x07060302u)};
        } else {
            char* d = S + lrow(tid >> 1) * RM_ROW_BYTES + (tid & 1) * 16;
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(d + (hh * NP + p) * RM_PLANE) = r.w[hh * NP + p];
        }
        }
    }
    // the 8 k values (8g .. 8g+7) of tile row r, plane p, as an MFMA operand
    static __device__ __forceinline__ bf16x8 frag(const char* __restrict__ S, int r, int g, int p) {
        if constexpr (MODE == X3_KMAJ || MODE == X3_KMAJ16) {
            const unsigned* s = reinterpret_cast<const unsigned*>(S + p * KM_PLANE) + (4 * g) * XB + r;
            return __builtin_bit_cast(bf16x8, u32x4{s[0], s[XB], s[2 * XB], s[3 * XB]});
        } else {
            return *reinterpret_cast<const bf16x8*>(S + p * RM_PLANE + r * RM_ROW_BYTES + g * 16);
        }
    }
};

static __device__ __forceinline__ int p_splits(const X3Args& p) { return (p.K + p.k_per_split - 1) / p.k_per_split; }

// epilogue of one work item
template <int PREC>
static __device__ __forceinline__ void store_tile(const X3Args& p, f32x16 (&acc)[2][2], char* smem, int m_blk, int n_blk, int z, int wave,
                                                  int wm, int wn, int lane) {
    const int fr = lane & 31, fg = lane >> 5;
    // ---- epilogue: accumulators -> this wave's 64 x 64 LDS image -> 16-byte rows ----------------------------------------
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    constexpr int IMG_ROWS = X3Cfg<PREC>::kImgRows;      // 64: the whole 64 x 64 block at once; 32: its two row halves one after the other
    float* img = reinterpret_cast<float*>(smem) + wave * (IMG_ROWS * EP_LD);
#pragma unroll
    for (int half = 0; half < 64 / IMG_ROWS; ++half) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (IMG_ROWS == 32 && i != half) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                img[((IMG_ROWS == 64 ? i * 32 : 0) + (r & 3) + 8 * (r >> 2) + 4 * fg) * EP_LD + j * 32 + fr] = PREC == 4 ? acc[i][j][r] * p.inv : acc[i][j][r];
    }
    // (each wave reads back only what it wrote: no workgroup barrier needed, the LDS is in order per wave)
    const int c4 = (lane & 15) * 4;
    const int col = n_blk + wn * 64 + c4;
    const bool col_ok = col < p.N;            // N is a multiple of 4 wherever this kernel is used (host check)
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias != nullptr && p.slab == nullptr) {
        if (col + 0 < p.nbias) bv.x = p.bias[col + 0];
        if (col + 1 < p.nbias) bv.y = p.bias[col + 1];
        if (col + 2 < p.nbias) bv.z = p.bias[col + 2];
        if (col + 3 < p.nbias) bv.w = p.bias[col + 3];
    }
    const bool to_c2 = p.n_split > 0 && col >= p.n_split;
#pragma unroll 4
    for (int it = 0; it < IMG_ROWS / 4; ++it) {
        const int rl = it * 4 + (lane >> 4);
        const int row = m_blk + wm * 64 + half * IMG_ROWS + rl;
        float4 v = *reinterpret_cast<const float4*>(img + rl * EP_LD + c4);
        if (row >= p.M || !col_ok) continue;
        if (p.slab != nullptr) {
            *reinterpret_cast<float4*>(p.slab + ((size_t)z * p.M + row) * p.N + col) = v;
            continue;
        }
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (p.relu) { v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w); }      // NaN-propagating (common.h)
        if (p.aux != nullptr) {
            float4 m;
            if (p.aux_bf16) {
                const uint2 mb = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.aux) + (size_t)row * p.ldaux + col);
                m = make_float4(__uint_as_float(mb.x << 16), __uint_as_float(mb.x & 0xffff0000u), __uint_as_float(mb.y << 16), __uint_as_float(mb.y & 0xffff0000u));
            } else {
                m = *reinterpret_cast<const float4*>(p.aux + (size_t)row * p.ldaux + col);
            }
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        }
        if (p.c_bf16) {       // bf16 storage: 8 bytes per lane (no accumulate / second output on this form: host check)
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.C) + (size_t)row * p.ldc + col) = make_uint2(cvt_pk_bf16(v.x, v.y), cvt_pk_bf16(v.z, v.w));
            continue;
        }
        float* c = to_c2 ? p.C2 + (size_t)row * p.ldc2 + (col - p.n_split) : p.C + (size_t)row * p.ldc + col;
        if (p.accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(c);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *reinterpret_cast<float4*>(c) = v;      // (as non-temporal stores: no change, +- 2 %, measured round 5)
    }
    }   // row halves
}

// Persistent: a workgroup walks the (row tile, column tile, K split) work items w = first, first + stride, ...; the loads
// of the NEXT item's first two K steps are issued before the epilogue of the current one.  Inside an item the loads run
// two K steps ahead of the MFMAs (two register staging sets), the split + LDS store one step ahead (two LDS stages).
template <int PREC, int A_MODE, int B_MODE>
__global__ __launch_bounds__(256, X3Cfg<PREC>::kOcc) void gemm_x3_kernel(X3Args p, int n_items, int mt, int nt) {
    constexpr int OPER_BYTES = X3Cfg<PREC>::kOper, STAGE_BYTES = X3Cfg<PREC>::kStage;      // (shadow the three-plane sizes)
    constexpr int NP = PREC == 1 ? 1 : (PREC == 4 ? 2 : 3);
    constexpr bool F16 = PREC == 4;
    constexpr int KH = x3_kh<PREC, A_MODE, B_MODE>(), NPL = NP * KH;    // sub-steps of 16 k per step; LDS planes per operand
    using LA = X3Loader<A_MODE, NP, F16, KH>;
    using LB = X3Loader<B_MODE, NP, F16, KH>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fg = lane >> 5;

    // work item -> (m tile, n tile, split).  With a multiple of 8 row tiles, the workgroups of one XCD (blockIdx & 7 under the
    // usual round-robin placement - a speed hint only) take the column tiles of the same row tile one after the other, so
    // the A rows they share stay in that XCD's L2.
    // Split-K products (the weight gradients) with a multiple of 8 splits: an XCD takes whole SPLITS (z = x, x + 8, ...) - every tile
    // of a K range runs in the same XCD, so both operands of that range are fetched from HBM once (by row tile, the x rows of a range
    // went through all eight L2s: the gates' weight gradient read 1.3 GB instead of 0.4).
    const int n_splits = p_splits(p);
#ifdef DC_X3_NO_ZMAP          // A/B build
    const bool z_map = false;
#else
    const bool z_map = n_splits >= 8 && (n_splits & 7) == 0 && (gridDim.x & 7) == 0;
#endif
    const bool xcd_map = !z_map && (mt & 7) == 0 && (gridDim.x & 7) == 0;
    auto decode = [&](int it, int& m_blk, int& n_blk, int& z) -> bool {
        int w;
        if (z_map) {
            const int per = n_items >> 3;                                   // items per XCD slice
            const int wi = it * (gridDim.x >> 3) + (blockIdx.x >> 3);
            if (wi >= per) return false;
            const int tiles = mt * nt;
            const int zl = wi / tiles, t = wi - zl * tiles;
            z = zl * 8 + (int)(blockIdx.x & 7);
            m_blk = (t / nt) * XB; n_blk = (t % nt) * XB;
            return true;
        }
        if (xcd_map) {
            const int per = n_items >> 3;                                   // items per XCD slice
            const int wi = it * (gridDim.x >> 3) + (blockIdx.x >> 3);
            if (wi >= per) return false;
            const int x = blockIdx.x & 7;
            const int per_m = nt * p_splits(p);
            const int ml = wi / per_m, rest = wi - ml * per_m;
            m_blk = (ml * 8 + x) * XB; n_blk = (rest % nt) * XB; z = rest / nt;
            (void)w;
            return true;
        }
        w = it * gridDim.x + blockIdx.x;
        if (w >= n_items) return false;
        const int tiles = mt * nt;
        z = w / tiles;
        const int t = w - z * tiles;
        m_blk = (t / nt) * XB; n_blk = (t % nt) * XB;
        return true;
    };

    LA la;
    LB lb;
    typename LA::Regs ra0, ra1;
    typename LB::Regs rb0, rb1;
    int nk = 0, n16 = 0;                                      // steps / sub-steps of sixteen k of the open item
    auto nsub = [&](int kt) { return min(KH, n16 - kt * KH); };
    auto open_item = [&](int m_blk, int n_blk, int z) {       // address state + the loads of K steps 0 and 1
        const int k_begin = z * p.k_per_split;
        n16 = (min(p.K, k_begin + p.k_per_split) - k_begin) / XK;
        nk = (n16 + KH - 1) / KH;
        // B of this column tile: the second operand of a pair behind n_split (workgroup-uniform)
        const bool second = p.n_split > 0 && n_blk >= p.n_split;
        la.init(p.A, p.lda, p.a_plane, m_blk, p.M, k_begin, tid, p.sa);
        lb.init(second ? p.B2 : p.B, second ? p.ldb2 : p.ldb, second ? p.b2_plane : p.b_plane, second ? n_blk - p.n_split : n_blk,
                p.n_split > 0 ? (second ? p.N - p.n_split : p.n_split) : p.N, k_begin, tid, p.sb);
        la.load(ra0, nsub(0)); lb.load(rb0, nsub(0));
        if (nk > 1) { la.load(ra1, nsub(1)); lb.load(rb1, nsub(1)); }
    };

    // developer builds (DC_DEV_TIMING): s_memtime stamps of (workgroup 0, thread 0), summed per phase (a stamp waits for the LDS
    // operations in flight: the phases are serialised in that wave)
    long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    const bool timing = DC_DEV_TIMING && p.dbg != nullptr && blockIdx.x == 0 && tid == 0;
    auto stamp = [&](int k) {
        if (DC_DEV_TIMING && timing) { const long long now = (long long)__builtin_amdgcn_s_memtime(); tm[k] += now - tlast; tlast = now; }
    };
    int m_blk, n_blk, z;
    if (!decode(0, m_blk, n_blk, z)) return;
    open_item(m_blk, n_blk, z);
    if (DC_DEV_TIMING && timing) tlast = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0;; ++it) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // column sums of A (k-major A only): the work items of column tile 0 add up what passes through their loader
        const bool do_cs = (A_MODE == X3_KMAJ || A_MODE == X3_KMAJ16) && p.a_colsum != nullptr && n_blk == 0;
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
        auto cs_add = [&](const typename LA::Regs& r) {
#pragma unroll
            for (int h = 0; h < 2 * KH; ++h) {
                if constexpr (A_MODE == X3_KMAJ) {
                    cs.x += r.v[h].x; cs.y += r.v[h].y; cs.z += r.v[h].z; cs.w += r.v[h].w;
                } else if constexpr (A_MODE == X3_KMAJ16) {
                    cs.x += __uint_as_float(r.h[h].x << 16); cs.y += __uint_as_float(r.h[h].x & 0xffff0000u);
                    cs.z += __uint_as_float(r.h[h].y << 16); cs.w += __uint_as_float(r.h[h].y & 0xffff0000u);
                }
            }
        };
        if (do_cs) cs_add(ra0);
        la.store(ra0, smem, tid); lb.store(rb0, smem + OPER_BYTES, tid);
        __syncthreads();
        // one K step: loads of step kt + 2 -> the staging set that step kt just vacated; MFMAs of step kt; split + store of kt + 1
        auto kstep = [&](int kt, typename LA::Regs& ra_cur, typename LB::Regs& rb_cur, const typename LA::Regs& ra_nxt,
                         const typename LB::Regs& rb_nxt) {
            stamp(7);      // (loop control)
            if (kt + 2 < nk) { la.load(ra_cur, nsub(kt + 2)); lb.load(rb_cur, nsub(kt + 2)); }
            stamp(0);      // global loads issued
            const char* a_s = smem + (kt & 1) * STAGE_BYTES;
            const char* b_s = a_s + OPER_BYTES;
            bf16x8 a[2][NPL], b[2][NPL];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    a[i][q] = LA::frag(a_s, wm * 64 + i * 32 + fr, fg, q);
                    b[i][q] = LB::frag(b_s, wn * 64 + i * 32 + fr, fg, q);
                }
            stamp(1);      // fragment reads (+ the wait for them)
            // piece-major order, smallest terms first: the four accumulators take turns, consecutive MFMAs never chain
#define DC_X3_P(X, Y)                                                                                             \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                         \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                     \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][X], b[j][Y], acc[i][j], 0, 0, 0);
#define DC_X2H_P(X, Y)                                                                                            \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                         \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                     \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][X]), __builtin_bit_cast(f16x8, b[j][Y]), acc[i][j], 0, 0, 0);
            if constexpr (PREC == 4) { DC_X2H_MM(DC_X2H_P(1, 1)) DC_X2H_P(1, 0) DC_X2H_P(0, 1) DC_X2H_P(0, 0) }
            else {
            if constexpr (PREC == 6) { DC_X3_P(2, 0) DC_X3_P(0, 2) DC_X3_P(1, 1) DC_X3_P(1, 0) DC_X3_P(0, 1) }
            DC_X3_P(0, 0)
            if constexpr (KH == 2) { DC_X3_P(1, 1) }      // PREC = 1: plane 1 = the next sixteen k
            }
#undef DC_X2H_P
#undef DC_X3_P
            stamp(2);      // MFMAs issued
            if (kt + 1 < nk) {
                char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
                if (do_cs) cs_add(ra_nxt);
                la.store(ra_nxt, nxt, tid); lb.store(rb_nxt, nxt + OPER_BYTES, tid);
            }
            stamp(3);      // wait for the next step's global loads + split + LDS stores
            __syncthreads();
            stamp(4);      // barrier
        };
        for (int kt = 0; kt < nk; kt += 2) {
            kstep(kt, ra0, rb0, ra1, rb1);
            if (kt + 1 < nk) kstep(kt + 1, ra1, rb1, ra0, rb0);
        }

        if (do_cs) {     // thread (kp = tid >> 5, c4 = tid & 31) holds the sums of rows m_blk + 4 c4 .. + 3 over its k of every pair: fold the
                         // eight k lanes through the 4 KB of LDS behind the epilogue images, one atomic per row (like colsum_kernel's)
            float* red = reinterpret_cast<float*>(smem + 4 * X3Cfg<PREC>::kImgRows * EP_LD * 4);
            *reinterpret_cast<float4*>(red + (tid >> 5) * XB + (tid & 31) * 4) = cs;
            __syncthreads();
            if (tid < XB && m_blk + tid < p.M) {
                float acc = 0.f;
#pragma unroll
                for (int g = 0; g < 8; ++g) acc += red[g * XB + tid];
                atomicAdd(p.a_colsum + m_blk + tid, acc);
            }
        }
        // the next item's first loads fly during this item's epilogue
        const int cm = m_blk, cn = n_blk, cz = z;
        const bool have_next = decode(it + 1, m_blk, n_blk, z);
        if (have_next) open_item(m_blk, n_blk, z);
        stamp(5);      // item switch: first store + barrier of the item, next item's first loads
        // (round 5 ablations on the network's shapes, profiles/r05/gemm_x3_ablation.txt: without this epilogue the x W^T product of the gates
        //  runs 180 -> 133 us (prec 4) and configs[4]'s 680 -> 405 us (prec 1, f32 C): the K loop alone reaches 0.28-0.31 of the f16x2
        //  ceiling / 0.27 of the bf16 peak; starting the workgroups of a CU a third of an item apart changed nothing)
        store_tile<PREC>(p, acc, smem, cm, cn, cz, wave, wm, wn, lane);
        if (!have_next) break;
        __syncthreads();          // the epilogue images live in the stage buffers the next item is about to fill
        stamp(6);      // epilogue
    }
    if (DC_DEV_TIMING && timing) { for (int k = 0; k < 8; ++k) p.dbg[k] = tm[k]; }
}

// weights -> bf16 planes, in the orientation the consumer contracts over
struct SplitJob { const float* src; uint16_t* dst; int rows, cols, ld, transpose, rows_pad; };
struct SplitJobs { SplitJob j[16]; int n; float scale; };
template <int NP, bool F16 = false>
__global__ __launch_bounds__(256) void split_planes_kernel(SplitJobs jobs) {
    const SplitJob jb = jobs.j[blockIdx.y];
    const int n_out_rows = jb.transpose ? jb.cols : jb.rows_pad, n_out_cols = jb.transpose ? jb.rows_pad : jb.cols;
    const long long total = (long long)n_out_rows * n_out_cols;
    for (long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 2; e < total; e += (long long)gridDim.x * 512) {
        const int orow = (int)(e / n_out_cols), ocol = (int)(e - (long long)orow * n_out_cols);   // n_out_cols is even
        float x[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = jb.transpose ? ocol + i : orow, c = jb.transpose ? orow : ocol + i;
            x[i] = r < jb.rows ? jb.src[(size_t)r * jb.ld + c] : 0.f;   // rows_pad > rows: zero rows (K padding of dH)
        }
        unsigned o[3];
        X3Loader<X3_ROW, NP, F16>::split2(x[0], x[1], o, jobs.scale);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<unsigned*>(jb.dst + p * total + e) = o[p];
    }
}

}  // namespace

bool gemm_x3_shape_ok(int M, int N, int K, int lda, int ldb, int a_mode, int b_mode) {
    if (K % XK || (N & 3) || M <= 0 || N <= 0) return false;
    if (a_mode == X3_ROW && (lda & 3)) return false;
    if ((a_mode == X3_KMAJ || a_mode == X3_KMAJ16) && ((lda & 3) || lda < 4)) return false;
    if ((b_mode == X3_KMAJ || b_mode == X3_KMAJ16) && ((ldb & 3) || ldb < 4)) return false;
    if (a_mode == X3_PLANES && (lda & 7)) return false;
    if (b_mode == X3_PLANES && (ldb & 7)) return false;
    return true;
}

template <int PREC, int AM, int BM_>
static int launch_x3(const X3Args& a, int splits, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_x3_kernel<PREC, AM, BM_>, hipFuncAttributeMaxDynamicSharedMemorySize, X3Cfg<PREC>::kLds);
        if (e != hipSuccess) { set_error("gemm_x3: hipFuncSetAttribute", (int)e); return (int)e; }
        attr_done = true;
    }
    const int mt = (a.M + XB - 1) / XB, nt = (a.N + XB - 1) / XB;
    const int n_items = mt * nt * splits;
    static const int slots = [] {      // two (three: X3Cfg) workgroups per CU (LDS-limited)
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        return X3Cfg<PREC>::kOcc * cus;
    }();
    const int grid = n_items < slots ? n_items : slots;
#if DC_DEV_TIMING
    static long long* dbg = nullptr;
    if (!dbg) (void)hipMalloc(&dbg, 64);
    (void)hipMemsetAsync(dbg, 0, 64, s);
    X3Args at = a;
    at.dbg = dbg;
    hipLaunchKernelGGL((gemm_x3_kernel<PREC, AM, BM_>), dim3(grid), dim3(256), X3Cfg<PREC>::kLds, s, at, n_items, mt, nt);
    {
        long long h[8];
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        const int kh = x3_kh<PREC, AM, BM_>();
        const double items = (double)((n_items + grid - 1) / grid), steps = items * ((a.k_per_split / XK + kh - 1) / kh);
        fprintf(stderr, "gemm_x3<%d,%d,%d> M %d N %d K %d items/wg %.0f steps/item %.0f | clocks per step: loads %.0f  frag reads %.0f  mfma issue %.0f  split+store %.0f  "
                        "barrier %.0f  loop %.0f | per item: switch %.0f  epilogue %.0f\n", PREC, AM, BM_, a.M, a.N, a.K, items, steps / items, h[0] / steps, h[1] / steps,
                h[2] / steps, h[3] / steps, h[4] / steps, h[7] / steps, h[5] / items, h[6] / items);
    }
#else
    hipLaunchKernelGGL((gemm_x3_kernel<PREC, AM, BM_>), dim3(grid), dim3(256), X3Cfg<PREC>::kLds, s, a, n_items, mt, nt);
#endif
    return launch_check("gemm_x3");
}

// C[M,N] (op)= A * B^T-ish: A per a_mode (X3_ROW [M][lda] or X3_KMAJ [K][lda]), B per b_mode (X3_PLANES [NP][N][ldb] bf16,
// X3_KMAJ [K][ldb] f32); optional second B behind column n_split (the dW_ih | dW_hh pair); split-K through the slab.
int gemm_x3(const X3Gemm& g, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0) return 0;
    if (!gemm_x3_shape_ok(g.M, g.N, g.K, g.lda, g.ldb, g.a_mode, g.b_mode)) { set_error("gemm_x3: unsupported shape", 1004); return 1004; }
    X3Args a{};
    a.A = g.A; a.B = g.B; a.B2 = g.B2; a.C = g.C; a.C2 = g.C2; a.bias = g.bias; a.aux = g.aux; a.slab = nullptr;
    a.a_plane = 0; a.b_plane = g.b_plane; a.b2_plane = g.b2_plane;
    a.M = g.M; a.N = g.N; a.K = g.K; a.lda = g.lda; a.ldb = g.ldb; a.ldb2 = g.ldb2; a.ldc = g.ldc; a.ldc2 = g.ldc2; a.ldaux = g.ldaux;
    a.n_split = g.n_split; a.relu = g.relu; a.accumulate = g.accumulate; a.nbias = g.bias ? g.nbias : 0;
    // bf16 storage of configs[4] (prec 1 only): an activation / gradient operand that lives in HBM as bf16 takes the loader of its layout
    // that does no arithmetic; the output and the relu mask may be bf16 as well
    const int a_mode = g.a_bf16 ? (g.a_mode == X3_ROW ? X3_PLANES : X3_KMAJ16) : g.a_mode;
    const int b_mode = g.b_bf16 ? X3_KMAJ16 : g.b_mode;
    if ((g.a_bf16 || g.b_bf16 || g.c_bf16 || g.aux_bf16) &&
        (g.prec != 1 || (g.a_bf16 && g.a_mode == X3_PLANES) || (g.b_bf16 && g.b_mode != X3_KMAJ) ||
         (g.c_bf16 && (g.accumulate || g.C2 != nullptr || g.scratch.p != nullptr)) || (a_mode == X3_PLANES && (g.lda & 7)))) {
        set_error("gemm_x3: bf16 storage asked for a form that is not built", 1005);
        return 1005;
    }
    a.a_colsum = (g.a_mode == X3_KMAJ) ? g.a_colsum : nullptr;
    a.c_bf16 = g.c_bf16; a.aux_bf16 = g.aux_bf16;
    a.sa = g.sa; a.sb = g.sb; a.inv = 1.f / (g.sa * g.sb);
    int splits = 1;
    const long tiles = (long)((g.M + XB - 1) / XB) * ((g.N + XB - 1) / XB);
    if (tiles < 256 && g.K >= 4096 && !g.relu && g.aux == nullptr && g.scratch.p != nullptr) {
        long want = (g.prec != 6 && DC_X3_OCC == 3 ? 768 : 512) / tiles;      // two (three) workgroups per CU
        if (want < 1) want = 1;
        const long maxs = g.K / 512;
        splits = (int)(want < maxs ? want : maxs);
        if (splits < 1) splits = 1;
        while (splits > 1 && (long long)splits * g.M * g.N > g.scratch.floats) --splits;
        if (splits >= 8) splits = splits / 8 * 8;       // whole splits per XCD (the kernel's z_map)
    }
    int kper = (g.K + splits - 1) / splits;
    const int kgran = g.prec == 1 ? XK * X3Cfg<1>::kKHmax : XK;      // whole steps per split
    kper = (kper + kgran - 1) / kgran * kgran;
    splits = (g.K + kper - 1) / kper;
    a.k_per_split = kper;
    if (splits > 1) a.slab = g.scratch.p;
    const char* name = g.a_mode == X3_KMAJ ? "gemm_f32_dW(TN,split-K)" : (g.transposed_w ? "gemm_f32_dX(NN)" : "gemm_f32_fwd(NT)");
    ProfScope prof(name, 2.0 * g.M * (double)g.N * g.K,
                   (g.a_bf16 ? 2.0 : 4.0) * g.M * g.K + (g.b_bf16 || g.b_mode == X3_PLANES ? 2.0 : 4.0) * g.N * g.K + (g.c_bf16 ? 2.0 : 4.0) * g.M * g.N, stream);
    if (!g.tile128 && gemm_x3s_eligible(g)) return gemm_x3s(g, stream);      // x W^T / dy W in f16x2 mode: the row-streaming kernel (gemm_x3s.hip)
    int rc;
    if (a_mode == X3_PLANES && b_mode == X3_PLANES) rc = launch_x3<1, X3_PLANES, X3_PLANES>(a, splits, stream);
    else if (a_mode == X3_KMAJ16 && b_mode == X3_KMAJ) rc = launch_x3<1, X3_KMAJ16, X3_KMAJ>(a, splits, stream);
    else if (a_mode == X3_KMAJ16 && b_mode == X3_KMAJ16) rc = launch_x3<1, X3_KMAJ16, X3_KMAJ16>(a, splits, stream);
    else if (a_mode == X3_KMAJ && b_mode == X3_KMAJ16) rc = launch_x3<1, X3_KMAJ, X3_KMAJ16>(a, splits, stream);
    else if (g.a_mode == X3_ROW && g.b_mode == X3_PLANES)
        rc = g.prec == 1 ? launch_x3<1, X3_ROW, X3_PLANES>(a, splits, stream)
                         : (g.prec == 4 ? launch_x3<4, X3_ROW, X3_PLANES>(a, splits, stream) : launch_x3<6, X3_ROW, X3_PLANES>(a, splits, stream));
    else if (g.a_mode == X3_KMAJ && g.b_mode == X3_KMAJ)
        rc = g.prec == 1 ? launch_x3<1, X3_KMAJ, X3_KMAJ>(a, splits, stream)
                         : (g.prec == 4 ? launch_x3<4, X3_KMAJ, X3_KMAJ>(a, splits, stream) : launch_x3<6, X3_KMAJ, X3_KMAJ>(a, splits, stream));
    else { set_error("gemm_x3: operand layout combination not built", 1005); return 1005; }
    if (rc == 0 && splits > 1) {
        // C = (C +) sum_z slab[z]; the pair form scatters columns >= n_split into C2
        rc = splitk_reduce_pair(a.slab, g.C, g.M, g.N, g.ldc, splits, g.accumulate, g.C2, g.ldc2, g.n_split, stream);
    }
    return rc;
}

int split_weight_planes(const X3SplitJob* jobs, int n, int prec, hipStream_t s, float scale) {
    if (n <= 0) return 0;
    if (n > 16) { set_error("split_weight_planes: too many jobs", 1006); return 1006; }
    SplitJobs sj{};
    sj.n = n; sj.scale = scale;
    for (int i = 0; i < n; ++i) {
        sj.j[i] = SplitJob{jobs[i].src, jobs[i].dst, jobs[i].rows, jobs[i].cols, jobs[i].ld, jobs[i].transpose,
                           jobs[i].rows_pad > jobs[i].rows ? jobs[i].rows_pad : jobs[i].rows};
    }
    if (prec == 1) hipLaunchKernelGGL(split_planes_kernel<1>, dim3(64, n), dim3(256), 0, s, sj);
    else if (prec == 4) hipLaunchKernelGGL((split_planes_kernel<2, true>), dim3(64, n), dim3(256), 0, s, sj);
    else hipLaunchKernelGGL(split_planes_kernel<3>, dim3(64, n), dim3(256), 0, s, sj);
    return launch_check("split_weight_planes");
}

}  // namespace dc
