// fp32 GEMM on the CDNA4 matrix cores: C[M,N] (op)= A[M,K] * B[K,N]  (+bias, relu, mask).
//
// This is the dense-contraction workhorse of the network (every nn.Linear of
// /root/reference/policy.py:54-75 forward, and the dX / dW products of its backward, which the
// reference gets from torch autograd at /root/reference/optimizer.py:672).  Inputs and accumulation
// are exact fp32 (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain), which is what lets the 1e-4 parity
// bar of BASELINE.json hold without a reduced-precision path.
//
// Layout: each operand is described by where its K index lives.
//   A_KM = false : A stored row-major [M][lda], k contiguous  (activations, x of y = x W^T)
//   A_KM = true  : A stored [K][lda],  m contiguous           (dY^T of dW = dY^T X)
//   B_KM = false : B stored [N][ldb],  k contiguous           (a torch Linear weight [out,in])
//   B_KM = true  : B stored [K][ldb],  n contiguous           (W of dX = dY W, X of dW = dY^T X)
//
// Tiling: 256 threads = 4 waves in a 2x2 grid; block tile BM x BN, K step 32; each wave owns
// (BM/2)x(BN/2) as (BM/64)x(BN/64) MFMA tiles of 32x32.  Operand tiles are staged global -> VGPR
// -> LDS with the next tile's global loads in flight during the MFMAs of the current one (two LDS
// buffers, one barrier per K step).  k-contiguous tiles sit in LDS as [row][33] (pad 1 makes both
// the 4-scalar staging writes and the 32-lane fragment reads conflict-free for ds_*_b32);
// k-major tiles sit as [k][rows] (fragment reads are lane-consecutive, staging is ds_write_b128).
// Split-K (gridDim.z) accumulates with fp32 global atomics into a caller-zeroed / live C.
#include "kernels.h"
#include "gemm_tiles.h"

namespace dc {

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;  // [N] or nullptr (added by split 0 only)
    const float* aux;   // [M][ldaux] or nullptr: result is zeroed where aux <= 0 (relu backward)
    int M, N, K;
    int lda, ldb, ldc, ldaux;
    int relu;        // apply max(0, .) (needs splits == 1)
    int accumulate;  // C += result (non-atomic read-modify-write; splits == 1)
    int atomic;      // C += result with atomics (split-K without scratch)
    int k_per_split; // multiple of 32
    float* slab;     // split-K with scratch: partial [split][M][N] slabs, reduced by splitk_reduce_kernel
    // two B operands side by side (gemm_f32_tn_pair): columns >= n_split come from B2 (n_split: multiple of the column
    // tile, 0 = off)
    const float* B2;
    int ldb2, n_split;
};


template <int BR, bool KM>
struct TileLoader {
    // registers: BR/32 float4 per thread
    static constexpr int NV = BR / 32;
    float4 v[NV];

    __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int r_base, int R, int k_base,
                                         int k_end, bool vec, int tid) {
        if constexpr (!KM) {
            const int c4 = tid & 7, r0 = tid >> 3;
            const int k = k_base + c4 * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int r = r_base + r0 + 32 * i;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < R) {
                    const float* p = P + (size_t)r * ld + k;
                    if (vec && k + 3 < k_end) {
                        x = *reinterpret_cast<const float4*>(p);
                    } else {
                        if (k + 0 < k_end) x.x = p[0];
                        if (k + 1 < k_end) x.y = p[1];
                        if (k + 2 < k_end) x.z = p[2];
                        if (k + 3 < k_end) x.w = p[3];
                    }
                }
                v[i] = x;
            }
        } else {
            constexpr int V = BR / 4;       // float4 per k-row
            constexpr int KSTEP = 256 / V;  // k-rows covered per pass
            const int c4 = tid % V, k0 = tid / V;
            const int r = r_base + c4 * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k = k_base + k0 + KSTEP * i;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < k_end) {
                    const float* p = P + (size_t)k * ld + r;
                    if (vec && r + 3 < R) {
                        x = *reinterpret_cast<const float4*>(p);
                    } else {
                        if (r + 0 < R) x.x = p[0];
                        if (r + 1 < R) x.y = p[1];
                        if (r + 2 < R) x.z = p[2];
                        if (r + 3 < R) x.w = p[3];
                    }
                }
                v[i] = x;
            }
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ S, int tid) const {
        if constexpr (!KM) {
            const int c4 = tid & 7, r0 = tid >> 3;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                float* d = S + (r0 + 32 * i) * (GEMM_BK + 1) + c4 * 4;
                d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
            }
        } else {
            constexpr int V = BR / 4;
            constexpr int KSTEP = 256 / V;
            const int c4 = tid % V, k0 = tid / V;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                *reinterpret_cast<float4*>(S + (k0 + KSTEP * i) * BR + c4 * 4) = v[i];
            }
        }
    }

    static constexpr int LDS_FLOATS = KM ? GEMM_BK * BR : BR * (GEMM_BK + 1);
    // element (row r, k) of the staged tile
    static __device__ __forceinline__ float frag(const float* __restrict__ S, int r, int k) {
        if constexpr (!KM) return S[r * (GEMM_BK + 1) + k];
        else return S[k * BR + r];
    }
};

template <int BM, int BN, bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
    using LA = TileLoader<BM, A_KM>;
    using LB = TileLoader<BN, B_KM>;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_FL = (LA::LDS_FLOATS + 3) / 4 * 4, B_FL = (LB::LDS_FLOATS + 3) / 4 * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE_FL = A_FL + B_FL;  // buffer b: A at smem + b*STAGE_FL, B right after it

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m_blk = blockIdx.y * BM, n_blk = blockIdx.x * BN;
    const int k_begin = blockIdx.z * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    if (k_begin >= k_end) return;

    const bool vecA = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
    const bool vecB = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.B) & 15) == 0);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    LA la;
    LB lb;
    la.load(p.A, p.lda, m_blk, p.M, k_begin, k_end, vecA, tid);
    lb.load(p.B, p.ldb, n_blk, p.N, k_begin, k_end, vecB, tid);
    la.store(smem, tid);
    lb.store(smem + A_FL, tid);
    __syncthreads();

    const int nk = (k_end - k_begin + GEMM_BK - 1) / GEMM_BK;
    const int fr = lane & 31, fk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nk);
        if (more) {
            la.load(p.A, p.lda, m_blk, p.M, k_begin + (kt + 1) * GEMM_BK, k_end, vecA, tid);
            lb.load(p.B, p.ldb, n_blk, p.N, k_begin + (kt + 1) * GEMM_BK, k_end, vecB, tid);
        }
        const float* a_s = smem + cur * STAGE_FL;
        const float* b_s = a_s + A_FL;
#pragma unroll
        for (int kk = 0; kk < GEMM_BK; kk += 2) {
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = LA::frag(a_s, wm * (BM / 2) + i * 32 + fr, kk + fk);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = LB::frag(b_s, wn * (BN / 2) + j * 32 + fr, kk + fk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            la.store(smem + (cur ^ 1) * STAGE_FL, tid);
            lb.store(smem + (cur ^ 1) * STAGE_FL + A_FL, tid);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const bool first_split = (blockIdx.z == 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n_blk + wn * (BN / 2) + j * 32 + fr;
            if (col >= p.N) continue;
            const float bv = (p.bias != nullptr && first_split) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m_blk + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (p.relu) v = relu_nan(v);
                if (p.aux != nullptr && !(p.aux[(size_t)row * p.ldaux + col] > 0.f)) v = 0.f;
                if (p.slab != nullptr) {
                    p.slab[((size_t)blockIdx.z * p.M + row) * p.N + col] = acc[i][j][r];
                    continue;
                }
                float* c = p.C + (size_t)row * p.ldc + col;
                if (p.atomic) atomicAdd(c, v);
                else if (p.accumulate) *c += v;
                else *c = v;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Fast path: full tiles only (M % BM == 0, N % BN == 0, every split's K range a multiple of 32,
// 16-byte aligned operands) - which is every large product of the network; the kernel above stays as
// the bounds-checked fallback for the 154-wide head projections.
//
// What differs from the fallback:
//   * tiles go global -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no ds_write): the next
//     tile's DMA is issued before the MFMAs of the current tile and lands behind them; hipcc cannot sink
//     it (it writes LDS) the way it sinks ordinary prefetch loads next to their consumers;
//   * the DMA writes LDS lane-linearly (wave-uniform base + 16 B x lane), so a k-contiguous tile is the
//     unpadded image [row][32] and bank conflicts are avoided by an XOR swizzle applied on the SOURCE
//     side: the 16-byte slot p of row r holds k chunk p ^ ((r >> 1) & 7).  A fragment read of chunk c
//     of row i then hits slot (i&1)*8 + (c ^ ((i>>1)&7)) of the 256-byte bank row - 16 distinct slots
//     for the 16 rows of every ds_read_b128 lane group;
//   * one ds_read_b128 feeds four MFMAs: lane (i = lane&31, q = lane>>5) reads k = 8j+4q .. 8j+4q+3
//     of its row, i.e. MFMA step e of group j contracts k = 8j+e (q=0) and 8j+4+e (q=1).  Both
//     operands use the same k permutation, so the sum over k is complete and exact;
//   * k-major tiles ([k][rows], the operands of the weight-gradient products) are lane-linear as they
//     are; their fragments are 4 conflict-free ds_read_b32;
//   * epilogue variants are compile-time (EPI_*), no per-element branches.
// ---------------------------------------------------------------------------------------------------
enum { EPI_PLAIN = 0, EPI_RELU = 1, EPI_MASK = 2, EPI_ACC = 3, EPI_SLAB = 4, EPI_ATOMIC = 5 };

template <int BM, int BN, bool A_KM, bool B_KM, int EPI, bool EDGE>
__global__ __launch_bounds__(256) void gemm_fast_kernel(GemmArgs p) {
    using LA = FastTile<BM, A_KM>;
    using LB = FastTile<BN, B_KM>;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_FL = LA::LDS_FLOATS, B_FL = LB::LDS_FLOATS, STAGE_FL = A_FL + B_FL;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m_blk = blockIdx.y * BM, n_blk = blockIdx.x * BN;
    const int k_begin = blockIdx.z * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    const int nk = (k_end - k_begin) / GEMM_BK;
    const int fr = lane & 31, fq = lane >> 5;
    if (nk <= 0) return;   // host never launches an empty split

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    size_t offa[LA::NI], offb[LB::NI];
    LA::template src_offsets<EDGE>(offa, p.lda, m_blk, p.M, wave, lane);
    // B of this column tile: the second operand of a pair behind n_split (workgroup-uniform)
    const bool second = p.n_split > 0 && n_blk >= p.n_split;
    const float* Bp = second ? p.B2 : p.B;
    const int ldb = second ? p.ldb2 : p.ldb;
    const int nb_blk = second ? n_blk - p.n_split : n_blk;
    const int nb_ext = p.n_split > 0 ? (second ? p.N - p.n_split : p.n_split) : p.N;
    LB::template src_offsets<EDGE>(offb, ldb, nb_blk, nb_ext, wave, lane);
    // operand origins at k = k_begin; advancing one K step adds BK floats (k-contiguous) or BK rows (k-major)
    const float* ga = A_KM ? p.A + (size_t)k_begin * p.lda : p.A + k_begin;
    const float* gb = B_KM ? Bp + (size_t)k_begin * ldb : Bp + k_begin;
    const size_t sa = A_KM ? (size_t)GEMM_BK * p.lda : GEMM_BK;
    const size_t sb = B_KM ? (size_t)GEMM_BK * ldb : GEMM_BK;

    LA::issue(ga, offa, smem, wave);
    LB::issue(gb, offb, smem + A_FL, wave);
    __syncthreads();   // (hipcc drains the DMA - vmcnt(0) - in front of the barrier)

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {   // next tile's DMA lands in the other buffer while this one is multiplied
            float* nxt = smem + ((kt + 1) & 1) * STAGE_FL;
            ga += sa; gb += sb;
            LA::issue(ga, offa, nxt, wave);
            LB::issue(gb, offb, nxt + A_FL, wave);
        }
        const float* a_s = smem + (kt & 1) * STAGE_FL;
        const float* b_s = a_s + A_FL;
        mma_kstep<LA, LB, TM, TN>(a_s, b_s, wm * (BM / 2), wn * (BN / 2), fr, fq, acc);
        __syncthreads();
    }

    // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n_blk + wn * (BN / 2) + j * 32 + fr;
            const int row0 = m_blk + wm * (BM / 2) + i * 32 + 4 * fq;
            if constexpr (EDGE) { if (col >= p.N) continue; }
            auto row_ok = [&](int r) { return !EDGE || row0 + (r & 3) + 8 * (r >> 2) < p.M; };
            if constexpr (EPI == EPI_SLAB) {
                float* c = p.slab + ((size_t)blockIdx.z * p.M + row0) * p.N + col;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (row_ok(r)) c[(size_t)((r & 3) + 8 * (r >> 2)) * p.N] = acc[i][j][r];
            } else {
                float bv = 0.f;
                if constexpr (EPI != EPI_ATOMIC) bv = p.bias != nullptr ? p.bias[col] : 0.f;
                else bv = (p.bias != nullptr && blockIdx.z == 0) ? p.bias[col] : 0.f;
                float* c = p.C + (size_t)row0 * p.ldc + col;
                float mk[16];
                if constexpr (EPI == EPI_MASK) {
                    const float* ax = p.aux + (size_t)row0 * p.ldaux + col;
#pragma unroll
                    for (int r = 0; r < 16; ++r) mk[r] = row_ok(r) ? ax[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldaux] : 0.f;
                }
                if constexpr (EPI == EPI_ACC) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mk[r] = row_ok(r) ? c[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldc] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] + bv;
                    if constexpr (EPI == EPI_RELU) v = relu_nan(v);
                    if constexpr (EPI == EPI_MASK) v = mk[r] > 0.f ? v : 0.f;
                    if constexpr (EPI == EPI_ACC) v += mk[r];
                    float* cr = c + (size_t)((r & 3) + 8 * (r >> 2)) * p.ldc;
                    if (row_ok(r)) {
                        if constexpr (EPI == EPI_ATOMIC) atomicAdd(cr, v);
                        else *cr = v;
                    }
                }
            }
        }
    }
}

// C[m][n] (+)= bias[n] + sum_z slab[z][m][n]   (second stage of split-K: far cheaper than one fp32 atomic
// per partial element - an M x N x splits slab is a few MB, L2-resident)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, int M,
                                                            int N, int ldc, int splits, const float* __restrict__ bias,
                                                            int accumulate, float* __restrict__ C2 = nullptr, int ldc2 = 0,
                                                            int n_split = 0) {
    // 64 consecutive outputs per block, the splits dealt round-robin to 4 thread groups (a 128x128
    // product has only 16K outputs: one thread per output left most of the chip idle)
    __shared__ float sh[256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long mn = (long long)M * N;
    const long long idx = (long long)blockIdx.x * 64 + tx;
    float acc = 0.f;
    if (idx < mn) {
        int z = ty;
        for (; z + 12 < splits; z += 16) {
            const float a0 = slab[(size_t)z * mn + idx], a1 = slab[(size_t)(z + 4) * mn + idx];
            const float a2 = slab[(size_t)(z + 8) * mn + idx], a3 = slab[(size_t)(z + 12) * mn + idx];
            acc += (a0 + a1) + (a2 + a3);
        }
        for (; z < splits; z += 4) acc += slab[(size_t)z * mn + idx];
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (ty != 0 || idx >= mn) return;
    acc = (sh[tx] + sh[64 + tx]) + (sh[128 + tx] + sh[192 + tx]);
    const int m = (int)(idx / N), n = (int)(idx - (long long)m * N);
    if (bias) acc += bias[n];
    float* c = (n_split > 0 && n >= n_split) ? C2 + (size_t)m * ldc2 + (n - n_split) : C + (size_t)m * ldc + n;
    *c = accumulate ? *c + acc : acc;
}

// grouped form: group g (blockIdx.y) owns slabs [begin[g], begin[g+1]) and writes C + g*M*N (dense M x N)
struct SplitkGroups { int begin[9]; };
__global__ __launch_bounds__(256) void splitk_reduce_grouped_kernel(const float* __restrict__ slab, float* __restrict__ C,
                                                                    int mn, SplitkGroups gr) {
    __shared__ float sh[256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int g = blockIdx.y;
    const int idx = blockIdx.x * 64 + tx;
    const float* sl = slab + (size_t)gr.begin[g] * mn;
    const int splits = gr.begin[g + 1] - gr.begin[g];
    float acc = 0.f;
    if (idx < mn) {
        int z = ty;
        for (; z + 12 < splits; z += 16) {
            const float a0 = sl[(size_t)z * mn + idx], a1 = sl[(size_t)(z + 4) * mn + idx];
            const float a2 = sl[(size_t)(z + 8) * mn + idx], a3 = sl[(size_t)(z + 12) * mn + idx];
            acc += (a0 + a1) + (a2 + a3);
        }
        for (; z < splits; z += 4) acc += sl[(size_t)z * mn + idx];
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (ty != 0 || idx >= mn) return;
    C[(size_t)g * mn + idx] = (sh[tx] + sh[64 + tx]) + (sh[128 + tx] + sh[192 + tx]);
}

int splitk_reduce_grouped(const float* slab, float* C, int M, int N, int n_groups, const int* begin, hipStream_t s) {
    if (n_groups > 8) { set_error("splitk_reduce_grouped: more than 8 groups", 1050); return 1050; }
    SplitkGroups gr;
    for (int i = 0; i <= n_groups; ++i) gr.begin[i] = begin[i];
    const int mn = M * N;
    hipLaunchKernelGGL(splitk_reduce_grouped_kernel, dim3((unsigned)((mn + 63) / 64), n_groups), dim3(256), 0, s, slab, C, mn, gr);
    return launch_check("splitk_reduce_grouped");
}

int splitk_reduce(const float* slab, float* C, int M, int N, int ldc, int splits, hipStream_t s) {
    const long long mn = (long long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((mn + 63) / 64)), dim3(256), 0, s, slab, C, M, N, ldc, splits,
                       (const float*)nullptr, 0);
    return launch_check("splitk_reduce");
}

int splitk_reduce_pair(const float* slab, float* C, int M, int N, int ldc, int splits, int accumulate, float* C2, int ldc2,
                       int n_split, hipStream_t s) {
    const long long mn = (long long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((mn + 63) / 64)), dim3(256), 0, s, slab, C, M, N, ldc, splits,
                       (const float*)nullptr, accumulate, C2, ldc2, n_split);
    return launch_check("splitk_reduce_pair");
}

template <int BM, int BN, bool A_KM, bool B_KM>
static int launch_gemm(const GemmArgs& a, int splits, hipStream_t stream) {
    using LA = TileLoader<BM, A_KM>;
    using LB = TileLoader<BN, B_KM>;
    constexpr int A_FL = (LA::LDS_FLOATS + 3) / 4 * 4, B_FL = (LB::LDS_FLOATS + 3) / 4 * 4;
    const size_t lds = (size_t)(A_FL + B_FL) * 2 * sizeof(float);
    static bool attr_done = false;
    if (lds > 64 * 1024 && !attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_f32_kernel<BM, BN, A_KM, B_KM>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("gemm: hipFuncSetAttribute", (int)e); return (int)e; }
        attr_done = true;
    }
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, splits);
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, A_KM, B_KM>), grid, dim3(256), lds, stream, a);
    return launch_check("gemm_f32");
}

template <bool A_KM, bool B_KM>
static int dispatch_tile(const GemmArgs& a, int splits, hipStream_t stream) {
    // big tile when both dimensions can fill it and the grid still covers the chip
    const long tiles128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * splits;
    if (a.M >= 128 && a.N >= 128 && tiles128 >= 256) return launch_gemm<128, 128, A_KM, B_KM>(a, splits, stream);
    if (a.M >= 128 && a.N > 32) {
        const long tiles = (long)((a.M + 127) / 128) * ((a.N + 63) / 64) * splits;
        if (tiles >= 128) return launch_gemm<128, 64, A_KM, B_KM>(a, splits, stream);
    }
    return launch_gemm<64, 64, A_KM, B_KM>(a, splits, stream);
}


template <int BM, int BN, bool A_KM, bool B_KM, int EPI, bool EDGE>
static int launch_fast(const GemmArgs& a, int splits, hipStream_t stream) {
    constexpr int A_FL = FastTile<BM, A_KM>::LDS_FLOATS, B_FL = FastTile<BN, B_KM>::LDS_FLOATS;
    const size_t lds = (size_t)(A_FL + B_FL) * 2 * sizeof(float);
    static bool attr_done = false;
    if (lds > 48 * 1024 && !attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_fast_kernel<BM, BN, A_KM, B_KM, EPI, EDGE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("gemm: hipFuncSetAttribute", (int)e); return (int)e; }
        attr_done = true;
    }
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, splits);
    hipLaunchKernelGGL((gemm_fast_kernel<BM, BN, A_KM, B_KM, EPI, EDGE>), grid, dim3(256), lds, stream, a);
    return launch_check("gemm_f32(fast)");
}

template <int BM, int BN, bool A_KM, bool B_KM, bool EDGE>
static int fast_epi(const GemmArgs& a, int splits, hipStream_t stream) {
    if (a.slab != nullptr) return launch_fast<BM, BN, A_KM, B_KM, EPI_SLAB, EDGE>(a, splits, stream);
    if (a.atomic) return launch_fast<BM, BN, A_KM, B_KM, EPI_ATOMIC, EDGE>(a, splits, stream);
    if (a.aux != nullptr) return launch_fast<BM, BN, A_KM, B_KM, EPI_MASK, EDGE>(a, splits, stream);
    if (a.relu) return launch_fast<BM, BN, A_KM, B_KM, EPI_RELU, EDGE>(a, splits, stream);
    if (a.accumulate) return launch_fast<BM, BN, A_KM, B_KM, EPI_ACC, EDGE>(a, splits, stream);
    return launch_fast<BM, BN, A_KM, B_KM, EPI_PLAIN, EDGE>(a, splits, stream);
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// returns -1 when the shape does not qualify for the fast path
template <bool A_KM, bool B_KM>
static int dispatch_fast(const GemmArgs& a, int splits, hipStream_t stream) {
    if ((a.k_per_split % GEMM_BK) || (a.K % GEMM_BK) || (a.lda & 3) || (a.ldb & 3) || !aligned16(a.A) || !aligned16(a.B))
        return -1;
    if (a.relu && a.aux != nullptr) return -1;
    if ((a.M & 63) || (a.N & 63)) {
        // ragged M/N (the 154-wide head projections): clamped source rows + masked stores, 64x64 tiles.
        // k-major operands are read in 16-byte pieces up to their physical row width.
        if ((A_KM && a.lda < 4) || (B_KM && a.ldb < 4)) return -1;
        return fast_epi<64, 64, A_KM, B_KM, true>(a, splits, stream);
    }
    const bool m128 = (a.M % 128) == 0, n128 = (a.N % 128) == 0;
    // prefer the big tile while it still gives every CU two workgroups; otherwise shrink M first
    const long t_big = (long)(a.M / 128) * (a.N / 128) * splits;
    if (m128 && n128 && t_big >= 512) return fast_epi<128, 128, A_KM, B_KM, false>(a, splits, stream);
    if (n128 && (long)(a.M / 64) * (a.N / 128) * splits >= 384) return fast_epi<64, 128, A_KM, B_KM, false>(a, splits, stream);
    if (m128 && n128 && t_big >= 256) return fast_epi<128, 128, A_KM, B_KM, false>(a, splits, stream);
    if (n128) return fast_epi<64, 128, A_KM, B_KM, false>(a, splits, stream);
    return fast_epi<64, 64, A_KM, B_KM, false>(a, splits, stream);
}

// C[M,N] (op)= A*B.  splits <= 0 -> chosen automatically (split-K is used when the output tile
// grid alone cannot fill 256 CUs, i.e. for the weight-gradient products with K = rows).
int gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
             int a_kmajor, int b_kmajor, const float* bias, int relu, const float* aux, int ldaux,
             int accumulate, int splits, hipStream_t stream, GemmScratch sc) {
    if (M <= 0 || N <= 0) return 0;
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux = aux;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux;
    a.relu = relu; a.accumulate = accumulate; a.atomic = 0; a.slab = nullptr;
    a.B2 = nullptr; a.ldb2 = 0; a.n_split = 0;
    if (K <= 0) {
        set_error("gemm_f32: K <= 0", 1002);
        return 1002;
    }
    if (splits <= 0) {
        const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
        splits = 1;
        if (tiles < 256 && K >= 2048 && !relu && aux == nullptr) {
            // workgroups the dispatch below will launch per split (64 x 128 tiles when N allows, else 64 x 64)
            const bool wide = !(M & 63) && !(N & 127);
            const long wgs = (long)((M + 63) / 64) * (wide ? N / 128 : (N + 63) / 64);
            // one workgroup per CU in a single round is the sweet spot (tools/gemm_split_sweep.py: dW_pre 256 x 896 takes
            // 80 us at 224-252 or 448-504 workgroups, 112 us at 280 - a second, nearly empty round); two per CU when one
            // round would leave a quarter of the chip idle
            long want = 256 / wgs;
            if (want < 1) want = 1;
            if (wgs * want < 192) want = 512 / wgs;
            long maxs = K / 512;
            splits = (int)(want < maxs ? want : maxs);
            if (splits < 1) splits = 1;
        }
    }
    int kper = (K + splits - 1) / splits;
    kper = (kper + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
    splits = (K + kper - 1) / kper;
    a.k_per_split = kper;
    if (splits > 1) {
        if (relu || aux != nullptr) {
            set_error("gemm_f32: relu/mask epilogue is incompatible with split-K", 1003);
            return 1003;
        }
        if (sc.p != nullptr && (long long)splits * M * N <= sc.floats) {
            a.slab = sc.p;
        } else {
            if (!accumulate) {
                // split-K accumulates with atomics: start from zero
                if (int rc0 = zero2d_f32_async(C, ldc, N, M, stream)) return rc0;
            }
            a.atomic = 1;
            a.accumulate = 0;
        }
    }
    ProfScope prof(a_kmajor ? "gemm_f32_dW(TN,split-K)" : (b_kmajor ? "gemm_f32_dX(NN)" : "gemm_f32_fwd(NT)"),
                   2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), stream);
    int rc;
    if (!a_kmajor && !b_kmajor) rc = dispatch_fast<false, false>(a, splits, stream);
    else if (!a_kmajor && b_kmajor) rc = dispatch_fast<false, true>(a, splits, stream);
    else if (a_kmajor && !b_kmajor) rc = dispatch_fast<true, false>(a, splits, stream);
    else rc = dispatch_fast<true, true>(a, splits, stream);
    if (rc == -1) {
        if (!a_kmajor && !b_kmajor) rc = dispatch_tile<false, false>(a, splits, stream);
        else if (!a_kmajor && b_kmajor) rc = dispatch_tile<false, true>(a, splits, stream);
        else if (a_kmajor && !b_kmajor) rc = dispatch_tile<true, false>(a, splits, stream);
        else rc = dispatch_tile<true, true>(a, splits, stream);
    }
    if (rc == 0 && a.slab != nullptr) {
        const long long mn = (long long)M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((mn + 63) / 64)), dim3(256), 0, stream, a.slab, C, M, N,
                           ldc, splits, bias, accumulate);
        rc = launch_check("splitk_reduce");
    }
    return rc;
}

// [C1 | C2] = A^T [B1 | B2] in ONE split-K launch: A [K][M] k-major (lda), B1 [K][N1], B2 [K][N2] k-major, C1 [M][N1],
// C2 [M][N2] accumulated into (like the callers' gemm_f32(.., accumulate = 1)).  The two weight-gradient products of a recurrent layer (dW_ih = dgates^T x, dW_hh = dgates^T h_prev)
// share their A operand; as one product they are one launch of ~256 workgroups and one reduction instead of two each.
// Falls back to two gemm_f32 calls when the shapes do not fit (N1, N2 multiples of 128, M of 64, scratch large enough).
int gemm_f32_tn_pair(const float* A, int lda, const float* B1, int ldb1, int N1, const float* B2, int ldb2, int N2, float* C1,
                     int ldc1, float* C2, int ldc2, int M, int K, hipStream_t stream, GemmScratch sc) {
    const int N = N1 + N2;
    const long wgs = (long)(M / 64) * (N / 128);
    long want = wgs > 0 ? 256 / wgs : 1;
    if (want < 1) want = 1;
    if (wgs * want < 192 && wgs > 0) want = 512 / wgs;
    const long maxs = K / 512;
    int splits = (int)(want < maxs ? want : maxs);
    bool ok = (M % 64) == 0 && (N1 % 128) == 0 && (N2 % 128) == 0 && (K % GEMM_BK) == 0 && splits >= 2 && sc.p != nullptr &&
              !(lda & 3) && !(ldb1 & 3) && !(ldb2 & 3) && aligned16(A) && aligned16(B1) && aligned16(B2);
    int kper = 0;
    if (ok) {
        kper = (K + splits - 1) / splits;
        kper = (kper + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
        splits = (K + kper - 1) / kper;
        ok = (long long)splits * M * N <= sc.floats;
    }
    if (!ok) {
        if (int e = gemm_f32(A, B1, C1, M, N1, K, lda, ldb1, ldc1, 1, 1, nullptr, 0, nullptr, 0, 1, 0, stream, sc)) return e;
        return gemm_f32(A, B2, C2, M, N2, K, lda, ldb2, ldc2, 1, 1, nullptr, 0, nullptr, 0, 1, 0, stream, sc);
    }
    GemmArgs a;
    a.A = A; a.B = B1; a.C = C1; a.bias = nullptr; a.aux = nullptr;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb1; a.ldc = ldc1; a.ldaux = 0;
    a.relu = 0; a.accumulate = 0; a.atomic = 0; a.k_per_split = kper; a.slab = sc.p;
    a.B2 = B2; a.ldb2 = ldb2; a.n_split = N1;
    ProfScope prof("gemm_f32_dW(TN,split-K)", 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), stream);
    if (int e = launch_fast<64, 128, true, true, EPI_SLAB, false>(a, splits, stream)) return e;
    const long long mn = (long long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((mn + 63) / 64)), dim3(256), 0, stream, a.slab, C1, M, N, ldc1, splits,
                       (const float*)nullptr, 1, C2, ldc2, N1);
    return launch_check("gemm_f32_tn_pair");
}

}  // namespace dc
