// Backward of the per-unit embedding MLP for the two 16-unit types (allied / enemy non-heroes = 32 of the 40
// units), exploiting what the max-pool leaves of the gradient.
//
// Replaces, for these types, the part of /root/reference/optimizer.py:672 (autograd) that flows through
// policy.py:102-136: d(emb) of an env-step has, per (type, channel), exactly ONE non-zero among the 16 units -
// the arg-max unit gets d(xcat) (torch.max backward) - plus, on the steps whose target-unit head is live, the
// rank-one attention term dtu[u] * q (policy.py:152).  The dense formulation (embed_scatter_bwd writes the
// 16 x 128 block, embed_bwd_dw2 / embed_bwd_dw1 multiply it by basic^T and W2 on the matrix cores) does
// 16 x 128 x 128 MACs per step, type and product; the products actually needed are
//     dW2[c][k]   += d[c] * basic[a(c)][k]                       (a(c) = arg-max unit of channel c)
//     dbasic[u][k] = sum over {c : a(c) = u} of d[c] * W2[c][k]
// = 128 x 128 MACs each, a sixteenth.  They are gathers scaled per channel - no matrix-product structure, so
// they run on the packed-f32 VALU with W2_t stationary in LDS; d(emb) of these types is never written to or
// read from HBM (2 x 268 MB per pass at the bench batch), and d(basic) exists only in registers.
//
// Two passes: embed_pool16_prepare_kernel sorts every step's channels by arg-max unit (wave ballots; ascending
// channel inside a unit: deterministic sums); embed_bwd_pool16_kernel (one workgroup = one type x a contiguous range of
// env-steps) keeps its accumulators in registers for the whole range.  Two lane <-> data maps, each where it saves
// instructions: the dW2 update runs CHANNEL-per-lane (lane l owns channels l and l + 64, wave w the k range
// [16w, 16w + 16): the scale d[c] and the source row a(c) are lane-local - no broadcast, one 8-byte read for both - and a
// lane gathers its 64 bytes of basic[a(c)] with four ds_read_b128, rows 528 bytes apart so that sixteen different rows
// at one column never share a bank); d(basic) runs K-per-lane (wave w owns units w, w + 8, lane l the k pair 2l, 2l + 1:
// the rows of W2 a unit sums over are wave-uniform and read as 512 contiguous bytes).  Steps with a live target-unit head additionally need R[k] = sum_c q[c] W2[c][k] and
// s[k] = sum_u dtu[u] basic[u][k] (two workgroup reductions through LDS) for the rank-one terms.
// Outputs are per-workgroup partials in the formats the dense path already reduces:
//   slab[wg][128][128] (splitk_reduce_grouped), part1[wg][13][128] (unit_basic_reduce), part2[wg][128] (colsum).
#include <stdio.h>
#include <stdlib.h>
#include <utility>
#include "kernels.h"

namespace dc {
namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;
enum { SP_THREADS = 512, SP_LD = 128, SP_BLD = 132, SP_PREP = 320 };   // SP_BLD: floats per LDS row of `basic` (4 banks apart)
enum { SP_OLD_ = 0 };   // SP_PREP: floats per (step, type) of the prepared channel lists   // padded LDS rows: consecutive rows start 4 banks apart, so 16 lanes reading
                                         // 16 bytes at one column offset of 16 different rows never share a bank
constexpr int SP_OBS = 483, SP_XCAT = 896;

struct SparseArgs {
    const float* obs; const float* dxcat; const uint8_t* amax; const float* dtu; const float* q; int ldq;
    const float* W1; const float* b1; const float* W2;
    float* slab; float* part1; float* part2;
    long long nr; int wg_per_type; int steps_per_wg;
    long long* dbg;
    float* prep;      // [2][nr][SP_PREP]: per step and type {d, W2-row byte offset} sorted by arg-max unit [144] | {first, count} [16]
};

// LDS carve-up (floats).  A workgroup runs NS = 2 independent step streams in lock step (one barrier per iteration):
// twice the independent work per wave between barriers - the per-row chains (broadcast -> address -> LDS -> FMA) are
// latency-bound with two waves per SIMD, so the second stream is nearly free.
enum {
    NS = 2,
    L_W2 = 0,                               // [128][128] W2_t rows
    L_BAS = L_W2 + 128 * SP_LD,             // 2 x NS x [16][SP_BLD] basic of a step (written one iteration ahead)
    L_RED = L_BAS + 2 * NS * 16 * SP_BLD,   // NS x [2][8][128] per-wave partials of R and s
    L_STG = L_RED + NS * 2 * 8 * 128,       // 3 x NS staging blocks: the inputs of a step
    STG_Q = 0,                         //   q[128]
    STG_PB = 128,                      //   per channel {d, byte offset of basic row a(c) (SP_BLD rows)}  [128] x 8 B
    STG_LIST = 384,                    //   channels sorted by arg-max unit: {d, byte offset of W2 row c}, [128 + 16] x 8 B
    STG_SC = 672,                      //   per unit {first list entry, number of channels}          [16] x 8 B
    STG_DT = 704,                      //   dtu[16], [16] = their sum
    STG_FLAG = 736,                    //   1 if any dtu != 0
    STG_X = 752,                       //   unit records [16][12]
    STG_SIZE = 944,
    L_RS = L_STG + 3 * NS * STG_SIZE,       // NS x (R[128] | s[128])
    L_W1 = L_RS + NS * 256,                 // [128][12] W1
    L_TOTAL = L_W1 + 128 * 12
};
enum { SP_SLOTS = 12 };   // channels of a unit handled in straight-line code (a unit holds 8 on average); the rest in a loop

__device__ __forceinline__ f32x2 mk2(float a, float b) { f32x2 r; r.x = a; r.y = b; return r; }

// entry J (0..15) of a gather16 vector in every lane: each 16-lane DPP row holds the sixteen entries, row_newbcast:J
// copies lane J of every row to the whole row - one VALU instruction (foldable into its consumer), no SGPR round trip
template <int J>
__device__ __forceinline__ int bcast16_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xf, 0xf, true); }
template <int J>
__device__ __forceinline__ float bcast16_f(float v) { return __int_as_float(bcast16_i<J>(__float_as_int(v))); }

template <class F, int... Js>
__device__ __forceinline__ void for_consts(F&& f, std::integer_sequence<int, Js...>) { (f(std::integral_constant<int, Js>{}), ...); }

// acc += row * d with d wave-uniform: the scalar goes in as an SGPR pair (d, 0) whose low dword feeds both halves
// (op_sel_hi) - no v_mov pair per row to splat it into vector registers
__device__ __forceinline__ void pk_fma_s(f32x2& acc, f32x2 row, float d_uniform) {
    const unsigned long long dd = (unsigned long long)__float_as_uint(d_uniform);
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(row), "s"(dd));
}

}  // namespace

// Pass 1: the channel list of every (env-step, type), sorted by arg-max unit.  One wave per (step, type); lane l owns
// channels l and l + 64.  Per unit: two ballots, popcounts (count, running first position), v_mbcnt ranks - ascending
// channel inside a unit, so the sums of pass 2 are deterministic.
__global__ __launch_bounds__(256) void embed_pool16_prepare_kernel(SparseArgs p) {
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= 2 * p.nr) return;
    const int t = item < p.nr ? 2 : 3;
    const long long n = item < p.nr ? item : item - p.nr;
    const float* dx = p.dxcat + n * SP_XCAT;
    float d[2];
    int a[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = lane + 64 * h;
        d[h] = t == 2 ? dx[3 * 128 + c] : dx[4 * 128 + c] + dx[6 * 128 + c];   // policy.py:127: enh feeds two slots
        a[h] = p.amax[(n * 3 + (t - 1)) * 128 + c];
    }
    int pos[2] = {0, 0};
    int start = 0, my_start = 0, my_cnt = 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const unsigned long long m0 = __ballot(a[0] == u), m1 = __ballot(a[1] == u);
        const int c0 = __builtin_popcountll(m0), c1 = __builtin_popcountll(m1);
        const int r0 = __builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0));
        const int r1 = __builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0));
        pos[0] = a[0] == u ? start + r0 : pos[0];
        pos[1] = a[1] == u ? start + c0 + r1 : pos[1];
        my_start = lane == u ? start : my_start;
        my_cnt = lane == u ? c0 + c1 : my_cnt;
        start += c0 + c1;
    }
    float* out = p.prep + (size_t)item * SP_PREP;
#pragma unroll
    for (int h = 0; h < 2; ++h)
        *reinterpret_cast<float2*>(out + 2 * pos[h]) = make_float2(d[h], __int_as_float((lane + 64 * h) * SP_LD * 4));
    if (lane < 16) {
        // pass 2 reads SP_SLOTS entries from a unit's first one whatever its count: 16 zero-valued entries of row 0 behind
        *reinterpret_cast<float2*>(out + 2 * (128 + lane)) = make_float2(0.f, __int_as_float(0));
        *reinterpret_cast<int2*>(out + 288 + 2 * lane) = make_int2(my_start, my_cnt);
    }
}

// Pass 2.  Workgroup = one type x a contiguous range of env-steps, split into two streams that advance one step per
// iteration each (one barrier per iteration).
// Wave w owns units w and w + 8 and the dW2 rows of channels w + 8i (i < 16); lane l owns k = 2l, 2l + 1 - so every
// row of W2 / basic a wave touches is read by its 64 lanes as 512 contiguous bytes, and WHICH row is wave-uniform: the
// {value, row offset} entries are fetched sixteen at a time (one 8-byte LDS read by sixteen lanes) and broadcast with
// v_readlane into scalar registers.  One LDS round trip per row instead of two dependent ones, no divergent loops.
//   iteration i: every thread hands its prefetched values of iteration i+1 to staging[(i+1) % 3] and issues the loads of
//   iteration i+2  -- barrier --  phase A of iteration i+1 (basic -> bas[(i+1) & 1]), phases B, C, (live), D of iteration i.
template <bool TIMING>   // TIMING (DC_SP_TIMING=1): s_memtime phase sums of every wave of workgroup 0 -> p.dbg[8][6]
__global__ __launch_bounds__(SP_THREADS) void embed_bwd_pool16_kernel(SparseArgs p) {
    long long tm[6] = {0, 0, 0, 0, 0, 0}, tm0 = 0;
    auto stamp = [&](int i) {
        if constexpr (TIMING) { const long long x = __builtin_amdgcn_s_memtime(); tm[i] += x - tm0; tm0 = x; }
    };
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k0 = 2 * lane;
    const int t = 2 + blockIdx.x / p.wg_per_type;          // 2 = allied non-heroes, 3 = enemy non-heroes
    const int wgi = blockIdx.x % p.wg_per_type;
    const long long n0 = (long long)wgi * p.steps_per_wg;
    const long long n1 = min(p.nr, n0 + p.steps_per_wg);
    const int cum = t == 2 ? 6 : 22;                        // first unit of the type inside the 40
    const float* prep_t = p.prep + (size_t)(t - 2) * p.nr * SP_PREP;

    // ---- stationary operands ---------------------------------------------------------------------------
    {
        const float4* src = reinterpret_cast<const float4*>(p.W2 + (size_t)t * 128 * 128);
        for (int e = tid; e < 128 * 32; e += SP_THREADS) *reinterpret_cast<float4*>(smem + L_W2 + 4 * e) = src[e];
    }
    for (int e = tid; e < 128 * 12; e += SP_THREADS) smem[L_W1 + e] = p.W1[e];
    const f32x2 b1r = mk2(p.b1[k0], p.b1[k0 + 1]);
    f32x2 D[2][8];                 // dW2[c][16w + 2j, 16w + 2j + 1] of channels c = lane (D[0]) and lane + 64 (D[1])
    float dW1a[2][12], db1a[2];
    float db2a = 0.f;              // threads 0..127: second-layer bias gradient of channel tid
#pragma unroll
    for (int i = 0; i < 8; ++i) { D[0][i] = mk2(0.f, 0.f); D[1][i] = mk2(0.f, 0.f); }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        db1a[e] = 0.f;
#pragma unroll
        for (int f = 0; f < 12; ++f) dW1a[e][f] = 0.f;
    }

    // ---- two step streams: stream s covers steps [nb[s], ne[s]); iteration i works on step nb[s] + i of both ----------
    const long long half = (n1 - n0 + 1) / 2;
    const long long nb[NS] = {n0, n0 + half}, ne[NS] = {min(n1, n0 + half), n1};
    const long long iters = half;

    // ---- loader roles: tid < 128 channel tid (d, q, arg-max) | 128..143 dtu | 144..335 record floats | 336..495 prepared list
    float st_0[NS] = {0.f, 0.f}, st_1[NS] = {0.f, 0.f};
    int st_a[NS] = {0, 0};
    auto stage_load = [&](long long i) {
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            const long long n = nb[s2] + i;
            if (n >= ne[s2]) continue;
            if (tid < 128) {
                const float* dx = p.dxcat + n * SP_XCAT;
                st_0[s2] = t == 2 ? dx[3 * 128 + tid] : dx[4 * 128 + tid] + dx[6 * 128 + tid];   // policy.py:127: enh feeds two slots
                st_1[s2] = p.q[n * p.ldq + tid];
                st_a[s2] = p.amax[(n * 3 + (t - 1)) * 128 + tid];
            } else if (tid < 144) {
                st_0[s2] = p.dtu[n * 40 + cum + (tid - 128)];
            } else if (tid < 336) {
                st_0[s2] = p.obs[n * SP_OBS + 3 + cum * 12 + (tid - 144)];
            } else if (tid < 496) {
                const float2 v = *reinterpret_cast<const float2*>(prep_t + (size_t)n * SP_PREP + 2 * (tid - 336));
                st_0[s2] = v.x; st_1[s2] = v.y;
            }
        }
    };
    auto stg_of = [&](long long i, int s2) { return smem + L_STG + ((int)(i % 3) * NS + s2) * STG_SIZE; };
    auto hand_over = [&](long long i) {           // registers -> staging[i % 3][stream]
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            if (nb[s2] + i >= ne[s2]) continue;
            float* stg = stg_of(i, s2);
            if (tid < 128) {
                stg[STG_Q + tid] = st_1[s2];
                *reinterpret_cast<float2*>(stg + STG_PB + 2 * tid) = make_float2(st_0[s2], __int_as_float(st_a[s2] * SP_BLD * 4));
            } else if (tid < 144) {        // lanes 0..15 of wave 2
                stg[STG_DT + (tid - 128)] = st_0[s2];
                float sum = st_0[s2];          // the sixteen lanes are one DPP row: rotate-and-add, no LDS round trips
                sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x128, 0xf, 0xf, true));
                sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x124, 0xf, 0xf, true));
                sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x122, 0xf, 0xf, true));
                sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x121, 0xf, 0xf, true));
                const unsigned long long nz = __ballot(st_0[s2] != 0.f) & 0xffffull;
                if (tid == 128) { stg[STG_DT + 16] = sum; reinterpret_cast<int*>(stg)[STG_FLAG] = nz != 0ull; }
            } else if (tid < 336) {
                stg[STG_X + (tid - 144)] = st_0[s2];
            } else if (tid < 496) {        // LIST [288] and SC [32] are contiguous in the prepared block and in staging
                *reinterpret_cast<float2*>(stg + STG_LIST + 2 * (tid - 336)) = make_float2(st_0[s2], st_1[s2]);
            }
        }
    };
    // sixteen {value, offset} entries starting at `first`, stride `stride` entries -> lanes 0..15 of (val, off)
    auto gather16 = [&](const float* base, int first, int stride, float& val, int& off) {
        const float2 e = *reinterpret_cast<const float2*>(base + 2 * (first + stride * (lane & 15)));
        val = e.x;
        off = __float_as_int(e.y);
    };
    auto bas_of = [&](long long i, int s2) { return smem + L_BAS + ((int)(i & 1) * NS + s2) * 16 * SP_BLD; };
    auto phase_a = [&](long long i) {             // basic[u][k0..k0+1] of iteration i's steps, u = w and w + 8
        // W1 rows k0, k0 + 1 (24 floats); the k-ordered fmaf chain of the MFMA-generated first layer (embed_fused.hip),
        // bias last: bitwise the forward's value, hence its relu mask
        const float4* wp = reinterpret_cast<const float4*>(smem + L_W1 + k0 * 12);
        float w1[24];
#pragma unroll
        for (int v = 0; v < 6; ++v) { const float4 q4 = wp[v]; w1[4 * v] = q4.x; w1[4 * v + 1] = q4.y; w1[4 * v + 2] = q4.z; w1[4 * v + 3] = q4.w; }
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            if (nb[s2] + i >= ne[s2]) continue;
            const float* stg = stg_of(i, s2);
            float* bas = bas_of(i, s2);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int u = w + 8 * h;
                const float4* xp = reinterpret_cast<const float4*>(stg + STG_X + u * 12);     // wave-uniform address
                const float4 xa = xp[0], xb = xp[1], xc = xp[2];
                const float x[12] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w, xc.x, xc.y, xc.z, xc.w};
                float a0 = x[0] * w1[0], a1 = x[0] * w1[12];
#pragma unroll
                for (int f = 1; f < 12; ++f) { a0 = fmaf(x[f], w1[f], a0); a1 = fmaf(x[f], w1[12 + f], a1); }
                *reinterpret_cast<float2*>(bas + u * SP_BLD + k0) = make_float2(fmaxf(a0 + b1r.x, 0.f), fmaxf(a1 + b1r.y, 0.f));
            }
        }
    };

    stage_load(0);
    hand_over(0);
    stage_load(1);
    __syncthreads();          // W2 / W1 / staging[0]
    phase_a(0);

    const char* w2b = reinterpret_cast<const char*>(smem + L_W2 + k0);     // + W2-row byte offset
    for (long long i = 0; i < iters; ++i) {
        if constexpr (TIMING) tm0 = __builtin_amdgcn_s_memtime();
        hand_over(i + 1);
        stage_load(i + 2);
        stamp(0);
        __syncthreads();
        stamp(1);
        phase_a(i + 1);
        const float* stg[NS];
        const char* basb[NS];
        bool on[NS], live[NS];
        f32x2 basic[NS][2], db[NS][2];
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            on[s2] = nb[s2] + i < ne[s2];                                         // workgroup-uniform
            stg[s2] = stg_of(i, s2);
            basb[s2] = reinterpret_cast<const char*>(bas_of(i, s2) + k0);
            live[s2] = on[s2] && reinterpret_cast<const int*>(stg[s2])[STG_FLAG] != 0;
            db[s2][0] = mk2(0.f, 0.f); db[s2][1] = mk2(0.f, 0.f);
            basic[s2][0] = mk2(0.f, 0.f); basic[s2][1] = mk2(0.f, 0.f);
        }
        // ---- phase B: dW2[c][k] += d[c] * basic[a(c)][k], c = w + 8 i;  phase C: dbasic[u][k] = sum over the unit's
        // channels of d[c] * W2[c][k], u = w, w + 8 - for both streams
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            if (!on[s2]) continue;
#pragma unroll
            for (int h = 0; h < 2; ++h) basic[s2][h] = *reinterpret_cast<const f32x2*>(basb[s2] + (w + 8 * h) * SP_BLD * 4);
            {   // phase B, channel per lane: D[h][j] += d[c] * basic[a(c)][16w + 2j .. +1], c = lane + 64 h
                const char* brow = reinterpret_cast<const char*>(bas_of(i, s2)) + w * 64;       // k range of this wave
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float2 e = *reinterpret_cast<const float2*>(stg[s2] + STG_PB + 2 * (lane + 64 * h));   // {d, row byte offset}
                    const float4* rp = reinterpret_cast<const float4*>(brow + __float_as_int(e.y));
                    const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
                    const f32x2 dd = mk2(e.x, e.x);
                    D[h][0] = __builtin_elementwise_fma(dd, mk2(r0.x, r0.y), D[h][0]);
                    D[h][1] = __builtin_elementwise_fma(dd, mk2(r0.z, r0.w), D[h][1]);
                    D[h][2] = __builtin_elementwise_fma(dd, mk2(r1.x, r1.y), D[h][2]);
                    D[h][3] = __builtin_elementwise_fma(dd, mk2(r1.z, r1.w), D[h][3]);
                    D[h][4] = __builtin_elementwise_fma(dd, mk2(r2.x, r2.y), D[h][4]);
                    D[h][5] = __builtin_elementwise_fma(dd, mk2(r2.z, r2.w), D[h][5]);
                    D[h][6] = __builtin_elementwise_fma(dd, mk2(r3.x, r3.y), D[h][6]);
                    D[h][7] = __builtin_elementwise_fma(dd, mk2(r3.z, r3.w), D[h][7]);
                }
            }
            float scv; int scc;
            gather16(stg[s2] + STG_SC, 0, 1, scv, scc);          // lane u: {first entry, count} of unit u
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int first = __builtin_amdgcn_readlane(__float_as_int(scv), w + 8 * h);
                const int cnt = __builtin_amdgcn_readlane(scc, w + 8 * h);
                float dvec; int ovec;
                gather16(stg[s2] + STG_LIST, first, 1, dvec, ovec);
                dvec = (lane & 15) < cnt ? dvec : 0.f;        // past the unit's segment: the next unit's entries, weight 0
                f32x2& dbh = db[s2][h];
                f32x2 dbo = mk2(0.f, 0.f);       // odd slots: a second chain (a unit's 8-12 dependent packed FMAs are the
                                                 // longest serial chain of the step; order of the sum stays fixed)
                auto slot = [&](auto JC) {
                    constexpr int J = decltype(JC)::value;
                    const f32x2 wv = *reinterpret_cast<const f32x2*>(w2b + bcast16_i<J>(ovec));
                    const float d = bcast16_f<J>(dvec);
                    f32x2& acc = (J & 1) ? dbo : dbh;
                    acc.x = fmaf(d, wv.x, acc.x);
                    acc.y = fmaf(d, wv.y, acc.y);
                };
                for_consts(slot, std::integer_sequence<int, 0, 1, 2, 3, 4, 5, 6, 7>{});         // a unit holds 8 channels on average
                if (cnt > 8) for_consts(slot, std::integer_sequence<int, 8, 9, 10, 11>{});      // wave-uniform
                for (int j = SP_SLOTS; j < cnt; ++j) {        // wave-uniform trip count
                    const float2 e = *reinterpret_cast<const float2*>(stg[s2] + STG_LIST + 2 * (first + j));
                    const f32x2 wv = *reinterpret_cast<const f32x2*>(w2b + __float_as_int(e.y));
                    db[s2][h] = __builtin_elementwise_fma(mk2(e.x, e.x), wv, db[s2][h]);
                }
                db[s2][h] += dbo;
            }
        }
        stamp(2);
        stamp(3);
        if (live[0] || live[1]) {
            // rank-one attention terms: d(emb)[u][c] += dtu[u] q[c]
            //   dbasic[u][k] += dtu[u] * R[k],  R[k] = sum_c q[c] W2[c][k]
            //   dW2[c][k]    += q[c] * s[k],    s[k] = sum_u dtu[u] basic[u][k]
            float dt0[NS] = {0.f, 0.f}, dt1[NS] = {0.f, 0.f}, qvec[NS] = {0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                if (!live[s2]) continue;
                dt0[s2] = stg[s2][STG_DT + w]; dt1[s2] = stg[s2][STG_DT + w + 8];
                qvec[s2] = stg[s2][STG_Q + w + 8 * (lane & 15)];       // lane c: q of channel w + 8 c
                float* red = smem + L_RED + s2 * 2048;
                f32x2 r = mk2(0.f, 0.f);
                const float qv16 = qvec[s2];
                for_consts([&](auto CC) {
                    constexpr int C = decltype(CC)::value;
                    const float qc = bcast16_f<C>(qv16);
                    const f32x2 wv = *reinterpret_cast<const f32x2*>(w2b + (w + 8 * C) * SP_LD * 4);
                    r.x = fmaf(qc, wv.x, r.x);
                    r.y = fmaf(qc, wv.y, r.y);
                }, std::make_integer_sequence<int, 16>{});
                *reinterpret_cast<f32x2*>(red + w * 128 + k0) = r;
                *reinterpret_cast<f32x2*>(red + 1024 + w * 128 + k0) =
                    mk2(dt0[s2], dt0[s2]) * basic[s2][0] + mk2(dt1[s2], dt1[s2]) * basic[s2][1];
            }
            __syncthreads();
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                if (!live[s2]) continue;
                if (tid < 256) {     // R[k] (tid < 128) and s[k] (128 <= tid < 256): sums over the 8 waves, fixed order
                    const float* src = smem + L_RED + s2 * 2048 + (tid >> 7) * 1024 + (tid & 127);
                    float acc = 0.f;
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += src[u * 128];
                    smem[L_RS + s2 * 256 + tid] = acc;
                }
            }
            __syncthreads();
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                if (!live[s2]) continue;
                const f32x2 R = *reinterpret_cast<const f32x2*>(smem + L_RS + s2 * 256 + k0);
                db[s2][0] = __builtin_elementwise_fma(mk2(dt0[s2], dt0[s2]), R, db[s2][0]);
                db[s2][1] = __builtin_elementwise_fma(mk2(dt1[s2], dt1[s2]), R, db[s2][1]);
                // dW2[c][k] += q[c] * s[k] in the channel-per-lane layout: q lane-local, s[16w .. 16w + 15] wave-uniform
                const float4* sp = reinterpret_cast<const float4*>(smem + L_RS + s2 * 256 + 128 + 16 * w);
                const float4 s0 = sp[0], s1 = sp[1], s2v = sp[2], s3 = sp[3];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float qc = stg[s2][STG_Q + lane + 64 * h];
                    const f32x2 qq = mk2(qc, qc);
                    D[h][0] = __builtin_elementwise_fma(qq, mk2(s0.x, s0.y), D[h][0]);
                    D[h][1] = __builtin_elementwise_fma(qq, mk2(s0.z, s0.w), D[h][1]);
                    D[h][2] = __builtin_elementwise_fma(qq, mk2(s1.x, s1.y), D[h][2]);
                    D[h][3] = __builtin_elementwise_fma(qq, mk2(s1.z, s1.w), D[h][3]);
                    D[h][4] = __builtin_elementwise_fma(qq, mk2(s2v.x, s2v.y), D[h][4]);
                    D[h][5] = __builtin_elementwise_fma(qq, mk2(s2v.z, s2v.w), D[h][5]);
                    D[h][6] = __builtin_elementwise_fma(qq, mk2(s3.x, s3.y), D[h][6]);
                    D[h][7] = __builtin_elementwise_fma(qq, mk2(s3.z, s3.w), D[h][7]);
                }
            }
        }
        stamp(4);
        // ---- phase D: through the relu into dW1 / db1; second-layer bias gradient -----------------------
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            if (!on[s2]) continue;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4* xp = reinterpret_cast<const float4*>(stg[s2] + STG_X + (w + 8 * h) * 12);
                const float4 xa = xp[0], xb = xp[1], xc = xp[2];
                const float x[12] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w, xc.x, xc.y, xc.z, xc.w};
                const float dbm[2] = {basic[s2][h].x > 0.f ? db[s2][h].x : 0.f, basic[s2][h].y > 0.f ? db[s2][h].y : 0.f};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
#pragma unroll
                    for (int f = 0; f < 12; ++f) dW1a[e][f] = fmaf(dbm[e], x[f], dW1a[e][f]);
                    db1a[e] += dbm[e];
                }
            }
            if (tid < 128) db2a += stg[s2][STG_PB + 2 * tid] + (live[s2] ? stg[s2][STG_Q + tid] * stg[s2][STG_DT + 16] : 0.f);   // column sum of d(emb)
        }
        stamp(5);
    }

    if constexpr (TIMING) {
        if (blockIdx.x == 0 && (tid & 63) == 0) for (int i = 0; i < 6; ++i) p.dbg[(tid >> 6) * 6 + i] = tm[i];   // every wave's sums
    }
    // ---- results --------------------------------------------------------------------------------------
    {
        float* out = p.slab + (size_t)blockIdx.x * 128 * 128;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(out + (size_t)(lane + 64 * h) * 128 + 16 * w + 4 * j) =
                    make_float4(D[h][2 * j].x, D[h][2 * j].y, D[h][2 * j + 1].x, D[h][2 * j + 1].y);
        if (tid < 128) p.part2[(size_t)blockIdx.x * 128 + tid] = db2a;
    }
    // dW1 / db1: sum the 8 waves in fixed order through LDS -> part1[wg][f][k] (f = 12: db1)
    __syncthreads();
    float* acc = smem + L_RED;      // [13][128]
    for (int e = tid; e < 13 * 128; e += SP_THREADS) acc[e] = 0.f;
    __syncthreads();
    for (int ww = 0; ww < 8; ++ww) {
        if (w == ww) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int f = 0; f < 12; ++f) acc[f * 128 + k0 + e] += dW1a[e][f];
                acc[12 * 128 + k0 + e] += db1a[e];
            }
        }
        __syncthreads();
    }
    float* o = p.part1 + (size_t)blockIdx.x * 1664;
    for (int e = tid; e < 1664; e += SP_THREADS) o[e] = acc[e];
}

// slab: 2 * wg_per_type x [128][128]; part1: 2 * wg_per_type x [13][128]; part2: 2 * wg_per_type x [128];
// prep: 2 * nr * 320 floats of scratch for the prepared channel lists
int embed_bwd_pool16(const float* obs, const float* dxcat, const uint8_t* amax, const float* dtu, const float* q, int ldq,
                     const float* W1, const float* b1, const float* W2, float* slab, float* part1, float* part2, float* prep,
                     long long nr, int wg_per_type, hipStream_t s) {
    SparseArgs a{obs, dxcat, amax, dtu, q, ldq, W1, b1, W2, slab, part1, part2, nr, wg_per_type,
                 (int)((nr + wg_per_type - 1) / wg_per_type), nullptr, prep};
    const size_t lds = (size_t)L_TOTAL * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)embed_bwd_pool16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)embed_bwd_pool16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("embed_bwd_pool16: hipFuncSetAttribute", (int)e); return (int)e; }
        attr = true;
    }
    hipLaunchKernelGGL(embed_pool16_prepare_kernel, dim3((unsigned)((2 * nr + 3) / 4)), dim3(256), 0, s, a);
    if (int e = launch_check("embed_pool16_prepare")) return e;
    // algorithmic work: basic + dW1 fold 2 x 16 x 128 x 12 MACs, the two gathers 2 x 128 x 128 MACs per step and type
    ProfScope prof("embed_bwd_pool16", 2.0 * 2.0 * nr * (2.0 * 16 * 128 * 12 + 2.0 * 128 * 128),
                   4.0 * 2.0 * nr * (16 * 12 + 3 * 128 + 16 + 32), s);
    constexpr bool timing = DC_DEV_TIMING != 0;
    if (timing) {   // debugging aid: per-step phase cycles of one wave, printed per launch
        static long long* dbg = nullptr;
        if (!dbg) (void)hipMalloc(&dbg, 8 * 6 * sizeof(long long));
        a.dbg = dbg;
        hipLaunchKernelGGL(embed_bwd_pool16_kernel<true>, dim3(2 * wg_per_type), dim3(SP_THREADS), lds, s, a);
        long long h[8][6];
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        const double st = (double)a.steps_per_wg;
        for (int wv = 0; wv < 8; ++wv)
            fprintf(stderr, "embed_bwd_pool16 timing, wave %d (cycles per step): handover+A %.0f  barrier %.0f  B+C %.0f  live %.0f  D %.0f\n", wv,
                    h[wv][0] / st, h[wv][1] / st, (h[wv][2] + h[wv][3]) / st, h[wv][4] / st, h[wv][5] / st);
        return launch_check("embed_bwd_pool16");
    }
    hipLaunchKernelGGL(embed_bwd_pool16_kernel<false>, dim3(2 * wg_per_type), dim3(SP_THREADS), lds, s, a);
    return launch_check("embed_bwd_pool16");
}

}  // namespace dc
