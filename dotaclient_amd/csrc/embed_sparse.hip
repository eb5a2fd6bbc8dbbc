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
// Two passes: embed_pool16_prepare_kernel sorts every step's channels by arg-max unit (wave ballots; ascending channel inside
// a unit: deterministic sums) and leaves the result as an image of the head of pass 2's LDS staging block;
// embed_bwd_pool16_kernel (one workgroup = one type x a contiguous range of env-steps, two steps in flight) keeps its
// accumulators in registers for the whole range and fetches everything per step by LDS-DMA.  Two lane <-> data maps, each
// where it saves instructions: the dW2 update runs CHANNEL-per-lane (lane l owns channels l and l + 64, wave w the k range
// [16w, 16w + 16): the scale d[c] and the source row a(c) are lane-local - no broadcast, one 8-byte read for both - and a
// lane gathers its 64 bytes of basic[a(c)] with four ds_read_b128, rows 528 bytes apart so that sixteen different rows at
// one column never share a bank); the first layer, d(basic) and the dW1 fold run UNIT-per-half-wave (wave w, lanes 0..31:
// unit w, lanes 32..63: unit w + 8; lane l of a half owns k = 4l .. 4l + 3: a row of W2 is read by a half as 512 contiguous
// bytes with one ds_read_b128 per lane, the two units of a wave advance in the same instructions, every FMA is packed).
// Steps with a live target-unit head additionally need R[k] = sum_c q[c] W2[c][k] and s[k] = sum_u dtu[u] basic[u][k] for
// the rank-one terms: R is a dense product over all steps (q W2_t, 2 x 2 GFLOP on the matrix cores, fetched like q), s a
// sixteen-lane DPP sum per wave - no workgroup reduction.
// Outputs are per-workgroup partials in the formats the dense path already reduces:
//   slab[wg][128][128] (splitk_reduce_grouped), part1[wg][13][128] (unit_basic_reduce), part2[wg][128] (colsum).
#include <stdio.h>
#include <stdlib.h>
#include <utility>
#include "kernels.h"

namespace dc {
namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;
enum { SP_THREADS = 512, SP_LD = 128, SP_BLD = 132 };   // SP_BLD: floats per LDS row of `basic`: consecutive rows start 4 banks apart, so
                                                        // 16 lanes reading 16 bytes at one column offset of 16 different rows never share a bank
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int SP_OBS = 483, SP_XCAT = 896;

struct SparseArgs {
    const float* obs; const float* dxcat; const uint8_t* amax; const float* dtu; const float* q; int ldq;
    const float* W1; const float* b1; const float* W2;
    float* slab; float* part1; float* part2;
    long long nr; int wg_per_type; int steps_per_wg;
    long long* dbg;
    float* prep;      // [2][nr][IMG_SIZE]: pass 1's image of the head of pass 2's staging block per (type, step)
};

// LDS carve-up (floats).  A workgroup runs NS = 2 independent step streams in lock step (one barrier per iteration):
// twice the independent work per wave between barriers - the per-row chains (broadcast -> address -> LDS -> FMA) are
// latency-bound with two waves per SIMD, so the second stream is nearly free.
enum {
    NS = 2,
    L_W2 = 0,                               // [128][128] W2_t rows
    L_BAS = L_W2 + 128 * SP_LD,             // 2 x NS x [16][SP_BLD] basic of a step (written one iteration ahead)
    L_RED = L_BAS + 2 * NS * 16 * SP_BLD,   // [13][128]: the final sum of the half-waves' dW1 / db1
    L_STG = L_RED + 13 * 128,               // 3 x NS staging blocks: the inputs of a step
    // pass 1's image of a step (IMG_SIZE floats, contiguous in HBM and in LDS) ...
    STG_PB = 0,                        //   per channel {d, byte offset of basic row a(c) (SP_BLD rows)}  [128] x 8 B
    STG_LIST = 256,                    //   channels sorted by arg-max unit: {d, byte offset of W2 row c}, [128 + 16] x 8 B
    STG_SC = 544,                      //   per unit {first list entry, number of channels}          [16] x 8 B
    STG_DT = 576,                      //   dtu[16], [16] = their sum
    STG_FLAG = 600,                    //   1 if any dtu != 0
    STG_R = 608,                       //   R[k] = sum_c q[c] W2[c][k]: written into the image by a dense product over all steps (embed_bwd_pool16)
    IMG_SIZE = 736,
    // ... and what pass 2 fetches from where it already lies
    STG_Q = 736,                       //   q[128]                                  <- headout row
    STG_X = 864,                       //   unit records [16][12], behind 0..3 floats of slack <- obs row, fetched from its 16-byte aligned floor
    STG_SIZE = 1060,
    L_T = L_STG + 3 * NS * STG_SIZE,        // per wave [4 (stream, unit) rows][T_LD]: relu-masked d(basic), re-read as an MFMA operand
    T_LD = 144,                             //   (rows 16 banks apart: the two rows a 32-lane group reads never share a bank)
    L_TOTAL = L_T + 8 * 4 * T_LD
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

// Pass 1: what pass 2 needs of an (env-step, type) beyond plain copies, laid out exactly as the head of pass 2's LDS staging
// block (STG_*), so that pass 2 fetches it with LDS-DMA (global_load_lds, 16 bytes per lane, no registers, no per-thread loader
// roles): the per-channel {d, basic-row offset} pairs, the channel list sorted by arg-max unit, the units' {first, count},
// dtu with its sum and the "head is live" flag.  One wave per (step, type); lane l owns channels l and l + 64.
// The sort - per unit: two ballots, popcounts (count, running first position), v_mbcnt ranks - keeps ascending channel
// order inside a unit, so the sums of pass 2 are deterministic.
__global__ __launch_bounds__(256) void embed_pool16_prepare_kernel(SparseArgs p) {
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= 2 * p.nr) return;
    const int t = item < p.nr ? 2 : 3;
    const long long n = item < p.nr ? item : item - p.nr;
    const int cum = t == 2 ? 6 : 22;                        // first unit of the type inside the 40
    const float* dx = p.dxcat + n * SP_XCAT;
    float* out = p.prep + (size_t)item * IMG_SIZE;
    float d[2];
    int a[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = lane + 64 * h;
        d[h] = t == 2 ? dx[3 * 128 + c] : dx[4 * 128 + c] + dx[6 * 128 + c];   // policy.py:127: enh feeds two slots
        a[h] = p.amax[(n * 3 + (t - 1)) * 128 + c];
        *reinterpret_cast<float2*>(out + STG_PB + 2 * c) = make_float2(d[h], __int_as_float(a[h] * SP_BLD * 4));
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
#pragma unroll
    for (int h = 0; h < 2; ++h)
        *reinterpret_cast<float2*>(out + STG_LIST + 2 * pos[h]) = make_float2(d[h], __int_as_float((lane + 64 * h) * SP_LD * 4));
    {   // dtu of the sixteen units (lanes 0..15 = one DPP row: rotate-and-add leaves the sum in every lane), live flag
        const float v = lane < 16 ? p.dtu[n * 40 + cum + lane] : 0.f;
        float sum = v;
        sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x128, 0xf, 0xf, true));
        sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x124, 0xf, 0xf, true));
        sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x122, 0xf, 0xf, true));
        sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x121, 0xf, 0xf, true));
        const unsigned long long nz = __ballot(v != 0.f);
        if (lane < 16) {
            out[STG_DT + lane] = v;
            // pass 2 reads SP_SLOTS entries from a unit's first one whatever its count: 16 zero-valued entries of row 0 behind
            *reinterpret_cast<float2*>(out + STG_LIST + 2 * (128 + lane)) = make_float2(0.f, __int_as_float(0));
            *reinterpret_cast<int2*>(out + STG_SC + 2 * lane) = make_int2(my_start, my_cnt);
        }
        if (lane == 0) { out[STG_DT + 16] = sum; reinterpret_cast<int*>(out)[STG_FLAG] = nz != 0ull; }
    }
}

// Pass 2.  Workgroup = one type x a contiguous range of env-steps, split into two streams that advance one step per
// iteration each (one barrier per iteration).
// Lane maps (header): dW2 update - lane l owns channels l, l + 64, wave w the k range [16w, 16w + 16).  Everything else -
// half hh = lane >> 5 of wave w owns unit u = w + 8 hh, lane l = lane & 31 of the half owns k4 = 4l .. 4l + 3.  The
// {value, W2-row offset} entries of a unit's channels are fetched sixteen at a time (one 8-byte LDS read per lane, every
// 16-lane DPP row of a half holding the same sixteen entries) and broadcast with row_newbcast: the two units of a wave
// advance in the same instructions, one LDS round trip per W2 row, no divergent loops.
//   iteration i: every thread hands its prefetched values of iteration i+1 to staging[(i+1) % 3] and issues the loads of
//   iteration i+2  -- barrier --  phase A of iteration i+1 (basic -> bas[(i+1) & 1]), phases B, C, (live), D of iteration i.
template <bool TIMING>   // TIMING (DC_DEV_TIMING build): s_memtime phase sums of every wave of workgroup 0 -> p.dbg[8][6]
__global__ __launch_bounds__(SP_THREADS) void embed_bwd_pool16_kernel(SparseArgs p) {
    long long tm[6] = {0, 0, 0, 0, 0, 0}, tm0 = 0;
    auto stamp = [&](int i) {
        if constexpr (TIMING) { const long long x = __builtin_amdgcn_s_memtime(); tm[i] += x - tm0; tm0 = x; }
    };
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, l32 = lane & 31;
    const int k4 = 4 * l32;
    const int u_own = w + 8 * hh;                           // this half-wave's unit
    const int t = 2 + blockIdx.x / p.wg_per_type;          // 2 = allied non-heroes, 3 = enemy non-heroes
    const int wgi = blockIdx.x % p.wg_per_type;
    const long long n0 = (long long)wgi * p.steps_per_wg;
    const long long n1 = min(p.nr, n0 + p.steps_per_wg);
    const int cum = t == 2 ? 6 : 22;                        // first unit of the type inside the 40
    const float* prep_t = p.prep + (size_t)(t - 2) * p.nr * IMG_SIZE;

    // ---- stationary operands ---------------------------------------------------------------------------
    {
        const float4* src = reinterpret_cast<const float4*>(p.W2 + (size_t)t * 128 * 128);
        for (int e = tid; e < 128 * 32; e += SP_THREADS) *reinterpret_cast<float4*>(smem + L_W2 + 4 * e) = src[e];
    }
    // The first layer (12 -> 128, phase A) and the dW1 / db1 fold (phase D) run as 16 x 16 x 4 f32 MFMA products on the otherwise idle
    // matrix pipe (v_mfma_f32_16x16x4_f32: A[i][q], B[q][n] in lane (i or n = lane & 15, q = lane >> 4); D[4 (lane >> 4) + r][lane & 15]):
    //   first layer  basic[u][16w + n] = sum_f x[u][f] W1[16w + n][f]: wave w owns the k block 16w .. 16w + 15 of all sixteen units;
    //                B = W1 (three K = 4 steps) stays in registers for the whole kernel
    //   dW1 fold     out[f][k] += sum over the wave's four (stream, unit) rows of x[row][f] dbm[row][k], f = 12: ones (db1)
    // - a fifth of the kernel's instructions and its two longest FMA chains less than the packed-f32 form (measured: 838 -> 795 us).
    const int mi = lane & 15, mq = lane >> 4;
    float w1b[3];
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) w1b[kk] = p.W1[(16 * w + mi) * 12 + 4 * kk + mq];
    const float b1c = p.b1[16 * w + mi];
    f32x4 accD[8];                 // out[4 mq + r][16 kb + mi]
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) accD[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 D[2][8];                 // dW2[c][16w + 2j, 16w + 2j + 1] of channels c = lane (D[0]) and lane + 64 (D[1])
    float db2a = 0.f;              // threads 0..127: second-layer bias gradient of channel tid
#pragma unroll
    for (int i = 0; i < 8; ++i) { D[0][i] = mk2(0.f, 0.f); D[1][i] = mk2(0.f, 0.f); }

    // ---- two step streams: stream s covers steps [nb[s], ne[s]); iteration i works on step nb[s] + i of both ----------
    const long long half = (n1 - n0 + 1) / 2;
    const long long nb[NS] = {n0, n0 + half}, ne[NS] = {min(n1, n0 + half), n1};
    const long long iters = half;
    // ---- staging: the block of (stream, iteration) goes global -> LDS by DMA, five wave-instructions per stream (the address
    // path takes 100+ cycles per instruction and all of them are issued at once behind the barrier: with sixteen per iteration the
    // last wave in line sat 600 cycles per step in the queue - s_memtime per wave, roles swapped to tell): the image incl. R in three
    // 1 KB pieces, q (512 bytes), the unit records (768 bytes at a 4-byte aligned address) as 49 16-byte lanes from the aligned floor
    // of their address - consumers add the 0..3 floats of slack.  Issued behind
    // the barrier of iteration i for iteration i + 2, awaited (s_waitcnt vmcnt(0): the kernel's only vector-memory traffic) before
    // the barrier of iteration i + 1.
    auto stg_of = [&](long long i, int s2) { return smem + L_STG + ((int)(i % 3) * NS + s2) * STG_SIZE; };
    auto rec_off = [&](long long n) { return (size_t)n * SP_OBS + 3 + cum * 12; };      // first float of the type's records in obs
    auto dma_issue = [&](long long i) {
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            const long long n = nb[s2] + i;
            if (n >= ne[s2]) continue;                             // workgroup-uniform
            float* dst = stg_of(i, s2);
            // this wave's piece of the stream (5..7: none).  q and the records come from buffers last touched a pass ago and take long to
            // issue; they go to the older waves 0..3 (which win their SIMD's arbitration and have slack at the barrier), the image pieces
            // (written by pass 1 a moment ago) to waves 4..7
            const int j = w < 4 ? ((w >> 1) == s2 ? 3 + (w & 1) : 7) : (w < 6 ? w - 4 : (w - 6 == s2 ? 2 : 7));
            if (j < 3) {
                if (j * 1024 + lane * 16 < IMG_SIZE * 4)
                    __builtin_amdgcn_global_load_lds((gptr_t)(prep_t + (size_t)n * IMG_SIZE + j * 256 + lane * 4), (lptr_t)(dst + j * 256), 16, 0, 0);
            } else if (j == 3) {
                if (lane < 32) __builtin_amdgcn_global_load_lds((gptr_t)(p.q + (size_t)n * p.ldq + lane * 4), (lptr_t)(dst + STG_Q), 16, 0, 0);
            } else if (j == 4) {
                if (lane < 49) __builtin_amdgcn_global_load_lds((gptr_t)(p.obs + (rec_off(n) & ~(size_t)3) + lane * 4), (lptr_t)(dst + STG_X), 16, 0, 0);
            }
        }
    };
    auto bas_of = [&](long long i, int s2) { return smem + L_BAS + ((int)(i & 1) * NS + s2) * 16 * SP_BLD; };
    // BOTH (here and in the loop body): both streams have a step in this iteration, known at compile time - no branch splits
    // the two streams' instruction chains into separate blocks, so the scheduler interleaves them (the point of having two)
    auto phase_a = [&](long long i, auto both_c) {             // basic[all units][16w .. 16w + 15] of iteration i's steps
        constexpr bool BOTH = decltype(both_c)::value;
        // (the forward's first layer is an f32 MFMA as well, 32 x 32 x 2: the two may differ in the last bit, i.e. in the relu
        // mask of pre-activations within an ulp of zero)
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            if (!BOTH && nb[s2] + i >= ne[s2]) continue;
            const float* xs = stg_of(i, s2) + STG_X + (int)(rec_off(nb[s2] + i) & 3) + mi * 12 + mq;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[4 * kk], w1b[kk], acc, 0, 0, 0);
            float* bo = bas_of(i, s2) + 4 * mq * SP_BLD + 16 * w + mi;
#pragma unroll
            for (int r = 0; r < 4; ++r) bo[r * SP_BLD] = fmaxf(acc[r] + b1c, 0.f);
        }
    };

    dma_issue(0);
    dma_issue(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();          // W2 / W1 / staging[0], staging[1]
    phase_a(0, std::false_type{});

    const char* w2b = reinterpret_cast<const char*>(smem + L_W2 + k4);     // + W2-row byte offset
    // dW1 fold, deferred by one iteration: an iteration ends by leaving its four relu-masked d(basic) rows in the wave's T block and
    // the matching record operand in xa_prev; the eight MFMAs that fold them run at the top of the NEXT iteration, under its phase A / B
    // instructions, instead of at the end of a chain (LDS round trip + 8 x 32 pipe cycles) with nothing beside them
    float* T = smem + L_T + w * (4 * T_LD);
    for (int e = lane; e < 4 * T_LD; e += 64) T[e] = 0.f;
    float xa_prev = 0.f;
    auto fold_prev = [&]() {
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) accD[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa_prev, T[mq * T_LD + 16 * kb + mi], accD[kb], 0, 0, 0);
    };
    auto iteration = [&](long long i, auto both_c) {     // BOTH: both streams on in iterations i and i + 1
        constexpr bool BOTH = decltype(both_c)::value;
        if constexpr (TIMING) tm0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // staging[i + 1] of this wave's quarter has landed
        stamp(0);
        __syncthreads();
        stamp(1);
        dma_issue(i + 2);
        fold_prev();
        phase_a(i + 1, both_c);
        const float* stg[NS];
        bool on[NS], live[NS];
        float4 basic[NS];
        f32x2 dbl[NS], dbh[NS];       // d(basic)[u_own][k4, k4 + 1] and [k4 + 2, k4 + 3]
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            on[s2] = BOTH || nb[s2] + i < ne[s2];                                 // workgroup-uniform
            stg[s2] = stg_of(i, s2);
            live[s2] = on[s2] && reinterpret_cast<const int*>(stg[s2])[STG_FLAG] != 0;
            dbl[s2] = mk2(0.f, 0.f); dbh[s2] = mk2(0.f, 0.f);
            basic[s2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // ---- phase B: dW2[c][k] += d[c] * basic[a(c)][k], c = lane, lane + 64;  phase C: dbasic[u][k] = sum over the
        // unit's channels of d[c] * W2[c][k], u = u_own - for both streams
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            if (!on[s2]) continue;
            basic[s2] = *reinterpret_cast<const float4*>(bas_of(i, s2) + u_own * SP_BLD + k4);
            {   // phase B, channel per lane: D[h][j] += d[c] * basic[a(c)][16w + 2j .. +1], c = lane + 64 h
                const char* brow = reinterpret_cast<const char*>(bas_of(i, s2)) + w * 64;       // k range of this wave
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float2 e = *reinterpret_cast<const float2*>(stg[s2] + STG_PB + 2 * (lane + 64 * h));   // {d, row byte offset}
                    const float4* rp = reinterpret_cast<const float4*>(brow + __float_as_int(e.y));
                    const f32x2 dd = mk2(e.x, e.x);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float4 r4 = rp[v];
                        D[h][2 * v] = __builtin_elementwise_fma(dd, mk2(r4.x, r4.y), D[h][2 * v]);
                        D[h][2 * v + 1] = __builtin_elementwise_fma(dd, mk2(r4.z, r4.w), D[h][2 * v + 1]);
                    }
                }
            }
            // phase C: {first entry, count} of the own unit, then its entries first .. first + 15 in every 16-lane row
            const int2 sc = *reinterpret_cast<const int2*>(stg[s2] + STG_SC + 2 * u_own);
            const int first = sc.x, cnt = sc.y;
            const float2 ent = *reinterpret_cast<const float2*>(stg[s2] + STG_LIST + 2 * (first + (lane & 15)));
            const float dvec = (lane & 15) < cnt ? ent.x : 0.f;       // past the unit's segment: the next unit's entries, weight 0
            const int ovec = __float_as_int(ent.y);
            f32x2 ol = mk2(0.f, 0.f), oh = mk2(0.f, 0.f);    // odd slots: a second chain (a unit's 8-12 dependent packed FMAs are the
                                                             // longest serial chain of the step; order of the sum stays fixed)
            auto slot = [&](auto JC) {
                constexpr int J = decltype(JC)::value;
                const float4 wv = *reinterpret_cast<const float4*>(w2b + bcast16_i<J>(ovec));
                const float d = bcast16_f<J>(dvec);
                const f32x2 dd = mk2(d, d);
                if constexpr (J & 1) {
                    ol = __builtin_elementwise_fma(dd, mk2(wv.x, wv.y), ol); oh = __builtin_elementwise_fma(dd, mk2(wv.z, wv.w), oh);
                } else {
                    dbl[s2] = __builtin_elementwise_fma(dd, mk2(wv.x, wv.y), dbl[s2]); dbh[s2] = __builtin_elementwise_fma(dd, mk2(wv.z, wv.w), dbh[s2]);
                }
            };
            for_consts(slot, std::integer_sequence<int, 0, 1, 2, 3, 4, 5, 6, 7>{});         // a unit holds 8 channels on average
            const int cmax = max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 32));
            if (cmax > 8) for_consts(slot, std::integer_sequence<int, 8, 9, 10, 11>{});     // wave-uniform
            for (int j = SP_SLOTS; j < cmax; ++j) {       // wave-uniform trip count; a half past its own count adds zeros
                const float2 e = *reinterpret_cast<const float2*>(stg[s2] + STG_LIST + 2 * (first + min(j, cnt - 1 < 0 ? 0 : cnt - 1)));
                const float4 wv = *reinterpret_cast<const float4*>(w2b + __float_as_int(e.y));
                const float d = j < cnt ? e.x : 0.f;
                dbl[s2] = __builtin_elementwise_fma(mk2(d, d), mk2(wv.x, wv.y), dbl[s2]);
                dbh[s2] = __builtin_elementwise_fma(mk2(d, d), mk2(wv.z, wv.w), dbh[s2]);
            }
            dbl[s2] += ol; dbh[s2] += oh;
        }
        stamp(2);
        stamp(3);
        if (live[0] || live[1]) {
            // rank-one attention terms: d(emb)[u][c] += dtu[u] q[c]
            //   dbasic[u][k] += dtu[u] * R[k],  R[k] = sum_c q[c] W2[c][k]       (R: staging, a dense product over all steps)
            //   dW2[c][k]    += q[c] * s[k],    s[k] = sum_u dtu[u] basic[u][k]
            // Both streams whenever either is live (dtu = 0 adds zeros): one block, the streams' chains interleave.
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                if (!on[s2]) continue;
                const float dt = stg[s2][STG_DT + u_own];
                const float4 R = *reinterpret_cast<const float4*>(stg[s2] + STG_R + k4);
                dbl[s2] = __builtin_elementwise_fma(mk2(dt, dt), mk2(R.x, R.y), dbl[s2]);
                dbh[s2] = __builtin_elementwise_fma(mk2(dt, dt), mk2(R.z, R.w), dbh[s2]);
                // s[16w .. 16w + 15] (this wave's k range of the dW2 update): lane (u = lane & 15, g = lane >> 4) takes
                // dtu[u] * basic[u][16w + 4g .. + 3]; rotate-and-add over the sixteen lanes of each DPP row
                const float du = stg[s2][STG_DT + (lane & 15)];
                const float4 bv = *reinterpret_cast<const float4*>(bas_of(i, s2) + (lane & 15) * SP_BLD + 16 * w + 4 * (lane >> 4));
                float sv[4] = {du * bv.x, du * bv.y, du * bv.z, du * bv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sv[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv[e]), 0x128, 0xf, 0xf, true));
                    sv[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv[e]), 0x124, 0xf, 0xf, true));
                    sv[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv[e]), 0x122, 0xf, 0xf, true));
                    sv[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv[e]), 0x121, 0xf, 0xf, true));
                }
                // dW2[c][k] += q[c] * s[k] in the channel-per-lane layout: q lane-local, s wave-uniform (row g of the wave -> SGPRs)
                const float q0 = stg[s2][STG_Q + lane], q1 = stg[s2][STG_Q + lane + 64];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    auto rl = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16 * g)); };
                    const float sa = rl(sv[0]), sb = rl(sv[1]), sc2 = rl(sv[2]), sd = rl(sv[3]);
                    D[0][2 * g] = __builtin_elementwise_fma(mk2(q0, q0), mk2(sa, sb), D[0][2 * g]);
                    D[0][2 * g + 1] = __builtin_elementwise_fma(mk2(q0, q0), mk2(sc2, sd), D[0][2 * g + 1]);
                    D[1][2 * g] = __builtin_elementwise_fma(mk2(q1, q1), mk2(sa, sb), D[1][2 * g]);
                    D[1][2 * g + 1] = __builtin_elementwise_fma(mk2(q1, q1), mk2(sc2, sd), D[1][2 * g + 1]);
                }
            }
        }
        stamp(4);
        // ---- phase D: through the relu into dW1 / db1 (MFMA, above); second-layer bias gradient
        {
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2)        // a stream without a step this iteration: db = basic = 0, a row of zeros
                *reinterpret_cast<float4*>(T + (2 * s2 + hh) * T_LD + k4) =
                    make_float4(basic[s2].x > 0.f ? dbl[s2].x : 0.f, basic[s2].y > 0.f ? dbl[s2].y : 0.f,
                                basic[s2].z > 0.f ? dbh[s2].x : 0.f, basic[s2].w > 0.f ? dbh[s2].y : 0.f);
            // A[f][row mq]: the record of row mq = (stream mq >> 1, unit w + 8 (mq & 1)); f = 12: ones (db1); f > 12: zeros.  Read now:
            // the next iteration's DMA reuses this staging slot
            const int sq = mq >> 1;
            const float* sx = (sq ? stg[1] : stg[0]) + STG_X + (int)(rec_off(nb[sq] + i) & 3);
            xa_prev = mi < 12 ? sx[(w + 8 * (mq & 1)) * 12 + mi] : (mi == 12 ? 1.f : 0.f);
        }
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            if (!on[s2]) continue;
            if (tid < 128) db2a += stg[s2][STG_PB + 2 * tid] + (live[s2] ? stg[s2][STG_Q + tid] * stg[s2][STG_DT + 16] : 0.f);   // column sum of d(emb)
        }
        stamp(5);
    };
    {
        const long long full = ne[1] - nb[1];       // stream 1 has as many steps as stream 0 or one fewer
        long long i = 0;
        for (; i + 1 < full; ++i) iteration(i, std::true_type{});
        for (; i < iters; ++i) iteration(i, std::false_type{});
        fold_prev();
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
            for (int kb = 0; kb < 8; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * mq + r < 13) acc[(4 * mq + r) * 128 + 16 * kb + mi] += accD[kb][r];
        }
        __syncthreads();
    }
    float* o = p.part1 + (size_t)blockIdx.x * 1664;
    for (int e = tid; e < 1664; e += SP_THREADS) o[e] = acc[e];
}


// ---------------------------------------------------------------------------------------------------
// Pass 2, sixteen-wave form (round 4; the default - same box: 709-723 us against the eight-wave kernel's 750-752 us, profiles/r04/pool16_16w_vs_8w.txt).  The eight-wave kernel above keeps BOTH step streams in every wave: 256 registers,
// two waves per SIMD, and the counters say its waves are parked 51 % of their cycles (s_waitcnt on the LDS round trips of the gathers)
// with the issue ports two-thirds busy - latency-bound, not throughput-bound.  Here a workgroup has 1 024 threads: waves 0..7 work on
// stream 0, waves 8..15 on stream 1, each with the lane maps of the eight-wave kernel for ITS stream only - half the live state per
// wave (<= 128 registers), FOUR waves per SIMD to hide the same round trips, the same one barrier per iteration, the same LDS
// footprint (one workgroup per CU: W2 stays whole).  What changes with it:
//   * the dW2 update is split over k, not over the streams: wave W owns dW2[.][8 W .. 8 W + 7] and takes the steps of BOTH streams (half the
//     accumulators per wave, every entry has one owner);
//   * the dW1 / db1 fold works on wave PAIRS: the pair's four (stream, unit) rows are one full K = 4 MFMA operand, each wave folds them for
//     half of the k blocks (four MFMAs); the T rows live per pair, double-buffered by iteration parity (the partner reads them one barrier
//     later).  With two rows per wave (half-empty K, eight MFMAs each) the kernel was 5 % SLOWER than the eight-wave one: right behind the
//     barrier every wave issues its f32 MFMAs, and four in-order waves per SIMD queued behind 1 408 matrix-pipe cycles per iteration;
//   * staging by DMA: piece j of stream s is issued by wave (s, j), j = 0..4.
// ---------------------------------------------------------------------------------------------------
// LDS: the staging ring starts where the eight-wave kernel keeps its final reduction area; that area is aliased onto the T blocks (every
// fold is done by then) - the doubled T blocks need the 6.6 KB.
enum { LW_STG = L_RED, LW_T = LW_STG + 3 * NS * STG_SIZE, LW_TOTAL = LW_T + 8 * 2 * 4 * T_LD, LW_ACC = LW_T, LW_TZ = 0 };

__global__ __launch_bounds__(1024) void embed_bwd_pool16w_kernel(SparseArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int W = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int so = W >> 3, w = W & 7;                       // own stream, wave index inside the stream's group
    const int hh = lane >> 5, l32 = lane & 31;
    const int k4 = 4 * l32;
    const int u_own = w + 8 * hh;                           // this half-wave's unit
    const int t = 2 + blockIdx.x / p.wg_per_type;          // 2 = allied non-heroes, 3 = enemy non-heroes
    const int wgi = blockIdx.x % p.wg_per_type;
    const long long n0 = (long long)wgi * p.steps_per_wg;
    const long long n1 = min(p.nr, n0 + p.steps_per_wg);
    const int cum = t == 2 ? 6 : 22;
    const float* prep_t = p.prep + (size_t)(t - 2) * p.nr * IMG_SIZE;

    {
        const float4* src = reinterpret_cast<const float4*>(p.W2 + (size_t)t * 128 * 128);
        for (int e = tid; e < 128 * 32; e += 1024) *reinterpret_cast<float4*>(smem + L_W2 + 4 * e) = src[e];
    }
    const int mi = lane & 15, mq = lane >> 4;
    float w1b[3];
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) w1b[kk] = p.W1[(16 * w + mi) * 12 + 4 * kk + mq];
    const float b1c = p.b1[16 * w + mi];
    constexpr int NKB = 4;         // this wave folds k blocks 4 (W & 1) .. + 3 of its pair's four rows
    f32x4 accD[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) accD[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 D[2][4];                 // dW2[c = lane + 64 h][8 W + 2 j, + 1]: the sixteen waves split k, each takes BOTH streams' steps
    float db2a = 0.f;              // threads 0..127: second-layer bias gradient of channel tid (both streams)
#pragma unroll
    for (int i = 0; i < 4; ++i) { D[0][i] = mk2(0.f, 0.f); D[1][i] = mk2(0.f, 0.f); }

    const long long half = (n1 - n0 + 1) / 2;
    const long long nb[NS] = {n0, n0 + half}, ne[NS] = {min(n1, n0 + half), n1};
    const long long iters = half;
    const long long nbo = so ? nb[1] : nb[0], neo = so ? ne[1] : ne[0];     // own stream's step range
    auto stg_of = [&](long long i, int s2) { return smem + LW_STG + ((int)(i % 3) * NS + s2) * STG_SIZE; };
    auto rec_off = [&](long long n) { return (size_t)n * SP_OBS + 3 + cum * 12; };
    auto dma_issue = [&](long long i) {          // own stream; piece j = w: 0..2 image (3 x 1 KB), 3 q, 4 the unit records
        const long long n = nbo + i;
        if (n >= neo || w > 4) return;                             // wave-uniform
        float* dst = stg_of(i, so);
        if (w < 3) {
            if (w * 1024 + lane * 16 < IMG_SIZE * 4)
                __builtin_amdgcn_global_load_lds((gptr_t)(prep_t + (size_t)n * IMG_SIZE + w * 256 + lane * 4), (lptr_t)(dst + w * 256), 16, 0, 0);
        } else if (w == 3) {
            if (lane < 32) __builtin_amdgcn_global_load_lds((gptr_t)(p.q + (size_t)n * p.ldq + lane * 4), (lptr_t)(dst + STG_Q), 16, 0, 0);
        } else {
            if (lane < 49) __builtin_amdgcn_global_load_lds((gptr_t)(p.obs + (rec_off(n) & ~(size_t)3) + lane * 4), (lptr_t)(dst + STG_X), 16, 0, 0);
        }
    };
    auto bas_of = [&](long long i, int s2) { return smem + L_BAS + ((int)(i & 1) * NS + s2) * 16 * SP_BLD; };
    // FULL (here and in the loop body): both streams have a step in this iteration and the next, known at compile time - no uniform
    // branch splits the body into scheduling regions, so the f32 MFMAs of the fold / first layer can be placed among the gathers
    auto phase_a = [&](long long i, auto full_c) {   // basic[all units][16w .. 16w + 15] of the own stream's step of iteration i
        if (!decltype(full_c)::value && nbo + i >= neo) return;
        const float* xs = stg_of(i, so) + STG_X + (int)(rec_off(nbo + i) & 3) + mi * 12 + mq;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[4 * kk], w1b[kk], acc, 0, 0, 0);
        float* bo = bas_of(i, so) + 4 * mq * SP_BLD + 16 * w + mi;
#pragma unroll
        for (int r = 0; r < 4; ++r) bo[r * SP_BLD] = fmaxf(acc[r] + b1c, 0.f);
    };

    // T rows of the wave PAIR (W >> 1): [iteration parity][row = 2 (W & 1) + hh][T_LD]
    float* Tp = smem + LW_T + (W >> 1) * (2 * 4 * T_LD);
    for (int e = lane; e < 4 * T_LD; e += 64) Tp[(W & 1) * 4 * T_LD + e] = 0.f;     // wave W & 1 clears buffer W & 1

    dma_issue(0);
    dma_issue(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();          // W2 / staging[0], staging[1] / T
    phase_a(0, std::false_type{});

    const char* w2b = reinterpret_cast<const char*>(smem + L_W2 + k4);
    float xa_prev = 0.f;
    auto fold_prev = [&](long long i_prev) {          // the pair's four rows of iteration i_prev (buffer i_prev & 1), this wave's four k blocks
        const float* tbp = Tp + ((int)(i_prev & 1) * 4 + mq) * T_LD + 64 * (W & 1) + mi;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) accD[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa_prev, tbp[16 * kb], accD[kb], 0, 0, 0);
    };
    auto iteration = [&](long long i, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's piece of staging[i + 1] has landed
        __syncthreads();
        dma_issue(i + 2);
        fold_prev(i - 1);                                      // (iteration 0: both T buffers are zero)
        phase_a(i + 1, full_c);
        const bool on = FULL || nbo + i < neo;                                 // uniform over the stream's eight waves
        const float* stg = stg_of(i, so);
        const bool live = on && reinterpret_cast<const int*>(stg)[STG_FLAG] != 0;
        f32x2 dbl = mk2(0.f, 0.f), dbh = mk2(0.f, 0.f);       // d(basic)[u_own][k4, k4 + 1] and [k4 + 2, k4 + 3]
        float4 basic = make_float4(0.f, 0.f, 0.f, 0.f);
        // phase B, channel per lane, BOTH streams: D[h][j] += d[c] * basic[a(c)][8 W + 2 j .. + 1], c = lane + 64 h
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            if (!FULL && nb[s2] + i >= ne[s2]) continue;                          // workgroup-uniform
            const float* sg = stg_of(i, s2);
            const char* brow = reinterpret_cast<const char*>(bas_of(i, s2)) + W * 32;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float2 e = *reinterpret_cast<const float2*>(sg + STG_PB + 2 * (lane + 64 * h));   // {d, row byte offset}
                const float4* rp = reinterpret_cast<const float4*>(brow + __float_as_int(e.y));
                const f32x2 dd = mk2(e.x, e.x);
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const float4 r4 = rp[v];
                    D[h][2 * v] = __builtin_elementwise_fma(dd, mk2(r4.x, r4.y), D[h][2 * v]);
                    D[h][2 * v + 1] = __builtin_elementwise_fma(dd, mk2(r4.z, r4.w), D[h][2 * v + 1]);
                }
            }
        }
        if (on) {
            basic = *reinterpret_cast<const float4*>(bas_of(i, so) + u_own * SP_BLD + k4);
            // phase C: {first entry, count} of the own unit, then its entries first .. first + 15 in every 16-lane row
            const int2 sc = *reinterpret_cast<const int2*>(stg + STG_SC + 2 * u_own);
            const int first = sc.x, cnt = sc.y;
            const float2 ent = *reinterpret_cast<const float2*>(stg + STG_LIST + 2 * (first + (lane & 15)));
            const float dvec = (lane & 15) < cnt ? ent.x : 0.f;
            const int ovec = __float_as_int(ent.y);
            f32x2 ol = mk2(0.f, 0.f), oh = mk2(0.f, 0.f);
            auto slot = [&](auto JC) {
                constexpr int J = decltype(JC)::value;
                const float4 wv = *reinterpret_cast<const float4*>(w2b + bcast16_i<J>(ovec));
                const float d = bcast16_f<J>(dvec);
                const f32x2 dd = mk2(d, d);
                if constexpr (J & 1) {
                    ol = __builtin_elementwise_fma(dd, mk2(wv.x, wv.y), ol); oh = __builtin_elementwise_fma(dd, mk2(wv.z, wv.w), oh);
                } else {
                    dbl = __builtin_elementwise_fma(dd, mk2(wv.x, wv.y), dbl); dbh = __builtin_elementwise_fma(dd, mk2(wv.z, wv.w), dbh);
                }
            };
            for_consts(slot, std::integer_sequence<int, 0, 1, 2, 3, 4, 5, 6, 7>{});
#ifdef DC_P16W_SCHED
            // A/B build: ask the scheduler to SPREAD the seven f32 MFMAs of this block (four of the fold, three of the first layer) through
            // its vector / LDS instructions instead of issuing them close together behind the barrier: with four waves per SIMD at the
            // same point of the same code, back-to-back MFMAs of one wave wait for the other three's
            if constexpr (FULL) {
#pragma unroll
                for (int gsb = 0; gsb < 7; ++gsb) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002 | 0x100 | 0x200, DC_P16W_SCHED, 0);
                }
            }
#endif
            const int cmax = max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 32));
            if (cmax > 8) for_consts(slot, std::integer_sequence<int, 8, 9, 10, 11>{});     // wave-uniform
            for (int j = SP_SLOTS; j < cmax; ++j) {
                const float2 e = *reinterpret_cast<const float2*>(stg + STG_LIST + 2 * (first + min(j, cnt - 1 < 0 ? 0 : cnt - 1)));
                const float4 wv = *reinterpret_cast<const float4*>(w2b + __float_as_int(e.y));
                const float d = j < cnt ? e.x : 0.f;
                dbl = __builtin_elementwise_fma(mk2(d, d), mk2(wv.x, wv.y), dbl);
                dbh = __builtin_elementwise_fma(mk2(d, d), mk2(wv.z, wv.w), dbh);
            }
            dbl += ol; dbh += oh;
            if (live) {       // rank-one attention terms (see the eight-wave kernel)
                const float dt = stg[STG_DT + u_own];
                const float4 R = *reinterpret_cast<const float4*>(stg + STG_R + k4);
                dbl = __builtin_elementwise_fma(mk2(dt, dt), mk2(R.x, R.y), dbl);
                dbh = __builtin_elementwise_fma(mk2(dt, dt), mk2(R.z, R.w), dbh);
            }
        }
        // dW2[c][k] += q[c] * s[k], s[k] = sum_u dtu[u] basic[u][k], for the live steps of BOTH streams over this wave's k range
        // 8 W .. 8 W + 7: lane (u = lane & 15, g = lane >> 4; g < 2) takes dtu[u] * basic[u][8 W + 4 g .. + 3], sixteen-lane DPP sums
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            if (!FULL && nb[s2] + i >= ne[s2]) continue;
            const float* sg = stg_of(i, s2);
            if (reinterpret_cast<const int*>(sg)[STG_FLAG] == 0) continue;        // workgroup-uniform
            {
                const float du = sg[STG_DT + (lane & 15)];
                const float4 bv = *reinterpret_cast<const float4*>(bas_of(i, s2) + (lane & 15) * SP_BLD + 8 * W + 4 * ((lane >> 4) & 1));
                float sv[4] = {du * bv.x, du * bv.y, du * bv.z, du * bv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sv[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv[e]), 0x128, 0xf, 0xf, true));
                    sv[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv[e]), 0x124, 0xf, 0xf, true));
                    sv[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv[e]), 0x122, 0xf, 0xf, true));
                    sv[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv[e]), 0x121, 0xf, 0xf, true));
                }
                const float q0 = sg[STG_Q + lane], q1 = sg[STG_Q + lane + 64];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    auto rl = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16 * g)); };
                    const float sa = rl(sv[0]), sb = rl(sv[1]), sc2 = rl(sv[2]), sd = rl(sv[3]);
                    D[0][2 * g] = __builtin_elementwise_fma(mk2(q0, q0), mk2(sa, sb), D[0][2 * g]);
                    D[0][2 * g + 1] = __builtin_elementwise_fma(mk2(q0, q0), mk2(sc2, sd), D[0][2 * g + 1]);
                    D[1][2 * g] = __builtin_elementwise_fma(mk2(q1, q1), mk2(sa, sb), D[1][2 * g]);
                    D[1][2 * g + 1] = __builtin_elementwise_fma(mk2(q1, q1), mk2(sc2, sd), D[1][2 * g + 1]);
                }
            }
        }
        // ---- phase D: through the relu into the wave's T row (folded at the top of the next iteration); the fold's A operand
        *reinterpret_cast<float4*>(Tp + ((int)(i & 1) * 4 + 2 * (W & 1) + hh) * T_LD + k4) =
            make_float4(basic.x > 0.f ? dbl.x : 0.f, basic.y > 0.f ? dbl.y : 0.f, basic.z > 0.f ? dbh.x : 0.f, basic.w > 0.f ? dbh.y : 0.f);
        {
            // A[f][row mq] (f = 12: ones -> db1).  Read now: the next iteration's DMA reuses this staging slot
            const float* sx = stg + STG_X + (int)(rec_off(nbo + i) & 3);
            // rows = the pair's (wave, half): wave (w & ~1) | (mq >> 1), unit that wave + 8 (mq & 1)
            xa_prev = on ? (mi < 12 ? sx[(((w & ~1) | (mq >> 1)) + 8 * (mq & 1)) * 12 + mi] : (mi == 12 ? 1.f : 0.f)) : 0.f;
        }
        if (tid < 128) {      // column sum of d(emb) over both streams' steps (threads 0..127 = the first two waves)
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                if (!FULL && nb[s2] + i >= ne[s2]) continue;
                const float* sg = stg_of(i, s2);
                const bool lv = reinterpret_cast<const int*>(sg)[STG_FLAG] != 0;
                db2a += sg[STG_PB + 2 * tid] + (lv ? sg[STG_Q + tid] * sg[STG_DT + 16] : 0.f);
            }
        }
    };
    {
        const long long full = ne[1] - nb[1];       // stream 1 has as many steps as stream 0 or one fewer
        long long i = 0;
        for (; i + 1 < full; ++i) iteration(i, std::true_type{});
        for (; i < iters; ++i) iteration(i, std::false_type{});
    }
    __syncthreads();           // the partner's rows of the last iteration
    if (iters > 0) fold_prev(iters - 1);

    // ---- results --------------------------------------------------------------------------------------
    {
        float* out = p.slab + (size_t)blockIdx.x * 128 * 128;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<float4*>(out + (size_t)(lane + 64 * h) * 128 + 8 * W + 4 * j) =
                    make_float4(D[h][2 * j].x, D[h][2 * j].y, D[h][2 * j + 1].x, D[h][2 * j + 1].y);
        if (tid < 128) p.part2[(size_t)blockIdx.x * 128 + tid] = db2a;
    }
    // dW1 / db1: sum the 16 waves in fixed order through LDS -> part1[wg][f][k] (f = 12: db1)
    __syncthreads();
    float* acc = smem + LW_ACC;     // [13][128]  (pair variant: on top of the T blocks - every fold is done)
    for (int e = tid; e < 13 * 128; e += 1024) acc[e] = 0.f;
    __syncthreads();
    for (int ww = 0; ww < 16; ++ww) {
        if (W == ww) {
            const int kb0 = 4 * (W & 1);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * mq + r < 13) acc[(4 * mq + r) * 128 + 16 * (kb0 + kb) + mi] += accD[kb][r];
        }
        __syncthreads();
    }
    float* o = p.part1 + (size_t)blockIdx.x * 1664;
    for (int e = tid; e < 1664; e += 1024) o[e] = acc[e];
}

// slab: 2 * wg_per_type x [128][128]; part1: 2 * wg_per_type x [13][128]; part2: 2 * wg_per_type x [128];
// prep: 2 * nr * 736 floats of scratch for the prepared staging blocks
int embed_bwd_pool16(const float* obs, const float* dxcat, const uint8_t* amax, const float* dtu, const float* q, int ldq,
                     const float* W1, const float* b1, const float* W2, float* slab, float* part1, float* part2, float* prep,
                     long long nr, int wg_per_type, hipStream_t s, int eight_waves) {
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
    // R[n][k] = sum_c q[n][c] W2_t[c][k] of every step and type, straight into the R section of the images (row stride IMG_SIZE):
    // 2 x 2 GFLOP on the matrix cores instead of 128 x 128 MACs per live step on the vector unit
    for (int t = 2; t < 4; ++t)
        if (int e = gemm_f32(q, W2 + (size_t)t * 128 * 128, prep + (size_t)(t - 2) * nr * IMG_SIZE + STG_R, (int)nr, 128, 128, ldq, 128,
                             IMG_SIZE, 0, 1, nullptr, 0, nullptr, 0, 0, 1, s))
            return e;
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
    if (eight_waves) {
        hipLaunchKernelGGL(embed_bwd_pool16_kernel<false>, dim3(2 * wg_per_type), dim3(SP_THREADS), lds, s, a);
        return launch_check("embed_bwd_pool16");
    }
    const size_t ldsw = (size_t)LW_TOTAL * sizeof(float);
    static bool attrw = false;
    if (!attrw) {
        hipError_t e = hipFuncSetAttribute((const void*)embed_bwd_pool16w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
        if (e != hipSuccess) { set_error("embed_bwd_pool16 (16 waves): hipFuncSetAttribute", (int)e); return (int)e; }
        attrw = true;
    }
    hipLaunchKernelGGL(embed_bwd_pool16w_kernel, dim3(2 * wg_per_type), dim3(1024), ldsw, s, a);
    return launch_check("embed_bwd_pool16 (16 waves)");
}

}  // namespace dc
