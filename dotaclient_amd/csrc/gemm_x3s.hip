// Row-streaming form of the f16x2 dense products x W^T / dy W (round 6): the default kernel for the activations-times-weights
// products of /root/reference/policy.py:54-75,138-155 (affine_pre_rnn, the recurrent input projection, the head block) and of their
// input gradients (torch autograd at /root/reference/optimizer.py:672) whenever the shape fills the chip; gemm_x3.hip's 128 x 128
// split-on-load kernel keeps the weight gradients, the bf16 mode, the bf16x3 fallback and the small shapes.
//
// Why another kernel.  gemm_x3.hip (PREC 4) sits at 0.23-0.27 of the three-MFMA ceiling with its matrix pipes ~30 % busy: per K = 16 step
// a workgroup stages both operands through registers (split VALU + ds_write at ~80 B/clk) and meets at a barrier after only twelve MFMAs
// per wave; LDS stores, LDS reads, VALU and MFMA each take about as long as the others and do not overlap (profiles/r05/
// gemm_x3_ablation.txt).  Here (measurements of every step on the way: profiles/r06/gemm_x3s_development.txt):
//   * BOTH operands go global -> LDS by DMA (global_load_lds_dwordx4: no VGPRs, no ds_write): the activations as the f32 they are in HBM
//     (FastTile's lane-linear [row][32 k] image, XOR-swizzled on the source side), the weights as their pre-split f16 planes
//     (PlaneTile<128, 2>).  The DMA is issued from inline asm and counted by hand: hipcc puts a vmcnt(0) in front of every ds_read that
//     follows a builtin LDS-DMA it knows of - the whole queue drained every stage (tools/x3s_isa_check.py keeps its waits out of the K loop);
//   * the eight waves of a workgroup split the 256 ROWS of its tile and each takes all 128 columns: an activation element is read from
//     LDS, scaled and split into its two f16 pieces by exactly ONE wave (16 elements per lane and stage), in registers - the weight
//     fragments need no arithmetic at all;
//   * three LDS stages of 48 KB (K = 32 each), DMA two stages ahead, raw s_barrier with COUNTED vmcnt (the DMA of the stage after next
//     stays in flight across the barriers), 24 MFMAs per wave and stage;
//   * TWO WAVE GROUPS HALF A STAGE APART: a stage is a MEMORY phase (DMA pieces, fragment reads into registers, split) and a MATRIX phase
//     (24 MFMAs from registers, s_setprio 1); waves 4-7 run one barrier behind waves 0-3, so the two waves of a SIMD are always in
//     opposite phases - its matrix pipe belongs to one while the other issues its DMA and LDS reads (all eight in step: ~170 cycles per
//     DMA piece blocked behind the others' on the 64 B/clk address path with nobody issuing MFMAs, ~3 700 cycles per stage for 1 536 of
//     matrix work);
//   * D = x_tile W_tile^T leaves one output column per lane and one row per register: the epilogue is scale - bias - relu - mask on the
//     accumulators and stores of whole 128-byte lines (a half-wave each), no LDS round trip; the bias comes from an LDS table filled once;
//   * persistent workgroups (one per CU: all 160 KB of LDS, two waves per SIMD) walk the (row tile, column tile) items of an XCD in
//     row-tile order, so the column tiles of a row tile meet in that XCD's L2 (HBM traffic 1.15 x the algorithmic bytes).
// What bounds it now: the MATRIX phases (2 x ~1 120 ticks of ~2 900 per stage: 24 MFMAs measure ~46 ticks each at the socket's power cap,
// not 32) and what the phases leave uncovered - barriers, loop control, the DMA wait ~700 per stage -, then the epilogue (~3 000 ticks per
// wave group and item: 64 stores of a whole line each).  Not kept: staggered workgroup starts, stores trickled out of parked registers
// under the next item's MFMAs (slower: vmcnt retires in order, every DMA wait then waits for the stores in front of it).
// Arithmetic is gemm_x3.hip's PREC 4 exactly: x sa = h + m (two f16 pieces), a b = (hh + hm + mh) / (sa sb), f32 accumulate.
#include <cstdio>
#include <type_traits>
#include "kernels.h"
#include "gemm_tiles.h"

#ifndef X3S_TIMING
#define X3S_TIMING 0      // developer build: s_memtime stamps of (workgroup 0, wave 0), summed per phase; a blocking read-back and a line on stderr per launch
#endif

namespace dc {
namespace {

enum { SB_M = 256, SB_N = 128, SB_K = 32, SB_WAVES = 8, SB_THREADS = 512, SB_NST = 3 };
enum { SA_BYTES = SB_M * SB_K * 4 /* 32768 */, SPLANE_BYTES = SB_N * SB_K * 2 /* 8192 */, SB_BYTES = 2 * SPLANE_BYTES,
       SSTAGE_BYTES = SA_BYTES + SB_BYTES /* 49152 */, X3S_LDS = SB_NST * SSTAGE_BYTES + 4 * 4096 /* 163840 = all of a CU's LDS: three stages + the epilogue images */ };
enum { SA_PIECES = SA_BYTES / 1024 / SB_WAVES /* 4 */, SB_PIECES = SB_BYTES / 1024 / SB_WAVES /* 2 */, S_DMA = SA_PIECES + SB_PIECES /* 6 per lane and stage */ };

// D = x_tile W_tile^T: a lane holds one output column, a register one row - a half-wave's 32 lanes are a whole 128-byte line of C
#define X3S_MFMA(w, x, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(x, w, c, 0, 0, 0)

struct X3SArgs {
    const float* A; const uint16_t* B;
    float* C; const float* bias; const float* aux;
    long long b_plane;
    int M, N, K, lda, ldb, ldc, ldaux, relu, nbias;
    float sa, inv;
    long long* dbg;
};

// (row tile, column tile) of the n-th item this workgroup takes; false past the end.  With a multiple of 8 row tiles XCD x (blockIdx & 7
// under the usual round-robin placement - a speed hint only) owns the row tiles x, x + 8, ...; its workgroups walk that list in row-tile
// order, `run` consecutive items each (run = the column tiles of a row tile when there are at most two: both tiles of a row tile then
// run on one CU back to back - that also balances the 128 + 32 column split of the head block).
struct ItemMap {
    int mt, nt, n_items, run, xcd;
    __device__ __forceinline__ bool decode(int n, int& m_blk, int& n_blk) const {
        const int r = n / run, i = n - r * run;
        if (xcd) {
            const int per = n_items >> 3;
            const int li = (r * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3)) * run + i;
            if (li >= per) return false;
            const int ml = li / nt;
            m_blk = (ml * 8 + (int)(blockIdx.x & 7)) * SB_M; n_blk = (li - ml * nt) * SB_N;
            return true;
        }
        const int w = (r * (int)gridDim.x + (int)blockIdx.x) * run + i;
        if (w >= n_items) return false;
        const int ml = w / nt;
        m_blk = ml * SB_M; n_blk = (w - ml * nt) * SB_N;
        return true;
    }
};

// One stage's six DMA pieces of this wave (4 KB of the activation image, 2 KB of the weight planes) as ONE asm statement: hipcc must not
// know that these are LDS writes - it would drain the whole DMA queue (vmcnt(0)) in front of the next ds_read, whatever buffer that read
// touches (seen in the .s of the builtin form of this kernel) - so the counting is done by hand below (cdna_hip_programming.md 5.7).
// Source = 64-bit scalar base (advanced per stage on the SALU) + this lane's 32-bit byte offset (constant per item); M0 = LDS base of
// the piece, lane l lands at M0 + 16 l.
__device__ __forceinline__ void dma_stage(const float* a_base, const uint16_t* b_base, const unsigned (&ao)[SA_PIECES], const unsigned (&bo)[SB_PIECES],
                                          unsigned lds_a, unsigned lds_b) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %9\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %7\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %7\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %7\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %7\n\t"
        "s_mov_b32 m0, %10\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %5, %8\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %6, %8\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(ao[0]), "v"(ao[1]), "v"(ao[2]), "v"(ao[3]), "v"(bo[0]), "v"(bo[1]), "s"(a_base), "s"(b_base), "s"(lds_a), "s"(lds_b)
        : "memory", "scc");
}
// One piece, issued BETWEEN the MFMA groups of a stage.  All eight waves issuing their six pieces together right behind the barrier block for
// ~1 000 cycles each - 48 wave-instructions of 1 KB through a 64 B/clk address path - with the matrix pipes idle, and reach the next
// barrier ~1 000 cycles apart (phase clocks of the first version: DMA issue 800-1 200, barrier 900-1 400 of ~4 100 cycles per stage); one
// piece per four MFMAs keeps the address path at half load and the waves in step.  `pin` (an accumulator, untouched) orders the statement
// between the MFMA that last wrote it and the one that reads it next: hipcc moves register-only instructions across an asm otherwise.
__device__ __forceinline__ void dma_piece(const void* base, unsigned voff, unsigned lds, f32x16& pin) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "+v"(pin)
        : "v"(voff), "s"(base), "s"(lds)
        : "memory");
}
static_assert(SA_PIECES == 4 && SB_PIECES == 2, "dma_stage is written for 4 + 2 pieces per wave");

// PARTIAL: N is not a multiple of 128 - the last column tile multiplies only its live 32-column blocks
// EPI: what the epilogue does besides scale + bias - compile-time, because it is the epilogue's instruction count that costs (below)
enum { X3S_PLAIN = 0, X3S_RELU = 1, X3S_MASK = 2 };
template <bool PARTIAL, int EPI>
__global__ __launch_bounds__(SB_THREADS, 2) void gemm_x3s_kernel(X3SArgs p, ItemMap im) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fg = lane >> 5;
    const int nk = p.K / SB_K;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);

    // ---- DMA side: a stream of stages (item, k step) that runs two stages ahead of the MFMAs ----------------------------------------
    // A piece n = wave * 4 + i: tile rows 8 n .. 8 n + 7, lane = (row & 7) * 8 + slot; slot holds k chunk slot ^ ((row >> 1) & 7)
    // B piece n = wave * 2 + i: plane n >> 3, tile rows 16 (n & 7) .. + 15, lane = (row & 15) * 4 + slot; chunk slot ^ ((row >> 2) & 3)
    unsigned a_voff[SA_PIECES], b_voff[SB_PIECES];      // byte offsets of this lane's pieces from A / B at k = 0 (per item)
    int d_item = 0, d_kt = 0;                           // next stage to issue
    bool d_on;
    auto d_open = [&]() __attribute__((always_inline)) {
        int m_blk, n_blk;
        d_on = im.decode(d_item, m_blk, n_blk);
        if (!d_on) return;
#pragma unroll
        for (int i = 0; i < SA_PIECES; ++i) {
            const int row = (wave * SA_PIECES + i) * 8 + (lane >> 3);
            const int rg = min(m_blk + row, p.M - 1);      // rows past M re-read the last one (never stored)
            a_voff[i] = ((unsigned)rg * (unsigned)p.lda + 4u * ((lane & 7) ^ ((row >> 1) & 7))) * 4u;
        }
#pragma unroll
        for (int i = 0; i < SB_PIECES; ++i) {
            const int n = wave * SB_PIECES + i;
            const int row = (n & 7) * 16 + (lane >> 2);
            const int rg = min(n_blk + row, p.N - 1);      // columns past N likewise
            b_voff[i] = (unsigned)((n >> 3) * p.b_plane * 2) + ((unsigned)rg * (unsigned)p.ldb + 8u * ((lane & 3) ^ ((row >> 2) & 3))) * 2u;
        }
    };
    auto d_issue = [&](int buf) __attribute__((always_inline)) {        // one stage into LDS buffer `buf`; advances the stream
        const unsigned st = lds0 + (unsigned)buf * SSTAGE_BYTES;
#ifndef X3S_NO_DMA           // ablation build: timing only
        dma_stage(p.A + d_kt * SB_K, p.B + d_kt * SB_K, a_voff, b_voff, st + wave * (SA_PIECES * 1024), st + SA_BYTES + wave * (SB_PIECES * 1024));
#endif
        if (++d_kt == nk) { d_kt = 0; ++d_item; d_open(); }
    };

    // ---- MFMA side ---------------------------------------------------------------------------------------------------------------------
    // activation fragment of sub-step s: row 32 wave + fr, k = 16 s + 8 fg .. + 7 = k chunks 4 s + 2 fg, + 1 (two float4);
    // weight fragment of block j, plane pl: row 32 j + fr, chunk (2 s + fg) ^ ((fr >> 2) & 3)
    int a_off[2][2], b_off[2];
    {
        const int r = wave * 32 + fr, sw = (r >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int e = 0; e < 2; ++e) a_off[s][e] = r * 128 + 16 * ((4 * s + 2 * fg + e) ^ sw);
            b_off[s] = SA_BYTES + fr * 64 + 16 * ((2 * s + fg) ^ ((fr >> 2) & 3));
        }
    }

    int m_blk, n_blk;
    if (!im.decode(0, m_blk, n_blk)) return;
    long long tm[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
    const bool timing = X3S_TIMING && p.dbg != nullptr && tid == 0;      // (every workgroup's wave 0: the per-phase sums of workgroup 0, the totals of all)
    const long long t_begin = X3S_TIMING ? (long long)__builtin_amdgcn_s_memtime() : 0;
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if (X3S_TIMING && timing) { const long long now = (long long)__builtin_amdgcn_s_memtime(); tm[k] += now - tlast; tlast = now; }
    };
    if (p.bias != nullptr)               // the bias vector into its LDS table (N <= 4096: gemm_x3s_eligible); the prologue's barrier publishes it
        for (int i = tid; i < p.N; i += SB_THREADS) reinterpret_cast<float*>(smem + SB_NST * SSTAGE_BYTES)[i] = i < p.nbias ? p.bias[i] : 0.f;
    d_open();
    d_issue(0);
    {
        const bool second = d_on;        // a second stage exists (K > 32 or another item)
        if (second) d_issue(1);
        // stage 0 has landed (this wave's pieces; the barrier makes it everybody's)
        if (second) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S_DMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- two wave groups half a stage apart ------------------------------------------------------------------------------------------
    // A stage is two phases, each closed by a workgroup barrier: MEMORY (DMA pieces of the stage after next, fragment reads of this stage
    // into registers, split) and MATRIX (24 MFMAs from registers).  Waves 4-7 take one extra barrier up front and none behind their last
    // MATRIX phase: from then on a SIMD's two waves (w and w + 4) are always in OPPOSITE phases - the matrix pipe of a SIMD belongs to one
    // wave while the other one issues its DMA and LDS reads.  (First form of this kernel, all eight waves in step: every wave blocked
    // ~170 cycles per DMA piece behind the seven others' on the 64 B/clk address path with nobody left to issue MFMAs, ~3 700 cycles per
    // stage for 1 536 of matrix work per SIMD.)
    //   global phase p:        2t          2t + 1        2t + 2        2t + 3
    //   waves 0-3 (G0):     MEMORY(t)    MATRIX(t)    MEMORY(t+1)   MATRIX(t+1)
    //   waves 4-7 (G1):     MATRIX(t-1)  MEMORY(t)    MATRIX(t)     MEMORY(t+1)
    // DMA(t + 2) goes out in MEMORY(t) into buffer (t + 2) % 3 = (t - 1) % 3, last read in phase 2t - 1 (G1's MEMORY(t - 1)): free.
    // Stage t + 1 is first read in phase 2t + 2 (G0), so every piece of it must have landed before the barrier that closes phase 2t + 1:
    //   G0 waits at the END of MATRIX(t) for "at most the S_DMA pieces of DMA(t + 2) in flight" (its DMA(t + 1) went out in phase 2t - 2);
    //   G1 waits at the START of MEMORY(t) for everything (its DMA(t + 1) went out in phase 2t - 1 and nothing younger exists yet).
    // Stores of an epilogue are older than the DMA that follows them in program order, so the same counts cover them.
    const bool g1 = wave >= SB_WAVES / 2;
    if (g1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    if (X3S_TIMING && timing) tlast = (long long)__builtin_amdgcn_s_memtime();

    int buf = 0;
    f32x16 acc[4];
    int nj = PARTIAL ? min(4, (p.N - n_blk + 31) >> 5) : 4;

    // Epilogue straight from the accumulators.  With D = x_tile W_tile^T the 32x32 MFMA leaves output column n0 + (lane & 31) in every lane
    // and row (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) in register reg: one store instruction writes two whole 128-byte lines (one per
    // half-wave), 64 per wave and item, no LDS round trip.  What an epilogue costs is its VALU instructions - 5 400 ticks per wave and item
    // for ~11 per element (two 64-bit multiplies for the address of every store, a compare + select for a mask that is not there, a select
    // for a relu that is not asked for; the forms before this one, with D = W x^T: 16-byte stores of one row per lane 7 500, whole lines
    // through 4 KB LDS images 5 400 - phase clocks in profiles/r06/gemm_x3s_development.txt, where staggered workgroup starts and stores
    // trickled out of parked registers are recorded as well: no gain / slower).  So: relu / mask are compile-time (EPI), a row's address
    // is a SCALAR base (advanced on the SALU) plus ONE per-lane offset for the whole kernel, the accumulators are not zeroed (the next
    // item's first MFMAs take a zero C operand): a fused multiply-add (+ a max) per element.  The bias comes from an LDS table of the
    // whole vector (the 16 KB behind the stage buffers), filled once per workgroup.  Rows past M (the last row tile): the guarded form.
    const float* bias_lds = reinterpret_cast<const float*>(smem + SB_NST * SSTAGE_BYTES);
    const unsigned lane_c = (unsigned)(4 * fg) * (unsigned)p.ldc + (unsigned)fr;            // this lane's element offset inside a wave's 32-row block
    const unsigned lane_x = (unsigned)(4 * fg) * (unsigned)p.ldaux + (unsigned)fr;
    auto epilogue = [&](int em, int en, bool) __attribute__((always_inline)) {
        const int enj = PARTIAL ? min(4, (p.N - en + 31) >> 5) : 4; // live 32-column blocks (N % 32 == 0: a block is all in or all out)
        float bj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bj[j] = p.bias != nullptr ? bias_lds[min(en + j * 32 + fr, p.N - 1)] : 0.f;
        if (em + SB_M <= p.M) {
            float* const cb = p.C + (size_t)(em + wave * 32) * p.ldc + en;                 // workgroup-uniform: scalar registers
            const float* const xb = EPI == X3S_MASK ? p.aux + (size_t)(em + wave * 32) * p.ldaux + en : nullptr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (PARTIAL && j >= enj) continue;
                float mv[16];
                if constexpr (EPI == X3S_MASK) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mv[r] = (xb + (size_t)((r & 3) + 8 * (r >> 2)) * p.ldaux + j * 32)[lane_x];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = fmaf(acc[j][r], p.inv, bj[j]);
                    if constexpr (EPI == X3S_RELU) v = relu_nan(v);          // NaN-propagating (common.h)
                    if constexpr (EPI == X3S_MASK) v = mv[r] > 0.f ? v : 0.f;
#ifdef X3S_NO_STORE          // ablation build: timing only
                    if (v == 12345.678f)
#endif
                    (cb + (size_t)((r & 3) + 8 * (r >> 2)) * p.ldc + j * 32)[lane_c] = v;
                }
            }
        } else {
            const int row0 = em + wave * 32 + 4 * fg;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (PARTIAL && j >= enj) continue;
                const int col = en + j * 32 + fr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + (r & 3) + 8 * (r >> 2), rowc = min(row, p.M - 1);      // (clamped: no load under a lane-divergent branch)
                    float v = fmaf(acc[j][r], p.inv, bj[j]);
                    if constexpr (EPI == X3S_RELU) v = relu_nan(v);
                    if constexpr (EPI == X3S_MASK) v = p.aux[(size_t)rowc * p.ldaux + col] > 0.f ? v : 0.f;
                    if (row < p.M) p.C[(size_t)rowc * p.ldc + col] = v;
                }
            }
        }
    };
    bool fresh = true;                    // the next MATRIX phase is an item's first: its first MFMA group takes a zero C operand
    int pend_m = -1, pend_n = 0;          // the item whose accumulators still wait for their epilogue (written at the start of the next MEMORY phase)
    for (int c_item = 0;; ++c_item) {
        for (int kt = 0; kt < nk; ++kt) {
            // ================= MEMORY phase =================
            stamp(0);
            if (g1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // G1: its pieces of the next stage have landed (see above)
            if (pend_m >= 0) {
                epilogue(pend_m, pend_n, false);
                pend_m = -1;
                nj = PARTIAL ? min(4, (p.N - n_blk + 31) >> 5) : 4;
            }
            stamp(3);      // (epilogue, when there was one)
            const bool ahead = d_on;
            if (ahead) d_issue(buf >= 1 ? buf - 1 : SB_NST - 1);
            const char* st = smem + buf * SSTAGE_BYTES;
            f16x8 wh[2][4], wm[2][4];
            Split2h a[2];
#ifndef X3S_NO_MFMA          // ablation build: timing only
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float4 x0 = *reinterpret_cast<const float4*>(st + a_off[s][0]);
                const float4 x1 = *reinterpret_cast<const float4*>(st + a_off[s][1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wh[s][j] = *reinterpret_cast<const f16x8*>(st + b_off[s] + j * 2048);
                    wm[s][j] = *reinterpret_cast<const f16x8*>(st + b_off[s] + j * 2048 + SPLANE_BYTES);
                }
                a[s] = split2h<true>(x0, x1, p.sa);
            }
#endif
            stamp(1);      // DMA issue, fragment reads, split
            __builtin_amdgcn_sched_barrier(0);      // nothing of the MEMORY phase sinks below its barrier
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            stamp(5);      // barrier
            // ================= MATRIX phase =================
#ifndef X3S_NO_MFMA
            __builtin_amdgcn_s_setprio(1);      // the SIMD's other wave is in its MEMORY phase: its VALU / DS / DMA issue must not delay these MFMAs
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // piece-major, smallest terms first: the four accumulators take turns
                if (s == 0 && fresh) {
                    f32x16 zero;
#pragma unroll
                    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = X3S_MFMA(wm[s][j], a[s].h, zero);      // (every block, live or not: the epilogue skips the dead ones)
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (!PARTIAL || j < nj) acc[j] = X3S_MFMA(wm[s][j], a[s].h, acc[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (!PARTIAL || j < nj) acc[j] = X3S_MFMA(wh[s][j], a[s].m, acc[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (!PARTIAL || j < nj) acc[j] = X3S_MFMA(wh[s][j], a[s].h, acc[j]);
            }
            __builtin_amdgcn_s_setprio(0);
#endif
            stamp(2);      // MFMAs
            buf = buf == SB_NST - 1 ? 0 : buf + 1;
            const bool last = kt == nk - 1;
            fresh = last;
            if (last) { pend_m = m_blk; pend_n = n_blk; }
            bool more = true;
            if (last) more = im.decode(c_item + 1, m_blk, n_blk);
            if (!g1) {
                // G0: the next stage's pieces of this wave have landed - at most the stage just issued may still be in flight
                if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S_DMA) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            stamp(4);      // wait for the next stage's DMA
            if (!(g1 && last && !more)) {          // (G1 took its extra barrier up front)
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            stamp(5);
            if (last && !more) goto done;
        }
    }
done:
    if (pend_m >= 0) epilogue(pend_m, pend_n, true);
    if (X3S_TIMING && timing) {
        if (blockIdx.x == 0) for (int k = 0; k < 6; ++k) p.dbg[k] = tm[k];
        p.dbg[8 + 2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memtime() - t_begin;
        p.dbg[9 + 2 * blockIdx.x] = tm[4] + tm[5];       // waits: DMA + barriers
    }
}

}  // namespace

// Shapes this kernel takes: whole K = 32 steps, 16-byte aligned rows, and enough (256 x 128) items to give every CU one.
bool gemm_x3s_eligible(const X3Gemm& g) {
    if (g.prec != 4 || g.a_mode != X3_ROW || g.b_mode != X3_PLANES) return false;
    if (g.accumulate || g.C2 != nullptr || g.B2 != nullptr || g.n_split != 0 || g.a_colsum != nullptr) return false;
    if (g.a_bf16 || g.b_bf16 || g.c_bf16 || g.aux_bf16 || (g.relu && g.aux != nullptr)) return false;
    if (g.N > 4096) return false;                                  // the bias table in LDS
    if (g.K < SB_K || g.K % SB_K || (g.N & 31) || (g.lda & 3) || (g.ldb & 7) || (g.ldc & 3) || (g.aux != nullptr && (g.ldaux & 3))) return false;
    if ((long long)g.M * g.lda * 4 >= (1LL << 32) - (1 << 20) || 2 * g.b_plane * 2 + (long long)g.N * g.ldb * 2 >= (1LL << 32) - (1 << 20)) return false;      // 32-bit DMA offsets
    const long items = (long)((g.M + SB_M - 1) / SB_M) * ((g.N + SB_N - 1) / SB_N);
    return items >= 192;
}

template <bool PARTIAL, int EPI>
static int launch_x3s(const X3SArgs& a, const ItemMap& im, int grid, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_x3s_kernel<PARTIAL, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, X3S_LDS);
        if (e != hipSuccess) { set_error("gemm_x3s: hipFuncSetAttribute", (int)e); return (int)e; }
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_x3s_kernel<PARTIAL, EPI>), dim3(grid), dim3(SB_THREADS), X3S_LDS, stream, a, im);
    return launch_check("gemm_x3s");
}

int gemm_x3s(const X3Gemm& g, hipStream_t stream) {
    const bool partial = (g.N % SB_N) != 0;
    X3SArgs a{};
    a.A = static_cast<const float*>(g.A); a.B = static_cast<const uint16_t*>(g.B);
    a.C = g.C; a.bias = g.bias; a.aux = g.aux; a.b_plane = g.b_plane;
    a.M = g.M; a.N = g.N; a.K = g.K; a.lda = g.lda; a.ldb = g.ldb; a.ldc = g.ldc; a.ldaux = g.ldaux; a.relu = g.relu;
    a.nbias = g.bias ? g.nbias : 0;
    a.sa = g.sa; a.inv = 1.f / (g.sa * g.sb);
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    ItemMap im;
    im.mt = (g.M + SB_M - 1) / SB_M; im.nt = (g.N + SB_N - 1) / SB_N;
    im.n_items = im.mt * im.nt;
    // run = 2 only for the 128 + 32 column split of the head block (its two tiles cost 4 : 1 - one workgroup takes both); two EQUAL column
    // tiles go to two neighbouring workgroups of the XCD, which stream the same 256 activation rows at the same time
    im.run = (im.nt == 2 && g.N % SB_N != 0) ? 2 : 1;
    int grid = (im.n_items + im.run - 1) / im.run;
    if (grid > cus) grid = cus;
    im.xcd = ((im.mt & 7) == 0 && (grid & 7) == 0) ? 1 : 0;
#if X3S_TIMING
    static long long* dbg = nullptr;
    if (!dbg) (void)hipMalloc(&dbg, 8192);
    (void)hipMemsetAsync(dbg, 0, 8192, stream);
    a.dbg = dbg;
#endif
    const int epi = g.aux != nullptr ? X3S_MASK : (g.relu ? X3S_RELU : X3S_PLAIN);
    int rc;
    if (partial) rc = epi == X3S_MASK ? launch_x3s<true, X3S_MASK>(a, im, grid, stream) : (epi == X3S_RELU ? launch_x3s<true, X3S_RELU>(a, im, grid, stream) : launch_x3s<true, X3S_PLAIN>(a, im, grid, stream));
    else rc = epi == X3S_MASK ? launch_x3s<false, X3S_MASK>(a, im, grid, stream) : (epi == X3S_RELU ? launch_x3s<false, X3S_RELU>(a, im, grid, stream) : launch_x3s<false, X3S_PLAIN>(a, im, grid, stream));
    if (rc) return rc;
#if X3S_TIMING
    {
        long long h[8 + 2 * 256];
        (void)hipMemcpy(h, dbg, sizeof(long long) * (8 + 2 * (grid < 256 ? grid : 256)), hipMemcpyDeviceToHost);
        long long tmin = 1LL << 60, tmax = 0, wmin = 1LL << 60, wmax = 0; double tsum = 0, wsum = 0;
        for (int b = 0; b < grid && b < 256; ++b) {
            const long long t = h[8 + 2 * b], w = h[9 + 2 * b];
            tmin = t < tmin ? t : tmin; tmax = t > tmax ? t : tmax; tsum += (double)t;
            wmin = w < wmin ? w : wmin; wmax = w > wmax ? w : wmax; wsum += (double)w;
        }
        fprintf(stderr, "gemm_x3s workgroup totals (ticks): min %lld avg %.0f max %lld (wg0 %lld) | waits min %lld avg %.0f max %lld\n", tmin, tsum / grid, tmax, h[8], wmin, wsum / grid, wmax);
        const double items = (double)((im.n_items + grid - 1) / grid), stages = items * (g.K / SB_K);
        fprintf(stderr, "gemm_x3s M %d N %d K %d items/wg %.0f stages/item %d | clocks per stage (wave 0): loop %.0f  MEMORY phase (dma issue, reads, split) %.0f  MATRIX phase %.0f  dma wait %.0f  barriers %.0f | per item: epilogue %.0f\n",
                g.M, g.N, g.K, items, g.K / SB_K, h[0] / stages, h[1] / stages, h[2] / stages, h[4] / stages, h[5] / stages, h[3] / items);
    }
#endif
    return 0;
}

}  // namespace dc
