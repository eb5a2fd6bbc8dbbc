// Persistent LSTM, latency variant: ONE sequence per workgroup, recurrent product on the packed f32
// VALU (v_pk_fma_f32) instead of the matrix cores.
//
// Same contract as rnn_persist.hip (the BASELINE.json LSTM extension of /root/reference/policy.py:66,141
// and its BPTT, /root/reference/optimizer.py:672): all time steps of a layer in one launch, W_hh
// stationary in registers, one trip of the state through LDS and one barrier per step.
//
// Why a second variant.  On gfx950 the f32 MFMA and the packed f32 VALU have the SAME rate
// (64 FLOP/clk/SIMD, MI355X_MICROARCH.md): the matrix instruction buys nothing for a product whose
// "batch" dimension is the handful of sequences a workgroup owns.  The MFMA variant needs four
// sequences per workgroup to fill the 4x4 blocks, so a step costs 4 x 4H x H MACs on one CU =
// 2048 issue cycles at H = 128, and the bench batch (64 trajectories, BASELINE.json configs[1]) runs on
// 16 of the 256 CUs.  Here a workgroup owns one sequence: 4H x H MACs = 512 issue cycles per step on
// 64 CUs.  Per-step latency is what a recurrence is bound by, so this is the variant to use while
// sequences <= CUs; above ~2 sequences per CU the MFMA variant's 4-sequence packing wins again
// (lstm_forward_persist picks).
//
// Lane roles, 4H threads (H = 128: 8 waves, two per SIMD, 128 weights + ~50 live registers each).
// In BOTH kernels thread tid finishes the step holding gate q = tid & 3 of hidden unit u = tid >> 2,
// so the four gates of a cell sit in one DPP quad and the cell maths is quad broadcasts.
//   forward : the 16 lanes of a DPP row split k (H/16 each, their slice of h_{t-1} = one or two
//             ds_read_b128), and each lane accumulates the row's 16 gate columns (4 units x 4 gates)
//             as 8 packed pairs: H/16 x 8 v_pk_fma_f32 with h[k] broadcast by op_sel.  A
//             reduce-scatter over the row (row_ror:8, row_half_mirror, two quad_perms: 8+4+2+1
//             v_add_f32_dpp) leaves each lane with the complete sum of ONE column.  The
//             register -> column assignment is permuted per lane at weight-load time (colmap) so that
//             every stage adds register c+half of the partner into register c - no selects at run time.
//   backward: dh_rec = W_hh^T dgates has a 4H-long contraction, so the 64 lanes of a wave split k
//             (the 4 gate gradients of one or two cells = one or two ds_read_b128) and each wave owns
//             16 outputs: again H/16 x 8 v_pk_fma_f32.  Reduce-scatter: v_permlane32_swap (8 pairs),
//             v_permlane16_swap (4 pairs), row_ror:8, row_half_mirror, then a quad all-reduce, which
//             lands output 16*wave + (lane >> 2) in all four lanes of its quad.
// The index logic of both trees is checked on the CPU by tests/test_host_logic.py (numpy emulation of
// the lane permutations) and on the GPU against the MFMA variant and the oracle.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <utility>
#include "kernels.h"
#include "valu_util.h"

namespace dc {
namespace {

constexpr int PF = 4;   // global loads run this many steps ahead (L2/MALL latency ~ 2 steps)

}  // namespace

// ---------------------------------------------------------------------------------------------------
// forward.  gates[row][4H]: W_ih x + b_ih on entry, activated i,f,g,o on exit; hprev/cprev[first row] <-
// h0/c0 (written here); h_t, c_t -> hseq/cseq[row] and hprev/cprev[row+1] (exactly lstm_fwd_persist_kernel's contract).
// ---------------------------------------------------------------------------------------------------
template <int H, bool TIMING = false>   // TIMING (DC_LSTM_TIMING=1): s_memtime phase sums of wave 3 of workgroup 0 -> p.dbg
__global__ __launch_bounds__(4 * H) void lstm_fwd_valu_kernel(RnnStepArgs p) {
    long long tm[4] = {0, 0, 0, 0}, tm0 = 0;
    auto stamp = [&](int i) {
        if constexpr (TIMING) { const long long x = __builtin_amdgcn_s_memtime(); tm[i] += x - tm0; tm0 = x; }
    };
    constexpr int KPL = H / 16;   // k per lane
    constexpr int NRD = KPL / 4;  // ds_read_b128 per step
    __shared__ __attribute__((aligned(16))) float h_lds[2][H];
    const int tid = threadIdx.x;
    const int kg = tid & 15, jg = tid >> 4;
    const int q = tid & 3, u = tid >> 2;
    const int b = blockIdx.x;
    const int len = p.seq_len[b];
    if (len <= 0) return;
    const size_t row0 = (size_t)p.seq_off[b];

    // ---- weights: pair m = registers 2m, 2m+1; element kk <-> k = 64*(kk>>2) + 4*kg + (kk&3) ----------
    f32x2 wp[8][KPL];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = colmap(kg, 2 * m + e);
            const float* src = p.Whh + (size_t)((c & 3) * H + 4 * jg + (c >> 2)) * H + 4 * kg;
#pragma unroll
            for (int i = 0; i < NRD; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(src + 64 * i);
                wp[m][4 * i + 0][e] = v.x; wp[m][4 * i + 1][e] = v.y; wp[m][4 * i + 2][e] = v.z; wp[m][4 * i + 3][e] = v.w;
            }
        }
    }
    const float bq = p.bhh[q * H + u];
    const bool is_g = q == 2;                                     // the tanh gate
    const float sc = is_g ? -2.8853900817779268f : -1.4426950408889634f;
    const float am = is_g ? 2.f : 1.f, aa = is_g ? -1.f : 0.f;   // act = am * rcp(1 + exp2(sc * x)) + aa
    // initial state: read from h0/c0 (zeros when absent) and written to the first row's hprev/cprev, where the backward and
    // the dW_hh product expect it
    float c = p.c0 ? p.c0[(size_t)b * H + u] : 0.f;
    if (q == 0) p.cprev[row0 * H + u] = c;
    if (tid < H) {
        const float h = p.h0 ? p.h0[(size_t)b * H + tid] : 0.f;
        h_lds[0][tid] = h;
        p.hprev[row0 * H + tid] = h;
    }
    float* const gp = p.gates + row0 * (4 * H) + q * H + u;      // this lane's gate column, row 0
    float* const sA = ((q & 1) ? p.cseq : p.hseq) + row0 * H + u;
    float* const sB = ((q & 1) ? p.cprev : p.hprev) + row0 * H + u + H;
    // gate pre-activations of the CURRENT group of PF steps; the next group's are loaded at the top of a
    // group and waited for at its end (see the loop)
    float xc[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) xc[j] = gp[(size_t)min(j, len - 1) * (4 * H)];
    asm volatile("" : "+v"(xc[0]), "+v"(xc[1]), "+v"(xc[2]), "+v"(xc[3]) : : "memory");
    __syncthreads();

    auto step = [&](const int t, const float x) {
        const float* hl = &h_lds[t & 1][4 * kg];
        f32x2 hv[KPL / 2];           // (h[k], h[k+1]) pairs: the broadcast of either half is an op_sel, not a move
#pragma unroll
        for (int i = 0; i < NRD; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(hl + 64 * i);
            hv[2 * i] = __builtin_shufflevector(v, v, 0, 1);
            hv[2 * i + 1] = __builtin_shufflevector(v, v, 2, 3);
        }
        stamp(0);      // barrier release .. h slice in registers
        f32x2 acc[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = pk_mul_bcast0(wp[m][0], hv[0]);
#pragma unroll
        for (int kk = 1; kk < KPL; ++kk)
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (kk & 1) pk_fma_bcast<1>(acc[m], wp[m][kk], hv[kk >> 1]);
                else pk_fma_bcast<0>(acc[m], wp[m][kk], hv[kk >> 1]);
            }
        float a[16];
#pragma unroll
        for (int m = 0; m < 8; ++m) { a[2 * m] = acc[m].x; a[2 * m + 1] = acc[m].y; }
        stamp(1);      // packed FMAs
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) a[cc] += dpp<DPP_ROR8>(a[8 + cc]);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) a[cc] += dpp<DPP_HALF_MIRROR>(a[4 + cc]);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) a[cc] += dpp<DPP_XOR2>(a[2 + cc]);
        a[0] += dpp<DPP_XOR1>(a[1]);
        // ---- this lane's gate, then the cell (all four lanes of the quad compute it) ------------------
        const float pre = a[0] + (x + bq);
        const float act = __builtin_fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(sc * pre)), am, aa);
        const float ig = dpp<DPP_Q0>(act), fg = dpp<DPP_Q1>(act), gg = dpp<DPP_Q2>(act), og = dpp<DPP_Q3>(act);
        const float cn = fg * c + ig * gg;
        const float hn = og * tanh_hw(cn);
        c = cn;
        if (q == 0) h_lds[(t & 1) ^ 1][u] = hn;
        gp[(size_t)t * (4 * H)] = act;
        // lanes q = 0,1: h_t, c_t -> hseq/cseq[row]; q = 2,3: -> hprev/cprev[row+1], or (last step) the same
        // value to the same address as lanes 0,1 - no divergent branch around a memory instruction
        float* const dst = (q >= 2 && t + 1 < len) ? sB : sA;
        dst[(size_t)t * H] = (q & 1) ? cn : hn;
        stamp(2);      // reduce + cell + stores
        __syncthreads();
        stamp(3);      // barrier
    };
    int t = 0;
    if constexpr (TIMING) tm0 = __builtin_amdgcn_s_memtime();
    for (; t + PF <= len; t += PF) {
        float xn[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) xn[j] = gp[(size_t)min(t + PF + j, len - 1) * (4 * H)];
        step(t, xc[0]);
        step(t + 1, xc[1]);
        step(t + 2, xc[2]);
        step(t + 3, xc[3]);
#pragma unroll
        for (int j = 0; j < PF; ++j) xc[j] = xn[j];
        // the waits for this group's loads happen HERE, a whole group after their issue and with an exact
        // count (only the group's stores are behind them; vmcnt retires in order) - not at the loop top, where
        // the pre-header's freshly issued loads would force vmcnt(0) on every iteration
        asm volatile("" : "+v"(xc[0]), "+v"(xc[1]), "+v"(xc[2]), "+v"(xc[3]) : : "memory");
    }
    if (t < len) {
        step(t, xc[0]);
        if (t + 1 < len) {
            step(t + 1, xc[1]);
            if (t + 2 < len) step(t + 2, xc[2]);
        }
    }
    if constexpr (TIMING) {
        if (blockIdx.x == 0 && tid == 192 && p.dbg != nullptr) { for (int i = 0; i < 4; ++i) p.dbg[i] = tm[i]; p.dbg[4] = len; }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward through time.  In: dh[row][H] (from above), the forward's activated gates / cseq / cprev.
// Out: dgx[row][4H] (exactly lstm_bwd_persist_kernel's contract).
// ---------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(4 * H) void lstm_bwd_valu_kernel(RnnStepArgs p) {
    constexpr int KPL = H / 16;   // gate columns per lane = 4H / 64
    constexpr int NRD = KPL / 4;
    __shared__ __attribute__((aligned(16))) float g_lds[2][4 * H];   // position 4*unit + gate
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = tid & 3, u = tid >> 2;
    const int Q = (lane >> 2) & 3;
    const int b = blockIdx.x;
    const int len = p.seq_len[b];
    if (len <= 0) return;
    const size_t row0 = (size_t)p.seq_off[b];

    // ---- weights.  Register r <-> output 16*wave + ((r & 12) | ((r & 3) ^ Q)); element kk <-> LDS position
    // 256*(kk>>2) + 4*lane + (kk&3) = gate (kk&3) of unit 64*(kk>>2) + lane.  The 16 outputs are 64
    // contiguous bytes of a W_hh row; the ^Q permutation inside each float4 is two conditional swaps.
    f32x2 wp[8][KPL];
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
        const int col = (kk & 3) * H + 64 * (kk >> 2) + lane;
        const float4* src = reinterpret_cast<const float4*>(p.Whh + (size_t)col * H + 16 * wave);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 v = src[g4];
            // register 4*g4 + j holds component j ^ Q
            const float s0 = (Q & 1) ? v.y : v.x, s1 = (Q & 1) ? v.x : v.y, s2 = (Q & 1) ? v.w : v.z, s3 = (Q & 1) ? v.z : v.w;
            const float r0 = (Q & 2) ? s2 : s0, r1 = (Q & 2) ? s3 : s1, r2 = (Q & 2) ? s0 : s2, r3 = (Q & 2) ? s1 : s3;
            wp[2 * g4][kk][0] = r0; wp[2 * g4][kk][1] = r1; wp[2 * g4 + 1][kk][0] = r2; wp[2 * g4 + 1][kk][1] = r3;
        }
    }
    const float* const gp = p.gates + row0 * (4 * H) + q * H + u;
    const float* const shp = (q == 0 ? p.cseq : (q == 1 ? p.cprev : p.dh)) + row0 * H + u;   // lane 3: dh again (unused)
    float* const dgp = p.dgx + row0 * (4 * H) + q * H + u;
    float aoc[PF], shc[PF];       // current group's operands (see the forward)
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const size_t r = (size_t)max(len - 1 - j, 0);
        aoc[j] = gp[r * (4 * H)];
        shc[j] = shp[r * H];
    }
    asm volatile("" : "+v"(aoc[0]), "+v"(aoc[1]), "+v"(aoc[2]), "+v"(aoc[3]), "+v"(shc[0]), "+v"(shc[1]), "+v"(shc[2]),
                 "+v"(shc[3]) : : "memory");
    const bool is_q0 = q == 0, is_q1 = q == 1, is_q2 = q == 2, is_q3 = q == 3;
    float dc_next = 0.f, f_next = 0.f;
    g_lds[0][tid] = 0.f;          // "step len" has no gate gradient
    __syncthreads();

    auto step = [&](const int t, const int cur, const float a_own, const float shv) {
        const float* gl = &g_lds[cur][4 * lane];
        f32x2 dv[KPL / 2];
#pragma unroll
        for (int i = 0; i < NRD; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(gl + 256 * i);
            dv[2 * i] = __builtin_shufflevector(v, v, 0, 1);
            dv[2 * i + 1] = __builtin_shufflevector(v, v, 2, 3);
        }
        f32x2 acc[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = pk_mul_bcast0(wp[m][0], dv[0]);
#pragma unroll
        for (int kk = 1; kk < KPL; ++kk)
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (kk & 1) pk_fma_bcast<1>(acc[m], wp[m][kk], dv[kk >> 1]);
                else pk_fma_bcast<0>(acc[m], wp[m][kk], dv[kk >> 1]);
            }
        float a[16];
#pragma unroll
        for (int m = 0; m < 8; ++m) { a[2 * m] = acc[m].x; a[2 * m + 1] = acc[m].y; }
        float s8[8], s4[4];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) s8[cc] = swap32_sum(a[cc], a[8 + cc]);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) s4[cc] = swap16_sum(s8[cc], s8[4 + cc]);
        s4[0] += dpp<DPP_ROR8>(s4[2]);
        s4[1] += dpp<DPP_ROR8>(s4[3]);
        float rec = s4[0] + dpp<DPP_HALF_MIRROR>(s4[1]);
        rec += dpp<DPP_XOR1>(rec);
        rec += dpp<DPP_XOR2>(rec);          // dh_rec[u]: zero at the sequence's last step (g_lds starts zeroed)
        // ---- cell ------------------------------------------------------------------------------------
        const float ig = dpp<DPP_Q0>(a_own), fg = dpp<DPP_Q1>(a_own), gg = dpp<DPP_Q2>(a_own), og = dpp<DPP_Q3>(a_own);
        const float cs = dpp<DPP_Q0>(shv), cp = dpp<DPP_Q1>(shv), dhx = dpp<DPP_Q2>(shv);
        const float dh = dhx + rec;
        const float tc = tanh_hw(cs);
        const float dcv = dh * og * (1.f - tc * tc) + dc_next * f_next;
        // i: dc*g*i(1-i)   f: dc*c_prev*f(1-f)   g: dc*i*(1-g^2)   o: dh*tanh(c)*o(1-o)
        const float M = is_q3 ? dh * tc : dcv;
        float X = is_q2 ? ig : 1.f;      // selects on loop-invariant lane masks, no branches
        X = is_q1 ? cp : X;
        X = is_q0 ? gg : X;
        const float D = __builtin_fmaf(-a_own, a_own, is_q2 ? 1.f : a_own);
        const float d = M * X * D;
        g_lds[cur ^ 1][tid] = d;
        dgp[(size_t)t * (4 * H)] = d;
        dc_next = dcv;
        f_next = fg;
        __syncthreads();
    };
    int i = 0;   // step index from the end: t = len - 1 - i
    for (; i + PF <= len; i += PF) {
        float aon[PF], shn[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const size_t r = (size_t)max(len - 1 - i - PF - j, 0);
            aon[j] = gp[r * (4 * H)];
            shn[j] = shp[r * H];
        }
        step(len - 1 - i, 0, aoc[0], shc[0]);
        step(len - 2 - i, 1, aoc[1], shc[1]);
        step(len - 3 - i, 0, aoc[2], shc[2]);
        step(len - 4 - i, 1, aoc[3], shc[3]);
#pragma unroll
        for (int j = 0; j < PF; ++j) { aoc[j] = aon[j]; shc[j] = shn[j]; }
        asm volatile("" : "+v"(aoc[0]), "+v"(aoc[1]), "+v"(aoc[2]), "+v"(aoc[3]), "+v"(shc[0]), "+v"(shc[1]), "+v"(shc[2]),
                     "+v"(shc[3]) : : "memory");
    }
    if (i < len) {
        step(len - 1 - i, 0, aoc[0], shc[0]);
        if (i + 1 < len) {
            step(len - 2 - i, 1, aoc[1], shc[1]);
            if (i + 2 < len) step(len - 3 - i, 0, aoc[2], shc[2]);
        }
    }
}

// DC_DIMS_LSTM_MFMA / DC_DIMS_LSTM_VALU force a variant (A/B measurements, parity tests of both); default: by size
bool lstm_persist_use_valu(int n_seq, int flags) {
    if (flags & DC_DIMS_LSTM_MFMA) return false;
    if (flags & DC_DIMS_LSTM_VALU) return true;
    return n_seq <= 512;   // <= 2 sequences per CU: one-sequence workgroups; above, the 4-sequence MFMA packing
}

int lstm_forward_valu(RnnStepArgs a, hipStream_t s) {
    constexpr bool timing = DC_DEV_TIMING != 0;
    if (timing && a.H == 128) {   // debugging aid: per-step phase cycles of one wave, printed per launch
        static long long* dbg = nullptr;
        if (!dbg) (void)hipMalloc(&dbg, 64);
        a.dbg = dbg;
        hipLaunchKernelGGL((lstm_fwd_valu_kernel<128, true>), dim3(a.n_seq), dim3(512), 0, s, a);
        long long h[5];
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        const double st = (double)h[4];
        fprintf(stderr, "lstm_fwd_valu timing (cycles per step, %lld steps): h read %.0f  packed FMAs %.0f  reduce+cell+stores %.0f  barrier %.0f\n",
                h[4], h[0] / st, h[1] / st, h[2] / st, h[3] / st);
        return launch_check("lstm_forward_valu");
    }
    if (a.H == 128) hipLaunchKernelGGL((lstm_fwd_valu_kernel<128>), dim3(a.n_seq), dim3(512), 0, s, a);
    else if (a.H == 64) hipLaunchKernelGGL((lstm_fwd_valu_kernel<64>), dim3(a.n_seq), dim3(256), 0, s, a);
    else { set_error("lstm_forward_valu: unsupported hidden size", 1011); return 1011; }
    return launch_check("lstm_forward_valu");
}

int lstm_backward_valu(RnnStepArgs a, hipStream_t s) {
    if (a.H == 128) hipLaunchKernelGGL((lstm_bwd_valu_kernel<128>), dim3(a.n_seq), dim3(512), 0, s, a);
    else if (a.H == 64) hipLaunchKernelGGL((lstm_bwd_valu_kernel<64>), dim3(a.n_seq), dim3(256), 0, s, a);
    else { set_error("lstm_backward_valu: unsupported hidden size", 1011); return 1011; }
    return launch_check("lstm_backward_valu");
}

}  // namespace dc
