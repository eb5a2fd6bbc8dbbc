// Persistent LSTM: ALL time steps of a layer in ONE launch, recurrent weights stationary in registers.
//
// Replaces the S sequential cell steps of nn.LSTM (the BASELINE.json extension of
// /root/reference/policy.py:66,141) and their BPTT (/root/reference/optimizer.py:672) for H <= 128,
// where W_hh (4H x H fp32 = 256 KB at H = 128) fits the register file of one CU: 4 waves x 64 lanes
// x 256 registers.  Sequences are independent (optimizer.py:591 stacks them as batch rows), so a
// workgroup owns FOUR sequences for the whole trajectory and never talks to another workgroup: no
// launch boundary (1.5 us each, 2 x 256 of them per epoch before), no grid barrier, no weight
// re-fetch.  64 trajectories -> 16 workgroups, 256 -> 64.
//
// Matrix instruction: v_mfma_f32_4x4x1_16b_f32 = sixteen independent 4x4 outer products per issue
// (K = 1), exact fp32, 64 FLOP/clk/SIMD like every f32 MFMA.  Block b = lanes 4b..4b+3;
//   A: lane 4b+i holds A_b[i]     B: lane 4b+j holds B_b[j]     D: lane 4b+j, register i = D_b[i][j].
// Row i = sequence (all blocks get the same four rows), column = whatever weight column the lane
// keeps - so each lane simply owns output columns, with its 2 x H (forward) / 2H (backward) weights
// held in registers for the whole launch.  The state h_t (4 x H) / the gate gradients (4 x 4H) make
// one trip through LDS per step (double-buffered, one barrier per step).
//
// Lane roles (wave w, lane l, hi = l >> 5): hidden unit u = 32 w + (l & 31).
//   forward : lanes hi=0 accumulate the i,f gate columns of u, lanes hi=1 the g,o columns; one
//             cross-half exchange later every lane finishes two (sequence, unit) cells:
//             sequences 2*hi and 2*hi+1.  c_t of a cell lives in that lane's registers throughout.
//   backward: lane accumulates dh_rec[seq][u] over gate columns [2H*hi, 2H*hi+2H), halves are summed
//             across the wave halves, same cell ownership; dc_{t+1} and f_{t+1} stay in registers.
#include <stdio.h>
#include <stdlib.h>
#include <utility>
#include "kernels.h"
#include "persist_util.h"

namespace dc {

// ---------------------------------------------------------------------------------------------------
// forward.  gates[row][4H] holds W_ih x + b_ih on entry and the activated gates i,f,g,o on exit.
// hprev/cprev[first row of a sequence] hold h0/c0 (rnn_seed_state); h_t, c_t go to hseq/cseq[row] and
// to hprev/cprev[row+1] (the shifted copies the weight-gradient GEMM and the backward read).
//
// Per step the critical path is product -> exchange -> gate maths -> h to LDS -> barrier.  Everything
// else is issued from the hooks between MFMA pairs: the loads of next step's gate pre-activations and
// the stores of the PREVIOUS step's results, which wait in registers for one step.  A cell past its
// sequence end keeps re-storing its last results to the same rows (idempotent) and re-loading its last
// row - no branches anywhere in the step.  Steps are unrolled by two with ping-pong registers, so the
// prefetched values are never copied.
// ---------------------------------------------------------------------------------------------------
template <int H, bool TIMING = false, int HOOKMODE = 0>   // HOOKMODE (timing experiments only): 1 no hook work, 2 loads only, 3 stores only
__global__ __launch_bounds__(PersistCfg<H>::THREADS, 1) void lstm_fwd_persist_kernel(RnnStepArgs p) {
    long long tm0 = 0, tm_prod = 0, tm_gate = 0, tm_bar = 0;   // TIMING: s_memtime phase sums (shader cycles)
    using C = PersistCfg<H>;
    __shared__ __attribute__((aligned(16))) float h_lds[2][4][C::HLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5;
    const int u = 32 * wave + (lane & 31);
    int bmap[4], tmax;
    if (!map_slots(p, blockIdx.x * 4, bmap, tmax)) return;

    // ---- weights: column (2*hi + m)*H + u of W_hh, all k, for m = 0,1 ------------------------------
    float w0[H], w1[H];
    {
        const float4* r0 = reinterpret_cast<const float4*>(p.Whh + (size_t)((2 * hi + 0) * H + u) * H);
        const float4* r1 = reinterpret_cast<const float4*>(p.Whh + (size_t)((2 * hi + 1) * H + u) * H);
#pragma unroll
        for (int k = 0; k < H / 4; ++k) {
            const float4 x = r0[k], y = r1[k];
            w0[4 * k] = x.x; w0[4 * k + 1] = x.y; w0[4 * k + 2] = x.z; w0[4 * k + 3] = x.w;
            w1[4 * k] = y.x; w1[4 * k + 1] = y.y; w1[4 * k + 2] = y.z; w1[4 * k + 3] = y.w;
        }
    }
    float bh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bh[g] = p.bhh[g * H + u];

    // ---- this lane's two cells: sequences slot 2*hi and 2*hi+1 --------------------------------------
    int len[2];
    float c[2];
    unsigned goff[2], soff[2];          // element offsets of the CURRENT row: gates[row][u], state[row][u]
    unsigned st_g[2], st_s[2], st_p[2]; // rows the deferred stores go to (frozen once the cell is done)
    float sv[2][6], svp[2][2];          // deferred: i,f,g,o,c,h of the last finished step / c,h for row+1
    auto init_cell = [&](auto CC) {
        constexpr int cc = decltype(CC)::value;
        const int b = hi ? bmap[2 + cc] : bmap[cc];
        len[cc] = p.seq_len[b];
        const unsigned row0 = (unsigned)p.seq_off[b];
        goff[cc] = row0 * (4 * H) + u;
        soff[cc] = row0 * H + u;
        c[cc] = p.cprev[soff[cc]];
        st_g[cc] = goff[cc]; st_s[cc] = soff[cc]; st_p[cc] = soff[cc];
#pragma unroll
        for (int q = 0; q < 6; ++q) sv[cc][q] = 0.f;      // step 0 "stores" these over rows it rewrites at step 1
        svp[cc][0] = c[cc];                               // row 0 of cprev/hprev keeps c0/h0
        svp[cc][1] = p.hprev[soff[cc]];
    };
    init_cell(std::integral_constant<int, 0>{});
    init_cell(std::integral_constant<int, 1>{});
    // h0 -> LDS buffer 0
    for (int e = tid; e < 4 * H; e += C::THREADS) {
        const int q = e / H, j = e - q * H;
        const int bq = q == 0 ? bmap[0] : (q == 1 ? bmap[1] : (q == 2 ? bmap[2] : bmap[3]));   // static indices: no scratch
        (&h_lds[0][0][0])[bcast_pos<H / 16, C::HLD>(q, j)] = p.hprev[(size_t)p.seq_off[bq] * H + j];
    }
    float xc[2][4], xn[2][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { xc[0][g] = p.gates[goff[0] + g * H]; xc[1][g] = p.gates[goff[1] + g * H]; }
    __syncthreads();

    // one step; xcur = this step's gate pre-activations, xnext receives the next step's
    // UNI: all four slots have the workgroup's full length (the training case: equal chunks) - every
    // "is this cell still running" select folds away
    auto step = [&](const int t, float (&xcur)[2][4], float (&xnext)[2][4], auto CUR, auto UNI_) {
        constexpr int cur = decltype(CUR)::value;
        constexpr bool UNI = decltype(UNI_)::value;
        // (indices of the per-cell integer state are compile-time constants everywhere - cells are visited
        // through integral_constant, not loops: hipcc leaves loop-indexed copies of them in scratch)
        const bool on1_0 = UNI ? t + 1 < tmax : t + 1 < len[0], on1_1 = UNI ? t + 1 < tmax : t + 1 < len[1];
        const unsigned gnx[2] = {goff[0] + (on1_0 ? 4 * H : 0), goff[1] + (on1_1 ? 4 * H : 0)};
        // all addresses of the step up front (they issue while the first LDS read is in flight); the
        // hooks are then bare memory instructions with immediate offsets
        const float* const lp[2] = {p.gates + gnx[0], p.gates + gnx[1]};
        float* const gs[2] = {p.gates + st_g[0], p.gates + st_g[1]};
        float* const cs[2] = {p.cseq + st_s[0], p.cseq + st_s[1]};
        float* const hs[2] = {p.hseq + st_s[0], p.hseq + st_s[1]};
        float* const cp[2] = {p.cprev + st_p[0], p.cprev + st_p[1]};
        float* const hp[2] = {p.hprev + st_p[0], p.hprev + st_p[1]};
        auto hook = [&](auto K) {
            constexpr int k = decltype(K)::value;
            constexpr int SP = FwdProduct<H>::HOOKS >= 52 ? 2 : 1;   // 25 items, one per SP hooks
            if constexpr (k % SP == 0 && k / SP < 25) {
                constexpr int it = k / SP;
                if constexpr (it == 0) {
                } else if constexpr (it <= 8) {
                    constexpr int cc = (it - 1) >> 2, gg = (it - 1) & 3;
                    if constexpr (HOOKMODE == 0 || HOOKMODE == 2) xnext[cc][gg] = lp[cc][gg * H];
                    else xnext[cc][gg] = 0.f;
                } else if constexpr (HOOKMODE == 1 || HOOKMODE == 2) {
                } else if constexpr (it <= 16) {
                    constexpr int cc = (it - 9) >> 2, gg = (it - 9) & 3;
                    gs[cc][gg * H] = sv[cc][gg];
                } else if constexpr (it <= 18) {
                    *cs[it - 17] = sv[it - 17][4];
                } else if constexpr (it <= 20) {
                    *hs[it - 19] = sv[it - 19][5];
                } else if constexpr (it <= 22) {
                    *cp[it - 21] = svp[it - 21][0];
                } else {
                    *hp[it - 23] = svp[it - 23][1];
                }
            }
        };

        // ---- recurrent product: rows = the 4 sequences, this lane's two columns ----------------------
        f32x4 pa[4];
        if constexpr (TIMING) tm0 = __builtin_amdgcn_s_memtime();
        FwdProduct<H>::run(pa, w0, w1, lds_addr(&h_lds[cur][lane & 3][(lane >> 2) * (H / 16)]), hook);
        if constexpr (TIMING) { const long long x = __builtin_amdgcn_s_memtime(); tm_prod += x - tm0; tm0 = x; }
        f32x4 acc0 = pa[0] + pa[2], acc1 = pa[1] + pa[3];   // lanes hi=0: (i,f) of 4 sequences; hi=1: (g,o)

        // ---- cross-half exchange: afterwards every lane holds i,f,g,o of its own two cells ------------
        float ri[2], rf[2], rg[2], ro[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            float y0 = acc0[cc], x0 = acc0[2 + cc], y1 = acc1[cc], x1 = acc1[2 + cc];
            half_swap(y0, x0);   // y0 = i of own cell, x0 = g of own cell
            half_swap(y1, x1);   // y1 = f,             x1 = o
            ri[cc] = y0; rg[cc] = x0; rf[cc] = y1; ro[cc] = x1;
        }
        auto cell = [&](auto CC, const bool on1c) {
            constexpr int cc = decltype(CC)::value;
            const bool on = UNI ? true : t < len[cc];
            const float ig = fast_sigmoid(xcur[cc][0] + (ri[cc] + bh[0]));
            const float fg = fast_sigmoid(xcur[cc][1] + (rf[cc] + bh[1]));
            const float gg = fast_tanh(xcur[cc][2] + (rg[cc] + bh[2]));
            const float og = fast_sigmoid(xcur[cc][3] + (ro[cc] + bh[3]));
            const float cn = fg * c[cc] + ig * gg;
            const float hn = og * fast_tanh(cn);
            (&h_lds[cur ^ 1][0][0])[bcast_pos<H / 16, C::HLD>(2 * hi + cc, u)] = on ? hn : 0.f;
            c[cc] = on ? cn : c[cc];
            // results of an active step replace the deferred ones; a finished cell keeps its last set
            sv[cc][0] = on ? ig : sv[cc][0]; sv[cc][1] = on ? fg : sv[cc][1]; sv[cc][2] = on ? gg : sv[cc][2];
            sv[cc][3] = on ? og : sv[cc][3]; sv[cc][4] = on ? cn : sv[cc][4]; sv[cc][5] = on ? hn : sv[cc][5];
            st_g[cc] = on ? goff[cc] : st_g[cc];
            st_s[cc] = on ? soff[cc] : st_s[cc];
            svp[cc][0] = on1c ? cn : svp[cc][0];
            svp[cc][1] = on1c ? hn : svp[cc][1];
            st_p[cc] = on1c ? soff[cc] + H : st_p[cc];
            goff[cc] = gnx[cc];
            soff[cc] += on1c ? H : 0;
        };
        cell(std::integral_constant<int, 0>{}, on1_0);
        cell(std::integral_constant<int, 1>{}, on1_1);
        if constexpr (TIMING) { const long long x = __builtin_amdgcn_s_memtime(); tm_gate += x - tm0; tm0 = x; }
        __syncthreads();
        if constexpr (TIMING) { const long long x = __builtin_amdgcn_s_memtime(); tm_bar += x - tm0; }
    };

    const bool uni = p.seq_len[bmap[0]] == tmax && p.seq_len[bmap[1]] == tmax && p.seq_len[bmap[2]] == tmax &&
                     p.seq_len[bmap[3]] == tmax;
    if (uni) {
        for (int t = 0; t < tmax; t += 2) {
            step(t, xc, xn, std::integral_constant<int, 0>{}, std::true_type{});
            if (t + 1 < tmax) step(t + 1, xn, xc, std::integral_constant<int, 1>{}, std::true_type{});
        }
    } else {
        for (int t = 0; t < tmax; t += 2) {
            step(t, xc, xn, std::integral_constant<int, 0>{}, std::false_type{});
            if (t + 1 < tmax) step(t + 1, xn, xc, std::integral_constant<int, 1>{}, std::false_type{});
        }
    }
    // drain: the deferred stores of the last step
    auto drain = [&](auto CC) {
        constexpr int cc = decltype(CC)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) p.gates[st_g[cc] + g * H] = sv[cc][g];
        p.cseq[st_s[cc]] = sv[cc][4];
        p.hseq[st_s[cc]] = sv[cc][5];
        p.cprev[st_p[cc]] = svp[cc][0];
        p.hprev[st_p[cc]] = svp[cc][1];
    };
    drain(std::integral_constant<int, 0>{});
    drain(std::integral_constant<int, 1>{});
    if constexpr (TIMING) {
        if (tid == 0 && blockIdx.x == 0 && p.dbg != nullptr) { p.dbg[0] = tm_prod; p.dbg[1] = tm_gate; p.dbg[2] = tm_bar; p.dbg[3] = tmax; }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward through time.  In: dh[row][H] = dL/dh_t from the layer above (heads), the forward's saved
// gates / cseq / cprev.  Out: dgx[row][4H] = gradient w.r.t. the gate pre-activations (for an LSTM the
// same tensor serves W_ih x + b_ih and W_hh h + b_hh).  Same hook discipline as the forward: the
// operands of step t-1 are loaded, and the gate gradients of step t+1 stored, between the MFMA pairs.
// A cell whose sequence is shorter than the workgroup's longest idles at its last row until the sweep
// reaches it (stores there are overwritten by its first real step).
// ---------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(PersistCfg<H>::THREADS, 1) void lstm_bwd_persist_kernel(RnnStepArgs p) {
    using C = PersistCfg<H>;
    constexpr int KH = 2 * H;   // gate columns per wave half
    __shared__ __attribute__((aligned(16))) float g_lds[2][4][C::GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5;
    const int u = 32 * wave + (lane & 31);
    int bmap[4], tmax;
    if (!map_slots(p, blockIdx.x * 4, bmap, tmax)) return;

    // ---- weights: W_hh[KH*hi + kk][u], kk = 0..KH-1 --------------------------------------------------
    float w[KH];
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) w[kk] = p.Whh[(size_t)(KH * hi + kk) * H + u];

    int len[2];
    unsigned goff[2], soff[2], st_g[2];
    float dc_next[2] = {0.f, 0.f}, f_next[2] = {0.f, 0.f};
    float cur_v[2][7], nxt_v[2][7];   // i, f, g, o, c, c_prev, dh
    float sv[2][4];                   // gate gradients of the last finished step, stored one step late
    auto init_cell = [&](auto CC) {
        constexpr int cc = decltype(CC)::value;
        const int b = hi ? bmap[2 + cc] : bmap[cc];
        len[cc] = p.seq_len[b];
        const unsigned row = (unsigned)p.seq_off[b] + (unsigned)min(tmax - 1, len[cc] - 1);
        goff[cc] = row * (4 * H) + u;
        soff[cc] = row * H + u;
        st_g[cc] = goff[cc];
#pragma unroll
        for (int g = 0; g < 4; ++g) { cur_v[cc][g] = p.gates[goff[cc] + g * H]; sv[cc][g] = 0.f; }
        cur_v[cc][4] = p.cseq[soff[cc]];
        cur_v[cc][5] = p.cprev[soff[cc]];
        cur_v[cc][6] = p.dh[soff[cc]];
    };
    init_cell(std::integral_constant<int, 0>{});
    init_cell(std::integral_constant<int, 1>{});
    for (int e = tid; e < 4 * C::GLD; e += C::THREADS) (&g_lds[0][0][0])[e] = 0.f;   // "step tmax" has no gradient
    __syncthreads();

    auto step = [&](const int t, float (&cv)[2][7], float (&nv)[2][7], auto CUR, auto UNI_) {
        constexpr int cur = decltype(CUR)::value;
        constexpr bool UNI = decltype(UNI_)::value;
        const bool dec0 = UNI ? t > 0 : (t < len[0] && t > 0), dec1 = UNI ? t > 0 : (t < len[1] && t > 0);   // row below is next
        const unsigned gnx[2] = {goff[0] - (dec0 ? 4 * H : 0), goff[1] - (dec1 ? 4 * H : 0)};
        const unsigned snx[2] = {soff[0] - (dec0 ? H : 0), soff[1] - (dec1 ? H : 0)};
        const float* const lg[2] = {p.gates + gnx[0], p.gates + gnx[1]};
        const float* const lc[2] = {p.cseq + snx[0], p.cseq + snx[1]};
        const float* const lcp[2] = {p.cprev + snx[0], p.cprev + snx[1]};
        const float* const ldh[2] = {p.dh + snx[0], p.dh + snx[1]};
        float* const gs[2] = {p.dgx + st_g[0], p.dgx + st_g[1]};
        auto hook = [&](auto K) {
            constexpr int k = decltype(K)::value;
            constexpr int SP = BwdProduct<KH>::HOOKS >= 48 ? 2 : 1;   // 23 items
            if constexpr (k % SP == 0 && k / SP < 23) {
                constexpr int it = k / SP;
                if constexpr (it == 0) {
                } else if constexpr (it <= 14) {
                    constexpr int cc = (it - 1) / 7, q = (it - 1) % 7;
                    if constexpr (q < 4) nv[cc][q] = lg[cc][q * H];
                    else if constexpr (q == 4) nv[cc][4] = *lc[cc];
                    else if constexpr (q == 5) nv[cc][5] = *lcp[cc];
                    else nv[cc][6] = *ldh[cc];
                } else {
                    constexpr int cc = (it - 15) >> 2, gg = (it - 15) & 3;
                    gs[cc][gg * H] = sv[cc][gg];
                }
            }
        };

        // ---- dh_rec[seq][u] = sum_k dgates_{t+1}[seq][k] * W_hh[k][u], this half's k range ------------
        f32x4 pa[4];
        BwdProduct<KH>::run(pa, w, lds_addr(&g_lds[cur][lane & 3][KH * hi + ((lane >> 2) & 7) * BwdProduct<KH>::NJ]), hook);
        const f32x4 acc = (pa[0] + pa[1]) + (pa[2] + pa[3]);
        float rec[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            float y = acc[cc], x = acc[2 + cc];
            half_swap(y, x);     // low lanes: own acc[cc] | partner's acc[cc]; high lanes: partner's acc[2+cc] | own
            rec[cc] = y + x;
        }

        auto cell = [&](auto CC) {
            constexpr int cc = decltype(CC)::value;
            const bool on = UNI ? true : t < len[cc];
            const bool has_next = UNI ? (t + 1) < tmax : (t + 1) < len[cc];
            float dh = cv[cc][6];
            dh += has_next ? rec[cc] : 0.f;
            const float ig = cv[cc][0], fg = cv[cc][1], gg = cv[cc][2], og = cv[cc][3];
            const float tc = fast_tanh(cv[cc][4]);
            float dcv = dh * og * (1.f - tc * tc);
            dcv += has_next ? dc_next[cc] * f_next[cc] : 0.f;
            const float di = dcv * gg * ig * (1.f - ig);
            const float df = dcv * cv[cc][5] * fg * (1.f - fg);
            const float dg = dcv * ig * (1.f - gg * gg);
            const float dO = dh * tc * og * (1.f - og);
            // gate column g*H+u -> wave half (col / KH), position inside that half's broadcast-ordered block
            float* gl = &g_lds[cur ^ 1][2 * hi + cc][0];
            auto gpos = [&](int col) { return BwdProduct<KH>::pos(col); };
            gl[gpos(u)] = on ? di : 0.f; gl[gpos(H + u)] = on ? df : 0.f;
            gl[gpos(2 * H + u)] = on ? dg : 0.f; gl[gpos(3 * H + u)] = on ? dO : 0.f;
            sv[cc][0] = on ? di : sv[cc][0]; sv[cc][1] = on ? df : sv[cc][1];
            sv[cc][2] = on ? dg : sv[cc][2]; sv[cc][3] = on ? dO : sv[cc][3];
            st_g[cc] = on ? goff[cc] : st_g[cc];
            dc_next[cc] = on ? dcv : dc_next[cc];
            f_next[cc] = on ? fg : f_next[cc];
            goff[cc] = gnx[cc];
            soff[cc] = snx[cc];
        };
        cell(std::integral_constant<int, 0>{});
        cell(std::integral_constant<int, 1>{});
        __syncthreads();
    };

    const bool uni = p.seq_len[bmap[0]] == tmax && p.seq_len[bmap[1]] == tmax && p.seq_len[bmap[2]] == tmax &&
                     p.seq_len[bmap[3]] == tmax;
    if (uni) {
        for (int t = tmax - 1; t >= 0; t -= 2) {
            step(t, cur_v, nxt_v, std::integral_constant<int, 0>{}, std::true_type{});
            if (t - 1 >= 0) step(t - 1, nxt_v, cur_v, std::integral_constant<int, 1>{}, std::true_type{});
        }
    } else {
        for (int t = tmax - 1; t >= 0; t -= 2) {
            step(t, cur_v, nxt_v, std::integral_constant<int, 0>{}, std::false_type{});
            if (t - 1 >= 0) step(t - 1, nxt_v, cur_v, std::integral_constant<int, 1>{}, std::false_type{});
        }
    }
    // drain the deferred stores of step 0
#pragma unroll
    for (int g = 0; g < 4; ++g) { p.dgx[st_g[0] + g * H] = sv[0][g]; p.dgx[st_g[1] + g * H] = sv[1][g]; }
}

bool lstm_persist_supported(int H) { return H == 64 || H == 128; }

int lstm_forward_persist(RnnStepArgs a, int max_len, hipStream_t s) {
    const dim3 grid((a.n_seq + 3) / 4);
    ProfScope prof("lstm_fwd_persist", 2.0 * a.n_seq * 4.0 * a.H * a.H * max_len,
                   4.0 * a.n_seq * max_len * a.H * (2.0 * 4 + 4.0), s);
    if (lstm_persist_use_valu(a.n_seq, a.flags)) return lstm_forward_valu(a, s);
    static long long* dbg = nullptr;
    constexpr bool timing = DC_DEV_TIMING != 0;
    if (timing && a.H == 128) {   // debugging aid: per-phase cycle counts of workgroup 0, printed per launch
        if (!dbg) (void)hipMalloc(&dbg, 64);
        a.dbg = dbg;
        constexpr int mode = DC_DEV_HOOKMODE;
        if (mode == 1) hipLaunchKernelGGL((lstm_fwd_persist_kernel<128, true, 1>), grid, dim3(PersistCfg<128>::THREADS), 0, s, a);
        else if (mode == 2) hipLaunchKernelGGL((lstm_fwd_persist_kernel<128, true, 2>), grid, dim3(PersistCfg<128>::THREADS), 0, s, a);
        else if (mode == 3) hipLaunchKernelGGL((lstm_fwd_persist_kernel<128, true, 3>), grid, dim3(PersistCfg<128>::THREADS), 0, s, a);
        else hipLaunchKernelGGL((lstm_fwd_persist_kernel<128, true>), grid, dim3(PersistCfg<128>::THREADS), 0, s, a);
        long long h[4];
        (void)hipMemcpy(h, dbg, 32, hipMemcpyDeviceToHost);
        fprintf(stderr, "lstm_fwd timing: steps %lld  product %.0f  gates %.0f  barrier %.0f cycles/step\n", h[3],
                (double)h[0] / h[3], (double)h[1] / h[3], (double)h[2] / h[3]);
        return launch_check("lstm_forward_persist");
    }
    if (a.H == 128) hipLaunchKernelGGL((lstm_fwd_persist_kernel<128>), grid, dim3(PersistCfg<128>::THREADS), 0, s, a);
    else if (a.H == 64) hipLaunchKernelGGL((lstm_fwd_persist_kernel<64>), grid, dim3(PersistCfg<64>::THREADS), 0, s, a);
    else { set_error("lstm_forward_persist: unsupported hidden size", 1011); return 1011; }
    return launch_check("lstm_forward_persist");
}

int lstm_backward_persist(RnnStepArgs a, int max_len, hipStream_t s) {
    const dim3 grid((a.n_seq + 3) / 4);
    ProfScope prof("lstm_bwd_persist", 2.0 * a.n_seq * 4.0 * a.H * a.H * max_len,
                   4.0 * a.n_seq * max_len * a.H * (2.0 * 4 + 3.0), s);
    if (lstm_persist_use_valu(a.n_seq, a.flags)) return lstm_backward_valu(a, s);
    if (a.H == 128) hipLaunchKernelGGL((lstm_bwd_persist_kernel<128>), grid, dim3(PersistCfg<128>::THREADS), 0, s, a);
    else if (a.H == 64) hipLaunchKernelGGL((lstm_bwd_persist_kernel<64>), grid, dim3(PersistCfg<64>::THREADS), 0, s, a);
    else { set_error("lstm_backward_persist: unsupported hidden size", 1011); return 1011; }
    return launch_check("lstm_backward_persist");
}

}  // namespace dc
