// LSTM / GRU with H = 256 in teams of EIGHT workgroups (taken for 65 .. 128 sequences; DC_DIMS_TEAM8 forces, DC_DIMS_TEAM4 forbids).
//
// Same job as rnn_team_mfma.hip (the S sequential cell steps of nn.LSTM / nn.GRU behind /root/reference/policy.py:66,141 in one
// launch, W_hh resident in AGPRs, four sequences of a team on the rows of the 4x4x1 f32 MFMA, 8-byte {h, tag} granules between
// the members) with the weights cut twice as fine: a member holds the gate columns of 32 hidden units (128 AGPRs per lane instead
// of 256).
//
// Why.  A step of the four-member kernels is 0.85-0.9 us of MFMA issue and ~1 us of everything else (gate math, publish, the
// hand-off's round trip through L2, barriers), and with one wave per SIMD nothing overlaps.  With 128 sequences (BASELINE.json
// configs[3]'s per-GPU shard) 32 teams of four leave half the chip idle; 32 teams of eight use every CU once, and a member's product
// is half as long: 1.5-1.6 us per step instead of 1.85-2.0.  (128 AGPRs would let TWO workgroups share a CU - which is what more than
// 128 sequences get when the flag forces these kernels - but two workgroups on a CU disturb each other: 2.5 us per step.  That, the teams
// claimed in pairs on shared CUs, and two groups in flight per team in one instruction stream were all measured and are slower than the
// four-member kernels: profiles/r04/team8_vs_team4.txt.)
//
// Two forward kernels: team8_fwd_half_kernel (the default: k split inside the wave, described at the kernel) and team8_fwd_kernel
// (the first form, DC_DIMS_TEAM_NS(2), described here).
// Forward roles (256 threads; member m of a team = hidden units [32 m, 32 m + 32), all gates):
//   product:  wave w = k quarter (k in [64 w, 64 w + 64)); lane l: gate pair hi = l >> 5 (0: i, f; 1: g, o), unit 32 m + (l & 31);
//             the lane's two gate columns over its k quarter = 128 AGPRs; FwdProduct<64>, rnn_persist.hip's product phase.
//   k quarters: every wave leaves its partial sums of all four sequences in LDS; wave w then owns SEQUENCE SLOT w and adds the four
//             quarters in index order (the sum of a cell does not depend on the slot it sits in: duplicate slots of a ragged group
//             store identical values).  v_permlane32_swap hands both wave halves all four gates: the halves compute the same cell
//             (128 cells on 256 lanes); the lower half publishes.
//   exchange: ring of rnn_team_mfma.hip (one granule per cell and step, four slots deep); lower half collects members m + 1 .. m + 4,
//             upper half m + 5 .. m + 8 (= m itself: its own granule, already there).
// Backward roles (row-parallel like rnn_team_mfma.hip: a member contracts ITS OWN 128 gate gradients with W_hh[own column][u'] for
// all 256 output units u' and the partial sums are reduce-scattered to the owners of u'):
//   product:  lane tid = output unit u'; K = 128 own columns kk = 32 gate + own unit: 128 AGPRs; BwdProduct<128>.
//   scatter:  lane u' publishes its four sums to member u' >> 5 (ring [owner][source][sequence][unit]); the owner's own go through LDS.
//   cell:     wave w = sequence slot w, unit 32 m + (l & 31); lower half adds own + sources m + 1 .. m + 4, upper half m + 5 .. m + 7,
//             v_permlane32_swap adds the halves (fixed order); both halves then compute the cell's gate gradients.
// Timeouts, roles by ticket, same-XCD hand-off: team_util.h, with eight members (four teams per XCD at one workgroup per CU).
#include <stdio.h>
#include <stdlib.h>
#include "kernels.h"
#include "persist_util.h"
#include "team_util.h"

namespace dc {
namespace {

enum { T8_H = TEAM_H, T8_M = 8, T8_US = 32, T8_KQ = 64, T8_HLD = 80, T8_THREADS = 256 };
// ring words of one team: forward [sequence slot][tag & 3][256 units]; backward [tag & 3][owner][source][sequence][32 units]
enum { T8_FWD_RING = 4 * TEAM_SLOTS * T8_H, T8_BWD_STAGE = T8_M * T8_M * 4 * T8_US, T8_BWD_RING = TEAM_SLOTS * T8_BWD_STAGE };

// float index of h[seq][k] inside one buffer of the LDS image: [k quarter][seq][T8_HLD], k in plain order inside a row; rows 80 floats
// apart: the sixteen lanes of a ds_read_b128 group (four sequences x four 16-byte pieces) start 16 floats x 0..3 + 4 floats x 0..3
// apart - sixteen different bank quads.  FwdProduct<64> hands MFMA number kk' (abid = kk' & 15 of A register kk' >> 4) the four
// consecutive floats lane 4b + i read at [i][4b ..]: it contracts k = 4 (kk' & 15) + (kk' >> 4) - the weights are loaded in that order.
__device__ __forceinline__ int t8_hpos(int seq, int k) { return ((k >> 6) * 4 + seq) * T8_HLD + (k & 63); }
__device__ __forceinline__ constexpr int t8_korder(int kk) { return 4 * (kk & 15) + (kk >> 4); }

template <int CELL>     // 1: LSTM, 0: GRU
__global__ __launch_bounds__(T8_THREADS, 2) void team8_fwd_kernel(RnnStepArgs p, u64* __restrict__ xbuf_all, int n_teams, int allow_plain) {
    constexpr bool LSTM = CELL == 1;
    constexpr int H = T8_H, G = LSTM ? 4 : 3, GH = G * H;
    __shared__ __attribute__((aligned(16))) float h_lds[2][4 * 4 * T8_HLD];
    __shared__ float2 xch[4][4][64];                   // [k quarter = source wave][sequence][lane]: partial sums of the lane's two gate columns
    __shared__ int dead;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, ul = lane & 31;
    int team, member;
    team_claim_role_m<T8_M>(reinterpret_cast<unsigned*>(xbuf_all), n_teams, team, member);
    if (team < 0) return;
    u64* const xbuf = xbuf_all + TEAM_HDR + TEAM_MAX * T8_M;
    const int plain = team_same_xcd_m<T8_M>(xbuf_all + TEAM_HDR + team * T8_M, member, allow_plain);
    const int u = T8_US * member + ul;
    const int slot = wave;                             // cell role: this wave's sequence slot (both halves: the same cell)
    if (tid == 0) dead = 0;

    // ---- weights: rows (2 hi + m) H + u of W_hh, k in [64 wave, 64 wave + 64) ----------------------------------------
    float w0[T8_KQ], w1[T8_KQ];
    {
        const bool has1 = 2 * hi + 1 < G;              // the GRU has no gate 3: zero weights in that slot
        const float* r0 = p.Whh + (size_t)((2 * hi + 0) * H + u) * H + T8_KQ * wave;
        const float* r1 = p.Whh + (size_t)((has1 ? 2 * hi + 1 : 0) * H + u) * H + T8_KQ * wave;
#pragma unroll
        for (int kk = 0; kk < T8_KQ; ++kk) { w0[kk] = r0[t8_korder(kk)]; w1[kk] = has1 ? r1[t8_korder(kk)] : 0.f; }
    }
    float bh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bh[g] = g < G ? p.bhh[g * H + u] : 0.f;

    unsigned tag = 0;                                  // the team's running step counter (continues across sequence groups)
    bool failed = false;
    const int n_groups = (p.n_seq + 3) >> 2;
    for (int grp = team; grp < n_groups && !failed; grp += n_teams) {
        int bmap[4], tmax;
        if (!map_slots(p, 4 * grp, bmap, tmax)) continue;
        const int b = slot == 0 ? bmap[0] : (slot == 1 ? bmap[1] : (slot == 2 ? bmap[2] : bmap[3]));
        const int len = p.seq_len[b];
        const unsigned row0 = (unsigned)p.seq_off[b];
        unsigned goff = row0 * GH + u, soff = row0 * H + u;
        unsigned st_g = goff, st_s = soff, st_p = soff;
        const float h0v = p.h0 ? p.h0[(size_t)b * H + u] : 0.f;
        float c = LSTM ? (p.c0 ? p.c0[(size_t)b * H + u] : 0.f) : h0v;     // the cell's carried state: c (LSTM) / h (GRU)
        float sv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // results of the last finished step: gates 0..3 (GRU: [3] = W_hn h + b_hn), c, h
        float svp0 = c, svp1 = h0v;                    // row 0 of cprev / hprev keeps c0 / h0
        for (int e = tid; e < 4 * H; e += T8_THREADS) {        // h0 of all 256 units of the four slots -> LDS buffer 0
            const int q = e >> 8, j = e & (H - 1);
            const int bq = q == 0 ? bmap[0] : (q == 1 ? bmap[1] : (q == 2 ? bmap[2] : bmap[3]));
            h_lds[0][t8_hpos(q, j)] = p.h0 ? p.h0[(size_t)bq * H + j] : 0.f;
        }
        float xc[4] = {0.f, 0.f, 0.f, 0.f}, xn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < G; ++g) xc[g] = p.gates[goff + g * H];
        u64* const xb = xbuf + (size_t)(team * 4 + slot) * (TEAM_SLOTS * H);      // ring of this wave's sequence slot
        __syncthreads();

        auto step = [&](const int t, float (&xcur)[4], float (&xnext)[4], auto CUR) {
            constexpr int cur = decltype(CUR)::value;
            const bool on = t < len, on1 = t + 1 < len;
            const unsigned gnx = goff + (on1 ? GH : 0);
            const float* const lp = p.gates + gnx;
            float* const gs = p.gates + st_g;
            float* const cs = (LSTM ? p.cseq : p.hn) + st_s;     // GRU: W_hn h + b_hn of the step (the backward's r-gate term)
            float* const hs = p.hseq + st_s;
            float* const cp = LSTM ? p.cprev + st_p : nullptr;
            float* const hp = p.hprev + st_p;
            // between the MFMA pairs: the loads of the next step's gate pre-activations, the stores of the previous step's results
            // (both wave halves: the same addresses, the same values)
            auto hook = [&](auto K) {
                constexpr int k = decltype(K)::value;          // 0 .. 31
                if constexpr (k >= 1 && k <= G) xnext[k - 1] = tm_ld(lp + (k - 1) * H);
                else if constexpr (k >= 8 && k < 8 + G) tm_st(gs + (k - 8) * H, sv[k - 8]);
                else if constexpr (k == 12) tm_st(cs, sv[LSTM ? 4 : 3]);
                else if constexpr (k == 13) tm_st(hs, sv[5]);
                else if constexpr (k == 14 && LSTM) tm_st(cp, svp0);
                else if constexpr (k == 15) tm_st(hp, svp1);
            };
            f32x4 pa[4];
            FwdProduct<T8_KQ>::run(pa, w0, w1, lds_addr(&h_lds[cur][(wave * 4 + (lane & 3)) * T8_HLD + (lane >> 2) * 4]), hook);
            const f32x4 acc0 = pa[0] + pa[2], acc1 = pa[1] + pa[3];   // gate columns 2 hi, 2 hi + 1 of the four sequences, this k quarter
            // ---- k quarters: all partial sums through LDS, every wave adds its own sequence's four in index order -------------
#pragma unroll
            for (int q = 0; q < 4; ++q) xch[wave][q][lane] = make_float2(acc0[q], acc1[q]);
            __syncthreads();
            float y0, y1;
            {
                const float2 a = xch[0][slot][lane], bq = xch[1][slot][lane], cq = xch[2][slot][lane], dq = xch[3][slot][lane];
                y0 = ((a.x + bq.x) + cq.x) + dq.x;     // gate 2 hi     of (sequence slot, unit u)
                y1 = ((a.y + bq.y) + cq.y) + dq.y;     // gate 2 hi + 1
            }
            // ---- gate pairs: every lane gets all four (lower half held i, f; upper half g, o) ------------------------------
            float x0 = y0, x1 = y1;
            half_swap(y0, x0);                         // y0 = gate 0 (i / r), x0 = gate 2 (g / n part) in all lanes
            half_swap(y1, x1);                         // y1 = gate 1 (f / z), x1 = gate 3 (o)
            // LSTM: i, f, g, o -> c' = f c + i g, h' = o tanh(c');   GRU: r, z, n (og = W_hn h + b_hn) -> h' = (1 - z) n + z h
            const float ig = fast_sigmoid(xcur[0] + (y0 + bh[0]));
            const float fg = fast_sigmoid(xcur[1] + (y1 + bh[1]));
            const float og = LSTM ? fast_sigmoid(xcur[3] + (x1 + bh[3])) : x0 + bh[2];
            const float gg = LSTM ? fast_tanh(xcur[2] + (x0 + bh[2])) : fast_tanh(xcur[2] + ig * og);
            const float cn = LSTM ? fg * c + ig * gg : (1.f - fg) * gg + fg * c;
            const float hn = LSTM ? og * fast_tanh(cn) : cn;
            const float hpub = on ? hn : 0.f;
            ++tag;
            if (hi == 0) granule_store(xb + (tag & 3) * H + u, hpub, tag, plain);      // publish first: the peers are waiting for it
            h_lds[cur ^ 1][t8_hpos(slot, u)] = hpub;
            c = on ? cn : c;
            sv[0] = on ? ig : sv[0]; sv[1] = on ? fg : sv[1]; sv[2] = on ? gg : sv[2];
            sv[3] = on ? og : sv[3]; sv[4] = on ? cn : sv[4]; sv[5] = on ? hn : sv[5];
            st_g = on ? goff : st_g;
            st_s = on ? soff : st_s;
            svp0 = on1 ? cn : svp0;
            svp1 = on1 ? hn : svp1;
            st_p = on1 ? soff + H : st_p;
            goff = gnx;
            soff += on1 ? H : 0;
            // ---- the same unit index of the other members, same sequence -> LDS (lower half: m + 1 .. m + 4, upper: m + 5 .. m + 8) ---
            if (t + 1 < tmax) {
                u64 gr[4];
                const u64* ga[4];
                int uu[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uu[j] = T8_US * ((member + 1 + 4 * hi + j) & 7) + ul;
                    ga[j] = xb + (tag & 3) * H + uu[j];
                    gr[j] = granule_load(ga[j]);
                }
                if (!granule_wait_all<4>(gr, ga, tag)) { dead = 1; team_report_timeout(p.fault, TEAM_K_T8_FWD, p.layer, team, member, t, b, tag); }
#pragma unroll
                for (int j = 0; j < 4; ++j) h_lds[cur ^ 1][t8_hpos(slot, uu[j])] = __uint_as_float((unsigned)gr[j]);
            }
            __syncthreads();
            return dead == 0;
        };
        for (int t = 0; t < tmax; t += 2) {
            if (!step(t, xc, xn, std::integral_constant<int, 0>{})) { failed = true; break; }
            if (t + 1 < tmax && !step(t + 1, xn, xc, std::integral_constant<int, 1>{})) { failed = true; break; }
        }
        // drain: the deferred stores of the group's last step
#pragma unroll
        for (int g = 0; g < G; ++g) p.gates[st_g + g * H] = sv[g];
        if constexpr (LSTM) { p.cseq[st_s] = sv[4]; p.cprev[st_p] = svp0; }
        else p.hn[st_s] = sv[3];
        p.hseq[st_s] = failed ? __builtin_nanf("") : sv[5];      // a peer never answered: make the failure visible downstream
        p.hprev[st_p] = svp1;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// Forward with the k split INSIDE the wave (the default; DC_DIMS_TEAM_NS(2) keeps the kernel above for A/B).  Above, four waves hold a k
// quarter each and meet in LDS behind a barrier.  Here a wave owns 32 gate columns (8 units x 4 gates) and its two halves contract the two
// k halves of the SAME columns (the 4x4x1 MFMA with the broadcast kept inside each half, cbsz:3 - rnn_persist.hip's BwdProduct): lane =
// (k half, gate, unit), 128 AGPRs, 128 MFMAs per wave and step as before.  The halves are added by v_permlane32_swap, the four gates of a
// cell brought together by a 4 x 4 transpose between the 8-lane groups and four registers (v_permlane16_swap + DPP): no LDS round trip, ONE
// barrier per step.  Cell role: lane group s' = (lane >> 3) & 3 -> sequence slot s', unit 32 m + 8 w + (lane & 7); both wave halves
// compute the same cell, the lower publishes; lower / upper half collect members m + 1 .. m + 4 / m + 5 .. m + 8.
// ---------------------------------------------------------------------------------------------------
// BwdProduct<128> hands MFMA number kk' (abid = kk' & 7 of A register kk' >> 3) the 16 consecutive floats lane 4b' + i read: it contracts
// half-local k = 16 (kk' & 7) + (kk' >> 3)
__device__ __forceinline__ constexpr int t8c_korder(int kk) { return 16 * (kk & 7) + (kk >> 3); }
enum { T8C_HLD = 260 };    // floats per sequence row of the LDS image (plain k order): lane (half, b', i) reads [i][128 half + 16 b' ..] as four
                           // ds_read_b128; 260 / 4 = 1 mod 16: the sixteen lanes of a read group start at 16-byte units i + 4 b' - all different

template <int CELL>     // 1: LSTM, 0: GRU
__global__ __launch_bounds__(T8_THREADS, 2) void team8_fwd_half_kernel(RnnStepArgs p, u64* __restrict__ xbuf_all, int n_teams, int allow_plain) {
    constexpr bool LSTM = CELL == 1;
    constexpr int H = T8_H, G = LSTM ? 4 : 3, GH = G * H, KH = 128;
    __shared__ __attribute__((aligned(16))) float h_lds[2][4 * T8C_HLD];
    __shared__ int dead;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5, grp = (lane >> 3) & 3, j = lane & 7;      // product role: k half, gate `grp`; cell role: sequence slot `grp`
    int team, member;
    team_claim_role_m<T8_M>(reinterpret_cast<unsigned*>(xbuf_all), n_teams, team, member);
    if (team < 0) return;
    u64* const xbuf = xbuf_all + TEAM_HDR + TEAM_MAX * T8_M;
    const int plain = team_same_xcd_m<T8_M>(xbuf_all + TEAM_HDR + team * T8_M, member, allow_plain);
    const int ul = 8 * wave + j;                       // unit inside the member's 32
    const int u = T8_US * member + ul;
    const int slot = grp;
    if (tid == 0) dead = 0;

    // ---- weights: row (gate H + u) of W_hh, k in [128 kh, 128 kh + 128), in the order the product contracts them ----
    float w[KH];
    {
        const bool has = grp < G;                      // the GRU has no gate 3: zero weights there
        const float* r0 = p.Whh + (size_t)((has ? grp : 0) * H + u) * H + KH * kh;
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) w[kk] = has ? r0[t8c_korder(kk)] : 0.f;
    }
    float bh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bh[g] = g < G ? p.bhh[g * H + u] : 0.f;

    unsigned tag = 0;
    bool failed = false;
    const int n_groups = (p.n_seq + 3) >> 2;
    for (int gq = team; gq < n_groups && !failed; gq += n_teams) {
        int bmap[4], tmax;
        if (!map_slots(p, 4 * gq, bmap, tmax)) continue;
        const int b = slot == 0 ? bmap[0] : (slot == 1 ? bmap[1] : (slot == 2 ? bmap[2] : bmap[3]));
        const int len = p.seq_len[b];
        const unsigned row0 = (unsigned)p.seq_off[b];
        unsigned goff = row0 * GH + u, soff = row0 * H + u;
        unsigned st_g = goff, st_s = soff, st_p = soff;
        const float h0v = p.h0 ? p.h0[(size_t)b * H + u] : 0.f;
        float c = LSTM ? (p.c0 ? p.c0[(size_t)b * H + u] : 0.f) : h0v;
        float sv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float svp0 = c, svp1 = h0v;
        for (int e = tid; e < 4 * H; e += T8_THREADS) {
            const int q = e >> 8, jj = e & (H - 1);
            const int bq = q == 0 ? bmap[0] : (q == 1 ? bmap[1] : (q == 2 ? bmap[2] : bmap[3]));
            h_lds[0][q * T8C_HLD + jj] = p.h0 ? p.h0[(size_t)bq * H + jj] : 0.f;
        }
        float xc[4] = {0.f, 0.f, 0.f, 0.f}, xn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < G; ++g) xc[g] = p.gates[goff + g * H];
        u64* const xb = xbuf + (size_t)(team * 4 + slot) * (TEAM_SLOTS * H);
        __syncthreads();

        auto step = [&](const int t, float (&xcur)[4], float (&xnext)[4], auto CUR) {
            constexpr int cur = decltype(CUR)::value;
            const bool on = t < len, on1 = t + 1 < len;
            const unsigned gnx = goff + (on1 ? GH : 0);
            const float* const lp = p.gates + gnx;
            float* const gs = p.gates + st_g;
            float* const cs = (LSTM ? p.cseq : p.hn) + st_s;
            float* const hs = p.hseq + st_s;
            float* const cp = LSTM ? p.cprev + st_p : nullptr;
            float* const hp = p.hprev + st_p;
            auto hook = [&](auto K) {
                constexpr int k = decltype(K)::value;          // 0 .. 31
                if constexpr (k >= 1 && k <= G) xnext[k - 1] = tm_ld(lp + (k - 1) * H);
                else if constexpr (k >= 8 && k < 8 + G) tm_st(gs + (k - 8) * H, sv[k - 8]);
                else if constexpr (k == 12) tm_st(cs, sv[LSTM ? 4 : 3]);
                else if constexpr (k == 13) tm_st(hs, sv[5]);
                else if constexpr (k == 14 && LSTM) tm_st(cp, svp0);
                else if constexpr (k == 15) tm_st(hp, svp1);
            };
            f32x4 pa[4];
            BwdProduct<KH>::run(pa, w, lds_addr(&h_lds[cur][(lane & 3) * T8C_HLD + KH * kh + ((lane >> 2) & 7) * 16]), hook);
            const f32x4 part = (pa[0] + pa[1]) + (pa[2] + pa[3]);    // gate `grp` of unit u for the four sequences, this lane's k half
            float a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {                            // k halves: lower + upper, in that order, in every lane
                float lo = part[s], hi2 = part[s];
                half_swap(lo, hi2);
                a[s] = lo + hi2;
            }
            groups8_transpose4(a, lane);                             // a[g] = gate g of (sequence `grp`, unit u)
            const float y0 = a[0], y1 = a[1], x0 = a[2], x1 = a[3];
            const float ig = fast_sigmoid(xcur[0] + (y0 + bh[0]));
            const float fg = fast_sigmoid(xcur[1] + (y1 + bh[1]));
            const float og = LSTM ? fast_sigmoid(xcur[3] + (x1 + bh[3])) : x0 + bh[2];
            const float gg = LSTM ? fast_tanh(xcur[2] + (x0 + bh[2])) : fast_tanh(xcur[2] + ig * og);
            const float cn = LSTM ? fg * c + ig * gg : (1.f - fg) * gg + fg * c;
            const float hn = LSTM ? og * fast_tanh(cn) : cn;
            const float hpub = on ? hn : 0.f;
            ++tag;
            if (kh == 0) granule_store(xb + (tag & 3) * H + u, hpub, tag, plain);
            h_lds[cur ^ 1][slot * T8C_HLD + u] = hpub;
            c = on ? cn : c;
            sv[0] = on ? ig : sv[0]; sv[1] = on ? fg : sv[1]; sv[2] = on ? gg : sv[2];
            sv[3] = on ? og : sv[3]; sv[4] = on ? cn : sv[4]; sv[5] = on ? hn : sv[5];
            st_g = on ? goff : st_g;
            st_s = on ? soff : st_s;
            svp0 = on1 ? cn : svp0;
            svp1 = on1 ? hn : svp1;
            st_p = on1 ? soff + H : st_p;
            goff = gnx;
            soff += on1 ? H : 0;
            if (t + 1 < tmax) {
                u64 gr[4];
                const u64* ga[4];
                int uu[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uu[q] = T8_US * ((member + 1 + 4 * kh + q) & 7) + ul;
                    ga[q] = xb + (tag & 3) * H + uu[q];
                    gr[q] = granule_load(ga[q]);
                }
                if (!granule_wait_all<4>(gr, ga, tag)) { dead = 1; team_report_timeout(p.fault, TEAM_K_T8_FWD, p.layer, team, member, t, b, tag); }
#pragma unroll
                for (int q = 0; q < 4; ++q) h_lds[cur ^ 1][slot * T8C_HLD + uu[q]] = __uint_as_float((unsigned)gr[q]);
            }
            __syncthreads();
            return dead == 0;
        };
        for (int t = 0; t < tmax; t += 2) {
            if (!step(t, xc, xn, std::integral_constant<int, 0>{})) { failed = true; break; }
            if (t + 1 < tmax && !step(t + 1, xn, xc, std::integral_constant<int, 1>{})) { failed = true; break; }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) p.gates[st_g + g * H] = sv[g];
        if constexpr (LSTM) { p.cseq[st_s] = sv[4]; p.cprev[st_p] = svp0; }
        else p.hn[st_s] = sv[3];
        p.hseq[st_s] = failed ? __builtin_nanf("") : sv[5];
        p.hprev[st_p] = svp1;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// backward through time (row-parallel, see the header and rnn_team_mfma.hip)
// ---------------------------------------------------------------------------------------------------
// LDS image of the A operand: per sequence eight blocks of 16 floats (one per broadcast group b'), 20 floats apart, rows 208 floats
// apart: the sixteen lanes of a ds_read_b128 group (four sequences x four blocks) start 16 bytes x {4 i + 5 b'} apart - sixteen
// different residues mod 16, no bank conflicts.
enum { T8B_KH = 128, T8B_BLK = 20, T8B_GLD = 208 };
__device__ __forceinline__ int t8b_pos(int kk) { return T8B_BLK * (kk >> 4) + (kk & 15); }
// BwdProduct<128> hands MFMA number kk' (abid = kk' & 7 of A register kk' >> 3) the 16 consecutive floats lane 4b' + i read at
// [i][16 b' ..]: with the image in plain order it contracts kk = 16 (kk' & 7) + (kk' >> 3); the weights are loaded in that order
__device__ __forceinline__ constexpr int t8b_korder(int kk) { return 16 * (kk & 7) + (kk >> 3); }

template <int CELL>     // 1: LSTM, 0: GRU (contracts dgh = d(W_hh h + b_hh) of step t + 1; writes dgx and dgh)
__global__ __launch_bounds__(T8_THREADS, 2) void team8_bwd_kernel(RnnStepArgs p, u64* __restrict__ xbuf_all, int n_teams, int allow_plain) {
    constexpr bool LSTM = CELL == 1;
    constexpr int H = T8_H, G = LSTM ? 4 : 3, GH = G * H, KH = T8B_KH;
    __shared__ __attribute__((aligned(16))) float g_lds[2][4 * T8B_GLD];      // own gate gradients [seq][pos(kk)], kk = 32 gate + own unit
    __shared__ float own[4][T8_US];                                            // own partial sums dh_rec[seq][own unit]
    __shared__ int dead;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, ul = lane & 31;
    int team, member;
    team_claim_role_m<T8_M>(reinterpret_cast<unsigned*>(xbuf_all), n_teams, team, member);
    if (team < 0) return;
    u64* const xbuf = xbuf_all + TEAM_HDR + TEAM_MAX * T8_M;
    const int plain = team_same_xcd_m<T8_M>(xbuf_all + TEAM_HDR + team * T8_M, member, allow_plain);
    const int up = tid;                                // product role: output unit u' = tid, owner member = tid >> 5
    const int owner = tid >> 5;
    const int slot = wave;                             // cell role: sequence slot, own unit (both wave halves: the same cell)
    const int u = T8_US * member + ul;
    if (tid == 0) dead = 0;

    // ---- weights: W_hh[(kk >> 5) H + 32 m + (kk & 31)][u'], kk = 0 .. 127 ---------------------------------------------
    float w[KH];
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) {
        const int k = t8b_korder(kk);                  // own gate column 32 gate + own unit (GRU: slot 3 is empty)
        w[kk] = (k >> 5) < G ? p.Whh[(size_t)((k >> 5) * H + T8_US * member + (k & 31)) * H + up] : 0.f;
    }

    u64* const ring0 = xbuf + (size_t)team * T8_BWD_RING;
    unsigned tag = 0;
    bool failed = false;
    const int n_groups = (p.n_seq + 3) >> 2;
    for (int grp = team; grp < n_groups && !failed; grp += n_teams) {
        int bmap[4], tmax;
        if (!map_slots(p, 4 * grp, bmap, tmax)) continue;
        const int b = slot == 0 ? bmap[0] : (slot == 1 ? bmap[1] : (slot == 2 ? bmap[2] : bmap[3]));
        const int len = p.seq_len[b];
        const unsigned row = (unsigned)p.seq_off[b] + (unsigned)min(tmax - 1, len - 1);
        unsigned goff = row * GH + u, soff = row * H + u, st_g = goff;
        float dc_next = 0.f, f_next = 0.f;             // LSTM: dc, f of step t + 1;  GRU: total dh, z of step t + 1 (the direct path z h)
        float cur_v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, nxt_v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // LSTM: i, f, g, o, c, c_prev, dh;  GRU: r, z, n, W_hn h + b_hn, -, h_prev, dh
        float sv[4] = {0.f, 0.f, 0.f, 0.f};            // gate gradients of the last finished step, stored one step late
        float svh2 = 0.f;                              // GRU: the n column of dgh (= sv[2] * r; the r, z columns equal dgx's)
#pragma unroll
        for (int g = 0; g < G; ++g) cur_v[g] = p.gates[goff + g * H];
        if constexpr (LSTM) { cur_v[4] = p.cseq[soff]; cur_v[5] = p.cprev[soff]; }
        else { cur_v[3] = p.hn[soff]; cur_v[5] = p.hprev[soff]; }
        cur_v[6] = p.dh[soff];
        for (int e = tid; e < 4 * T8B_GLD; e += T8_THREADS) g_lds[0][e] = 0.f;      // "step tmax" has no gradient
        __syncthreads();

        auto step = [&](const int t, float (&cv)[7], float (&nv)[7], auto CUR) {
            constexpr int cur = decltype(CUR)::value;
            const bool on = t < len, has_next = t + 1 < len, dec = t < len && t > 0;      // dec: the row below is next
            const bool xchg = t + 1 < tmax;                                               // workgroup- and team-uniform
            const unsigned gnx = goff - (dec ? GH : 0), snx = soff - (dec ? H : 0);
            const float* const lg = p.gates + gnx;
            const float* const lc = (LSTM ? p.cseq : p.hn) + snx;
            const float* const lcp = (LSTM ? p.cprev : p.hprev) + snx;
            const float* const ldh = p.dh + snx;
            float* const gs = p.dgx + st_g;
            float* const ghs = LSTM ? nullptr : p.dgh + st_g;
            auto hook = [&](auto K) {
                constexpr int k = decltype(K)::value;          // 0 .. 31
                if constexpr (k >= 1 && k <= G) nv[k - 1] = tm_ld(lg + (k - 1) * H);
                else if constexpr (k == 5) nv[LSTM ? 4 : 3] = tm_ld(lc);
                else if constexpr (k == 6) nv[5] = tm_ld(lcp);
                else if constexpr (k == 7) nv[6] = tm_ld(ldh);
                else if constexpr (k >= 10 && k < 10 + G) tm_st(gs + (k - 10) * H, sv[k - 10]);
                else if constexpr (!LSTM && (k == 14 || k == 15)) tm_st(ghs + (k - 14) * H, sv[k - 14]);
                else if constexpr (!LSTM && k == 16) tm_st(ghs + 2 * H, svh2);
            };
            // ---- partial dh_rec[seq 0..3][u'] over this member's 128 gate columns ------------------------------------------
            f32x4 pa[4];
            BwdProduct<KH>::run(pa, w, lds_addr(&g_lds[cur][(lane & 3) * T8B_GLD + ((lane >> 2) & 7) * T8B_BLK]), hook);
            const f32x4 acc = (pa[0] + pa[1]) + (pa[2] + pa[3]);
            ++tag;
            u64* const ring = ring0 + (size_t)(tag & 3) * T8_BWD_STAGE;
            if (xchg) {
                if (owner == member) {                                 // my own units: through LDS
#pragma unroll
                    for (int q = 0; q < 4; ++q) own[q][ul] = acc[q];
                } else {                                               // the owner's: [owner][source = member][seq][unit]
                    u64* dst = ring + ((size_t)(owner * T8_M + member) * 4) * T8_US + ul;
#pragma unroll
                    for (int q = 0; q < 4; ++q) granule_store(dst + q * T8_US, acc[q], tag, plain);
                }
            }
            __syncthreads();
            float rec = 0.f;
            if (xchg) {
                // lower half: own + sources m + 1 .. m + 4; upper half: m + 5 .. m + 7 (its fourth address repeats the third, weight 0)
                u64 gr[4];
                const u64* ga[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int src = (member + 1 + 4 * hi + (hi && j == 3 ? 2 : j)) & 7;
                    ga[j] = ring + ((size_t)(member * T8_M + src) * 4 + slot) * T8_US + ul;
                    gr[j] = granule_load(ga[j]);
                }
                float part = hi ? 0.f : own[slot][ul];
                if (!granule_wait_all<4>(gr, ga, tag)) { dead = 1; team_report_timeout(p.fault, TEAM_K_T8_BWD, p.layer, team, member, t, b, tag); }
#pragma unroll
                for (int j = 0; j < 4; ++j) part += (hi && j == 3) ? 0.f : __uint_as_float((unsigned)gr[j]);
                float p0 = part, p1 = part;
                half_swap(p0, p1);                     // p0 = the lower half's sum, p1 = the upper half's, in all lanes
                rec = p0 + p1;
            }
            float dh = cv[6];
            dh += has_next ? rec : 0.f;
            const float ig = cv[0], fg = cv[1], gg = cv[2], og = cv[3];
            float dgr[4];                                  // what the next product contracts: d(W_hh h + b_hh) of this step
            if constexpr (LSTM) {
                const float tc = fast_tanh(cv[4]);
                float dcv = dh * og * (1.f - tc * tc);
                dcv += has_next ? dc_next * f_next : 0.f;
                dgr[0] = on ? dcv * gg * ig * (1.f - ig) : 0.f; dgr[1] = on ? dcv * cv[5] * fg * (1.f - fg) : 0.f;
                dgr[2] = on ? dcv * ig * (1.f - gg * gg) : 0.f; dgr[3] = on ? dh * tc * og * (1.f - og) : 0.f;
                sv[0] = on ? dgr[0] : sv[0]; sv[1] = on ? dgr[1] : sv[1]; sv[2] = on ? dgr[2] : sv[2]; sv[3] = on ? dgr[3] : sv[3];
                dc_next = on ? dcv : dc_next;
            } else {                                       // r = ig, z = fg, n = gg, og = W_hn h + b_hn, cv[5] = h_{t-1}  (rnn.hip)
                dh += has_next ? dc_next * f_next : 0.f;   // direct path h_{t+1} = ... + z_{t+1} h_t
                const float dn_pre = dh * (1.f - fg) * (1.f - gg * gg);
                const float dz_pre = dh * (cv[5] - gg) * fg * (1.f - fg);
                const float dr_pre = dn_pre * og * ig * (1.f - ig);
                dgr[0] = on ? dr_pre : 0.f; dgr[1] = on ? dz_pre : 0.f; dgr[2] = on ? dn_pre * ig : 0.f; dgr[3] = 0.f;
                sv[0] = on ? dr_pre : sv[0]; sv[1] = on ? dz_pre : sv[1]; sv[2] = on ? dn_pre : sv[2];
                svh2 = on ? dn_pre * ig : svh2;
                dc_next = on ? dh : dc_next;
            }
            if (hi == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) g_lds[cur ^ 1][slot * T8B_GLD + t8b_pos(T8_US * g + ul)] = dgr[g];
            }
            st_g = on ? goff : st_g;
            f_next = on ? fg : f_next;
            goff = gnx;
            soff = snx;
            __syncthreads();
            return dead == 0;
        };
        for (int t = tmax - 1; t >= 0; t -= 2) {
            if (!step(t, cur_v, nxt_v, std::integral_constant<int, 0>{})) { failed = true; break; }
            if (t - 1 >= 0 && !step(t - 1, nxt_v, cur_v, std::integral_constant<int, 1>{})) { failed = true; break; }
        }
        // drain the deferred stores of step 0
#pragma unroll
        for (int g = 0; g < G; ++g) p.dgx[st_g + g * H] = failed ? __builtin_nanf("") : sv[g];
        if constexpr (!LSTM) { p.dgh[st_g] = sv[0]; p.dgh[st_g + H] = sv[1]; p.dgh[st_g + 2 * H] = svh2; }
        __syncthreads();
    }
}

// teams that can be resident together: two 256-thread workgroups per CU, eight per team
int team8_capacity() {
    constexpr int MAXDEV = 64;
    static int cap_of[MAXDEV];             // 0 = not queried yet
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return 0;
    if (cap_of[dev] == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        int t = 2 * cus / T8_M;
        t = t > TEAM_MAX ? TEAM_MAX : t;
        cap_of[dev] = t >= 8 ? (t & ~7) : (t > 0 ? t : -1);
    }
    return cap_of[dev] > 0 ? cap_of[dev] : 0;
}

// teams to launch: as rnn_team_mfma.hip's plan (a multiple of 8 teams keeps every team on one XCD)
int team8_plan(int n_seq) {
    const int cap = team8_capacity();
    const int groups = (n_seq + 3) / 4;
    const int nt_f = groups < cap ? groups : cap;
    const int nt_q = nt_f >= 8 ? (nt_f & ~7) : nt_f;
    const int rounds_f = (groups + nt_f - 1) / nt_f, rounds_q = (groups + nt_q - 1) / nt_q;
    return (nt_q != nt_f && rounds_f * 1.25 < rounds_q) ? nt_f : nt_q;
}

}  // namespace

long long rnn_team8_xbuf_bytes() {
    return (long long)(((size_t)TEAM_HDR + (size_t)TEAM_MAX * T8_M + (size_t)TEAM_MAX * T8_BWD_RING) * sizeof(u64));
}

// Forced by DC_DIMS_TEAM8 (any number of sequences); by itself where it measured faster than the alternatives (profiles/r04/
// team8_vs_team4.txt, 256 steps, forward / backward): 65 .. 128 sequences - every team gets a CU per member, ONE workgroup per CU -
// 423 / 421 us against 473 / 498 (VALU team forward, four-member MFMA backward).  With more sequences two workgroups share a CU and
// disturb each other (256 sequences: 643 / 708 us against 534 / 532 for the four-member kernels); with 64 or fewer the VALU team
// kernels with one sequence per team are as fast (390 / 450).  DC_DIMS_TEAM4 keeps it off.
bool rnn_team8_supported(int cell, int H, int n_seq, int flags) {
    if (!((cell == 0 || cell == 1) && H == T8_H && n_seq >= 1) || (flags & (DC_DIMS_TEAM_VALU | DC_DIMS_TEAM4)) || team8_capacity() < 1) return false;
    if (flags & DC_DIMS_TEAM8) return true;
    return n_seq > 64 && (n_seq + 3) / 4 <= team8_capacity() / 2;
}

int rnn_team8_forward(int cell, RnnStepArgs a, int max_len, hipStream_t s) {
    u64* xb = static_cast<u64*>(a.xbuf);
    if (!xb) { set_error("rnn_team8_forward: no exchange buffer (RnnStepArgs::xbuf)", 1012); return 1012; }
    const int nt = team8_plan(a.n_seq);
    const double G = cell == 1 ? 4.0 : 3.0;
    ProfScope prof(cell == 1 ? "lstm_fwd_team" : "gru_fwd_team", 2.0 * a.n_seq * G * a.H * a.H * max_len, 4.0 * a.n_seq * max_len * a.H * (2.0 * G + 4.0), s);
    if (int rc = zero_async(xb, ((size_t)TEAM_HDR + (size_t)TEAM_MAX * T8_M + (size_t)nt * T8_FWD_RING) * sizeof(u64), s)) return rc;
    const bool quarters = ((a.flags >> DC_DIMS_TEAM_NS_SHIFT) & 7) == 2;    // DC_DIMS_TEAM_NS(2): the forward with k quarters across the waves (A/B)
    const dim3 grid(nt * T8_M), block(T8_THREADS);
    const int allow = !(a.flags & DC_DIMS_TEAM_DEVICE_SCOPE);
    if (quarters) {
        if (cell == 1) hipLaunchKernelGGL(team8_fwd_kernel<1>, grid, block, 0, s, a, xb, nt, allow);
        else hipLaunchKernelGGL(team8_fwd_kernel<0>, grid, block, 0, s, a, xb, nt, allow);
    } else {
        if (cell == 1) hipLaunchKernelGGL(team8_fwd_half_kernel<1>, grid, block, 0, s, a, xb, nt, allow);
        else hipLaunchKernelGGL(team8_fwd_half_kernel<0>, grid, block, 0, s, a, xb, nt, allow);
    }
    return launch_check("rnn_team8_forward");
}

int rnn_team8_backward(int cell, RnnStepArgs a, int max_len, hipStream_t s) {
    u64* xb = static_cast<u64*>(a.xbuf);
    if (!xb) { set_error("rnn_team8_backward: no exchange buffer (RnnStepArgs::xbuf)", 1012); return 1012; }
    const int nt = team8_plan(a.n_seq);
    const double G = cell == 1 ? 4.0 : 3.0;
    ProfScope prof(cell == 1 ? "lstm_bwd_team" : "gru_bwd_team", 2.0 * a.n_seq * G * a.H * a.H * max_len, 4.0 * a.n_seq * max_len * a.H * (3.0 * G + 6.0), s);
    if (int rc = zero_async(xb, ((size_t)TEAM_HDR + (size_t)TEAM_MAX * T8_M + (size_t)nt * T8_BWD_RING) * sizeof(u64), s)) return rc;
    if (cell == 1) hipLaunchKernelGGL(team8_bwd_kernel<1>, dim3(nt * T8_M), dim3(T8_THREADS), 0, s, a, xb, nt, !(a.flags & DC_DIMS_TEAM_DEVICE_SCOPE));
    else hipLaunchKernelGGL(team8_bwd_kernel<0>, dim3(nt * T8_M), dim3(T8_THREADS), 0, s, a, xb, nt, !(a.flags & DC_DIMS_TEAM_DEVICE_SCOPE));
    return launch_check("rnn_team8_backward");
}

}  // namespace dc
