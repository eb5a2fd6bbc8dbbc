// LSTM with H = 256 (BASELINE.json configs[2] / [3]) for batches of more than ~128 sequences: the team scheme of
// rnn_team.hip - a sequence's W_hh spread over the registers of FOUR workgroups that exchange the state every step -
// with the arithmetic of rnn_persist.hip: a team advances FOUR sequences together, one v_mfma_f32_4x4x1_16b_f32 product
// phase per time step, instead of giving each of its sequences a turn on the packed-f32 VALU.
//
// Replaces the S sequential cell steps of nn.LSTM (the BASELINE.json extension of /root/reference/policy.py:66,141) and of the
// reference's own nn.GRU (CELL = 0: three gates r, z, n in the same four-slot layout, zero weights in slot 3 - a step is bound by the
// hand-off, not by the product; n = tanh(W_in x + b_in + r (W_hn h + b_hn)), h' = (1 - z) n + z h, torch 1.0 cell maths as in rnn.hip).
//
// Why.  rnn_team.hip at 256 sequences (64 teams, four sequences each, one "step call" per sequence in turn): 2 100
// cycles per call - 1 080 of packed-FMA arithmetic with its DPP reduction trees, the rest poll, barrier and loop control -
// i.e. 8 400 cycles per time step of the four sequences.  The 4x4x1 MFMA takes the four sequences as the four ROWS of
// every block, so the same 4 x 65 536 MACs of a member are 1 024 MFMAs = 2 048 matrix-pipe cycles per SIMD, with one
// poll, two barriers and one loop iteration per time step.  What is left exposed is one hand-off per step (publish ->
// visible -> read, ~0.7 us): four sequences per team leave nothing to hide it behind, and eight would halve the number
// of teams.
//
// Roles of the FIRST forward form (team_mfma_fwd_kernel, now behind DC_DIMS_TEAM_NS(2); the default since round 4 is
// team_mfma_fwd_col_kernel further down, without the k split) - 256 threads, one workgroup per CU, member m of a team = hidden units
// [64m, 64m + 64), all four gates:
//   wave w: k half kh = w >> 1 (k in [128 kh, 128 kh + 128)), unit block ub = w & 1;  lane l: gate pair hi = l >> 5
//   (0: i, f; 1: g, o), unit u = 64 m + 32 ub + (l & 31).  A lane keeps the two gate columns of its unit over its k half
//   in 256 AGPRs - rnn_persist.hip's register budget and its FwdProduct<128> product phase, unchanged.
//   After the product: the two k halves swap the partial sums of the sequences the OTHER half owns (one 16-byte LDS
//   round trip), the gate pairs swap within the wave (v_permlane32_swap), and every lane holds the four gates of ONE cell:
//   unit u of sequence slot 2 kh + hi.  256 lanes = 64 units x 4 sequences.
// Exchange: the protocol of rnn_team.hip (team_util.h) - one 8-byte {h, tag} granule per cell per step, ring of four
// slots, roles by ticket, L2-scope stores when the four members share an XCD, bounded spins.  A lane publishes its cell's h
// and collects the same unit index of the three other members for the same sequence.
#include <stdio.h>
#include <stdlib.h>
#include "kernels.h"
#include "persist_util.h"
#include "team_util.h"

namespace dc {
namespace {


enum { TM_H = TEAM_H, TM_KH = 128, TM_HLD = TM_KH + 4, TM_THREADS = 256 };

// float index of h[seq][k] inside one buffer of the LDS image: [k half][seq][TM_HLD], k in PLAIN order inside a row.  The
// product phase (FwdProduct<128>) hands MFMA number kk' (block broadcast abid = kk' & 15 of A register kk' >> 4) the eight
// consecutive floats lane 4b+i read at [i][8b ..]: it contracts k = 8 (kk' & 15) + (kk' >> 4) - so the WEIGHTS are loaded
// into the registers in that order (tm_korder), once, and the per-step LDS stores of h (consecutive lanes = consecutive
// units) are consecutive words instead of the 8-word stride of rnn_persist.hip's broadcast order (measured there: 48 % of
// the LDS cycles were bank conflicts).
__device__ __forceinline__ int tm_hpos(int seq, int k) { return ((k >> 7) * 4 + seq) * TM_HLD + (k & 127); }
__device__ __forceinline__ constexpr int tm_korder(int kk) { return 8 * (kk & 15) + (kk >> 4); }

template <int CELL>     // 1: LSTM, 0: GRU
__global__ __launch_bounds__(TM_THREADS, 1) void team_mfma_fwd_kernel(RnnStepArgs p, u64* __restrict__ xbuf_all, int n_teams, int allow_plain) {
    constexpr bool LSTM = CELL == 1;
    constexpr int H = TM_H, G = LSTM ? 4 : 3, GH = G * H;
    __shared__ __attribute__((aligned(16))) float h_lds[2][2 * 4 * TM_HLD];
    __shared__ __attribute__((aligned(16))) float4 xch[2][2][64];          // [k half][unit block][lane]: partial sums for the partner
    __shared__ int dead;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 1, ub = wave & 1, hi = lane >> 5;
    int team, member;
    team_claim_role(reinterpret_cast<unsigned*>(xbuf_all), n_teams, team, member);
    if (team < 0) return;
    u64* const xbuf = xbuf_all + TEAM_HDR + TEAM_MAX * TEAM_M;
    const int plain = team_same_xcd(xbuf_all + TEAM_HDR + team * TEAM_M, member, allow_plain);
    const int ul = 32 * ub + (lane & 31);              // unit inside the member's 64
    const int u = TEAM_US * member + ul;
    const int slot = 2 * kh + hi;                      // this lane's cell: (sequence slot, unit u)
    if (tid == 0) dead = 0;

    // ---- weights: rows (2 hi + m) H + u of W_hh, k in [128 kh, 128 kh + 128) ----------------------------------------
    float w0[TM_KH], w1[TM_KH];
    {
        const bool has1 = 2 * hi + 1 < G;              // the GRU has no gate 3: zero weights in that slot
        const float* r0 = p.Whh + (size_t)((2 * hi + 0) * H + u) * H + TM_KH * kh;
        const float* r1 = p.Whh + (size_t)((has1 ? 2 * hi + 1 : 0) * H + u) * H + TM_KH * kh;
#pragma unroll
        for (int kk = 0; kk < TM_KH; ++kk) { w0[kk] = r0[tm_korder(kk)]; w1[kk] = has1 ? r1[tm_korder(kk)] : 0.f; }
    }
    float bh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bh[g] = g < G ? p.bhh[g * H + u] : 0.f;

    unsigned tag = 0;                                  // the team's running step counter (continues across sequence groups)
    bool failed = false;
    const int n_groups = (p.n_seq + 3) >> 2;
    for (int grp = team; grp < n_groups && !failed; grp += n_teams) {
        int bmap[4], tmax;
        if (!map_slots(p, 4 * grp, bmap, tmax)) continue;      // (all four slots empty: nothing to publish either, for any member)
        const int b = slot == 0 ? bmap[0] : (slot == 1 ? bmap[1] : (slot == 2 ? bmap[2] : bmap[3]));
        // is this slot a duplicate of an earlier one (ragged last group / empty sequence)?  Duplicates compute and store
        // bit-identical values to the same addresses; they exchange through their own ring stream like any other slot.
        const int len = p.seq_len[b];
        const unsigned row0 = (unsigned)p.seq_off[b];
        unsigned goff = row0 * GH + u, soff = row0 * H + u;
        unsigned st_g = goff, st_s = soff, st_p = soff;
        const float h0v = p.h0 ? p.h0[(size_t)b * H + u] : 0.f;
        float c = LSTM ? (p.c0 ? p.c0[(size_t)b * H + u] : 0.f) : h0v;     // the cell's carried state: c (LSTM) / h (GRU)
        float sv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // results of the last finished step: gates 0..3 (GRU: [3] = W_hn h + b_hn), c, h
        float svp0 = c, svp1 = h0v;                    // row 0 of cprev / hprev keeps c0 / h0
        // h0 of all 256 units of the four slots -> LDS buffer 0 (the previous group's last reads ended at its final barrier)
        for (int e = tid; e < 4 * H; e += TM_THREADS) {
            const int q = e >> 8, j = e & (H - 1);
            const int bq = q == 0 ? bmap[0] : (q == 1 ? bmap[1] : (q == 2 ? bmap[2] : bmap[3]));
            h_lds[0][tm_hpos(q, j)] = p.h0 ? p.h0[(size_t)bq * H + j] : 0.f;
        }
        float xc[4] = {0.f, 0.f, 0.f, 0.f}, xn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < G; ++g) xc[g] = p.gates[goff + g * H];
        u64* const xb = xbuf + (size_t)(team * 4 + slot) * (TEAM_SLOTS * H);      // ring of this lane's sequence slot
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
            auto hook = [&](auto K) {
                constexpr int k = decltype(K)::value;          // 0 .. 63
                if constexpr (k >= 1 && k <= G) xnext[k - 1] = tm_ld(lp + (k - 1) * H);
                else if constexpr (k >= 8 && k < 8 + G) tm_st(gs + (k - 8) * H, sv[k - 8]);
                else if constexpr (k == 12) tm_st(cs, sv[LSTM ? 4 : 3]);
                else if constexpr (k == 13) tm_st(hs, sv[5]);
                else if constexpr (k == 14 && LSTM) tm_st(cp, svp0);
                else if constexpr (k == 15) tm_st(hp, svp1);
            };
            f32x4 pa[4];
            FwdProduct<TM_KH>::run(pa, w0, w1, lds_addr(&h_lds[cur][(kh * 4 + (lane & 3)) * TM_HLD + (lane >> 2) * 8]), hook);
            f32x4 acc0 = pa[0] + pa[2], acc1 = pa[1] + pa[3];   // gate columns 2 hi, 2 hi + 1 of the four sequences, this k half
            // ---- k halves: hand the partner the partial sums of ITS two sequences, add its partials of mine -----------
            const int other = 2 * (kh ^ 1);
            xch[kh][ub][lane] = kh ? make_float4(acc0[0], acc0[1], acc1[0], acc1[1]) : make_float4(acc0[2], acc0[3], acc1[2], acc1[3]);
            (void)other;
            __syncthreads();
            const float4 pr = xch[kh ^ 1][ub][lane];
            float y0 = (kh ? acc0[2] : acc0[0]) + pr.x, x0 = (kh ? acc0[3] : acc0[1]) + pr.y;      // gate 2 hi     of sequences 2 kh, 2 kh + 1
            float y1 = (kh ? acc1[2] : acc1[0]) + pr.z, x1 = (kh ? acc1[3] : acc1[1]) + pr.w;      // gate 2 hi + 1
            // ---- gate pairs: afterwards y0 = i, x0 = g, y1 = f, x1 = o of this lane's own cell (sequence 2 kh + hi) --------
            half_swap(y0, x0);
            half_swap(y1, x1);
            // LSTM: i, f, g, o -> c' = f c + i g, h' = o tanh(c');   GRU: r, z, n (og = W_hn h + b_hn) -> h' = (1 - z) n + z h
            const float ig = fast_sigmoid(xcur[0] + (y0 + bh[0]));
            const float fg = fast_sigmoid(xcur[1] + (y1 + bh[1]));
            const float og = LSTM ? fast_sigmoid(xcur[3] + (x1 + bh[3])) : x0 + bh[2];
            const float gg = LSTM ? fast_tanh(xcur[2] + (x0 + bh[2])) : fast_tanh(xcur[2] + ig * og);
            const float cn = LSTM ? fg * c + ig * gg : (1.f - fg) * gg + fg * c;
            const float hn = LSTM ? og * fast_tanh(cn) : cn;
            const float hpub = on ? hn : 0.f;
            ++tag;
            granule_store(xb + (tag & 3) * H + u, hpub, tag, plain);      // publish first: the peers are waiting for it
            h_lds[cur ^ 1][tm_hpos(slot, u)] = hpub;
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
            // ---- the same unit index of the three other members, same sequence -> LDS ------------------------------------
            if (t + 1 < tmax) {
                u64 gr[3];
                const u64* ga[3];
#pragma unroll
                for (int j = 1; j < TEAM_M; ++j) {
                    ga[j - 1] = xb + (tag & 3) * H + TEAM_US * ((member + j) & 3) + ul;
                    gr[j - 1] = granule_load(ga[j - 1]);
                }
                if (!granule_wait_all<3>(gr, ga, tag)) { dead = 1; team_report_timeout(p.fault, TEAM_K_MFMA_FWD, p.layer, team, member, t, b, tag); }
#pragma unroll
                for (int j = 1; j < TEAM_M; ++j)
                    h_lds[cur ^ 1][tm_hpos(slot, TEAM_US * ((member + j) & 3) + ul)] = __uint_as_float((unsigned)gr[j - 1]);
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
        if ((tmax & 1) && !failed) {                   // an odd number of steps ended in buffer 1: the next group starts from buffer 0
            // (nothing to copy: the next group overwrites buffer 0 with its own h0 before its first step)
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------
// Forward without the k split (round 4; the default - DC_DIMS_TEAM_NS(2) keeps the kernel above for A/B).  Above, a wave owns a k half of
// 32 units' gate-column pairs, so every step the halves meet in LDS behind a barrier before anybody knows a gate.  Here a lane owns ONE gate
// column over the whole K = 256 (still 256 AGPRs, still 256 MFMAs per wave and step): wave w, 16-lane row g, lane j = gate g of unit
// 64 m + 16 w + j.  After the product a lane holds that gate's sums for the four sequences; a 4 x 4 transpose between the wave's four
// rows and four registers (two v_permlane32_swap + two v_permlane16_swap) leaves row s with the four gates of (sequence s, unit j):
// 64 cells per wave, one per lane.  One barrier per step instead of two, no LDS round trip between product and gate math.
// ---------------------------------------------------------------------------------------------------
enum { TN_HLD = 260 };     // floats per sequence row of the LDS image (k in plain order): lane 4b + i reads [i][16 b .. 16 b + 15] as four
                           // ds_read_b128; 260 / 4 = 65 = 1 mod 16: the sixteen lanes of a read group start at 16-byte units i + 4 b - all different
__device__ __forceinline__ constexpr int tn_korder(int kk) { return 16 * (kk & 15) + (kk >> 4); }

// F16P (round 6, DC_DIMS_F16X2): W_hh and h as two f16 planes each (x 2^s = h + m, the x W^T products' arithmetic: weights x 2^8, states x 2^4)
// on v_mfma_f32_4x4x4_16b_f16 - three instructions per four k instead of four (persist_util.h, FwdProductColH).  The LDS image is then
// two planes of halfs (rows TN_HLD halfs apart: 520 bytes = 2 banks more than a multiple of 64, so the 32 lanes of a ds_read_b64 half
// touch 64 different banks) and a granule carries the two halfs of a state instead of its float.  A weight beyond 65504 / 2^8 becomes
// inf - inf = NaN in every step's output, the f16x2 contract of the other products (include/dotaclient_hip.h).
__device__ __forceinline__ unsigned f16_planes(float v, float scale) {
    const float x = v * scale;
    const _Float16 h = (_Float16)x;
    const _Float16 m = (_Float16)(x - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, m) << 16);
}
template <int CELL, bool F16P, bool KEEP>     // CELL 1: LSTM, 0: GRU
__global__ __launch_bounds__(TM_THREADS, 1) void team_mfma_fwd_col_kernel(RnnStepArgs p, u64* __restrict__ xbuf_all, int n_teams, int allow_plain) {
    // fwd_only (DC_DIMS_FWD_ONLY: the optimizer's no-grad rollout pass, optimizer.py:344-385): what only a backward would read - the activated
    // gates, h / c of the step before - is not written (562 -> ~135 MB per launch: h and c go out, the gate rows stay as the input projection left them)
    // KEEP is a template parameter, not a run-time flag: with the stores behind uniform branches the compiler's vmcnt bookkeeping assumes the
    // path with the fewest stores, and its wait for the NEXT step's input-projection loads then also waits for this step's first stores to be
    // acknowledged by memory (measured: it ate two thirds of what the f16 planes saved)
    constexpr bool keep = KEEP;
    constexpr bool LSTM = CELL == 1;
    constexpr int H = TM_H, G = LSTM ? 4 : 3, GH = G * H;
    __shared__ __attribute__((aligned(16))) float h_lds[2][4 * TN_HLD];      // F16P: [buffer][plane][4 * TN_HLD] halfs (same bytes)
    unsigned short* const h_img = reinterpret_cast<unsigned short*>(&h_lds[0][0]);
    constexpr int IMG_PLANE = 4 * TN_HLD, IMG_BUF = 2 * IMG_PLANE;           // halfs
    __shared__ int dead;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane >> 4, j = lane & 15;          // product role: gate `row`; cell role: sequence slot `row`
    int team, member;
    team_claim_role(reinterpret_cast<unsigned*>(xbuf_all), n_teams, team, member);
    if (team < 0) return;
    u64* const xbuf = xbuf_all + TEAM_HDR + TEAM_MAX * TEAM_M;
    const int plain = team_same_xcd(xbuf_all + TEAM_HDR + team * TEAM_M, member, allow_plain);
    const int ul = 16 * wave + j;                      // unit inside the member's 64
    const int u = TEAM_US * member + ul;
    const int slot = row;
    if (tid == 0) dead = 0;

    // ---- weights: row (gate H + u) of W_hh, all k, in the order the product contracts them ---------------------------
    float w[F16P ? 1 : H];
    f16x4 wh[F16P ? H / 4 : 1], wm[F16P ? H / 4 : 1];
    {
        const bool has = row < G;                      // the GRU has no gate 3: zero weights in that row
        const float* r0 = p.Whh + (size_t)((has ? row : 0) * H + u) * H;
        if constexpr (F16P) {
#pragma unroll
            for (int g = 0; g < H / 4; ++g) {          // k-group g: k = 16 (g & 15) + 4 (g >> 4) + e
                const float4 v = *reinterpret_cast<const float4*>(r0 + 16 * (g & 15) + 4 * (g >> 4));
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = has ? e[q] * 256.f : 0.f;
                    const _Float16 hi = (_Float16)x;
                    wh[g][q] = hi;
                    wm[g][q] = (_Float16)(x - (float)hi);
                }
                // pin each plane entry to ONE 64-bit AGPR pair here: left to itself the compiler keeps the four dwords of (wh, wm) apart and
                // gathers them with four v_accvgpr_mov in front of every k-group of every step
                asm volatile("" : "+a"(wh[g]), "+a"(wm[g]));
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < H; ++kk) w[kk] = has ? r0[tn_korder(kk)] : 0.f;
        }
    }
    float bh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bh[g] = g < G ? p.bhh[g * H + u] : 0.f;

    unsigned tag = 0;
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
        float c = LSTM ? (p.c0 ? p.c0[(size_t)b * H + u] : 0.f) : h0v;
        float sv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float svp0 = c, svp1 = h0v;
        for (int e = tid; e < 4 * H; e += TM_THREADS) {
            const int q = e >> 8, jj = e & (H - 1);
            const int bq = q == 0 ? bmap[0] : (q == 1 ? bmap[1] : (q == 2 ? bmap[2] : bmap[3]));
            const float hv = p.h0 ? p.h0[(size_t)bq * H + jj] : 0.f;
            if constexpr (F16P) {
                const unsigned pk = f16_planes(hv, 16.f);
                h_img[q * TN_HLD + jj] = (unsigned short)pk;
                h_img[IMG_PLANE + q * TN_HLD + jj] = (unsigned short)(pk >> 16);
            } else h_lds[0][q * TN_HLD + jj] = hv;
        }
        float xc[4] = {0.f, 0.f, 0.f, 0.f}, xn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < G; ++g) xc[g] = p.gates[goff + g * H];
        u64* const xb = xbuf + (size_t)(team * 4 + slot) * (TEAM_SLOTS * H);      // ring of this lane's sequence slot
        __syncthreads();

#ifdef TM_TIMING
        long long tacc[5] = {0, 0, 0, 0, 0};
#endif
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
                constexpr int k = decltype(K)::value;          // 0 .. 63
                if constexpr (k >= 1 && k <= G) xnext[k - 1] = tm_ld(lp + (k - 1) * H);
                else if constexpr (k >= 8 && k < 8 + G) { if (keep) tm_st(gs + (k - 8) * H, sv[k - 8]); }
                else if constexpr (k == 12) { if (keep || LSTM) tm_st(cs, sv[LSTM ? 4 : 3]); }       // (the GRU's W_hn h + b_hn: the backward's only)
                else if constexpr (k == 13) tm_st(hs, sv[5]);
                else if constexpr (k == 14 && LSTM) { if (keep) tm_st(cp, svp0); }
                else if constexpr (k == 15) { if (keep) tm_st(hp, svp1); }
            };
#ifdef TM_TIMING
            const long long tq0 = __builtin_amdgcn_s_memtime();
#endif
            f32x4 tot;                                               // gate `row` of unit u for the four sequences
            if constexpr (F16P) {
                f32x4 pa[3];
                FwdProductColH<H, 2 * IMG_PLANE>::run(pa, wh, wm, lds_addr(h_img + cur * IMG_BUF + (lane & 3) * TN_HLD + (lane >> 2) * 16), hook);
                tot = (pa[0] + (pa[1] + pa[2])) * (1.f / 4096.f);
            } else {
                f32x4 pa[4];
                FwdProductCol<H>::run(pa, w, lds_addr(&h_lds[cur][(lane & 3) * TN_HLD + (lane >> 2) * 16]), hook);
                tot = (pa[0] + pa[1]) + (pa[2] + pa[3]);
            }
#ifdef TM_TIMING
            const long long tq1 = __builtin_amdgcn_s_memtime();
#endif
            float a[4] = {tot[0], tot[1], tot[2], tot[3]};
            rows_transpose4(a);                                      // a[g] = gate g of (sequence `row`, unit u)
            const float y0 = a[0], y1 = a[1], x0 = a[2], x1 = a[3];
            const float ig = fast_sigmoid(xcur[0] + (y0 + bh[0]));
            const float fg = fast_sigmoid(xcur[1] + (y1 + bh[1]));
            const float og = LSTM ? fast_sigmoid(xcur[3] + (x1 + bh[3])) : x0 + bh[2];
            const float gg = LSTM ? fast_tanh(xcur[2] + (x0 + bh[2])) : fast_tanh(xcur[2] + ig * og);
            const float cn = LSTM ? fg * c + ig * gg : (1.f - fg) * gg + fg * c;
            const float hn = LSTM ? og * fast_tanh(cn) : cn;
            const float hpub = on ? hn : 0.f;
            ++tag;
            if constexpr (F16P) {
                const unsigned pk = f16_planes(hpub, 16.f);
                granule_store(xb + (tag & 3) * H + u, __uint_as_float(pk), tag, plain);
                h_img[(cur ^ 1) * IMG_BUF + slot * TN_HLD + u] = (unsigned short)pk;
                h_img[(cur ^ 1) * IMG_BUF + IMG_PLANE + slot * TN_HLD + u] = (unsigned short)(pk >> 16);
            } else {
                granule_store(xb + (tag & 3) * H + u, hpub, tag, plain);
                h_lds[cur ^ 1][slot * TN_HLD + u] = hpub;
            }
#ifdef TM_TIMING
            const long long tq2 = __builtin_amdgcn_s_memtime();
#endif
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
                u64 gr[3];
                const u64* ga[3];
#pragma unroll
                for (int q = 1; q < TEAM_M; ++q) {
                    ga[q - 1] = xb + (tag & 3) * H + TEAM_US * ((member + q) & 3) + ul;
                    gr[q - 1] = granule_load(ga[q - 1]);
                }
                if (!granule_wait_all<3>(gr, ga, tag)) { dead = 1; team_report_timeout(p.fault, TEAM_K_MFMA_FWD, p.layer, team, member, t, b, tag); }
#pragma unroll
                for (int q = 1; q < TEAM_M; ++q) {
                    const int at = slot * TN_HLD + TEAM_US * ((member + q) & 3) + ul;
                    if constexpr (F16P) {
                        h_img[(cur ^ 1) * IMG_BUF + at] = (unsigned short)gr[q - 1];
                        h_img[(cur ^ 1) * IMG_BUF + IMG_PLANE + at] = (unsigned short)((unsigned)gr[q - 1] >> 16);
                    } else h_lds[cur ^ 1][at] = __uint_as_float((unsigned)gr[q - 1]);
                }
            }
#ifdef TM_TIMING
            const long long tq3 = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();
#ifdef TM_TIMING
            const long long tq4 = __builtin_amdgcn_s_memtime();
            tacc[0] += tq1 - tq0; tacc[1] += tq2 - tq1; tacc[2] += tq3 - tq2; tacc[3] += tq4 - tq3; tacc[4] += 1;
#endif
            return dead == 0;
        };
#ifdef TM_TIMING
        const long long tg0 = __builtin_amdgcn_s_memtime();
#endif
        for (int t = 0; t < tmax; t += 2) {
            if (!step(t, xc, xn, std::integral_constant<int, 0>{})) { failed = true; break; }
            if (t + 1 < tmax && !step(t + 1, xn, xc, std::integral_constant<int, 1>{})) { failed = true; break; }
        }
#ifdef TM_TIMING
        if (tid == 0 && (team == 0 || team == 37) && member < 2)
            printf("team_fwd_col F16P=%d team %d member %d plain %d: steps %lld  product %.0f  cell+publish %.0f  gather %.0f  barrier %.0f  loop total/step %.0f (s_memtime ticks per step)\n",
                   (int)F16P, team, member, plain, tacc[4], (double)tacc[0] / tacc[4], (double)tacc[1] / tacc[4], (double)tacc[2] / tacc[4], (double)tacc[3] / tacc[4],
                   (double)(__builtin_amdgcn_s_memtime() - tg0) / tacc[4]);
#endif
        if (keep) {
#pragma unroll
            for (int g = 0; g < G; ++g) p.gates[st_g + g * H] = sv[g];
        }
        if constexpr (LSTM) { p.cseq[st_s] = sv[4]; if (keep) p.cprev[st_p] = svp0; }
        else if (keep) p.hn[st_s] = sv[3];
        p.hseq[st_s] = failed ? __builtin_nanf("") : sv[5];
        if (keep) p.hprev[st_p] = svp1;
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------
// Forward, f16 planes, OWN UNITS FIRST (round 6; the default with DC_DIMS_F16X2).  In the kernel above a step is a chain
//   product (all 256 k) -> cell -> publish h -> wait for the three other members' h -> barrier -> next product,
// and the wait is two L2 trips long (TM_TIMING: product 1 945 ticks, cell + publish 395, gather 980, barrier 155).  But a member's own 64
// units never leave the CU: with the LDS image in member-LOCAL k order (k_local = (k - 64 member) mod 256, weights loaded to match) the 16
// k-groups of broadcast blocks 0 .. 3 contract own states only.  So a step becomes
//   publish h_t | own h_t -> LDS, barrier | phase A: 48 MFMAs on the own quarter of k (the poll loads for the others' granules are issued
//   from one of its hooks: late enough to find them in L2, early enough to be back by its end) | gather, foreign h_t -> LDS, barrier |
//   phase B: 144 MFMAs on the other three quarters, next step's input loads and this step's stores from its hooks | cell -> publish ...
// and the hand-off hides behind phase A and the first barrier.  One more barrier and one more set of LDS reads per step.
// PG: the phase A hook the poll loads are issued from.
// ---------------------------------------------------------------------------------------------------
#ifndef TM_OF_PG
#define TM_OF_PG 4
#endif
template <int CELL, bool KEEP>     // CELL 1: LSTM, 0: GRU; KEEP: also store what only a backward reads (see above)
__global__ __launch_bounds__(TM_THREADS, 1) void team_mfma_fwd_of_kernel(RnnStepArgs p, u64* __restrict__ xbuf_all, int n_teams, int allow_plain) {
    constexpr bool keep = KEEP;
    constexpr bool LSTM = CELL == 1;
    constexpr int H = TM_H, G = LSTM ? 4 : 3, GH = G * H, PG = TM_OF_PG;
    constexpr int IMG_PLANE = 4 * TN_HLD, IMG_BUF = 2 * IMG_PLANE;           // halfs
    __shared__ __attribute__((aligned(16))) unsigned short h_img[2 * IMG_BUF];      // [buffer][plane][sequence][TN_HLD], member-local k order
    __shared__ int dead;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane >> 4, j = lane & 15;          // product role: gate `row`; cell role: sequence slot `row`
    int team, member;
    team_claim_role(reinterpret_cast<unsigned*>(xbuf_all), n_teams, team, member);
    if (team < 0) return;
    u64* const xbuf = xbuf_all + TEAM_HDR + TEAM_MAX * TEAM_M;
    const int plain = team_same_xcd(xbuf_all + TEAM_HDR + team * TEAM_M, member, allow_plain);
    const int ul = 16 * wave + j;                      // unit inside the member's 64 = its local k
    const int u = TEAM_US * member + ul;
    const int slot = row;
    if (tid == 0) dead = 0;

    // ---- weights: row (gate H + u) of W_hh as two f16 planes (x 2^8); k-group g holds local k = 16 (g & 15) + 4 (g >> 4) + e ----------
    f16x4 wh[H / 4], wm[H / 4];
    {
        const bool has = row < G;                      // the GRU has no gate 3: zero weights in that row
        const float* r0 = p.Whh + (size_t)((has ? row : 0) * H + u) * H;
#pragma unroll
        for (int g = 0; g < H / 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(r0 + ((16 * (g & 15) + 4 * (g >> 4) + TEAM_US * member) & (H - 1)));
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x = has ? e[q] * 256.f : 0.f;
                const _Float16 hi = (_Float16)x;
                wh[g][q] = hi;
                wm[g][q] = (_Float16)(x - (float)hi);
            }
            asm volatile("" : "+a"(wh[g]), "+a"(wm[g]));      // one 64-bit AGPR pair each (see the kernel above)
        }
    }
    float bh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bh[g] = g < G ? p.bhh[g * H + u] : 0.f;

    unsigned tag = 0;
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
        float c = LSTM ? (p.c0 ? p.c0[(size_t)b * H + u] : 0.f) : h0v;
        float sv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float svp0 = c, svp1 = h0v;
        for (int e = tid; e < 4 * H; e += TM_THREADS) {
            const int q = e >> 8, jj = e & (H - 1);
            const int bq = q == 0 ? bmap[0] : (q == 1 ? bmap[1] : (q == 2 ? bmap[2] : bmap[3]));
            const unsigned pk = f16_planes(p.h0 ? p.h0[(size_t)bq * H + jj] : 0.f, 16.f);
            const int at = q * TN_HLD + ((jj - TEAM_US * member) & (H - 1));
            h_img[at] = (unsigned short)pk;
            h_img[IMG_PLANE + at] = (unsigned short)(pk >> 16);
        }
        float xc[4] = {0.f, 0.f, 0.f, 0.f}, xn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < G; ++g) xc[g] = p.gates[goff + g * H];
        u64* const xb = xbuf + (size_t)(team * 4 + slot) * (TEAM_SLOTS * H);      // ring of this lane's sequence slot
        __syncthreads();
#ifdef TM_TIMING
        long long tacc[6] = {0, 0, 0, 0, 0, 0}, tpub = 0, tpub0 = 0;
        int tmiss = 0;
#endif
        // step t: gates of step t from h_{t-1} in image `cur` (own quarter complete on entry, the other three arrive during phase A)
        auto step = [&](const int t, float (&xcur)[4], float (&xnext)[4], auto CUR) __attribute__((always_inline)) {
            constexpr int cur = decltype(CUR)::value;
            const bool on = t < len, on1 = t + 1 < len;
            const unsigned gnx = goff + (on1 ? GH : 0);
            const float* const lp = p.gates + gnx;
            float* const gs = p.gates + st_g;
            float* const cs = (LSTM ? p.cseq : p.hn) + st_s;
            float* const hs = p.hseq + st_s;
            float* const cp = LSTM ? p.cprev + st_p : nullptr;
            float* const hp = p.hprev + st_p;
            const bool gather = t > 0;                 // (step 0 reads the initial state, complete in the image)
            u64 gr[3] = {0, 0, 0};
            const u64* ga[3];
#pragma unroll
            for (int q = 1; q < TEAM_M; ++q) ga[q - 1] = xb + (tag & 3) * H + TEAM_US * ((member + q) & 3) + ul;
            auto hook = [&](auto K) __attribute__((always_inline)) {
                constexpr int k = decltype(K)::value;          // 0 .. 15: phase A, 16 .. 63: phase B
                if constexpr (k == PG) {
                    if (gather) {
#pragma unroll
                        for (int q = 0; q < 3; ++q) gr[q] = granule_load(ga[q]);
                    }
                }
                else if constexpr (k >= 17 && k <= 16 + G) xnext[k - 17] = tm_ld(lp + (k - 17) * H);
                else if constexpr (k >= 24 && k < 24 + G) { if constexpr (keep) tm_st(gs + (k - 24) * H, sv[k - 24]); }
                else if constexpr (k == 28) { if constexpr (keep || LSTM) tm_st(cs, sv[LSTM ? 4 : 3]); }       // (the GRU's W_hn h + b_hn: the backward's only)
                else if constexpr (k == 29) tm_st(hs, sv[5]);
                else if constexpr (k == 30 && LSTM) { if constexpr (keep) tm_st(cp, svp0); }
                else if constexpr (k == 31) { if constexpr (keep) tm_st(hp, svp1); }
            };
            using Prod = FwdProductOwnFirst<H, 2 * IMG_PLANE>;
            const uint32_t addr = lds_addr(h_img + cur * IMG_BUF + (lane & 3) * TN_HLD + (lane >> 2) * 16);
#ifdef TM_TIMING
            const long long tq0 = __builtin_amdgcn_s_memtime();
#endif
            f32x4 pa[3];
            Prod::phase_a(pa, wh, wm, addr, hook);
#ifdef TM_TIMING
            const long long tq1 = __builtin_amdgcn_s_memtime();
#endif
            if (gather) {
#ifdef TM_TIMING
                for (int q = 0; q < 3; ++q) if ((unsigned)(gr[q] >> 32) != tag) { ++tmiss; break; }
                tpub += tq1 - tpub0;
#endif
                if (!granule_wait_all<3>(gr, ga, tag)) { dead = 1; team_report_timeout(p.fault, TEAM_K_MFMA_FWD, p.layer, team, member, t, b, tag); }
#pragma unroll
                for (int q = 1; q < TEAM_M; ++q) {
                    const int at = cur * IMG_BUF + slot * TN_HLD + TEAM_US * q + ul;
                    h_img[at] = (unsigned short)gr[q - 1];
                    h_img[IMG_PLANE + at] = (unsigned short)((unsigned)gr[q - 1] >> 16);
                }
            }
#ifdef TM_TIMING
            const long long tq2 = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();
#ifdef TM_TIMING
            const long long tq3 = __builtin_amdgcn_s_memtime();
#endif
            Prod::phase_b(pa, wh, wm, addr, hook);
#ifdef TM_TIMING
            const long long tq4 = __builtin_amdgcn_s_memtime();
#endif
            const f32x4 tot = (pa[0] + (pa[1] + pa[2])) * (1.f / 4096.f);      // gate `row` of unit u for the four sequences
            float a[4] = {tot[0], tot[1], tot[2], tot[3]};
            rows_transpose4(a);                                      // a[g] = gate g of (sequence `row`, unit u)
            const float y0 = a[0], y1 = a[1], x0 = a[2], x1 = a[3];
            const float ig = fast_sigmoid(xcur[0] + (y0 + bh[0]));
            const float fg = fast_sigmoid(xcur[1] + (y1 + bh[1]));
            const float og = LSTM ? fast_sigmoid(xcur[3] + (x1 + bh[3])) : x0 + bh[2];
            const float gg = LSTM ? fast_tanh(xcur[2] + (x0 + bh[2])) : fast_tanh(xcur[2] + ig * og);
            const float cn = LSTM ? fg * c + ig * gg : (1.f - fg) * gg + fg * c;
            const float hn = LSTM ? og * fast_tanh(cn) : cn;
            const float hpub = on ? hn : 0.f;
            ++tag;
            const unsigned pk = f16_planes(hpub, 16.f);
            if (plain) granule_store(xb + (tag & 3) * H + u, __uint_as_float(pk), tag, 1);
            else granule_store(xb + (tag & 3) * H + u, __uint_as_float(pk), tag, 0);
            h_img[(cur ^ 1) * IMG_BUF + slot * TN_HLD + ul] = (unsigned short)pk;
            h_img[(cur ^ 1) * IMG_BUF + IMG_PLANE + slot * TN_HLD + ul] = (unsigned short)(pk >> 16);
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
#ifdef TM_TIMING
            const long long tq5 = __builtin_amdgcn_s_memtime();
            tpub0 = tq5;
#endif
            __syncthreads();
#ifdef TM_TIMING
            const long long tq6 = __builtin_amdgcn_s_memtime();
            tacc[0] += tq1 - tq0; tacc[1] += tq2 - tq1; tacc[2] += tq3 - tq2; tacc[3] += tq4 - tq3; tacc[4] += tq5 - tq4; tacc[5] += tq6 - tq5;
#endif
            return dead == 0;
        };
#ifdef TM_TIMING
        const long long tg0 = __builtin_amdgcn_s_memtime();
#endif
        for (int t = 0; t < tmax; t += 2) {
            if (!step(t, xc, xn, std::integral_constant<int, 0>{})) { failed = true; break; }
            if (t + 1 < tmax && !step(t + 1, xn, xc, std::integral_constant<int, 1>{})) { failed = true; break; }
        }
#ifdef TM_TIMING
        if (tid == 0 && (team == 0 || team == 37) && member < 2)
            printf("team_fwd_of PG=%d team %d member %d plain %d: steps %d  first poll stale in %d steps (lane 0), publish -> end of phase A %.0f;  phase A %.0f  gather %.0f  barrier %.0f  phase B %.0f  cell+publish %.0f  barrier %.0f  loop total/step %.0f (s_memtime ticks per step)\n",
                   PG, team, member, plain, tmax, tmiss, (double)tpub / tmax, (double)tacc[0] / tmax, (double)tacc[1] / tmax, (double)tacc[2] / tmax, (double)tacc[3] / tmax, (double)tacc[4] / tmax,
                   (double)tacc[5] / tmax, (double)(__builtin_amdgcn_s_memtime() - tg0) / tmax);
#endif
        if constexpr (keep) {
#pragma unroll
            for (int g = 0; g < G; ++g) p.gates[st_g + g * H] = sv[g];
        }
        if constexpr (LSTM) { p.cseq[st_s] = sv[4]; if constexpr (keep) p.cprev[st_p] = svp0; }
        else if constexpr (keep) p.hn[st_s] = sv[3];
        p.hseq[st_s] = failed ? __builtin_nanf("") : sv[5];
        if constexpr (keep) p.hprev[st_p] = svp1;
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------
// backward through time.  The forward is COLUMN-parallel (a member owns the gate columns of its 64 units and needs all of
// h_{t-1}: an all-gather of 192 foreign values per sequence).  Done the same way, the backward would need all 1 024 gate
// gradients of every sequence in every member - 3 072 eight-byte reads per member and step, and that exchange, not the
// arithmetic, is what a step then costs (measured: 3.8 us per step, the VALU team kernel's time).  So the backward is
// ROW-parallel: a member contracts ITS OWN 256 gate gradients (4 gates x its 64 units - no input from anyone) with the rows
// W_hh[own column][u'] for ALL 256 output units u' - the very weights of its forward slice - and the partial sums
// dh_rec[seq][u'] are reduce-scattered: a lane whose u' belongs to another member publishes its four sums to the owner
// (768 granules out, 768 in per member and step: a quarter of the reads), the owner adds the three it receives to its own.
//   lane (wave w, l): output unit u' = 64 w + l, K = 256 own columns kk = 64 g + j (gate g, own unit j): 256 AGPRs,
//   rnn_persist.hip's BwdProduct<256> with both wave halves on the same k range - no cross-lane reduction at all.
//   cell (unit 64 m + (tid & 63), sequence slot tid >> 6): own partial through LDS + three granules -> gate gradients ->
//   the LDS image of the next product (all local) and dgx.
// In: dh (from the layer above), the forward's gates / cseq / cprev; out: dgx.
// ---------------------------------------------------------------------------------------------------
// LDS image of the A operand: per sequence eight blocks of 32 floats (one per broadcast group b'), 36 floats apart, rows 336
// floats apart: the sixteen lanes of every ds_read_b128 lane group (four sequences x four blocks) then start 16 bytes x an odd
// pattern apart - no bank conflicts (32-float blocks in rows of 264 gave two-way conflicts on every read)
enum { TB_KH = 256, TB_BLK = 36, TB_GLD = 336 };
__device__ __forceinline__ int tb_pos(int kk) { return TB_BLK * (kk >> 5) + (kk & 31); }
// BwdProduct<256> hands MFMA number kk' (abid = kk' & 7 of A register kk' >> 3) the 32 consecutive floats lane 4b'+i read at
// [i][32 b' ..]: with the image in PLAIN order it contracts kk = 32 (kk' & 7) + (kk' >> 3); the weights are loaded in that order
__device__ __forceinline__ constexpr int tb_korder(int kk) { return 32 * (kk & 7) + (kk >> 3); }

// F16P (round 6, DC_DIMS_F16X2 with RnnStepArgs::s_grad > 0): the product on f16 planes like the forward's - W_hh x 2^8, the gate gradients
// x s_grad (the power of two the other backward products of the mode use, policy.hip: f16x2_grad_scale) - with FwdProductColH: a lane's
// output unit is its "column", the image holds the gate gradients as two planes of halfs in plain kk order.
template <int CELL, bool F16P>     // 1: LSTM, 0: GRU (contracts dgh = d(W_hh h + b_hh) of step t + 1; writes dgx and dgh)
__global__ __launch_bounds__(TM_THREADS, 1) void team_mfma_bwd_kernel(RnnStepArgs p, u64* __restrict__ xbuf_all, int n_teams, int allow_plain) {
    constexpr bool LSTM = CELL == 1;
    constexpr int H = TM_H, G = LSTM ? 4 : 3, GH = G * H, KH = TB_KH;
    __shared__ __attribute__((aligned(16))) float g_lds[2][4 * TB_GLD];      // own gate gradients [seq][pos(kk)], kk = 64 gate + own unit
    static_assert(sizeof(float) * 2 * 4 * TB_GLD >= sizeof(unsigned short) * 2 * 2 * 4 * TN_HLD, "the planes fit the f32 image");
    unsigned short* const g_img = reinterpret_cast<unsigned short*>(&g_lds[0][0]);      // F16P: [buffer][plane][4 * TN_HLD] halfs
    constexpr int IMG_PLANE = 4 * TN_HLD, IMG_BUF = 2 * IMG_PLANE;
    __shared__ float own[4][TEAM_US];                                         // own partial sums dh_rec[seq][own unit]
    __shared__ int dead;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int team, member;
    team_claim_role(reinterpret_cast<unsigned*>(xbuf_all), n_teams, team, member);
    if (team < 0) return;
    u64* const xbuf = xbuf_all + TEAM_HDR + TEAM_MAX * TEAM_M;
    const int plain = team_same_xcd(xbuf_all + TEAM_HDR + team * TEAM_M, member, allow_plain);
    const int up = tid;                                // product role: output unit u' = tid, owner member = wave
    const int slot = wave, ul = lane;                  // cell role: sequence slot, own unit
    const int u = TEAM_US * member + ul;
    if (tid == 0) dead = 0;

    // ---- weights: W_hh[(kk >> 6) H + 64 m + (kk & 63)][u'], kk = 0 .. 255 ---------------------------------------------
    float w[F16P ? 1 : KH];
    f16x4 wh[F16P ? KH / 4 : 1], wm[F16P ? KH / 4 : 1];
    if constexpr (F16P) {
#pragma unroll
        for (int g = 0; g < KH / 4; ++g) {             // k-group g: kk = 16 (g & 15) + 4 (g >> 4) + e
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 16 * (g & 15) + 4 * (g >> 4) + e;
                const float x = (k >> 6) < G ? p.Whh[(size_t)((k >> 6) * H + TEAM_US * member + (k & 63)) * H + up] * 256.f : 0.f;
                const _Float16 hi = (_Float16)x;
                wh[g][e] = hi;
                wm[g][e] = (_Float16)(x - (float)hi);
            }
            asm volatile("" : "+a"(wh[g]), "+a"(wm[g]));      // one 64-bit AGPR pair each (see the forward)
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) {
            const int k = tb_korder(kk);                   // own gate column 64 gate + own unit (GRU: slot 3 is empty)
            w[kk] = (k >> 6) < G ? p.Whh[(size_t)((k >> 6) * H + TEAM_US * member + (k & 63)) * H + up] : 0.f;
        }
    }
    const float s_grad = p.s_grad, inv_scale = F16P ? 1.f / (256.f * p.s_grad) : 1.f;

    // ring of a team: [tag & 3][owner member][source member][sequence slot][64 units] granules
    u64* const ring0 = xbuf + (size_t)team * (TEAM_SLOTS * 4 * 4 * 4 * TEAM_US);
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
        for (int e = tid; e < 4 * TB_GLD; e += TM_THREADS) g_lds[0][e] = 0.f;      // "step tmax" has no gradient (F16P: buffer 0 lies inside)
        __syncthreads();

#ifdef TM_TIMING
        long long tacc[6] = {0, 0, 0, 0, 0, 0};
#endif
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
                constexpr int k = decltype(K)::value;          // 0 .. 63
                if constexpr (k >= 1 && k <= G) nv[k - 1] = tm_ld(lg + (k - 1) * H);
                else if constexpr (k == 5) nv[LSTM ? 4 : 3] = tm_ld(lc);
                else if constexpr (k == 6) nv[5] = tm_ld(lcp);
                else if constexpr (k == 7) nv[6] = tm_ld(ldh);
                else if constexpr (k >= 10 && k < 10 + G) tm_st(gs + (k - 10) * H, sv[k - 10]);
                else if constexpr (!LSTM && (k == 14 || k == 15)) tm_st(ghs + (k - 14) * H, sv[k - 14]);
                else if constexpr (!LSTM && k == 16) tm_st(ghs + 2 * H, svh2);
            };
            // ---- partial dh_rec[seq 0..3][u'] over this member's 256 gate columns ------------------------------------------
#ifdef TM_TIMING
            const long long tq0 = __builtin_amdgcn_s_memtime();
#endif
            f32x4 acc;
            if constexpr (F16P) {
                f32x4 pa[3];
                FwdProductColH<KH, 2 * IMG_PLANE>::run(pa, wh, wm, lds_addr(g_img + cur * IMG_BUF + (lane & 3) * TN_HLD + (lane >> 2) * 16), hook);
                acc = (pa[0] + (pa[1] + pa[2])) * inv_scale;
            } else {
                f32x4 pa[4];
                BwdProduct<KH>::run(pa, w, lds_addr(&g_lds[cur][(lane & 3) * TB_GLD + ((lane >> 2) & 7) * TB_BLK]), hook);
                acc = (pa[0] + pa[1]) + (pa[2] + pa[3]);
            }
#ifdef TM_TIMING
            const long long tq1 = __builtin_amdgcn_s_memtime();
#endif
            ++tag;
            u64* const ring = ring0 + (size_t)(tag & 3) * (4 * 4 * 4 * TEAM_US);
            if (xchg) {
                if (wave == member) {                                  // my own units: through LDS
#pragma unroll
                    for (int q = 0; q < 4; ++q) own[q][lane] = acc[q];
                } else {                                               // the owner's: [owner = wave][source = member][seq][unit]
                    u64* dst = ring + ((size_t)(wave * 4 + member) * 4) * TEAM_US + lane;
                    if (plain) {                                       // (one uniform branch, not one per store)
#pragma unroll
                        for (int q = 0; q < 4; ++q) granule_store(dst + q * TEAM_US, acc[q], tag, 1);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) granule_store(dst + q * TEAM_US, acc[q], tag, 0);
                    }
                }
            }
#ifdef TM_TIMING
            const long long tq2 = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();
#ifdef TM_TIMING
            const long long tq3 = __builtin_amdgcn_s_memtime();
#endif
            float rec = 0.f;
            u64 gr[3] = {0, 0, 0};
            const u64* ga[3];
            if (xchg) {
#pragma unroll
                for (int j = 1; j < TEAM_M; ++j) {
                    ga[j - 1] = ring + ((size_t)(member * 4 + ((member + j) & 3)) * 4 + slot) * TEAM_US + ul;
                    gr[j - 1] = granule_load(ga[j - 1]);
                }
                rec = own[slot][ul];
            }
            // everything of the cell's backward that does not need dh_rec, while the granules are in flight: afterwards the gate gradients
            // are one fma and four multiplies away.  LSTM: dcv = dh A + B, dg = dcv C0..2, dh C3;  GRU: dg = dh D0..2, dgh_n = dh D3
            const float ig = cv[0], fg = cv[1], gg = cv[2], og = cv[3];
            float cA = 0.f, cB = 0.f, cC[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (LSTM) {
                const float tc = fast_tanh(cv[4]);
                cA = og * (1.f - tc * tc);
                cB = has_next ? dc_next * f_next : 0.f;
                cC[0] = on ? gg * ig * (1.f - ig) : 0.f; cC[1] = on ? cv[5] * fg * (1.f - fg) : 0.f;
                cC[2] = on ? ig * (1.f - gg * gg) : 0.f; cC[3] = on ? tc * og * (1.f - og) : 0.f;
            } else {                                       // r = ig, z = fg, n = gg, og = W_hn h + b_hn, cv[5] = h_{t-1}  (rnn.hip)
                cB = has_next ? dc_next * f_next : 0.f;    // direct path h_{t+1} = ... + z_{t+1} h_t
                const float d0 = (1.f - fg) * (1.f - gg * gg);
                cC[2] = d0;                                // dn_pre = dh d0
                cC[1] = (cv[5] - gg) * fg * (1.f - fg);    // dz_pre
                cC[0] = d0 * og * ig * (1.f - ig);         // dr_pre
                cC[3] = d0 * ig;                           // the n column of dgh
            }
            if (xchg) {
                if (!granule_wait_all<3>(gr, ga, tag)) { dead = 1; team_report_timeout(p.fault, TEAM_K_MFMA_BWD, p.layer, team, member, t, b, tag); }
#pragma unroll
                for (int j = 1; j < TEAM_M; ++j) rec += __uint_as_float((unsigned)gr[j - 1]);
            }
#ifdef TM_TIMING
            asm volatile("" : "+v"(rec));
            const long long tq4 = __builtin_amdgcn_s_memtime();
#endif
            float dh = cv[6];
            dh += has_next ? rec : 0.f;
            float dgr[4];                                  // what the next product contracts: d(W_hh h + b_hh) of this step
            if constexpr (LSTM) {
                const float dcv = dh * cA + cB;
                dgr[0] = dcv * cC[0]; dgr[1] = dcv * cC[1]; dgr[2] = dcv * cC[2]; dgr[3] = dh * cC[3];
                sv[0] = on ? dgr[0] : sv[0]; sv[1] = on ? dgr[1] : sv[1]; sv[2] = on ? dgr[2] : sv[2]; sv[3] = on ? dgr[3] : sv[3];
                dc_next = on ? dcv : dc_next;
            } else {
                dh += cB;
                const float dn_pre = dh * cC[2], dz_pre = dh * cC[1], dr_pre = dh * cC[0], dnr = dh * cC[3];
                dgr[0] = on ? dr_pre : 0.f; dgr[1] = on ? dz_pre : 0.f; dgr[2] = on ? dnr : 0.f; dgr[3] = 0.f;
                sv[0] = on ? dr_pre : sv[0]; sv[1] = on ? dz_pre : sv[1]; sv[2] = on ? dn_pre : sv[2];
                svh2 = on ? dnr : svh2;
                dc_next = on ? dh : dc_next;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (F16P) {
                    const unsigned pk = f16_planes(dgr[g], s_grad);
                    g_img[(cur ^ 1) * IMG_BUF + slot * TN_HLD + TEAM_US * g + ul] = (unsigned short)pk;
                    g_img[(cur ^ 1) * IMG_BUF + IMG_PLANE + slot * TN_HLD + TEAM_US * g + ul] = (unsigned short)(pk >> 16);
                } else g_lds[cur ^ 1][slot * TB_GLD + tb_pos(TEAM_US * g + ul)] = dgr[g];
            }
            st_g = on ? goff : st_g;
            f_next = on ? fg : f_next;
            goff = gnx;
            soff = snx;
#ifdef TM_TIMING
            const long long tq5 = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();
#ifdef TM_TIMING
            const long long tq6 = __builtin_amdgcn_s_memtime();
            tacc[0] += tq1 - tq0; tacc[1] += tq2 - tq1; tacc[2] += tq3 - tq2; tacc[3] += tq4 - tq3; tacc[4] += tq5 - tq4; tacc[5] += tq6 - tq5;
#endif
            return dead == 0;
        };
#ifdef TM_TIMING
        const long long tg0 = __builtin_amdgcn_s_memtime();
#endif
        for (int t = tmax - 1; t >= 0; t -= 2) {
            if (!step(t, cur_v, nxt_v, std::integral_constant<int, 0>{})) { failed = true; break; }
            if (t - 1 >= 0 && !step(t - 1, nxt_v, cur_v, std::integral_constant<int, 1>{})) { failed = true; break; }
        }
#ifdef TM_TIMING
        if (lane == 0 && (team == 0 || team == 37) && member == 1)
            printf("team_bwd F16P=%d team %d member %d wave %d plain %d: steps %d  product %.0f  publish %.0f  barrier %.0f  gather %.0f  cell+LDS %.0f  barrier %.0f  loop total/step %.0f (s_memtime ticks per step)\n",
                   (int)F16P, team, member, wave, plain, tmax, (double)tacc[0] / tmax, (double)tacc[1] / tmax, (double)tacc[2] / tmax, (double)tacc[3] / tmax, (double)tacc[4] / tmax,
                   (double)tacc[5] / tmax, (double)(__builtin_amdgcn_s_memtime() - tg0) / tmax);
#endif
        // drain the deferred stores of step 0
#pragma unroll
        for (int g = 0; g < G; ++g) p.dgx[st_g + g * H] = failed ? __builtin_nanf("") : sv[g];
        if constexpr (!LSTM) { p.dgh[st_g] = sv[0]; p.dgh[st_g + H] = sv[1]; p.dgh[st_g + 2 * H] = svh2; }
        __syncthreads();
    }
}

}  // namespace

// Teams to launch for n_seq sequences and the number of rounds (groups of four sequences per team, one after the other) that
// follow.  A multiple of 8 teams lets the four members of every team share an XCD (team_claim_role: L2-scope hand-off); another
// count keeps the teams busy in fewer rounds at the price of device-scope hand-offs (about a quarter slower per step, measured).
static int team_mfma_plan(int n_seq, int n_teams, double& cost_rounds) {
    const int groups = (n_seq + 3) / 4;
    const int nt_f = groups < n_teams ? groups : n_teams;
    const int nt_q = nt_f >= 8 ? (nt_f & ~7) : nt_f;
    const int rounds_f = (groups + nt_f - 1) / nt_f, rounds_q = (groups + nt_q - 1) / nt_q;
    if (nt_q != nt_f && rounds_f * 1.25 < rounds_q) { cost_rounds = rounds_f * 1.25; return nt_f; }
    cost_rounds = rounds_q;
    return nt_q;
}

// MFMA team kernels or VALU team kernels (rnn_team.hip)?  Cost per 256 time steps in us, measured at the end of round 2 (GRU- and
// LSTM-256 alike within 10 %; the MFMA figures: end of round 4).  Both are quantised: the MFMA kernels take a round of 480 (forward) / 510 (backward) per 64 x 4
// sequences; the VALU kernels keep s = ceil(n / 64) sequences in flight per team - s = 1: 390 / 450, 2: 486 / 540, 3: 790 / 940,
// 4: 883 / 1 050, and more than four one after the other (260 sequences: 1 650 / 1 890) - and stop at 768 sequences (per-step
// launches instead: ~19 us per step).  E.g. 64 sequences: VALU (390 / 450); 65 .. 128: the eight-member kernels of rnn_team8.hip take them
// before this model is asked (423 / 421); 129 .. 256: MFMA; 260: MFMA in two rounds; 1 065 chunks (the reference's default shape): MFMA in five.
bool lstm_team_mfma_supported(int cell, int H, int n_seq, int flags, bool backward) {
    if ((cell != 1 && cell != 0) || H != TM_H || n_seq < 65 || (flags & DC_DIMS_TEAM_VALU)) return false;
    double rounds;
    (void)team_mfma_plan(n_seq, 64, rounds);
    const double mfma = rounds * (backward ? 510.0 : 480.0);      // round 4: forward without the k split 473-488 us, backward 490-518
    static const double kValu[2][4] = {{390.0, 486.0, 790.0, 883.0}, {450.0, 540.0, 940.0, 1050.0}};
    const int s = (n_seq + 63) / 64;
    const double valu = n_seq > 768 ? 256.0 * 19.0 : (s <= 4 ? kValu[backward][s - 1] : ((s + 3) / 4) * kValu[backward][3]);
    return mfma <= valu;
}

int lstm_team_mfma_forward(int cell, RnnStepArgs a, int max_len, int n_teams, hipStream_t s) {
    u64* xb = static_cast<u64*>(a.xbuf);
    if (!xb) { set_error("lstm_team_mfma_forward: no exchange buffer (RnnStepArgs::xbuf)", 1012); return 1012; }
    double rounds;
    const int nt = team_mfma_plan(a.n_seq, n_teams, rounds);
    const double G = cell == 1 ? 4.0 : 3.0;
    ProfScope prof(cell == 1 ? "lstm_fwd_team" : "gru_fwd_team", 2.0 * a.n_seq * G * a.H * a.H * max_len, 4.0 * a.n_seq * max_len * a.H * (2.0 * G + 4.0), s);
    if (int rc = zero_async(xb, ((size_t)TEAM_HDR + (size_t)TEAM_MAX * TEAM_M + (size_t)nt * 4 * TEAM_SLOTS * TEAM_H) * sizeof(u64), s)) return rc;
    const bool ksplit = ((a.flags >> DC_DIMS_TEAM_NS_SHIFT) & 7) == 2;      // DC_DIMS_TEAM_NS(2): round 2's forward with the k halves (A/B)
    const dim3 grid(nt * TEAM_M), block(TM_THREADS);
    const int allow = !(a.flags & DC_DIMS_TEAM_DEVICE_SCOPE);
    if (ksplit) {
        if (cell == 1) hipLaunchKernelGGL(team_mfma_fwd_kernel<1>, grid, block, 0, s, a, xb, nt, allow);
        else hipLaunchKernelGGL(team_mfma_fwd_kernel<0>, grid, block, 0, s, a, xb, nt, allow);
    } else {
        // DC_DIMS_F16X2: W_hh and h as f16 planes like the other products of that mode (-DTEAM_FWD_F32: keep the f32 instruction, A/B)
#ifdef TEAM_FWD_F32
        const bool planes = false;
#else
        const bool planes = (a.flags & DC_DIMS_F16X2) && !(a.flags & DC_DIMS_BF16);
#endif
        void (*kern)(RnnStepArgs, u64*, int, int);
        const bool keep = a.fwd_only == 0;
#ifdef TEAM_FWD_COL_PLANES
        const bool own_first = false;                   // A/B: the f16 planes without the own-units-first split
#else
        const bool own_first = planes;
#endif
        if (own_first) {
            if (cell == 1) kern = keep ? team_mfma_fwd_of_kernel<1, true> : team_mfma_fwd_of_kernel<1, false>;
            else kern = keep ? team_mfma_fwd_of_kernel<0, true> : team_mfma_fwd_of_kernel<0, false>;
        } else if (planes) {
            if (cell == 1) kern = keep ? team_mfma_fwd_col_kernel<1, true, true> : team_mfma_fwd_col_kernel<1, true, false>;
            else kern = keep ? team_mfma_fwd_col_kernel<0, true, true> : team_mfma_fwd_col_kernel<0, true, false>;
        } else if (cell == 1) kern = keep ? team_mfma_fwd_col_kernel<1, false, true> : team_mfma_fwd_col_kernel<1, false, false>;
        else kern = keep ? team_mfma_fwd_col_kernel<0, false, true> : team_mfma_fwd_col_kernel<0, false, false>;
        hipLaunchKernelGGL(kern, grid, block, 0, s, a, xb, nt, allow);
    }
    return launch_check("lstm_team_mfma_forward");
}

int lstm_team_mfma_backward(int cell, RnnStepArgs a, int max_len, int n_teams, hipStream_t s) {
    u64* xb = static_cast<u64*>(a.xbuf);
    if (!xb) { set_error("lstm_team_mfma_backward: no exchange buffer (RnnStepArgs::xbuf)", 1012); return 1012; }
    double rounds;
    const int nt = team_mfma_plan(a.n_seq, n_teams, rounds);
    const double G = cell == 1 ? 4.0 : 3.0;
    ProfScope prof(cell == 1 ? "lstm_bwd_team" : "gru_bwd_team", 2.0 * a.n_seq * G * a.H * a.H * max_len, 4.0 * a.n_seq * max_len * a.H * (3.0 * G + 6.0), s);
    if (int rc = zero_async(xb, ((size_t)TEAM_HDR + (size_t)TEAM_MAX * TEAM_M + (size_t)nt * 4 * TEAM_SLOTS * 4 * TEAM_H) * sizeof(u64), s)) return rc;
    // (the member's own partial sums through the ring instead of LDS + barrier - one barrier per step - was measured: 535-540 us against
    // 517-518, profiles/r04/team_fwd_without_k_split.txt)
#ifdef TEAM_BWD_F32
    const bool planes = false;
#else
    const bool planes = (a.flags & DC_DIMS_F16X2) && !(a.flags & DC_DIMS_BF16) && a.s_grad > 0.f;
#endif
    void (*kern)(RnnStepArgs, u64*, int, int);
    if (cell == 1) kern = planes ? team_mfma_bwd_kernel<1, true> : team_mfma_bwd_kernel<1, false>;
    else kern = planes ? team_mfma_bwd_kernel<0, true> : team_mfma_bwd_kernel<0, false>;
    hipLaunchKernelGGL(kern, dim3(nt * TEAM_M), dim3(TM_THREADS), 0, s, a, xb, nt, !(a.flags & DC_DIMS_TEAM_DEVICE_SCOPE));
    return launch_check("lstm_team_mfma_backward");
}

}  // namespace dc
