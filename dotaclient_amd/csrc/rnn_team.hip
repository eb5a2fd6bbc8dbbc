// Persistent recurrent kernels for H = 256 (the reference's GRU, /root/reference/policy.py:66,141, and
// the LSTM-256 configs of BASELINE.json): all time steps of a layer in ONE launch, W_hh stationary in
// registers, the sequence's state exchanged between FOUR workgroups per step.
//
// Why a team.  W_hh of a 256-wide cell is 768 KB (GRU) / 1 MB (LSTM) of f32: more than the 512 KB of
// vector registers of one CU, so the one-workgroup-per-sequence scheme of rnn_persist_valu.hip stops at
// H = 128 and these cells used to run one launch per time step (5.7-6.4 us per launch: 19 of the 21 ms
// of a GRU-256 64x256 bench step).  Here a sequence is owned by a TEAM of four workgroups on four CUs;
// member m keeps the recurrent weights of hidden units [64m, 64m+64) (all gates, all k: 128 weights per
// lane, 512 lanes - exactly the register budget of the H = 128 kernel) and per step
//   forward : computes its 64 units' gates and h_t, publishes the 64 values, collects the other 192;
//   backward: computes its 64 units' gate gradients (the contraction W_hh^T dgates runs over ALL gates,
//             so it keeps columns [64m, 64m+64) of W_hh), publishes 256 values, collects the other 768.
//
// Exchange protocol (placement independent: nothing below assumes which CU or XCD a workgroup lands on;
// MI355X_MICROARCH.md "handoff-1to1"): every value travels as an 8-byte granule {f32 value, u32 tag}
// written with ONE device-scope (sc1, write-through) store and read with device-scope loads, so a granule
// is either old or complete - no separate flag, no fence, no ordering between granules needed.  The tag
// is the team's running step counter (+1 per step, continuing across the sequences a team works through;
// the buffer is zeroed before the launch and tags start at 1).  Granules live in a ring of four slots
// (tag & 3): a member can be at most one step plus one sequence boundary ahead of a peer, so the slot it
// overwrites (tag - 4) has been consumed by everyone.  Speed only, never correctness: blocks 8 apart form a team, which the
// usual round-robin placement puts on one XCD; the members compare their XCC ids at start (one granule each) and, if they
// do share an XCD (hence an L2), publish with L2-scope stores instead of write-through ones (a lone sequence's step:
// 1.19 -> 0.97 us); any other placement keeps the device-scope stores.  Every spin is bounded; a member that gives up
// poisons its outputs with NaN (the loss turns NaN and the optimizer raises, optimizer.py:667) instead of
// hanging the GPU.  The grid does NOT have to be resident at once: roles are taken by ticket when a workgroup starts
// (team_claim_role), so a team only ever waits for workgroups that are running; the host still sizes the grid from
// the CU count (one 512-thread workgroup with ~180 registers per lane per CU) because that is the fastest shape.
//
// Lane roles (512 threads).  Both kernels end a step with gate q = lane & 3 of one hidden unit in every
// DPP quad, each (unit, gate) held by two lanes ("dup" 0/1, which share the stores).
//   forward : row = tid >> 4 owns local units 2*row, 2*row+1 (8 gate columns); its 16 lanes split k
//             (16 each = four ds_read_b128 of h_{t-1}); 8 columns x 16 k = 64 v_pk_fma_f32 per lane.
//             Reduce-scatter half_mirror / xor2 / xor1 (4+2+1 v_add_f32_dpp) then row_ror:8 as an
//             all-reduce: lane l holds column l & 7.  The register -> column permutation is colmap & 7.
//   backward: wave w owns local units 8w..8w+7; its 64 lanes split the 1024-long contraction (the four
//             gate gradients of units lane, 64+lane, 128+lane, 192+lane); v_permlane32_swap (4),
//             v_permlane16_swap (2), row_ror:8 (1) scatter, then an 8-lane all-reduce: lane l holds
//             output l >> 3.
// The GRU has three gates: it runs in the same four-slot layout with zero weights in slot 3 (a quarter of
// the FMAs is wasted; a step is bound by the exchange latency, not by FMA issue).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "kernels.h"
#include "valu_util.h"
#include "team_util.h"

namespace dc {

// ---------------------------------------------------------------------------------------------------
// forward.  Same contract as the per-step kernels of rnn.hip: gates[row][G*H] = W_ih x + b_ih on entry,
// activated gates on exit; hprev/cprev[first row] = h0/c0; h_t (c_t) -> hseq (cseq)[row] and
// hprev (cprev)[row+1]; GRU: hn[row] = W_hn h + b_hn.
// NS = sequences a team works on concurrently ("streams", round-robin one step each): a step is ~0.45 us of
// work and ~1 us of waiting for the peers' granules, so with more sequences than teams the wait of one
// stream is filled with the work of the others.  Stream s of team T owns sequences T*NS + s + k*NS*teams.
// ---------------------------------------------------------------------------------------------------
// early granule read of a lone stream right behind its publish (see the end of step): measured SLOWER (forward 305 -> 386 us
// per 256 steps): the peers publish at the same moment, a read issued now mostly comes back stale and costs a second round trip
constexpr bool EARLY1 = false;

template <int CELL, int NS, bool TIMING = false>   // TIMING (DC_TEAM_TIMING=1): s_memtime phase sums of wave 0 of block 0 -> p.dbg
__global__ __launch_bounds__(512) void rnn_team_fwd_kernel(RnnStepArgs p, u64* __restrict__ xbuf_all, int n_teams, int allow_plain) {
    constexpr int H = TEAM_H, G = CELL == CELL_GRU ? 3 : 4, GH = G * H;
    constexpr int KPL = 16, NRD = 4;
    long long tm[6] = {0, 0, 0, 0, 0, 0}, tm0 = 0;
    auto stamp = [&](int i) {
        if constexpr (TIMING) { const long long x = __builtin_amdgcn_s_memtime(); tm[i] += x - tm0; tm0 = x; }
    };
    __shared__ __attribute__((aligned(16))) float h_lds[NS][2][H];
    __shared__ int dead;
    const int tid = threadIdx.x;
    const int kg = tid & 15, row = tid >> 4;
    const int q = tid & 3, dup = (tid >> 3) & 1;
    int team, member;
    team_claim_role(reinterpret_cast<unsigned*>(xbuf_all), n_teams, team, member);
    if (team < 0) return;                                          // more workgroups than roles: cannot happen (grid = 4 x teams)
    u64* const xbuf = xbuf_all + TEAM_HDR + TEAM_MAX * TEAM_M;      // [tickets | handshake granules | rings]
    const int plain = team_same_xcd(xbuf_all + TEAM_HDR + team * TEAM_M, member, allow_plain);
    const int U0 = member * TEAM_US;
    const int u = U0 + 2 * row + ((tid >> 2) & 1);

    // ---- weights: pair m = registers 2m, 2m+1; element kk <-> k = 64*(kk>>2) + 4*kg + (kk&3) ----------
    f32x2 wp[4][KPL];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = colmap(kg, 2 * m + e) & 7;
            const int gate = c & 3;
            const bool on = gate < G;
            const float* src = p.Whh + (size_t)((on ? gate : 0) * H + U0 + 2 * row + (c >> 2)) * H + 4 * kg;
#pragma unroll
            for (int i = 0; i < NRD; ++i) {
                float4 v = *reinterpret_cast<const float4*>(src + 64 * i);
                if (!on) v = make_float4(0.f, 0.f, 0.f, 0.f);
                wp[m][4 * i + 0][e] = v.x; wp[m][4 * i + 1][e] = v.y; wp[m][4 * i + 2][e] = v.z; wp[m][4 * i + 3][e] = v.w;
            }
        }
    }
    const int gq = q < G ? q : G - 1;                              // slot 3 of the GRU shadows the n lane's addresses
    const float bq = q < G ? p.bhh[q * H + u] : 0.f;
    const bool is_t = q == 2;                                     // the tanh gate (LSTM g, GRU n)
    const float sc = (CELL == CELL_LSTM && is_t) ? -2.8853900817779268f : -1.4426950408889634f;
    const float am = (CELL == CELL_LSTM && is_t) ? 2.f : 1.f, aa = (CELL == CELL_LSTM && is_t) ? -1.f : 0.f;
    if (tid == 0) dead = 0;

    // ---- per-stream state (every loop over s is unrolled: registers / SGPRs) ---------------------------------
    int sq[NS], len[NS], tt[NS];
    size_t row0[NS];
    unsigned tag[NS];      // tag of the stream's previous step output (running counter, continues across sequences)
    bool on[NS];
    float st[NS];          // c_{t-1} (LSTM) / h_{t-1} (GRU) of this lane's unit
    float x0[NS], x1[NS];  // gate pre-activations of steps t, t+1 (loaded two steps ahead)
    const int pidx = (U0 + TEAM_US + tid) & (H - 1);   // the granule lanes 0..191 collect
    u64 pre = 0;           // early first read of a stream's granule (see step)
    int pre_owner = -1;
    auto gate_ptr = [&](int s) { return p.gates + row0[s] * GH + gq * H + u; };
    auto open = [&](int s) {   // next non-empty sequence of stream s, or retire the stream
        for (;;) {
            sq[s] += NS * n_teams;
            if (sq[s] >= p.n_seq) { on[s] = false; return; }
            len[s] = p.seq_len[sq[s]];
            if (len[s] > 0) break;
        }
        row0[s] = (size_t)p.seq_off[sq[s]];
        tt[s] = 0;
        // initial state from h0/c0 (zeros when absent); every member writes its own 64 units of the first row's
        // hprev/cprev, where the backward and the dW_hh product expect it
        const size_t sb = (size_t)sq[s] * H;
        if constexpr (CELL == CELL_LSTM) {
            st[s] = p.c0 ? p.c0[sb + u] : 0.f;
            if (dup == 0 && q == 0) p.cprev[row0[s] * H + u] = st[s];
        } else {
            st[s] = p.h0 ? p.h0[sb + u] : 0.f;
        }
        const float* gp = gate_ptr(s);
        x0[s] = gp[0];
        x1[s] = gp[(size_t)min(1, len[s] - 1) * GH];
        __syncthreads();                                           // the previous sequence's last reads of h_lds[s]
        if (tid < H) {
            const float h = p.h0 ? p.h0[sb + tid] : 0.f;
            h_lds[s][0][tid] = h;
            if (tid >= U0 && tid < U0 + TEAM_US) p.hprev[row0[s] * H + tid] = h;
        }
        // wait for these loads inside this (rare) path: otherwise the wait lands after the join with the path that
        // opens nothing, as a vmcnt(0) behind that path's freshly issued stores
        asm volatile("" : "+v"(x0[s]), "+v"(x1[s]), "+v"(st[s]) : : "memory");
    };
    // outputs of a step: one store per lane plus the granules
    float* pend_ptr = nullptr; float pend_val = 0.f, pend_h = 0.f; bool pend_on = false, pending = false;
    u64* pend_gr = nullptr; unsigned pend_tag = 0;
    auto flush = [&]() {
        if (!pending) return;
        if (dup == 0 && q == 0) granule_store(pend_gr, pend_h, pend_tag, plain);
        if (pend_on) *pend_ptr = pend_val;
        pending = false;
    };
    auto step = [&](int s) -> bool {
        stamp(0);      // between step calls (loop control, open)
        const int t = tt[s], par = t & 1;
        u64* const xb = xbuf + (size_t)(team * NS + s) * (TEAM_SLOTS * H);
        float* const gp = gate_ptr(s);
        const float x = x0[s];
        auto next_operands = [&]() {
            x0[s] = x1[s];
            x1[s] = gp[(size_t)min(t + 2, len[s] - 1) * GH];
        };
        // several streams: issued first, waited for with the early granule read at the end of the arithmetic (the hot path
        // of the poll below has no wait of its own).  One stream: after the poll, whose vmcnt(0) must not include it.
        if constexpr (NS > 1) next_operands();
        if (t > 0 && tid < H - TEAM_US) {                           // the other members' h_{t-1}
            const u64* gptr = xb + (tag[s] & 3) * H + pidx;
            float v = 0.f;
            u64 g = pre;
            if (pre_owner != s) {   // no early read for this step: read now, and wait for it inside this branch (a wait after
                g = granule_load(gptr);                            // the join would be a vmcnt(0) behind the last stores)
                asm volatile("" : "+v"(g) : : "memory");
            }
            if constexpr (TIMING) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(g) : : "memory");
                stamp(1);                                          // first read back
                if ((unsigned)(g >> 32) != tag[s]) tm[5] += 1;     // ... and it was stale
            }
            if (!granule_wait(g, gptr, tag[s], v)) { dead = 1; team_report_timeout(p.fault, TEAM_K_VALU_FWD, p.layer, team, member, t, sq[s], tag[s]); }
            h_lds[s][par][pidx] = v;
        }
        stamp(2);      // spinning
        if constexpr (!(NS == 1 && EARLY1)) pre_owner = -1;
        __syncthreads();
        stamp(3);      // barrier
        if (NS > 1) {
            // first read of the NEXT stream's granules now, so that its round trip (~0.7 us even when the data is
            // there) runs under this step's arithmetic; that stream's step starts from the value read here
            const int n = (s + 1) % NS;
            if (on[n] && tt[n] > 0) {
                if (tid < H - TEAM_US) pre = granule_load(xbuf + (size_t)(team * NS + n) * (TEAM_SLOTS * H) + (tag[n] & 3) * H + pidx);
                pre_owner = n;
            }
        }
        if constexpr (NS == 1) next_operands();
        const float* hl = &h_lds[s][par][4 * kg];
        f32x2 hv[KPL / 2];
#pragma unroll
        for (int i = 0; i < NRD; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(hl + 64 * i);
            hv[2 * i] = __builtin_shufflevector(v, v, 0, 1);
            hv[2 * i + 1] = __builtin_shufflevector(v, v, 2, 3);
        }
        const int is_dead = dead;
        f32x2 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = pk_mul_bcast0(wp[m][0], hv[0]);
#pragma unroll
        for (int kk = 1; kk < KPL; ++kk)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (kk & 1) pk_fma_bcast<1>(acc[m], wp[m][kk], hv[kk >> 1]);
                else pk_fma_bcast<0>(acc[m], wp[m][kk], hv[kk >> 1]);
            }
        float a[8];
#pragma unroll
        for (int m = 0; m < 4; ++m) { a[2 * m] = acc[m].x; a[2 * m + 1] = acc[m].y; }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) a[cc] += dpp<DPP_HALF_MIRROR>(a[4 + cc]);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) a[cc] += dpp<DPP_XOR2>(a[2 + cc]);
        a[0] += dpp<DPP_XOR1>(a[1]);
        a[0] += dpp<DPP_ROR8>(a[0]);
        const float ah = a[0] + bq;                                // W_hh h + b_hh of this lane's gate
        const size_t r = row0[s] + t;
        const bool more = t + 1 < len[s];
        float hn;
        // every lane has (at most) ONE output store per step: dup 0 the activated gate (GRU slot 3: hn), dup 1 the states
        float* o_ptr;
        float o_val;
        bool o_on = true;
        if constexpr (CELL == CELL_LSTM) {
            const float act = __builtin_fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(sc * (ah + x))), am, aa);
            const float ig = dpp<DPP_Q0>(act), fg = dpp<DPP_Q1>(act), gg = dpp<DPP_Q2>(act), og = dpp<DPP_Q3>(act);
            const float cn = fg * st[s] + ig * gg;
            hn = og * tanh_hw(cn);
            st[s] = cn;
            // dup 1: q 0,1: h_t, c_t -> hseq/cseq[row]; q 2,3: -> hprev/cprev[row+1] (last step: the same value again)
            float* base = (q & 1) ? ((q >= 2 && more) ? p.cprev + H : p.cseq) : ((q >= 2 && more) ? p.hprev + H : p.hseq);
            o_ptr = dup == 0 ? gp + (size_t)t * GH : base + r * H + u;
            o_val = dup == 0 ? act : ((q & 1) ? cn : hn);
        } else {
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(sc * (ah + x)));   // r, z lanes
            const float rg = dpp<DPP_Q0>(sg);
            const float ng = tanh_hw(x + rg * ah);                                                  // n lane
            const float act = is_t ? ng : sg;
            const float zg = dpp<DPP_Q1>(act), nn = dpp<DPP_Q2>(act), hnv = dpp<DPP_Q2>(ah);
            hn = (1.f - zg) * nn + zg * st[s];
            st[s] = hn;
            if (dup == 0) {
                o_ptr = q < 3 ? gp + (size_t)t * GH : p.hn + r * H + u;
                o_val = q < 3 ? act : hnv;
            } else {
                o_ptr = (q == 0 ? p.hseq : p.hprev + H) + r * H + u;
                o_val = hn;
                o_on = q == 0 || (q == 1 && more);
            }
        }
        ++tag[s];
        if (dup == 0 && q == 0) h_lds[s][par ^ 1][u] = hn;
        pend_ptr = o_ptr; pend_val = o_val; pend_on = o_on;
        pend_gr = xb + (tag[s] & 3) * H + u; pend_h = hn; pend_tag = tag[s];
        pending = true;
        // Several streams: the loads of this step call (next operands, the next stream's granules) are waited for HERE,
        // with a step's arithmetic behind them and before this step's stores are issued.  Left to the compiler the wait
        // lands at the loop back-edge (register rotation of x0/x1) as a vmcnt(0) behind the stores: their acknowledgement
        // latency in every step call (measured: 480-700 cycles).  One stream: the poll is the critical path and the
        // publish must not wait for anything.
        if constexpr (NS > 1) asm volatile("" : "+v"(x1[s]), "+v"(pre) : : "memory");
        // (holding the stores back until after the next step call's poll was measured and is slower: the later publish
        // costs the peers more than it saves)
        flush();
        if constexpr (NS == 1 && EARLY1) {
            // one stream: a first read of the peers' granules of THIS step right behind the publish, so that its round trip
            // overlaps the wait for the store acknowledgements at the loop back-edge instead of following it
            if (more && tid < H - TEAM_US) pre = granule_load(xb + (tag[s] & 3) * H + pidx);
            pre_owner = more ? s : -1;
        }
        stamp(4);      // arithmetic + stores
        return is_dead == 0;
    };

#pragma unroll
    for (int s = 0; s < NS; ++s) {
        sq[s] = team * NS + s - NS * n_teams; tag[s] = 0; on[s] = true; len[s] = 0; tt[s] = 0; row0[s] = 0;
        st[s] = x0[s] = x1[s] = 0.f;
        open(s);
    }
    if constexpr (TIMING) tm0 = __builtin_amdgcn_s_memtime();
    bool failed = false;
    size_t fail_row = 0;
    for (bool any = true; any && !failed;) {
        any = false;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (!on[s] || failed) continue;
            const size_t r_now = row0[s] + tt[s];
            if (!step(s)) { failed = true; fail_row = r_now; continue; }
            if (++tt[s] == len[s]) open(s);
            any |= on[s];
        }
    }
    flush();
    // a peer never answered: make the failure visible downstream
    if (failed && tid < TEAM_US) p.hseq[fail_row * H + U0 + tid] = __builtin_nanf("");
    if constexpr (TIMING) {
        if (blockIdx.x == 0 && tid == 0 && p.dbg != nullptr) for (int i = 0; i < 6; ++i) p.dbg[i] = tm[i];
    }
}

// ---------------------------------------------------------------------------------------------------
// backward through time.  In: dh[row][H] (from above), the forward's activated gates, cseq/cprev (LSTM)
// or hn/hprev (GRU).  Out: dgx[row][G*H] and, GRU, dgh[row][G*H] (the n gate's differs by the factor r).
// ---------------------------------------------------------------------------------------------------
template <int CELL, int NS>
__global__ __launch_bounds__(512) void rnn_team_bwd_kernel(RnnStepArgs p, u64* __restrict__ xbuf_all, int n_teams, int allow_plain) {
    constexpr int H = TEAM_H, G = CELL == CELL_GRU ? 3 : 4, GH = G * H, NG = 4 * H;   // NG: granules / LDS positions per step
    constexpr int KPL = 16, NRD = 4;
    __shared__ __attribute__((aligned(16))) float g_lds[NS][2][NG];   // position 4*unit + gate slot
    __shared__ int dead;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 3, dup = (lane >> 2) & 1, b3 = (lane >> 3) & 1;
    int team, member;
    team_claim_role(reinterpret_cast<unsigned*>(xbuf_all), n_teams, team, member);
    if (team < 0) return;                                          // more workgroups than roles: cannot happen (grid = 4 x teams)
    u64* const xbuf = xbuf_all + TEAM_HDR + TEAM_MAX * TEAM_M;      // [tickets | handshake granules | rings]
    const int plain = team_same_xcd(xbuf_all + TEAM_HDR + team * TEAM_M, member, allow_plain);
    const int U0 = member * TEAM_US;
    const int u = U0 + 8 * wave + (lane >> 3);

    // ---- weights.  Register r <-> output U0 + 8*wave + ((r & 6) | ((r & 1) ^ b3)); element kk <-> LDS position
    // 256*(kk>>2) + 4*lane + (kk&3) = gate slot (kk&3) of unit 64*(kk>>2) + lane.
    f32x2 wp[4][KPL];
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
        const int gate = kk & 3;
        const bool on = gate < G;
        const int col = (on ? gate : 0) * H + 64 * (kk >> 2) + lane;
        const float4* src = reinterpret_cast<const float4*>(p.Whh + (size_t)col * H + U0 + 8 * wave);
#pragma unroll
        for (int g4 = 0; g4 < 2; ++g4) {
            float4 v = src[g4];
            if (!on) v = make_float4(0.f, 0.f, 0.f, 0.f);
            // register 4*g4 + j holds component j ^ b3
            wp[2 * g4][kk][0] = b3 ? v.y : v.x; wp[2 * g4][kk][1] = b3 ? v.x : v.y;
            wp[2 * g4 + 1][kk][0] = b3 ? v.w : v.z; wp[2 * g4 + 1][kk][1] = b3 ? v.z : v.w;
        }
    }
    const int gq = q < G ? q : G - 1;
    const bool is_q0 = q == 0, is_q1 = q == 1, is_q2 = q == 2, is_q3 = q == 3;
    // LSTM: q0 c_t, q1 c_{t-1}, q2 dh (q3: dh again, unused).  GRU: q0 hn, q1 h_{t-1}, q2 dh.
    const float* const sh_base = CELL == CELL_LSTM ? (q == 0 ? p.cseq : (q == 1 ? p.cprev : p.dh)) : (q == 0 ? p.hn : (q == 1 ? p.hprev : p.dh));
    if (tid == 0) dead = 0;

    int sq[NS], len[NS], ii[NS];   // ii: step index from the end, t = len - 1 - ii
    size_t row0[NS];
    unsigned tag[NS];
    bool on[NS];
    float carry[NS], gate_next[NS];      // LSTM: dc_{t+1}, f_{t+1};  GRU: dh_{t+1}, z_{t+1}
    float a0[NS], a1[NS], s0[NS], s1[NS];   // own gate activation / state operand of steps ii, ii+1
    // the 768 granules of the other members: every lane collects one, lanes 0..255 a second
    const int i0 = (4 * U0 + 4 * TEAM_US + tid) & (NG - 1), i1 = (4 * U0 + 4 * TEAM_US + 512 + tid) & (NG - 1);
    u64 pre0 = 0, pre1 = 0;
    int pre_owner = -1;
    auto open = [&](int s) {
        for (;;) {
            sq[s] += NS * n_teams;
            if (sq[s] >= p.n_seq) { on[s] = false; return; }
            len[s] = p.seq_len[sq[s]];
            if (len[s] > 0) break;
        }
        row0[s] = (size_t)p.seq_off[sq[s]];
        ii[s] = 0;
        carry[s] = 0.f; gate_next[s] = 0.f;
        const float* gp = p.gates + row0[s] * GH + gq * H + u;
        const float* shp = sh_base + row0[s] * H + u;
        const size_t r0 = (size_t)(len[s] - 1), r1 = (size_t)max(len[s] - 2, 0);
        a0[s] = gp[r0 * GH]; s0[s] = shp[r0 * H];
        a1[s] = gp[r1 * GH]; s1[s] = shp[r1 * H];
        __syncthreads();                                           // the previous sequence's last reads of g_lds[s]
        g_lds[s][0][tid] = 0.f; g_lds[s][0][512 + tid] = 0.f;      // "step len" has no gate gradient
        asm volatile("" : "+v"(a0[s]), "+v"(a1[s]), "+v"(s0[s]), "+v"(s1[s]) : : "memory");   // (see the forward)
    };
    float* pend_ptr = nullptr; float pend_val = 0.f, pend_d = 0.f; bool pend_on = false, pending = false;
    u64* pend_gr = nullptr; unsigned pend_tag = 0;
    auto flush = [&]() {
        if (!pending) return;
        if (dup == 0) granule_store(pend_gr, pend_d, pend_tag, plain);
        if (pend_on) *pend_ptr = pend_val;
        pending = false;
    };
    auto step = [&](int s) -> bool {
        const int i = ii[s], t = len[s] - 1 - i, cur = i & 1;
        u64* const xb = xbuf + (size_t)(team * NS + s) * (TEAM_SLOTS * NG);
        const float a_own = a0[s], shv = s0[s];
        auto next_operands = [&]() {   // operands of step i + 2
            const size_t r2 = (size_t)max(t - 2, 0);
            a0[s] = a1[s]; s0[s] = s1[s];
            a1[s] = p.gates[(row0[s] + r2) * GH + gq * H + u];
            s1[s] = sh_base[(row0[s] + r2) * H + u];
        };
        if constexpr (NS > 1) next_operands();   // (placement: see the forward)
        if (i > 0) {                                               // the other members' gate gradients of step t+1
            float v0 = 0.f, v1 = 0.f;
            const u64* base = xb + (tag[s] & 3) * NG;
            u64 g0 = pre0, g1 = pre1;
            if (pre_owner != s) {   // (see the forward)
                g0 = granule_load(base + i0);
                if (tid < 256) g1 = granule_load(base + i1);
                asm volatile("" : "+v"(g0), "+v"(g1) : : "memory");
            }
            bool ok = granule_wait(g0, base + i0, tag[s], v0);
            g_lds[s][cur][i0] = v0;
            if (tid < 256) {
                ok = granule_wait(g1, base + i1, tag[s], v1) && ok;
                g_lds[s][cur][i1] = v1;
            }
            if (!ok) { dead = 1; team_report_timeout(p.fault, TEAM_K_VALU_BWD, p.layer, team, member, t, sq[s], tag[s]); }
        }
        pre_owner = -1;
        __syncthreads();
        if (NS > 1) {   // early first read of the next stream's granules (see the forward)
            const int n = (s + 1) % NS;
            if (on[n] && ii[n] > 0) {
                const u64* nb = xbuf + (size_t)(team * NS + n) * (TEAM_SLOTS * NG) + (tag[n] & 3) * NG;
                pre0 = granule_load(nb + i0);
                if (tid < 256) pre1 = granule_load(nb + i1);
                pre_owner = n;
            }
        }
        if constexpr (NS == 1) next_operands();
        const float* gl = &g_lds[s][cur][4 * lane];
        f32x2 dv[KPL / 2];
#pragma unroll
        for (int j = 0; j < NRD; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(gl + 256 * j);
            dv[2 * j] = __builtin_shufflevector(v, v, 0, 1);
            dv[2 * j + 1] = __builtin_shufflevector(v, v, 2, 3);
        }
        const int is_dead = dead;
        f32x2 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = pk_mul_bcast0(wp[m][0], dv[0]);
#pragma unroll
        for (int kk = 1; kk < KPL; ++kk)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (kk & 1) pk_fma_bcast<1>(acc[m], wp[m][kk], dv[kk >> 1]);
                else pk_fma_bcast<0>(acc[m], wp[m][kk], dv[kk >> 1]);
            }
        float a[8];
#pragma unroll
        for (int m = 0; m < 4; ++m) { a[2 * m] = acc[m].x; a[2 * m + 1] = acc[m].y; }
        float s4[4], s2[2];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) s4[cc] = swap32_sum(a[cc], a[4 + cc]);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) s2[cc] = swap16_sum(s4[cc], s4[2 + cc]);
        float rec = s2[0] + dpp<DPP_ROR8>(s2[1]);
        rec += dpp<DPP_HALF_MIRROR>(rec);
        rec += dpp<DPP_XOR1>(rec);
        rec += dpp<DPP_XOR2>(rec);            // dh_rec[u]: zero at the sequence's last step (g_lds starts zeroed)
        const size_t r = row0[s] + t;
        float d, dh_out;
        float* o_ptr; float o_val; bool o_on;      // the lane's one output store of the step
        if constexpr (CELL == CELL_LSTM) {
            const float ig = dpp<DPP_Q0>(a_own), fg = dpp<DPP_Q1>(a_own), gg = dpp<DPP_Q2>(a_own), og = dpp<DPP_Q3>(a_own);
            const float cs = dpp<DPP_Q0>(shv), cp = dpp<DPP_Q1>(shv), dhx = dpp<DPP_Q2>(shv);
            const float dh = dhx + rec;
            const float tc = tanh_hw(cs);
            const float dcv = dh * og * (1.f - tc * tc) + carry[s] * gate_next[s];
            const float M = is_q3 ? dh * tc : dcv;
            float X = is_q2 ? ig : 1.f;
            X = is_q1 ? cp : X;
            X = is_q0 ? gg : X;
            const float D = __builtin_fmaf(-a_own, a_own, is_q2 ? 1.f : a_own);
            d = M * X * D;
            dh_out = d;
            carry[s] = dcv;
            gate_next[s] = fg;
            o_ptr = p.dgx + r * GH + q * H + u; o_val = d; o_on = dup == 0;
        } else {
            const float rg = dpp<DPP_Q0>(a_own), zg = dpp<DPP_Q1>(a_own), ng = dpp<DPP_Q2>(a_own);
            const float hnv = dpp<DPP_Q0>(shv), hpv = dpp<DPP_Q1>(shv), dhx = dpp<DPP_Q2>(shv);
            const float dh = dhx + rec + carry[s] * gate_next[s];
            const float dn = dh * (1.f - zg) * (1.f - ng * ng);
            const float dz = dh * (hpv - ng) * zg * (1.f - zg);
            const float dr = dn * hnv * rg * (1.f - rg);
            d = is_q0 ? dr : (is_q1 ? dz : (is_q2 ? dn : 0.f));
            dh_out = is_q2 ? dn * rg : d;
            carry[s] = dh;
            gate_next[s] = zg;
            o_ptr = (dup == 0 ? p.dgx : p.dgh) + r * GH + gq * H + u; o_val = dup == 0 ? d : dh_out; o_on = q < 3;
        }
        ++tag[s];
        if (dup == 0) g_lds[s][cur ^ 1][4 * u + q] = dh_out;
        if constexpr (NS > 1) asm volatile("" : "+v"(a1[s]), "+v"(s1[s]), "+v"(pre0), "+v"(pre1) : : "memory");   // (see the forward)
        pend_ptr = o_ptr; pend_val = o_val; pend_on = o_on;
        pend_gr = xb + (tag[s] & 3) * NG + 4 * u + q; pend_d = dh_out; pend_tag = tag[s];
        pending = true;
        flush();
        return is_dead == 0;
    };

#pragma unroll
    for (int s = 0; s < NS; ++s) {
        sq[s] = team * NS + s - NS * n_teams; tag[s] = 0; on[s] = true; len[s] = 0; ii[s] = 0; row0[s] = 0;
        carry[s] = gate_next[s] = a0[s] = a1[s] = s0[s] = s1[s] = 0.f;
        open(s);
    }
    bool failed = false;
    size_t fail_row = 0;
    for (bool any = true; any && !failed;) {
        any = false;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (!on[s] || failed) continue;
            const size_t r_now = row0[s] + (len[s] - 1 - ii[s]);
            if (!step(s)) { failed = true; fail_row = r_now; continue; }
            if (++ii[s] == len[s]) open(s);
            any |= on[s];
        }
    }
    flush();
    if (failed && tid < TEAM_US) p.dgx[fail_row * GH + U0 + tid] = __builtin_nanf("");
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
namespace {

// exchange buffer = [ticket counters | handshake granules | ring of the largest case (backward: 4H granules per slot)];
// it is part of the CALLER's workspace (DC_WS_TEAM_XBUF, RnnStepArgs::xbuf): nothing is allocated here, and two engines
// on two streams never share one
constexpr size_t TEAM_XBUF_WORDS = (size_t)TEAM_HDR + (size_t)TEAM_MAX * TEAM_M + (size_t)TEAM_MAX * TEAM_NS_MAX * TEAM_SLOTS * 4 * TEAM_H;

// teams that can be resident together: one 512-thread workgroup per CU, four per team
int team_capacity() {
    static const int cap = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return 0;
        int t = cus / TEAM_M;
        t = t > TEAM_MAX ? TEAM_MAX : t;
        return t >= 8 ? (t & ~7) : t;   // multiples of 8 keep the members of a team on one XCD
    }();
    return cap;
}

int team_count(int n_seq) {
    const int cap = team_capacity();
    int t = n_seq < cap ? n_seq : cap;
    if (t >= 8) t &= ~7;
    return t;
}

// DC_DIMS_TEAM_DEVICE_SCOPE: device-scope (write-through) granule stores even for teams that sit on one XCD (A/B measurements)
int team_plain(int flags) { return !(flags & DC_DIMS_TEAM_DEVICE_SCOPE); }

// sequences a team keeps in flight: as many as it has to walk through anyway, up to four (a step is ~1/3 work,
// ~2/3 waiting for the peers).  DC_DIMS_TEAM_NS(1 | 2 | 4) forces a count (A/B measurements, tests).
int team_streams(int n_seq, int nt, int flags) {
    const int forced = (flags >> DC_DIMS_TEAM_NS_SHIFT) & 7;
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    const int per = (n_seq + nt - 1) / nt;
    return per >= 3 ? 4 : (per == 2 ? 2 : 1);
}

template <int CELL>
void launch_team_fwd(int ns, int nt, const RnnStepArgs& a, u64* xb, hipStream_t s) {
    const dim3 grid(nt * TEAM_M), block(512);
    if (ns == 4) hipLaunchKernelGGL((rnn_team_fwd_kernel<CELL, 4>), grid, block, 0, s, a, xb, nt, team_plain(a.flags));
    else if (ns == 2) hipLaunchKernelGGL((rnn_team_fwd_kernel<CELL, 2>), grid, block, 0, s, a, xb, nt, team_plain(a.flags));
    else hipLaunchKernelGGL((rnn_team_fwd_kernel<CELL, 1>), grid, block, 0, s, a, xb, nt, team_plain(a.flags));
}
template <int CELL>
void launch_team_bwd(int ns, int nt, const RnnStepArgs& a, u64* xb, hipStream_t s) {
    const dim3 grid(nt * TEAM_M), block(512);
    if (ns == 4) hipLaunchKernelGGL((rnn_team_bwd_kernel<CELL, 4>), grid, block, 0, s, a, xb, nt, team_plain(a.flags));
    else if (ns == 2) hipLaunchKernelGGL((rnn_team_bwd_kernel<CELL, 2>), grid, block, 0, s, a, xb, nt, team_plain(a.flags));
    else hipLaunchKernelGGL((rnn_team_bwd_kernel<CELL, 1>), grid, block, 0, s, a, xb, nt, team_plain(a.flags));
}

}  // namespace

long long rnn_team_xbuf_bytes() {
    const long long a = (long long)(TEAM_XBUF_WORDS * sizeof(u64)), b = rnn_team8_xbuf_bytes();
    return a > b ? a : b;
}

// (DC_DIMS_RNN_PER_STEP, checked by the caller, forces the launch-per-step kernels)
bool rnn_team_supported(int cell, int H, int n_seq, int flags) {
    if (H != TEAM_H || (cell != CELL_GRU && cell != CELL_LSTM) || team_capacity() < 1) return false;
    if (rnn_team8_supported(cell, H, n_seq, flags)) return true;       // DC_DIMS_TEAM8: any number of sequences
    // The MFMA team kernels (rnn_team_mfma.hip, both cells) where their rounds of 64 x 4 sequences are cheaper than the alternative
    // (cost model there): no upper limit on the number of sequences - 1 065 chunks of 16 steps, the reference's default shape,
    // take 215 / 205 us per forward / backward pass against 16 launches of 17-27 us.
    if (lstm_team_mfma_supported(cell, H, n_seq, flags, false)) return true;
    // VALU team kernels, measured at LSTM-256, 256 steps: 64 sequences 0.39 vs 2.2 ms per pass, 256: 0.86 vs 2.4 ms, 1024: 3.4 vs
    // 3.8 ms - beyond that the batched per-step launches (f32 MFMA, all sequences at once) win again
    return n_seq <= 12 * TEAM_MAX;
}

int rnn_team_forward(int cell, RnnStepArgs a, int max_len, hipStream_t s) {
    if (rnn_team8_supported(cell, a.H, a.n_seq, a.flags)) return rnn_team8_forward(cell, a, max_len, s);
    if (lstm_team_mfma_supported(cell, a.H, a.n_seq, a.flags, false)) return lstm_team_mfma_forward(cell, a, max_len, team_capacity(), s);
    u64* xb = static_cast<u64*>(a.xbuf);
    if (!xb) { set_error("rnn_team_forward: no exchange buffer (RnnStepArgs::xbuf)", 1012); return 1012; }
    const int nt = team_count(a.n_seq), ns = team_streams(a.n_seq, nt, a.flags);
    const double G = cell == CELL_GRU ? 3 : 4;
    ProfScope prof(cell == CELL_GRU ? "gru_fwd_team" : "lstm_fwd_team", 2.0 * a.n_seq * G * a.H * a.H * max_len,
                   4.0 * a.n_seq * max_len * a.H * (2.0 * G + 4.0), s);
    if (int rc = zero_async(xb, ((size_t)TEAM_HDR + (size_t)TEAM_MAX * TEAM_M + (size_t)nt * ns * TEAM_SLOTS * TEAM_H) * sizeof(u64), s)) return rc;
    constexpr bool timing = DC_DEV_TIMING != 0;
    if (timing && cell == CELL_LSTM && (ns == 1 || ns == 4)) {   // debugging aid: phase cycles of one wave, printed per launch
        static long long* dbg = nullptr;
        if (!dbg) (void)hipMalloc(&dbg, 64);
        a.dbg = dbg;
        if (ns == 1) hipLaunchKernelGGL((rnn_team_fwd_kernel<CELL_LSTM, 1, true>), dim3(nt * TEAM_M), dim3(512), 0, s, a, xb, nt, team_plain(a.flags));
        else hipLaunchKernelGGL((rnn_team_fwd_kernel<CELL_LSTM, 4, true>), dim3(nt * TEAM_M), dim3(512), 0, s, a, xb, nt, team_plain(a.flags));
        long long h[6];
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        const double calls = (double)((a.n_seq + nt - 1) / nt) * max_len;   // step calls of one team (uniform lengths)
        fprintf(stderr, "rnn_team_fwd timing (NS=%d, cycles per step call, %.0f calls): between %.0f  first read %.0f  spin %.0f  barrier %.0f  "
                        "arithmetic+stores %.0f  stale first reads %.2f\n", ns, calls, h[0] / calls, h[1] / calls, h[2] / calls, h[3] / calls,
                h[4] / calls, h[5] / calls);
        return launch_check("rnn_team_forward");
    }
    if (cell == CELL_GRU) launch_team_fwd<CELL_GRU>(ns, nt, a, xb, s);
    else launch_team_fwd<CELL_LSTM>(ns, nt, a, xb, s);
    return launch_check("rnn_team_forward");
}

int rnn_team_backward(int cell, RnnStepArgs a, int max_len, hipStream_t s) {
    if (rnn_team8_supported(cell, a.H, a.n_seq, a.flags)) return rnn_team8_backward(cell, a, max_len, s);
    if (lstm_team_mfma_supported(cell, a.H, a.n_seq, a.flags, true)) return lstm_team_mfma_backward(cell, a, max_len, team_capacity(), s);
    u64* xb = static_cast<u64*>(a.xbuf);
    if (!xb) { set_error("rnn_team_backward: no exchange buffer (RnnStepArgs::xbuf)", 1012); return 1012; }
    const int nt = team_count(a.n_seq), ns = team_streams(a.n_seq, nt, a.flags);
    const double G = cell == CELL_GRU ? 3 : 4;
    ProfScope prof(cell == CELL_GRU ? "gru_bwd_team" : "lstm_bwd_team", 2.0 * a.n_seq * G * a.H * a.H * max_len,
                   4.0 * a.n_seq * max_len * a.H * (3.0 * G + 6.0), s);
    if (int rc = zero_async(xb, ((size_t)TEAM_HDR + (size_t)TEAM_MAX * TEAM_M + (size_t)nt * ns * TEAM_SLOTS * 4 * TEAM_H) * sizeof(u64), s)) return rc;
    if (cell == CELL_GRU) launch_team_bwd<CELL_GRU>(ns, nt, a, xb, s);
    else launch_team_bwd<CELL_LSTM>(ns, nt, a, xb, s);
    return launch_check("rnn_team_backward");
}

}  // namespace dc
