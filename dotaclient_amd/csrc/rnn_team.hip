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
// overwrites (tag - 4) has been consumed by everyone.  Every spin is bounded; a member that gives up
// poisons its outputs with NaN (the loss turns NaN and the optimizer raises, optimizer.py:667) instead of
// hanging the GPU.  All 4 x teams workgroups must be resident at once: the host sizes the grid from the
// CU count (one 512-thread workgroup with ~180 registers per lane per CU).
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

namespace dc {
namespace {

enum { TEAM_H = 256, TEAM_US = 64, TEAM_M = 4, TEAM_SLOTS = 4, TEAM_MAX = 64 };
enum { CELL_GRU = 0, CELL_LSTM = 1 };
constexpr int PF = 4;
constexpr int SPIN_LIMIT = 1 << 21;   // polls (~0.5-1 us each) before a member gives up

typedef unsigned long long u64;

__device__ __forceinline__ u64 granule_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void granule_store(u64* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// spins until the granule carries `tag`; false on timeout
__device__ __forceinline__ bool granule_wait(const u64* p, unsigned tag, float& v) {
    u64 g = granule_load(p);
    int n = 0;
    while ((unsigned)(g >> 32) != tag) {
        if (++n > SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(1);
        g = granule_load(p);
    }
    v = __uint_as_float((unsigned)g);
    return true;
}

// blockIdx -> (team, member): members of a team 8 blocks apart (same XCD under the usual round-robin
// placement - a speed hint only)
__device__ __forceinline__ void team_of_block(int n_teams, int& team, int& member) {
    const int b = blockIdx.x;
    if ((n_teams & 7) == 0) {
        const int x = b & 7, j = b >> 3;
        team = (j >> 2) * 8 + x;
        member = j & 3;
    } else {
        team = b >> 2;
        member = b & 3;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// forward.  Same contract as the per-step kernels of rnn.hip: gates[row][G*H] = W_ih x + b_ih on entry,
// activated gates on exit; hprev/cprev[first row] = h0/c0; h_t (c_t) -> hseq (cseq)[row] and
// hprev (cprev)[row+1]; GRU: hn[row] = W_hn h + b_hn.
// ---------------------------------------------------------------------------------------------------
template <int CELL>
__global__ __launch_bounds__(512) void rnn_team_fwd_kernel(RnnStepArgs p, u64* __restrict__ xbuf, int n_teams) {
    constexpr int H = TEAM_H, G = CELL == CELL_GRU ? 3 : 4, GH = G * H;
    constexpr int KPL = 16, NRD = 4;
    __shared__ __attribute__((aligned(16))) float h_lds[2][H];
    __shared__ int dead;
    const int tid = threadIdx.x;
    const int kg = tid & 15, row = tid >> 4;
    const int q = tid & 3, dup = (tid >> 3) & 1;
    int team, member;
    team_of_block(n_teams, team, member);
    const int U0 = member * TEAM_US;
    const int u = U0 + 2 * row + ((tid >> 2) & 1);
    u64* const xb = xbuf + (size_t)team * (TEAM_SLOTS * H);

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

    unsigned tag = 0;   // tag of the previous step's output (team-wide running counter)
    for (int b = team; b < p.n_seq; b += n_teams) {
        const int len = p.seq_len[b];
        if (len <= 0) continue;
        const size_t row0 = (size_t)p.seq_off[b];
        float* const gp = p.gates + row0 * GH + gq * H + u;
        float st = CELL == CELL_LSTM ? p.cprev[row0 * H + u] : p.hprev[row0 * H + u];   // c_{t-1} (LSTM) / h_{t-1} (GRU) of this unit
        __syncthreads();                                           // the previous sequence's last reads of h_lds
        if (tid < H) h_lds[0][tid] = p.hprev[row0 * H + tid];
        float xc[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) xc[j] = gp[(size_t)min(j, len - 1) * GH];
        asm volatile("" : "+v"(xc[0]), "+v"(xc[1]), "+v"(xc[2]), "+v"(xc[3]) : : "memory");

        auto step = [&](const int t, const float x) -> bool {
            const int par = t & 1;
            if (t > 0 && tid < H - TEAM_US) {                       // the other members' h_{t-1}
                const int idx = (U0 + TEAM_US + tid) & (H - 1);
                float v = 0.f;
                if (!granule_wait(xb + (tag & 3) * H + idx, tag, v)) dead = 1;
                h_lds[par][idx] = v;
            }
            __syncthreads();
            const float* hl = &h_lds[par][4 * kg];
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
            const float ah = a[0] + bq;                            // W_hh h + b_hh of this lane's gate
            const size_t r = (size_t)t;
            float hn;
            if constexpr (CELL == CELL_LSTM) {
                const float act = __builtin_fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(sc * (ah + x))), am, aa);
                const float ig = dpp<DPP_Q0>(act), fg = dpp<DPP_Q1>(act), gg = dpp<DPP_Q2>(act), og = dpp<DPP_Q3>(act);
                const float cn = fg * st + ig * gg;
                hn = og * tanh_hw(cn);
                st = cn;
                if (dup == 0) gp[r * GH] = act;
                else {
                    // q 0,1: h_t, c_t -> hseq/cseq[row]; q 2,3: -> hprev/cprev[row+1] (last step: the same value again)
                    float* base = (q & 1) ? ((q >= 2 && t + 1 < len) ? p.cprev + H : p.cseq) : ((q >= 2 && t + 1 < len) ? p.hprev + H : p.hseq);
                    base[(row0 + r) * H + u] = (q & 1) ? cn : hn;
                }
            } else {
                const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(sc * (ah + x)));   // r, z lanes
                const float rg = dpp<DPP_Q0>(s);
                const float ng = tanh_hw(x + rg * ah);                                                  // n lane
                const float act = is_t ? ng : s;
                const float zg = dpp<DPP_Q1>(act), nn = dpp<DPP_Q2>(act), hnv = dpp<DPP_Q2>(ah);
                hn = (1.f - zg) * nn + zg * st;
                st = hn;
                if (dup == 0) {
                    if (q < 3) gp[r * GH] = act;
                    else p.hn[(row0 + r) * H + u] = hnv;
                } else if (q == 0) {
                    p.hseq[(row0 + r) * H + u] = hn;
                } else if (q == 1 && t + 1 < len) {
                    p.hprev[(row0 + r + 1) * H + u] = hn;
                }
            }
            ++tag;
            if (dup == 0 && q == 0) {
                granule_store(xb + (tag & 3) * H + u, hn, tag);
                h_lds[par ^ 1][u] = hn;
            }
            return is_dead == 0;
        };
        auto poison = [&](int t) {   // a peer never answered: make the failure visible downstream
            if (tid < TEAM_US) p.hseq[(row0 + t) * H + U0 + tid] = __builtin_nanf("");
        };
        int t = 0;
        bool ok = true;
        for (; ok && t + PF <= len; t += PF) {
            float xn[PF];
#pragma unroll
            for (int j = 0; j < PF; ++j) xn[j] = gp[(size_t)min(t + PF + j, len - 1) * GH];
            ok = step(t, xc[0]) && step(t + 1, xc[1]) && step(t + 2, xc[2]) && step(t + 3, xc[3]);
#pragma unroll
            for (int j = 0; j < PF; ++j) xc[j] = xn[j];
            asm volatile("" : "+v"(xc[0]), "+v"(xc[1]), "+v"(xc[2]), "+v"(xc[3]) : : "memory");
        }
        if (ok && t < len) {
            ok = step(t, xc[0]);
            if (ok && t + 1 < len) {
                ok = step(t + 1, xc[1]);
                if (ok && t + 2 < len) ok = step(t + 2, xc[2]);
            }
        }
        if (!ok) { poison(min(t, len - 1)); return; }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward through time.  In: dh[row][H] (from above), the forward's activated gates, cseq/cprev (LSTM)
// or hn/hprev (GRU).  Out: dgx[row][G*H] and, GRU, dgh[row][G*H] (the n gate's differs by the factor r).
// ---------------------------------------------------------------------------------------------------
template <int CELL>
__global__ __launch_bounds__(512) void rnn_team_bwd_kernel(RnnStepArgs p, u64* __restrict__ xbuf, int n_teams) {
    constexpr int H = TEAM_H, G = CELL == CELL_GRU ? 3 : 4, GH = G * H, NG = 4 * H;   // NG: granules / LDS positions per step
    constexpr int KPL = 16, NRD = 4;
    __shared__ __attribute__((aligned(16))) float g_lds[2][NG];   // position 4*unit + gate slot
    __shared__ int dead;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 3, dup = (lane >> 2) & 1, b3 = (lane >> 3) & 1;
    int team, member;
    team_of_block(n_teams, team, member);
    const int U0 = member * TEAM_US;
    const int u = U0 + 8 * wave + (lane >> 3);
    u64* const xb = xbuf + (size_t)team * (TEAM_SLOTS * NG);

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
    if (tid == 0) dead = 0;

    unsigned tag = 0;
    for (int b = team; b < p.n_seq; b += n_teams) {
        const int len = p.seq_len[b];
        if (len <= 0) continue;
        const size_t row0 = (size_t)p.seq_off[b];
        const float* const gp = p.gates + row0 * GH + gq * H + u;
        // LSTM: q0 c_t, q1 c_{t-1}, q2 dh (q3: dh again, unused).  GRU: q0 hn, q1 h_{t-1}, q2 dh.
        const float* const shp = (CELL == CELL_LSTM ? (q == 0 ? p.cseq : (q == 1 ? p.cprev : p.dh))
                                                    : (q == 0 ? p.hn : (q == 1 ? p.hprev : p.dh))) + row0 * H + u;
        float aoc[PF], shc[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const size_t r = (size_t)max(len - 1 - j, 0);
            aoc[j] = gp[r * GH];
            shc[j] = shp[r * H];
        }
        asm volatile("" : "+v"(aoc[0]), "+v"(aoc[1]), "+v"(aoc[2]), "+v"(aoc[3]), "+v"(shc[0]), "+v"(shc[1]), "+v"(shc[2]),
                     "+v"(shc[3]) : : "memory");
        float carry = 0.f, gate_next = 0.f;   // LSTM: dc_{t+1}, f_{t+1};  GRU: dh_{t+1}, z_{t+1}
        __syncthreads();
        g_lds[0][tid] = 0.f; g_lds[0][512 + tid] = 0.f;            // "step len" has no gate gradient
        // (the barrier of the first step orders these writes)

        auto step = [&](const int i, const float a_own, const float shv) -> bool {
            const int t = len - 1 - i, cur = i & 1;
            if (i > 0) {                                           // the other members' gate gradients of step t+1
                const int i0 = (4 * U0 + 4 * TEAM_US + tid) & (NG - 1);
                float v0 = 0.f, v1 = 0.f;
                const u64* base = xb + (tag & 3) * NG;
                bool ok = granule_wait(base + i0, tag, v0);
                g_lds[cur][i0] = v0;
                if (tid < 256) {
                    const int i1 = (4 * U0 + 4 * TEAM_US + 512 + tid) & (NG - 1);
                    ok = granule_wait(base + i1, tag, v1) && ok;
                    g_lds[cur][i1] = v1;
                }
                if (!ok) dead = 1;
            }
            __syncthreads();
            const float* gl = &g_lds[cur][4 * lane];
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
            float d, dh_out;
            if constexpr (CELL == CELL_LSTM) {
                const float ig = dpp<DPP_Q0>(a_own), fg = dpp<DPP_Q1>(a_own), gg = dpp<DPP_Q2>(a_own), og = dpp<DPP_Q3>(a_own);
                const float cs = dpp<DPP_Q0>(shv), cp = dpp<DPP_Q1>(shv), dhx = dpp<DPP_Q2>(shv);
                const float dh = dhx + rec;
                const float tc = tanh_hw(cs);
                const float dcv = dh * og * (1.f - tc * tc) + carry * gate_next;
                const float M = is_q3 ? dh * tc : dcv;
                float X = is_q2 ? ig : 1.f;
                X = is_q1 ? cp : X;
                X = is_q0 ? gg : X;
                const float D = __builtin_fmaf(-a_own, a_own, is_q2 ? 1.f : a_own);
                d = M * X * D;
                dh_out = d;
                carry = dcv;
                gate_next = fg;
                if (dup == 0) p.dgx[(row0 + t) * GH + q * H + u] = d;
            } else {
                const float rg = dpp<DPP_Q0>(a_own), zg = dpp<DPP_Q1>(a_own), ng = dpp<DPP_Q2>(a_own);
                const float hnv = dpp<DPP_Q0>(shv), hpv = dpp<DPP_Q1>(shv), dhx = dpp<DPP_Q2>(shv);
                const float dh = dhx + rec + carry * gate_next;
                const float dn = dh * (1.f - zg) * (1.f - ng * ng);
                const float dz = dh * (hpv - ng) * zg * (1.f - zg);
                const float dr = dn * hnv * rg * (1.f - rg);
                d = is_q0 ? dr : (is_q1 ? dz : (is_q2 ? dn : 0.f));
                dh_out = is_q2 ? dn * rg : d;
                carry = dh;
                gate_next = zg;
                if (q < 3) {
                    if (dup == 0) p.dgx[(row0 + t) * GH + q * H + u] = d;
                    else p.dgh[(row0 + t) * GH + q * H + u] = dh_out;
                }
            }
            ++tag;
            if (dup == 0) {
                granule_store(xb + (tag & 3) * NG + 4 * u + q, dh_out, tag);
                g_lds[cur ^ 1][4 * u + q] = dh_out;
            }
            return is_dead == 0;
        };
        int i = 0;   // step index from the end: t = len - 1 - i
        bool ok = true;
        for (; ok && i + PF <= len; i += PF) {
            float aon[PF], shn[PF];
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const size_t r = (size_t)max(len - 1 - i - PF - j, 0);
                aon[j] = gp[r * GH];
                shn[j] = shp[r * H];
            }
            ok = step(i, aoc[0], shc[0]) && step(i + 1, aoc[1], shc[1]) && step(i + 2, aoc[2], shc[2]) && step(i + 3, aoc[3], shc[3]);
#pragma unroll
            for (int j = 0; j < PF; ++j) { aoc[j] = aon[j]; shc[j] = shn[j]; }
            asm volatile("" : "+v"(aoc[0]), "+v"(aoc[1]), "+v"(aoc[2]), "+v"(aoc[3]), "+v"(shc[0]), "+v"(shc[1]), "+v"(shc[2]),
                         "+v"(shc[3]) : : "memory");
        }
        if (ok && i < len) {
            ok = step(i, aoc[0], shc[0]);
            if (ok && i + 1 < len) {
                ok = step(i + 1, aoc[1], shc[1]);
                if (ok && i + 2 < len) ok = step(i + 2, aoc[2], shc[2]);
            }
        }
        if (!ok) {
            if (tid < TEAM_US) p.dgx[(row0 + max(len - 1 - i, 0)) * GH + U0 + tid] = __builtin_nanf("");
            return;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
namespace {

// exchange ring of the largest case (backward: 4H granules per slot), allocated once
u64* team_xbuf() {
    static u64* buf = nullptr;
    if (!buf && hipMalloc(&buf, (size_t)TEAM_MAX * TEAM_SLOTS * 4 * TEAM_H * sizeof(u64)) != hipSuccess) buf = nullptr;
    return buf;
}

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

}  // namespace

// DC_RNN_TEAM=0 forces the launch-per-step kernels (A/B measurements, parity tests of both)
bool rnn_team_supported(int cell, int H, int n_seq) {
    const char* e = getenv("DC_RNN_TEAM");
    const bool on = !(e && e[0] == '0');
    // above ~6 sequences per team the serial walk through a team's sequences loses to the batched per-step launches
    return on && H == TEAM_H && (cell == CELL_GRU || cell == CELL_LSTM) && n_seq <= 6 * TEAM_MAX && team_capacity() >= 1;
}

int rnn_team_forward(int cell, RnnStepArgs a, int max_len, hipStream_t s) {
    u64* xb = team_xbuf();
    if (!xb) { set_error("rnn_team_forward: exchange buffer allocation failed", 1012); return 1012; }
    const int nt = team_count(a.n_seq);
    const double G = cell == CELL_GRU ? 3 : 4;
    ProfScope prof(cell == CELL_GRU ? "gru_fwd_team" : "lstm_fwd_team", 2.0 * a.n_seq * G * a.H * a.H * max_len,
                   4.0 * a.n_seq * max_len * a.H * (2.0 * G + 4.0), s);
    if (hipMemsetAsync(xb, 0, (size_t)nt * TEAM_SLOTS * TEAM_H * sizeof(u64), s) != hipSuccess) return launch_check("rnn_team_forward memset");
    if (cell == CELL_GRU) hipLaunchKernelGGL((rnn_team_fwd_kernel<CELL_GRU>), dim3(nt * TEAM_M), dim3(512), 0, s, a, xb, nt);
    else hipLaunchKernelGGL((rnn_team_fwd_kernel<CELL_LSTM>), dim3(nt * TEAM_M), dim3(512), 0, s, a, xb, nt);
    return launch_check("rnn_team_forward");
}

int rnn_team_backward(int cell, RnnStepArgs a, int max_len, hipStream_t s) {
    u64* xb = team_xbuf();
    if (!xb) { set_error("rnn_team_backward: exchange buffer allocation failed", 1012); return 1012; }
    const int nt = team_count(a.n_seq);
    const double G = cell == CELL_GRU ? 3 : 4;
    ProfScope prof(cell == CELL_GRU ? "gru_bwd_team" : "lstm_bwd_team", 2.0 * a.n_seq * G * a.H * a.H * max_len,
                   4.0 * a.n_seq * max_len * a.H * (3.0 * G + 6.0), s);
    if (hipMemsetAsync(xb, 0, (size_t)nt * TEAM_SLOTS * 4 * TEAM_H * sizeof(u64), s) != hipSuccess) return launch_check("rnn_team_backward memset");
    if (cell == CELL_GRU) hipLaunchKernelGGL((rnn_team_bwd_kernel<CELL_GRU>), dim3(nt * TEAM_M), dim3(512), 0, s, a, xb, nt);
    else hipLaunchKernelGGL((rnn_team_bwd_kernel<CELL_LSTM>), dim3(nt * TEAM_M), dim3(512), 0, s, a, xb, nt);
    return launch_check("rnn_team_backward");
}

}  // namespace dc
