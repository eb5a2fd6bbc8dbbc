// Persistent recurrent kernels for H = 512 on the bf16 path (DC_DIMS_BF16, BASELINE.json configs[4]: "5v5 hidden=512 2-layer LSTM,
// bf16 MFMA path"): the whole time loop of a layer in ONE launch.  Replaces nn.LSTM's recurrence at /root/reference/policy.py:66,141
// (cell parametrised as BASELINE.json asks) and its BPTT (/root/reference/optimizer.py:672); same arithmetic as the launch-per-step
// kernels of rnn_step_bf16.hip (bf16 operands, f32 accumulate, f32 state and gate maths), which took 13 / 23 us per time step -
// bound by the launch boundary and by gathering one row of every sequence per launch.
//
// W_hh of a 512-wide LSTM is 2 MB as bf16: no CU holds it.  A TEAM of 16 workgroups (16 CUs) does: member m owns the hidden units
// 32 m .. 32 m + 31, i.e. 128 gate columns x 512 = 128 KB of bf16 weights, resident in the registers of its four waves (128 VGPRs
// per lane) for the whole launch.  A team advances a TILE of 32 sequences together on v_mfma_f32_32x32x16_bf16 (the sequences are
// the rows of every product) and works through its tiles one after the other.
//
//   forward  (column-parallel): wave g of member m multiplies h_{t-1}[32 seq][512] (bf16, LDS) with its gate-g rows of W_hh:
//            32 MFMAs; the four gate blocks meet in LDS, 256 threads finish four (sequence, unit) cells each; the member's
//            h_t[32 seq][32 units] goes to the 15 peers as 512 tagged granules {bf16, bf16, tag} (team_util.h) - an all-gather of
//            7 680 granule reads per member and step.
//   backward (row-parallel): a member contracts ITS OWN 128 gate gradients of step t + 1 (bf16, LDS - no input from anyone)
//            with W_hh[own column][u'] for all 512 output units: 32 MFMAs per wave; the partial sums dh_rec[32 seq][32 units] for
//            each owner go out as 512 granules of two bf16 each (a reduce-scatter: 7 680 out, 7 680 in) and the owner adds the
//            fifteen it receives to its own.  Rounding the PARTIAL sums to bf16 is the one place this kernel is coarser than the
//            per-step form (whose 2 048-term sums stay f32): sixteen roundings of 2^-9 each on sums of 128 terms - the same order
//            as the bf16 rounding of the 2 048 operands themselves; tolerances in tests/test_gpu_bf16.py are unchanged.
//
// Roles by ticket, XCD-local teams when the team count is a multiple of 8, L2-scope granule stores when a team shares an XCD,
// timeouts reported in DC_WS_FAULT: as in rnn_team.hip / team_util.h.
#include "kernels.h"
#include "gemm_tiles.h"
#include "team_util.h"

namespace dc {
namespace {

enum {
    T5_H = 512, T5_M = 16, T5_US = 32, T5_NS = 32, T5_THREADS = 256, T5_SLOTS = 4, T5_MAXTEAMS = 16,
    T5_PAIRS = T5_NS * T5_US / 2,            // granules a member publishes per step (forward) / per owner (backward)
    T5_HROW = T5_H * 2 + 16,                 // bytes: LDS row of the h tile [32 seq][512 k] (16 bytes of padding: conflict-free b128 reads)
    T5_GROW = 4 * T5_US * 2 + 16,            // bytes: LDS row of the own gate-gradient tile [32 seq][128 k]
    T5_RED_LD = 33,
    T5_HDR = 16,                             // u64 words: ticket counters
    T5_HS = T5_MAXTEAMS * T5_M,              // handshake granules
    T5_FWD_RING = T5_SLOTS * T5_M * T5_PAIRS,            // words per team
    T5_BWD_RING = T5_SLOTS * T5_M * T5_M * T5_PAIRS,     // words per team: [slot][owner][source][512]
    T5_FWD_LDS = 2 * T5_NS * T5_HROW + 4 * T5_NS * T5_RED_LD * 4,
    T5_BWD_LDS = 2 * T5_NS * T5_GROW + T5_NS * T5_RED_LD * 4,
    T5_K_FWD = 5, T5_K_BWD = 6,              // kernel ids in the fault record (include/dotaclient_hip.h)
};

__device__ __forceinline__ u32x4 t5_to_bf16x8(const float4& a, const float4& b) {
    return u32x4{cvt_pk_bf16(a.x, a.y), cvt_pk_bf16(a.z, a.w), cvt_pk_bf16(b.x, b.y), cvt_pk_bf16(b.z, b.w)};
}
__device__ __forceinline__ float t5_bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float t5_bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// team_util.h's team_claim_role for teams of sixteen
__device__ __forceinline__ void t5_claim_role(unsigned* claim, int n_teams, int& team, int& member) {
    __shared__ int role_sh[2];
    if (threadIdx.x == 0) {
        int t = -1, m = 0;
        if ((n_teams & 7) == 0) {
            const int quota = (n_teams >> 3) * T5_M;
            const int x = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7);   // HW_REG_XCC_ID
            for (int k = 0; k < 8 && t < 0; ++k) {
                const int y = (x + k) & 7;
                const unsigned o = atomicAdd(&claim[y], 1u);
                if ((int)o < quota) { t = (int)(o / T5_M) * 8 + y; m = (int)(o % T5_M); }
            }
        } else {
            const unsigned o = atomicAdd(&claim[0], 1u);
            if ((int)o < n_teams * T5_M) { t = (int)(o / T5_M); m = (int)(o % T5_M); }
        }
        role_sh[0] = t; role_sh[1] = m;
    }
    __syncthreads();
    team = role_sh[0]; member = role_sh[1];
}

__device__ __forceinline__ int t5_same_xcd(u64* hs, int member, int allow) {
    __shared__ int same_sh;
    if (threadIdx.x == 0) {
        const unsigned my = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        granule_store(hs + member, __uint_as_float(my), TEAM_HS_TAG);
        bool same = allow != 0;
        for (int m = 0; m < T5_M; ++m) {
            if (m == member) continue;
            float v = 0.f;
            const bool ok = granule_wait(granule_load(hs + m), hs + m, TEAM_HS_TAG, v);
            same = same && ok && __float_as_uint(v) == my;
        }
        same_sh = same ? 1 : 0;
    }
    __syncthreads();
    return same_sh;
}

// All N granules of a thread in ONE batch, the whole batch re-issued until every tag matches: a retry costs one L2 round trip
// (~0.8 us) for all of them.  (Waiting for them one after the other, as the four-member kernels do with their three granules, made
// every stale first read pay its own round trip: 30 granules per thread, 5.6 us per step.)  false on timeout.
enum { T5_SPIN = 1 << 20 };
template <int N, class Addr>
__device__ __forceinline__ bool t5_poll_all(Addr addr, unsigned tag, u64 (&g)[N]) {
    for (int spins = 0;; ++spins) {
#pragma unroll
        for (int n = 0; n < N; ++n) g[n] = granule_load(addr(n));
        bool ok = true;
#pragma unroll
        for (int n = 0; n < N; ++n) ok = ok && (unsigned)(g[n] >> 32) == tag;
        if (ok) return true;
        if (spins > T5_SPIN) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// longest sequence of the tile that starts at sequence b0 (all 256 threads call; result uniform)
__device__ __forceinline__ int t5_tile_tmax(const RnnStepArgs& p, int b0) {
    __shared__ int tmax_sh;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int b = b0 + (int)threadIdx.x;
        int v = (threadIdx.x < T5_NS && b < p.n_seq) ? p.seq_len[b] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
        if (threadIdx.x == 0) tmax_sh = v;
    }
    __syncthreads();
    return tmax_sh;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward.  In: gates = W_ih x + b_ih (overwritten by the activated gates), hprev / cprev rows 0 of every sequence = h0 / c0 (seeded
// by rnn_forward_layer); out: gates, cseq, hseq, hprev / cprev (the rows the backward and the weight gradients read).
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(T5_THREADS, 1) void lstm512_team_fwd_kernel(RnnStepArgs p, const uint16_t* __restrict__ Wb,
                                                                          u64* __restrict__ xbuf_all, int n_teams, int allow_plain) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ht0 = smem;                                            // h tiles, double-buffered by the parity of t
    float* const red = reinterpret_cast<float*>(smem + 2 * T5_NS * T5_HROW);   // [gate][seq][unit] pre-activations W_hh h
    __shared__ int dead;
    constexpr int H = T5_H, GH = 4 * T5_H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // = the gate this wave computes
    const int fr = lane & 31, fq = lane >> 5;
    int team, member;
    t5_claim_role(reinterpret_cast<unsigned*>(xbuf_all), n_teams, team, member);
    if (team < 0) return;
    const int plain = t5_same_xcd(xbuf_all + T5_HDR + team * T5_M, member, allow_plain);
    u64* const ring = xbuf_all + T5_HDR + T5_HS + (size_t)team * T5_FWD_RING;
    const int U0 = T5_US * member;
    if (tid == 0) dead = 0;

    // ---- weights: rows (gate = wave, unit U0 + fr) of W_hh, all of K, as MFMA operands -------------------------------------------------
    bf16x8 wreg[32];
    {
        const uint16_t* wrow = Wb + (size_t)(wave * H + U0 + fr) * H + fq * 8;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) wreg[ks] = *reinterpret_cast<const bf16x8*>(wrow + ks * 16);
    }
    // ---- this thread's cells: sequences (tid >> 4) and 16 + (tid >> 4) of the tile, units 2 (tid & 15), + 1 ----------------------------
    const int up = tid & 15, j0 = U0 + 2 * up;
    float bh[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) { bh[g][0] = p.bhh[g * H + j0]; bh[g][1] = p.bhh[g * H + j0 + 1]; }

    unsigned tag = 0;
    bool failed = false;
    const int n_tiles = (p.n_seq + T5_NS - 1) / T5_NS;
    for (int tile = team; tile < n_tiles && !failed; tile += n_teams) {
        const int b0 = tile * T5_NS;
        const int tmax = t5_tile_tmax(p, b0);
        int len[2];
        size_t row0[2];
        float c[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int b = b0 + (tid >> 4) + 16 * i;
            len[i] = b < p.n_seq ? p.seq_len[b] : 0;
            row0[i] = len[i] > 0 ? (size_t)p.seq_off[b] : 0;
            c[i][0] = len[i] > 0 ? p.cprev[row0[i] * H + j0] : 0.f;
            c[i][1] = len[i] > 0 ? p.cprev[row0[i] * H + j0 + 1] : 0.f;
        }
        // h0 of all 512 units of the tile's sequences -> tile buffer 0
        for (int e = tid; e < T5_NS * 64; e += T5_THREADS) {
            const int s = e >> 6, k8 = (e & 63) * 8;
            const int b = b0 + s;
            u32x4 v = u32x4{0, 0, 0, 0};
            if (b < p.n_seq && p.seq_len[b] > 0) {
                const float* src = p.hprev + (size_t)p.seq_off[b] * H + k8;
                v = t5_to_bf16x8(*reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4));
            }
            *reinterpret_cast<u32x4*>(ht0 + s * T5_HROW + k8 * 2) = v;
        }
        // the input projections W_ih x + b_ih of a step are fetched ONE STEP AHEAD: vector memory operations complete in order, so a
        // load from HBM (2 us) issued right before the granule loads would sit in front of every one of them
        float2 gxn[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* gt = p.gates + row0[i] * GH + j0;
#pragma unroll
            for (int g = 0; g < 4; ++g) gxn[i][g] = len[i] > 0 ? *reinterpret_cast<const float2*>(gt + g * H) : make_float2(0.f, 0.f);
        }
        __syncthreads();

#pragma unroll 1
        for (int t = 0; t < tmax; ++t) {
            char* const hcur = ht0 + (t & 1) * (T5_NS * T5_HROW);          // holds h_{t-1}
            char* const hnxt = ht0 + ((t + 1) & 1) * (T5_NS * T5_HROW);    // receives h_t
            ++tag;
            // (a) this step's input projections (fetched during the previous step)
            bool on[2];
            float2 gx[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                on[i] = t < len[i];
#pragma unroll
                for (int g = 0; g < 4; ++g) gx[i][g] = gxn[i][g];
            }
            // (b) the fifteen peers' h_{t-1}: 30 granules per thread (granule nn: peer nn >> 1, pair 256 (nn & 1) + tid)
            if (t > 0) {
                const unsigned rt = tag - 1;
                const u64* const slot = ring + (size_t)(rt & (T5_SLOTS - 1)) * (T5_M * T5_PAIRS);
                u64 g[30];
                const bool ok = t5_poll_all<30>([&](int nn) {
                    const int pi = nn >> 1, mp = pi + (pi >= member ? 1 : 0);
                    return slot + (size_t)mp * T5_PAIRS + (nn & 1) * 256 + tid;
                }, rt, g);
                if (!ok) {
                    dead = 1;
                    team_report_timeout(p.fault, T5_K_FWD, p.layer, team, member, t, b0 + (tid >> 4), rt);
                }
#pragma unroll
                for (int nn = 0; nn < 30; ++nn) {
                    const int pi = nn >> 1, mp = pi + (pi >= member ? 1 : 0);
                    const int pair = (nn & 1) * 256 + tid;
                    *reinterpret_cast<unsigned*>(hcur + (pair >> 4) * T5_HROW + (T5_US * mp + 2 * (pair & 15)) * 2) = (unsigned)g[nn];
                }
            }
            // the next step's input projections: behind the granule loads in the queue, a whole step ahead of their use
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool onn = t + 1 < len[i];
                const float* gt = p.gates + (row0[i] + (size_t)(t + 1)) * GH + j0;
#pragma unroll
                for (int g = 0; g < 4; ++g) gxn[i][g] = onn ? *reinterpret_cast<const float2*>(gt + g * H) : make_float2(0.f, 0.f);
            }
            __syncthreads();
            if (dead) { failed = true; break; }
            // (c) [32 seq] x [32 gate columns] += h_{t-1} W_hh^T over K = 512: two accumulator chains
            f32x16 acc0, acc1;
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
            const char* arow = hcur + fr * T5_HROW + fq * 16;
#pragma unroll
            for (int ks = 0; ks < 32; ks += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(arow + ks * 32), wreg[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(arow + ks * 32 + 32), wreg[ks + 1], acc1, 0, 0, 0);
            }
            // (d) C layout: column (= unit) lane & 31, row (= sequence) (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[(wave * T5_NS + (r & 3) + 8 * (r >> 2) + 4 * fq) * T5_RED_LD + fr] = acc0[r] + acc1[r];
            __syncthreads();
            // (e) the cells
            const bool publish = t + 1 < tmax;
            u64* const out_slot = ring + (size_t)(tag & (T5_SLOTS - 1)) * (T5_M * T5_PAIRS) + (size_t)member * T5_PAIRS;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int s = (tid >> 4) + 16 * i;
                unsigned packed = 0;
                if (on[i]) {
                    float act[4][2], hv[2], cn[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        float pre[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            pre[g] = (e == 0 ? gx[i][g].x : gx[i][g].y) + red[(g * T5_NS + s) * T5_RED_LD + 2 * up + e] + bh[g][e];
                        const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
                        cn[e] = fg * c[i][e] + ig * gg;
                        hv[e] = og * tanhf(cn[e]);
                        c[i][e] = cn[e];
                        act[0][e] = ig; act[1][e] = fg; act[2][e] = gg; act[3][e] = og;
                    }
                    const size_t r = row0[i] + (size_t)t;
                    float* gt = p.gates + r * GH + j0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<float2*>(gt + g * H) = make_float2(act[g][0], act[g][1]);
                    *reinterpret_cast<float2*>(p.cseq + r * H + j0) = make_float2(cn[0], cn[1]);
                    *reinterpret_cast<float2*>(p.hseq + r * H + j0) = make_float2(hv[0], hv[1]);
                    if (t + 1 < len[i]) {
                        *reinterpret_cast<float2*>(p.cprev + (r + 1) * H + j0) = make_float2(cn[0], cn[1]);
                        *reinterpret_cast<float2*>(p.hprev + (r + 1) * H + j0) = make_float2(hv[0], hv[1]);
                    }
                    packed = cvt_pk_bf16(hv[0], hv[1]);
                }
                // own units of h_t for the next product, and the same two values to the peers (finished sequences publish zeros:
                // every peer waits for all 512 granules of every member)
                *reinterpret_cast<unsigned*>(hnxt + s * T5_HROW + j0 * 2) = packed;
                if (publish) granule_store(out_slot + i * 256 + tid, __uint_as_float(packed), tag, plain);
            }
        }
        __syncthreads();
    }
    if (failed && tid < T5_US) {        // a peer never answered: make the failure visible downstream (NaN loss -> status 1)
        const int b = min(team * T5_NS, p.n_seq - 1);
        p.hseq[(size_t)p.seq_off[b] * H + U0 + tid] = __builtin_nanf("");
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward through time.  In: dh (from the layer above / the heads), the forward's gates / cseq / cprev; out: dgx (gradient w.r.t.
// W_ih x + b_ih = w.r.t. W_hh h + b_hh), dh (total), dc.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(T5_THREADS, 1) void lstm512_team_bwd_kernel(RnnStepArgs p, const uint16_t* __restrict__ WTb,
                                                                          u64* __restrict__ xbuf_all, int n_teams, int allow_plain) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const dg0 = smem;                                             // own gate gradients [seq][gate * 32 + unit] bf16, double-buffered
    float* const own = reinterpret_cast<float*>(smem + 2 * T5_NS * T5_GROW);   // own partial sums [seq][unit]
    __shared__ int dead;
    constexpr int H = T5_H, GH = 4 * T5_H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fq = lane >> 5;
    int team, member;
    t5_claim_role(reinterpret_cast<unsigned*>(xbuf_all), n_teams, team, member);
    if (team < 0) return;
    const int plain = t5_same_xcd(xbuf_all + T5_HDR + team * T5_M, member, allow_plain);
    u64* const ring = xbuf_all + T5_HDR + T5_HS + (size_t)team * T5_BWD_RING;
    const int U0 = T5_US * member;
    if (tid == 0) dead = 0;

    // ---- weights: for the four owners o = 4 wave + i: rows W_hh^T[unit 32 o + fr][own columns], K = 128 own gate columns ----------------
    bf16x8 wreg[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint16_t* wrow = WTb + (size_t)(T5_US * (4 * wave + i) + fr) * GH + U0 + fq * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wreg[i][ks] = *reinterpret_cast<const bf16x8*>(wrow + (ks >> 1) * H + (ks & 1) * 16);
    }
    // ---- this thread's cells: unit U0 + (lane & 31); sequence pairs sb[k], sb[k] + 1 where the granule (register pair rp = wave + 4 k,
    //      lane) of a source's 32 x 32 block lands: row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) with r = 2 rp --------------------------------
    const int u = fr, j = U0 + u;
    int sb[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) { const int r = 2 * (wave + 4 * k); sb[k] = (r & 3) + 8 * (r >> 2) + 4 * fq; }

    unsigned tag = 0;
    bool failed = false;
    const int n_tiles = (p.n_seq + T5_NS - 1) / T5_NS;
    for (int tile = team; tile < n_tiles && !failed; tile += n_teams) {
        const int b0 = tile * T5_NS;
        const int tmax = t5_tile_tmax(p, b0);
        int len[4];
        size_t row0[4];
        float nf[4], ndc[4];                                  // f_{t+1} and dc_{t+1} of the cell (carried from the previous iteration)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b = b0 + sb[q >> 1] + (q & 1);
            len[q] = b < p.n_seq ? p.seq_len[b] : 0;
            row0[q] = len[q] > 0 ? (size_t)p.seq_off[b] : 0;
            nf[q] = 0.f; ndc[q] = 0.f;
        }
        // no gate gradients yet: the first product (skipped) would read zeros
        for (int e = tid; e < 2 * T5_NS * T5_GROW / 4; e += T5_THREADS) reinterpret_cast<unsigned*>(dg0)[e] = 0u;
        // the cells' operands of a step are fetched ONE STEP AHEAD (see the forward)
        float gvn[4][4], dhn[4], csn[4], cpn[4];
        auto fetch = [&](int tt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool o = tt >= 0 && tt < len[q];
                const size_t rr = row0[q] + (size_t)(tt < 0 ? 0 : tt);
                const float* gt = p.gates + rr * GH + j;
#pragma unroll
                for (int g = 0; g < 4; ++g) gvn[q][g] = o ? gt[g * H] : 0.f;
                dhn[q] = o ? p.dh[rr * H + j] : 0.f;
                csn[q] = o ? p.cseq[rr * H + j] : 0.f;
                cpn[q] = o ? p.cprev[rr * H + j] : 0.f;
            }
        };
        fetch(tmax - 1);
        __syncthreads();

#pragma unroll 1
        for (int t = tmax - 1; t >= 0; --t) {
            char* const dcur = dg0 + (t & 1) * (T5_NS * T5_GROW);          // holds the gate gradients of step t + 1
            char* const dnxt = dg0 + ((t + 1) & 1) * (T5_NS * T5_GROW);    // receives those of step t
            ++tag;
            // (a) the cells' operands (fetched during the previous iteration)
            bool on[4], has_next[4];
            float gv[4][4], dhv[4], cs[4], cp[4];
            size_t r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                on[q] = t < len[q];
                has_next[q] = t + 1 < len[q];
                r[q] = row0[q] + (size_t)t;
#pragma unroll
                for (int g = 0; g < 4; ++g) gv[q][g] = gvn[q][g];
                dhv[q] = dhn[q]; cs[q] = csn[q]; cp[q] = cpn[q];
            }
            float rec[4] = {0.f, 0.f, 0.f, 0.f};
            if (t + 1 < tmax) {
                // (b) partial dh_rec[32 seq][32 units of owner o] over the 128 own gate columns, o = 4 wave + i
                f32x16 acc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
                const char* arow = dcur + fr * T5_GROW + fq * 16;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + ks * 32);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wreg[i][ks], acc[i], 0, 0, 0);
                }
                // (c) to the owners: own block through LDS, the others as granules [slot][owner][source = member][rp * 64 + lane]
                u64* const out_slot = ring + (size_t)(tag & (T5_SLOTS - 1)) * (T5_M * T5_M * T5_PAIRS);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = 4 * wave + i;
                    if (o == member) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) own[((q & 3) + 8 * (q >> 2) + 4 * fq) * T5_RED_LD + fr] = acc[i][q];
                    } else {
                        u64* dst = out_slot + ((size_t)o * T5_M + member) * T5_PAIRS + lane;
#pragma unroll
                        for (int rp = 0; rp < 8; ++rp)
                            granule_store(dst + rp * 64, __uint_as_float(cvt_pk_bf16(acc[i][2 * rp], acc[i][2 * rp + 1])), tag, plain);
                    }
                }
                __syncthreads();
                // (d) own partial + the fifteen sources'
                const u64* const in_slot = ring + (size_t)(tag & (T5_SLOTS - 1)) * (T5_M * T5_M * T5_PAIRS) + (size_t)member * T5_M * T5_PAIRS;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    rec[2 * k] = own[sb[k] * T5_RED_LD + u];
                    rec[2 * k + 1] = own[(sb[k] + 1) * T5_RED_LD + u];
                }
                {
                    u64 g[30];         // granule nn: source nn >> 1 (skipping this member), register-pair block k = nn & 1
                    const bool ok = t5_poll_all<30>([&](int nn) {
                        const int n = nn >> 1, src = n + (n >= member ? 1 : 0);
                        return in_slot + (size_t)src * T5_PAIRS + (nn & 1) * 256 + tid;
                    }, tag, g);
                    if (!ok) {
                        dead = 1;
                        team_report_timeout(p.fault, T5_K_BWD, p.layer, team, member, t, b0 + sb[0], tag);
                    }
#pragma unroll
                    for (int nn = 0; nn < 30; ++nn) {
                        const unsigned w = (unsigned)g[nn];
                        rec[2 * (nn & 1)] += t5_bf16_lo(w);
                        rec[2 * (nn & 1) + 1] += t5_bf16_hi(w);
                    }
                }
            }
            fetch(t - 1);        // behind the granule loads in the queue, a whole iteration ahead of their use
            // (e) the cells (rnn_step_bf16.hip's epilogue)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = sb[q >> 1] + (q & 1);
                float d4[4] = {0.f, 0.f, 0.f, 0.f};
                if (on[q]) {
                    float dh = dhv[q];
                    if (has_next[q]) dh += rec[q];
                    p.dh[r[q] * H + j] = dh;
                    const float ig = gv[q][0], fg = gv[q][1], gg = gv[q][2], og = gv[q][3];
                    const float tc = tanhf(cs[q]);
                    float dcv = dh * og * (1.f - tc * tc);
                    if (has_next[q]) dcv += ndc[q] * nf[q];              // dc_{t+1} * f_{t+1}
                    p.dc[r[q] * H + j] = dcv;
                    d4[0] = dcv * gg * ig * (1.f - ig);
                    d4[1] = dcv * cp[q] * fg * (1.f - fg);
                    d4[2] = dcv * ig * (1.f - gg * gg);
                    d4[3] = dh * tc * og * (1.f - og);
                    float* gx = p.dgx + r[q] * GH + j;
#pragma unroll
                    for (int g = 0; g < 4; ++g) gx[g * H] = d4[g];
                    nf[q] = fg; ndc[q] = dcv;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<uint16_t*>(dnxt + s * T5_GROW + (g * T5_US + u) * 2) = (uint16_t)(cvt_pk_bf16(d4[g], 0.f) & 0xffffu);
            }
            __syncthreads();
            if (dead) { failed = true; break; }
        }
        __syncthreads();
    }
    if (failed && tid < T5_US) {
        const int b = min(team * T5_NS, p.n_seq - 1);
        p.dgx[(size_t)p.seq_off[b] * GH + U0 + tid] = __builtin_nanf("");
    }
}

int t5_capacity() {
    static const int cap = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        return cus / T5_M;                 // one workgroup per CU (its registers hold 128 KB of weights, its LDS 83 KB)
    }();
    return cap < T5_MAXTEAMS ? cap : T5_MAXTEAMS;
}

int t5_teams(int n_seq) {
    const int tiles = (n_seq + T5_NS - 1) / T5_NS;
    int nt = tiles < t5_capacity() ? tiles : t5_capacity();
    if (nt > 8) nt = nt / 8 * 8;           // whole XCD slices: a team's sixteen members then share an L2
    return nt;
}

template <class K>
int t5_attr(K kern, int bytes, bool* done) {
    if (*done) return 0;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { set_error("lstm512_team: hipFuncSetAttribute", (int)e); return (int)e; }
    *done = true;
    return 0;
}

}  // namespace

long long lstm_team512_xbuf_bytes() { return (long long)((size_t)T5_HDR + T5_HS + (size_t)T5_MAXTEAMS * T5_BWD_RING) * (long long)sizeof(u64); }

// DC_DIMS_BF16 LSTM-512 layers whose W_hh arrived as bf16; DC_DIMS_RNN_STEP_BF16 (or DC_DIMS_RNN_PER_STEP) keeps the launch-per-step kernels
bool lstm_team512_supported(int cell, int H, int flags, const void* Wb) {
    return cell == 1 && H == T5_H && (flags & DC_DIMS_BF16) && !(flags & (DC_DIMS_RNN_PER_STEP | DC_DIMS_RNN_STEP_BF16)) && Wb != nullptr &&
           t5_capacity() >= 1;
}

int lstm_team512_forward(RnnStepArgs a, int max_len, hipStream_t s) {
    u64* xb = static_cast<u64*>(a.xbuf);
    if (!xb) { set_error("lstm_team512_forward: no exchange buffer (RnnStepArgs::xbuf)", 1012); return 1012; }
    static bool attr = false;
    if (int e = t5_attr(lstm512_team_fwd_kernel, T5_FWD_LDS, &attr)) return e;
    const int nt = t5_teams(a.n_seq);
    ProfScope prof("lstm_fwd_team", 2.0 * a.n_seq * 4.0 * a.H * a.H * max_len, 4.0 * a.n_seq * max_len * a.H * 12.0, s);
    if (int rc = zero_async(xb, ((size_t)T5_HDR + T5_HS + (size_t)nt * T5_FWD_RING) * sizeof(u64), s)) return rc;
    hipLaunchKernelGGL(lstm512_team_fwd_kernel, dim3(nt * T5_M), dim3(T5_THREADS), T5_FWD_LDS, s, a, a.Whh_bf, xb, nt,
                       !(a.flags & DC_DIMS_TEAM_DEVICE_SCOPE));
    return launch_check("lstm_team512_forward");
}

int lstm_team512_backward(RnnStepArgs a, int max_len, hipStream_t s) {
    u64* xb = static_cast<u64*>(a.xbuf);
    if (!xb) { set_error("lstm_team512_backward: no exchange buffer (RnnStepArgs::xbuf)", 1012); return 1012; }
    static bool attr = false;
    if (int e = t5_attr(lstm512_team_bwd_kernel, T5_BWD_LDS, &attr)) return e;
    const int nt = t5_teams(a.n_seq);
    ProfScope prof("lstm_bwd_team", 2.0 * a.n_seq * 4.0 * a.H * a.H * max_len, 4.0 * a.n_seq * max_len * a.H * 18.0, s);
    if (int rc = zero_async(xb, ((size_t)T5_HDR + T5_HS + (size_t)nt * T5_BWD_RING) * sizeof(u64), s)) return rc;
    hipLaunchKernelGGL(lstm512_team_bwd_kernel, dim3(nt * T5_M), dim3(T5_THREADS), T5_BWD_LDS, s, a, a.WhhT_bf, xb, nt,
                       !(a.flags & DC_DIMS_TEAM_DEVICE_SCOPE));
    return launch_check("lstm_team512_backward");
}

}  // namespace dc
