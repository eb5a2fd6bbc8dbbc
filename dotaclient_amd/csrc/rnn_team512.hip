// Persistent recurrent kernels for H = 512 on the bf16 path (DC_DIMS_BF16, BASELINE.json configs[4]: "5v5 hidden=512 2-layer LSTM,
// bf16 MFMA path"): the whole time loop of a layer in ONE launch.  Replaces nn.LSTM's recurrence at /root/reference/policy.py:66,141
// (cell parametrised as BASELINE.json asks) and its BPTT (/root/reference/optimizer.py:672); same arithmetic as the launch-per-step
// kernels of rnn_step_bf16.hip (bf16 operands, f32 accumulate, f32 state and gate maths), which took 13 / 23 us per time step -
// bound by the launch boundary and by gathering one row of every sequence per launch.
//
// W_hh of a 512-wide LSTM is 2 MB as bf16: no CU holds it.  A TEAM of 16 workgroups (16 CUs) does: member m owns the hidden units
// 32 m .. 32 m + 31, i.e. 128 gate columns x 512 = 128 KB of bf16 weights, resident in the registers of its four waves (128 VGPRs
// per lane) for the whole launch.  A team advances a TILE of NS = 16 or 32 sequences together on v_mfma_f32_32x32x16_bf16 (the
// sequences are the rows of every product) and works through its tiles one after the other.  NS = 16 while the batch then still fits
// one round of teams: 256 sequences = 16 teams on all 256 CUs instead of 8 on 128, and a member's per-step exchange, LDS reads,
// spill and cell work halve (measured: 4.6 / 6.3 -> 3.15 / 4.9 us per time step; the MFMA tile stays 32 rows, half of them unused -
// the product is not what bounds a step).
//
//   forward  (column-parallel): wave g of member m multiplies h_{t-1}[NS seq][512] (bf16, LDS) with its gate-g rows of W_hh:
//            32 MFMAs; the four gate blocks meet in LDS, a thread finishes four consecutive units of one sequence (one row address,
//            16-byte loads and stores); the member's h_t[NS seq][32 units] goes to the 15 peers as NS x 16 tagged granules
//            {bf16, bf16, tag} (team_util.h) - an all-gather of 15 x NS x 16 granule reads per member and step.
//   backward (row-parallel): a member contracts ITS OWN 128 gate gradients of step t + 1 (bf16, LDS - no input from anyone)
//            with W_hh[own column][u'] for all 512 output units: 32 MFMAs per wave; the partial sums dh_rec[NS seq][32 units] for
//            each owner go out as granules of two bf16 each (a reduce-scatter) and the owner adds the fifteen it receives to its
//            own.  Rounding the PARTIAL sums to bf16 is the one place this kernel is coarser than the per-step form (whose
//            2 048-term sums stay f32): sixteen roundings of 2^-9 each on sums of 128 terms - the same order as the bf16 rounding of
//            the 2 048 operands themselves; gradients agree with the step kernels to 1e-5 (tests/test_gpu_bf16.py).
//
// What shapes the step (profiles/r03/team512_phases.md): one wave per SIMD - nothing hides an instruction's latency, a step is as long
// as its instruction stream - and vector memory operations that complete IN ORDER.  Hence: all granules of a thread are polled as
// ONE batch and re-issued as a batch (t5_poll_all); the HBM operands of a step are fetched one (forward) / two (backward) steps
// ahead, behind the granule loads in the queue; the forward stores a step's results one step late; gates on v_exp / v_rcp.
//
// Roles by ticket, XCD-local teams when the team count is a multiple of 8, L2-scope granule stores when a team shares an XCD,
// timeouts reported in DC_WS_FAULT: as in rnn_team.hip / team_util.h.
#include <cstdio>
#include <type_traits>
#include "kernels.h"
#include "gemm_tiles.h"
#include "team_util.h"
#include "persist_util.h"

namespace dc {
namespace {

enum {
    T5_H = 512, T5_M = 16, T5_US = 32, T5_NS_MAX = 32, T5_THREADS = 256, T5_SLOTS = 4, T5_MAXTEAMS = 16,
    T5_HROW = T5_H * 2 + 16,                 // bytes: LDS row of the h tile [32 seq][512 k] (16 bytes of padding: conflict-free b128 reads)
    T5_GROW = 4 * T5_US * 2 + 16,            // bytes: LDS row of the own gate-gradient tile [32 seq][128 k]
    T5_RED_LD = 33,
    T5_HDR = 16,                             // u64 words: ticket counters
    T5_HS = T5_MAXTEAMS * T5_M,              // handshake granules
    T5_K_FWD = 5, T5_K_BWD = 6,              // kernel ids in the fault record (include/dotaclient_hip.h)
};

// A tile is NS = 32 or 16 sequences (template parameter of the kernels; the MFMA tile is 32 rows either way, the upper half unused at
// 16).  16 when the batch then still fits one round of teams: 256 sequences = 16 teams = all 256 CUs instead of 8 teams on 128, and a
// member's per-step exchange, LDS reads, spill and cell work halve.
constexpr int t5_pairs(int ns) { return ns * T5_US / 2; }                           // granules a member publishes per step (forward) / per owner (backward)
constexpr size_t t5_fwd_ring(int ns) { return (size_t)T5_SLOTS * T5_M * t5_pairs(ns); }          // u64 words per team
constexpr size_t t5_bwd_ring(int ns) { return (size_t)T5_SLOTS * T5_M * T5_M * t5_pairs(ns); }   // [slot][owner][source][pairs]
constexpr int t5_fwd_lds(int ns) { return 2 * ns * T5_HROW + 4 * ns * T5_RED_LD * 4; }
// (16-sequence tiles: + the staging images of the backward's wide HBM accesses: gates 4 KB, dh / c / c_prev 2 KB each, dh / dc out 2 x 2 KB each)
constexpr int t5_bwd_wide(int ns) { return ns == 16 ? 4096 + 3 * 2048 + 4 * 2048 : 0; }
constexpr int t5_bwd_lds(int ns) { return 2 * ns * T5_GROW + ns * T5_RED_LD * 4 + t5_bwd_wide(ns); }

__device__ __forceinline__ u32x4 t5_to_bf16x8(const float4& a, const float4& b) {
    return u32x4{cvt_pk_bf16(a.x, a.y), cvt_pk_bf16(a.z, a.w), cvt_pk_bf16(b.x, b.y), cvt_pk_bf16(b.z, b.w)};
}
typedef __attribute__((ext_vector_type(4))) float t5_f32x4;
// The activation streams (gate rows, states, gradients: each byte touched once per launch) are marked NON-TEMPORAL: they pass through
// the L2 the team's granules live in, and as ordinary accesses they evict them - measured on the backward, 256 x 512: 2 216 -> 1 917 us
// per pass (-DT5_NT=0: ordinary accesses, for A/B).
#ifndef T5_NT
#define T5_NT 1
#endif
typedef __attribute__((ext_vector_type(2))) unsigned t5_u32x2;
template <class T> __device__ __forceinline__ T t5_ld(const T* p) {
#if T5_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <class T> __device__ __forceinline__ void t5_st(T* p, T v) {
#if T5_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
// (HIP's float4 / uint2 are structs: through the matching ext vectors)
__device__ __forceinline__ float4 t5_ld(const float4* p) { const t5_f32x4 v = t5_ld(reinterpret_cast<const t5_f32x4*>(p)); return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ uint2 t5_ld(const uint2* p) { const t5_u32x2 v = t5_ld(reinterpret_cast<const t5_u32x2*>(p)); return make_uint2(v[0], v[1]); }
__device__ __forceinline__ void t5_st(float4* p, float4 v) { t5_st(reinterpret_cast<t5_f32x4*>(p), t5_f32x4{v.x, v.y, v.z, v.w}); }
__device__ __forceinline__ void t5_st(uint2* p, uint2 v) { t5_st(reinterpret_cast<t5_u32x2*>(p), t5_u32x2{v.x, v.y}); }
#define T5_LD(p) t5_ld(p)
#define T5_ST(p, v) t5_st(p, v)
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
    team = __builtin_amdgcn_readfirstlane(role_sh[0]);        // uniform: everything derived from them stays on the scalar unit
    member = __builtin_amdgcn_readfirstlane(role_sh[1]);
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
    return __builtin_amdgcn_readfirstlane(same_sh);
}

// All N granules of a thread in ONE batch, the whole batch re-issued until every tag matches: a retry costs one L2 round trip
// (~0.8 us) for all of them.  (Waiting for them one after the other, as the four-member kernels do with their three granules, made
// every stale first read pay its own round trip: 30 granules per thread, 5.6 us per step.)  false on timeout.
enum { T5_SPIN = 1 << 20 };
template <int N, class Addr>
__device__ __forceinline__ bool t5_poll_all(Addr addr, unsigned tag, u64 (&g)[N], long long* retries = nullptr) {
    for (int spins = 0;; ++spins) {
        if (DC_DEV_TIMING && retries) *retries += 1;
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

// The same in two halves, so that independent work can be issued between the first batch of loads and the wait for it (memory
// operations complete in order: what is issued BEHIND the granule loads does not delay them).
template <int N, class Addr>
__device__ __forceinline__ void t5_poll_issue(Addr addr, u64 (&g)[N]) {
#pragma unroll
    for (int n = 0; n < N; ++n) g[n] = granule_load(addr(n));
}
template <int N, class Addr>
__device__ __forceinline__ bool t5_poll_wait(Addr addr, unsigned tag, u64 (&g)[N]) {
    for (int spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int n = 0; n < N; ++n) ok = ok && (unsigned)(g[n] >> 32) == tag;
        if (ok) return true;
        if (spins > T5_SPIN) return false;
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int n = 0; n < N; ++n) g[n] = granule_load(addr(n));
    }
}

// longest sequence of the tile that starts at sequence b0 (all 256 threads call; result uniform)
__device__ __forceinline__ int t5_tile_tmax(const RnnStepArgs& p, int b0, int ns) {
    __shared__ int tmax_sh;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int b = b0 + (int)threadIdx.x;
        int v = ((int)threadIdx.x < ns && b < p.n_seq) ? p.seq_len[b] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
        if (threadIdx.x == 0) tmax_sh = v;
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(tmax_sh);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward.  In: gates = W_ih x + b_ih (overwritten by the activated gates), hprev / cprev rows 0 of every sequence = h0 / c0 (seeded
// by rnn_forward_layer); out: gates, cseq, hseq, hprev / cprev (the rows the backward and the weight gradients read).
// With one wave per SIMD nothing hides an instruction's latency and a wave64 VALU instruction occupies its SIMD for four cycles:
// the step is as long as its instruction stream.  Hence: a thread's cells are FOUR CONSECUTIVE UNITS OF ONE SEQUENCE (one row
// address, 16-byte loads and stores), row pointers advance by a stride, transcendental gates on v_exp / v_rcp.
// ---------------------------------------------------------------------------------------------------------------------------------
// BS (RnnStepArgs::bf16_store, configs[4]'s bf16 storage): `gates` is a bf16 [rows][4H] buffer - the input projections arrive rounded to
// bf16 (gemm_x3's epilogue) and the activated gates are left as bf16 for the backward: 8-byte instead of 16-byte accesses per thread.
template <int NS, bool BS>
__global__ __launch_bounds__(T5_THREADS, 1) void lstm512_team_fwd_kernel(RnnStepArgs p, const uint16_t* __restrict__ Wb,
                                                                          u64* __restrict__ xbuf_all, int n_teams, int allow_plain) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ht0 = smem;                                            // h tiles, double-buffered by the parity of t
    float* const red = reinterpret_cast<float*>(smem + 2 * NS * T5_HROW);   // [gate][seq][unit] pre-activations W_hh h
    __shared__ int dead;
    constexpr int H = T5_H, GH = 4 * T5_H, PAIRS = t5_pairs(NS), NPOLL = 15 * PAIRS / T5_THREADS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // = the gate this wave computes
    const int fr = lane & 31, fq = lane >> 5;
    int team, member;
    t5_claim_role(reinterpret_cast<unsigned*>(xbuf_all), n_teams, team, member);
    if (team < 0) return;
    const int plain = t5_same_xcd(xbuf_all + T5_HDR + team * T5_M, member, allow_plain);
    u64* const ring = xbuf_all + T5_HDR + T5_HS + (size_t)team * t5_fwd_ring(NS);
    const int U0 = T5_US * member;
    if (tid == 0) dead = 0;

    // ---- weights: rows (gate = wave, unit U0 + fr) of W_hh, all of K, as MFMA operands -------------------------------------------------
    // NS = 16: v_mfma_f32_16x16x32_bf16 - the sixteen sequences are ALL of the tile's rows (the 32 x 32 form spent half of its 32 cycles on
    // rows nobody reads, and read every h row twice): two column tiles ct (units U0 + 16 ct + (lane & 15)) x sixteen K = 32 steps, a lane
    // holds B[k = 32 ks + 8 (lane >> 4) + j][column lane & 15] and A[row lane & 15][the same k]
    const int l15 = lane & 15, l4 = lane >> 4;
    bf16x8 wreg[32];
    if constexpr (NS == 16) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const uint16_t* wrow = Wb + (size_t)(wave * H + U0 + 16 * ct + l15) * H + l4 * 8;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) wreg[ct * 16 + ks] = *reinterpret_cast<const bf16x8*>(wrow + ks * 32);
        }
    } else {
        const uint16_t* wrow = Wb + (size_t)(wave * H + U0 + fr) * H + fq * 8;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) wreg[ks] = *reinterpret_cast<const bf16x8*>(wrow + ks * 16);
    }
    // ---- this thread's cells: sequence tid >> 3 of the tile (NS = 16: the upper half of the threads has none), units 4 (tid & 7) .. + 3
    //      of the member's 32 ----------------------------------------------------------------------------------------------------------------
    const int cs_ = tid >> 3, ub = 4 * (tid & 7), j0 = U0 + ub;
    const bool has_cell = cs_ < NS;
    float4 bh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bh[g] = *reinterpret_cast<const float4*>(p.bhh + g * H + j0);

    // developer builds (DC_DEV_TIMING): phase clocks of (team 0, member 0, thread 0), summed over the steps
    long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    const bool timing = DC_DEV_TIMING && p.dbg != nullptr && team == 0 && member == 0 && tid == 0;
    auto stamp = [&](int k) {
        if (DC_DEV_TIMING && timing) { const long long now = (long long)__builtin_amdgcn_s_memtime(); tm[k] += now - tlast; tlast = now; }
    };
    unsigned tag = 0;
    bool failed = false;
    const int n_tiles = (p.n_seq + NS - 1) / NS;
    for (int tile = team; tile < n_tiles && !failed; tile += n_teams) {
        const int b0 = tile * NS;
        const int tmax = t5_tile_tmax(p, b0, NS);
        const int bq = b0 + cs_;
        const int len = (has_cell && bq < p.n_seq) ? p.seq_len[bq] : 0;
        const size_t row0 = len > 0 ? (size_t)p.seq_off[bq] : 0;
        using GT = std::conditional_t<BS, uint16_t, float>;
        GT* const gbase = reinterpret_cast<GT*>(p.gates) + row0 * GH + j0;      // row t: + t * GH
        // (the prefetched row stays RAW in its registers until the step that uses it: any arithmetic on it at the point of the load
        //  would wait for the load there - measured: + 3 600 clocks per step in the prefetch phase)
        using GR = std::conditional_t<BS, uint2, float4>;
        auto gload = [](const GT* q) -> GR { return T5_LD(reinterpret_cast<const GR*>(q)); };
        auto gdecode = [](const GR& w) -> float4 {
            if constexpr (BS) return make_float4(t5_bf16_lo(w.x), t5_bf16_hi(w.x), t5_bf16_lo(w.y), t5_bf16_hi(w.y));
            else return w;
        };
        const GR gzero = GR{};
        float* const sbase = p.cseq + row0 * H + j0;                   // cseq; hseq / cprev / hprev at the same offset of their buffers
        const ptrdiff_t d_h = p.hseq - p.cseq, d_cp = p.cprev - p.cseq, d_hp = p.hprev - p.cseq;
        float4 c = len > 0 ? *reinterpret_cast<const float4*>(sbase + d_cp) : make_float4(0.f, 0.f, 0.f, 0.f);
        // h0 of all 512 units of the tile's sequences -> tile buffer 0
        for (int e = tid; e < NS * 64; e += T5_THREADS) {
            const int s = e >> 6, k8 = (e & 63) * 8;
            const int b = b0 + s;
            u32x4 v = u32x4{0, 0, 0, 0};
            if (b < p.n_seq && p.seq_len[b] > 0) {
                if constexpr (BS) {
                    v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p.hprev) + (size_t)p.seq_off[b] * H + k8);
                } else {
                    const float* src = p.hprev + (size_t)p.seq_off[b] * H + k8;
                    v = t5_to_bf16x8(*reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4));
                }
            }
            *reinterpret_cast<u32x4*>(ht0 + s * T5_HROW + k8 * 2) = v;
        }
        // A step's results are STORED one step late, behind the next step's granule loads: vector memory operations complete in
        // order, so the scattered stores of a step would otherwise sit in front of the loads the next step waits for ...
        float4 pact[4], pc, ph;
        bool pend = false, pnext = false;
        int pt = 0;
        auto flush = [&]() {
            if (!pend) return;
            GT* gt = gbase + (size_t)pt * GH;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (BS) T5_ST(reinterpret_cast<uint2*>(gt + g * H), make_uint2(cvt_pk_bf16(pact[g].x, pact[g].y), cvt_pk_bf16(pact[g].z, pact[g].w)));
                else T5_ST(reinterpret_cast<float4*>(gt + g * H), pact[g]);
            }
            float* st = sbase + (size_t)pt * H;
            T5_ST(reinterpret_cast<float4*>(st), pc);
            if (pnext) T5_ST(reinterpret_cast<float4*>(st + H + d_cp), pc);
            if constexpr (BS) {      // hseq / hprev as bf16: the values the next product and the peers get anyway
                const uint2 hb = make_uint2(cvt_pk_bf16(ph.x, ph.y), cvt_pk_bf16(ph.z, ph.w));
                const size_t e = (row0 + (size_t)pt) * H + j0;
                T5_ST(reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.hseq) + e), hb);
                if (pnext) T5_ST(reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.hprev) + e + H), hb);
            } else {
                T5_ST(reinterpret_cast<float4*>(st + d_h), ph);
                if (pnext) T5_ST(reinterpret_cast<float4*>(st + H + d_hp), ph);
            }
            pend = false;
        };
        // ... and the input projections W_ih x + b_ih of a step are fetched ONE STEP AHEAD, for the same reason (a load from HBM
        // issued right before the granule loads would sit in front of every one of them)
        GR gxn[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) gxn[g] = len > 0 ? gload(gbase + g * H) : gzero;
        __syncthreads();

#pragma unroll 1
        for (int t = 0; t < tmax; ++t) {
            char* const hcur = ht0 + (t & 1) * (NS * T5_HROW);          // holds h_{t-1}
            char* const hnxt = ht0 + ((t + 1) & 1) * (NS * T5_HROW);    // receives h_t
            ++tag;
            if (DC_DEV_TIMING && timing && t == 0) tlast = (long long)__builtin_amdgcn_s_memtime();
            const bool on = t < len;
            float4 gx[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) gx[g] = gdecode(gxn[g]);
            // (b) the fifteen peers' h_{t-1}: NPOLL granules per thread (granule nn: peer 256 nn / PAIRS, pair 256 nn % PAIRS + tid)
            // (issuing the stores and the prefetch below BETWEEN the poll's loads and the wait for them, as the backward does, was measured
            //  here: 1 354 -> 1 680 us per pass - the wait then sits behind thirteen more queue entries)
            if (t > 0) {
                const unsigned rt = tag - 1;
                const u64* const slot = ring + (size_t)(rt & (T5_SLOTS - 1)) * (T5_M * PAIRS);
                u64 g[NPOLL];
                const bool ok = t5_poll_all<NPOLL>([&](int nn) {
                    const int pi = (nn * T5_THREADS) / PAIRS, mp = pi + (pi >= member ? 1 : 0);
                    return slot + (size_t)mp * PAIRS + (nn * T5_THREADS) % PAIRS + tid;
                }, rt, g, timing ? &tm[6] : nullptr);
                if (!ok) {
                    dead = 1;
                    team_report_timeout(p.fault, T5_K_FWD, p.layer, team, member, t, b0 + (tid >> 4), rt);
                }
#pragma unroll
                for (int nn = 0; nn < NPOLL; ++nn) {
                    const int pi = (nn * T5_THREADS) / PAIRS, mp = pi + (pi >= member ? 1 : 0);
                    const int pair = (nn * T5_THREADS) % PAIRS + tid;  // pair = sequence * 16 + unit pair of the peer's 32 units
                    *reinterpret_cast<unsigned*>(hcur + (pair >> 4) * T5_HROW + (T5_US * mp + 2 * (pair & 15)) * 2) = (unsigned)g[nn];
                }
            }
            stamp(0);      // poll + LDS writes
            // (ablation, round 5, 256 x 512: without these stores 1 363 -> 1 170 us per pass, without the prefetch 1 173, without both 1 134:
            //  the exchange / product / cell chain itself is 2.2 us per step, the HBM traffic costs 0.45 us on top)
            flush();       // the previous step's results
            {              // the next step's input projections
                const bool onn = t + 1 < len;
                const GT* gt = gbase + (size_t)(t + 1) * GH;
#pragma unroll
                for (int g = 0; g < 4; ++g) gxn[g] = onn ? gload(gt + g * H) : gzero;
            }
            __syncthreads();
            stamp(1);      // stores + prefetch issue + barrier
            if (dead) { failed = true; break; }
            // (c) [NS seq] x [32 gate columns] += h_{t-1} W_hh^T over K = 512
            if constexpr (NS == 16) {
                t5_f32x4 acc[2][2];                      // [column tile][chain]
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { acc[ct][0][q] = 0.f; acc[ct][1][q] = 0.f; }
                const char* arow = hcur + l15 * T5_HROW + l4 * 16;
                bf16x8 af[16];              // every fragment read is issued before the first MFMA (one wave per SIMD: nothing else hides LDS latency)
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(arow + ks * 64);
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    acc[0][ks & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], wreg[ks], acc[0][ks & 1], 0, 0, 0);
                    acc[1][ks & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], wreg[16 + ks], acc[1][ks & 1], 0, 0, 0);
                }
                // (d) C layout: column (= unit 16 ct + (lane & 15)), row (= sequence) 4 (lane >> 4) + r
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(wave * NS + 4 * l4 + r) * T5_RED_LD + 16 * ct + l15] = acc[ct][0][r] + acc[ct][1][r];
            } else {
            // two accumulator chains of v_mfma_f32_32x32x16_bf16
            f32x16 acc0, acc1;
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
            const char* arow = hcur + fr * T5_HROW + fq * 16;
            bf16x8 af[32];
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(arow + ks * 32);
#pragma unroll
            for (int ks = 0; ks < 32; ks += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], wreg[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks + 1], wreg[ks + 1], acc1, 0, 0, 0);
            }
            // (d) C layout: column (= unit) lane & 31, row (= sequence) (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * NS + (r & 3) + 8 * (r >> 2) + 4 * fq) * T5_RED_LD + fr] = acc0[r] + acc1[r];
            }
            stamp(2);      // product + spill
            __syncthreads();
            stamp(3);      // barrier
            // (e) the cells
            unsigned packed0 = 0, packed1 = 0;
            if (on) {
                float pre[4][4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float* rr = red + (g * NS + cs_) * T5_RED_LD + ub;
                    pre[g][0] = gx[g].x + rr[0] + bh[g].x; pre[g][1] = gx[g].y + rr[1] + bh[g].y;
                    pre[g][2] = gx[g].z + rr[2] + bh[g].z; pre[g][3] = gx[g].w + rr[3] + bh[g].w;
                }
                float cv[4] = {c.x, c.y, c.z, c.w}, hv[4], act[4][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ig = fast_sigmoid(pre[0][e]), fg = fast_sigmoid(pre[1][e]), gg = fast_tanh(pre[2][e]), og = fast_sigmoid(pre[3][e]);
                    cv[e] = fg * cv[e] + ig * gg;
                    hv[e] = og * fast_tanh(cv[e]);
                    act[0][e] = ig; act[1][e] = fg; act[2][e] = gg; act[3][e] = og;
                }
                c = make_float4(cv[0], cv[1], cv[2], cv[3]);
#pragma unroll
                for (int g = 0; g < 4; ++g) pact[g] = make_float4(act[g][0], act[g][1], act[g][2], act[g][3]);
                pc = c;
                ph = make_float4(hv[0], hv[1], hv[2], hv[3]);
                pt = t; pnext = t + 1 < len; pend = true;
                packed0 = cvt_pk_bf16(hv[0], hv[1]);
                packed1 = cvt_pk_bf16(hv[2], hv[3]);
            }
            // own units of h_t for the next product, and the same values to the peers as pairs 2 tid, 2 tid + 1 (finished sequences
            // publish zeros: every peer waits for all 512 granules of every member)
            if (has_cell) *reinterpret_cast<uint2*>(hnxt + cs_ * T5_HROW + j0 * 2) = make_uint2(packed0, packed1);
            if (has_cell && t + 1 < tmax) {
                u64* const out = ring + (size_t)(tag & (T5_SLOTS - 1)) * (T5_M * PAIRS) + (size_t)member * PAIRS + 2 * tid;
                granule_store(out, __uint_as_float(packed0), tag, plain);
                granule_store(out + 1, __uint_as_float(packed1), tag, plain);
            }
            stamp(4);      // cells + publish
        }
        flush();
        __syncthreads();
    }
    if (DC_DEV_TIMING && timing) { tm[7] = plain; for (int k = 0; k < 8; ++k) p.dbg[k] = tm[k]; }
    if (failed && tid < T5_US) {        // a peer never answered: make the failure visible downstream (NaN loss -> status 1)
        const int b = min(team * NS, p.n_seq - 1);
        if constexpr (BS) reinterpret_cast<uint16_t*>(p.hseq)[(size_t)p.seq_off[b] * H + U0 + tid] = 0x7fc0u;      // bf16 NaN
        else p.hseq[(size_t)p.seq_off[b] * H + U0 + tid] = __builtin_nanf("");
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward through time.  In: dh (from the layer above / the heads), the forward's gates / cseq / cprev; out: dgx (gradient w.r.t.
// W_ih x + b_ih = w.r.t. W_hh h + b_hh), dh (total), dc.
// ---------------------------------------------------------------------------------------------------------------------------------
// BS: the forward's gates and this kernel's gate gradients `dgx` are bf16 [rows][4H] buffers (the weight- and input-gradient products
// read dgx as a bf16 operand anyway)
template <int NS, bool BS>
__global__ __launch_bounds__(T5_THREADS, 1) void lstm512_team_bwd_kernel(RnnStepArgs p, const uint16_t* __restrict__ WTb,
                                                                          u64* __restrict__ xbuf_all, int n_teams, int allow_plain) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const dg0 = smem;                                             // own gate gradients [seq][gate * 32 + unit] bf16, double-buffered
    float* const own = reinterpret_cast<float*>(smem + 2 * NS * T5_GROW);   // own partial sums [seq][unit]
    __shared__ int dead;
    constexpr int H = T5_H, GH = 4 * T5_H, PAIRS = t5_pairs(NS), NPOLL = 15 * PAIRS / T5_THREADS, KP = NS / 16, NC = 2 * KP;   // NC cells per thread
    // WIDE (bf16 storage, 16-sequence tiles - configs[4]'s case): every HBM access of a step is a 16-byte-per-lane access of a whole
    // 64 / 128-byte row segment, staged through LDS images - three loads and two stores per thread and step instead of fourteen 2 / 4-byte
    // loads and twelve stores (measured by ablation, 256 x 512: the narrow loads cost 900 us and the stores 370 us of a 2 260 us pass).
    //   gates / dgx image [seq 16][gate 4][unit 32] bf16: chunk tid = 16 bytes = (seq tid >> 4, gate (tid >> 2) & 3, units 8 (tid & 3) ..)
    //   dh / c / c_prev / dh out / dc out images [seq 16][unit 32] f32: chunk cf = tid & 127 = (seq cf >> 3, units 4 (cf & 7) ..);
    //   waves 0, 1 move dh and c_prev in and dh out, waves 2, 3 move c in and dc out.
    constexpr bool WIDE = BS && NS == 16;
    char* const wbase = smem + 2 * NS * T5_GROW + NS * T5_RED_LD * 4;
    uint16_t* const SG = reinterpret_cast<uint16_t*>(wbase);                 // the step's gates
    float* const SD = reinterpret_cast<float*>(wbase + 4096);               // dh (in), c, c_prev
    float* const SC = SD + 512;
    float* const SP = SC + 512;
    float* const OD = SP + 512;                                             // dh (total), dc: [parity of t][seq][unit]
    float* const OC = OD + 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fq = lane >> 5;
    int team, member;
    t5_claim_role(reinterpret_cast<unsigned*>(xbuf_all), n_teams, team, member);
    if (team < 0) return;
    const int plain = t5_same_xcd(xbuf_all + T5_HDR + team * T5_M, member, allow_plain);
    u64* const ring = xbuf_all + T5_HDR + T5_HS + (size_t)team * t5_bwd_ring(NS);
    const int U0 = T5_US * member;
    if (tid == 0) dead = 0;

    // ---- weights: for the four owners o = 4 wave + i: rows W_hh^T[unit 32 o + fr][own columns], K = 128 own gate columns ----------------
    // NS = 16: v_mfma_f32_16x16x32_bf16 (see the forward): per owner two column tiles ct (units 16 ct + (lane & 15) of the owner) x four
    // K = 32 steps g (the own 32 columns of gate g: k = 8 (lane >> 4) + j)
    const int l15 = lane & 15, l4 = lane >> 4;
    bf16x8 wreg[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (NS == 16) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const uint16_t* wrow = WTb + (size_t)(T5_US * (4 * wave + i) + 16 * ct + l15) * GH + U0 + l4 * 8;
#pragma unroll
                for (int g = 0; g < 4; ++g) wreg[i][ct * 4 + g] = *reinterpret_cast<const bf16x8*>(wrow + g * H);
            }
        } else {
            const uint16_t* wrow = WTb + (size_t)(T5_US * (4 * wave + i) + fr) * GH + U0 + fq * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) wreg[i][ks] = *reinterpret_cast<const bf16x8*>(wrow + (ks >> 1) * H + (ks & 1) * 16);
        }
    }
    // ---- this thread's cells: unit U0 + (lane & 31); sequence pairs sb[k], sb[k] + 1 where the granule (register pair rp = wave + 4 k,
    //      lane) of a source's 32 x 32 block lands: row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) with r = 2 rp --------------------------------
    //      (NS = 16, 16 x 16 accumulators: granule (column tile ct, register pair rp) of a lane = rows 4 (lane >> 4) + 2 rp, + 1 of column
    //      16 ct + (lane & 15); the thread that receives granule index 64 (2 ct + rp) + lane is (wave = 2 ct + rp, lane))
    const int u = NS == 16 ? 16 * (wave >> 1) + l15 : fr, j = U0 + u;
    int sb[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) { const int r = 2 * (wave + 4 * k); sb[k] = NS == 16 ? 4 * l4 + 2 * (wave & 1) : (r & 3) + 8 * (r >> 2) + 4 * fq; }

    long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    const bool timing = DC_DEV_TIMING && p.dbg != nullptr && team == 0 && member == 0 && tid == 0;
    auto stamp = [&](int k) {
        if (DC_DEV_TIMING && timing) { const long long now = (long long)__builtin_amdgcn_s_memtime(); tm[k] += now - tlast; tlast = now; }
    };
    unsigned tag = 0;
    bool failed = false;
    const int n_tiles = (p.n_seq + NS - 1) / NS;
    for (int tile = team; tile < n_tiles && !failed; tile += n_teams) {
        const int b0 = tile * NS;
        const int tmax = t5_tile_tmax(p, b0, NS);
        int len[NC];
        size_t row0[NC];
        float nf[NC], ndc[NC];                                // f_{t+1} and dc_{t+1} of the cell (carried from the previous iteration)
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int b = b0 + sb[q >> 1] + (q & 1);
            len[q] = b < p.n_seq ? p.seq_len[b] : 0;
            row0[q] = len[q] > 0 ? (size_t)p.seq_off[b] : 0;
            nf[q] = 0.f; ndc[q] = 0.f;
        }
        // WIDE: the sequences of this thread's chunks
        const int cf = tid & 127;
        const int bG = b0 + (tid >> 4), bF = b0 + (cf >> 3);
        const int lenG = (WIDE && bG < p.n_seq) ? p.seq_len[bG] : 0, lenF = (WIDE && bF < p.n_seq) ? p.seq_len[bF] : 0;
        const size_t rowG = lenG > 0 ? (size_t)p.seq_off[bG] : 0, rowF = lenF > 0 ? (size_t)p.seq_off[bF] : 0;
        const int gcol = ((tid >> 2) & 3) * H + U0 + 8 * (tid & 3), fcol = U0 + 4 * (cf & 7);
        // no gate gradients yet: the first product (skipped) would read zeros
        for (int e = tid; e < 2 * NS * T5_GROW / 4; e += T5_THREADS) reinterpret_cast<unsigned*>(dg0)[e] = 0u;
        // The cells' operands of a step are fetched TWO STEPS AHEAD into one of two register sets (static indices: the loop is unrolled by
        // two).  Two, not one: the loads come from HBM (> 2 us) and are issued behind a step's granule loads; a set loaded during step
        // t + 1 and copied out at the top of step t was waited for there (about 1 us per step).
        // (BS: the gates stay the raw 16 bits until the step that uses them - a shift at the point of the load would wait for it there)
        using GV = std::conditional_t<BS, unsigned, float>;
        GV gvn[2][NC][4];
        float dhn[2][NC], csn[2][NC], cpn[2][NC];
        u32x4 rg[2], ra[2], rb[2];                           // WIDE: the raw chunks of a step, by parity (they stay raw until they are staged)
        auto fetch = [&](int tt, auto PAR) {
            constexpr int P = decltype(PAR)::value;
            if constexpr (WIDE) {
                // NO select on the loaded data (it would wait for the load right here): a step outside the sequence re-reads a row inside it
                // (the first row of the buffer for an absent sequence) - the cells ignore what is staged for a step they are not `on`
                const size_t tG = (size_t)max(min(tt, lenG - 1), 0), tF = (size_t)max(min(tt, lenF - 1), 0);
                rg[P] = T5_LD(reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p.gates) + (rowG + tG) * GH + gcol));
                if (wave < 2) {
                    ra[P] = T5_LD(reinterpret_cast<const u32x4*>(p.dh + (rowF + tF) * H + fcol));
                    rb[P] = T5_LD(reinterpret_cast<const u32x4*>(p.cprev + (rowF + tF) * H + fcol));
                } else {
                    ra[P] = T5_LD(reinterpret_cast<const u32x4*>(p.cseq + (rowF + tF) * H + fcol));
                }
                return;
            }
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const bool o = tt >= 0 && tt < len[q];
                const size_t rr = row0[q] + (size_t)(tt < 0 ? 0 : tt);
                if constexpr (BS) {
                    const uint16_t* gt = reinterpret_cast<const uint16_t*>(p.gates) + rr * GH + j;
#pragma unroll
                    for (int g = 0; g < 4; ++g) gvn[P][q][g] = o ? (unsigned)T5_LD(gt + g * H) : 0u;
                } else {
                    const float* gt = p.gates + rr * GH + j;
#pragma unroll
                    for (int g = 0; g < 4; ++g) gvn[P][q][g] = o ? T5_LD(gt + g * H) : 0.f;
                }
                dhn[P][q] = o ? T5_LD(p.dh + rr * H + j) : 0.f;
                csn[P][q] = o ? T5_LD(p.cseq + rr * H + j) : 0.f;
                cpn[P][q] = o ? T5_LD(p.cprev + rr * H + j) : 0.f;
            }
        };
        fetch(tmax - 1, std::integral_constant<int, 0>{});
        fetch(tmax - 2, std::integral_constant<int, 1>{});
        // a step's results are stored one step late, BEHIND the next step's granule loads and while those are in flight
        float pd[NC][6];
        size_t prow[NC];
        bool pend[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) { pend[q] = false; prow[q] = 0; }
        // WIDE: the results of step tt from their LDS images (the gate-gradient tile the next product reads; the dh / dc images)
        auto flush_wide = [&](int tt) {
            if (tt < lenG) {
                const char* tile = dg0 + ((tt + 1) & 1) * (NS * T5_GROW);
                const u32x4 v = *reinterpret_cast<const u32x4*>(tile + (tid >> 4) * T5_GROW + (((tid >> 2) & 3) * T5_US + 8 * (tid & 3)) * 2);
                T5_ST(reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.dgx) + (rowG + (size_t)tt) * GH + gcol), v);
            }
            if (tt < lenF) {
                const float* img = (wave < 2 ? OD : OC) + (tt & 1) * 512 + cf * 4;
                float* dst = (wave < 2 ? p.dh : p.dc) + (rowF + (size_t)tt) * H + fcol;
                T5_ST(reinterpret_cast<u32x4*>(dst), *reinterpret_cast<const u32x4*>(img));
            }
        };
        auto flush = [&]() {
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                if (!pend[q]) continue;
                T5_ST(p.dh + prow[q] * H + j, pd[q][4]);
                T5_ST(p.dc + prow[q] * H + j, pd[q][5]);
                if constexpr (BS) {
                    uint16_t* gx = reinterpret_cast<uint16_t*>(p.dgx) + prow[q] * GH + j;
#pragma unroll
                    for (int g = 0; g < 4; ++g) T5_ST(gx + g * H, (uint16_t)(cvt_pk_bf16(pd[q][g], 0.f) & 0xffffu));
                } else {
                    float* gx = p.dgx + prow[q] * GH + j;
#pragma unroll
                    for (int g = 0; g < 4; ++g) T5_ST(gx + g * H, pd[q][g]);
                }
                pend[q] = false;
            }
        };
        __syncthreads();

        auto step = [&](const int t, auto PAR) -> bool {
            constexpr int P = decltype(PAR)::value;
            char* const dcur = dg0 + (t & 1) * (NS * T5_GROW);          // holds the gate gradients of step t + 1
            char* const dnxt = dg0 + ((t + 1) & 1) * (NS * T5_GROW);    // receives those of step t
            ++tag;
            if (DC_DEV_TIMING && timing && t == tmax - 1) tlast = (long long)__builtin_amdgcn_s_memtime();
            stamp(5);      // (loop control)
            // (a) the cells' operands (fetched during the previous iteration)
            bool on[NC], has_next[NC];
            float gv[NC][4], dhv[NC], cs[NC], cp[NC];
            size_t r[NC];
            if constexpr (WIDE) {      // the step's chunks (fetched two steps ago) -> their images; the cells read them behind the next barrier
                reinterpret_cast<u32x4*>(SG)[tid] = rg[P];
                if (wave < 2) { reinterpret_cast<u32x4*>(SD)[cf] = ra[P]; reinterpret_cast<u32x4*>(SP)[cf] = rb[P]; }
                else reinterpret_cast<u32x4*>(SC)[cf] = ra[P];
            }
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                on[q] = t < len[q];
                has_next[q] = t + 1 < len[q];
                r[q] = row0[q] + (size_t)t;
                if constexpr (!WIDE) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if constexpr (BS) gv[q][g] = __uint_as_float(gvn[P][q][g] << 16);
                        else gv[q][g] = gvn[P][q][g];
                    }
                    dhv[q] = dhn[P][q]; cs[q] = csn[P][q]; cp[q] = cpn[P][q];
                }
            }
            float rec[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) rec[q] = 0.f;
            if (t + 1 < tmax) {
                // (b) partial dh_rec[NS seq][32 units of owner o] over the 128 own gate columns, o = 4 wave + i
                u64* const out_slot = ring + (size_t)(tag & (T5_SLOTS - 1)) * (T5_M * T5_M * PAIRS);
                if constexpr (NS == 16) {
                    t5_f32x4 acc[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) { acc[i][0][q] = 0.f; acc[i][1][q] = 0.f; }
                    const char* arow = dcur + l15 * T5_GROW + l4 * 16;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + g * 64);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wreg[i][g], acc[i][0], 0, 0, 0);
                            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wreg[i][4 + g], acc[i][1], 0, 0, 0);
                        }
                    }
                    stamp(0);      // operand copies + product
                    // (c) to the owners: own block through LDS, the others as granules [slot][owner][source = member][64 (2 ct + rp) + lane]
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int o = 4 * wave + i;
                        if (o == member) {
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                                for (int q = 0; q < 4; ++q) own[(4 * l4 + q) * T5_RED_LD + 16 * ct + l15] = acc[i][ct][q];
                        } else {
                            u64* dst = out_slot + ((size_t)o * T5_M + member) * PAIRS + lane;
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                                for (int rp = 0; rp < 2; ++rp)
                                    granule_store(dst + (2 * ct + rp) * 64, __uint_as_float(cvt_pk_bf16(acc[i][ct][2 * rp], acc[i][ct][2 * rp + 1])), tag, plain);
                        }
                    }
                } else {
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
                stamp(0);      // operand copies + product
                // (c) to the owners: own block through LDS, the others as granules [slot][owner][source = member][rp * 64 + lane]
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = 4 * wave + i;
                    if (o == member) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) own[((q & 3) + 8 * (q >> 2) + 4 * fq) * T5_RED_LD + fr] = acc[i][q];
                    } else {
                        u64* dst = out_slot + ((size_t)o * T5_M + member) * PAIRS + lane;
#pragma unroll
                        for (int rp = 0; rp < 8; ++rp)
                            granule_store(dst + rp * 64, __uint_as_float(cvt_pk_bf16(acc[i][2 * rp], acc[i][2 * rp + 1])), tag, plain);
                    }
                }
                }
                stamp(1);      // publish
                __syncthreads();
                stamp(2);      // barrier
                // (d) the fifteen sources' granules: issued now, waited for after the work below
                const u64* const in_slot = ring + (size_t)(tag & (T5_SLOTS - 1)) * (T5_M * T5_M * PAIRS) + (size_t)member * T5_M * PAIRS;
                auto addr = [&](int nn) {
                    const int n = (nn * T5_THREADS) / PAIRS, src = n + (n >= member ? 1 : 0);
                    return in_slot + (size_t)src * PAIRS + (nn * T5_THREADS) % PAIRS + tid;
                };
                u64 g[NPOLL];      // granule nn: source 256 nn / PAIRS (skipping this member), register-pair block k = (256 nn % PAIRS) / 256
                t5_poll_issue<NPOLL>(addr, g);
                // ... meanwhile: the previous step's stores, the operand fetch for two steps ahead, and everything of the cells that does
                // not need the received sums.  (Round 5, 256 x 512, per pass: with these accesses AFTER the wait 2 240 us, here 1 922, without the
                // stores 1 726, without any of them 1 177.  What they cost is not their count (16-byte staged accesses: - 2 %) and no longer the
                // L2 (non-temporal: - 13 %); cause unknown.  Also tried: fetching for a PAIR of steps in its first step, none in its second: 1 898 - within noise, so it is
                // not an HBM load holding the in-order queue's head while a step's sixty granule stores are issued.)
                if constexpr (WIDE) flush_wide(t + 1);
                else flush();
                fetch(t - 2, PAR);
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    rec[2 * k] = own[sb[k] * T5_RED_LD + u];
                    rec[2 * k + 1] = own[(sb[k] + 1) * T5_RED_LD + u];
                }
                if (!t5_poll_wait<NPOLL>(addr, tag, g)) {
                    dead = 1;
                    team_report_timeout(p.fault, T5_K_BWD, p.layer, team, member, t, b0 + sb[0], tag);
                }
#pragma unroll
                for (int nn = 0; nn < NPOLL; ++nn) {
                    const int k = ((nn * T5_THREADS) % PAIRS) / T5_THREADS;
                    const unsigned w = (unsigned)g[nn];
                    rec[2 * k] += t5_bf16_lo(w);
                    rec[2 * k + 1] += t5_bf16_hi(w);
                }
            } else {
                if constexpr (WIDE) __syncthreads();      // (the images; with an exchange the barrier behind the publish has done it)
                else flush();
                fetch(t - 2, PAR);
            }
            if constexpr (WIDE) {
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const int e = (sb[q >> 1] + (q & 1)) * T5_US + u;
#pragma unroll
                    for (int g = 0; g < 4; ++g) gv[q][g] = __uint_as_float((unsigned)SG[e + (3 * (sb[q >> 1] + (q & 1)) + g) * T5_US] << 16);
                    dhv[q] = SD[e]; cs[q] = SC[e]; cp[q] = SP[e];
                }
            }
            stamp(3);      // poll (+ stores, prefetch under it) + sum
            // (e) the cells (rnn_step_bf16.hip's epilogue)
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int s = sb[q >> 1] + (q & 1);
                float d4[4] = {0.f, 0.f, 0.f, 0.f};
                if (on[q]) {
                    float dh = dhv[q];
                    if (has_next[q]) dh += rec[q];
                    const float ig = gv[q][0], fg = gv[q][1], gg = gv[q][2], og = gv[q][3];
                    const float tc = fast_tanh(cs[q]);
                    float dcv = dh * og * (1.f - tc * tc);
                    if (has_next[q]) dcv += ndc[q] * nf[q];              // dc_{t+1} * f_{t+1}
                    d4[0] = dcv * gg * ig * (1.f - ig);
                    d4[1] = dcv * cp[q] * fg * (1.f - fg);
                    d4[2] = dcv * ig * (1.f - gg * gg);
                    d4[3] = dh * tc * og * (1.f - og);
                    if constexpr (WIDE) {
                        OD[(t & 1) * 512 + s * T5_US + u] = dh;
                        OC[(t & 1) * 512 + s * T5_US + u] = dcv;
                    } else {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) pd[q][g4] = d4[g4];
                        pd[q][4] = dh; pd[q][5] = dcv;
                        prow[q] = r[q];
                        pend[q] = true;
                    }
                    nf[q] = fg; ndc[q] = dcv;
                }
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    *reinterpret_cast<uint16_t*>(dnxt + s * T5_GROW + (g4 * T5_US + u) * 2) = (uint16_t)(cvt_pk_bf16(d4[g4], 0.f) & 0xffffu);
            }
            stamp(4);      // prefetch issue + cells + stores
            __syncthreads();
            stamp(6);      // barrier
            return dead == 0;
        };
#pragma unroll 1
        for (int t = tmax - 1; t >= 0; t -= 2) {
            if (!step(t, std::integral_constant<int, 0>{})) { failed = true; break; }
            if (t - 1 >= 0 && !step(t - 1, std::integral_constant<int, 1>{})) { failed = true; break; }
        }
        if constexpr (WIDE) { if (!failed) flush_wide(0); }
        else flush();
        __syncthreads();
    }
    if (DC_DEV_TIMING && timing) { tm[7] = plain; for (int k = 0; k < 8; ++k) p.dbg[k] = tm[k]; }
    if (failed && tid < T5_US) {
        const int b = min(team * NS, p.n_seq - 1);
        if constexpr (BS) reinterpret_cast<uint16_t*>(p.dgx)[(size_t)p.seq_off[b] * GH + U0 + tid] = 0x7fc0u;      // bf16 NaN
        else p.dgx[(size_t)p.seq_off[b] * GH + U0 + tid] = __builtin_nanf("");
    }
}

// Teams of sixteen one-per-CU workgroups the CURRENT device can hold (queried per device id: a process may drive several different
// devices).  Roles are taken by ticket, so a team never waits for a workgroup that is not running - but the kernels do
// assume that nt * 16 workgroups get a CU each; a CU mask or a co-tenant that takes CUs away shows up as the DC_WS_FAULT timeout.
int t5_capacity() {
    enum { MAXDEV = 64 };
    static int cap_of[MAXDEV];             // 0 = not queried yet
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return 0;
    int cap = __atomic_load_n(&cap_of[dev], __ATOMIC_RELAXED);
    if (cap == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cap = cus / T5_M;                  // one workgroup per CU (its registers hold 128 KB of weights, its LDS 83 KB)
        __atomic_store_n(&cap_of[dev], cap, __ATOMIC_RELAXED);
    }
    return cap < T5_MAXTEAMS ? cap : T5_MAXTEAMS;
}

// sequences per tile: 16 when the batch then still runs in one round of teams (twice the teams, half the per-step work of a member)
int t5_tile_seqs(int n_seq) { return n_seq <= 16 * t5_capacity() ? 16 : 32; }

int t5_teams(int n_seq, int ns) {
    const int tiles = (n_seq + ns - 1) / ns;
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

long long lstm_team512_xbuf_bytes() { return (long long)((size_t)T5_HDR + T5_HS + (size_t)T5_MAXTEAMS * t5_bwd_ring(T5_NS_MAX)) * (long long)sizeof(u64); }

// DC_DIMS_BF16 LSTM-512 layers whose W_hh arrived as bf16; DC_DIMS_RNN_STEP_BF16 (or DC_DIMS_RNN_PER_STEP) keeps the launch-per-step kernels
bool lstm_team512_supported(int cell, int H, int flags, const void* Wb) {
    return cell == 1 && H == T5_H && (flags & DC_DIMS_BF16) && !(flags & (DC_DIMS_RNN_PER_STEP | DC_DIMS_RNN_STEP_BF16)) && Wb != nullptr &&
           t5_capacity() >= 1;
}

int lstm_team512_forward(RnnStepArgs a, int max_len, hipStream_t s) {
    u64* xb = static_cast<u64*>(a.xbuf);
    if (!xb) { set_error("lstm_team512_forward: no exchange buffer (RnnStepArgs::xbuf)", 1012); return 1012; }
    static bool attr[4] = {false, false, false, false};
    if (int e = t5_attr(lstm512_team_fwd_kernel<16, false>, t5_fwd_lds(16), &attr[0])) return e;
    if (int e = t5_attr(lstm512_team_fwd_kernel<32, false>, t5_fwd_lds(32), &attr[1])) return e;
    if (int e = t5_attr(lstm512_team_fwd_kernel<16, true>, t5_fwd_lds(16), &attr[2])) return e;
    if (int e = t5_attr(lstm512_team_fwd_kernel<32, true>, t5_fwd_lds(32), &attr[3])) return e;
    const int ns = (a.flags & DC_DIMS_TEAM_NS(2)) ? 32 : t5_tile_seqs(a.n_seq);     // DC_DIMS_TEAM_NS(2): force 32-sequence tiles (A/B)
    const int nt = t5_teams(a.n_seq, ns);
    // per cell: gates in + out (4 + 4 values), h, hprev (f32 or bf16), c, cprev (f32)
    ProfScope prof("lstm_fwd_team", 2.0 * a.n_seq * 4.0 * a.H * a.H * max_len, (double)a.n_seq * max_len * a.H * (a.bf16_store ? 10 * 2.0 + 8.0 : 48.0), s);
    if (int rc = zero_async(xb, ((size_t)T5_HDR + T5_HS + (size_t)nt * (ns == 16 ? t5_fwd_ring(16) : t5_fwd_ring(32))) * sizeof(u64), s)) return rc;
#if DC_DEV_TIMING
    static long long* dbg = nullptr;
    if (!dbg) (void)hipMalloc(&dbg, 64);
    (void)hipMemsetAsync(dbg, 0, 64, s);
    a.dbg = dbg;
#endif
    const int plain_ok = !(a.flags & DC_DIMS_TEAM_DEVICE_SCOPE);
    const dim3 grid(nt * T5_M), block(T5_THREADS);
    if (ns == 16 && a.bf16_store) hipLaunchKernelGGL((lstm512_team_fwd_kernel<16, true>), grid, block, t5_fwd_lds(16), s, a, a.Whh_bf, xb, nt, plain_ok);
    else if (ns == 16) hipLaunchKernelGGL((lstm512_team_fwd_kernel<16, false>), grid, block, t5_fwd_lds(16), s, a, a.Whh_bf, xb, nt, plain_ok);
    else if (a.bf16_store) hipLaunchKernelGGL((lstm512_team_fwd_kernel<32, true>), grid, block, t5_fwd_lds(32), s, a, a.Whh_bf, xb, nt, plain_ok);
    else hipLaunchKernelGGL((lstm512_team_fwd_kernel<32, false>), grid, block, t5_fwd_lds(32), s, a, a.Whh_bf, xb, nt, plain_ok);
#if DC_DEV_TIMING
    {
        long long h[8];
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        const double n = (double)max_len * (((a.n_seq + ns - 1) / ns + nt - 1) / nt);
        fprintf(stderr, "lstm512_team_fwd timing (clocks per step, %.0f steps): poll %.0f  prefetch+barrier %.0f  product+spill %.0f  barrier %.0f  "
                        "cells+publish %.0f  | poll rounds per step %.2f  plain %lld\n", n, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[6] / n, h[7]);
    }
#endif
    return launch_check("lstm_team512_forward");
}

int lstm_team512_backward(RnnStepArgs a, int max_len, hipStream_t s) {
    u64* xb = static_cast<u64*>(a.xbuf);
    if (!xb) { set_error("lstm_team512_backward: no exchange buffer (RnnStepArgs::xbuf)", 1012); return 1012; }
    static bool attr[4] = {false, false, false, false};
    if (int e = t5_attr(lstm512_team_bwd_kernel<16, false>, t5_bwd_lds(16), &attr[0])) return e;
    if (int e = t5_attr(lstm512_team_bwd_kernel<32, false>, t5_bwd_lds(32), &attr[1])) return e;
    if (int e = t5_attr(lstm512_team_bwd_kernel<16, true>, t5_bwd_lds(16), &attr[2])) return e;
    if (int e = t5_attr(lstm512_team_bwd_kernel<32, true>, t5_bwd_lds(32), &attr[3])) return e;
    const int ns = (a.flags & DC_DIMS_TEAM_NS(2)) ? 32 : t5_tile_seqs(a.n_seq);
    const int nt = t5_teams(a.n_seq, ns);
#if DC_DEV_TIMING
    static long long* dbg = nullptr;
    if (!dbg) (void)hipMalloc(&dbg, 64);
    (void)hipMemsetAsync(dbg, 0, 64, s);
    a.dbg = dbg;
#endif
    // per cell: gates in, dgx out (4 + 4 values, f32 or bf16), dh in + out, dc out, c, cprev (f32)
    ProfScope prof("lstm_bwd_team", 2.0 * a.n_seq * 4.0 * a.H * a.H * max_len, (double)a.n_seq * max_len * a.H * (a.bf16_store ? 8 * 2.0 + 20.0 : 52.0), s);
    if (int rc = zero_async(xb, ((size_t)T5_HDR + T5_HS + (size_t)nt * (ns == 16 ? t5_bwd_ring(16) : t5_bwd_ring(32))) * sizeof(u64), s)) return rc;
    const int plain_ok = !(a.flags & DC_DIMS_TEAM_DEVICE_SCOPE);
    const dim3 grid(nt * T5_M), block(T5_THREADS);
    if (ns == 16 && a.bf16_store) hipLaunchKernelGGL((lstm512_team_bwd_kernel<16, true>), grid, block, t5_bwd_lds(16), s, a, a.WhhT_bf, xb, nt, plain_ok);
    else if (ns == 16) hipLaunchKernelGGL((lstm512_team_bwd_kernel<16, false>), grid, block, t5_bwd_lds(16), s, a, a.WhhT_bf, xb, nt, plain_ok);
    else if (a.bf16_store) hipLaunchKernelGGL((lstm512_team_bwd_kernel<32, true>), grid, block, t5_bwd_lds(32), s, a, a.WhhT_bf, xb, nt, plain_ok);
    else hipLaunchKernelGGL((lstm512_team_bwd_kernel<32, false>), grid, block, t5_bwd_lds(32), s, a, a.WhhT_bf, xb, nt, plain_ok);
#if DC_DEV_TIMING
    {
        long long h[8];
        (void)hipMemcpy(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost);
        const double n = (double)max_len * (((a.n_seq + ns - 1) / ns + nt - 1) / nt);
        fprintf(stderr, "lstm512_team_bwd timing (clocks per step, %.0f steps, NS %d): copies+product %.0f  publish %.0f  barrier %.0f  poll+sum %.0f  "
                        "prefetch+cells+stores %.0f  barrier %.0f  loop %.0f\n", n, ns, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[6] / n, h[5] / n);
    }
#endif
    return launch_check("lstm_team512_backward");
}

}  // namespace dc
