// Exchange protocol of the H = 256 team kernels (rnn_team.hip, rnn_team_mfma.hip): 8-byte {value, tag} granules, ticketed
// roles, the same-XCD handshake.  See the header comment of rnn_team.hip.
#pragma once
#include "kernels.h"

namespace dc {
namespace {

enum { TEAM_H = 256, TEAM_US = 64, TEAM_M = 4, TEAM_SLOTS = 4, TEAM_MAX = 64, TEAM_NS_MAX = 4 };
enum { CELL_GRU = 0, CELL_LSTM = 1 };
constexpr int SPIN_LIMIT = 1 << 21;   // polls (~0.5-1 us each) before a member gives up

typedef unsigned long long u64;

__device__ __forceinline__ u64 granule_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void granule_store(u64* p, float v, unsigned tag, int plain = 0) {
    const u64 g = ((u64)tag << 32) | (u64)__float_as_uint(v);
    // plain: a store that stops in the XCD's L2 instead of writing through to memory - valid (and ~0.15 us per hand-off
    // faster) when all four members of the team run on the same XCD, which they check at start (team_same_xcd)
    if (plain) __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// spins until the granule carries `tag`; false on timeout.  g = a first read of the granule (possibly issued long ago)
__device__ __forceinline__ bool granule_wait(u64 g, const u64* p, unsigned tag, float& v) {
    int n = 0;
    while ((unsigned)(g >> 32) != tag) {
        if (++n > SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(1);
        g = granule_load(p);
    }
    v = __uint_as_float((unsigned)g);
    return true;
}

// N granules at once: every granule whose tag is still missing is re-read in the SAME round, so once the data is there the wait ends
// one L2 round trip later - waited for one after the other, each stale first read paid a round trip of its own (rnn_team512.hip measured
// that on 30 granules per thread).  g = first reads (possibly issued long ago); false on timeout.
// The usual case - every lane's first reads already carry the tag - leaves through one ballot and a wave-uniform branch: the loop's
// lane-divergent exit and re-reads cost ~450 cycles of exec-mask bookkeeping per call even when nothing is missing (round 6, TM_TIMING).
template <int N>
__device__ __forceinline__ bool granule_wait_all(u64 (&g)[N], const u64* const (&p)[N], unsigned tag) {
    {
        bool missing = false;
#pragma unroll
        for (int i = 0; i < N; ++i) missing = missing || (unsigned)(g[i] >> 32) != tag;
        if (__builtin_amdgcn_ballot_w64(missing) == 0) return true;
    }
    for (int n = 0;; ++n) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i) ok = ok && (unsigned)(g[i] >> 32) == tag;
        if (ok) return true;
        if (n > SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int i = 0; i < N; ++i)
            if ((unsigned)(g[i] >> 32) != tag) g[i] = granule_load(p[i]);
    }
}

// A wait timed out: first writer wins, the record stays until the workspace's owner clears it (include/dotaclient_hip.h, DC_WS_FAULT).
enum { TEAM_K_VALU_FWD = 1, TEAM_K_VALU_BWD = 2, TEAM_K_MFMA_FWD = 3, TEAM_K_MFMA_BWD = 4, TEAM_K_T8_FWD = 7, TEAM_K_T8_BWD = 8 };   // (5, 6: rnn_team512.hip)
__device__ __noinline__ void team_report_timeout(int* fault, int kernel_id, int layer, int team, int member, int step, int seq, unsigned tag) {
    if (fault == nullptr) return;
    if (atomicCAS(&fault[0], 0, DC_FAULT_TEAM_TIMEOUT + kernel_id) == 0) {
        fault[1] = layer; fault[2] = team; fault[3] = member; fault[4] = step; fault[5] = seq; fault[6] = (int)tag;
        __threadfence();
    }
}

// Which (team, member) a workgroup plays is decided when it STARTS RUNNING, by a ticket, not by its block index: the
// first four workgroups to start form team 0, the next four team 1, ...  A workgroup that has started stays resident
// until it exits, so every team whose four tickets are taken is fully resident and makes progress - whatever else
// occupies the chip (an RCCL kernel, a second engine's launch on another stream, a CU mask): workgroups that have not
// been dispatched yet hold no role anyone waits for, at most one team per ticket counter is incomplete at any moment,
// and it completes as soon as any running workgroup on the chip exits.  No co-residency of the whole grid is assumed.
// Speed only: with a multiple of 8 teams there is one ticket counter per XCD (the workgroup reads its XCC id), so a
// team's members share an L2 under any placement; a workgroup whose XCD has no role left takes one of another XCD.
enum { TEAM_HDR = 16 };   // u64 words in front of the handshake granules: ticket counters (u32 each)
template <int M>     // members per team: 4 (rnn_team.hip, rnn_team_mfma.hip) or 8 (rnn_team8.hip)
__device__ __forceinline__ void team_claim_role_m(unsigned* claim, int n_teams, int& team, int& member) {
    __shared__ int role_sh[2];
    if (threadIdx.x == 0) {
        int t = -1, m = 0;
        if ((n_teams & 7) == 0) {
            const int quota = (n_teams >> 3) * M;        // roles per XCD slice: teams x, x + 8, x + 16, ...
            const int x = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7);   // HW_REG_XCC_ID
            for (int k = 0; k < 8 && t < 0; ++k) {
                const int y = (x + k) & 7;
                const unsigned o = atomicAdd(&claim[y], 1u);
                if ((int)o < quota) { t = (int)(o / M) * 8 + y; m = (int)(o % M); }
            }
        } else {
            const unsigned o = atomicAdd(&claim[0], 1u);
            if ((int)o < n_teams * M) { t = (int)(o / M); m = (int)(o % M); }
        }
        role_sh[0] = t; role_sh[1] = m;
    }
    __syncthreads();
    team = __builtin_amdgcn_readfirstlane(role_sh[0]);      // uniform values: what is derived from them stays on the scalar unit
    member = __builtin_amdgcn_readfirstlane(role_sh[1]);
}
__device__ __forceinline__ void team_claim_role(unsigned* claim, int n_teams, int& team, int& member) {
    team_claim_role_m<TEAM_M>(claim, n_teams, team, member);
}

// Once per launch: do the four members of this team share an XCD (hence an L2)?  Each publishes its XCC id as a granule
// (device scope, like the ring) and reads the three others'.  All members see the same four ids, so they agree on the answer;
// a member that cannot read a peer answers "no" (the ring will then time out and poison the outputs anyway).
enum { TEAM_HS_TAG = 0xFFFFFFFFu };
template <int M>
__device__ __forceinline__ int team_same_xcd_m(u64* hs, int member, int allow) {
    __shared__ int same_sh;
    if (threadIdx.x == 0) {
        const unsigned my = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID[3:0]
        granule_store(hs + member, __uint_as_float(my), TEAM_HS_TAG);
        bool same = allow != 0;
        for (int m = 0; m < M; ++m) {
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
__device__ __forceinline__ int team_same_xcd(u64* hs, int member, int allow) { return team_same_xcd_m<TEAM_M>(hs, member, allow); }

// The activation streams of the team kernels (gate rows, states, gradients: each byte touched once per launch) are NON-TEMPORAL
// accesses: as ordinary ones they pass through - and evict from - the L2 the team's granules live in (measured round 5: LSTM-512 backward
// 2 216 -> 1 917 us per pass, LSTM-256 backward 503 -> 479).  -DTM_NT=0: ordinary accesses (A/B).
#ifndef TM_NT
#define TM_NT 1
#endif
__device__ __forceinline__ float tm_ld(const float* p) {
#if TM_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void tm_st(float* p, float v) {
#if TM_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

}  // namespace
}  // namespace dc
