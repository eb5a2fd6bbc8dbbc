// Action heads: target-unit attention, masked log-softmax, PPO clipped-ratio + value + entropy loss.
//
// Replaces /root/reference/policy.py:152 (query . unit_embedding), policy.py:169-178
// (masked_softmax), /root/reference/optimizer.py:387-390 (old log-probs of the selected actions),
// optimizer.py:587-665 (advantage normalisation + loss) and the backward seeds autograd derives
// from them (optimizer.py:672).
//
// Per env-step record: headout[n][160] (cols: 0..127 query, 128..131 enum, 132..140 x, 141..149 y,
// 150..152 ability, 153 value), tu[n][40] target-unit logits; act/mask uint8[n][65] with heads in
// order enum(4) x(9) y(9) target_unit(40) ability(3).
#include "kernels.h"

namespace dc {

enum { HO_LD = 160, HO_ENUM = 128, HO_X = 132, HO_Y = 141, HO_ABILITY = 150, HO_VALUE = 153, ACT = 65, NUNITS = 40,
       EMBW = 128 };
__constant__ int c_head_off[6] = {0, 4, 13, 22, 62, 65};   // offsets inside the 65-wide act/mask row
__constant__ int c_t_units[6] = {1, 5, 16, 16, 1, 1};
__constant__ int c_t_cum[7] = {0, 1, 6, 22, 38, 39, 40};

// stats block (doubles), shared by the loss kernels and the finaliser
enum { ST_ADV_SUM = 0, ST_ADV_SQ = 1, ST_NSEL = 2 /*5*/, ST_POL = 7 /*5*/, ST_ENT = 12 /*5*/, ST_VAL = 17, ST_COUNT = 18 };
// Layout of DC_WS_STATS (doubles; policy.hip sizes it): [0, 64) the totals above (written by the last block of ppo_loss_kernel,
// for inspection), [ST_PART1, +ST_G1 * 8) batch_stats_kernel's per-block partial sums, [ST_PART2, +ST_G2 * 12) ppo_loss_kernel's,
// [ST_TICKET] a 32-bit arrival counter.  Per-block partials summed in a fixed order instead of f64 atomics on eleven words:
// 4 096 blocks x 11 same-address atomics serialised at one L2 channel (the kernel's tail), and made the sums order-dependent.
#ifndef DC_LOSS_GRID
#define DC_LOSS_GRID 1024
#endif
enum { ST_G1 = 256, ST_G2 = DC_LOSS_GRID,      // (policy.hip sizes DC_WS_STATS for up to 4 096 blocks)
       ST_PART1 = 64, ST_PART2 = ST_PART1 + ST_G1 * 8, ST_TICKET = ST_PART2 + ST_G2 * 12, ST_L1 = 32, ST_GPART = ST_TICKET + 8 + ST_L1, ST_DOUBLES = ST_GPART + ST_L1 * 12 };      // [ST_TICKET]: second-level counter, [ST_TICKET + 8 ..): ST_L1 first-level ones, [ST_GPART ..): the groups' rows

// sixteen-lane (DPP row) all-reduces
template <int CTRL>
__device__ __forceinline__ float row_dpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int row_dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ float row_sum16(float x) {      // all 16 lanes of the row get the sum
    x += row_dpp<0x128>(x); x += row_dpp<0x124>(x); x += row_dpp<0x122>(x); x += row_dpp<0x121>(x);   // row_ror 8,4,2,1
    return x;
}
__device__ __forceinline__ int row_sum16_i(int x) {
    x += row_dpp_i<0x128>(x); x += row_dpp_i<0x124>(x); x += row_dpp_i<0x122>(x); x += row_dpp_i<0x121>(x);
    return x;
}
__device__ __forceinline__ int row_min16_i(int x) {
    x = min(x, row_dpp_i<0x128>(x)); x = min(x, row_dpp_i<0x124>(x)); x = min(x, row_dpp_i<0x122>(x)); x = min(x, row_dpp_i<0x121>(x));
    return x;
}

// ---------------------------------------------------------------------------------------------------
// target-unit logits: tu[n][u] = sum_c q[n][c] * emb[n][u][c];  16 lanes per unit, 4 units per wave pass
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_logits_kernel(const float* __restrict__ headout, const float* __restrict__ emb,
                                                          float* __restrict__ tu, long long nr, long long nrp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 4, l16 = lane & 15;
    for (long long n = (long long)blockIdx.x * 4 + wave; n < nr; n += (long long)gridDim.x * 4) {
        const float4* qp = reinterpret_cast<const float4*>(headout + n * HO_LD);
        const float4 q0 = qp[l16], q1 = qp[16 + l16];
#pragma unroll
        for (int u0 = 0; u0 < NUNITS; u0 += 4) {
            const int u = u0 + sub;
            int t = 0;
#pragma unroll
            for (int i = 1; i < 6; ++i) if (u >= c_t_cum[i]) t = i;
            const float4* ep = reinterpret_cast<const float4*>(
                emb + (nrp * c_t_cum[t] + n * c_t_units[t] + (u - c_t_cum[t])) * EMBW);
            const float4 e0 = ep[l16], e1 = ep[16 + l16];
            float s = q0.x * e0.x + q0.y * e0.y + q0.z * e0.z + q0.w * e0.w + q1.x * e1.x + q1.y * e1.y + q1.z * e1.z +
                      q1.w * e1.w;
            s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
            if (l16 == 0) tu[n * NUNITS + u] = s;
        }
    }
}

// type-major row of unit u (0..39) of env-step n.  Selects, not the c_t_* tables: a table entry at a run-time index is a memory
// round trip (scalar or vector) in front of every embedding-row address.
__device__ __forceinline__ long long unit_row(long long nr, long long n, int u) {
    int cum = 0, un = 1;                              // types: 1, 5, 16, 16, 1, 1 units; first units 0, 1, 6, 22, 38, 39
    if (u >= 1) { cum = 1; un = 5; }
    if (u >= 6) { cum = 6; un = 16; }
    if (u >= 22) cum = 22;
    if (u >= 38) { cum = 38; un = 1; }
    if (u >= 39) cum = 39;
    return nr * cum + n * un + (u - cum);
}

// Mask-aware form (DC_DIMS_LAZY_TU): only units whose target_unit mask byte is set are read - the actors' masks
// (agent.py:666-671) leave the head empty unless the step's action type targets a unit, so most steps cost 40
// bytes instead of 20 KB.  One wave per env-step: the 40 mask bytes become a ballot, each 16-lane quarter of the
// wave takes every fourth set bit.
__global__ __launch_bounds__(256) void attn_logits_masked_kernel(const float* __restrict__ headout, const float* __restrict__ emb,
                                                                 const uint8_t* __restrict__ mask, float* __restrict__ tu,
                                                                 long long nr, long long nrp) {
    // A live step has ~24 units: taken four per trip (one per quarter) that was six dependent memory round trips per step.  Now the set
    // lanes leave their unit numbers in a wave-private LDS list (rank by v_mbcnt), quarter `sub` takes entries sub, sub + 4, ... and ALL
    // of its (up to ten) rows are requested before the first dot product.
    __shared__ uint8_t list_s[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 4, l16 = lane & 15;
    uint8_t* const list = list_s[wave];
    for (long long n = (long long)blockIdx.x * 4 + wave; n < nr; n += (long long)gridDim.x * 4) {
        const bool set = lane < NUNITS && mask[n * ACT + 22 + lane] != 0;
        const unsigned long long bits = __ballot(set);
        if (lane < NUNITS && !set) tu[n * NUNITS + lane] = 0.f;
        if (bits == 0ull) continue;
        const int cnt = __popcll(bits);
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bits >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bits, 0u));
        if (set) list[rank] = (uint8_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // LDS is in order per wave; this keeps the compiler from moving the reads up
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float4* qp = reinterpret_cast<const float4*>(headout + n * HO_LD);
        const float4 q0 = qp[l16], q1 = qp[16 + l16];
        // two chunks of five entries per quarter (20 units each; the second only when the step has more).  No branch around a load:
        // an entry past the end re-reads the last unit's row (an L1 hit) and is dropped - a conditional load makes the compiler wait
        // for it at the end of its block, which is the serial chain again.
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            if (20 * ch < cnt) {                                      // wave-uniform
                float4 e0[5], e1[5];
                int u[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    u[i] = (int)list[min(20 * ch + 4 * i + sub, cnt - 1)];
                    const float4* ep = reinterpret_cast<const float4*>(emb + unit_row(nrp, n, u[i]) * EMBW);
                    e0[i] = ep[l16]; e1[i] = ep[16 + l16];
                }
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    float s = q0.x * e0[i].x + q0.y * e0[i].y + q0.z * e0[i].z + q0.w * e0[i].w + q1.x * e1[i].x + q1.y * e1[i].y +
                              q1.z * e1[i].z + q1.w * e1[i].w;
                    s = row_sum16(s);                                 // (the same association as the xor butterfly it replaces)
                    if (l16 == 0 && 20 * ch + 4 * i + sub < cnt) tu[n * NUNITS + u[i]] = s;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the list is rewritten by the wave's next step
        __builtin_amdgcn_wave_barrier();
    }
}

// dq[n][c] = sum_u dtu[n][u] * emb[n][u][c]  -> dheadout[n][0..127]; 128 threads (two waves) per env-step.
// Only units with a non-zero gradient are read: dtu is zero outside the masked-in units of the steps whose
// target_unit head is live (most steps have none and cost 160 bytes instead of 20 KB).
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float* __restrict__ dtu, const float* __restrict__ emb,
                                                         float* __restrict__ dheadout, long long nr, long long nrp) {
    const int c = threadIdx.x & 127, sub = threadIdx.x >> 7, lane = threadIdx.x & 63;
    for (long long n = (long long)blockIdx.x * 2 + sub; n < nr; n += (long long)gridDim.x * 2) {
        const float* dt = dtu + n * NUNITS;
        const float mine = lane < NUNITS ? dt[lane] : 0.f;
        unsigned long long bits = __ballot(mine != 0.f);        // the same in both waves of the step
        float acc = 0.f;
        while (bits) {      // sixteen units per trip: their rows are in flight together (the loop is a chain of memory round trips;
                            // a live step has ~24 units: two trips, where four per trip made six)
            int u[16];
            float w[16], e[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                u[i] = bits ? __builtin_ctzll(bits) : -1;         // wave-uniform
                bits &= bits - 1;
                w[i] = u[i] >= 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), u[i] < 0 ? 0 : u[i])) : 0.f;
                e[i] = u[i] >= 0 ? emb[unit_row(nrp, n, u[i]) * EMBW + c] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = fmaf(w[i], e[i], acc);   // ascending unit order, as before
        }
        dheadout[n * HO_LD + c] = acc;
    }
}

__device__ __forceinline__ float head_logit(const float* __restrict__ ho, const float* __restrict__ tu, int k, int c) {
    switch (k) {
        case 0: return ho[HO_ENUM + c];
        case 1: return ho[HO_X + c];
        case 2: return ho[HO_Y + c];
        case 3: return tu[c];
        default: return ho[HO_ABILITY + c];
    }
}
__device__ __forceinline__ int head_grad_col(int k, int c) {
    switch (k) {
        case 0: return HO_ENUM + c;
        case 1: return HO_X + c;
        case 2: return HO_Y + c;
        case 3: return -1;
        default: return HO_ABILITY + c;
    }
}

// ---------------------------------------------------------------------------------------------------
// rollout-pass epilogue (optimizer.py:387-390): log-prob of the selected action per head (0 where the
// head has no action in that step), the value, and the masked argmax per head (-1 on an empty mask):
// select_logp_kernel, further down (next to ppo_loss_kernel, whose lane layout and helpers it shares).
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// batch statistics needed before the loss: sum / sum-of-squares of the advantages (optimizer.py:588)
// and the number of steps that took an action per head (optimizer.py:626,643)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void batch_stats_kernel(const float* __restrict__ adv, const uint8_t* __restrict__ act,
                                                          double* __restrict__ stats, long long nr) {
    double s = 0.0, sq = 0.0;
    double cnt[5] = {0, 0, 0, 0, 0};
    for (long long n = (long long)blockIdx.x * 256 + threadIdx.x; n < nr; n += (long long)gridDim.x * 256) {
        const double a = adv[n];
        s += a; sq += a * a;
        const uint8_t* ar = act + n * ACT;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            int any = 0;
            for (int c = c_head_off[k]; c < c_head_off[k + 1]; ++c) any |= ar[c];
            cnt[k] += any ? 1.0 : 0.0;
        }
    }
    __shared__ double sh[4][7];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double v[7] = {s, sq, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4]};
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const double r = wave_sum(v[i]);
        if (lane == 0) sh[wave][i] = r;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const double r = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        stats[ST_PART1 + blockIdx.x * 8 + threadIdx.x] = r;   // ST_ADV_SUM, ST_ADV_SQ, ST_NSEL..: summed by ppo_loss_kernel's blocks
    }
    if (blockIdx.x == 0 && threadIdx.x <= ST_L1)            // ppo_loss_kernel's arrival counters (second level, then the ST_L1 first-level ones)
        *reinterpret_cast<unsigned*>(stats + ST_TICKET + (threadIdx.x == 0 ? 0 : 7 + threadIdx.x)) = 0u;
}

struct LossArgs {
    const float* headout; const float* tu;
    const uint8_t* act; const uint8_t* mask;
    const float* old_logp;   // [nr][5]
    const float* adv;        // [nr] raw advantages
    const float* ret;        // [nr]
    double* stats;
    float* dheadout;         // [nr][160] cols 128..153 written here (0..127 by attn_bwd_q)
    float* dtu;              // [nr][40]
    long long nr;
    float e_clip, entropy_coef, vf_coef, adv_eps;
    int g1;                  // blocks of batch_stats_kernel (rows of its partial sums)
    float* losses_out; int32_t* head_on;    // written by the last block to finish (loss_finalize)
};

// 16 lanes (one DPP row) per env-step: lane j owns the act/mask columns j, j+16, j+32, j+48 (and lane 0 column
// 64) of the step's 65, whatever head they belong to; per-head soft-max sums, entropies, the selected column and
// its gradient seed are row all-reduces (four v_add_f32_dpp / v_min_u32_dpp each).  The previous one-thread-per-
// (step, head) form walked up to 40 logits three times in a serial loop while the other heads' lanes idled
// (66 us per launch; this one is bound by its ~1 KB per step of traffic).  Same arithmetic per element:
// no max-subtraction (policy.py:172), tie/clamp gradient rules of torch.min / clamp, batch sums in f64.
__device__ __forceinline__ int head_of_col(int c) { return c < 4 ? 0 : (c < 13 ? 1 : (c < 22 ? 2 : (c < 62 ? 3 : 4))); }
// Slot m of a lane is column j + 16 m: columns 0..15 belong to heads 0, 1, 2 only, 16..31 to heads 2, 3, 32..47 to head 3, 48..63 to heads 3, 4, 64 to
// head 4.  The loops over slots are unrolled, so these fold to constants and the per-head select chains below keep 9 of their 25 links (the
// kernel is bound by its instruction count: ~1 800 per 16-lane group and step before).
__device__ __forceinline__ constexpr bool slot_has(int m, int kk) {
    return m == 0 ? kk <= 2 : (m == 1 ? (kk == 2 || kk == 3) : (m == 2 ? kk == 3 : (m == 3 ? (kk == 3 || kk == 4) : kk == 4)));
}
__device__ __forceinline__ constexpr int slot_top(int m) { return m == 0 ? 2 : (m <= 2 ? 3 : 4); }      // highest head a slot can hold: the chain's default
template <class T>
__device__ __forceinline__ T pick_head(int m, int k, const T (&a)[5]) {
    T r = a[slot_top(m)];
#pragma unroll
    for (int kk = 3; kk >= 0; --kk)
        if (slot_has(m, kk) && kk != slot_top(m)) r = k == kk ? a[kk] : r;
    return r;
}

// rollout-pass epilogue (optimizer.py:387-390; see the banner further up): 16 lanes per env-step like ppo_loss_kernel - lane j owns columns
// j, j + 16, .. of the 65 - instead of one thread per (step, head) walking up to 40 logits in a serial loop behind dependent loads
// (37-41 us per launch for 24 MB).  Same arithmetic per element (no max-subtraction, policy.py:172); the soft-max sum is the row
// all-reduce ppo_loss_kernel takes too, so the first epoch's ratio starts from the very same log-prob.
__global__ __launch_bounds__(256) void select_logp_kernel(const float* __restrict__ headout, const float* __restrict__ tu,
                                                          const uint8_t* __restrict__ act, const uint8_t* __restrict__ mask,
                                                          float* __restrict__ logp_sel, float* __restrict__ values,
                                                          int32_t* __restrict__ argmax, long long nr) {
    const int j = threadIdx.x & 15;
    const long long n = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool on = n < nr;
    const long long nn = on ? n : nr - 1;           // idle rows of the last block recompute a valid step and store nothing
    const float* ho = headout + nn * HO_LD;
    const float* tun = tu + nn * NUNITS;
    const uint8_t* mrow = mask + nn * ACT;
    const uint8_t* arow = act + nn * ACT;
    float z[5];
    bool mk[5];
    float se_c[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, mx_c[5] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int amin_c[5] = {127, 127, 127, 127, 127};
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const int c = j + 16 * m;
        const bool valid = c < ACT;
        const int cc = valid ? c : 0;
        const int k = head_of_col(cc);
        z[m] = k == 3 ? tun[cc - 22] : ho[(k == 4 ? 88 : 128) + cc];
        mk[m] = valid && mrow[cc] != 0;
        const bool ac = valid && arow[cc] != 0;
        const float e = mk[m] ? expf(z[m]) : 0.f;
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
            if (!slot_has(m, kk)) continue;
            const bool mine = valid && k == kk;
            se_c[kk] += mine ? e : 0.f;
            amin_c[kk] = (mine && ac) ? min(amin_c[kk], c) : amin_c[kk];
            mx_c[kk] = (mine && mk[m]) ? fmaxf(mx_c[kk], z[m]) : mx_c[kk];      // (a NaN logit is never the maximum: like `z > best`)
        }
    }
    float lse[5], mx[5];
    int amin[5];
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
        lse[kk] = logf(row_sum16(se_c[kk]));
        amin[kk] = row_min16_i(amin_c[kk]);
        float v = mx_c[kk];
        v = fmaxf(v, row_dpp<0x128>(v)); v = fmaxf(v, row_dpp<0x124>(v)); v = fmaxf(v, row_dpp<0x122>(v)); v = fmaxf(v, row_dpp<0x121>(v));
        mx[kk] = v;
    }
    // the selected action's logit (one owner lane per head) and the first column that holds the masked maximum
    float zsel_c[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    int best_c[5] = {127, 127, 127, 127, 127};
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const int c = j + 16 * m;
        const bool valid = c < ACT;
        const int k = head_of_col(valid ? c : 0);
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
            if (!slot_has(m, kk)) continue;
            const bool mine = valid && k == kk;
            zsel_c[kk] += (mine && c == amin[kk]) ? z[m] : 0.f;
            best_c[kk] = (mine && mk[m] && z[m] == mx[kk] && mx[kk] > -INFINITY) ? min(best_c[kk], c) : best_c[kk];
        }
    }
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
        const float zs = row_sum16(zsel_c[kk]);               // exact: one non-zero lane
        const int bc = row_min16_i(best_c[kk]);
        if (on && j == kk) {
            logp_sel[n * 5 + kk] = amin[kk] < 127 ? zs - lse[kk] : 0.f;
            if (argmax) argmax[n * 5 + kk] = bc < 127 ? bc - c_head_off[kk] : -1;
        }
    }
    if (on && j == 5) values[n] = ho[HO_VALUE];
}

// losses[0..3] = loss, policy_loss, entropy_loss, value_loss ; losses[4..8] = entropies per head
// (optimizer.py:649-665, 682-689); flags[0..4] = 1 if head k had at least one action in the batch.  One thread.
__device__ __forceinline__ void loss_finalize(const double* stats, float* out, int32_t* head_on, long long nr, float entropy_coef,
                                              float vf_coef) {
    double pol_sum = 0.0, ent_sum = 0.0;
    for (int k = 0; k < 5; ++k) {
        const double nsel = stats[ST_NSEL + k];
        // fp32 like the reference's 0-d tensors: mean of the per-step terms, then mean over the 5 heads
        const float lk = nsel > 0.0 ? (float)(stats[ST_POL + k] / nsel) : 0.f;
        const float hk = nsel > 0.0 ? (float)(stats[ST_ENT + k] / nsel) : 0.f;
        pol_sum += (double)lk;
        ent_sum += (double)hk;
        out[4 + k] = hk;
        head_on[k] = nsel > 0.0 ? 1 : 0;
    }
    const float policy_loss = (float)(pol_sum / 5.0);
    const float entropy_loss = entropy_coef > 0.f ? -entropy_coef * (float)ent_sum : 0.f;
    const float value_loss = vf_coef > 0.f ? vf_coef * (0.5f * (float)(stats[ST_VAL] / (double)nr)) : 0.f;
    out[0] = policy_loss + entropy_loss + value_loss;
    out[1] = policy_loss;
    out[2] = entropy_loss;
    out[3] = value_loss;
}

__global__ __launch_bounds__(256) void ppo_loss_kernel(LossArgs p) {
#ifdef PL_TIMING
    long long ts[10]; int nts = 0;
#define PL_STAMP() do { if (threadIdx.x == 0) ts[nts++] = __builtin_readcyclecounter(); } while (0)
#else
#define PL_STAMP() do { } while (0)
#endif
    PL_STAMP();
    __shared__ float sh_norm[2];
    __shared__ double sh_tot[ST_COUNT];
    __shared__ double sh[4][11];
    __shared__ double accs[16][12];      // running sums of the block: [row of the group][5 policy, 5 entropy, value]; the row's lane 0 owns its entry
    __shared__ double sh_inv[7];         // loop invariants, one f64 division each per BLOCK: 1 / (5 nsel_k), 1 / sd, 1 / N
    __shared__ float sh_gent[5], sh_nsel[5];
    __shared__ int sh_last;
    // batch statistics: every block sums batch_stats_kernel's partial rows itself, in one fixed order (lane-strided, then the wave
    // butterfly): bit-identical in every block and from run to run
    if (threadIdx.x < 64) {
        double v[7] = {0, 0, 0, 0, 0, 0, 0};
        // (all ST_G1 / 64 x 7 loads requested before the first add: a `for (b < g1)` loop was four dependent round trips in front of every block)
        double ld[ST_G1 / 64][7];
#pragma unroll
        for (int i = 0; i < ST_G1 / 64; ++i) {
            const int b = min(threadIdx.x + 64 * i, p.g1 - 1);
#pragma unroll
            for (int q = 0; q < 7; ++q) ld[i][q] = p.stats[ST_PART1 + b * 8 + q];
        }
#pragma unroll
        for (int i = 0; i < ST_G1 / 64; ++i)
#pragma unroll
            for (int q = 0; q < 7; ++q) v[q] += (int)threadIdx.x + 64 * i < p.g1 ? ld[i][q] : 0.0;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const double r = wave_sum(v[q]);
            if (threadIdx.x == 0) sh_tot[q] = r;
        }
        if (threadIdx.x == 0) {
            const double N = (double)p.nr;
            const double mean = sh_tot[ST_ADV_SUM] / N;
            double var = (sh_tot[ST_ADV_SQ] - N * mean * mean) / (N - 1.0);   // torch.std: unbiased (N-1); optimizer.py:588
            if (var < 0.0) var = 0.0;
            sh_norm[0] = (float)mean;
            sh_norm[1] = (float)(sqrt(var) + (double)p.adv_eps);
            sh_inv[5] = 1.0 / (double)sh_norm[1];
            sh_inv[6] = 1.0 / N;
#pragma unroll
            for (int kk = 0; kk < 5; ++kk) {
                const float ns = (float)sh_tot[ST_NSEL + kk];
                sh_nsel[kk] = ns;
                sh_inv[kk] = ns > 0.f ? 1.0 / (5.0 * (double)ns) : 0.0;
                sh_gent[kk] = (ns > 0.f && p.entropy_coef > 0.f) ? (float)((double)p.entropy_coef / (double)ns) : 0.f;
            }
        }
    }
    if (threadIdx.x < 192) (&accs[0][0])[threadIdx.x] = 0.0;
    __syncthreads();
    PL_STAMP();      // 1: prologue
    const int j = threadIdx.x & 15;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long groups = (p.nr + 15) / 16;
#pragma unroll 1
    for (int grp_i = blockIdx.x; grp_i < (int)groups; grp_i += gridDim.x) {      // persistent blocks: one set of partial sums per block
    const long long grp = __builtin_amdgcn_readfirstlane(grp_i);
    // this group's terms go into the block's running sums in LDS (accs): every one of them has ONE contributing lane per row (the
    // selected action's owner / the row's lane 0), so a row's lane 0 adds them to its own entry - no cross-lane f64 reduction per group
    // (eleven ds_bpermute wave sums per group were a fifth of the kernel's instructions), and nothing carried in registers across the loop.
    float polc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const long long n = grp * 16 + (threadIdx.x >> 4);
    const bool on = n < p.nr;
    const long long nn = on ? n : p.nr - 1;     // idle rows of the last group recompute a valid step and store nothing

    const float* ho = p.headout + nn * HO_LD;
    const float* tun = p.tu + nn * NUNITS;
    const uint8_t* mrow = p.mask + nn * ACT;
    const uint8_t* arow = p.act + nn * ACT;
    // (double) like the reference's fp32 tensor op on a float64-free path: A is an fp32 value
    const float A = (float)(((double)p.adv[nn] - (double)sh_norm[0]) * sh_inv[5]);

    float z[5], e[5];
    bool mk[5];
    int cand[5];
    int cnt_packed = 0;
    float se_c[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    int amin_c[5] = {127, 127, 127, 127, 127};
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const int c = j + 16 * m;
        const bool valid = c < ACT;
        const int cc = valid ? c : 0;
        const int k = head_of_col(cc);
        z[m] = k == 3 ? tun[cc - 22] : ho[(k == 4 ? 88 : 128) + cc];
        mk[m] = valid && mrow[cc] != 0;
        const bool ac = valid && arow[cc] != 0;
        e[m] = mk[m] ? expf(z[m]) : 0.f;
        cand[m] = ac ? c : 127;
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
            if (!slot_has(m, kk)) continue;
            se_c[kk] += (valid && k == kk) ? e[m] : 0.f;
            amin_c[kk] = (valid && k == kk) ? min(amin_c[kk], cand[m]) : amin_c[kk];
        }
        cnt_packed += mk[m] ? (1 << (6 * k)) : 0;
    }
    float lse[5], nselv[5];
    int amin[5];
    const int cnt = row_sum16_i(cnt_packed);
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
        lse[kk] = logf(row_sum16(se_c[kk]));
        amin[kk] = row_min16_i(amin_c[kk]);
        nselv[kk] = sh_nsel[kk];
    }
    // log-probs of this lane's columns, entropy terms, the selected action's surrogate
    float lp[5], pc[5];
    float h_c[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, glp_c[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const int c = j + 16 * m;
        const bool valid = c < ACT;
        const int k = head_of_col(valid ? c : 0);
        const float lsek = pick_head(m, k, lse);
        lp[m] = z[m] - lsek;
        pc[m] = mk[m] ? expf(lp[m]) : 0.f;
        const float ht = mk[m] ? pc[m] * lp[m] : 0.f;
#pragma unroll
        for (int kk = 0; kk < 5; ++kk)
            if (slot_has(m, kk)) h_c[kk] -= (valid && k == kk) ? ht : 0.f;
        // owner of the head's selected action (lowest set column)
        const int amk = pick_head(m, k, amin);
        const float nsk = pick_head(m, k, nselv);
        if (valid && c == amk && nsk > 0.f) {     // at most one column per head in the whole row
            const float ratio = expf(lp[m] - p.old_logp[nn * 5 + k]);
            const float s1 = ratio * A;
            const float rc = fminf(fmaxf(ratio, 1.f - p.e_clip), 1.f + p.e_clip);
            const float s2 = rc * A;
            const float polv = -fminf(s1, s2);
            // d min(s1,s2)/d ratio: torch splits ties evenly; clamp passes gradient inside [lo,hi]
            const bool inr = (ratio >= 1.f - p.e_clip) && (ratio <= 1.f + p.e_clip);
            float w1, w2;
            if (s1 < s2) { w1 = 1.f; w2 = 0.f; } else if (s1 > s2) { w1 = 0.f; w2 = 1.f; } else { w1 = 0.5f; w2 = 0.5f; }
            const float dmin_dr = A * (w1 + (inr ? w2 : 0.f));
            const double i5v[5] = {sh_inv[0], sh_inv[1], sh_inv[2], sh_inv[3], sh_inv[4]};
            const double i5n = pick_head(m, k, i5v);
            const float g = (float)(-(double)dmin_dr * (double)ratio * i5n);
#pragma unroll
            for (int kk = 0; kk < 5; ++kk) {
                if (!slot_has(m, kk)) continue;
                glp_c[kk] += k == kk ? g : 0.f;
                polc[kk] += k == kk ? polv : 0.f;
            }
        }
    }
    float Hrow[5], glp[5], gent[5];
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
        Hrow[kk] = row_sum16(h_c[kk]);
        glp[kk] = row_sum16(glp_c[kk]);
        const bool many = ((cnt >> (6 * kk)) & 63) != 0;
        if (!many) Hrow[kk] = 0.f;
        gent[kk] = sh_gent[kk];
        const float polr = row_sum16(polc[kk]);                      // exact: one non-zero lane per head and row
        if (on && j == 0) {
            double* const a = &accs[threadIdx.x >> 4][0];
            a[kk] += (double)polr;
            if (nselv[kk] > 0.f && many) a[5 + kk] += (double)Hrow[kk];
        }
    }
    // d loss / d logit_c = g_lp * (delta_{c,a} - [mask_c] p_c) + g_ent * [mask_c] p_c (logp_c + Hrow)
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const int c = j + 16 * m;
        if (c < ACT && on) {
            const int k = head_of_col(c);
            const float gl = pick_head(m, k, glp), ge = pick_head(m, k, gent), hr = pick_head(m, k, Hrow);
            const int amk = pick_head(m, k, amin);
            float g = mk[m] ? (-gl * pc[m] + ge * pc[m] * (lp[m] + hr)) : 0.f;
            // (k == 3: an action on a masked-out unit - the actors never send one, agent.py:666-671 - gets no gradient, so that d(tu) is
            // exactly zero outside the mask: the backward may then rely on "dtu != 0 => the unit's embedding row was stored")
            if (c == amk && (k != 3 || mk[m])) g += gl;
            if (k == 3) p.dtu[n * NUNITS + (c - 22)] = g;
            else p.dheadout[n * HO_LD + (k == 4 ? 88 : 128) + c] = g;
        }
    }
    if (on && j == 0) {
        const float v = ho[HO_VALUE];
        const float d = p.ret[n] - v;
        accs[threadIdx.x >> 4][10] += (double)d * (double)d;
        // value_loss = vf_coef * 0.5 * mean((R - V)^2)  (optimizer.py:658-661)
        p.dheadout[n * HO_LD + HO_VALUE] = (p.vf_coef > 0.f) ? (float)((double)p.vf_coef * (double)(v - p.ret[n]) * sh_inv[6]) : 0.f;
#pragma unroll
        for (int c = HO_VALUE + 1; c < HO_LD; ++c) p.dheadout[n * HO_LD + c] = 0.f;
    }
    }   // groups of this block
    __syncthreads();
    PL_STAMP();      // 2: body
    // The last block to arrive sums all rows (fixed order) and finalises the losses.  Hand-off without fences (an agent-scope release
    // writes back every dirty line of the XCD's L2 - here the d(headout) rows the whole chip has just written - once per block):
    // the row goes out as 8-byte write-through (sc1) stores, drained (vmcnt(0)) before the ticket is taken; the last block
    // reads the rows with sc1 loads (MI355X_MICROARCH.md: "8-B agent atomics both sides").
    if (threadIdx.x < 11) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += accs[r][threadIdx.x];     // fixed order
        __hip_atomic_store(&p.stats[ST_PART2 + blockIdx.x * 12 + threadIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // Two levels, tickets and sums alike: block b belongs to group b mod ST_L1; the last block of a group to arrive adds the group's rows
    // (fixed order) into ONE group row, then takes a second-level ticket; the last group adds the ST_L1 group rows and finalises.  A thousand
    // blocks finishing together queued at one L2 address for ~17 us, and the one last block then walked 1 024 x 11 partial sums through
    // eleven f64 wave reductions: 17 us more (PL_TIMING) - now the groups reduce side by side and the tail holds 32 x 11 values.
    __shared__ double sh_rows[ST_L1][12];
    const unsigned grp = blockIdx.x % ST_L1, n_grp = (gridDim.x - grp + ST_L1 - 1) / ST_L1;
    if (threadIdx.x == 0)
        sh_last = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(p.stats + ST_TICKET + 8 + grp), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_grp - 1;
    __syncthreads();
    PL_STAMP();      // 3: partials stored + ticket
    if (!sh_last) return;
    for (int e = threadIdx.x; e < (int)n_grp * 11; e += 256) {
        const int i = e / 11, q = e - 11 * i;
        sh_rows[i][q] = __hip_atomic_load(&p.stats[ST_PART2 + (size_t)(grp + ST_L1 * i) * 12 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (threadIdx.x < 11) {
        double t = 0.0;
        for (int i = 0; i < (int)n_grp; ++i) t += sh_rows[i][threadIdx.x];      // fixed order
        __hip_atomic_store(&p.stats[ST_GPART + grp * 12 + threadIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const unsigned n_top = min((unsigned)gridDim.x, (unsigned)ST_L1);
    if (threadIdx.x == 0)
        sh_last = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(p.stats + ST_TICKET), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_top - 1;
    __syncthreads();
    if (!sh_last) return;
    for (int e = threadIdx.x; e < (int)n_top * 11; e += 256) {
        const int i = e / 11, q = e - 11 * i;
        sh_rows[i][q] = __hip_atomic_load(&p.stats[ST_GPART + i * 12 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (threadIdx.x < 11) {
        double t = 0.0;
        for (int i = 0; i < (int)n_top; ++i) t += sh_rows[i][threadIdx.x];
        p.stats[ST_POL + threadIdx.x] = t;
    }
    if (threadIdx.x >= 64 && threadIdx.x < 71) p.stats[threadIdx.x - 64] = sh_tot[threadIdx.x - 64];
    __syncthreads();
    if (threadIdx.x == 0) {
        loss_finalize(p.stats, p.losses_out, p.head_on, p.nr, p.entropy_coef, p.vf_coef);
#ifdef PL_TIMING
        ts[nts++] = __builtin_readcyclecounter();
        printf("ppo_loss last block %d: prologue %lld body %lld ticket %lld finalised %lld\n", (int)blockIdx.x, ts[1] - ts[0], ts[2] - ts[0], ts[3] - ts[0], ts[4] - ts[0]);
#endif
    }
}

// blocks of the mask-aware attention kernels: a wave per env-step (a capped grid of 4 096 blocks walked four steps per wave one after the
// other - a chain of dependent round trips each: 63 -> 56 us for the logits, profiles/r04: tools/gpu_r4_v30.sh)
#ifndef ATTN_GRID_CAP
#define ATTN_GRID_CAP 65536
#endif
static inline int grid1d(long long items, int per_block, int cap) {
    long long g = (items + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

int attn_logits(const float* headout, const float* emb, float* tu, long long nr, long long nrp, hipStream_t s) {
    ProfScope prof("attn_logits", 2.0 * nr * 40 * 128, 4.0 * nr * (40 * 128 + 128 + 40), s);
    hipLaunchKernelGGL(attn_logits_kernel, dim3(grid1d(nr, 4, 256 * 16)), dim3(256), 0, s, headout, emb, tu, nr, nrp);
    return launch_check("attn_logits");
}

int attn_logits_masked(const float* headout, const float* emb, const uint8_t* mask, float* tu, long long nr, long long nrp, hipStream_t s) {
    ProfScope prof("attn_logits", 2.0 * nr * 40 * 128, 4.0 * nr * (40 * 128 + 128 + 40), s);
    hipLaunchKernelGGL(attn_logits_masked_kernel, dim3(grid1d(nr, 4, ATTN_GRID_CAP)), dim3(256), 0, s, headout, emb, mask, tu, nr, nrp);
    return launch_check("attn_logits_masked");
}

int attn_bwd_q(const float* dtu, const float* emb, float* dheadout, long long nr, long long nrp, hipStream_t s) {
    ProfScope prof("attn_bwd_q", 2.0 * nr * 40 * 128, 4.0 * nr * (40 * 128 + 128 + 40), s);
    hipLaunchKernelGGL(attn_bwd_q_kernel, dim3(grid1d(nr, 2, ATTN_GRID_CAP)), dim3(256), 0, s, dtu, emb, dheadout, nr, nrp);
    return launch_check("attn_bwd_q");
}

int select_logp(const float* headout, const float* tu, const uint8_t* act, const uint8_t* mask, float* logp_sel,
                float* values, int32_t* argmax, long long nr, hipStream_t s) {
    ProfScope prof("select_logp", 0.0, (double)nr * (4.0 * (26 + 40) + 2.0 * 65 + 4.0 * (5 + 1 + 5)), s);
    hipLaunchKernelGGL(select_logp_kernel, dim3((unsigned)((nr + 15) / 16)), dim3(256), 0, s, headout, tu, act, mask,
                       logp_sel, values, argmax, nr);
    return launch_check("select_logp");
}

int ppo_loss_fwd_bwd(const float* headout, const float* tu, const uint8_t* act, const uint8_t* mask, const float* old_logp,
                     const float* adv, const float* ret, double* stats, float* dheadout, float* dtu, float* losses_out,
                     int32_t* head_on, long long nr, float e_clip, float entropy_coef, float vf_coef, hipStream_t s) {
    ProfScope prof("ppo_loss(stats+loss+finalize)", 0.0, (double)nr * (4.0 * (26 + 40 + 5 + 2) + 2.0 * 65 + 4.0 * (32 + 40)), s);
    // two launches: batch statistics (per-block partial sums; also clears the arrival counter), then the loss, whose last block
    // to finish sums the blocks' partial losses and finalises them - no clear launch, no finalise launch, no same-address atomics
    const int g1 = grid1d(nr, 256, ST_G1);
    hipLaunchKernelGGL(batch_stats_kernel, dim3(g1), dim3(256), 0, s, adv, act, stats, nr);
    LossArgs a;
    a.headout = headout; a.tu = tu; a.act = act; a.mask = mask; a.old_logp = old_logp; a.adv = adv; a.ret = ret;
    a.stats = stats; a.dheadout = dheadout; a.dtu = dtu; a.nr = nr;
    a.e_clip = e_clip; a.entropy_coef = entropy_coef; a.vf_coef = vf_coef;
    a.adv_eps = 1.1920928955078125e-07f;   // np.finfo(np.float32).eps, optimizer.py:38
    a.g1 = g1; a.losses_out = losses_out; a.head_on = head_on;
    hipLaunchKernelGGL(ppo_loss_kernel, dim3(grid1d(nr, 16, ST_G2)), dim3(256), 0, s, a);
    return launch_check("ppo_loss_fwd_bwd");
}

}  // namespace dc
