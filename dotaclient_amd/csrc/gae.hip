// GAE(lambda) advantage / rewards-to-go scan.
//
// Replaces /root/reference/optimizer.py:53-64 (discount + advantage_returns) together with the
// per-step sub-reward sum of optimizer.py:397 and the terminal-zero append of optimizer.py:417-420.
//
//   r_t   = sum_k rewards[t, k]                         (float32, numpy's pairwise order for n = 10)
//   d_t   = r_t + gamma * V_{t+1} - V_t                 (float32, one rounding per op; V_L = 0)
//   A_t   = d_t + (gamma*lam) * A_{t+1}                 (float64 accumulate, cast to float32)
//   R_t   = r_t + gamma * R_{t+1}                       (float64 accumulate, cast to float32)
//
// One wavefront per rollout (rollouts longer than one LDS block: block by block from the end, carrying the float64 state).  The
// steps of a block are split into 64 contiguous lane chunks; each
// lane reduces its chunk to the affine map y_in -> a*y_in + b, the 64 maps are combined with a
// wavefront (Hillis-Steele, shuffle-based) suffix scan, and each lane then replays its chunk with the
// exact carry-in.  Everything is staged through LDS so that global reads/writes are coalesced.
// HBM-bound: 44 B read + 8 B written per env-step.
#include "kernels.h"

// hipcc defaults to -ffp-contract=fast, which would fuse the reference's separately-rounded
// multiply/add pairs into fma (the __f*_rn helpers are inline header functions and do not help);
// with contraction off every plain operator in this file rounds exactly once.
#pragma clang fp contract(off)

namespace dc {

// numpy's pairwise summation for a contiguous run of 10 float32 (loops_utils.h.src, n in [8,128]):
// eight running lanes, tree-combined, then the two leftovers added sequentially.
__device__ __forceinline__ float reward_sum10(const float* r) {
    const float a = (r[0] + r[1]) + (r[2] + r[3]);
    const float b = (r[4] + r[5]) + (r[6] + r[7]);
    float s = a + b;
    s = s + r[8];
    s = s + r[9];
    return s;
}

struct Affine {  // y -> a*y + b
    double a, b;
};
// apply `inner` first, then `outer`
__device__ __forceinline__ Affine compose(const Affine& outer, const Affine& inner) {
    Affine r;
    r.a = outer.a * inner.a;
    r.b = outer.b + outer.a * inner.b;
    return r;
}

// y[t] = x[t] + g * y[t+1] over s[0..L) in place (float32 in, float64 accumulate, float32 out), y[L] = carry_term:
// lane-local chunk -> affine map (reverse time), wavefront suffix scan of the 64 maps, replay of the chunk with the
// exact carry-in.  Called by all 64 lanes of ONE wavefront; s lives in LDS and was written by this wavefront.
// Returns (to every lane) the float64 value of y at the block's FIRST step, before its rounding to float32: the carry into the
// block in front of it when a rollout is longer than one LDS block.
__device__ __forceinline__ double suffix_scan_replay(float* s, int L, int lane, double g, double carry_term) {
    const int chunk = (L + 63) / 64;
    const int lo = min(lane * chunk, L);
    const int hi = min(lo + chunk, L);
    Affine m = {1.0, 0.0};
    for (int t = hi - 1; t >= lo; --t) {
        m.b = (double)s[t] + g * m.b;
        m.a *= g;
    }
    // suffix scan over lanes: after it, lane i holds the map of chunks i..63
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        Affine n;
        n.a = __shfl_down(m.a, o, 64); n.b = __shfl_down(m.b, o, 64);
        if (lane + o < 64) m = compose(m, n);
    }
    // carry into lane i = value at the first step of lane i+1's suffix
    const double na = __shfl_down(m.a, 1, 64), nb = __shfl_down(m.b, 1, 64);
    double c = (lane == 63) ? carry_term : (carry_term == 0.0 ? nb : nb + na * carry_term);
    // replay the chunk with the exact carry: one rounding per multiply and per add, like the C loop inside lfilter
    for (int t = hi - 1; t >= lo; --t) {
        const double p = g * c;
        c = (double)s[t] + p;
        s[t] = (float)c;
    }
    return __shfl(c, 0, 64);
}

// A rollout longer than `cap` steps (what fits the LDS) is scanned in blocks of `cap` steps from its END to its front: the
// recurrences run backwards in time, so a block's float64 value at its first step is the terminal carry of the block before it
// (the reference's lfilter has no length limit, optimizer.py:53-54).  L <= cap: one block, carry 0.
__global__ __launch_bounds__(64) void gae_scan_kernel(const float* __restrict__ rewards,   // [rows,10]
                                                      const float* __restrict__ values,    // [rows]
                                                      const int64_t* __restrict__ seq_off, // [n_seq]
                                                      const int32_t* __restrict__ seq_len, // [n_seq]
                                                      float gamma, double gamma_d, double gl_d, int cap,
                                                      float* __restrict__ adv, float* __restrict__ ret) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int seq = blockIdx.x;
    const int lane = threadIdx.x;
    const int L = seq_len[seq];
    if (L <= 0) return;
    const int64_t base = seq_off[seq];
    const int CH = min(L, cap);
    float* s_delta = lds;        // [CH]  later overwritten by advantages
    float* s_rsum = lds + CH;    // [CH]  later overwritten by returns
    double carry_a = 0.0, carry_r = 0.0;   // terminal carry 0: the appended zeros of optimizer.py:417-420

    for (int hi = L; hi > 0; hi -= CH) {
        const int lo = max(hi - CH, 0), n = hi - lo;
        // pass 1 (coalesced): per-step reward sum and TD residual
        for (int i = lane; i < n; i += 64) {
            const int t = lo + i;
            const float* rp = rewards + (base + t) * 10;
            float rr[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) rr[k] = rp[k];
            const float r = reward_sum10(rr);
            const float v0 = values[base + t];
            const float v1 = (t + 1 < L) ? values[base + t + 1] : 0.0f;
            // reference: rewards[:-1] + gamma * values[1:] - values[:-1]  in float32, left to right
            const float gv = gamma * v1;
            const float d = (r + gv) - v0;
            s_rsum[i] = r;
            s_delta[i] = d;
        }
        __syncthreads();
        // passes 2 + 3: the two reverse recurrences
        carry_a = suffix_scan_replay(s_delta, n, lane, gl_d, carry_a);
        carry_r = suffix_scan_replay(s_rsum, n, lane, gamma_d, carry_r);
        __syncthreads();
        for (int i = lane; i < n; i += 64) {
            adv[base + lo + i] = s_delta[i];
            ret[base + lo + i] = s_rsum[i];
        }
        __syncthreads();
    }
}

enum { GAE_CAP = 20480, DISCOUNT_CAP = 40960 };   // steps per LDS block: 2 x 4 B resp. 4 B per step in 160 KB

int gae_scan(const float* rewards, const float* values, const int64_t* seq_off, const int32_t* seq_len,
             int n_seq, int max_len, double gamma, double lam, float* adv, float* ret, hipStream_t stream) {
    if (n_seq <= 0) return 0;
    const size_t lds_bytes = (size_t)(max_len < GAE_CAP ? max_len : GAE_CAP) * 2 * sizeof(float);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)gae_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) { set_error("gae_scan: hipFuncSetAttribute", (int)e); return (int)e; }
    }
    // The reference keeps gamma and lam as python doubles: gamma*lam is a double product
    // (optimizer.py:61), lfilter runs in double with the double gamma (optimizer.py:63), and only the
    // TD residuals use float32(gamma) (numpy scalar-times-float32-array arithmetic, optimizer.py:60).
    // algorithmic bytes: 10 sub-rewards + value read, advantage + return written per env-step (44 + 8 B)
    ProfScope prof("gae_scan", 0.0, 52.0 * n_seq * max_len, stream);
    hipLaunchKernelGGL(gae_scan_kernel, dim3(n_seq), dim3(64), lds_bytes, stream, rewards, values, seq_off,
                       seq_len, (float)gamma, gamma, gamma * lam, (int)GAE_CAP, adv, ret);
    return launch_check("gae_scan");
}

// optimizer.py:53-54 `discount` for one vector: y[t] = x[t] + gamma * y[t+1], y[n] = 0 (float64 accumulate); blocks of `cap`
// entries from the end, like gae_scan_kernel.
__global__ __launch_bounds__(64) void discount_kernel(const float* __restrict__ x, int n, double g, int cap, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const int CH = min(n, cap);
    double carry = 0.0;
    for (int hi = n; hi > 0; hi -= CH) {
        const int lo = max(hi - CH, 0), m = hi - lo;
        for (int i = lane; i < m; i += 64) lds[i] = x[lo + i];
        __syncthreads();
        carry = suffix_scan_replay(lds, m, lane, g, carry);
        __syncthreads();
        for (int i = lane; i < m; i += 64) y[lo + i] = lds[i];
        __syncthreads();
    }
}

// optimizer.py:57-64 `advantage_returns` for one rollout with ANY terminal entries: rewards / values are the (L+1)-long
// vectors of the reference (already summed over the sub-rewards; entry L is whatever the caller appended):
//   deltas = rewards[:-1] + gamma * values[1:] - values[:-1]   (float32)
//   adv    = discount(deltas, gamma * lam)
//   ret    = discount(rewards, gamma)[:-1]                      -> the scan over L entries enters with y[L] = rewards[L]
__global__ __launch_bounds__(64) void advantage_returns_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
                                                               int L, float gamma, double gamma_d, double gl_d, int cap,
                                                               float* __restrict__ adv, float* __restrict__ ret) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const int CH = min(L, cap);
    float* s_delta = lds;
    float* s_r = lds + CH;
    double carry_a = 0.0, carry_r = (double)rewards[L];
    for (int hi = L; hi > 0; hi -= CH) {
        const int lo = max(hi - CH, 0), n = hi - lo;
        for (int i = lane; i < n; i += 64) {
            const int t = lo + i;
            const float r = rewards[t];
            const float gv = gamma * values[t + 1];
            s_delta[i] = (r + gv) - values[t];
            s_r[i] = r;
        }
        __syncthreads();
        carry_a = suffix_scan_replay(s_delta, n, lane, gl_d, carry_a);
        carry_r = suffix_scan_replay(s_r, n, lane, gamma_d, carry_r);
        __syncthreads();
        for (int i = lane; i < n; i += 64) {
            adv[lo + i] = s_delta[i];
            ret[lo + i] = s_r[i];
        }
        __syncthreads();
    }
}

static int scan_lds_attr(const void* fn, size_t lds_bytes) {
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) { set_error("gae: hipFuncSetAttribute", (int)e); return (int)e; }
    }
    return 0;
}

int discount(const float* x, int n, double gamma, float* y, hipStream_t stream) {
    if (n <= 0) return 0;
    const size_t lds_bytes = (size_t)(n < DISCOUNT_CAP ? n : DISCOUNT_CAP) * sizeof(float);
    if (int e = scan_lds_attr((const void*)discount_kernel, lds_bytes)) return e;
    hipLaunchKernelGGL(discount_kernel, dim3(1), dim3(64), lds_bytes, stream, x, n, gamma, (int)DISCOUNT_CAP, y);
    return launch_check("discount");
}

int advantage_returns(const float* rewards, const float* values, int L, double gamma, double lam, float* adv, float* ret,
                      hipStream_t stream) {
    if (L <= 0) return 0;
    const size_t lds_bytes = (size_t)(L < GAE_CAP ? L : GAE_CAP) * 2 * sizeof(float);
    if (int e = scan_lds_attr((const void*)advantage_returns_kernel, lds_bytes)) return e;
    hipLaunchKernelGGL(advantage_returns_kernel, dim3(1), dim3(64), lds_bytes, stream, rewards, values, L, (float)gamma, gamma,
                       gamma * lam, (int)GAE_CAP, adv, ret);
    return launch_check("advantage_returns");
}

}  // namespace dc
