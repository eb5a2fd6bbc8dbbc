// Gradient-norm metrics, clip-by-global-norm and Adam over the flat fp32 parameter buffer.
//
// Replaces /root/reference/optimizer.py:674-681: mean_gradient_norm (optimizer.py:691-695, called
// before and after the clip), torch.nn.utils.clip_grad_norm_(params, 0.5) (coef = 0.5/(norm+1e-6)
// clamped to <= 1) and torch.optim.Adam(lr).step() (betas 0.9/0.999, eps 1e-8, no weight decay;
// optimizer.py:275), including the two NaN guards (optimizer.py:667-669, 678-679): when the loss or
// the gradient norm is NaN nothing is updated and `status` is set so the host can raise ValueError.
// `status` is STICKY: a non-zero word makes every later call skip its update as well (epochs are enqueued back to back
// and read by the host once; the reference would have raised before the next epoch) - the caller zeroes it.
//
// The 34 named parameters are "segments" of the flat buffer (offset, length).  A segment whose head
// took no action in the batch has no gradient in the reference (torch leaves .grad = None): it is
// excluded from both norms, is not clipped and Adam does not advance its step counter.
// HBM-bound: 4 B (norm pass) + 28 B (update: p,g,m,v read, p,g,m,v... written) per parameter - at this model's 0.9 M parameters
// everything lives in L2 and the launch is a chain of latencies: one kernel, below.
#include "kernels.h"

namespace dc {

enum { ADAM_CHUNK = 4096 };

struct AdamSegs {
    const int64_t* seg_off;    // [n_seg] offsets (floats)
    const int32_t* seg_len;    // [n_seg]
    const int32_t* seg_gate;   // [n_seg] -1: always has a grad; 0..4: needs head k active; 5: needs vf_coef > 0
    int n_seg;
};

__device__ __forceinline__ bool seg_active(const AdamSegs& sg, int seg, const int32_t* head_on, float vf_coef) {
    const int g = sg.seg_gate[seg];
    if (g < 0) return true;
    if (g < 5) return head_on[g] != 0;
    return vf_coef > 0.f;
}

// ctl[0] = clip coefficient, ctl[1] = ok flag (1.0 = apply the update), ctl[2] (as u32) = second-level arrival counter of
// gradnorm_clip_adam_kernel's blocks (zero between calls), ctl[3] (as u32) = its release generation (one more per call).
// The norms: every (segment, 4096-element chunk) block leaves its partial sum of squares in segsq[n_seg + seg * nchunk + chunk] - no
// clear launch, no same-address atomics, a fixed summation order - and the LAST block to arrive (release / ticket / acquire) does
// what used to be a launch of its own: per-segment norms in the reference's order, clip coefficient, the two NaN guards, step counters.
// ---------------------------------------------------------------------------------------------------
// One launch (round 6; two before: the norms with a last-block finalise, then the update - 23 + 11 us for 29 MB that live in L2).
// Every microsecond of it is a chain of memory round trips or a queue at one L2 address, so the work is arranged to have few of either:
// a fixed grid of ADAM_GRID co-resident blocks of 1 024 threads walks the 4096-element chunks of all segments (chunk c of the
// concatenated list -> block c mod grid; four elements per thread):
//   1. everything the update needs except the clip coefficient is requested up front: the chunk's g, p, m, v, the segment's step
//      counter (-> bias corrections of the step about to be taken), and - by every block, for whichever turns out to finalise - the
//      segments' gates, head_on, the loss, the status word and the release generation ctl[3];
//   2. the chunk's sum of squares -> segsq[n_seg + c]; arrival tickets in TWO levels (sixteen first-level counters in the spare tail of
//      segsq, then ctl[2]): 250 same-address atomics queue for ~4 us, 16 + 16 do not.  The last block to arrive loads all partial sums
//      in one round trip, sums them per segment in a fixed order (LDS), finalises (clip coefficient, NaN guards, step counters), zeroes the
//      counters and bumps the generation ctl[3]; the others poll the 16 bytes of ctl, which bring coefficient and ok flag with it;
//   3. the update, from the registers of step 1 for a block's first chunk (further chunks - models beyond ADAM_GRID x 4096 parameters -
//      are read again).  No ticket on the way out: nothing is left to reset.
// Every block of the grid is resident (ADAM_GRID = one per CU), so the wait cannot starve the block it waits for.
// ---------------------------------------------------------------------------------------------------
enum { ADAM_GRID = 256, ADAM_THREADS = 1024, ADAM_WAVES = ADAM_THREADS / 64, ADAM_MAX_SEGS = 256, ADAM_PT = ADAM_CHUNK / ADAM_THREADS,
       ADAM_LDS_PART = 2048, ADAM_L1 = 16 };

struct AdamArgs {
    AdamSegs sg;
    float* param; float* grad; float* m; float* v;
    double* segsq; const int32_t* head_on; const float* losses; float* norms_out; float* ctl; int32_t* seg_step; int32_t* status;
    float max_norm, vf_coef, eps;
    double lr, beta1, beta2;
    int slots;                 // doubles in segsq: n_seg * (1 + ceil(max_seg_len / ADAM_CHUNK))
};

// beta^n by squaring (n = Adam's step counter): a handful of f64 multiplies instead of the library's pow()
__device__ __forceinline__ double ipow(double b, int n) {
    double r = 1.0;
    for (; n > 0; n >>= 1) { if (n & 1) r *= b; b *= b; }
    return r;
}

// the norms' finalisation from values already on chip - per-segment norms in the reference's order, clip coefficient, the two NaN guards,
// step counters: one wave
__device__ __forceinline__ void clip_finalize_lds(int n_seg, const double* seg_tot, const int* gate, const int* head_on, float vf_coef, bool loss_nan,
                                                  int st, float max_norm, float* __restrict__ norms_out, float* __restrict__ ctl,
                                                  int32_t* __restrict__ seg_step, int32_t* __restrict__ status, int lane) {
    auto active = [&](int s) { const int g = gate[s]; return g < 0 ? true : (g < 5 ? head_on[g] != 0 : vf_coef > 0.f); };
    double sum_norm = 0.0, tot_sq = 0.0, n_act = 0.0;
    for (int s = lane; s < n_seg; s += 64) {
        if (!active(s)) continue;
        const float nrm = (float)sqrt(seg_tot[s]);
        sum_norm += (double)nrm;
        tot_sq += (double)nrm * (double)nrm;
        n_act += 1.0;
    }
    sum_norm = wave_sum(sum_norm);
    tot_sq = wave_sum(tot_sq);
    n_act = wave_sum(n_act);
    const float unclipped = (float)(sum_norm / n_act);
    const float total = (float)sqrt(tot_sq);
    float coef = max_norm / (total + 1e-6f);
    if (coef > 1.f) coef = 1.f;
    // NaN like the reference (optimizer.py:678-679) - and an INFINITE norm too: with the two-f16-piece products an out-of-range gradient
    // operand becomes inf where the reference holds a finite number; the reference's clip would scale by 0.5 / inf = 0 and write
    // inf * 0 = NaN into the parameters.  Same status word (2), nothing updated.
    const bool norm_nan = unclipped != unclipped || !(total <= 3.0e38f);
    if (st == 0) {                // sticky: once an epoch tripped a guard, later epochs skip their update too until the caller
        if (loss_nan) st = 1; else if (norm_nan) st = 2;      // clears the word (the reference raises at the first NaN epoch: nothing runs after it)
    }
    // (what other blocks of this launch read - ctl[0..1], the step counters - goes out as write-through stores; the caller drains them
    // before it bumps the generation: no fences, which at agent scope write back / invalidate whole caches)
    if (lane == 0) {
        norms_out[0] = unclipped;
        norms_out[1] = unclipped * coef;   // every per-parameter norm scales by the same coefficient
        *status = st;
        __hip_atomic_store(&ctl[0], coef, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl[1], st == 0 ? 1.f : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (st == 0)
        for (int s = lane; s < n_seg; s += 64)
            if (active(s)) __hip_atomic_store(&seg_step[s], seg_step[s] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(ADAM_THREADS) void gradnorm_clip_adam_kernel(AdamArgs a) {
    const AdamSegs& sg = a.sg;
    __shared__ int sh_first[ADAM_MAX_SEGS + 1];       // first chunk of every segment in the concatenated chunk list
    __shared__ int sh_gate[ADAM_MAX_SEGS];
    __shared__ int sh_misc[8];                        // head_on[0..4], loss is NaN, status, release generation at entry
    __shared__ double sh[ADAM_WAVES];
    __shared__ double sh_part[ADAM_LDS_PART];         // the last block: all partial sums
    __shared__ double sh_tot[ADAM_MAX_SEGS];          //                 per-segment sums of squares
    __shared__ float sh_bc[2], sh_ctl[2];
    __shared__ int sh_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef ADAM_TIMING
    long long ts[12];
    int nts = 0;
#define ADAM_STAMP() do { ts[nts++] = __builtin_readcyclecounter(); } while (0)
#else
#define ADAM_STAMP() do { } while (0)
#endif
    ADAM_STAMP();                                      // 0 start
    // chunk counts -> exclusive prefix: wave 0, one segment per lane, Hillis-Steele over the wave (a one-thread scan is a chain of n_seg LDS trips)
    if (wave == 0) {
        int carry = 0;
        for (int base = 0; base < sg.n_seg; base += 64) {
            const int s2 = base + lane;
            int x = s2 < sg.n_seg ? (sg.seg_len[s2] + ADAM_CHUNK - 1) / ADAM_CHUNK : 0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); x += lane >= d ? y : 0; }
            if (s2 < sg.n_seg) sh_first[s2 + 1] = carry + x;
            carry += __shfl(x, 63, 64);
        }
        if (lane == 0) sh_first[0] = 0;
    } else if (wave == 1) {                            // what a finaliser needs, requested by everybody now
        if (lane < 5) sh_misc[lane] = a.head_on[lane];
        else if (lane == 5) { const float l = a.losses[0]; sh_misc[5] = l != l; }
        else if (lane == 6) sh_misc[6] = *a.status;
        else if (lane == 7) sh_misc[7] = (int)__hip_atomic_load(reinterpret_cast<unsigned*>(a.ctl + 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (wave < 6) {
        for (int s2 = tid - 128; s2 < sg.n_seg; s2 += 256) sh_gate[s2] = sg.seg_gate[s2];
    }
    __syncthreads();
    const int total = sh_first[sg.n_seg];
    ADAM_STAMP();                                      // 1 prefix done
    if ((int)blockIdx.x >= total) return;              // (fewer chunks than blocks: the rest takes no part, and no ticket)
    const unsigned n_part = (unsigned)min((int)gridDim.x, total);
    unsigned* const cnt = reinterpret_cast<unsigned*>(a.ctl + 2);
    const unsigned gen0 = (unsigned)sh_misc[7];
    // two ticket levels when segsq has sixteen spare slots behind the partial sums (it has unless nearly every segment is max_seg_len long)
    const bool two_level = a.slots - sg.n_seg - total >= ADAM_L1 && n_part > ADAM_L1;
    double* const l1_base = a.segsq + sg.n_seg + total;

    // segment of chunk c: the number of segments that END at or before c (each lane tests some, one ballot per 64 segments)
    auto locate = [&](int c, int& seg, long long& c0) {
        int s2 = 0;
        for (int base = 0; base < sg.n_seg; base += 64) {
            const int q = base + lane;
            s2 += __popcll(__ballot(q < sg.n_seg && sh_first[q + 1] <= c));
        }
        seg = s2;
        c0 = (long long)(c - sh_first[s2]) * ADAM_CHUNK;
    };
    auto active = [&](int seg) { const int g = sh_gate[seg]; return g < 0 ? true : (g < 5 ? sh_misc[g] != 0 : a.vf_coef > 0.f); };
    auto bias_corrections = [&](int step) {              // -> sh_bc (one thread)
        const double bc1 = 1.0 - ipow(a.beta1, step);
        const double bc2 = 1.0 - ipow(a.beta2, step);
        sh_bc[0] = (float)(a.lr / bc1);
        sh_bc[1] = (float)sqrt(bc2);
    };
    // ---- 1. norms ------------------------------------------------------------------------------------------------------------
    float g0[ADAM_PT], m0[ADAM_PT], v0[ADAM_PT], p0[ADAM_PT];      // the block's FIRST chunk stays in registers across the ticket
    int seg0 = 0; long long c00 = 0;
    for (int c = blockIdx.x; c < total; c += gridDim.x) {
        int seg; long long c0;
        locate(c, seg, c0);
        const long long base = sg.seg_off[seg], len = sg.seg_len[seg];
        const long long i0 = base + c0 + tid, end = base + min(len, c0 + ADAM_CHUNK);
        const bool first = c == (int)blockIdx.x;
        float g[ADAM_PT];
#pragma unroll
        for (int j = 0; j < ADAM_PT; ++j) g[j] = i0 + ADAM_THREADS * j < end ? a.grad[i0 + ADAM_THREADS * j] : 0.f;
        if (first) {
            seg0 = seg; c00 = c0;
#pragma unroll
            for (int j = 0; j < ADAM_PT; ++j) {
                const long long i = i0 + ADAM_THREADS * j;
                const bool on = i < end;
                m0[j] = on ? a.m[i] : 0.f; v0[j] = on ? a.v[i] : 0.f; p0[j] = on ? a.param[i] : 0.f;
            }
            if (tid == ADAM_THREADS - 1) bias_corrections(a.seg_step[seg] + 1);      // (the finaliser increments the counter only after this block's ticket)
        }
        double sq = 0.0;
#pragma unroll
        for (int j = 0; j < ADAM_PT; ++j) { const double x = g[j]; sq += x * x; if (first) g0[j] = g[j]; }
        sq = wave_sum(sq);
        __syncthreads();                               // (sh of the chunk before)
        if (lane == 0) sh[wave] = sq;
        __syncthreads();
        // hand-off without fences: an 8-byte write-through (sc1) store, drained before the ticket; sc1 loads on the other side
        if (tid == 0) {
            double t8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t8[q] = sh[2 * q] + sh[2 * q + 1];
            __hip_atomic_store(&a.segsq[sg.n_seg + c], ((t8[0] + t8[1]) + (t8[2] + t8[3])) + ((t8[4] + t8[5]) + (t8[6] + t8[7])), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    ADAM_STAMP();                                      // 2 partial stored
    // ---- 2. ticket, finalise, release ----------------------------------------------------------------------------------------
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ADAM_STAMP();                                  // 3 loads + store drained
        bool last;
        if (two_level) {
            const unsigned grp = blockIdx.x % ADAM_L1, n_grp = (n_part - grp + ADAM_L1 - 1) / ADAM_L1;
            unsigned* const c1 = reinterpret_cast<unsigned*>(l1_base + grp);
            last = __hip_atomic_fetch_add(c1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_grp - 1;
            if (last) {
                __hip_atomic_store(c1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (nobody else touches it before the next call)
                last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ADAM_L1 - 1;
            }
        } else {
            last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_part - 1;
        }
        if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_last = last;
        ADAM_STAMP();                                  // 4 ticket taken
    }
    __syncthreads();
    if (sh_last) {
        // all partial sums in flight together (tiles of ADAM_LDS_PART), then per segment: lane-strided over its chunks + the fixed wave
        // butterfly, wave w taking segments w, w + 16, ... - from LDS, not one L2 round trip per segment
        for (int s2 = tid; s2 < sg.n_seg; s2 += ADAM_THREADS) sh_tot[s2] = 0.0;
        for (int t0 = 0; t0 < total; t0 += ADAM_LDS_PART) {
            const int t1 = min(total, t0 + ADAM_LDS_PART);
            __syncthreads();
            for (int c = t0 + tid; c < t1; c += ADAM_THREADS)
                sh_part[c - t0] = __hip_atomic_load(&a.segsq[sg.n_seg + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            for (int s2 = wave; s2 < sg.n_seg; s2 += ADAM_WAVES) {
                const int lo = max(sh_first[s2], t0), hi = min(sh_first[s2 + 1], t1);
                if (lo >= hi) continue;
                double sq = 0.0;
                for (int c = lo + lane; c < hi; c += 64) sq += sh_part[c - t0];
                sq = wave_sum(sq);
                if (lane == 0) sh_tot[s2] += sq;
            }
        }
        __syncthreads();
        if (tid == 0) ADAM_STAMP();                    // 5 (last) partials reduced
        for (int s2 = tid; s2 < sg.n_seg; s2 += ADAM_THREADS) a.segsq[s2] = sh_tot[s2];      // (for inspection)
        if (tid < 64)
            clip_finalize_lds(sg.n_seg, sh_tot, sh_gate, sh_misc, a.vf_coef, sh_misc[5] != 0, sh_misc[6], a.max_norm, a.norms_out, a.ctl, a.seg_step,
                              a.status, tid);
        __syncthreads();
        if (tid == 0) {
            ADAM_STAMP();                              // 6 (last) finalised
            // (ctl[0..1] and the step counters were written through and drained by the finalising wave, before the barrier above)
            __hip_atomic_store(reinterpret_cast<unsigned*>(a.ctl + 3), gen0 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_ctl[0] = __hip_atomic_load(&a.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_ctl[1] = __hip_atomic_load(&a.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (tid == 0) {
        // ctl[0..3] as ONE 16-byte read: the generation arrives together with what it guards (written, and drained, before it)
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 w;
        for (;;) {
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(a.ctl) : "memory");
            if (w[3] != gen0) break;
            __builtin_amdgcn_s_sleep(2);
        }
        sh_ctl[0] = __uint_as_float(w[0]); sh_ctl[1] = __uint_as_float(w[1]);
    }
    if (tid == 0) ADAM_STAMP();                        // 5 / 7 released
    __syncthreads();
    // ---- 3. update -----------------------------------------------------------------------------------------------------------
    const bool ok = sh_ctl[1] != 0.f;                              // NaN guard tripped: leave everything alone
    const float coef = sh_ctl[0];
    const float w1 = (float)(1.0 - a.beta1), b2 = (float)a.beta2, w2 = (float)(1.0 - a.beta2);
    for (int c = blockIdx.x; ok && c < total; c += gridDim.x) {
        const bool first = c == (int)blockIdx.x;
        int seg; long long c0;
        if (first) { seg = seg0; c0 = c00; } else locate(c, seg, c0);
        if (!active(seg)) continue;                    // grad is None in the reference
        const long long base = sg.seg_off[seg], len = sg.seg_len[seg];
        const long long i0 = base + c0 + tid, end = base + min(len, c0 + ADAM_CHUNK);
        float g[ADAM_PT], mi[ADAM_PT], vi[ADAM_PT], pi[ADAM_PT];
        if (first) {
#pragma unroll
            for (int j = 0; j < ADAM_PT; ++j) { g[j] = g0[j]; mi[j] = m0[j]; vi[j] = v0[j]; pi[j] = p0[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < ADAM_PT; ++j) {
                const long long i = i0 + ADAM_THREADS * j;
                const bool on = i < end;
                g[j] = on ? a.grad[i] : 0.f; mi[j] = on ? a.m[i] : 0.f; vi[j] = on ? a.v[i] : 0.f; pi[j] = on ? a.param[i] : 0.f;
            }
            __syncthreads();                           // (sh_bc of the chunk before)
            if (tid == 0) bias_corrections(__hip_atomic_load(&a.seg_step[seg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));      // already incremented for this update
            __syncthreads();
        }
        const float step_size = sh_bc[0], bc2_sqrt = sh_bc[1];
#pragma unroll
        for (int j = 0; j < ADAM_PT; ++j) {
            const long long i = i0 + ADAM_THREADS * j;
            if (i >= end) break;
            const float gj = g[j] * coef;                            // clip_grad_norm_ scales in place
            a.grad[i] = gj;
            const float mj = mi[j] + w1 * (gj - mi[j]);              // exp_avg.lerp_(grad, 1-beta1)
            const float vj = vi[j] * b2 + w2 * gj * gj;              // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
            const float denom = sqrtf(vj) / bc2_sqrt + a.eps;
            a.param[i] = pi[j] - step_size * mj / denom;             // param.addcdiv_(exp_avg, denom, -step_size)
            a.m[i] = mj;
            a.v[i] = vj;
        }
    }
#ifdef ADAM_TIMING
    if (tid == 0) {
        ADAM_STAMP();
        if (blockIdx.x == 0 || sh_last) {
            for (int i = nts; i < 12; ++i) ts[i] = ts[0];
            printf("adam block %3d last %d n %d: %lld %lld %lld %lld %lld %lld %lld %lld %lld (cycles since block start)\n", (int)blockIdx.x, sh_last, nts,
                   ts[1] - ts[0], ts[2] - ts[0], ts[3] - ts[0], ts[4] - ts[0], ts[5] - ts[0], ts[6] - ts[0], ts[7] - ts[0], ts[8] - ts[0], ts[9] - ts[0]);
        }
    }
#endif
}

// Data-parallel averaging (distributed.py:24-57): after the SUM all-reduce of the flat gradient bucket,
// each parameter is divided by the number of ranks that had a gradient for it.  counts[0..4] = ranks
// whose head k acted (all-reduced with the bucket), counts[5] = world size.  A rank whose own head was
// inactive keeps "no gradient" for those parameters (its Adam skips them), like the reference, where
// the reduced value is discarded on ranks with grad None (distributed.py:50-57).
__global__ __launch_bounds__(256) void dp_scale_kernel(AdamSegs sg, float* __restrict__ grad,
                                                       const float* __restrict__ counts, float vf_coef) {
    const int seg = blockIdx.y;
    const long long len = sg.seg_len[seg];
    const long long c0 = (long long)blockIdx.x * ADAM_CHUNK;
    if (c0 >= len) return;
    const int g = sg.seg_gate[seg];
    float cnt = counts[5];
    if (g >= 0 && g < 5) cnt = counts[g];
    if (cnt <= 0.f) return;
    const long long base = sg.seg_off[seg];
    const long long c1 = min(len, c0 + ADAM_CHUNK);
    for (long long i = base + c0 + threadIdx.x; i < base + c1; i += 256) grad[i] = grad[i] / cnt;   // grad_data /= has_grad_count (distributed.py:57)
}

int dp_average_grads(const int64_t* seg_off, const int32_t* seg_len, const int32_t* seg_gate, int n_seg, int max_seg_len,
                     float* grad, const float* counts, float vf_coef, hipStream_t s) {
    AdamSegs sg{seg_off, seg_len, seg_gate, n_seg};
    dim3 grid((max_seg_len + ADAM_CHUNK - 1) / ADAM_CHUNK, n_seg);
    hipLaunchKernelGGL(dp_scale_kernel, grid, dim3(256), 0, s, sg, grad, counts, vf_coef);
    return launch_check("dp_average_grads");
}

int gradnorm_clip_adam(const int64_t* seg_off, const int32_t* seg_len, const int32_t* seg_gate, int n_seg, int max_seg_len,
                       float* param, float* grad, float* m, float* v, double* segsq, const int32_t* head_on,
                       const float* losses, float* norms_out, float* ctl, int32_t* seg_step, int32_t* status,
                       float max_norm, float vf_coef, double lr, double beta1, double beta2, float eps, hipStream_t s) {
    if (n_seg > ADAM_MAX_SEGS) { set_error("gradnorm_clip_adam: more than 256 parameter segments", 1021); return 1021; }
    AdamArgs a{};
    a.sg = AdamSegs{seg_off, seg_len, seg_gate, n_seg};
    a.param = param; a.grad = grad; a.m = m; a.v = v; a.segsq = segsq; a.head_on = head_on; a.losses = losses; a.norms_out = norms_out;
    a.ctl = ctl; a.seg_step = seg_step; a.status = status; a.max_norm = max_norm; a.vf_coef = vf_coef; a.eps = eps;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2;
    a.slots = n_seg * (1 + (max_seg_len + ADAM_CHUNK - 1) / ADAM_CHUNK);      // segsq's documented size (include/dotaclient_hip.h)
    ProfScope prof("gradnorm_clip_adam", 0.0, 32.0 * n_seg * max_seg_len / 8, s);   // (bytes: a stand-in; bench.py prices the region by 32 B x parameters)
    hipLaunchKernelGGL(gradnorm_clip_adam_kernel, dim3(ADAM_GRID), dim3(ADAM_THREADS), 0, s, a);
    return launch_check("gradnorm_clip_adam");
}

}  // namespace dc
