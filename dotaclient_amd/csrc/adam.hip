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
// HBM-bound: 4 B (norm pass) + 28 B (update: p,g,m,v read, p,g,m,v... written) per parameter.
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

// ctl[0] = clip coefficient, ctl[1] = ok flag (1.0 = apply the update), ctl[2] (as u32) = arrival counter of grad_sqnorm_kernel's
// blocks (zero between calls), ctl[3] unused.
// The norms: every (segment, 4096-element chunk) block leaves its partial sum of squares in segsq[n_seg + seg * nchunk + chunk] - no
// clear launch, no same-address atomics, a fixed summation order - and the LAST block to arrive (release / ticket / acquire) does
// what used to be a launch of its own: per-segment norms in the reference's order, clip coefficient, the two NaN guards, step counters.
__device__ __forceinline__ void clip_finalize(const AdamSegs& sg, const double* __restrict__ seg_tot, const int32_t* __restrict__ head_on,
                                              const float* __restrict__ losses, float* __restrict__ norms_out, float* __restrict__ ctl,
                                              int32_t* __restrict__ seg_step, int32_t* __restrict__ status, float max_norm, float vf_coef,
                                              int lane) {
    double sum_norm = 0.0, tot_sq = 0.0, n_act = 0.0;
    for (int s = lane; s < sg.n_seg; s += 64) {
        if (!seg_active(sg, s, head_on, vf_coef)) continue;
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
    const bool loss_nan = losses[0] != losses[0];
    // NaN like the reference (optimizer.py:678-679) - and an INFINITE norm too: with the two-f16-piece products an out-of-range gradient
    // operand becomes inf where the reference holds a finite number; the reference's clip would scale by 0.5 / inf = 0 and write
    // inf * 0 = NaN into the parameters.  Same status word (2), nothing updated.
    const bool norm_nan = unclipped != unclipped || !(total <= 3.0e38f);
    int st = *status;             // sticky: once an epoch tripped a guard, later epochs skip their update too until the caller
    if (st == 0) {                // clears the word (the reference raises at the first NaN epoch: nothing runs after it)
        if (loss_nan) st = 1; else if (norm_nan) st = 2;
    }
    if (lane == 0) {
        norms_out[0] = unclipped;
        norms_out[1] = unclipped * coef;   // every per-parameter norm scales by the same coefficient
        *status = st;
        ctl[0] = coef;
        ctl[1] = st == 0 ? 1.f : 0.f;
    }
    if (st == 0)
        for (int s = lane; s < sg.n_seg; s += 64)
            if (seg_active(sg, s, head_on, vf_coef)) seg_step[s] += 1;
}

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(AdamSegs sg, const float* __restrict__ grad, double* __restrict__ segsq,
                                                          const int32_t* __restrict__ head_on,
                                                          const float* __restrict__ losses, float* __restrict__ norms_out,
                                                          float* __restrict__ ctl, int32_t* __restrict__ seg_step,
                                                          int32_t* __restrict__ status, float max_norm, float vf_coef) {
    const int seg = blockIdx.y;
    const long long len = sg.seg_len[seg];
    const long long c0 = (long long)blockIdx.x * ADAM_CHUNK;
    if (c0 >= len) return;                       // only blocks with a chunk take part (and take a ticket)
    __shared__ double sh[4];
    __shared__ int sh_last;
    // how many blocks take part: the chunks of all segments (every participating block computes the same number)
    int my_chunks = 0;
    for (int s2 = threadIdx.x; s2 < sg.n_seg; s2 += 256) my_chunks += (sg.seg_len[s2] + ADAM_CHUNK - 1) / ADAM_CHUNK;
    my_chunks = (int)wave_sum((float)my_chunks);
    __shared__ int sh_cnt[4];
    if ((threadIdx.x & 63) == 0) sh_cnt[threadIdx.x >> 6] = my_chunks;
    {
        const float* g = grad + sg.seg_off[seg];
        const long long c1 = min(len, c0 + ADAM_CHUNK);
        double s = 0.0;
        for (long long i = c0 + threadIdx.x; i < c1; i += 256) {
            const double v = g[i];
            s += v * v;
        }
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
        __syncthreads();
        // hand-off without fences: an 8-byte write-through (sc1) store, drained before the ticket; sc1 loads on the other side
        if (threadIdx.x == 0) {
            __hip_atomic_store(&segsq[sg.n_seg + seg * gridDim.x + blockIdx.x], (sh[0] + sh[1]) + (sh[2] + sh[3]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned total = (unsigned)(sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3]);
            unsigned* cnt = reinterpret_cast<unsigned*>(ctl + 2);
            const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = t == total - 1;
            if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero again for the next call
            sh_last = last;
        }
    }
    __syncthreads();
    if (!sh_last) return;
    // the last block: per-segment totals - wave w takes segments w, w + 4, ...; lane c the partial of chunk c (+ 64, ...), all loads of a
    // segment in flight together (one lane summing a segment's 64 partials one after the other is 64 dependent L2 round trips), then the
    // fixed wave butterfly - into segsq[0 .. n_seg), then one wave finalises from there
    {
        const int nchunk = (int)gridDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int s2 = wave; s2 < sg.n_seg; s2 += 4) {
            const int nc = (sg.seg_len[s2] + ADAM_CHUNK - 1) / ADAM_CHUNK;
            double sq = 0.0;
            for (int c = lane; c < nc; c += 64) sq += __hip_atomic_load(&segsq[sg.n_seg + s2 * nchunk + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sq = wave_sum(sq);
            if (lane == 0) segsq[s2] = sq;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) clip_finalize(sg, segsq, head_on, losses, norms_out, ctl, seg_step, status, max_norm, vf_coef, threadIdx.x);
}

// One block = ADAM_UPD consecutive elements of one segment, four per thread with all sixteen loads in flight before the first
// dependent instruction (a block of the 4096-element sqnorm chunks walked them in sixteen dependent load -> compute -> store rounds:
// ~220 working blocks on 256 CUs, 25 us for 29 MB).
enum { ADAM_UPD = 1024 };
__global__ __launch_bounds__(256) void adam_update_kernel(AdamSegs sg, float* __restrict__ param, float* __restrict__ grad,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          const float* __restrict__ ctl, const int32_t* __restrict__ seg_step,
                                                          const int32_t* __restrict__ head_on, float vf_coef, double lr,
                                                          double beta1, double beta2, float eps) {
    const int seg = blockIdx.y;
    const long long len = sg.seg_len[seg];
    const long long c0 = (long long)blockIdx.x * ADAM_UPD;
    if (c0 >= len) return;
    if (ctl[1] == 0.f) return;                                   // NaN guard tripped: leave everything alone
    if (!seg_active(sg, seg, head_on, vf_coef)) return;          // grad is None in the reference
    const float coef = ctl[0];
    const long long base = sg.seg_off[seg];
    const long long i0 = base + c0 + threadIdx.x, end = base + min(len, c0 + ADAM_UPD);
    float g[4], mi[4], vi[4], pi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long i = i0 + 256 * j;
        const bool on = i < end;
        g[j] = on ? grad[i] : 0.f; mi[j] = on ? m[i] : 0.f; vi[j] = on ? v[i] : 0.f; pi[j] = on ? param[i] : 0.f;
    }
    // bias corrections: two double pow() - once per block, not once per thread (behind the loads)
    __shared__ float sh_bc[2];
    if (threadIdx.x == 0) {
        const int step = seg_step[seg];                          // already incremented for this update
        const double bc1 = 1.0 - pow(beta1, (double)step);
        const double bc2 = 1.0 - pow(beta2, (double)step);
        sh_bc[0] = (float)(lr / bc1);
        sh_bc[1] = (float)sqrt(bc2);
    }
    __syncthreads();
    const float step_size = sh_bc[0];
    const float bc2_sqrt = sh_bc[1];
    const float w1 = (float)(1.0 - beta1), b2 = (float)beta2, w2 = (float)(1.0 - beta2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long i = i0 + 256 * j;
        if (i >= end) break;
        const float gj = g[j] * coef;                            // clip_grad_norm_ scales in place
        grad[i] = gj;
        const float mj = mi[j] + w1 * (gj - mi[j]);              // exp_avg.lerp_(grad, 1-beta1)
        const float vj = vi[j] * b2 + w2 * gj * gj;              // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
        const float denom = sqrtf(vj) / bc2_sqrt + eps;
        param[i] = pi[j] - step_size * mj / denom;               // param.addcdiv_(exp_avg, denom, -step_size)
        m[i] = mj;
        v[i] = vj;
    }
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
    AdamSegs sg{seg_off, seg_len, seg_gate, n_seg};
    ProfScope prof("gradnorm_clip_adam", 0.0, 32.0 * n_seg * max_seg_len / 8, s);   // (bytes: a stand-in; bench.py prices the region by 32 B x parameters)
    dim3 grid((max_seg_len + ADAM_CHUNK - 1) / ADAM_CHUNK, n_seg);
    hipLaunchKernelGGL(grad_sqnorm_kernel, grid, dim3(256), 0, s, sg, grad, segsq, head_on, losses, norms_out, ctl, seg_step,
                       status, max_norm, vf_coef);
    hipLaunchKernelGGL(adam_update_kernel, dim3((max_seg_len + ADAM_UPD - 1) / ADAM_UPD, n_seg), dim3(256), 0, s, sg, param, grad, m, v, ctl,
                       seg_step, head_on, vf_coef, lr, beta1, beta2, eps);
    return launch_check("gradnorm_clip_adam");
}

}  // namespace dc
