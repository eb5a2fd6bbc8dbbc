// Recurrent core (GRU as in the reference, LSTM for the BASELINE.json extension configs): forward and
// backward-through-time over packed variable-length trajectories.
//
// Replaces nn.GRU / nn.LSTM at /root/reference/policy.py:66,141 (torch==1.0.0 cell maths: gate order
// r,z,n with n = tanh(W_in x + b_in + r*(W_hn h + b_hn)), h' = (1-z)*n + z*h; LSTM gate order i,f,g,o)
// and the BPTT torch autograd performs at /root/reference/optimizer.py:672.
//
// Structure: the input projection W_ih x + b_ih for ALL steps is hoisted into one big GEMM (gemm.hip);
// what remains sequential is, per time step, the [B,H] x [H,G*H] recurrent product plus the gate
// maths.  One launch per time step (a dependent launch boundary costs ~1.5 us on MI355X, less than any
// grid-wide barrier), each launch tiled so that it spreads over many CUs:
//     workgroup = 16 sequences x 16 hidden units x all G gates, 4 waves splitting K,
//     v_mfma_f32_16x16x4_f32 (exact fp32), partial sums combined through LDS, then the 256 threads
//     each finish one (sequence, hidden unit): gates, state update, stores.
// Operand fragments come straight from L2 as 16-byte loads: lane (i = lane&15, q = lane>>4) holds
// k = 16*chunk + 4*q + e (e = 0..3) of row i for both A (state rows) and B (weight rows), so the four
// MFMAs of a chunk sum a permuted but identical set of k - no LDS staging, no transposes.
//
// Packed rows: sequence b occupies rows [seq_off[b], seq_off[b]+seq_len[b]); step t of sequence b is
// row seq_off[b]+t.  hprev[row] holds h_{t-1} (h0 at the first row of a sequence), hseq[row] holds
// h_t; the forward writes h_t to both hseq[row] and hprev[row+1].
#include "kernels.h"

namespace dc {

enum { CELL_GRU = 0, CELL_LSTM = 1 };


template <int G>
__device__ __forceinline__ void reduce_partials(float* red, const f32x4 (&acc)[G], int wave, int lane) {
    // C/D layout of the 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * G + g) * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[g][r];
}

// ---------------------------------------------------------------------------------------------------
// forward step t
// ---------------------------------------------------------------------------------------------------
template <int CELL>
__global__ __launch_bounds__(256) void rnn_fwd_step_kernel(RnnStepArgs p) {
    constexpr int G = (CELL == CELL_GRU) ? 3 : 4;
    __shared__ float red[4 * G * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, t = p.t;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int fi = lane & 15, fq = lane >> 4;

    // A rows: sequence b0+fi at step t (inactive rows contribute zeros)
    const int ab = b0 + fi;
    const bool a_on = ab < p.n_seq && t < p.seq_len[ab];
    const float* a_row = a_on ? p.hprev + (size_t)(p.seq_off[ab] + t) * H : nullptr;
    const int kw = H / 4;  // K range of this wave
    const int kbase = wave * kw + 4 * fq;

    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kc = 0; kc < kw; kc += 16) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_on) a = *reinterpret_cast<const float4*>(a_row + kbase + kc);
        float4 b[G];
#pragma unroll
        for (int g = 0; g < G; ++g)
            b[g] = *reinterpret_cast<const float4*>(p.Whh + (size_t)(g * H + j0 + fi) * H + kbase + kc);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[g].x, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[g].y, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[g].z, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[g].w, acc[g], 0, 0, 0);
        }
    }
    reduce_partials<G>(red, acc, wave, lane);
    __syncthreads();

    const int row_i = tid >> 4, col = tid & 15;
    const int b = b0 + row_i, j = j0 + col;
    if (b >= p.n_seq) return;
    const int len = p.seq_len[b];
    if (t >= len) return;
    const size_t r = (size_t)(p.seq_off[b] + t);
    float hh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int e = row_i * 16 + col;
        hh[g] = (red[(0 * G + g) * 256 + e] + red[(1 * G + g) * 256 + e]) +
                (red[(2 * G + g) * 256 + e] + red[(3 * G + g) * 256 + e]) + p.bhh[g * H + j];
    }
    float* gt = p.gates + r * (size_t)(G * H);
    const float hp = p.hprev[r * H + j];
    float hnew;
    if constexpr (CELL == CELL_GRU) {
        const float rg = sigmoidf_(gt[j] + hh[0]);
        const float zg = sigmoidf_(gt[H + j] + hh[1]);
        const float ng = tanhf(gt[2 * H + j] + rg * hh[2]);
        hnew = (1.f - zg) * ng + zg * hp;
        gt[j] = rg; gt[H + j] = zg; gt[2 * H + j] = ng;
        p.hn[r * H + j] = hh[2];
    } else {
        const float ig = sigmoidf_(gt[j] + hh[0]);
        const float fg = sigmoidf_(gt[H + j] + hh[1]);
        const float gg = tanhf(gt[2 * H + j] + hh[2]);
        const float og = sigmoidf_(gt[3 * H + j] + hh[3]);
        const float cn = fg * p.cprev[r * H + j] + ig * gg;
        hnew = og * tanhf(cn);
        gt[j] = ig; gt[H + j] = fg; gt[2 * H + j] = gg; gt[3 * H + j] = og;
        p.cseq[r * H + j] = cn;
        if (t + 1 < len) p.cprev[(r + 1) * H + j] = cn;
    }
    p.hseq[r * H + j] = hnew;
    if (t + 1 < len) p.hprev[(r + 1) * H + j] = hnew;
}

// ---------------------------------------------------------------------------------------------------
// backward step t: dh_t (total) and the gate gradients of step t, consuming step t+1's gate gradients
// ---------------------------------------------------------------------------------------------------
template <int CELL>
__global__ __launch_bounds__(256) void rnn_bwd_step_kernel(RnnStepArgs p) {
    constexpr int G = (CELL == CELL_GRU) ? 3 : 4;
    __shared__ float red[4 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, t = p.t, GH = G * H;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int fi = lane & 15, fq = lane >> 4;

    // recurrent term: sum_k dgh[row(t+1)][k] * Whh[k][j]  (only for sequences that have a step t+1)
    const int ab = b0 + fi;
    const bool a_on = ab < p.n_seq && (t + 1) < p.seq_len[ab];
    const float* a_row = a_on ? p.dgh + (size_t)(p.seq_off[ab] + t + 1) * GH : nullptr;
    const float* b_row = p.WhhT + (size_t)(j0 + fi) * GH;
    const int kw = GH / 4;
    const int kbase = wave * kw + 4 * fq;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < kw; kc += 16) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_on) a = *reinterpret_cast<const float4*>(a_row + kbase + kc);
        const float4 b = *reinterpret_cast<const float4*>(b_row + kbase + kc);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
    __syncthreads();

    const int row_i = tid >> 4, col = tid & 15;
    const int b = b0 + row_i, j = j0 + col;
    if (b >= p.n_seq) return;
    const int len = p.seq_len[b];
    if (t >= len) return;
    const size_t r = (size_t)(p.seq_off[b] + t);
    const int e = row_i * 16 + col;
    float dh = p.dh[r * H + j];
    const bool has_next = (t + 1) < len;
    if (has_next) dh += (red[e] + red[256 + e]) + (red[512 + e] + red[768 + e]);
    const float* gt = p.gates + r * (size_t)GH;
    if constexpr (CELL == CELL_GRU) {
        if (has_next) {
            // direct path h_{t+1} = ... + z_{t+1} * h_t
            const float zn = p.gates[(r + 1) * (size_t)GH + H + j];
            dh += p.dh[(r + 1) * H + j] * zn;
        }
        p.dh[r * H + j] = dh;
        const float rg = gt[j], zg = gt[H + j], ng = gt[2 * H + j];
        const float hnv = p.hn[r * H + j];
        const float hp = p.hprev[r * H + j];
        const float dn_pre = dh * (1.f - zg) * (1.f - ng * ng);
        const float dz_pre = dh * (hp - ng) * zg * (1.f - zg);
        const float dr_pre = dn_pre * hnv * rg * (1.f - rg);
        float* gx = p.dgx + r * (size_t)GH;
        float* gh = p.dgh + r * (size_t)GH;
        gx[j] = dr_pre; gx[H + j] = dz_pre; gx[2 * H + j] = dn_pre;
        gh[j] = dr_pre; gh[H + j] = dz_pre; gh[2 * H + j] = dn_pre * rg;
    } else {
        p.dh[r * H + j] = dh;
        const float ig = gt[j], fg = gt[H + j], gg = gt[2 * H + j], og = gt[3 * H + j];
        const float tc = tanhf(p.cseq[r * H + j]);
        float dcv = dh * og * (1.f - tc * tc);
        if (has_next) dcv += p.dc[(r + 1) * H + j] * p.gates[(r + 1) * (size_t)GH + H + j];
        p.dc[r * H + j] = dcv;
        const float cp = p.cprev[r * H + j];
        float* gx = p.dgx + r * (size_t)GH;
        gx[j] = dcv * gg * ig * (1.f - ig);
        gx[H + j] = dcv * cp * fg * (1.f - fg);
        gx[2 * H + j] = dcv * ig * (1.f - gg * gg);
        gx[3 * H + j] = dh * tc * og * (1.f - og);
    }
}

// out[c][r] = in[r][c]   (weight transposes for the backward recurrent product; tiny)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                        int cols) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(size_t)(r0 + i) * cols + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) out[(size_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

// hprev[first row of sequence b] = h0[b]   (and cprev likewise)
__global__ void seed_state_kernel(const float* __restrict__ h0, float* __restrict__ hprev,
                                  const int64_t* __restrict__ seq_off, const int32_t* __restrict__ seq_len, int n_seq,
                                  int H) {
    const int b = blockIdx.x;
    if (b >= n_seq || seq_len[b] <= 0) return;
    for (int j = threadIdx.x; j < H; j += blockDim.x) hprev[(size_t)seq_off[b] * H + j] = h0 ? h0[(size_t)b * H + j] : 0.f;
}

// hT[b] = hseq[last row of sequence b]
__global__ void final_state_kernel(const float* __restrict__ hseq, float* __restrict__ hT,
                                   const int64_t* __restrict__ seq_off, const int32_t* __restrict__ seq_len, int n_seq,
                                   int H) {
    const int b = blockIdx.x;
    if (b >= n_seq || seq_len[b] <= 0) return;
    for (int j = threadIdx.x; j < H; j += blockDim.x)
        hT[(size_t)b * H + j] = hseq[(size_t)(seq_off[b] + seq_len[b] - 1) * H + j];
}

int transpose(const float* in, float* out, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, s, in, out, rows, cols);
    return launch_check("transpose");
}

int rnn_seed_state(const float* h0, float* hprev, const int64_t* seq_off, const int32_t* seq_len, int n_seq, int H,
                   hipStream_t s) {
    hipLaunchKernelGGL(seed_state_kernel, dim3(n_seq), dim3(128), 0, s, h0, hprev, seq_off, seq_len, n_seq, H);
    return launch_check("rnn_seed_state");
}

int rnn_final_state(const float* hseq, float* hT, const int64_t* seq_off, const int32_t* seq_len, int n_seq, int H,
                    hipStream_t s) {
    hipLaunchKernelGGL(final_state_kernel, dim3(n_seq), dim3(128), 0, s, hseq, hT, seq_off, seq_len, n_seq, H);
    return launch_check("rnn_final_state");
}

// all steps of one layer, forward.  max_len = max(seq_len) (host value).
int rnn_forward_layer(int cell, RnnStepArgs a, int max_len, hipStream_t s) {
    if (a.H % 64 != 0) { set_error("rnn: hidden size must be a multiple of 64", 1010); return 1010; }
    dim3 grid(a.H / 16, (a.n_seq + 15) / 16);
    for (int t = 0; t < max_len; ++t) {
        a.t = t;
        const double G = cell == CELL_GRU ? 3 : 4;
        ProfScope prof("rnn_fwd_step", 2.0 * a.n_seq * G * a.H * a.H,
                       4.0 * (G * a.H * a.H + a.n_seq * a.H * (2.0 * G + 4.0)), s);
        if (cell == CELL_GRU) hipLaunchKernelGGL(rnn_fwd_step_kernel<CELL_GRU>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(rnn_fwd_step_kernel<CELL_LSTM>, grid, dim3(256), 0, s, a);
    }
    return launch_check("rnn_forward_layer");
}

int rnn_backward_layer(int cell, RnnStepArgs a, int max_len, hipStream_t s) {
    if (a.H % 64 != 0) { set_error("rnn: hidden size must be a multiple of 64", 1010); return 1010; }
    dim3 grid(a.H / 16, (a.n_seq + 15) / 16);
    for (int t = max_len - 1; t >= 0; --t) {
        a.t = t;
        const double G = cell == CELL_GRU ? 3 : 4;
        ProfScope prof("rnn_bwd_step", 2.0 * a.n_seq * G * a.H * a.H,
                       4.0 * (G * a.H * a.H + a.n_seq * a.H * (3.0 * G + 6.0)), s);
        if (cell == CELL_GRU) hipLaunchKernelGGL(rnn_bwd_step_kernel<CELL_GRU>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(rnn_bwd_step_kernel<CELL_LSTM>, grid, dim3(256), 0, s, a);
    }
    return launch_check("rnn_backward_layer");
}

}  // namespace dc
