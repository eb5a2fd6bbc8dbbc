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
#include <stdlib.h>
#include "kernels.h"
#include "gemm_tiles.h"

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
// forward step t.  NCH = 16-wide K chunks per wave (H/64), compile-time so that every global load of
// the launch - operand fragments AND the epilogue's gate inputs / previous state / biases - is issued
// before the first MFMA: a step costs one memory round trip, not two.
// ---------------------------------------------------------------------------------------------------
template <int CELL, int NCH>
__global__ __launch_bounds__(256) void rnn_fwd_step_kernel(RnnStepArgs p) {
    constexpr int G = (CELL == CELL_GRU) ? 3 : 4;
    __shared__ float red[4 * G * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, t = p.t;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int fi = lane & 15, fq = lane >> 4;

    // ---- epilogue operands of this thread's (sequence, hidden unit): issue the loads now -------------
    const int row_i = tid >> 4, col = tid & 15;
    const int eb = b0 + row_i, j = j0 + col;
    const int len = eb < p.n_seq ? p.seq_len[eb] : 0;
    const bool e_on = t < len;
    const size_t r = e_on ? (size_t)(p.seq_off[eb] + t) : 0;
    float* gt = p.gates + r * (size_t)(G * H);
    float gxv[G], bh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { gxv[g] = e_on ? gt[g * H + j] : 0.f; bh[g] = p.bhh[g * H + j]; }
    const float hp = e_on ? p.hprev[r * H + j] : 0.f;
    float cp = 0.f;
    if constexpr (CELL == CELL_LSTM) cp = e_on ? p.cprev[r * H + j] : 0.f;

    // ---- operand fragments: A rows = sequence b0+fi at step t (inactive rows contribute zeros) -------
    const int ab = b0 + fi;
    const bool a_on = ab < p.n_seq && t < p.seq_len[ab];
    const float* a_row = p.hprev + (a_on ? (size_t)(p.seq_off[ab] + t) * H : 0);
    const int kbase = wave * (NCH * 16) + 4 * fq;
    float4 a[NCH], b[NCH][G];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        a[c] = a_on ? *reinterpret_cast<const float4*>(a_row + kbase + 16 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int g = 0; g < G; ++g)
            b[c][g] = *reinterpret_cast<const float4*>(p.Whh + (size_t)(g * H + j0 + fi) * H + kbase + 16 * c);
    }
    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].x, b[c][g].x, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].y, b[c][g].y, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].z, b[c][g].z, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].w, b[c][g].w, acc[g], 0, 0, 0);
        }
    }
    reduce_partials<G>(red, acc, wave, lane);
    __syncthreads();
    if (!e_on) return;

    float hh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int e = row_i * 16 + col;
        hh[g] = (red[(0 * G + g) * 256 + e] + red[(1 * G + g) * 256 + e]) +
                (red[(2 * G + g) * 256 + e] + red[(3 * G + g) * 256 + e]) + bh[g];
    }
    float hnew;
    if constexpr (CELL == CELL_GRU) {
        const float rg = sigmoidf_(gxv[0] + hh[0]);
        const float zg = sigmoidf_(gxv[1] + hh[1]);
        const float ng = tanhf(gxv[2] + rg * hh[2]);
        hnew = (1.f - zg) * ng + zg * hp;
        gt[j] = rg; gt[H + j] = zg; gt[2 * H + j] = ng;
        p.hn[r * H + j] = hh[2];
    } else {
        const float ig = sigmoidf_(gxv[0] + hh[0]);
        const float fg = sigmoidf_(gxv[1] + hh[1]);
        const float gg = tanhf(gxv[2] + hh[2]);
        const float og = sigmoidf_(gxv[3] + hh[3]);
        const float cn = fg * cp + ig * gg;
        hnew = og * tanhf(cn);
        gt[j] = ig; gt[H + j] = fg; gt[2 * H + j] = gg; gt[3 * H + j] = og;
        p.cseq[r * H + j] = cn;
        if (t + 1 < len) p.cprev[(r + 1) * H + j] = cn;
    }
    p.hseq[r * H + j] = hnew;
    if (t + 1 < len) p.hprev[(r + 1) * H + j] = hnew;
}

// ---------------------------------------------------------------------------------------------------
// backward step t: dh_t (total) and the gate gradients of step t, consuming step t+1's gate gradients.
// NCH = 16-wide K chunks per wave (G*H/64).  Same prefetch discipline as the forward.
// ---------------------------------------------------------------------------------------------------
template <int CELL, int NCH>
__global__ __launch_bounds__(256) void rnn_bwd_step_kernel(RnnStepArgs p) {
    constexpr int G = (CELL == CELL_GRU) ? 3 : 4;
    __shared__ float red[4 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, t = p.t, GH = G * H;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int fi = lane & 15, fq = lane >> 4;

    // ---- epilogue operands first ---------------------------------------------------------------------
    const int row_i = tid >> 4, col = tid & 15;
    const int eb = b0 + row_i, j = j0 + col;
    const int len = eb < p.n_seq ? p.seq_len[eb] : 0;
    const bool e_on = t < len;
    const bool has_next = (t + 1) < len;
    const size_t r = e_on ? (size_t)(p.seq_off[eb] + t) : 0;
    const float* gt = p.gates + r * (size_t)GH;
    float gv[G];
#pragma unroll
    for (int g = 0; g < G; ++g) gv[g] = e_on ? gt[g * H + j] : 0.f;
    float dh = e_on ? p.dh[r * H + j] : 0.f;
    float x0 = 0.f, x1 = 0.f, n0 = 0.f, n1 = 0.f;   // cell-specific extras
    if constexpr (CELL == CELL_GRU) {
        x0 = e_on ? p.hn[r * H + j] : 0.f;
        x1 = e_on ? p.hprev[r * H + j] : 0.f;
        if (has_next) { n0 = p.gates[(r + 1) * (size_t)GH + H + j]; n1 = p.dh[(r + 1) * H + j]; }
    } else {
        x0 = e_on ? p.cseq[r * H + j] : 0.f;
        x1 = e_on ? p.cprev[r * H + j] : 0.f;
        if (has_next) { n0 = p.gates[(r + 1) * (size_t)GH + H + j]; n1 = p.dc[(r + 1) * H + j]; }
    }

    // ---- recurrent term: sum_k dgh[row(t+1)][k] * Whh[k][j] (sequences that have a step t+1) ---------
    const int ab = b0 + fi;
    const bool a_on = ab < p.n_seq && (t + 1) < p.seq_len[ab];
    const float* a_row = p.dgh + (a_on ? (size_t)(p.seq_off[ab] + t + 1) * GH : 0);
    const float* b_row = p.WhhT + (size_t)(j0 + fi) * GH;
    const int kbase = wave * (NCH * 16) + 4 * fq;
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int UN = NCH <= 8 ? NCH : (NCH % 8 == 0 ? 8 : 6);   // must divide NCH (3,6,12,24 | 4,8,16,32)
    static_assert(NCH % UN == 0, "chunk unroll must divide the chunk count");
#pragma unroll 1
    for (int c0 = 0; c0 < NCH; c0 += UN) {
        float4 a[UN], b[UN];
#pragma unroll
        for (int c = 0; c < UN; ++c) {
            a[c] = a_on ? *reinterpret_cast<const float4*>(a_row + kbase + 16 * (c0 + c)) : make_float4(0.f, 0.f, 0.f, 0.f);
            b[c] = *reinterpret_cast<const float4*>(b_row + kbase + 16 * (c0 + c));
        }
#pragma unroll
        for (int c = 0; c < UN; ++c) {
            // two independent accumulator chains hide the 40-cycle dependent-MFMA latency
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].x, b[c].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].y, b[c].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].z, b[c].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].w, b[c].w, acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave * 256 + (4 * (lane >> 4) + q) * 16 + (lane & 15)] = acc0[q] + acc1[q];
    __syncthreads();
    if (!e_on) return;

    const int e = row_i * 16 + col;
    if (has_next) dh += (red[e] + red[256 + e]) + (red[512 + e] + red[768 + e]);
    if constexpr (CELL == CELL_GRU) {
        if (has_next) dh += n1 * n0;                 // direct path h_{t+1} = ... + z_{t+1} * h_t
        p.dh[r * H + j] = dh;
        const float rg = gv[0], zg = gv[1], ng = gv[2];
        const float dn_pre = dh * (1.f - zg) * (1.f - ng * ng);
        const float dz_pre = dh * (x1 - ng) * zg * (1.f - zg);
        const float dr_pre = dn_pre * x0 * rg * (1.f - rg);
        float* gx = p.dgx + r * (size_t)GH;
        float* gh = p.dgh + r * (size_t)GH;
        gx[j] = dr_pre; gx[H + j] = dz_pre; gx[2 * H + j] = dn_pre;
        gh[j] = dr_pre; gh[H + j] = dz_pre; gh[2 * H + j] = dn_pre * rg;
    } else {
        p.dh[r * H + j] = dh;
        const float ig = gv[0], fg = gv[1], gg = gv[2], og = gv[3];
        const float tc = tanhf(x0);
        float dcv = dh * og * (1.f - tc * tc);
        if (has_next) dcv += n1 * n0;                // dc_{t+1} * f_{t+1}
        p.dc[r * H + j] = dcv;
        float* gx = p.dgx + r * (size_t)GH;
        gx[j] = dcv * gg * ig * (1.f - ig);
        gx[H + j] = dcv * x1 * fg * (1.f - fg);
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
// (bf16: the state buffer holds bf16 elements - configs[4]'s bf16 storage, policy.hip bf16_store())
__global__ void seed_state_kernel(const float* __restrict__ h0, float* __restrict__ hprev,
                                  const int64_t* __restrict__ seq_off, const int32_t* __restrict__ seq_len, int n_seq,
                                  int H, int bf16) {
    const int b = blockIdx.x;
    if (b >= n_seq || seq_len[b] <= 0) return;
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        const float v = h0 ? h0[(size_t)b * H + j] : 0.f;
        if (bf16) reinterpret_cast<uint16_t*>(hprev)[(size_t)seq_off[b] * H + j] = (uint16_t)(cvt_pk_bf16(v, 0.f) & 0xffffu);
        else hprev[(size_t)seq_off[b] * H + j] = v;
    }
}

// hT[b] = hseq[last row of sequence b]
__global__ void final_state_kernel(const float* __restrict__ hseq, float* __restrict__ hT,
                                   const int64_t* __restrict__ seq_off, const int32_t* __restrict__ seq_len, int n_seq,
                                   int H, int bf16) {
    const int b = blockIdx.x;
    if (b >= n_seq || seq_len[b] <= 0) return;
    const size_t r = (size_t)(seq_off[b] + seq_len[b] - 1) * H;
    for (int j = threadIdx.x; j < H; j += blockDim.x)
        hT[(size_t)b * H + j] = bf16 ? __uint_as_float((unsigned)reinterpret_cast<const uint16_t*>(hseq)[r + j] << 16) : hseq[r + j];
}

int transpose(const float* in, float* out, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, s, in, out, rows, cols);
    return launch_check("transpose");
}

int rnn_seed_state(const float* h0, float* hprev, const int64_t* seq_off, const int32_t* seq_len, int n_seq, int H,
                   hipStream_t s, int bf16) {
    hipLaunchKernelGGL(seed_state_kernel, dim3(n_seq), dim3(128), 0, s, h0, hprev, seq_off, seq_len, n_seq, H, bf16);
    return launch_check("rnn_seed_state");
}

int rnn_final_state(const float* hseq, float* hT, const int64_t* seq_off, const int32_t* seq_len, int n_seq, int H,
                    hipStream_t s, int bf16) {
    hipLaunchKernelGGL(final_state_kernel, dim3(n_seq), dim3(128), 0, s, hseq, hT, seq_off, seq_len, n_seq, H, bf16);
    return launch_check("rnn_final_state");
}

// out[b][:] = seq[prev_row[b]][:] (zeros when prev_row[b] < 0): the initial state of chunk b = the state after the last
// step of the previous chunk of the same rollout (optimizer.py:384,408), zeros for a rollout's first chunk (policy.py:77-78)
__global__ __launch_bounds__(128) void gather_state_kernel(const float* __restrict__ seq, const int64_t* __restrict__ prev_row,
                                                           float* __restrict__ out, int H, int bf16) {
    const int b = blockIdx.x;
    const int64_t r = prev_row[b];
    for (int j = threadIdx.x * 4; j < H; j += 128 * 4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= 0 && bf16) {
            const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(seq) + r * H + j);
            v = make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u));
        } else if (r >= 0) {
            v = *reinterpret_cast<const float4*>(seq + r * H + j);
        }
        *reinterpret_cast<float4*>(out + (size_t)b * H + j) = v;
    }
}

int rnn_gather_state(const float* seq, const int64_t* prev_row, float* out, int n, int H, hipStream_t s, int bf16) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gather_state_kernel, dim3(n), dim3(128), 0, s, seq, prev_row, out, H, bf16);
    return launch_check("rnn_gather_state");
}

template <int CELL>
static bool launch_fwd(const RnnStepArgs& a, dim3 grid, hipStream_t s) {
    switch (a.H / 64) {
        case 1: hipLaunchKernelGGL((rnn_fwd_step_kernel<CELL, 1>), grid, dim3(256), 0, s, a); return true;
        case 2: hipLaunchKernelGGL((rnn_fwd_step_kernel<CELL, 2>), grid, dim3(256), 0, s, a); return true;
        case 4: hipLaunchKernelGGL((rnn_fwd_step_kernel<CELL, 4>), grid, dim3(256), 0, s, a); return true;
        case 8: hipLaunchKernelGGL((rnn_fwd_step_kernel<CELL, 8>), grid, dim3(256), 0, s, a); return true;
        default: return false;
    }
}
template <int CELL>
static bool launch_bwd(const RnnStepArgs& a, dim3 grid, hipStream_t s) {
    constexpr int G = (CELL == CELL_GRU) ? 3 : 4;
    switch (a.H / 64) {
        case 1: hipLaunchKernelGGL((rnn_bwd_step_kernel<CELL, 1 * G>), grid, dim3(256), 0, s, a); return true;
        case 2: hipLaunchKernelGGL((rnn_bwd_step_kernel<CELL, 2 * G>), grid, dim3(256), 0, s, a); return true;
        case 4: hipLaunchKernelGGL((rnn_bwd_step_kernel<CELL, 4 * G>), grid, dim3(256), 0, s, a); return true;
        case 8: hipLaunchKernelGGL((rnn_bwd_step_kernel<CELL, 8 * G>), grid, dim3(256), 0, s, a); return true;
        default: return false;
    }
}

static int check_h(int H) {
    if (H == 64 || H == 128 || H == 256 || H == 512) return 0;
    set_error("rnn: hidden size must be 64, 128, 256 or 512", 1010);
    return 1010;
}

// all steps of one layer, forward.  max_len = max(seq_len) (host value).
// DC_DIMS_RNN_PER_STEP forces the launch-per-step kernels (A/B measurements, parity tests of both)
static bool persist_enabled(int flags) { return !(flags & DC_DIMS_RNN_PER_STEP); }

// true when the layer runs on the register-resident LSTM kernels (which read W_hh directly: no W_hh^T needed)
bool rnn_uses_persistent(int cell, int H, int flags) { return cell == CELL_LSTM && lstm_persist_supported(H) && persist_enabled(flags); }

int rnn_forward_layer(int cell, RnnStepArgs a, int max_len, hipStream_t s) {
    if (int e = check_h(a.H)) return e;
    // hprev/cprev[first row of a sequence] = h0/c0 (zeros when absent).  The one-sequence-per-workgroup LSTM kernels
    // and the team kernels read h0/c0 themselves and write those rows (two launches less per pass); the others are
    // seeded here.
    const bool lstm_persist = cell == CELL_LSTM && lstm_persist_supported(a.H) && persist_enabled(a.flags);
    const bool team = !lstm_persist && persist_enabled(a.flags) && rnn_team_supported(cell, a.H, a.n_seq, a.flags);
    const bool self_seeding = (lstm_persist && lstm_persist_use_valu(a.n_seq, a.flags)) || team;
    if (!self_seeding) {
        if (int e = rnn_seed_state(a.h0, a.hprev, a.seq_off, a.seq_len, a.n_seq, a.H, s, a.bf16_store)) return e;
        if (cell == CELL_LSTM)
            if (int e = rnn_seed_state(a.c0, a.cprev, a.seq_off, a.seq_len, a.n_seq, a.H, s)) return e;
    }
    if (lstm_persist) return lstm_forward_persist(a, max_len, s);
    if (team) return rnn_team_forward(cell, a, max_len, s);
    if (lstm_team512_supported(cell, a.H, a.flags, a.Whh_bf)) return lstm_team512_forward(a, max_len, s);
    if (lstm_step_bf16_supported(cell, a.H, a.flags, a.Whh_bf)) return lstm_forward_steps_bf16(a, max_len, s);
    dim3 grid(a.H / 16, (a.n_seq + 15) / 16);
    for (int t = 0; t < max_len; ++t) {
        a.t = t;
        const double G = cell == CELL_GRU ? 3 : 4;
        ProfScope prof("rnn_fwd_step", 2.0 * a.n_seq * G * a.H * a.H,
                       4.0 * (G * a.H * a.H + a.n_seq * a.H * (2.0 * G + 4.0)), s);
        if (cell == CELL_GRU) launch_fwd<CELL_GRU>(a, grid, s);
        else launch_fwd<CELL_LSTM>(a, grid, s);
    }
    return launch_check("rnn_forward_layer");
}

int rnn_backward_layer(int cell, RnnStepArgs a, int max_len, hipStream_t s) {
    if (int e = check_h(a.H)) return e;
    if (cell == CELL_LSTM && lstm_persist_supported(a.H) && persist_enabled(a.flags)) return lstm_backward_persist(a, max_len, s);
    if (persist_enabled(a.flags) && rnn_team_supported(cell, a.H, a.n_seq, a.flags)) return rnn_team_backward(cell, a, max_len, s);
    if (lstm_team512_supported(cell, a.H, a.flags, a.WhhT_bf)) return lstm_team512_backward(a, max_len, s);
    if (lstm_step_bf16_supported(cell, a.H, a.flags, a.WhhT_bf)) return lstm_backward_steps_bf16(a, max_len, s);
    dim3 grid(a.H / 16, (a.n_seq + 15) / 16);
    for (int t = max_len - 1; t >= 0; --t) {
        a.t = t;
        const double G = cell == CELL_GRU ? 3 : 4;
        ProfScope prof("rnn_bwd_step", 2.0 * a.n_seq * G * a.H * a.H,
                       4.0 * (G * a.H * a.H + a.n_seq * a.H * (3.0 * G + 6.0)), s);
        if (cell == CELL_GRU) launch_bwd<CELL_GRU>(a, grid, s);
        else launch_bwd<CELL_LSTM>(a, grid, s);
    }
    return launch_check("rnn_backward_layer");
}

}  // namespace dc
