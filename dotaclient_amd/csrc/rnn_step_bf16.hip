// Recurrent steps of the bf16 path (DC_DIMS_BF16, BASELINE.json configs[4]: "5v5 hidden=512 2-layer LSTM, bf16 MFMA path"):
// the [B,H] x [H,4H] product of a time step on v_mfma_f32_32x32x16_bf16 - bf16 operands (W_hh pre-converted once per pass,
// h_{t-1} / the gate gradients of step t+1 rounded while they are loaded), f32 accumulation, f32 state and gate maths.
//
// Replaces nn.LSTM's recurrence at /root/reference/policy.py:66,141 (cell parametrised as BASELINE.json asks) and its BPTT
// (/root/reference/optimizer.py:672) for hidden sizes no register-resident kernel covers (H = 512: W_hh is 4 MB per layer).
// The f32 kernels of rnn.hip give each workgroup 16 sequences x 16 units on the f32 MFMA (1/16 of the bf16 rate) and re-read
// their 128 KB slice of W_hh for every 16 sequences: 64 MB of L2 traffic and 20-24 us per step at 256 sequences.  Here:
//   forward : workgroup = 32 sequences x 16 units x 4 gates (two 32 x 32 blocks: gates i|f and g|o of the 16 units), the four
//             waves split K = H; the partial sums meet in LDS, then 256 threads finish two (sequence, unit) cells each;
//   backward: workgroup = 32 sequences x 32 units, K = 4H split over the four waves, four cells per thread.
// Launch per time step (a dependent launch boundary costs ~1.5 us, rnn.hip); every global load of a launch - operand fragments
// and the epilogue's inputs - is issued before the first MFMA.
#include "kernels.h"
#include "gemm_tiles.h"

namespace dc {
namespace {

// eight consecutive f32 -> one bf16 MFMA operand (round to nearest even)
__device__ __forceinline__ u32x4 to_bf16x8(const float4& a, const float4& b) {
    return u32x4{cvt_pk_bf16(a.x, a.y), cvt_pk_bf16(a.z, a.w), cvt_pk_bf16(b.x, b.y), cvt_pk_bf16(b.z, b.w)};
}

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ void spill_block(float* red, const f32x16& acc, int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
}

// LDS operand tiles: [rows][KC] bf16, rows SB_ROW bytes apart (16 bytes of padding: the sixteen lanes a ds_read_b128 serves
// per cycle read sixteen different rows at one column - 260 dwords apart, 4 banks - without sharing a bank)
enum { SB_KC = 512, SB_ROW = SB_KC * 2 + 16 };

// Operands reach the workgroup as ROWS: a wave-wide load covers one whole row segment (1 KB of bf16), i.e. whole cache lines.

// ---- forward step t, H = KC * NC ------------------------------------------------------------------------------------------
// workgroup = 32 sequences x 16 units x 4 gates; LDS row 16 g + u of the weight tile = gate g of unit j0 + u, i.e. block c of the
// product holds gates 2c (columns 0..15) and 2c + 1 (columns 16..31)
template <int NC>
__global__ __launch_bounds__(256) void lstm_fwd_step_bf16_kernel(RnnStepArgs p, const uint16_t* __restrict__ Wb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Bs = smem;                               // [64][KC]
    char* As = smem + 64 * SB_ROW;                 // [32][KC]
    float* red = reinterpret_cast<float*>(smem + 96 * SB_ROW);     // [4 waves][2 blocks][32][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, t = p.t;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 32;
    const int fr = lane & 31, fq = lane >> 5;

    // ---- epilogue operands of this thread's two (sequence, unit) cells: issue the loads now ----------------------------
    const int u = tid & 15, j = j0 + u;
    bool e_on[2];
    size_t r[2];
    int len[2];
    float gxv[2][4], cp[2], bh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bh[g] = p.bhh[g * H + j];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int eb = b0 + (tid >> 4) + 16 * i;
        len[i] = eb < p.n_seq ? p.seq_len[eb] : 0;
        e_on[i] = t < len[i];
        r[i] = e_on[i] ? (size_t)(p.seq_off[eb] + t) : 0;
        const float* gt = p.gates + r[i] * (size_t)(4 * H);
#pragma unroll
        for (int g = 0; g < 4; ++g) gxv[i][g] = e_on[i] ? gt[g * H + j] : 0.f;
        cp[i] = e_on[i] ? p.cprev[r[i] * H + j] : 0.f;
    }

    f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;

    // staging roles: weights - wave w, pass i: LDS row 4 i + w, lane = 16-byte piece of its 1 KB;  state - wave w, pass i: sequence
    // row 4 i + w, lane = 32-byte piece (8 floats) of its 2 KB
    // h_{t-1} of every sequence comes from the step-major bf16 copy the previous step's epilogue left (stepbf[(t - 1) & 1][b][H]:
    // one contiguous quarter megabyte), not from the packed f32 rows: those lie a whole trajectory apart per sequence (a page
    // each), and gathering 32 of them took 10 us per 64 KB - measured, see DESIGN.md.  Step 0 reads h0 from the f32 rows.
    u32x4 wreg[16], areg[8];
    const uint16_t* hb_in = p.stepbf + (size_t)((t + 1) & 1) * p.n_seq * H;
    auto issue = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rr = 4 * i + wave;
            wreg[i] = *reinterpret_cast<const u32x4*>(Wb + (size_t)((rr >> 4) * H + j0 + (rr & 15)) * H + kc * SB_KC + lane * 8);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ab = b0 + 4 * i + wave;
            const bool a_on = ab < p.n_seq && t < p.seq_len[ab];
            areg[i] = u32x4{0, 0, 0, 0};
            if (a_on && t > 0) {
                areg[i] = *reinterpret_cast<const u32x4*>(hb_in + (size_t)ab * H + kc * SB_KC + lane * 8);
            } else if (a_on) {
                const float* src = p.hprev + (size_t)p.seq_off[ab] * H + kc * SB_KC + lane * 8;
                areg[i] = to_bf16x8(*reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4));
            }
        }
    };
    issue(0);
#pragma unroll 1
    for (int kc = 0; kc < NC; ++kc) {
        if (kc > 0) __syncthreads();                      // the previous chunk's fragments have been read
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<u32x4*>(Bs + (4 * i + wave) * SB_ROW + lane * 16) = wreg[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(As + (4 * i + wave) * SB_ROW + lane * 16) = areg[i];
        __syncthreads();
        if (kc + 1 < NC) issue(kc + 1);
        // wave w contracts k = 128 w .. 128 w + 127 of the chunk
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int kb = (wave * 128 + ks * 16 + fq * 8) * 2;
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(As + fr * SB_ROW + kb);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bs + (32 * c + fr) * SB_ROW + kb);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) spill_block(red + (wave * 2 + c) * 1024, acc[c], lane);
    __syncthreads();
    uint16_t* hb_out = p.stepbf + (size_t)(t & 1) * p.n_seq * H;

#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (!e_on[i]) continue;
        const int row = (tid >> 4) + 16 * i;
        float hh[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int e = (g >> 1) * 1024 + row * 32 + (g & 1) * 16 + u;
            hh[g] = (red[e] + red[2048 + e]) + (red[4096 + e] + red[6144 + e]) + bh[g];
        }
        const float ig = sigmoidf_(gxv[i][0] + hh[0]);
        const float fg = sigmoidf_(gxv[i][1] + hh[1]);
        const float gg = tanhf(gxv[i][2] + hh[2]);
        const float og = sigmoidf_(gxv[i][3] + hh[3]);
        const float cn = fg * cp[i] + ig * gg;
        const float hnew = og * tanhf(cn);
        float* gt = p.gates + r[i] * (size_t)(4 * H);
        gt[j] = ig; gt[H + j] = fg; gt[2 * H + j] = gg; gt[3 * H + j] = og;
        p.cseq[r[i] * H + j] = cn;
        p.hseq[r[i] * H + j] = hnew;
        if (t + 1 < len[i]) { p.cprev[(r[i] + 1) * H + j] = cn; p.hprev[(r[i] + 1) * H + j] = hnew; }
        hb_out[(size_t)(b0 + row) * H + j] = (uint16_t)(cvt_pk_bf16(hnew, 0.f) & 0xffffu);
    }
}

// ---- backward step t: dh_t (total), dc_t and the gate gradients of step t, consuming step t + 1's gate gradients.
// workgroup = 32 sequences x 32 units; K = 4H in NC chunks of KC
template <int NC>
__global__ __launch_bounds__(256) void lstm_bwd_step_bf16_kernel(RnnStepArgs p, const uint16_t* __restrict__ WTb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Bs = smem;                               // [32][KC]
    char* As = smem + 32 * SB_ROW;                 // [32][KC]
    float* red = reinterpret_cast<float*>(smem + 64 * SB_ROW);     // [4 waves][32][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, t = p.t, GH = 4 * H;
    const int j0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int fr = lane & 31, fq = lane >> 5;

    // ---- epilogue operands of this thread's four cells first -------------------------------------------------------------
    const int u = tid & 31, j = j0 + u;
    bool e_on[4], has_next[4];
    size_t r[4];
    float gv[4][4], dhv[4], x0[4], x1[4], n0[4], n1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int eb = b0 + (tid >> 5) + 8 * i;
        const int len = eb < p.n_seq ? p.seq_len[eb] : 0;
        e_on[i] = t < len;
        has_next[i] = (t + 1) < len;
        r[i] = e_on[i] ? (size_t)(p.seq_off[eb] + t) : 0;
        const float* gt = p.gates + r[i] * (size_t)GH;
#pragma unroll
        for (int g = 0; g < 4; ++g) gv[i][g] = e_on[i] ? gt[g * H + j] : 0.f;
        dhv[i] = e_on[i] ? p.dh[r[i] * H + j] : 0.f;
        x0[i] = e_on[i] ? p.cseq[r[i] * H + j] : 0.f;
        x1[i] = e_on[i] ? p.cprev[r[i] * H + j] : 0.f;
        n0[i] = 0.f; n1[i] = 0.f;
        if (has_next[i]) { n0[i] = p.gates[(r[i] + 1) * (size_t)GH + H + j]; n1[i] = p.dc[(r[i] + 1) * H + j]; }
    }

    // ---- recurrent term: sum_k dg[row(t+1)][k] * Whh[k][j] (sequences that have a step t + 1) -----------------------------
    f32x16 acc0, acc1;
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
    // the gate gradients of step t + 1 come from the step-major bf16 copy its epilogue left (stepbf[(t + 1) & 1][b][4H]), see the forward
    u32x4 wreg[8], areg[8];
    const uint16_t* gb_in = p.stepbf + (size_t)((t + 1) & 1) * p.n_seq * GH;
    uint16_t* gb_out = p.stepbf + (size_t)(t & 1) * p.n_seq * GH;
    auto issue = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rr = 4 * i + wave;
            wreg[i] = *reinterpret_cast<const u32x4*>(WTb + (size_t)(j0 + rr) * GH + kc * SB_KC + lane * 8);
            const int ab = b0 + rr;
            const bool a_on = ab < p.n_seq && (t + 1) < p.seq_len[ab];
            areg[i] = a_on ? *reinterpret_cast<const u32x4*>(gb_in + (size_t)ab * GH + kc * SB_KC + lane * 8) : u32x4{0, 0, 0, 0};
        }
    };
    issue(0);
#pragma unroll 1
    for (int kc = 0; kc < NC; ++kc) {
        if (kc > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *reinterpret_cast<u32x4*>(Bs + (4 * i + wave) * SB_ROW + lane * 16) = wreg[i];
            *reinterpret_cast<u32x4*>(As + (4 * i + wave) * SB_ROW + lane * 16) = areg[i];
        }
        __syncthreads();
        if (kc + 1 < NC) issue(kc + 1);
#pragma unroll
        for (int ks = 0; ks < 8; ks += 2) {       // two accumulator chains
            const int kb = (wave * 128 + ks * 16 + fq * 8) * 2;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(As + fr * SB_ROW + kb),
                                                           *reinterpret_cast<const bf16x8*>(Bs + fr * SB_ROW + kb), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(As + fr * SB_ROW + kb + 32),
                                                           *reinterpret_cast<const bf16x8*>(Bs + fr * SB_ROW + kb + 32), acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc0[q] += acc1[q];
    spill_block(red + wave * 1024, acc0, lane);
    __syncthreads();

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!e_on[i]) continue;
        const int e = ((tid >> 5) + 8 * i) * 32 + u;
        float dh = dhv[i];
        if (has_next[i]) dh += (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
        p.dh[r[i] * H + j] = dh;
        const float ig = gv[i][0], fg = gv[i][1], gg = gv[i][2], og = gv[i][3];
        const float tc = tanhf(x0[i]);
        float dcv = dh * og * (1.f - tc * tc);
        if (has_next[i]) dcv += n1[i] * n0[i];                // dc_{t+1} * f_{t+1}
        p.dc[r[i] * H + j] = dcv;
        float* gx = p.dgx + r[i] * (size_t)GH;
        const float d4[4] = {dcv * gg * ig * (1.f - ig), dcv * x1[i] * fg * (1.f - fg), dcv * ig * (1.f - gg * gg), dh * tc * og * (1.f - og)};
        uint16_t* gb = gb_out + (size_t)(b0 + (tid >> 5) + 8 * i) * GH;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            gx[g * H + j] = d4[g];
            gb[g * H + j] = (uint16_t)(cvt_pk_bf16(d4[g], 0.f) & 0xffffu);
        }
    }
}

enum { FWD_LDS = 96 * SB_ROW + 4 * 2 * 1024 * 4, BWD_LDS = 64 * SB_ROW + 4 * 1024 * 4 };

template <class K>
static int set_lds(K kern, int bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { set_error("lstm_step_bf16: hipFuncSetAttribute", (int)e); return (int)e; }
    return 0;
}

}  // namespace

// LSTM layers in bf16 mode whose W_hh arrived as bf16 (both orientations); H = 256 keeps its team kernels (f32 recurrence)
bool lstm_step_bf16_supported(int cell, int H, int flags, const void* Wb) {
    return cell == 1 && (flags & DC_DIMS_BF16) && !(flags & DC_DIMS_RNN_PER_STEP) && Wb != nullptr && (H == 512 || H == 1024);   // (H = 512: behind rnn_team512.hip)
}
// (the callers also need 4 * n_seq <= rows: the step-major bf16 copies live in the layer's GRU-only `hn` buffer, policy.hip)

int lstm_forward_steps_bf16(RnnStepArgs a, int max_len, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        if (int e = set_lds(lstm_fwd_step_bf16_kernel<1>, FWD_LDS)) return e;
        if (int e = set_lds(lstm_fwd_step_bf16_kernel<2>, FWD_LDS)) return e;
        if (int e = set_lds(lstm_bwd_step_bf16_kernel<4>, BWD_LDS)) return e;
        if (int e = set_lds(lstm_bwd_step_bf16_kernel<8>, BWD_LDS)) return e;
        attr = true;
    }
    const dim3 grid(a.H / 16, (a.n_seq + 31) / 32);
    for (int t = 0; t < max_len; ++t) {
        a.t = t;
        ProfScope prof("rnn_fwd_step", 2.0 * a.n_seq * 4.0 * a.H * a.H, 2.0 * 4.0 * a.H * a.H + 4.0 * a.n_seq * a.H * 12.0, s);
        if (a.H == 512) hipLaunchKernelGGL((lstm_fwd_step_bf16_kernel<1>), grid, dim3(256), FWD_LDS, s, a, a.Whh_bf);
        else hipLaunchKernelGGL((lstm_fwd_step_bf16_kernel<2>), grid, dim3(256), FWD_LDS, s, a, a.Whh_bf);
    }
    return launch_check("lstm_forward_steps_bf16");
}

int lstm_backward_steps_bf16(RnnStepArgs a, int max_len, hipStream_t s) {
    const dim3 grid(a.H / 32, (a.n_seq + 31) / 32);
    for (int t = max_len - 1; t >= 0; --t) {
        a.t = t;
        ProfScope prof("rnn_bwd_step", 2.0 * a.n_seq * 4.0 * a.H * a.H, 2.0 * 4.0 * a.H * a.H + 4.0 * a.n_seq * a.H * 18.0, s);
        if (a.H == 512) hipLaunchKernelGGL((lstm_bwd_step_bf16_kernel<4>), grid, dim3(256), BWD_LDS, s, a, a.WhhT_bf);
        else hipLaunchKernelGGL((lstm_bwd_step_bf16_kernel<8>), grid, dim3(256), BWD_LDS, s, a, a.WhhT_bf);
    }
    return launch_check("lstm_backward_steps_bf16");
}

}  // namespace dc
