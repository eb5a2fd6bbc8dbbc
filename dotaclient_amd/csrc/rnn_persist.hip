// Persistent LSTM: ALL time steps of a layer in ONE launch, recurrent weights stationary in registers.
//
// Replaces the S sequential cell steps of nn.LSTM (the BASELINE.json extension of
// /root/reference/policy.py:66,141) and their BPTT (/root/reference/optimizer.py:672) for H <= 128,
// where W_hh (4H x H fp32 = 256 KB at H = 128) fits the register file of one CU: 4 waves x 64 lanes
// x 256 registers.  Sequences are independent (optimizer.py:591 stacks them as batch rows), so a
// workgroup owns FOUR sequences for the whole trajectory and never talks to another workgroup: no
// launch boundary (1.5 us each, 2 x 256 of them per epoch before), no grid barrier, no weight
// re-fetch.  64 trajectories -> 16 workgroups, 256 -> 64.
//
// Matrix instruction: v_mfma_f32_4x4x1_16b_f32 = sixteen independent 4x4 outer products per issue
// (K = 1), exact fp32, 64 FLOP/clk/SIMD like every f32 MFMA.  Block b = lanes 4b..4b+3;
//   A: lane 4b+i holds A_b[i]     B: lane 4b+j holds B_b[j]     D: lane 4b+j, register i = D_b[i][j].
// Row i = sequence (all blocks get the same four rows), column = whatever weight column the lane
// keeps - so each lane simply owns output columns, with its 2 x H (forward) / 2H (backward) weights
// held in registers for the whole launch.  The state h_t (4 x H) / the gate gradients (4 x 4H) make
// one trip through LDS per step (double-buffered, one barrier per step).
//
// Lane roles (wave w, lane l, hi = l >> 5): hidden unit u = 32 w + (l & 31).
//   forward : lanes hi=0 accumulate the i,f gate columns of u, lanes hi=1 the g,o columns; one
//             cross-half exchange later every lane finishes two (sequence, unit) cells:
//             sequences 2*hi and 2*hi+1.  c_t of a cell lives in that lane's registers throughout.
//   backward: lane accumulates dh_rec[seq][u] over gate columns [2H*hi, 2H*hi+2H), halves are summed
//             across the wave halves, same cell ownership; dc_{t+1} and f_{t+1} stay in registers.
#include <utility>
#include "kernels.h"

namespace dc {

template <int H>
struct PersistCfg {
    static constexpr int WAVES = H / 32;
    static constexpr int THREADS = WAVES * 64;
    static constexpr int HLD = H + 4;          // h rows in LDS: +4 floats -> the 4 rows of a b128 read hit disjoint banks
    static constexpr int GLD = 4 * H + 8;      // gate-gradient rows in LDS
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// Gate non-linearities on the hardware transcendentals (v_exp_f32 / v_rcp_f32, ~1 ulp each): the
// cell epilogue sits on the per-step critical path, and libm's range-reduced expf + IEEE division +
// branchy tanhf cost several hundred dependent cycles there.  Absolute error ~1e-7, far inside the
// 1e-4 parity bar (tests/test_gpu_parity.py).
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    // 2*sigmoid(2x) - 1; exp2 overflow to +inf gives rcp(inf) = 0 -> -1, underflow -> +1
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.0f;
}


// ---------------------------------------------------------------------------------------------------
// Hand-pipelined product phase.  hipcc (ROCm 7.2) sinks every LDS read to just before its first use
// and waits for it at once (~100 exposed cycles per 16 MFMAs), and copies the weights it keeps in
// AGPRs back to VGPRs with v_accvgpr_read + hazard nops before each MFMA.  So this phase is issued by
// hand: a ring of RING float4 LDS reads stays in flight (each has RING-1 MFMA groups of cover, waited
// for with a counted lgkmcnt), and the MFMAs take their B operand straight from AGPRs ("a"
// constraint: MFMA A/B operands may be AGPRs on gfx950).  Every statement is asm volatile, so the
// issue order - and with it the lgkmcnt arithmetic - is exactly the program order below; the compiler
// emits no LDS/SMEM operation of its own between the first read and the last wait (checked in the
// .s: the loop body has no s_load and no other ds_* before the epilogue).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

template <int OFF>
__device__ __forceinline__ void lds_read16(float4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}

// forward group: one float4 of h (4 consecutive k) against this lane's two columns.
// FIRST: accumulators start from the inline constant 0 (no VALU-written SrcC in front of an MFMA).
// LAST : pads the MFMA -> VALU read hazard by hand (nothing after an asm is padded by the compiler).
template <int WAIT, bool FIRST, bool LAST>
__device__ __forceinline__ void fwd_mma_group(const float4& a, f32x4& a00, f32x4& a10, f32x4& a01, f32x4& a11,
                                              float w00, float w10, float w01, float w11, float w02, float w12,
                                              float w03, float w13) {
    if constexpr (FIRST) {
        asm volatile(
            "s_waitcnt lgkmcnt(%16)\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, 0\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %1, %4, %9, 0\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %2, %5, %10, 0\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %3, %5, %11, 0\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %0, %6, %12, %0\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %1, %6, %13, %1\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %2, %7, %14, %2\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %3, %7, %15, %3"
            : "=&v"(a00), "=&v"(a10), "=&v"(a01), "=&v"(a11)
            : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "a"(w00), "a"(w10), "a"(w01), "a"(w11), "a"(w02), "a"(w12),
              "a"(w03), "a"(w13), "i"(WAIT));
    } else if constexpr (LAST) {
        asm volatile(
            "s_waitcnt lgkmcnt(%16)\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %1, %4, %9, %1\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %2, %5, %10, %2\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %3, %5, %11, %3\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %0, %6, %12, %0\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %1, %6, %13, %1\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %2, %7, %14, %2\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %3, %7, %15, %3\n\t"
            "s_nop 7\n\ts_nop 7"
            : "+v"(a00), "+v"(a10), "+v"(a01), "+v"(a11)
            : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "a"(w00), "a"(w10), "a"(w01), "a"(w11), "a"(w02), "a"(w12),
              "a"(w03), "a"(w13), "i"(WAIT));
    } else {
        asm volatile(
            "s_waitcnt lgkmcnt(%16)\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %1, %4, %9, %1\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %2, %5, %10, %2\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %3, %5, %11, %3\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %0, %6, %12, %0\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %1, %6, %13, %1\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %2, %7, %14, %2\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %3, %7, %15, %3"
            : "+v"(a00), "+v"(a10), "+v"(a01), "+v"(a11)
            : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "a"(w00), "a"(w10), "a"(w01), "a"(w11), "a"(w02), "a"(w12),
              "a"(w03), "a"(w13), "i"(WAIT));
    }
}

template <int H>
struct FwdProduct {
    static constexpr int NG = H / 4;      // float4 groups
    static constexpr int RING = 8;
    template <int G>
    static __device__ __forceinline__ void group(float4 (&av)[RING], f32x4 (&acc)[4], const float (&w0)[H],
                                                 const float (&w1)[H], uint32_t addr) {
        constexpr int outstanding = (NG - G < RING) ? (NG - G) : RING;   // reads in flight before this group
        fwd_mma_group<outstanding - 1, G == 0, G == NG - 1>(av[G % RING], acc[0], acc[1], acc[2], acc[3], w0[4 * G],
                                                            w1[4 * G], w0[4 * G + 1], w1[4 * G + 1], w0[4 * G + 2],
                                                            w1[4 * G + 2], w0[4 * G + 3], w1[4 * G + 3]);
        if constexpr (G + RING < NG) lds_read16<16 * (G + RING)>(av[G % RING], addr);
    }
    template <int... Gs>
    static __device__ __forceinline__ void groups(float4 (&av)[RING], f32x4 (&acc)[4], const float (&w0)[H],
                                                  const float (&w1)[H], uint32_t addr, std::integer_sequence<int, Gs...>) {
        (group<Gs>(av, acc, w0, w1, addr), ...);
    }
    template <int... Rs>
    static __device__ __forceinline__ void prologue(float4 (&av)[RING], uint32_t addr, std::integer_sequence<int, Rs...>) {
        (lds_read16<16 * Rs>(av[Rs], addr), ...);
    }
    static __device__ __forceinline__ void run(f32x4 (&acc)[4], const float (&w0)[H], const float (&w1)[H], uint32_t addr) {
        float4 av[RING];
        prologue(av, addr, std::make_integer_sequence<int, RING>{});
        groups(av, acc, w0, w1, addr, std::make_integer_sequence<int, NG>{});
    }
};

// backward group: one float4 of gate gradients (4 consecutive k), one column, four accumulator chains.
template <int WAIT, bool FIRST, bool LAST>
__device__ __forceinline__ void bwd_mma_group(const float4& a, f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, float w0,
                                              float w1, float w2, float w3) {
    if constexpr (FIRST) {
        asm volatile(
            "s_waitcnt lgkmcnt(%12)\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, 0\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %1, %5, %9, 0\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %2, %6, %10, 0\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %3, %7, %11, 0"
            : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
            : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "a"(w0), "a"(w1), "a"(w2), "a"(w3), "i"(WAIT));
    } else if constexpr (LAST) {
        asm volatile(
            "s_waitcnt lgkmcnt(%12)\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %1, %5, %9, %1\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %2, %6, %10, %2\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %3, %7, %11, %3\n\t"
            "s_nop 7\n\ts_nop 7"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
            : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "a"(w0), "a"(w1), "a"(w2), "a"(w3), "i"(WAIT));
    } else {
        asm volatile(
            "s_waitcnt lgkmcnt(%12)\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %1, %5, %9, %1\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %2, %6, %10, %2\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %3, %7, %11, %3"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
            : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "a"(w0), "a"(w1), "a"(w2), "a"(w3), "i"(WAIT));
    }
}

template <int KH>
struct BwdProduct {
    static constexpr int NG = KH / 4;
    static constexpr int RING = 16;       // 4 MFMAs per group -> deeper ring for the same cover (lgkmcnt <= 15)
    template <int G>
    static __device__ __forceinline__ void group(float4 (&av)[RING], f32x4 (&acc)[4], const float (&w)[KH], uint32_t addr) {
        constexpr int outstanding = (NG - G < RING) ? (NG - G) : RING;
        bwd_mma_group<outstanding - 1, G == 0, G == NG - 1>(av[G % RING], acc[0], acc[1], acc[2], acc[3], w[4 * G],
                                                            w[4 * G + 1], w[4 * G + 2], w[4 * G + 3]);
        if constexpr (G + RING < NG) lds_read16<16 * (G + RING)>(av[G % RING], addr);
    }
    template <int... Gs>
    static __device__ __forceinline__ void groups(float4 (&av)[RING], f32x4 (&acc)[4], const float (&w)[KH], uint32_t addr,
                                                  std::integer_sequence<int, Gs...>) {
        (group<Gs>(av, acc, w, addr), ...);
    }
    template <int... Rs>
    static __device__ __forceinline__ void prologue(float4 (&av)[RING], uint32_t addr, std::integer_sequence<int, Rs...>) {
        (lds_read16<16 * Rs>(av[Rs], addr), ...);
    }
    static __device__ __forceinline__ void run(f32x4 (&acc)[4], const float (&w)[KH], uint32_t addr) {
        float4 av[RING];
        prologue(av, addr, std::make_integer_sequence<int, RING>{});
        groups(av, acc, w, addr, std::make_integer_sequence<int, NG>{});
    }
};

// ---------------------------------------------------------------------------------------------------
// forward.  gates[row][4H] holds W_ih x + b_ih on entry and the activated gates i,f,g,o on exit.
// hprev/cprev[first row of a sequence] hold h0/c0 (rnn_seed_state); h_t, c_t go to hseq/cseq[row] and
// to hprev/cprev[row+1] (the shifted copies the weight-gradient GEMM and the backward read).
// ---------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(PersistCfg<H>::THREADS, 1) void lstm_fwd_persist_kernel(RnnStepArgs p) {
    using C = PersistCfg<H>;
    __shared__ __attribute__((aligned(16))) float h_lds[2][4][C::HLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5;
    const int u = 32 * wave + (lane & 31);
    const int b0 = blockIdx.x * 4;

    // ---- weights: column (2*hi + m)*H + u of W_hh, all k, for m = 0,1 ------------------------------
    float w0[H], w1[H];
    {
        const float4* r0 = reinterpret_cast<const float4*>(p.Whh + (size_t)((2 * hi + 0) * H + u) * H);
        const float4* r1 = reinterpret_cast<const float4*>(p.Whh + (size_t)((2 * hi + 1) * H + u) * H);
#pragma unroll
        for (int k = 0; k < H / 4; ++k) {
            const float4 x = r0[k], y = r1[k];
            w0[4 * k] = x.x; w0[4 * k + 1] = x.y; w0[4 * k + 2] = x.z; w0[4 * k + 3] = x.w;
            w1[4 * k] = y.x; w1[4 * k + 1] = y.y; w1[4 * k + 2] = y.z; w1[4 * k + 3] = y.w;
        }
    }
    float bh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bh[g] = p.bhh[g * H + u];

    // ---- this lane's two cells ----------------------------------------------------------------------
    int len[2];
    size_t row0[2];
    float c[2];
    int tmax = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = b0 + q;
        const int l = b < p.n_seq ? p.seq_len[b] : 0;
        tmax = max(tmax, l);
    }
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int b = b0 + 2 * hi + cc;
        len[cc] = b < p.n_seq ? p.seq_len[b] : 0;
        row0[cc] = len[cc] > 0 ? (size_t)p.seq_off[b] : 0;
        c[cc] = len[cc] > 0 ? p.cprev[row0[cc] * H + u] : 0.f;
    }
    // h0 -> LDS buffer 0
    for (int e = tid; e < 4 * H; e += C::THREADS) {
        const int q = e / H, j = e - q * H;
        const int b = b0 + q;
        const bool on = b < p.n_seq && p.seq_len[b] > 0;
        h_lds[0][q][j] = on ? p.hprev[(size_t)p.seq_off[b] * H + j] : 0.f;
    }
    // gate pre-activations of step 0
    float xc[2][4], xn[2][4];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int g = 0; g < 4; ++g) xc[cc][g] = len[cc] > 0 ? p.gates[row0[cc] * (size_t)(4 * H) + g * H + u] : 0.f;
    __syncthreads();

    for (int t = 0; t < tmax; ++t) {
        const int cur = t & 1;
        // prefetch next step's gate pre-activations (latency hidden behind the MFMAs)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                xn[cc][g] = (t + 1 < len[cc]) ? p.gates[(row0[cc] + t + 1) * (size_t)(4 * H) + g * H + u] : 0.f;

        // ---- recurrent product: rows = the 4 sequences, this lane's two columns ----------------------
        f32x4 pa[4];   // chains: [0] col0 even k, [1] col1 even k, [2] col0 odd k, [3] col1 odd k
        FwdProduct<H>::run(pa, w0, w1, lds_addr(&h_lds[cur][lane & 3][0]));
        const f32x4 acc0 = pa[0] + pa[2], acc1 = pa[1] + pa[3];   // lanes hi=0: (i,f) of 4 sequences; hi=1: (g,o)

        // ---- cross-half exchange: low lanes need g,o of sequences 0,1; high lanes i,f of 2,3 ---------
        float mine[2][2], recv[2][2];   // [m][cell]
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            mine[0][cc] = hi ? acc0[2 + cc] : acc0[cc];
            mine[1][cc] = hi ? acc1[2 + cc] : acc1[cc];
            recv[0][cc] = __shfl_xor(hi ? acc0[cc] : acc0[2 + cc], 32, 64);
            recv[1][cc] = __shfl_xor(hi ? acc1[cc] : acc1[2 + cc], 32, 64);
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const float ri = hi ? recv[0][cc] : mine[0][cc];
            const float rf = hi ? recv[1][cc] : mine[1][cc];
            const float rg = hi ? mine[0][cc] : recv[0][cc];
            const float ro = hi ? mine[1][cc] : recv[1][cc];
            const bool on = t < len[cc];
            const float ig = fast_sigmoid(xc[cc][0] + (ri + bh[0]));
            const float fg = fast_sigmoid(xc[cc][1] + (rf + bh[1]));
            const float gg = fast_tanh(xc[cc][2] + (rg + bh[2]));
            const float og = fast_sigmoid(xc[cc][3] + (ro + bh[3]));
            const float cn = fg * c[cc] + ig * gg;
            const float hn = og * fast_tanh(cn);
            h_lds[cur ^ 1][2 * hi + cc][u] = on ? hn : 0.f;
            if (on) {
                c[cc] = cn;
                const size_t r = row0[cc] + t;
                float* gt = p.gates + r * (size_t)(4 * H);
                gt[u] = ig; gt[H + u] = fg; gt[2 * H + u] = gg; gt[3 * H + u] = og;
                p.cseq[r * H + u] = cn;
                p.hseq[r * H + u] = hn;
                if (t + 1 < len[cc]) {
                    p.cprev[(r + 1) * H + u] = cn;
                    p.hprev[(r + 1) * H + u] = hn;
                }
            }
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int g = 0; g < 4; ++g) xc[cc][g] = xn[cc][g];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// backward through time.  In: dh[row][H] = dL/dh_t from the layer above (heads), the forward's saved
// gates / cseq / cprev.  Out: dgx[row][4H] = gradient w.r.t. the gate pre-activations (for an LSTM the
// same tensor serves W_ih x + b_ih and W_hh h + b_hh).
// ---------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(PersistCfg<H>::THREADS, 1) void lstm_bwd_persist_kernel(RnnStepArgs p) {
    using C = PersistCfg<H>;
    constexpr int KH = 2 * H;   // gate columns per wave half
    __shared__ __attribute__((aligned(16))) float g_lds[2][4][C::GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5;
    const int u = 32 * wave + (lane & 31);
    const int b0 = blockIdx.x * 4;

    // ---- weights: W_hh[KH*hi + kk][u], kk = 0..KH-1 --------------------------------------------------
    float w[KH];
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) w[kk] = p.Whh[(size_t)(KH * hi + kk) * H + u];

    int len[2];
    size_t row0[2];
    int tmax = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = b0 + q;
        tmax = max(tmax, b < p.n_seq ? p.seq_len[b] : 0);
    }
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int b = b0 + 2 * hi + cc;
        len[cc] = b < p.n_seq ? p.seq_len[b] : 0;
        row0[cc] = len[cc] > 0 ? (size_t)p.seq_off[b] : 0;
    }
    for (int e = tid; e < 4 * C::GLD; e += C::THREADS) (&g_lds[0][0][0])[e] = 0.f;   // "step tmax" has no gradient

    // per-cell carried state and the prefetched operands of the step about to be processed
    float dc_next[2] = {0.f, 0.f}, f_next[2] = {0.f, 0.f};
    float gv[2][4], cv[2], cpv[2], dhv[2];      // current step
    float gn[2][4], cn_[2], cpn[2], dhn[2];     // next (t-1)
    auto fetch = [&](int t, float (&G)[2][4], float (&Cv)[2], float (&Cp)[2], float (&Dh)[2]) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const bool on = t >= 0 && t < len[cc];
            const size_t r = row0[cc] + (on ? t : 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) G[cc][g] = on ? p.gates[r * (size_t)(4 * H) + g * H + u] : 0.f;
            Cv[cc] = on ? p.cseq[r * H + u] : 0.f;
            Cp[cc] = on ? p.cprev[r * H + u] : 0.f;
            Dh[cc] = on ? p.dh[r * H + u] : 0.f;
        }
    };
    fetch(tmax - 1, gv, cv, cpv, dhv);
    __syncthreads();

    for (int t = tmax - 1; t >= 0; --t) {
        const int cur = (tmax - 1 - t) & 1;
        fetch(t - 1, gn, cn_, cpn, dhn);

        // ---- dh_rec[seq][u] = sum_k dgates_{t+1}[seq][k] * W_hh[k][u], this half's k range ------------
        f32x4 pa[4];
        BwdProduct<KH>::run(pa, w, lds_addr(&g_lds[cur][lane & 3][KH * hi]));
        const f32x4 acc = (pa[0] + pa[1]) + (pa[2] + pa[3]);
        float rec[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const float other = __shfl_xor(hi ? acc[cc] : acc[2 + cc], 32, 64);
            rec[cc] = (hi ? acc[2 + cc] : acc[cc]) + other;
        }

#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const bool on = t < len[cc];
            const bool has_next = (t + 1) < len[cc];
            float dh = dhv[cc];
            if (has_next) dh += rec[cc];
            const float ig = gv[cc][0], fg = gv[cc][1], gg = gv[cc][2], og = gv[cc][3];
            const float tc = fast_tanh(cv[cc]);
            float dcv = dh * og * (1.f - tc * tc);
            if (has_next) dcv += dc_next[cc] * f_next[cc];
            const float di = dcv * gg * ig * (1.f - ig);
            const float df = dcv * cpv[cc] * fg * (1.f - fg);
            const float dg = dcv * ig * (1.f - gg * gg);
            const float dO = dh * tc * og * (1.f - og);
            float* gl = &g_lds[cur ^ 1][2 * hi + cc][0];
            gl[u] = on ? di : 0.f; gl[H + u] = on ? df : 0.f; gl[2 * H + u] = on ? dg : 0.f; gl[3 * H + u] = on ? dO : 0.f;
            if (on) {
                dc_next[cc] = dcv; f_next[cc] = fg;
                float* gx = p.dgx + (row0[cc] + t) * (size_t)(4 * H);
                gx[u] = di; gx[H + u] = df; gx[2 * H + u] = dg; gx[3 * H + u] = dO;
            }
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
            for (int g = 0; g < 4; ++g) gv[cc][g] = gn[cc][g];
            cv[cc] = cn_[cc]; cpv[cc] = cpn[cc]; dhv[cc] = dhn[cc];
        }
        __syncthreads();
    }
}

bool lstm_persist_supported(int H) { return H == 64 || H == 128; }

int lstm_forward_persist(RnnStepArgs a, int max_len, hipStream_t s) {
    const dim3 grid((a.n_seq + 3) / 4);
    ProfScope prof("lstm_fwd_persist", 2.0 * a.n_seq * 4.0 * a.H * a.H * max_len,
                   4.0 * a.n_seq * max_len * a.H * (2.0 * 4 + 4.0), s);
    if (a.H == 128) hipLaunchKernelGGL((lstm_fwd_persist_kernel<128>), grid, dim3(PersistCfg<128>::THREADS), 0, s, a);
    else if (a.H == 64) hipLaunchKernelGGL((lstm_fwd_persist_kernel<64>), grid, dim3(PersistCfg<64>::THREADS), 0, s, a);
    else { set_error("lstm_forward_persist: unsupported hidden size", 1011); return 1011; }
    return launch_check("lstm_forward_persist");
}

int lstm_backward_persist(RnnStepArgs a, int max_len, hipStream_t s) {
    const dim3 grid((a.n_seq + 3) / 4);
    ProfScope prof("lstm_bwd_persist", 2.0 * a.n_seq * 4.0 * a.H * a.H * max_len,
                   4.0 * a.n_seq * max_len * a.H * (2.0 * 4 + 3.0), s);
    if (a.H == 128) hipLaunchKernelGGL((lstm_bwd_persist_kernel<128>), grid, dim3(PersistCfg<128>::THREADS), 0, s, a);
    else if (a.H == 64) hipLaunchKernelGGL((lstm_bwd_persist_kernel<64>), grid, dim3(PersistCfg<64>::THREADS), 0, s, a);
    else { set_error("lstm_backward_persist: unsupported hidden size", 1011); return 1011; }
    return launch_check("lstm_backward_persist");
}

}  // namespace dc
