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

// lanes 32..63 of `lo` <-> lanes 0..31 of `hi_` (v_permlane32_swap): afterwards
//   lo  = [lo.low  | hi_.low ]      hi_ = [lo.high | hi_.high]
__device__ __forceinline__ void half_swap(float& lo, float& hi_) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi_), false, false);
    lo = __uint_as_float(r[0]);
    hi_ = __uint_as_float(r[1]);
}

template <int H>
struct FwdProduct {
    static constexpr int NG = H / 4;      // float4 groups
    static constexpr int RING = 8;
    template <int G, class Slot>
    static __device__ __forceinline__ void group(float4 (&av)[RING], f32x4 (&acc)[4], const float (&w0)[H],
                                                 const float (&w1)[H], uint32_t addr, Slot& slot) {
        constexpr int outstanding = (NG - G < RING) ? (NG - G) : RING;   // reads in flight before this group
        fwd_mma_group<outstanding - 1, G == 0, G == NG - 1>(av[G % RING], acc[0], acc[1], acc[2], acc[3], w0[4 * G],
                                                            w1[4 * G], w0[4 * G + 1], w1[4 * G + 1], w0[4 * G + 2],
                                                            w1[4 * G + 2], w0[4 * G + 3], w1[4 * G + 3]);
        if constexpr (G + RING < NG) lds_read16<16 * (G + RING)>(av[G % RING], addr);
        slot(std::integral_constant<int, G>{});   // off-critical-path work issued in the shadow of the MFMAs
    }
    template <class Slot, int... Gs>
    static __device__ __forceinline__ void groups(float4 (&av)[RING], f32x4 (&acc)[4], const float (&w0)[H],
                                                  const float (&w1)[H], uint32_t addr, Slot& slot,
                                                  std::integer_sequence<int, Gs...>) {
        (group<Gs>(av, acc, w0, w1, addr, slot), ...);
    }
    template <int... Rs>
    static __device__ __forceinline__ void prologue(float4 (&av)[RING], uint32_t addr, std::integer_sequence<int, Rs...>) {
        (lds_read16<16 * Rs>(av[Rs], addr), ...);
    }
    template <class Slot>
    static __device__ __forceinline__ void run(f32x4 (&acc)[4], const float (&w0)[H], const float (&w1)[H], uint32_t addr,
                                               Slot& slot) {
        float4 av[RING];
        prologue(av, addr, std::make_integer_sequence<int, RING>{});
        groups(av, acc, w0, w1, addr, slot, std::make_integer_sequence<int, NG>{});
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
    template <int G, class Slot>
    static __device__ __forceinline__ void group(float4 (&av)[RING], f32x4 (&acc)[4], const float (&w)[KH], uint32_t addr,
                                                 Slot& slot) {
        constexpr int outstanding = (NG - G < RING) ? (NG - G) : RING;
        bwd_mma_group<outstanding - 1, G == 0, G == NG - 1>(av[G % RING], acc[0], acc[1], acc[2], acc[3], w[4 * G],
                                                            w[4 * G + 1], w[4 * G + 2], w[4 * G + 3]);
        if constexpr (G + RING < NG) lds_read16<16 * (G + RING)>(av[G % RING], addr);
        slot(std::integral_constant<int, G>{});
    }
    template <class Slot, int... Gs>
    static __device__ __forceinline__ void groups(float4 (&av)[RING], f32x4 (&acc)[4], const float (&w)[KH], uint32_t addr,
                                                  Slot& slot, std::integer_sequence<int, Gs...>) {
        (group<Gs>(av, acc, w, addr, slot), ...);
    }
    template <int... Rs>
    static __device__ __forceinline__ void prologue(float4 (&av)[RING], uint32_t addr, std::integer_sequence<int, Rs...>) {
        (lds_read16<16 * Rs>(av[Rs], addr), ...);
    }
    template <class Slot>
    static __device__ __forceinline__ void run(f32x4 (&acc)[4], const float (&w)[KH], uint32_t addr, Slot& slot) {
        float4 av[RING];
        prologue(av, addr, std::make_integer_sequence<int, RING>{});
        groups(av, acc, w, addr, slot, std::make_integer_sequence<int, NG>{});
    }
};

// ---------------------------------------------------------------------------------------------------
// forward.  gates[row][4H] holds W_ih x + b_ih on entry and the activated gates i,f,g,o on exit.
// hprev/cprev[first row of a sequence] hold h0/c0 (rnn_seed_state); h_t, c_t go to hseq/cseq[row] and
// to hprev/cprev[row+1] (the shifted copies the weight-gradient GEMM and the backward read).
//
// Per step the critical path is product -> exchange -> gate maths -> h to LDS -> barrier.  Everything
// else - next step's gate pre-activation loads and the PREVIOUS step's global stores (results wait in
// registers for one step) - is issued from the slots between the MFMA groups, where it costs nothing.
// Loads of cells past their sequence end read a clamped (valid) row instead of branching.  Steps are
// unrolled by two with ping-pong registers, so the prefetched values are never copied (a copy would
// have to wait for the loads, and with them for every younger store, at the end of each step).
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
    float c[2];
    float* gp[2];     // &gates[row0][u]
    size_t so[2];     // offset of [row0][u] in the [rows][H] state arrays
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
        const size_t row0 = len[cc] > 0 ? (size_t)p.seq_off[b] : 0;
        gp[cc] = p.gates + row0 * (size_t)(4 * H) + u;
        so[cc] = row0 * H + u;
        c[cc] = len[cc] > 0 ? p.cprev[so[cc]] : 0.f;
    }
    // h0 -> LDS buffer 0
    for (int e = tid; e < 4 * H; e += C::THREADS) {
        const int q = e / H, j = e - q * H;
        const int b = b0 + q;
        const bool on = b < p.n_seq && p.seq_len[b] > 0;
        h_lds[0][q][j] = on ? p.hprev[(size_t)p.seq_off[b] * H + j] : 0.f;
    }
    // gate pre-activations of step 0 (clamped rows: always a valid address)
    float xc[2][4], xn[2][4];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int g = 0; g < 4; ++g) xc[cc][g] = gp[cc][g * H];
    // results of the previous step, stored one step late
    float sv[2][6];   // i, f, g, o, c, h
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int q = 0; q < 6; ++q) sv[cc][q] = 0.f;
    __syncthreads();

    // one step; xcur = this step's gate pre-activations, xnext receives the next step's
    auto step = [&](const int t, float (&xcur)[2][4], float (&xnext)[2][4], auto CUR) {
        constexpr int cur = decltype(CUR)::value;
        // slot work: loads for step t+1 (first: they get the whole product phase to land), then the
        // stores of step t-1
        auto slot = [&](auto G) {
            constexpr int g = decltype(G)::value;
            constexpr int per = FwdProduct<H>::NG / 16 > 0 ? FwdProduct<H>::NG / 16 : 1;   // 16 items over NG slots
            if constexpr (g % per == 0 && g / per < 16) {
                constexpr int it = g / per;
                if constexpr (it < 8) {
                    constexpr int cc = it >> 2, gg = it & 3;
                    const int tl = min(t + 1, max(len[cc] - 1, 0));
                    xnext[cc][gg] = gp[cc][(size_t)tl * (4 * H) + gg * H];
                } else {
                    constexpr int q = it - 8;
                    constexpr int cc = q >> 2, part = q & 3;
                    const int tp = t - 1;
                    if (tp >= 0 && tp < len[cc]) {
                        if constexpr (part < 2) {
                            float* gt = gp[cc] + (size_t)tp * (4 * H);
                            gt[(2 * part) * H] = sv[cc][2 * part];
                            gt[(2 * part + 1) * H] = sv[cc][2 * part + 1];
                        } else if constexpr (part == 2) {
                            p.cseq[so[cc] + (size_t)tp * H] = sv[cc][4];
                            p.hseq[so[cc] + (size_t)tp * H] = sv[cc][5];
                        } else {
                            if (tp + 1 < len[cc]) {
                                p.cprev[so[cc] + (size_t)(tp + 1) * H] = sv[cc][4];
                                p.hprev[so[cc] + (size_t)(tp + 1) * H] = sv[cc][5];
                            }
                        }
                    }
                }
            }
        };

        // ---- recurrent product: rows = the 4 sequences, this lane's two columns ----------------------
        f32x4 pa[4];   // chains: [0] col0 even k, [1] col1 even k, [2] col0 odd k, [3] col1 odd k
        FwdProduct<H>::run(pa, w0, w1, lds_addr(&h_lds[cur][lane & 3][0]), slot);
        f32x4 acc0 = pa[0] + pa[2], acc1 = pa[1] + pa[3];   // lanes hi=0: (i,f) of 4 sequences; hi=1: (g,o)

        // ---- cross-half exchange: afterwards every lane holds i,f,g,o of its own two cells ------------
        float ri[2], rf[2], rg[2], ro[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            float y0 = acc0[cc], x0 = acc0[2 + cc], y1 = acc1[cc], x1 = acc1[2 + cc];
            half_swap(y0, x0);   // y0 = i of own cell, x0 = g of own cell
            half_swap(y1, x1);   // y1 = f,             x1 = o
            ri[cc] = y0; rg[cc] = x0; rf[cc] = y1; ro[cc] = x1;
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const bool on = t < len[cc];
            const float ig = fast_sigmoid(xcur[cc][0] + (ri[cc] + bh[0]));
            const float fg = fast_sigmoid(xcur[cc][1] + (rf[cc] + bh[1]));
            const float gg = fast_tanh(xcur[cc][2] + (rg[cc] + bh[2]));
            const float og = fast_sigmoid(xcur[cc][3] + (ro[cc] + bh[3]));
            const float cn = fg * c[cc] + ig * gg;
            const float hn = og * fast_tanh(cn);
            h_lds[cur ^ 1][2 * hi + cc][u] = on ? hn : 0.f;
            c[cc] = on ? cn : c[cc];
            sv[cc][0] = ig; sv[cc][1] = fg; sv[cc][2] = gg; sv[cc][3] = og; sv[cc][4] = cn; sv[cc][5] = hn;
        }
        __syncthreads();
    };

    for (int t = 0; t < tmax; t += 2) {
        step(t, xc, xn, std::integral_constant<int, 0>{});
        if (t + 1 < tmax) step(t + 1, xn, xc, std::integral_constant<int, 1>{});
    }
    // drain: the deferred stores of the last step
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int tp = tmax - 1;
        if (tp >= 0 && tp < len[cc]) {
            float* gt = gp[cc] + (size_t)tp * (4 * H);
            gt[0] = sv[cc][0]; gt[H] = sv[cc][1]; gt[2 * H] = sv[cc][2]; gt[3 * H] = sv[cc][3];
            p.cseq[so[cc] + (size_t)tp * H] = sv[cc][4];
            p.hseq[so[cc] + (size_t)tp * H] = sv[cc][5];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward through time.  In: dh[row][H] = dL/dh_t from the layer above (heads), the forward's saved
// gates / cseq / cprev.  Out: dgx[row][4H] = gradient w.r.t. the gate pre-activations (for an LSTM the
// same tensor serves W_ih x + b_ih and W_hh h + b_hh).  Same slot discipline as the forward: the
// operands of step t-1 are loaded, and the gate gradients of step t+1 stored, between the MFMA groups.
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
    const float* gp[2];
    float* dgp[2];
    size_t so[2];
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
        const size_t row0 = len[cc] > 0 ? (size_t)p.seq_off[b] : 0;
        gp[cc] = p.gates + row0 * (size_t)(4 * H) + u;
        dgp[cc] = p.dgx + row0 * (size_t)(4 * H) + u;
        so[cc] = row0 * H + u;
    }
    for (int e = tid; e < 4 * C::GLD; e += C::THREADS) (&g_lds[0][0][0])[e] = 0.f;   // "step tmax" has no gradient

    // per-cell carried state, the operands of the step being processed, and the next ones in flight
    float dc_next[2] = {0.f, 0.f}, f_next[2] = {0.f, 0.f};
    float cur_v[2][7], nxt_v[2][7];   // i, f, g, o, c, c_prev, dh
    float sv[2][4];                   // gate gradients of the previous iteration (step t+1), stored one step late
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int tl = max(min(tmax - 1, len[cc] - 1), 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) cur_v[cc][g] = gp[cc][(size_t)tl * (4 * H) + g * H];
        cur_v[cc][4] = p.cseq[so[cc] + (size_t)tl * H];
        cur_v[cc][5] = p.cprev[so[cc] + (size_t)tl * H];
        cur_v[cc][6] = p.dh[so[cc] + (size_t)tl * H];
#pragma unroll
        for (int g = 0; g < 4; ++g) sv[cc][g] = 0.f;
    }
    __syncthreads();

    auto step = [&](const int t, float (&cv)[2][7], float (&nv)[2][7], auto CUR) {
        constexpr int cur = decltype(CUR)::value;
        auto slot = [&](auto G) {
            constexpr int g = decltype(G)::value;
            constexpr int per = BwdProduct<KH>::NG / 32 > 0 ? BwdProduct<KH>::NG / 32 : 1;   // 22 items over NG slots
            if constexpr (g % per == 0 && g / per < 22) {
                constexpr int it = g / per;
                if constexpr (it < 14) {
                    constexpr int cc = it / 7, q = it % 7;
                    const int tl = max(min(t - 1, len[cc] - 1), 0);
                    if constexpr (q < 4) nv[cc][q] = gp[cc][(size_t)tl * (4 * H) + q * H];
                    else if constexpr (q == 4) nv[cc][4] = p.cseq[so[cc] + (size_t)tl * H];
                    else if constexpr (q == 5) nv[cc][5] = p.cprev[so[cc] + (size_t)tl * H];
                    else nv[cc][6] = p.dh[so[cc] + (size_t)tl * H];
                } else {
                    constexpr int q = it - 14;
                    constexpr int cc = q >> 2, gg = q & 3;
                    const int tp = t + 1;
                    if (tp < len[cc]) dgp[cc][(size_t)tp * (4 * H) + gg * H] = sv[cc][gg];
                }
            }
        };

        // ---- dh_rec[seq][u] = sum_k dgates_{t+1}[seq][k] * W_hh[k][u], this half's k range ------------
        f32x4 pa[4];
        BwdProduct<KH>::run(pa, w, lds_addr(&g_lds[cur][lane & 3][KH * hi]), slot);
        const f32x4 acc = (pa[0] + pa[1]) + (pa[2] + pa[3]);
        float rec[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            float y = acc[cc], x = acc[2 + cc];
            half_swap(y, x);     // low lanes: own acc[cc] | partner's acc[cc]; high lanes: partner's acc[2+cc] | own
            rec[cc] = y + x;
        }

#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const bool on = t < len[cc];
            const bool has_next = (t + 1) < len[cc];
            float dh = cv[cc][6];
            if (has_next) dh += rec[cc];
            const float ig = cv[cc][0], fg = cv[cc][1], gg = cv[cc][2], og = cv[cc][3];
            const float tc = fast_tanh(cv[cc][4]);
            float dcv = dh * og * (1.f - tc * tc);
            if (has_next) dcv += dc_next[cc] * f_next[cc];
            const float di = dcv * gg * ig * (1.f - ig);
            const float df = dcv * cv[cc][5] * fg * (1.f - fg);
            const float dg = dcv * ig * (1.f - gg * gg);
            const float dO = dh * tc * og * (1.f - og);
            float* gl = &g_lds[cur ^ 1][2 * hi + cc][0];
            gl[u] = on ? di : 0.f; gl[H + u] = on ? df : 0.f; gl[2 * H + u] = on ? dg : 0.f; gl[3 * H + u] = on ? dO : 0.f;
            sv[cc][0] = di; sv[cc][1] = df; sv[cc][2] = dg; sv[cc][3] = dO;
            if (on) { dc_next[cc] = dcv; f_next[cc] = fg; }
        }
        __syncthreads();
    };

    for (int t = tmax - 1; t >= 0; t -= 2) {
        step(t, cur_v, nxt_v, std::integral_constant<int, 0>{});
        if (t - 1 >= 0) step(t - 1, nxt_v, cur_v, std::integral_constant<int, 1>{});
    }
    // drain the deferred stores of step 0
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
        if (0 < len[cc]) {
#pragma unroll
            for (int g = 0; g < 4; ++g) dgp[cc][g * H] = sv[cc][g];
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
