// One env-step of one hero as ONE kernel: the actor's call (policy.py:80-84 `Policy.single`, agent.py:652).
//
// Replaces, for B = 1 and S = 1, /root/reference/policy.py:92-167 (Policy.forward): unit embeddings (policy.py:100-105), max-pools and the
// env embedding (97, 118-127), the pre-rnn projection (138), the recurrent cell step (141), the heads and the target-unit attention
// (144-155).  Until round 6 `Policy.single` replayed the BATCH path's ~14 launches on 128-row padded tiles as one hipGraph (152 us per
// env-step, 290 us eager: VERDICT r5, missing item 4).  A single step is a chain of matrix-VECTOR products over 3.6 MB of weights: nothing for
// the matrix cores, everything for latency - so the kernel is laid out around the number of DEPENDENT memory round trips:
//   * 64 co-resident workgroups of 8 waves (256 VGPRs each: a stage's weight rows wait in registers); a wave per output row: the row as 16-byte loads along K (coalesced), one DPP wave reduction;
//     exact f32 fmas;
//   * the vectors that pass between the stages are {value, tag} granules in the caller's scratch (team_util.h's protocol: 8-byte relaxed
//     device-scope stores, consumers poll exactly the words they need; tag = a per-launch generation) - there is NO grid barrier: a hop
//     costs one store becoming visible plus one poll, and no wave waits for its stores to drain;
//   * a stage's weight rows do not depend on the stage before it: they are requested BEFORE the wave starts polling for its input, so
//     the weights' round trip hides behind the hop (the first version - stage after stage behind a ticket barrier, each row requested
//     when it was needed - took 54 us; see profiles/r06/policy_single.txt).
// Stages (hops between them):
//   A  basic[u] = relu(W1 x_u + b1) of all 40 units into LDS (every workgroup for itself: 61 k MACs), then a wave per (unit type, column):
//      emb[u][c] = W2_t[c] . basic[u] + b2_t[c] for the type's units and their max (policy.py:118-127; slot 6 = the enh max again, :127)
//   B  pre = relu(W_pre xcat + b): a row per wave (256 waves)
//   C  per recurrent layer: a wave per hidden unit - its G gate rows over [x | h], then the cell (LSTM i, f, g, o / GRU r, z, n)
//   D  workgroup 0: the query rows (128) into LDS, then tu[u] = q . emb[u] (emb polled into LDS long before); workgroup 1: the other
//      26 head rows
// Output: out[0..160) = the headout row (DC_WS_HEADOUT's columns, pad columns zero), out[160..200) = the target-unit logits; hT / cT.
#include "../../include/dotaclient_hip.h"
#include "team_util.h"

namespace dc {
namespace {

enum { PS_WG = 64, PS_THREADS = 512, PS_WPB = 8, PS_WAVES = PS_WG * PS_WPB, PS_OBS = 483, PS_EMB = 128, PS_XCAT = 896, PS_PRE = 256,
       PS_HO = 160, PS_HON = 154, PS_UNITS = 40, PS_HMAX = 512 };
// scratch as 8-byte granules: [0] the generation of the last finished launch, then the vectors
enum { PS_G_EMB = 8, PS_G_XCAT = PS_G_EMB + PS_UNITS * PS_EMB, PS_G_PRE = PS_G_XCAT + PS_XCAT, PS_G_H = PS_G_PRE + PS_PRE,
       PS_G_WORDS = PS_G_H + DC_MAX_LAYERS * PS_HMAX };
static_assert(2 * PS_G_WORDS <= DC_SINGLE_SCRATCH_FLOATS, "scratch documented in the header");
static_assert(PS_THREADS >= PS_OBS, "one thread per observation float");
constexpr int PS_SPIN_BUDGET = 1 << 20;      // polls a thread may spend in one launch before it gives up and hands on NaN (no hang)

struct SingleArgs {
    const float* params;
    long long off[DC_P_RNN0 + 4 * DC_MAX_LAYERS];
    const float* obs; const float* h0; const float* c0;
    float* out; float* hT; float* cT; u64* gran;
    int cell, H, layers;
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// sum over the 64 lanes, the same value in every lane: four steps inside the rows of 16 (quad_perm, row_half_mirror, row_mirror), then
// the four row sums through readlane - no LDS crossbar (common.h's wave_sum is six ds_bpermute round trips)
__device__ __forceinline__ float ps_wave_sum(float v) {
    v = dpp_add<0xB1, 0xf>(v);       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);       // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);      // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);      // row_mirror
    return (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48)));
}

// a weight row along K as 16-byte pieces: piece i of lane l covers k = 256 i + 4 l.  Buffer loads with the row as the buffer: pieces
// beyond K come back as zeros without a branch or a memory access (w wave-uniform, K a multiple of 4)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int N>
__device__ __forceinline__ void row_load(f32x4 (&r)[N], const float* w, int K, int lane) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, K * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 1024 + lane * 16, 0, 0);
        r[i] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
    }
}
// (the vector in LDS is zero beyond K: the row's pieces there are zeros too)
template <int N>
__device__ __forceinline__ float row_dot(const f32x4 (&r)[N], const float* x_lds, int lane) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(x_lds + i * 256 + lane * 4);
        s = fmaf(r[i][3], x[3], fmaf(r[i][2], x[2], fmaf(r[i][1], x[1], fmaf(r[i][0], x[0], s))));
    }
    return s;
}

__device__ __forceinline__ void pub(u64* g, float v, unsigned tag) { granule_store(g, v, tag); }
// the granule's value once it carries this launch's tag; NaN when the thread's poll budget is spent (the producers never ran)
__device__ __forceinline__ float poll(const u64* g, unsigned tag, int& budget) {
    u64 x = granule_load(g);
    while ((unsigned)(x >> 32) != tag) {
        if (--budget <= 0) return __builtin_nanf("");
        __builtin_amdgcn_s_sleep(1);
        x = granule_load(g);
    }
    return __uint_as_float((unsigned)x);
}

// stage A for one (type, column): the type's NU units against one row of W2_t; lanes 0 .. NU-1 publish emb, the max goes to xcat
template <int NU>
__device__ __forceinline__ void emb_item(const float2 w, const float b2, const float* sb, int u0, int c, int slot_a, int slot_b, u64* G, unsigned tag,
                                         int lane) {
    float mine = 0.f, m = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const float2 x = *reinterpret_cast<const float2*>(sb + (u0 + u) * PS_EMB + lane * 2);
        const float s = ps_wave_sum(fmaf(w.y, x.y, w.x * x.x)) + b2;
        if (lane == u) mine = s;
        m = u == 0 ? s : max_nan(m, s);
    }
    if (lane < NU) pub(G + PS_G_EMB + (u0 + lane) * PS_EMB + c, mine, tag);
    if (lane == 0 && slot_a >= 0) pub(G + PS_G_XCAT + slot_a * PS_EMB + c, m, tag);
    if (lane == 0 && slot_b >= 0) pub(G + PS_G_XCAT + slot_b * PS_EMB + c, m, tag);
}

}  // namespace

__global__ __launch_bounds__(PS_THREADS) void policy_single_kernel(SingleArgs a) {
    __shared__ __attribute__((aligned(16))) float sh_b[PS_UNITS * PS_EMB];     // stage A: basic of every unit; workgroup 0 later: emb
    __shared__ __attribute__((aligned(16))) float sh_x[2][1024];               // a stage's input vector (ping-pong: one barrier per stage)
    __shared__ __attribute__((aligned(16))) float sh_h[DC_MAX_LAYERS * PS_HMAX];   // h0 of every layer
    __shared__ __attribute__((aligned(16))) float sh_q[PS_EMB];
    __shared__ float sh_obs[PS_OBS + 1];                                       // the observation row: read ONCE per workgroup (it may be pinned host memory)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gw = blockIdx.x * PS_WPB + wave;                  // wave number in the grid
    const float* const P = a.params;
    u64* const G = a.gran;
    const int H = a.H, NG = a.cell == 1 ? 4 : 3, L = a.layers;
    int budget = PS_SPIN_BUDGET;

    // ---- everything that depends on nothing: requested up front ---------------------------------------------------------------------------
    const unsigned tag = (unsigned)__hip_atomic_load(G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;      // this launch's generation
    for (int e = tid; e < L * PS_HMAX; e += PS_THREADS) {                      // (zero beyond H: see row_dot)
        const int l = e / PS_HMAX, j = e % PS_HMAX;
        sh_h[e] = (a.h0 && j < H) ? a.h0[l * H + j] : 0.f;
    }
    if (tid < PS_OBS) sh_obs[tid] = a.obs[tid];
    // A: this thread's basic values: channel k of units ug, ug + 4, ..  (a wave = one unit: its 12 stats are scalar loads)
    const int bk = tid & 127, ug = __builtin_amdgcn_readfirstlane(tid >> 7);          // 0 .. 3
    f32x4 w1[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) w1[i] = *reinterpret_cast<const f32x4*>(P + a.off[DC_P_BASIC_W] + bk * 12 + i * 4);
    const float b1 = P[a.off[DC_P_BASIC_B] + bk];
    // A: this wave's (type, column) items
    int it_t[2] = {0, 0};
    const int ic = gw & 127;
    if (gw < 256) { it_t[0] = 2 + (gw >> 7); }                                   // anh / enh: 16 units
    else if (gw < 384) { it_t[0] = 1; it_t[1] = 0; }                             // eh (5 units) + ah (1)
    else { it_t[0] = 4; it_t[1] = 5; }                                           // ath + eth (1 each) (+ the env embedding)
    float2 w2[2]; float b2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        w2[i] = *reinterpret_cast<const float2*>(P + a.off[DC_P_UNIT_W] + ((size_t)it_t[i] * PS_EMB + ic) * PS_EMB + lane * 2);
        b2[i] = P[a.off[DC_P_UNIT_B] + it_t[i] * PS_EMB + ic];
    }
    // B: this wave's row of W_pre
    f32x4 wb[4]; float bb = 0.f;
    if (gw < PS_PRE) { row_load(wb, P + a.off[DC_P_PRE_W] + (size_t)gw * PS_XCAT, PS_XCAT, lane); bb = P[a.off[DC_P_PRE_B] + gw]; }

    // ---- A ---------------------------------------------------------------------------------------------------------------------------
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int u = ug + 4 * i;
        const float* x = sh_obs + 3 + u * 12;
        float s = b1;
#pragma unroll
        for (int f = 0; f < 12; ++f) s = fmaf(w1[f >> 2][f & 3], x[f], s);
        sh_b[u * PS_EMB + bk] = relu_nan(s);
    }
    __syncthreads();
    if (gw < 256) emb_item<16>(w2[0], b2[0], sh_b, it_t[0] == 2 ? 6 : 22, ic, it_t[0] + 1, it_t[0] == 3 ? 6 : -1, G, tag, lane);
    else if (gw < 384) { emb_item<5>(w2[0], b2[0], sh_b, 1, ic, 2, -1, G, tag, lane); emb_item<1>(w2[1], b2[1], sh_b, 0, ic, 1, -1, G, tag, lane); }
    else {
        emb_item<1>(w2[0], b2[0], sh_b, 38, ic, 5, -1, G, tag, lane);
        emb_item<1>(w2[1], b2[1], sh_b, 39, ic, -1, -1, G, tag, lane);
        if (lane == 0) {                                                         // policy.py:97
            const float* we = P + a.off[DC_P_ENV_W] + ic * 3;
            pub(G + PS_G_XCAT + ic, relu_nan(fmaf(sh_obs[2], we[2], fmaf(sh_obs[1], we[1], fmaf(sh_obs[0], we[0], P[a.off[DC_P_ENV_B] + ic])))), tag);
        }
    }

    // ---- the weight rows of a recurrent layer / of the heads, requested a hop ahead of their input ----------------------------------------------
    f32x4 wr[16];                                                               // C: [gate][x piece 0, 1 | h piece 0, 1]
    float br[8]; float cprev = 0.f;
    auto load_layer = [&](int l) {
        const int in = l == 0 ? PS_PRE : H;
        if (gw < H) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < NG) {
                    f32x4 t[2];
                    row_load(t, P + a.off[DC_P_RNN0 + 4 * l] + (size_t)(g * H + gw) * in, in, lane); wr[g * 4] = t[0]; wr[g * 4 + 1] = t[1];
                    row_load(t, P + a.off[DC_P_RNN0 + 4 * l + 1] + (size_t)(g * H + gw) * H, H, lane); wr[g * 4 + 2] = t[0]; wr[g * 4 + 3] = t[1];
                    br[g] = P[a.off[DC_P_RNN0 + 4 * l + 2] + g * H + gw];
                    br[4 + g] = P[a.off[DC_P_RNN0 + 4 * l + 3] + g * H + gw];
                }
            }
            if (a.cell == 1) cprev = a.c0 ? a.c0[(size_t)l * H + gw] : 0.f;
        }
    };
    load_layer(0);

    // ---- B: pre = relu(W_pre xcat + b) ------------------------------------------------------------------------------------------------------
    for (int e = tid; e < 1024; e += PS_THREADS) sh_x[0][e] = e < PS_XCAT ? poll(G + PS_G_XCAT + e, tag, budget) : 0.f;
    __syncthreads();
    if (gw < PS_PRE) {
        const float s = ps_wave_sum(row_dot(wb, sh_x[0], lane));
        if (lane == 0) pub(G + PS_G_PRE + gw, relu_nan(s + bb), tag);
    }
    if (blockIdx.x == 0) {            // workgroup 0 keeps every unit's embedding for D (sh_b's readers are all past the barrier above)
        for (int e = tid; e < PS_UNITS * PS_EMB; e += PS_THREADS) sh_b[e] = poll(G + PS_G_EMB + e, tag, budget);
    }

    // ---- C: the recurrent layers ------------------------------------------------------------------------------------------------------------
    int buf = 1;
    for (int l = 0; l < L; ++l, buf ^= 1) {
        const int in = l == 0 ? PS_PRE : H;
        const u64* const src = l == 0 ? G + PS_G_PRE : G + PS_G_H + (l - 1) * PS_HMAX;
        sh_x[buf][tid] = tid < in ? poll(src + tid, tag, budget) : 0.f;
        __syncthreads();
        if (gw < H) {
            const float* const hl = sh_h + l * PS_HMAX;
            float gx[4], gh[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < NG) {
                    const f32x4 tx[2] = {wr[g * 4], wr[g * 4 + 1]}, th[2] = {wr[g * 4 + 2], wr[g * 4 + 3]};
                    gx[g] = ps_wave_sum(row_dot(tx, sh_x[buf], lane)) + br[g];
                    gh[g] = ps_wave_sum(row_dot(th, hl, lane)) + br[4 + g];
                } else { gx[g] = gh[g] = 0.f; }
            }
            const float hp = hl[gw];
            float hn, cn = 0.f;
            if (a.cell == 1) {                                              // LSTM (torch order i, f, g, o)
                const float ig = sigmoidf_(gx[0] + gh[0]), fg = sigmoidf_(gx[1] + gh[1]), gg = tanhf(gx[2] + gh[2]), og = sigmoidf_(gx[3] + gh[3]);
                cn = fg * cprev + ig * gg;
                hn = og * tanhf(cn);
            } else {                                                        // GRU (r, z, n): n = tanh(W_in x + b_in + r (W_hn h + b_hn))
                const float r = sigmoidf_(gx[0] + gh[0]), z = sigmoidf_(gx[1] + gh[1]);
                const float n = tanhf(gx[2] + r * gh[2]);
                hn = (1.f - z) * n + z * hp;
            }
            if (lane == 0) {
                pub(G + PS_G_H + l * PS_HMAX + gw, hn, tag);
                a.hT[(size_t)l * H + gw] = hn;
                if (a.cell == 1 && a.cT) a.cT[(size_t)l * H + gw] = cn;
            }
        }
        if (l + 1 < L) load_layer(l + 1);
    }

    // ---- D: the heads (154 rows; the row's pad columns are zeros), the target-unit logits (policy.py:144-155) ------------------------------------
    if (blockIdx.x > 1) return;
    if (blockIdx.x == 1) {                                                      // rows 128 + wave + 8 i
        f32x4 wd[8]; float bd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = PS_EMB + i * 8 + wave;
            f32x4 t[2];
            row_load(t, P + a.off[DC_P_HEADS_W] + (size_t)r * H, r < PS_HON ? H : 0, lane); wd[i * 2] = t[0]; wd[i * 2 + 1] = t[1];
            bd[i] = r < PS_HON ? P[a.off[DC_P_HEADS_B] + r] : 0.f;
        }
        sh_x[buf][tid] = tid < H ? poll(G + PS_G_H + (L - 1) * PS_HMAX + tid, tag, budget) : 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = PS_EMB + i * 8 + wave;
            const f32x4 t[2] = {wd[i * 2], wd[i * 2 + 1]};
            const float s = ps_wave_sum(row_dot(t, sh_x[buf], lane)) + bd[i];
            if (lane == 0) a.out[r] = s;                                        // (rows 154 .. 159: zero weights, zero bias)
        }
        return;
    }
    {
        f32x4 wd[32]; float bd[16];                                             // query rows 16 wave .. 16 wave + 15
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f32x4 t[2];
            row_load(t, P + a.off[DC_P_HEADS_W] + (size_t)(wave * 16 + i) * H, H, lane); wd[i * 2] = t[0]; wd[i * 2 + 1] = t[1];
            bd[i] = P[a.off[DC_P_HEADS_B] + wave * 16 + i];
        }
        sh_x[buf][tid] = tid < H ? poll(G + PS_G_H + (L - 1) * PS_HMAX + tid, tag, budget) : 0.f;
        __syncthreads();
        float mine = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f32x4 t[2] = {wd[i * 2], wd[i * 2 + 1]};
            const float s = ps_wave_sum(row_dot(t, sh_x[buf], lane)) + bd[i];
            if (lane == i) mine = s;
        }
        if (lane < 16) { sh_q[wave * 16 + lane] = mine; a.out[wave * 16 + lane] = mine; }
    }
    __syncthreads();
    const float2 q = *reinterpret_cast<const float2*>(sh_q + lane * 2);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int u = wave + 8 * i;
        if (u < PS_UNITS) {
            const float2 e = *reinterpret_cast<const float2*>(sh_b + u * PS_EMB + lane * 2);
            const float s = ps_wave_sum(fmaf(q.y, e.y, q.x * e.x));
            if (lane == 0) a.out[PS_HO + u] = s;
        }
    }
    // the generation moves on when the last stage is through: every workgroup read it long ago (D's input needed all of them)
    if (tid == 0) __hip_atomic_store(G, (u64)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Supported: H a multiple of 64 up to 512, up to DC_MAX_LAYERS layers.  scratch: DC_SINGLE_SCRATCH_FLOATS floats (8-byte aligned), ZERO before
// the first call, then left to this function (granules and the launch generation).  One call at a time per scratch buffer.
int policy_single(const dc_dims* d, const float* params, const int64_t* poff, const float* obs, const float* h0, const float* c0, float* out,
                  float* hT, float* cT, float* scratch, hipStream_t s) {
    if (d->layers < 1 || d->layers > DC_MAX_LAYERS) { set_error("policy_single: layers out of range", 1020); return 1020; }
    if (d->cell != 0 && d->cell != 1) { set_error("policy_single: cell must be 0 (gru) or 1 (lstm)", 1021); return 1021; }
    if (d->hidden < 64 || d->hidden > PS_HMAX || d->hidden % 64) { set_error("policy_single: hidden must be a multiple of 64 up to 512", 1022); return 1022; }
    if (!out || !hT || !scratch || !obs || !params) { set_error("policy_single: null buffer", 1024); return 1024; }
    if (((uintptr_t)scratch & 7) || ((uintptr_t)params & 15)) { set_error("policy_single: scratch must be 8-byte, params 16-byte aligned", 1025); return 1025; }
    SingleArgs a{};
    a.params = params;
    for (int i = 0; i < DC_P_RNN0 + 4 * d->layers; ++i) a.off[i] = poff[i];
    a.obs = obs; a.h0 = h0; a.c0 = c0; a.out = out; a.hT = hT; a.cT = cT; a.gran = reinterpret_cast<u64*>(scratch);
    a.cell = d->cell; a.H = d->hidden; a.layers = d->layers;
    hipLaunchKernelGGL(policy_single_kernel, dim3(PS_WG), dim3(PS_THREADS), 0, s, a);
    return launch_check("policy_single");
}

}  // namespace dc
