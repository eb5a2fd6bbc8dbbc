// Whole-network orchestration: forward, loss seeds -> backward, on a caller-provided workspace.
//
// Replaces /root/reference/policy.py:92-167 (Policy.forward) and the reverse pass torch autograd
// builds for it at /root/reference/optimizer.py:672.  Every matrix product goes through gemm.hip
// (exact-fp32 MFMA); the sequential recurrent steps through rnn.hip; the HBM-bound glue through
// embed.hip / heads.hip.  Nothing is allocated here: `ws` is one device buffer laid out by
// workspace_layout() below (sizes depend only on dc_dims), so a whole epoch can be replayed as a
// hipGraph by the caller.
#include "../../include/dotaclient_hip.h"
#include <math.h>
#include <stdlib.h>
#include "kernels.h"

namespace dc {

static const int T_UNITS[6] = {1, 5, 16, 16, 1, 1};
static const int T_CUM[7] = {0, 1, 6, 22, 38, 39, 40};
enum { EMBW = 128, XCATW = 896, PREW = 256, HO_LD = 160, HO_N = 154, DC_SCRATCH_FLOATS = 16 << 20 };

// parameter offsets (floats) inside the flat buffer, in the order of dc_param_index (header)
struct Params {
    const float* base;
    const int64_t* off;
    const float* p(int i) const { return base + off[i]; }
};
struct Grads {
    float* base;
    const int64_t* off;
    float* p(int i) const { return base + off[i]; }
};

static inline int64_t align_up(int64_t x) { return (x + 255) / 256 * 256; }

// DC_DIMS_F16X2 (gemm_x3.hip, PREC = 4): the fixed power-of-two pre-scales of the three kinds of operand.  Activations of this
// network are O(1) (relu / tanh outputs, unit-variance inputs): 2^4 keeps their second f16 piece normal down to |x| ~ 2^-7 and
// overflows beyond 4094; weights 2^8 (|w| < 255); gradients of a MEAN loss over `rows` env-steps are O(1 / rows): 2^(ceil(log2
// rows) + 2) brings them to O(1) (overflow: an entry beyond ~ 16384 / rows).
static constexpr float F16X2_S_ACT = 16.f, F16X2_S_W = 256.f;
static inline float f16x2_grad_scale(long long rows) {
    int e = 0;
    while ((1LL << e) < rows && e < 40) ++e;
    return ldexpf(1.f, e + 2);
}

// The fused embedding kernels work on 128-row tiles of the type-major emb / d(emb) blocks: the blocks are laid out for the row
// count padded to a multiple of 128 (padding steps re-read the last real step; nobody reads their results, their gradients are
// zero).  DC_DIMS_EMBED_UNFUSED: the layer-by-layer path, exact row count.
static inline bool embed_fused_on(const dc_dims* d) { return !(d->flags & DC_DIMS_EMBED_UNFUSED) && d->rows > 0; }
static inline int64_t emb_rows(const dc_dims* d) { return embed_fused_on(d) ? (d->rows + 127) / 128 * 128 : d->rows; }

// bf16 STORAGE on configs[4]'s path (DC_DIMS_BF16, LSTM-512 on the persistent team kernels): the gate pre-activations / activated gates
// and the gate gradients - [rows][4H] each, two thirds of the bytes the dense products and the recurrent kernels move - and `pre`,
// `hseq`, `hprev` live in HBM as bf16 (in the lower half of their f32-sized workspace buffers: the layout does not change).  Their
// consumers round them to bf16 anyway (they are MFMA operands: for pre / hseq / hprev the stored form changes no result at all); what is
// new is the rounding of the input projections before the cell adds W_hh h, and of the gate activations / gate gradients the backward reads.
// DC_DIMS_BF16_F32_STORE keeps f32 storage (A/B, and the comparison with the launch-per-step kernels).
static inline bool bf16_store(const dc_dims* d) {
    // rows % 16: the weight gradients contract over the rows, and the split-on-load kernel - the only one that reads bf16 buffers - needs
    // whole K = 16 steps (gemm_x3_shape_ok); other row counts keep f32 storage and with it the f32 fallback products
    return (d->flags & DC_DIMS_BF16) && !(d->flags & (DC_DIMS_BF16_F32_STORE | DC_DIMS_RNN_PER_STEP | DC_DIMS_RNN_STEP_BF16 | DC_DIMS_GEMM_FASTTILE)) &&
           d->cell == 1 && d->hidden == 512 && (d->rows % 16) == 0 && 4 * (long long)d->n_seq <= d->rows && lstm_team512_supported(1, 512, d->flags, d);
}

bool policy_bf16_store(const dc_dims* d) { return bf16_store(d); }

// Weight matrices the dense products read as pre-split bf16 planes (gemm_x3.hip), in this order: affine_pre_rnn [256][896],
// the recurrent input projections [G*H][in_l], the head block zero-padded to [160][H].  Elements of one orientation.
static inline int64_t wplane_elems(const dc_dims* d) {
    const int64_t H = d->hidden, G = d->cell == 0 ? 3 : 4;
    int64_t e = (int64_t)PREW * XCATW + (int64_t)HO_LD * H + (int64_t)6 * EMBW * EMBW;   // + the six unit-type matrices
    for (int l = 0; l < d->layers; ++l) e += G * H * (l == 0 ? PREW : H) + G * H * H;   // W_ih and W_hh (bf16 recurrence steps)
    return e;
}
struct WPlanes {              // where each matrix's planes start inside DC_WS_WPLANES (element offsets per plane set)
    uint16_t* base;           // forward orientation at base, transposed orientation at base + 3 * total
    int64_t total, pre, heads, unit, ih[DC_MAX_LAYERS], hh[DC_MAX_LAYERS];
    uint16_t* fwd(int64_t off) const { return base + 3 * off; }                  // planes of one matrix are contiguous: [3][rows][cols]
    uint16_t* bwd(int64_t off) const { return base + 3 * total + 3 * off; }
};
static WPlanes wplanes_of(const dc_dims* d, char* ws_base, const int64_t* off) {
    WPlanes w;
    w.base = reinterpret_cast<uint16_t*>(ws_base + off[DC_WS_WPLANES]);
    w.total = wplane_elems(d);
    const int64_t H = d->hidden, G = d->cell == 0 ? 3 : 4;
    int64_t o = 0;
    w.pre = o; o += (int64_t)PREW * XCATW;
    w.heads = o; o += (int64_t)HO_LD * H;
    w.unit = o; o += (int64_t)6 * EMBW * EMBW;
    for (int l = 0; l < d->layers; ++l) { w.ih[l] = o; o += G * H * (l == 0 ? PREW : H); }
    for (int l = 0; l < d->layers; ++l) { w.hh[l] = o; o += G * H * H; }
    return w;
}

// offsets (bytes) of every workspace buffer; returns total bytes.  out must hold
// DC_WS_FIXED + DC_WS_PER_LAYER * layers entries.
int64_t workspace_layout(const dc_dims* d, int64_t* out) {
    const int64_t NR = emb_rows(d), H = d->hidden, G = d->cell == 0 ? 3 : 4, B = d->n_seq;   // (every buffer sized for the padded row count)
    (void)B;
    int64_t off = 0;
    auto put = [&](int idx, int64_t bytes) { out[idx] = off; off = align_up(off + bytes); };
    put(DC_WS_FAULT, 8 * 4);                      // always at offset 0: the sticky fault record (header)
    put(DC_WS_BASIC, NR * 40 * EMBW * 4);
    put(DC_WS_EMB, NR * 40 * EMBW * 4);
    put(DC_WS_DEMB, NR * 40 * EMBW * 4);
    put(DC_WS_XCAT, NR * XCATW * 4);
    put(DC_WS_AMAX, NR * 3 * EMBW);
    put(DC_WS_PRE, NR * PREW * 4);
    put(DC_WS_HEADOUT, NR * HO_LD * 4);
    put(DC_WS_TU, NR * 40 * 4);
    put(DC_WS_DHEADOUT, NR * HO_LD * 4);
    put(DC_WS_DTU, NR * 40 * 4);
    put(DC_WS_DPRE, NR * PREW * 4);
    put(DC_WS_DXCAT, NR * XCATW * 4);
    put(DC_WS_STATS, 64 * 8 + 256 * 8 * 8 + 4096 * 12 * 8 + 64 + 32 * 8 + 32 * 12 * 8);   // totals + per-block partial sums of the two loss kernels + arrival counter (heads.hip: ST_*)
    put(DC_WS_WHHT, H * G * H * 4);
    put(DC_WS_SCRATCH, (int64_t)DC_SCRATCH_FLOATS * 4);   // two-stage reductions / split-K slabs
    put(DC_WS_HEADW_PAD, (int64_t)HO_LD * H * 4);          // head weights zero-padded to 160 rows (K of dH)
    // exchange ring of the team kernels: H = 256 (rnn_team.hip, rnn_team_mfma.hip), H = 512 (rnn_team512.hip: 64 MB for sixteen teams)
    put(DC_WS_TEAM_XBUF, H == 256 ? rnn_team_xbuf_bytes() : (H == 512 ? lstm_team512_xbuf_bytes() : 0));
    put(DC_WS_WPLANES, 2 * 3 * 2 * wplane_elems(d));              // weights as bf16 planes: forward + transposed orientation
    for (int l = 0; l < d->layers; ++l) {
        const int b = DC_WS_FIXED + l * DC_WS_PER_LAYER;
        put(b + DC_WSL_GATES, NR * G * H * 4);
        put(b + DC_WSL_HN, NR * H * 4);
        put(b + DC_WSL_HSEQ, NR * H * 4);
        put(b + DC_WSL_HPREV, NR * H * 4);
        put(b + DC_WSL_CSEQ, d->cell == 1 ? NR * H * 4 : 0);
        put(b + DC_WSL_CPREV, d->cell == 1 ? NR * H * 4 : 0);
        put(b + DC_WSL_DGX, NR * G * H * 4);
        put(b + DC_WSL_DGH, d->cell == 0 ? NR * G * H * 4 : 0);
        put(b + DC_WSL_DC, d->cell == 1 ? NR * H * 4 : 0);
        put(b + DC_WSL_DH, NR * H * 4);
    }
    return off;
}

long long emb_rows_of(const dc_dims* d) { return emb_rows(d); }

struct Ws {
    char* base;
    int64_t off[DC_WS_FIXED + DC_WS_PER_LAYER * DC_MAX_LAYERS];
    float* f(int i) const { return reinterpret_cast<float*>(base + off[i]); }
    float* fl(int l, int i) const { return f(DC_WS_FIXED + l * DC_WS_PER_LAYER + i); }
};

static int check_dims(const dc_dims* d) {
    if (d->layers < 1 || d->layers > DC_MAX_LAYERS) { set_error("dims: layers out of range", 1020); return 1020; }
    if (d->cell != 0 && d->cell != 1) { set_error("dims: cell must be 0 (gru) or 1 (lstm)", 1021); return 1021; }
    if (d->hidden != 64 && d->hidden != 128 && d->hidden != 256 && d->hidden != 512) { set_error("dims: hidden must be 64, 128, 256 or 512", 1022); return 1022; }
    if (d->rows * 40 * 128 >= (1LL << 31) * 8) { set_error("dims: too many rows for one call", 1023); return 1023; }
    return 0;
}

#define DC_TRY(x) do { int _e = (x); if (_e) return _e; } while (0)

// Default mode (fused embedding + f16x2 products): EVERY plane set a pass needs - the unit types' W2, the dense matrices, and unless the pass
// is forward-only their transposes for the backward (incl. W2_t^T of the 16-unit types) - comes from ONE pre-pass launch at the top of
// policy_forward (round 6: 14 -> 5 split launches per configs[2] step).  policy_backward then relies on the workspace holding them: it
// always follows a forward over the same parameters and flags (whose activations it reads from the same workspace anyway).
static bool planes_in_one_launch(const dc_dims* d) {
    return embed_fused_on(d) && (d->flags & DC_DIMS_F16X2) && !(d->flags & DC_DIMS_BF16) && !(d->flags & DC_DIMS_GEMM_FASTTILE) &&
           !(d->flags & DC_DIMS_GEMM_X3_ALL) && d->layers <= 4;
}

// The two 16-unit types go through the sparse max-pool backward (embed_sparse.hip: a sixteenth of the MACs, no d(emb)
// in HBM; 290 us against 385 us for the dense MFMA kernels on the same units at the bench batch).  DC_DIMS_DENSE_POOL_BWD
// forces the dense kernels for every type (per call: the GPU tests run both paths in one process).
static bool embed_sparse_enabled(const dc_dims* d) { return !(d->flags & DC_DIMS_DENSE_POOL_BWD); }

int policy_forward(const dc_dims* d, const float* params, const int64_t* poff, const float* obs, const float* h0,
                   const float* c0, const int64_t* seq_off, const int32_t* seq_len, void* ws_base, float* hT, float* cT,
                   const uint8_t* unit_mask, hipStream_t s) {
    DC_TRY(check_dims(d));
    if (d->rows <= 0 || d->n_seq <= 0) return 0;
    Ws w; w.base = (char*)ws_base; workspace_layout(d, w.off);
    Params P{params, poff};
    const long long NR = d->rows;
    const int H = d->hidden, G = d->cell == 0 ? 3 : 4, B = d->n_seq;

    // per-unit embedding MLP (policy.py:100-126).  Fused (rows % 128 == 0): layer 1 recomputed on chip inside
    // the layer-2 product; otherwise layer 1 on VALU into `basic`, layer 2 as six dense GEMMs
    uint8_t* amax = reinterpret_cast<uint8_t*>(w.base + w.off[DC_WS_AMAX]);
    const bool fused = embed_fused_on(d);
    const long long NRp = emb_rows(d);               // rows per unit of the type-major blocks
    bool pools_done = false;
    const bool one_split = planes_in_one_launch(d);
    if (one_split) {
        const WPlanes wq = wplanes_of(d, w.base, w.off);
        X3SplitJob jobs[16];
        int nj = 0;
        jobs[nj++] = X3SplitJob{P.p(DC_P_UNIT_W), wq.fwd(wq.unit), 6 * EMBW, EMBW, EMBW, 0, 6 * EMBW};
        jobs[nj++] = X3SplitJob{P.p(DC_P_PRE_W), wq.fwd(wq.pre), PREW, XCATW, XCATW, 0, PREW};
        jobs[nj++] = X3SplitJob{P.p(DC_P_HEADS_W), wq.fwd(wq.heads), HO_N, H, H, 0, HO_LD};
        for (int l = 0; l < d->layers; ++l) jobs[nj++] = X3SplitJob{P.p(DC_P_RNN0 + 4 * l), wq.fwd(wq.ih[l]), G * H, l == 0 ? PREW : H, l == 0 ? PREW : H, 0, G * H};
        if (!(d->flags & DC_DIMS_FWD_ONLY)) {       // the transposes policy_backward contracts with
            jobs[nj++] = X3SplitJob{P.p(DC_P_PRE_W), wq.bwd(wq.pre), PREW, XCATW, XCATW, 1, PREW};
            jobs[nj++] = X3SplitJob{P.p(DC_P_HEADS_W), wq.bwd(wq.heads), HO_N, H, H, 1, HO_LD};
            for (int l = 0; l < d->layers; ++l) jobs[nj++] = X3SplitJob{P.p(DC_P_RNN0 + 4 * l), wq.bwd(wq.ih[l]), G * H, l == 0 ? PREW : H, l == 0 ? PREW : H, 1, G * H};
            for (int t = 2; t < 4; ++t)
                jobs[nj++] = X3SplitJob{P.p(DC_P_UNIT_W) + (size_t)t * EMBW * EMBW, wq.bwd(wq.unit) + 3 * (size_t)t * EMBW * EMBW, EMBW, EMBW, EMBW, 1, EMBW};
        }
        DC_TRY(split_weight_planes(jobs, nj, 4, s, F16X2_S_W));
    }
    if (fused) {
        // W2 of the six unit types as bf16 planes (one tiny pre-pass): the fused kernel's weight operand then needs no split
        const WPlanes wpe = wplanes_of(d, w.base, w.off);
        const bool bpl = !(d->flags & DC_DIMS_GEMM_FASTTILE);
        const bool eh = (d->flags & DC_DIMS_F16X2) && bpl;        // (the embedding MLP stays f32-grade in bf16 mode, so it may take the f16 pieces there too)
        if (bpl && !one_split) {
            X3SplitJob job{P.p(DC_P_UNIT_W), wpe.fwd(wpe.unit), 6 * EMBW, EMBW, EMBW, 0, 6 * EMBW};
            DC_TRY(split_weight_planes(&job, 1, eh ? 4 : 6, s, F16X2_S_W));
        }
        F16x2Scales fs;
        fs.on = eh; fs.s_act = F16X2_S_ACT; fs.s_w = F16X2_S_W;
        // (round 6) the f16x2 variant also takes the env embedding and the five-unit pool: no pool_env_fwd launch (DC_DIMS_POOL_ENV_SEPARATE: A/B)
        pools_done = eh && !(d->flags & DC_DIMS_POOL_ENV_SEPARATE);
        DC_TRY(embed_fwd_fused(obs, P.p(DC_P_BASIC_W), P.p(DC_P_BASIC_B), P.p(DC_P_UNIT_W), bpl ? wpe.fwd(wpe.unit) : nullptr, P.p(DC_P_UNIT_B),
                               w.f(DC_WS_EMB), w.f(DC_WS_XCAT), amax, NR, NRp, s, fs, (d->flags & DC_DIMS_LAZY_TU) ? unit_mask : nullptr,
                               pools_done ? P.p(DC_P_ENV_W) : nullptr, pools_done ? P.p(DC_P_ENV_B) : nullptr));
    } else {
        DC_TRY(unit_basic_fwd(obs, P.p(DC_P_BASIC_W), P.p(DC_P_BASIC_B), w.f(DC_WS_BASIC), NR, s));
        for (int t = 0; t < 6; ++t) {
            const size_t ro = (size_t)NR * T_CUM[t] * EMBW;
            DC_TRY(gemm_f32(w.f(DC_WS_BASIC) + ro, P.p(DC_P_UNIT_W) + (size_t)t * EMBW * EMBW, w.f(DC_WS_EMB) + ro,
                            (int)(NR * T_UNITS[t]), EMBW, EMBW, EMBW, EMBW, EMBW, 0, 0, P.p(DC_P_UNIT_B) + t * EMBW, 0, nullptr, 0,
                            0, 1, s));
        }
    }
    // env embedding + max-pools -> xcat (policy.py:97,102-136)
    // (fused: only the env embedding and the 5-unit type are left to do - the GEMM epilogue pooled the rest)
    if (!pools_done) DC_TRY(pool_env_fwd(obs, w.f(DC_WS_EMB), P.p(DC_P_ENV_W), P.p(DC_P_ENV_B), w.f(DC_WS_XCAT), amax, NR, NRp, fused ? 1 : 0, s));
    // the dense products read their weights as bf16 planes (gemm_x3.hip): one pre-pass per forward over the three / four matrices
    // (measured, tools/gemm_bench.py at 256 x 256: x W^T / dy W run at 145-150 TF on either kernel - the round-1 tile kernel with
    // its fragments split after the LDS reads needs no pre-pass and is the default there; the split-on-load kernel is 1.5x
    // faster on the weight-gradient products, 94 -> 146-152 TF, and is the only one with the bf16 mode)
    const bool f16x2 = (d->flags & DC_DIMS_F16X2) && !(d->flags & DC_DIMS_BF16) && !(d->flags & DC_DIMS_GEMM_FASTTILE);
    const bool x3 = ((d->flags & DC_DIMS_BF16) || (d->flags & DC_DIMS_GEMM_X3_ALL) || f16x2) && !(d->flags & DC_DIMS_GEMM_FASTTILE);
    const int prec = (d->flags & DC_DIMS_BF16) ? 1 : (f16x2 ? 4 : 6);
    const WPlanes wp = wplanes_of(d, w.base, w.off);
    if (x3 && !one_split) {
        X3SplitJob jobs[2 + DC_MAX_LAYERS];
        int nj = 0;
        jobs[nj++] = X3SplitJob{P.p(DC_P_PRE_W), wp.fwd(wp.pre), PREW, XCATW, XCATW, 0, PREW};
        jobs[nj++] = X3SplitJob{P.p(DC_P_HEADS_W), wp.fwd(wp.heads), HO_N, H, H, 0, HO_LD};
        for (int l = 0; l < d->layers; ++l) {
            const int in = l == 0 ? PREW : H;
            jobs[nj++] = X3SplitJob{P.p(DC_P_RNN0 + 4 * l), wp.fwd(wp.ih[l]), G * H, in, in, 0, G * H};
        }
        DC_TRY(split_weight_planes(jobs, nj, prec, s, F16X2_S_W));
    }
    // W_hh as bf16 for the recurrence steps; their step-major bf16 state copies (2 x n_seq x 4H) live in the layer's `hn` buffer
    // (rows x H floats, GRU only)
    const bool hh_bf = lstm_step_bf16_supported(d->cell, H, d->flags, wp.base) && 4 * (long long)B <= NR;
    if (hh_bf) {
        X3SplitJob jobs[DC_MAX_LAYERS];
        for (int l = 0; l < d->layers; ++l) jobs[l] = X3SplitJob{P.p(DC_P_RNN0 + 4 * l + 1), wp.fwd(wp.hh[l]), G * H, H, H, 0, G * H};
        DC_TRY(split_weight_planes(jobs, d->layers, 1, s));
    }
    // y[rows][N] = x[rows][K] W[N][K]^T + b, through the split-on-load kernel (or the round-1 kernel)
    const bool bs = bf16_store(d);
    auto linear = [&](const float* x, int K, const float* W, const uint16_t* Wp, int N, int Npad, const float* bias, int relu, float* y,
                      int ldy, int y_bf16 = 0, int x_bf16 = 0) -> int {
        if (!x3) return gemm_f32(x, W, y, (int)NR, N, K, K, K, ldy, 0, 0, bias, relu, nullptr, 0, 0, 1, s);
        X3Gemm g;
        g.A = x; g.a_mode = X3_ROW; g.lda = K;
        g.B = Wp; g.b_mode = X3_PLANES; g.ldb = K; g.b_plane = (long long)Npad * K;
        g.C = y; g.ldc = ldy; g.M = (int)NR; g.N = Npad; g.K = K; g.bias = bias; g.nbias = N; g.relu = relu; g.prec = prec;
        g.sa = F16X2_S_ACT; g.sb = F16X2_S_W;
        g.c_bf16 = y_bf16; g.a_bf16 = x_bf16;
        g.tile128 = (d->flags & DC_DIMS_GEMM_TILE128) ? 1 : 0;
        return gemm_x3(g, s);
    };
    // pre-rnn projection (policy.py:138)
    DC_TRY(linear(w.f(DC_WS_XCAT), XCATW, P.p(DC_P_PRE_W), wp.fwd(wp.pre), PREW, PREW, P.p(DC_P_PRE_B), 1, w.f(DC_WS_PRE), PREW, bs));
    // recurrent core (policy.py:141)
    const float* x = w.f(DC_WS_PRE);
    int in = PREW;
    for (int l = 0; l < d->layers; ++l) {
        const int pb = DC_P_RNN0 + 4 * l;
        DC_TRY(linear(x, in, P.p(pb + 0), wp.fwd(wp.ih[l]), G * H, G * H, P.p(pb + 2), 0, w.fl(l, DC_WSL_GATES), G * H, bs, bs));
        RnnStepArgs a{};
        a.bf16_store = bs;
        a.fwd_only = (d->flags & DC_DIMS_FWD_ONLY) ? 1 : 0;
        a.h0 = h0 ? h0 + (size_t)l * B * H : nullptr;
        a.c0 = (c0 && d->cell == 1) ? c0 + (size_t)l * B * H : nullptr;
        a.seq_off = seq_off; a.seq_len = seq_len; a.n_seq = B; a.H = H;
        a.flags = d->flags; a.xbuf = w.base + w.off[DC_WS_TEAM_XBUF];
        a.fault = reinterpret_cast<int*>(w.base + w.off[DC_WS_FAULT]); a.layer = l;
        a.Whh = P.p(pb + 1); a.bhh = P.p(pb + 3);
        a.Whh_bf = hh_bf ? wp.fwd(wp.hh[l]) : nullptr;
        a.stepbf = reinterpret_cast<uint16_t*>(w.fl(l, DC_WSL_HN));
        a.gates = w.fl(l, DC_WSL_GATES); a.hn = w.fl(l, DC_WSL_HN); a.hseq = w.fl(l, DC_WSL_HSEQ);
        a.hprev = w.fl(l, DC_WSL_HPREV); a.cseq = w.fl(l, DC_WSL_CSEQ); a.cprev = w.fl(l, DC_WSL_CPREV);
        DC_TRY(rnn_forward_layer(d->cell, a, d->max_len, s));
        if (hT) DC_TRY(rnn_final_state(w.fl(l, DC_WSL_HSEQ), hT + (size_t)l * B * H, seq_off, seq_len, B, H, s, bs));
        if (cT && d->cell == 1)
            DC_TRY(rnn_final_state(w.fl(l, DC_WSL_CSEQ), cT + (size_t)l * B * H, seq_off, seq_len, B, H, s));
        x = w.fl(l, DC_WSL_HSEQ);
        in = H;
    }
    // all head projections as one GEMM (policy.py:144-155), then the attention logits (policy.py:152)
    // (x3: the weight planes are zero-padded to 160 rows, so the pad columns 154..159 of headout are written as zeros)
    DC_TRY(linear(x, H, P.p(DC_P_HEADS_W), wp.fwd(wp.heads), HO_N, HO_LD, P.p(DC_P_HEADS_B), 0, w.f(DC_WS_HEADOUT), HO_LD, 0, bs));
    // (DC_DIMS_LAZY_TU: left to dc_select_logp / dc_ppo_loss_fwd_bwd, which know which units are unmasked)
    if (!(d->flags & DC_DIMS_LAZY_TU)) DC_TRY(attn_logits(w.f(DC_WS_HEADOUT), w.f(DC_WS_EMB), w.f(DC_WS_TU), NR, NRp, s));
    return 0;
}

// dst[0..n_total) (float4 units) = src[0..n_src) followed by zeros: the head weights [154][H] padded to [160][H]
__global__ __launch_bounds__(256) void copy_zero_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int n_src, int n_total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    reinterpret_cast<float4*>(dst)[i] = i < n_src ? reinterpret_cast<const float4*>(src)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// Consumes DHEADOUT[:,128:154] and DTU (written by ppo_loss_fwd_bwd) and everything policy_forward
// saved; overwrites the flat gradient buffer (total_floats long).
int policy_backward(const dc_dims* d, const float* params, const int64_t* poff, float* grads, int64_t total_floats,
                    const float* obs, const int64_t* seq_off, const int32_t* seq_len, void* ws_base, hipStream_t s) {
    DC_TRY(check_dims(d));
    // DC_DIMS_BWD_UPPER / DC_DIMS_BWD_EMBED: the two halves as separate calls (header); neither bit: both
    const bool do_upper = !(d->flags & DC_DIMS_BWD_EMBED) || (d->flags & DC_DIMS_BWD_UPPER);
    const bool do_embed = !(d->flags & DC_DIMS_BWD_UPPER) || (d->flags & DC_DIMS_BWD_EMBED);
    if (do_upper) {
        DC_TRY(zero_async(grads, (size_t)total_floats * sizeof(float), s));
    }
    if (d->rows <= 0 || d->n_seq <= 0) return 0;
    Ws w; w.base = (char*)ws_base; workspace_layout(d, w.off);
    Params P{params, poff};
    Grads Gd{grads, poff};
    const GemmScratch sc{w.f(DC_WS_SCRATCH), DC_SCRATCH_FLOATS};   // split-K slabs of this call's weight-gradient products
    const long long NR = d->rows;
    const int H = d->hidden, G = d->cell == 0 ? 3 : 4, B = d->n_seq, L = d->layers;
    const int TOP = L - 1;

    // dense products: split-on-load bf16-plane kernel (gemm_x3.hip) or the round-1 kernel
    const bool x3_tn = !(d->flags & DC_DIMS_GEMM_FASTTILE);                                              // weight gradients
    const bool f16x2 = (d->flags & DC_DIMS_F16X2) && !(d->flags & DC_DIMS_BF16) && x3_tn;
    const bool x3 = ((d->flags & DC_DIMS_BF16) || (d->flags & DC_DIMS_GEMM_X3_ALL) || f16x2) && x3_tn;    // input gradients (see policy_forward)
    const int prec = (d->flags & DC_DIMS_BF16) ? 1 : (f16x2 ? 4 : 6);
    const float s_grad = f16x2_grad_scale(NR);
    const WPlanes wp = wplanes_of(d, w.base, w.off);
    // dx[rows][N] = dy[rows][K] W[K][N] (optionally masked by aux > 0): reads W^T as bf16 planes [N][K]
    const bool bs = bf16_store(d);
    auto dgrad = [&](const float* dy, int K, const float* W, const uint16_t* WTp, int N, const float* aux, float* dx, int dy_bf16 = 0,
                     int aux_bf16 = 0) -> int {
        if (!x3) return gemm_f32(dy, W, dx, (int)NR, N, K, K, N, N, 0, 1, nullptr, 0, aux, N, 0, 1, s);
        X3Gemm g;
        g.A = dy; g.a_mode = X3_ROW; g.lda = K;
        g.B = WTp; g.b_mode = X3_PLANES; g.ldb = K; g.b_plane = (long long)N * K;
        g.C = dx; g.ldc = N; g.M = (int)NR; g.N = N; g.K = K; g.aux = aux; g.ldaux = N; g.prec = prec; g.transposed_w = 1;
        g.sa = s_grad; g.sb = F16X2_S_W;
        g.a_bf16 = dy_bf16; g.aux_bf16 = aux_bf16;
        g.tile128 = (d->flags & DC_DIMS_GEMM_TILE128) ? 1 : 0;
        return gemm_x3(g, s);
    };
    // dW[M][N] += dy[rows][lda: M]^T x[rows][ldb: N] (contraction over the env-steps, split-K); optional second x behind N
    // db (optional): the bias gradient that goes with dW, db[m] += sum over the env-steps of dy[.][m] - summed by the same kernel as dy
    // passes through its loader (a separate column-sum pass re-reads dy: 12 launches, 0.44 ms per configs[2] step)
    auto wgrad = [&](const float* dy, int lda, int M, const float* x1, int N1, float* dW1, const float* x2, int N2, float* dW2,
                     float* db = nullptr, int dy_bf16 = 0, int x_bf16 = 0) -> int {
        const bool pair = x2 != nullptr;
        if (x3_tn && gemm_x3_shape_ok(M, N1 + N2, (int)NR, lda, N1, X3_KMAJ, X3_KMAJ) && (!pair || (N1 % 128 == 0 && !(N2 & 3)))) {
            X3Gemm g;
            g.a_colsum = db;
            g.A = dy; g.a_mode = X3_KMAJ; g.lda = lda;
            g.B = x1; g.b_mode = X3_KMAJ; g.ldb = N1; g.B2 = x2; g.ldb2 = N2; g.n_split = pair ? N1 : 0;
            g.C = dW1; g.ldc = N1; g.C2 = dW2; g.ldc2 = N2; g.M = M; g.N = N1 + N2; g.K = (int)NR; g.accumulate = 1; g.prec = prec;
            g.sa = s_grad; g.sb = F16X2_S_ACT;
            g.scratch = sc;
            g.a_bf16 = dy_bf16; g.b_bf16 = x_bf16;
            return gemm_x3(g, s);
        }
        if (dy_bf16 || x_bf16) { set_error("policy_backward: bf16 storage needs the split-on-load products", 1005); return 1005; }
        if (db != nullptr) DC_TRY(colsum(dy, lda, NR, M, db, s));
        if (pair) return gemm_f32_tn_pair(dy, lda, x1, N1, N1, x2, N2, N2, dW1, N1, dW2, N2, M, (int)NR, s, sc);
        return gemm_f32(dy, x1, dW1, M, N1, (int)NR, lda, N1, N1, 1, 1, nullptr, 0, nullptr, 0, 1, 0, s, sc);
    };
    if (do_upper) {
    if (x3 && !planes_in_one_launch(d)) {   // W^T of the matrices the input gradients contract with, as bf16 planes: one pre-pass per backward
        X3SplitJob jobs[4 + DC_MAX_LAYERS];
        int nj = 0;
        jobs[nj++] = X3SplitJob{P.p(DC_P_PRE_W), wp.bwd(wp.pre), PREW, XCATW, XCATW, 1, PREW};          // -> [896][256]
        jobs[nj++] = X3SplitJob{P.p(DC_P_HEADS_W), wp.bwd(wp.heads), HO_N, H, H, 1, HO_LD};             // -> [H][160], zero k-padding
        if (f16x2)      // W2^T of the two 16-unit types: R = q W2_t (the attention term of their d(basic), embed_pool16m.hip) as an f16x2 product
            for (int t = 2; t < 4; ++t)
                jobs[nj++] = X3SplitJob{P.p(DC_P_UNIT_W) + (size_t)t * EMBW * EMBW, wp.bwd(wp.unit) + 3 * (size_t)t * EMBW * EMBW, EMBW, EMBW, EMBW, 1, EMBW};
        for (int l = 0; l < L; ++l) {
            const int in = l == 0 ? PREW : H;
            jobs[nj++] = X3SplitJob{P.p(DC_P_RNN0 + 4 * l), wp.bwd(wp.ih[l]), G * H, in, in, 1, G * H};  // -> [in][G*H]
        }
        DC_TRY(split_weight_planes(jobs, nj, prec, s, F16X2_S_W));
    }
    const bool hh_bf = lstm_step_bf16_supported(d->cell, H, d->flags, wp.base) && 4 * (long long)B <= NR;     // W_hh^T as bf16 for the recurrence steps
    if (hh_bf) {
        X3SplitJob jobs[DC_MAX_LAYERS];
        for (int l = 0; l < L; ++l) jobs[l] = X3SplitJob{P.p(DC_P_RNN0 + 4 * l + 1), wp.bwd(wp.hh[l]), G * H, H, H, 1, G * H};   // -> [H][G*H]
        DC_TRY(split_weight_planes(jobs, L, 1, s));
    }
    // heads (policy.py:144-155)
    DC_TRY(attn_bwd_q(w.f(DC_WS_DTU), w.f(DC_WS_EMB), w.f(DC_WS_DHEADOUT), NR, emb_rows(d), s));
    // dH = dheadout[:, 0:160] * [W_heads; 0]: K padded to 160 (dheadout's pad columns are zeroed by the loss kernel)
    if (!x3) {
        hipLaunchKernelGGL(copy_zero_pad_kernel, dim3((HO_LD * H / 4 + 255) / 256), dim3(256), 0, s, P.p(DC_P_HEADS_W), w.f(DC_WS_HEADW_PAD),
                           HO_N * H / 4, HO_LD * H / 4);   // one launch instead of a copy and a memset
        DC_TRY(launch_check("policy_backward: head weight pad"));
    }
    DC_TRY(dgrad(w.f(DC_WS_DHEADOUT), HO_LD, w.f(DC_WS_HEADW_PAD), wp.bwd(wp.heads), H, nullptr, w.fl(TOP, DC_WSL_DH)));
    DC_TRY(wgrad(w.f(DC_WS_DHEADOUT), HO_LD, HO_N, w.fl(TOP, DC_WSL_HSEQ), H, Gd.p(DC_P_HEADS_W), nullptr, 0, nullptr, Gd.p(DC_P_HEADS_B), 0, bs));

    // recurrent core, top layer first
    for (int l = TOP; l >= 0; --l) {
        const int pb = DC_P_RNN0 + 4 * l;
        if (!rnn_uses_persistent(d->cell, H, d->flags) && !hh_bf) DC_TRY(transpose(P.p(pb + 1), w.f(DC_WS_WHHT), G * H, H, s));
        RnnStepArgs a{};
        a.seq_off = seq_off; a.seq_len = seq_len; a.n_seq = B; a.H = H;
        a.flags = d->flags; a.xbuf = w.base + w.off[DC_WS_TEAM_XBUF];
        a.fault = reinterpret_cast<int*>(w.base + w.off[DC_WS_FAULT]); a.layer = l;
        a.gates = w.fl(l, DC_WSL_GATES); a.hn = w.fl(l, DC_WSL_HN); a.hseq = w.fl(l, DC_WSL_HSEQ);
        a.hprev = w.fl(l, DC_WSL_HPREV); a.cseq = w.fl(l, DC_WSL_CSEQ); a.cprev = w.fl(l, DC_WSL_CPREV);
        a.WhhT_bf = hh_bf ? wp.bwd(wp.hh[l]) : nullptr;
        a.stepbf = reinterpret_cast<uint16_t*>(w.fl(l, DC_WSL_HN));
        a.Whh = P.p(pb + 1); a.WhhT = w.f(DC_WS_WHHT); a.dh = w.fl(l, DC_WSL_DH); a.dc = w.fl(l, DC_WSL_DC);
        a.dgx = w.fl(l, DC_WSL_DGX);
        a.dgh = d->cell == 0 ? w.fl(l, DC_WSL_DGH) : w.fl(l, DC_WSL_DGX);
        a.s_grad = f16x2 ? s_grad : 0.f;
        a.bf16_store = bs;
        DC_TRY(rnn_backward_layer(d->cell, a, d->max_len, s));
        const float* xin = l == 0 ? w.f(DC_WS_PRE) : w.fl(l - 1, DC_WSL_HSEQ);
        const int in = l == 0 ? PREW : H;
        // dW_ih = dgx^T x ; dW_hh = dgh^T h_prev ; biases = column sums
        if (a.dgh == a.dgx) {   // LSTM: both products contract the same gate gradients - one launch
            DC_TRY(wgrad(a.dgx, G * H, G * H, xin, in, Gd.p(pb + 0), a.hprev, H, Gd.p(pb + 1), Gd.p(pb + 2), bs, bs));
            // dgh is dgx, so d(b_hh) = d(b_ih): copy 2 KB
            DC_TRY(copy_f32_async(Gd.p(pb + 3), Gd.p(pb + 2), (long long)G * H, s));
        } else {
            DC_TRY(wgrad(a.dgx, G * H, G * H, xin, in, Gd.p(pb + 0), nullptr, 0, nullptr, Gd.p(pb + 2)));
            DC_TRY(wgrad(a.dgh, G * H, G * H, a.hprev, H, Gd.p(pb + 1), nullptr, 0, nullptr, Gd.p(pb + 3)));
        }
        if (l > 0) {
            DC_TRY(dgrad(a.dgx, G * H, P.p(pb + 0), wp.bwd(wp.ih[l]), H, nullptr, w.fl(l - 1, DC_WSL_DH), bs));
        } else {
            // through relu(affine_pre_rnn) (policy.py:138): mask with the stored activation
            DC_TRY(dgrad(a.dgx, G * H, P.p(pb + 0), wp.bwd(wp.ih[0]), PREW, w.f(DC_WS_PRE), w.f(DC_WS_DPRE), bs, bs));
        }
    }
    DC_TRY(wgrad(w.f(DC_WS_DPRE), PREW, PREW, w.f(DC_WS_XCAT), XCATW, Gd.p(DC_P_PRE_W), nullptr, 0, nullptr, Gd.p(DC_P_PRE_B)));
    DC_TRY(dgrad(w.f(DC_WS_DPRE), PREW, P.p(DC_P_PRE_W), wp.bwd(wp.pre), XCATW, nullptr, w.f(DC_WS_DXCAT)));
    }   // do_upper
    if (!do_embed) return 0;

    // max-pool routing + attention keys -> per-unit embedding gradients; env embedding weights
    // Fused path: the two 16-unit types (32 of the 40 units) take the sparse max-pool backward (embed_sparse.hip);
    // DC_DIMS_DENSE_POOL_BWD: the dense kernels for all types.
    const bool fusedb = embed_fused_on(d);
    const long long NRp = emb_rows(d);
    const bool sparse16 = fusedb && embed_sparse_enabled(d);
    const uint8_t* amaxp = reinterpret_cast<const uint8_t*>(w.base + w.off[DC_WS_AMAX]);
    // the inputs of the on-chip backward kernels (embed_pool16m.hip, embed_small.hip), and which of them run
    const EmbSparseIn sp{w.f(DC_WS_DXCAT), amaxp, w.f(DC_WS_DTU), w.f(DC_WS_HEADOUT), HO_LD, Gd.p(DC_P_UNIT_B),
                         w.f(DC_WS_DEMB) + (size_t)NRp * T_CUM[2] * EMBW, (d->flags & DC_DIMS_POOL16_8W) ? 1 : 0,
                         (d->flags & DC_DIMS_POOL16_VALU) ? 1 : 0, (d->flags & DC_DIMS_SMALL_DENSE) ? 1 : 0,
                         (d->flags & DC_DIMS_DB2_SCATTER) ? 0 : 1, f16x2 ? wp.bwd(wp.unit) : nullptr};
    F16x2Scales fs;
    fs.on = (d->flags & DC_DIMS_F16X2) && !(d->flags & DC_DIMS_GEMM_FASTTILE);
    fs.s_act = F16X2_S_ACT; fs.s_w = F16X2_S_W; fs.s_grad = s_grad;
    const bool small_fused = embed_small_fused(sparse16, fs, &sp);      // then d(emb) is not written for ANY type
    DC_TRY(embed_scatter_bwd(obs, w.f(DC_WS_XCAT), w.f(DC_WS_DXCAT), w.f(DC_WS_DTU), w.f(DC_WS_HEADOUT), HO_LD, amaxp,
                             w.f(DC_WS_DEMB), Gd.p(DC_P_ENV_W), Gd.p(DC_P_ENV_B), Gd.p(DC_P_UNIT_B), w.f(DC_WS_SCRATCH), NR,
                             NRp, small_fused ? (sp.small_db2 ? 3 : 2) : (sparse16 ? 1 : 0), s));
    if (fusedb && NRp > NR && !small_fused) {
        // padding steps of the type-major d(emb) blocks the dense kernels read: zero gradients
        for (int t = 0; t < 6; ++t) {
            if (sparse16 && (t == 2 || t == 3)) continue;
            DC_TRY(zero_async(w.f(DC_WS_DEMB) + ((size_t)NRp * T_CUM[t] + (size_t)NR * T_UNITS[t]) * EMBW,
                              (size_t)(NRp - NR) * T_UNITS[t] * EMBW * sizeof(float), s));
        }
    }
    if (fusedb) {
        // dW2 (split-K with the first layer regenerated as B operand) and dW1/db1 (d(basic) kept in the
        // accumulators) - neither `basic` nor d(basic) exists in HBM on this path
        // the prepared staging blocks of the sparse path live in the d(emb) rows of the two 16-unit types (2 * 16 * 128 floats per
        // step, never written on that path; the prepared blocks take 2 * 736)
        DC_TRY(embed_bwd_fused(obs, w.f(DC_WS_DEMB), P.p(DC_P_BASIC_W), P.p(DC_P_BASIC_B), P.p(DC_P_UNIT_W), Gd.p(DC_P_UNIT_W),
                               Gd.p(DC_P_BASIC_W), Gd.p(DC_P_BASIC_B), w.f(DC_WS_SCRATCH), DC_SCRATCH_FLOATS, NR, NRp,
                               sparse16 ? &sp : nullptr, s, fs));
    } else {
        for (int t = 0; t < 6; ++t) {
            const size_t ro = (size_t)NR * T_CUM[t] * EMBW;
            const int rows_t = (int)(NR * T_UNITS[t]);
            DC_TRY(gemm_f32(w.f(DC_WS_DEMB) + ro, w.f(DC_WS_BASIC) + ro, Gd.p(DC_P_UNIT_W) + (size_t)t * EMBW * EMBW, EMBW, EMBW,
                            rows_t, EMBW, EMBW, EMBW, 1, 1, nullptr, 0, nullptr, 0, 1, 0, s, sc));
            // dbasic (stored over the no-longer-needed emb buffer), relu-masked by basic
            DC_TRY(gemm_f32(w.f(DC_WS_DEMB) + ro, P.p(DC_P_UNIT_W) + (size_t)t * EMBW * EMBW, w.f(DC_WS_EMB) + ro, rows_t, EMBW,
                            EMBW, EMBW, EMBW, EMBW, 0, 1, nullptr, 0, w.f(DC_WS_BASIC) + ro, EMBW, 0, 1, s));
        }
        DC_TRY(unit_basic_bwd(obs, w.f(DC_WS_EMB), Gd.p(DC_P_BASIC_W), Gd.p(DC_P_BASIC_B), w.f(DC_WS_SCRATCH), NR, s));
    }
    return 0;
}

}  // namespace dc
