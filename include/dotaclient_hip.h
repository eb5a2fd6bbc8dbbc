/* C ABI of libdotaclient_hip.so - the MI355X (gfx950) replacement for the arithmetic of the
 * reference PPO optimizer hot path (TimZaman/dotaclient, optimizer.py + policy.py).
 *
 * The reference has no FFI/plugin interface (it is pure Python on torch/scipy); the boundary that a
 * maintainer binds is therefore the set of Python functions named below, each replaced by one entry
 * point here.  INTEGRATION.md shows the ctypes stub that goes into the reference's optimizer.py.
 *
 * Conventions
 *   - plain C: device pointers + sizes, no torch types; all pointers are DEVICE pointers unless
 *     the parameter name ends in _host;
 *   - every function enqueues work on the caller's HIP stream (dc_stream_t = hipStream_t) and returns
 *     immediately; nothing is allocated, freed or synchronised inside, no state is kept between calls (all scratch comes
 *     from the caller: the workspace of dc_workspace_layout, the scratch argument of dc_gemm_f32) and no environment
 *     variable is read: two callers with their own buffers may drive two streams from two threads concurrently;
 *   - return value 0 = ok, anything else = HIP error code or a dc_* code >= 1000;
 *     dc_last_error() gives the message (the Python shim raises RuntimeError / ValueError from it);
 *   - inputs are never modified; outputs are fully overwritten unless documented as accumulating;
 *   - "rows" are env-steps in the packed trajectory layout: sequence b occupies rows
 *     [seq_off[b], seq_off[b] + seq_len[b]).
 */
#ifndef DOTACLIENT_HIP_H
#define DOTACLIENT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dc_stream_t; /* hipStream_t */

/* 3 (round 3): DC_WS_FAULT inserted at workspace index 0; round 2's unannounced changes (dc_gemm_f32's scratch arguments,
 * DC_WS_TEAM_XBUF / DC_WS_WPLANES) were version 2 in effect.
 * 4 (round 4): dc_policy_forward takes the action masks (unit_mask); otherwise same signatures, changed contracts - dc_gradnorm_clip_adam's status word is sticky (a non-zero word makes later calls
 * skip their update until the caller clears it); dc_gae_scan / dc_discount / dc_advantage_returns accept any length (error 1001 is gone).
 * (round 6 added dc_dims.flags / prec BITS only - DC_DIMS_GEMM_TILE128, DC_DIMS_FWD_ONLY, DC_DIMS_SMALL_DENSE, DC_GEMM_PREC_TILE128: a caller that does not set them gets
 * the same results as before from the same signatures, so the number stays.)
 * The Python binding refuses any other value. */
#define DC_ABI_VERSION 4
int dc_abi_version(void);
const char* dc_last_error(void);

/* Replaces optimizer.py:53-64 `discount` + `advantage_returns`, fused with the sub-reward sum of
 * optimizer.py:397 and the terminal-zero append of optimizer.py:417-420.
 *   rewards [rows,10] f32, values [rows] f32, seq_off [n_seq] i64, seq_len [n_seq] i32 (device),
 *   max_len = max(seq_len) (host), gamma/lam as the reference's python doubles (0.98, 0.97),
 *   adv/ret [rows] f32 out. */
int dc_gae_scan(const float* rewards, const float* values, const int64_t* seq_off, const int32_t* seq_len,
                int n_seq, int max_len, double gamma, double lam, float* adv, float* ret, dc_stream_t stream);

/* Replaces optimizer.py:53-54 `discount(x, gamma)` for one vector: y[t] = x[t] + gamma * y[t+1] (the reversed
 * scipy lfilter([1],[1,-gamma]) of the reference: float32 in, float64 accumulate, float32 out).
 *   x, y [n] f32 (device; y may alias x); any n (vectors longer than one 40 960-entry LDS block are scanned block by block
 *   from the end, carrying the float64 state - the reference's lfilter has no length limit). */
int dc_discount(const float* x, int n, double gamma, float* y, dc_stream_t stream);

/* Replaces optimizer.py:57-64 `advantage_returns(rewards, values, gamma, lam)` for ONE rollout, with the reference's
 * own argument shapes: rewards, values [L+1] f32 (entry L is whatever the caller appended - the reference's run()
 * appends 0, optimizer.py:417-420, but the function itself accepts any terminal reward / bootstrap value);
 * adv, ret [L] f32 out.  Any L (blocks of 20 480 steps from the end, float64 carry). */
int dc_advantage_returns(const float* rewards, const float* values, int L, double gamma, double lam, float* adv,
                         float* ret, dc_stream_t stream);

/* fp32 MFMA GEMM building block (every nn.Linear of policy.py:54-75 and its autograd products).
 *   C[M,N] (op)= A[M,K] * B[K,N] (+ bias[N]) ; a_kmajor: A stored [K][lda] else [M][lda];
 *   b_kmajor: B stored [K][ldb] else [N][ldb]; relu: max(0,.); aux/ldaux: zero where aux<=0;
 *   accumulate: C += ; splits: 0 = auto split-K.
 *   scratch / scratch_floats: device scratch of THIS call for the split-K partial slabs (reduced by a second kernel);
 *   NULL / 0 = fp32 atomics instead.  The library keeps no scratch pointer between calls. */
int dc_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                int a_kmajor, int b_kmajor, const float* bias, int relu, const float* aux, int ldaux,
                int accumulate, int splits, float* scratch, int64_t scratch_floats, dc_stream_t stream);

/* The same product through the split-on-load kernel the network's dense layers use (gemm_x3.hip): operands as exact 3-way bf16
 * splits on the bf16 matrix cores (prec 6: f32-grade results), as two f16 pieces with three MFMAs (prec 4, the network's default
 * arithmetic: f32-grade while sa * |a| and sb * |b| stay inside f16's exponent range; the power-of-two pre-scales sa = 2^la, sb = 2^lb
 * travel in the upper bits of `prec`, DC_GEMM_PREC_F16X2(la, lb) - plain 4 = unit scales; an entry beyond 65504 / scale becomes inf,
 * its output row / column non-finite: never a silently wrong finite number) or rounded to bf16 (prec 1), f32 accumulate.  Layouts:
 * a_kmajor = b_kmajor = 0 (x W^T), a_kmajor = 0 / b_kmajor = 1 (dy W), a_kmajor = b_kmajor = 1 (dy^T x, split-K).  B is a weight
 * matrix in the first two forms and is split into bf16 planes inside the call (into `scratch`, which must hold
 * 3 * N * K / 2 floats for it, plus splits * M * N for split-K).  K % 16 == 0, N % 4 == 0.  No relu/aux with split-K. */
#define DC_GEMM_PREC_F16X2(la, lb) (4 | (((la) & 0xff) << 8) | (((lb) & 0xff) << 16))
/* prec 4, x W^T / dy W: OR this in to keep the product on the 128 x 128 split-on-load kernel (default: the row-streaming kernel of
 * csrc/gemm_x3s.hip whenever K % 32 == 0 and the shape has at least 192 tiles of 256 x 128). */
#define DC_GEMM_PREC_TILE128 (1 << 24)
/* prec 1 with operands / results STORED as bf16 (BASELINE.json configs[4]'s path keeps its gate buffers that way): a, b, c, aux = 1 when
 * A, B, C, aux point at bf16 elements of the same shape and ld (in elements).  b: the k-major form only; c: no accumulate, no split-K. */
#define DC_GEMM_PREC_BF16_STORE(a, b, c, aux) (1 | ((a) << 8) | ((b) << 9) | ((c) << 10) | ((aux) << 11))
int dc_gemm_x3(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
               int a_kmajor, int b_kmajor, const float* bias, int relu, const float* aux, int ldaux,
               int accumulate, int prec, float* scratch, int64_t scratch_floats, dc_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Network + optimizer step.  Shapes: dc_dims; parameters: one flat fp32 buffer whose tensors are
 * addressed by an offset table `poff` (floats, host array, index = dc_param_index); scratch: one
 * device workspace laid out by dc_workspace_layout (buffers saved by the forward are consumed by the
 * loss / backward of the same batch).
 * ------------------------------------------------------------------------------------------------ */
#define DC_MAX_LAYERS 4

typedef struct dc_dims {
    int32_t cell;     /* 0 = GRU (the reference, policy.py:66), 1 = LSTM (BASELINE.json extension) */
    int32_t hidden;   /* H, multiple of 64 */
    int32_t layers;   /* 1..DC_MAX_LAYERS */
    int32_t n_seq;    /* B: sequences (trajectory chunks) in the batch */
    int32_t max_len;  /* max(seq_len) */
    int32_t flags;    /* DC_DIMS_* bits; 0 = plain */
    int64_t rows;     /* total env-steps = sum(seq_len) */
} dc_dims;

/* dc_dims.flags.  DC_DIMS_LAZY_TU: dc_policy_forward does NOT run the dense target-unit attention
 * (policy.py:152, 20 KB of embeddings per env-step); dc_select_logp / dc_ppo_loss_fwd_bwd - the consumers,
 * which hold the action masks - compute the logits of the UNMASKED units only (all others are written as 0;
 * masked_softmax, policy.py:169-178, never lets them reach a result).  Set by the optimizer's two passes;
 * leave clear when the caller wants DC_WS_TU for every unit (Policy.forward). */
#define DC_DIMS_LAZY_TU 1
/* dc_policy_backward in two calls, so that a data-parallel caller can start the all-reduce of the first part's
 * gradients while the second part runs (distributed.py:29-57 reduces after the whole backward):
 *   DC_DIMS_BWD_UPPER : zero the gradient buffer, then heads, recurrent core, pre-rnn projection - every parameter from
 *                       affine_pre_rnn on; leaves d(xcat) and the attention gradients in the workspace;
 *   DC_DIMS_BWD_EMBED : the unit / env embedding parameters from that workspace state (call after UPPER, same arguments).
 * Neither bit set: both parts, one call. */
#define DC_DIMS_BWD_UPPER 2
#define DC_DIMS_BWD_EMBED 4
/* Kernel-selection overrides (A/B measurements and the parity tests that pit one kernel family against another; the
 * default - none set - picks by shape).  They are part of the call, not of the process: nothing here reads the environment.
 *   DC_DIMS_DENSE_POOL_BWD : dense MFMA kernels for the max-pool backward of every unit type (default: the sparse
 *                            one-non-zero-per-channel form for the two 16-unit types when rows % 128 == 0);
 *   DC_DIMS_RNN_PER_STEP   : one launch per time step instead of the persistent (register-resident / team) recurrent kernels;
 *   DC_DIMS_LSTM_MFMA/_VALU: H <= 128 LSTM: force the 4-sequences-per-workgroup MFMA / 1-sequence-per-workgroup VALU variant;
 *   DC_DIMS_TEAM_DEVICE_SCOPE : H = 256 team kernels: write-through granule stores even when a team shares an XCD;
 *   DC_DIMS_TEAM_NS(n)     : H = 256 VALU team kernels: n = 1, 2 or 4 sequences in flight per team (0 = by batch size); where the MFMA team
 *                            kernels run, n = 2 selects their earlier forward forms (k split across the waves, meeting in LDS) for A/B;
 *                            H = 512 persistent bf16 kernels: n = 2 forces 32-sequence tiles (default: 16 while the batch fits one round). */
#define DC_DIMS_DENSE_POOL_BWD 8
#define DC_DIMS_RNN_PER_STEP 16
#define DC_DIMS_LSTM_MFMA 32
#define DC_DIMS_LSTM_VALU 64
#define DC_DIMS_TEAM_DEVICE_SCOPE 128
#define DC_DIMS_TEAM_NS_SHIFT 8
#define DC_DIMS_TEAM_NS(n) ((n) << DC_DIMS_TEAM_NS_SHIFT)   /* bits 8..10 */
/*   DC_DIMS_GEMM_FASTTILE  : every dense product through the round-1 tile kernel (f32 tiles by LDS-DMA, fragments split after
 *                            their LDS reads); DC_DIMS_GEMM_X3_ALL: every one through the split-on-load kernel of gemm_x3.hip
 *                            (default: that kernel for the weight gradients, the tile kernel for x W^T and dy W);
 * Precision.  Default: every product is f32-grade (f32 operands split exactly into three bf16 pieces, six bf16 MFMAs,
 * f32 accumulate - results within f32 round-off of an f32 fma chain).
 *   DC_DIMS_BF16           : BASELINE.json configs[4] "bf16 MFMA path": the operands of the dense products (pre-rnn, recurrent
 *                            input projection, heads and their gradients) are rounded to bf16 (one MFMA instead of six),
 *                            f32 accumulate; everything else stays f32.  Tolerance vs the f32 oracle: tests/test_gpu_bf16.py. */
#define DC_DIMS_GEMM_FASTTILE 2048
#define DC_DIMS_BF16 4096
#define DC_DIMS_GEMM_X3_ALL 8192
/*   DC_DIMS_TEAM_VALU      : H = 256 LSTM: the packed-f32 VALU team kernels (one sequence per turn) also for batches of more than
 *                            128 sequences, where the default is the MFMA team kernel that advances four sequences together. */
#define DC_DIMS_TEAM_VALU 16384
/*   DC_DIMS_EMBED_UNFUSED  : the unit-embedding MLP layer by layer (first layer materialised, six dense products) instead of the
 *                            fused kernels, which handle any row count by padding the type-major blocks to a multiple of 128 rows. */
#define DC_DIMS_EMBED_UNFUSED 32768
/*   DC_DIMS_RNN_STEP_BF16  : DC_DIMS_BF16 with H = 512: the launch-per-step bf16 recurrent kernels (rnn_step_bf16.hip) instead of the
 *                            persistent team kernel (rnn_team512.hip: sixteen workgroups hold W_hh as bf16 in their registers). */
#define DC_DIMS_RNN_STEP_BF16 65536
/*   DC_DIMS_F16X2          : f32-grade products from TWO f16 pieces per operand and three MFMAs (instead of three bf16 pieces and six):
 *                            x 2^s = h + m with h = f16(x 2^s), m = f16(x 2^s - h) carries 23 of the 24 significand bits; operands are
 *                            pre-scaled by fixed powers of two (activations 2^4, weights 2^8, gradients 2^(ceil(log2 rows) + 2)) and the
 *                            products scaled back - exact.  Same accuracy against f64 as the six-MFMA form (tools/ubench/gemm_x3.hip),
 *                            fewer MFMAs and LDS bytes per flop on products that run at the chip's power limit.  f16's exponent range
 *                            is the price: an operand entry beyond 65504 / scale (activation > 4094, weight > 255, gradient >
 *                            0.25 * 65536 / rows) turns into inf -> NaN -> the status word of dc_gradnorm_clip_adam (nothing
 *                            updated); the caller then repeats the iteration without this flag.  Ignored with DC_DIMS_BF16. */
#define DC_DIMS_F16X2 131072
/*   DC_DIMS_POOL16_8W      : sparse max-pool backward of the 16-unit types with round 2's eight-wave kernel (both step streams in every
 *                            wave, 256 registers, two waves per SIMD) instead of the sixteen-wave one (a step stream per group of eight
 *                            waves, dW2 update split over k, pair-wise dW1 fold: 128 registers, four waves per SIMD; ~5 % faster). */
#define DC_DIMS_POOL16_8W 262144
/*   DC_DIMS_POOL16_VALU    : with DC_DIMS_F16X2 the max-pool backward of the 16-unit types runs as DENSE products on the f16 matrix cores with
 *                            every operand generated on chip (csrc/embed_pool16m.hip, round 5: no prepare pass, no gathers, no barrier per
 *                            step); this flag keeps the sparse VALU kernels of csrc/embed_sparse.hip (what the bf16x3 products always use:
 *                            f32's exponent range). */
#define DC_DIMS_POOL16_VALU 2097152
/*   DC_DIMS_TEAM8          : H = 256 recurrent core in teams of EIGHT workgroups, two workgroups per CU (csrc/rnn_team8.hip): a member
 *                            holds the gate columns of 32 hidden units (128 AGPRs), the second workgroup on the CU works while the first
 *                            waits for its peers.  The library takes these kernels by itself for 65 .. 128 sequences (one workgroup per
 *                            CU then: BASELINE.json configs[3]'s per-GPU shard); the flag forces them for any number of sequences,
 *                            DC_DIMS_TEAM4 keeps them off (A/B). */
#define DC_DIMS_TEAM8 524288
#define DC_DIMS_TEAM4 1048576
/*   DC_DIMS_BF16_F32_STORE : with DC_DIMS_BF16 on the persistent LSTM-512 kernels (BASELINE.json configs[4]) the gate pre-activations /
 *                            activated gates and the gate gradients are STORED as bf16 (round 5: csrc/policy.hip bf16_store()); this flag
 *                            keeps them f32 (A/B; the comparison with the launch-per-step kernels, whose buffers are f32). */
#define DC_DIMS_BF16_F32_STORE 4194304
/*   DC_DIMS_GEMM_TILE128   : with DC_DIMS_F16X2 the x W^T / dy W products run on the row-streaming kernel (csrc/gemm_x3s.hip, round 6: both
 *                            operands by LDS-DMA, 256 x 128 tiles, eight waves that split the rows, epilogue from the registers); this flag
 *                            keeps them on the 128 x 128 split-on-load kernel of csrc/gemm_x3.hip (A/B, and the parity test between the two). */
#define DC_DIMS_GEMM_TILE128 8388608
/*   DC_DIMS_FWD_ONLY       : dc_policy_forward for a pass that NO dc_policy_backward will follow (the optimizer's no-grad rollout pass,
 *                            optimizer.py:344-385): kernels may skip what only a backward reads.  Used by the H = 256 MFMA team forward (the
 *                            activated gates and the previous step's h / c are not written: 562 -> ~135 MB per launch); DC_WS_HEADOUT, DC_WS_TU, hT / cT
 *                            and what dc_chunk_initial_state / dc_select_logp read are complete.  Calling dc_policy_backward after such a
 *                            forward is a caller error (the gradients would be garbage, not an error code). */
#define DC_DIMS_FWD_ONLY 16777216
/*   DC_DIMS_SMALL_DENSE    : with DC_DIMS_F16X2 the backward of the four small unit types (8 of the 40 units) runs as ONE kernel with every
 *                            operand formed on chip (csrc/embed_small.hip, round 6) and d(emb) is written for no type; this flag keeps rounds
 *                            1-5's path for them - d(emb) rows in HBM, embed_bwd_dw2 + embed_bwd_dw1 - for A/B. */
#define DC_DIMS_SMALL_DENSE 33554432
/* A/B: keep the env embedding and the five-unit max-pool as their own launch (pool_env_fwd) behind the fused embedding forward, as up to
 * round 5; default since round 6: the fused forward's epilogue takes them (24-step tiles for the five-unit type), no such launch */
#define DC_DIMS_POOL_ENV_SEPARATE 67108864
/* A/B: the small unit types' second-layer bias gradients from embed_scatter_bwd's own pass over d(xcat), q and dtu (the first half of round 6);
 * default: csrc/embed_small.hip sums them from the d(emb) patches it forms anyway, and embed_scatter_bwd reads the env slot only */
#define DC_DIMS_DB2_SCATTER 134217728

/* index into poff[]; policy.py:54-75 names in comments */
enum dc_param_index {
    DC_P_ENV_W = 0,   /* affine_env.weight [128,3] */
    DC_P_ENV_B,       /* affine_env.bias */
    DC_P_BASIC_W,     /* affine_unit_basic_stats.weight [128,12] */
    DC_P_BASIC_B,
    DC_P_UNIT_W,      /* affine_unit_{ah,eh,anh,enh,ath,eth}.weight, contiguous [6,128,128] */
    DC_P_UNIT_B,      /* ... .bias contiguous [6,128] */
    DC_P_PRE_W,       /* affine_pre_rnn.weight [256,896] */
    DC_P_PRE_B,
    DC_P_HEADS_W,     /* [154,H]: affine_unit_attention(128) | head_enum(4) | move_x(9) | move_y(9) |
                         head_ability(3) | affine_value(1), contiguous */
    DC_P_HEADS_B,     /* [154] same order */
    DC_P_RNN0,        /* + 4*l: rnn.weight_ih_l, weight_hh_l, bias_ih_l, bias_hh_l */
    DC_P_COUNT_FIXED = DC_P_RNN0
};

/* workspace buffer ids (for dc_workspace_layout's offsets[]; bytes) */
enum dc_ws_index {
    DC_WS_FAULT = 0,        /* i32[8] at workspace offset 0 (whatever the dims): fault record, see below */
    DC_WS_BASIC, DC_WS_EMB, DC_WS_DEMB, DC_WS_XCAT, DC_WS_AMAX, DC_WS_PRE, DC_WS_HEADOUT, DC_WS_TU,
    DC_WS_DHEADOUT, DC_WS_DTU, DC_WS_DPRE, DC_WS_DXCAT, DC_WS_STATS, DC_WS_WHHT, DC_WS_SCRATCH, DC_WS_HEADW_PAD,
    DC_WS_TEAM_XBUF, DC_WS_WPLANES,
    DC_WS_FIXED,            /* per-layer blocks follow */
    DC_WSL_GATES = 0, DC_WSL_HN, DC_WSL_HSEQ, DC_WSL_HPREV, DC_WSL_CSEQ, DC_WSL_CPREV, DC_WSL_DGX, DC_WSL_DGH,
    DC_WSL_DC, DC_WSL_DH,
    DC_WS_PER_LAYER
};

/* DC_WS_FAULT - where a failure of the team kernels (H = 256: four or eight workgroups per team; LSTM-512 in bf16 mode: sixteen) is
 * reported.  Those kernels hand state between the workgroups of a team through tagged granules; a member that polls one for ~1 s without seeing its tag gives up, NaN-poisons its outputs (the loss
 * turns NaN: status 1 of dc_gradnorm_clip_adam, the reference's own guard, optimizer.py:667-669) and - first writer wins - records
 *   [0] DC_FAULT_TEAM_TIMEOUT + kernel (1 rnn_team_fwd, 2 rnn_team_bwd, 3 team_mfma_fwd, 4 team_mfma_bwd, 5 lstm512_team_fwd,
 *   6 lstm512_team_bwd, 7 team8_fwd, 8 team8_bwd), [1] layer, [2] team,
 *   [3] member, [4] time step, [5] sequence, [6] the tag it waited for, [7] reserved.
 * The record is STICKY: the library never clears it.  The owner of the workspace zeroes these 32 bytes once after allocating it
 * (and again after reading a fault, if it wants to carry on). */
#define DC_FAULT_TEAM_TIMEOUT 16

/* Byte offsets of every workspace buffer into offsets[DC_WS_FIXED + DC_WS_PER_LAYER*layers]; returns
 * the total workspace size in bytes (host-only helper, no GPU work). */
int64_t dc_workspace_layout(const dc_dims* dims, int64_t* offsets);

/* Replaces Policy.forward (policy.py:92-167) for a packed batch.
 *   obs [rows,483] f32; h0/c0 [layers,B,H] (NULL = zeros; c0 LSTM only); seq_off i64[B], seq_len i32[B];
 *   hT/cT [layers,B,H] out (may be NULL).  Results live in the workspace: DC_WS_HEADOUT [rows,160]
 *   (cols 0..127 attention query, 128..131 enum, 132..140 x, 141..149 y, 150..152 ability, 153 value)
 *   and DC_WS_TU [rows,40] (target_unit logits).  *   unit_mask (may be NULL; only read with DC_DIMS_LAZY_TU and DC_DIMS_F16X2): the batch's action masks u8 [rows,65].  The per-unit
 *   embeddings of the two 16-unit types and of the three 1-unit types (35 of the 40 units, 1.2 GB per configs[2] pass) have ONE consumer -
 *   the target-unit attention of dc_select_logp / dc_ppo_loss_fwd_bwd / dc_policy_backward, which reads the rows of units whose mask byte
 *   (column 22 + unit) is set and nothing else - so with the masks at hand DC_WS_EMB rows of masked-out units of those types are NOT
 *   written (they keep whatever the buffer held).  NULL: every row is written. */
int dc_policy_forward(const dc_dims* dims, const float* params, const int64_t* poff_host, const float* obs,
                      const float* h0, const float* c0, const int64_t* seq_off, const int32_t* seq_len, void* ws,
                      float* hT, float* cT, const uint8_t* unit_mask, dc_stream_t stream);

/* Hidden-state carry of the rollout pass, optimizer.py:384,408 (`hidden = hidden.detach()` handed from chunk to chunk) and
 * policy.py:77-78 (zeros for a rollout's first chunk): after dc_policy_forward over whole rollouts (`dims`, `ws` = that
 * call's), the initial state of chunk b of the seq_len view is the recurrent state at row prev_row[b] (= the chunk's first
 * row - 1), or zeros when prev_row[b] < 0.
 *   prev_row i64[n_chunks] (device); h0 / c0 [layers, n_chunks, H] f32 out (c0: LSTM only, may be NULL for the GRU). */
int dc_chunk_initial_state(const dc_dims* dims, const void* ws, const int64_t* prev_row, int n_chunks, float* h0, float* c0,
                           dc_stream_t stream);

/* policy.py:80-84 `Policy.single` (what a rollout actor calls per env-step, agent.py:652): ONE env-step of ONE hero as one kernel
 * (csrc/policy_single.hip, round 6) - no workspace, no padding to tiles, exact f32 arithmetic, 17-21 us per launch.  An entry point added in
 * round 6: callers of ABI 4 that do not use it are unaffected.
 *   dims: cell, hidden (a multiple of 64 up to 512), layers are read; obs f32[483]: env(3) | 40 units x 12 (layout.py) - device memory OR
 *   page-locked host memory (every workgroup reads the row once: an actor hands over its pinned row, no copy);
 *   h0 / c0 f32[layers, H] (device; NULL = zeros; c0 LSTM only); out f32[200] = the head row (DC_WS_HEADOUT's 160 columns) | the 40
 *   target-unit logits; hT / cT f32[layers, H] out (cT may be NULL for the GRU; hT / cT must not alias h0 / c0);
 *   scratch f32[DC_SINGLE_SCRATCH_FLOATS], 8-byte aligned: ZERO before the first call, afterwards left to this function (the {value, tag}
 *   granules its stages hand over and its launch generation).  One call at a time per scratch buffer (calls on one stream are).
 *   Errors: 1020 layers, 1021 cell, 1022 hidden, 1024 a NULL buffer, 1025 alignment.  A stage whose input never arrives (its
 *   producers never ran) gives up after ~1 s and hands on NaN: no hang. */
#define DC_SINGLE_SCRATCH_FLOATS 20480
int dc_policy_single(const dc_dims* dims, const float* params, const int64_t* poff_host, const float* obs, const float* h0, const float* c0,
                     float* out, float* hT, float* cT, float* scratch, dc_stream_t stream);

/* Rollout-pass epilogue, optimizer.py:387-390 + policy.py:169-178: log-prob of the selected action
 * per head (0 where the head took no action), value, masked argmax per head (-1 on empty mask).
 *   act/mask u8 [rows,65]; logp_sel f32 [rows,5]; values f32 [rows]; argmax i32 [rows,5] or NULL. */
int dc_select_logp(const dc_dims* dims, const void* ws, const uint8_t* act, const uint8_t* mask, float* logp_sel,
                   float* values, int32_t* argmax, dc_stream_t stream);

/* Loss of DotaOptimizer.train, optimizer.py:587-665, and its gradient w.r.t. logits/value (written
 * into the workspace for dc_policy_backward).  adv: RAW advantages (normalised inside, optimizer.py:588).
 *   losses_out f32[9] = loss, policy_loss, entropy_loss, value_loss, entropy[enum,x,y,target_unit,ability]
 *   head_on i32[5] = 1 if that head took at least one action in the batch. */
int dc_ppo_loss_fwd_bwd(const dc_dims* dims, void* ws, const uint8_t* act, const uint8_t* mask, const float* old_logp,
                        const float* adv, const float* ret, float* losses_out, int32_t* head_on, float e_clip,
                        float entropy_coef, float vf_coef, dc_stream_t stream);

/* loss.backward() of optimizer.py:672 for the network: fills the flat gradient buffer
 * (total_floats long, same offsets as the parameters; overwritten). */
int dc_policy_backward(const dc_dims* dims, const float* params, const int64_t* poff_host, float* grads,
                       int64_t total_floats, const float* obs, const int64_t* seq_off, const int32_t* seq_len, void* ws,
                       dc_stream_t stream);

/* optimizer.py:674-681: mean_gradient_norm (before/after), clip_grad_norm_(0.5), NaN guards, Adam.
 *   seg_off i64 / seg_len i32 / seg_gate i32 [n_seg] (device): the named parameters inside the flat
 *   buffer; seg_gate -1 = always has a gradient, 0..4 = only when head k acted, 5 = only if vf_coef>0;
 *   m, v: Adam moments (flat); segsq f64[n_seg * (1 + ceil(max_seg_len / 4096))] (scratch, ZERO before the first call and not to be touched
 *   between calls: per-segment totals, the per-chunk partial sums, then sixteen arrival counters the call leaves at zero),
 *   ctl f32[4] (ZERO before the first call; [0] clip coefficient, [1] 1.0 = the update was applied, [2] an arrival counter the call leaves
 *   at zero, [3] a release generation that grows by one per call),
 *   seg_step i32[n_seg] (persistent step counters), status i32[1] (0 ok, 1 NaN loss, 2 NaN or infinite grad norm: nothing updated; STICKY - while it is non-zero every
 *   later call skips its update too, the caller clears it after handling the error) - all device;
 *   norms_out f32[2] = unclipped, clipped mean gradient norm. */
int dc_gradnorm_clip_adam(const int64_t* seg_off, const int32_t* seg_len, const int32_t* seg_gate, int n_seg,
                          int max_seg_len, float* params, float* grads, float* m, float* v, double* segsq,
                          const int32_t* head_on, const float* losses, float* norms_out, float* ctl,
                          int32_t* seg_step, int32_t* status, float max_norm, float vf_coef, double lr, double beta1,
                          double beta2, float eps, dc_stream_t stream);

/* Replaces the per-parameter averaging of DistributedDataParallelSparseParamCPU (distributed.py:24-57)
 * AFTER the caller has SUM-all-reduced the flat gradient bucket (RCCL via torch.distributed): every
 * parameter is divided by the number of ranks that had a gradient for it.
 *   counts f32[6] (device): [k<5] = ranks whose head k acted in their shard, [5] = world size.
 *   vf_coef: RESERVED, ignored (kept so that the signature stays ABI 4: the value head's parameters have a gradient on every rank
 *   whatever the coefficient - a zero one - so their divisor is always counts[5]). */
int dc_dp_average_grads(const int64_t* seg_off, const int32_t* seg_len, const int32_t* seg_gate, int n_seg,
                        int max_seg_len, float* grads, const float* counts, float vf_coef, dc_stream_t stream);

/* Host-side ingest (no device work): a list of 2-D copies src[i] (rows[i] x row_bytes[i], contiguous; 0 = write
 * zeros) -> dst[i] (row stride dst_stride[i] bytes), executed by up to n_threads host threads.  Fills the page-locked
 * staging buffers of a batch from the wire-format arrays of its rollouts in one call - replaces the per-key slicing
 * of optimizer.py:353-365 (one interpreter-level copy per key per chunk).  Pointers are passed as int64. */
int dc_pack_rows(const int64_t* src, const int64_t* dst, const int64_t* rows, const int64_t* row_bytes,
                 const int64_t* dst_stride, int64_t n_items, int n_threads);

/* Measurement aid for bench.py (not part of the reference surface): when enabled, every GEMM call and
 * every recurrent step launch is bracketed by two HIP events on the launch stream.  dc_profile_report
 * synchronises, fills per-region sums (names: 64-byte slots) and returns the region count. */
int dc_profile_enable(int on);
int dc_profile_report(char* names, int64_t* launches, double* total_ms, double* flops, double* bytes, int max_regions);

#ifdef __cplusplus
}
#endif
#endif
