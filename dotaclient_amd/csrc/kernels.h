// Internal prototypes shared by the translation units of libdotaclient_hip.so.
#pragma once
#include "../../include/dotaclient_hip.h"   // dc_dims, DC_DIMS_* flags, workspace ids
#include "common.h"

namespace dc {

struct RnnStepArgs {
    const int64_t* seq_off;
    const int32_t* seq_len;
    int n_seq, H, t;
    int flags;             // dc_dims.flags (DC_DIMS_* kernel-selection overrides)
    void* xbuf;            // DC_WS_TEAM_XBUF: exchange ring of the H = 256 team kernels (caller workspace)
    int* fault;            // DC_WS_FAULT: sticky fault record (nullptr: none)
    int layer;             // which recurrent layer this call works on (reported in the fault record)
    // forward
    const float* Whh;      // [G*H][H]
    const uint16_t* Whh_bf;   // bf16 mode: W_hh as bf16 [G*H][H] (nullptr: not provided)
    const uint16_t* WhhT_bf;  // bf16 mode: W_hh^T as bf16 [H][G*H]
    uint16_t* stepbf;         // bf16 mode: [2][n_seq][G*H] bf16, step-major copies of h_t (forward) / the gate gradients (backward)
    const float* bhh;      // [G*H]
    float* gates;          // [rows][G*H] in: W_ih x + b_ih, out: activated gates
    float* hn;             // [rows][H]   GRU: W_hn h + b_hn
    float* hseq;           // [rows][H]
    float* hprev;          // [rows][H]
    float* cseq;           // [rows][H]   LSTM
    float* cprev;          // [rows][H]   LSTM
    const float* h0;       // [n_seq][H] initial state of this layer (nullptr = zeros); rnn_forward_layer seeds hprev/cprev[first row]
    const float* c0;       // [n_seq][H] LSTM
    // backward
    const float* WhhT;     // [H][G*H]
    float* dh;             // [rows][H]  in: dL/dh_t from above, out: total dL/dh_t
    float* dc;             // [rows][H]  LSTM: total dL/dc_t
    float* dgx;            // [rows][G*H] grad wrt (W_ih x + b_ih)
    float* dgh;            // [rows][G*H] grad wrt (W_hh h + b_hh)   (GRU; LSTM: == dgx)
    float s_grad;          // DC_DIMS_F16X2: power-of-two pre-scale of the gate gradients for rnn_team_mfma.hip's backward on f16 planes (0: f32 product)
    int fwd_only;          // rnn_team_mfma.hip's forward (LSTM-256, > 128 sequences): no backward will read this pass - skip the stores only it would need (DC_DIMS_FWD_ONLY)
    int bf16_store;        // rnn_team512.hip only (configs[4]): `gates`, `dgx` [rows][G*H] and `hseq`, `hprev` [rows][H] are bf16 buffers (policy.hip: bf16_store())
    long long* dbg;        // DC_LSTM_TIMING=1: phase cycle sums of workgroup 0 (else nullptr)
};

// fill.hip: zero-fill / small copies as kernels of this library (a captured epoch holds kernel nodes only)
int zero_async(void* p, size_t bytes, hipStream_t s);
int zero2d_f32_async(float* p, long long ld, int width, long long rows, hipStream_t s);
int copy_f32_async(float* dst, const float* src, long long n, hipStream_t s);

// prof.hip (inert unless dc_profile_enable(1))
bool prof_enabled();
int prof_begin(const char* name, double flops, double bytes, hipStream_t s);
void prof_end(int handle, hipStream_t s);
struct ProfScope {
    int h; hipStream_t s;
    ProfScope(const char* name, double flops, double bytes, hipStream_t st) : h(prof_begin(name, flops, bytes, st)), s(st) {}
    ~ProfScope() { prof_end(h, s); }
};

// gae.hip
int gae_scan(const float* rewards, const float* values, const int64_t* seq_off, const int32_t* seq_len, int n_seq,
             int max_len, double gamma, double lam, float* adv, float* ret, hipStream_t stream);
int discount(const float* x, int n, double gamma, float* y, hipStream_t stream);
int advantage_returns(const float* rewards, const float* values, int L, double gamma, double lam, float* adv, float* ret,
                      hipStream_t stream);
// gemm.hip
// split-K partial slabs of ONE call (reduced by a second kernel); p == nullptr -> fp32 atomics.  Passed per call: the
// library keeps no scratch pointer of its own (two callers on two streams never share one)
struct GemmScratch { float* p = nullptr; long long floats = 0; };
int gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int a_kmajor,
             int b_kmajor, const float* bias, int relu, const float* aux, int ldaux, int accumulate, int splits,
             hipStream_t stream, GemmScratch sc = GemmScratch());
int gemm_f32_tn_pair(const float* A, int lda, const float* B1, int ldb1, int N1, const float* B2, int ldb2, int N2, float* C1,
                     int ldc1, float* C2, int ldc2, int M, int K, hipStream_t stream, GemmScratch sc);   // [C1 | C2] = A^T [B1 | B2], one launch
int splitk_reduce(const float* slab, float* C, int M, int N, int ldc, int splits, hipStream_t s);   // C = sum_z slab[z]
// C (+)= sum_z slab[z]; columns >= n_split (> 0) go to C2 (the dW_ih | dW_hh pair)
int splitk_reduce_pair(const float* slab, float* C, int M, int N, int ldc, int splits, int accumulate, float* C2, int ldc2,
                       int n_split, hipStream_t s);
// gemm_x3.hip: the dense products on the bf16 matrix cores (prec 6: f32-grade by exact 3-way splitting; prec 1: plain bf16)
enum { X3_ROW = 0, X3_KMAJ = 1, X3_PLANES = 2, X3_KMAJ16 = 3 /* internal: a k-major operand stored as bf16 */ };
struct X3Gemm {
    const void* A = nullptr; int a_mode = X3_ROW, lda = 0;             // X3_ROW: f32 [M][lda]; X3_KMAJ: f32 [K][lda]
    const void* B = nullptr; int b_mode = X3_PLANES, ldb = 0;          // X3_PLANES: bf16 [planes][N][ldb]; X3_KMAJ: f32 [K][ldb]
    long long b_plane = 0;                                            // elements between planes
    const void* B2 = nullptr; int ldb2 = 0; long long b2_plane = 0;   // second B behind output column n_split
    float* C = nullptr; int ldc = 0; float* C2 = nullptr; int ldc2 = 0; int n_split = 0;
    int M = 0, N = 0, K = 0;
    const float* bias = nullptr; int nbias = 0;                       // bias[col] for col < nbias
    int relu = 0; const float* aux = nullptr; int ldaux = 0;          // zero where aux <= 0
    int accumulate = 0, prec = 6, transposed_w = 0;     // prec 6: three bf16 pieces, six MFMAs; 4: two f16 pieces, three MFMAs; 1: bf16
    float sa = 1.f, sb = 1.f;                           // prec 4: power-of-two pre-scales of A / B (an X3_PLANES operand was scaled by sb
                                                        // when its planes were made); the product is scaled back by 1 / (sa sb)
    GemmScratch scratch;
    // bf16 storage (prec 1; configs[4]): the operand / the output / the relu mask `aux` live in HBM as bf16 with the same shape and ld
    // (elements).  b_bf16: X3_KMAJ B only (a second B is bf16 as well); c_bf16: no accumulate, no C2, no split-K (no scratch).
    int a_bf16 = 0, b_bf16 = 0, c_bf16 = 0, aux_bf16 = 0;
    int tile128 = 0;              // 1: keep the 128 x 128 split-on-load kernel where gemm_x3s.hip's row-streaming kernel would take the product (A/B, tests)
    float* a_colsum = nullptr;    // X3_KMAJ A only: a_colsum[m] += sum_k A[k][m] (the bias gradient that goes with a weight gradient),
                                  // summed while the tiles pass through the loader - no second pass over A
};
bool gemm_x3_shape_ok(int M, int N, int K, int lda, int ldb, int a_mode, int b_mode);
int gemm_x3(const X3Gemm& g, hipStream_t stream);
// gemm_x3s.hip: the row-streaming kernel for prec 4, X3_ROW x X3_PLANES (both operands by LDS-DMA, 256 x 128 tiles, epilogue from the
// registers); gemm_x3 dispatches to it when the shape is eligible and tile128 is clear
bool gemm_x3s_eligible(const X3Gemm& g);
int gemm_x3s(const X3Gemm& g, hipStream_t stream);
// weight matrices -> bf16 planes [prec == 1 ? 1 : 3][rows_pad (zero rows past `rows`)][cols], or of the transpose
struct X3SplitJob { const float* src; uint16_t* dst; int rows, cols, ld, transpose, rows_pad; };
int split_weight_planes(const X3SplitJob* jobs, int n, int prec, hipStream_t s, float scale = 1.f);   // prec 4: [2] f16 planes of w * scale
// n_groups (<= 8) independent reductions in one launch: C[g] (dense M x N) = sum of slabs [begin[g], begin[g+1])
int splitk_reduce_grouped(const float* slab, float* C, int M, int N, int n_groups, const int* begin, hipStream_t s);
// embed.hip
int unit_basic_fwd(const float* obs, const float* W1, const float* b1, float* basic, long long nr, hipStream_t s);
// nrp (here and below): env-steps per unit of the type-major emb / d(emb) blocks = nr padded to a multiple of 128 on the fused path
int pool_env_fwd(const float* obs, const float* emb, const float* Wenv, const float* benv, float* xcat, uint8_t* amax,
                 long long nr, long long nrp, int residual, hipStream_t s);
// skip16 = 1: the two 16-unit types are left out (no d(emb) rows written, no db2 contribution): embed_bwd_pool16 has them; 2: no d(emb) rows at all
// (embed_small.hip forms the small types' on chip), their db2 still summed here; 3: env-embedding gradient only (embed_env_bwd_kernel)
int embed_scatter_bwd(const float* obs, const float* xcat, const float* dxcat, const float* dtu, const float* q, int ldq,
                      const uint8_t* amax, float* demb, float* dWenv, float* dbenv, float* db2, float* scratch,
                      long long nr, long long nrp, int skip16, hipStream_t s);
int unit_basic_bwd(const float* obs, const float* dbasic, float* dW1, float* db1, float* scratch, long long nr,
                   hipStream_t s);
// p3 != nullptr: embed_small.hip's [256][128] column sums -> db2_small [6][128] rows 0, 1, 4, 5 (workgroups 0-31, 32-191, 192-223, 224-255)
int embed_tail_reduce(const float* pa, int na, const float* pb, int nb, float* dW1, float* db1, const float* p2, int n2,
                      float* db2, hipStream_t s, const float* p3 = nullptr, float* db2_small = nullptr);
int colsum(const float* X, int ld, long long rows, int cols, float* out, hipStream_t s);
int unit_basic_reduce(const float* partials, int nblk, float* dW1, float* db1, hipStream_t s);   // partials [nblk][13][128]
// embed_fused.hip (rows % 128 == 0: first embedding layer recomputed on chip, `basic` never stored)
bool embed_fused_supported(long long nr_padded);
// xcat/amax != nullptr: the max-pools of the one-unit and 16-unit types are produced by the epilogue (then call
// pool_env_fwd with residual = 1 for the env embedding and the 5-unit type only)
// W2p != nullptr: W2 also as pre-split bf16 planes [3][6 x 128][128] (split_weight_planes): no fragment split for that operand
// F16x2Scales (DC_DIMS_F16X2): the 128 x 128 layer and its two backward products from two f16 pieces per operand and three MFMAs; the
// power-of-two pre-scales of activations, weights, gradients (policy.hip).  W2p then holds [2] f16 planes of W2 * s_w.
struct F16x2Scales { bool on = false; float s_act = 1.f, s_w = 1.f, s_grad = 1.f; };
int embed_fwd_fused(const float* obs, const float* W1, const float* b1, const float* W2, const uint16_t* W2p, const float* b2, float* emb,
                    float* xcat, uint8_t* amax, long long nr_valid, long long nr_padded, hipStream_t s, F16x2Scales f16 = F16x2Scales(),
                    const uint8_t* unit_mask = nullptr,    // unit_mask (f16 variant only): emb rows of masked-out units are not stored
                    const float* Wenv = nullptr, const float* benv = nullptr);   // f16 variant only: the env embedding and the five-unit pool in
                                                                                  // this kernel too - pool_env_fwd is then not needed at all
// inputs of the sparse max-pool backward of the two 16-unit types (embed_sparse.hip); db2 [6][128] is accumulated into
// (prep: 2 * nr * 736 floats of scratch - it lives in the d(emb) rows of the two types, which the sparse path never writes)
struct EmbSparseIn { const float* dxcat; const uint8_t* amax; const float* dtu; const float* q; int ldq; float* db2; float* prep;
                     int eight_waves = 0; int valu = 0;         // valu: keep embed_sparse.hip's kernels in f16x2 mode too (DC_DIMS_POOL16_VALU)
                     int small_dense = 0;                       // small_dense: the four small types through d(emb) in HBM and the dense kernels (DC_DIMS_SMALL_DENSE)
                     int small_db2 = 0;
                     const uint16_t* w2t_planes = nullptr; };   // f16x2: [6][2 planes (of 3 slots)][128 k][128 c] = W2_t^T x s_w of types 2, 3 (policy_backward's pre-pass): R = q W2_t as a product of the same kind                      // embed_small.hip also sums the small types' second-layer bias gradients (then embed_scatter_bwd: env only)
// the four small types' backward fused (embed_small.hip) - decided in one place: policy.hip (what embed_scatter_bwd writes) and
// embed_bwd_fused (what it launches) must agree
inline bool embed_small_fused(bool sparse16, const F16x2Scales& f16, const EmbSparseIn* sp) {
    return sparse16 && f16.on && sp && !sp->valu && !sp->eight_waves && !sp->small_dense;
}
int embed_bwd_fused(const float* obs, const float* demb, const float* W1, const float* b1, const float* W2, float* dW2,
                    float* dW1, float* db1, float* scratch, long long scratch_floats, long long nr_valid, long long nr_padded,
                    const EmbSparseIn* sp, hipStream_t s, F16x2Scales f16 = F16x2Scales());
// embed_sparse.hip
int embed_bwd_pool16(const float* obs, const float* dxcat, const uint8_t* amax, const float* dtu, const float* q, int ldq,
                     const float* W1, const float* b1, const float* W2, float* slab, float* part1, float* part2, float* prep,
                     long long nr, int wg_per_type, hipStream_t s, int eight_waves = 0);   // eight_waves: round 2's 512-thread kernel (DC_DIMS_POOL16_8W)
// embed_pool16m.hip: the same gradient as dense f16x2 products with on-chip operands (same partial formats; needs F16x2Scales.on)
int embed_bwd_pool16m(const float* obs, const float* dxcat, const uint8_t* amax, const float* dtu, const float* q, int ldq,
                      const float* W1, const float* b1, const float* W2, float* slab, float* part1, float* part2, float* scratch_r,
                      long long nr, int wg_per_type, hipStream_t s, const F16x2Scales& f16,       // scratch_r: 2 * nr * 128 floats
                      const uint16_t* w2t_planes = nullptr);      // EmbSparseIn::w2t_planes (NULL: R = q W2_t by the exact-f32 tile kernel)
// embed_small.hip: the four small types (needs F16x2Scales.on); slab / part in the dense kernels' formats
int embed_bwd_small(const float* obs, const float* dxcat, const uint8_t* amax, const float* dtu, const float* q, int ldq, const float* W1,
                    const float* b1, const float* W2, float* slab, int slab_skip, float* part, float* db2part, long long nr, hipStream_t s,
                    const F16x2Scales& f16);
// policy_single.hip: Policy.single as one kernel
int policy_single(const dc_dims* d, const float* params, const int64_t* poff, const float* obs, const float* h0, const float* c0, float* out,
                  float* hT, float* cT, float* scratch, hipStream_t s);
// heads.hip
int attn_logits(const float* headout, const float* emb, float* tu, long long nr, long long nrp, hipStream_t s);
// target-unit logits of the units whose mask byte (mask[n][22 + u]) is set; 0 elsewhere
int attn_logits_masked(const float* headout, const float* emb, const uint8_t* mask, float* tu, long long nr, long long nrp, hipStream_t s);
int attn_bwd_q(const float* dtu, const float* emb, float* dheadout, long long nr, long long nrp, hipStream_t s);
int select_logp(const float* headout, const float* tu, const uint8_t* act, const uint8_t* mask, float* logp_sel,
                float* values, int32_t* argmax, long long nr, hipStream_t s);
int ppo_loss_fwd_bwd(const float* headout, const float* tu, const uint8_t* act, const uint8_t* mask, const float* old_logp,
                     const float* adv, const float* ret, double* stats, float* dheadout, float* dtu, float* losses_out,
                     int32_t* head_on, long long nr, float e_clip, float entropy_coef, float vf_coef, hipStream_t s);
// rnn.hip
int transpose(const float* in, float* out, int rows, int cols, hipStream_t s);
// bf16 (here and below): the state buffer (hprev / hseq / seq) holds bf16 elements - configs[4]'s bf16 storage (policy.hip)
int rnn_seed_state(const float* h0, float* hprev, const int64_t* seq_off, const int32_t* seq_len, int n_seq, int H,
                   hipStream_t s, int bf16 = 0);
int rnn_final_state(const float* hseq, float* hT, const int64_t* seq_off, const int32_t* seq_len, int n_seq, int H,
                    hipStream_t s, int bf16 = 0);
int rnn_gather_state(const float* seq, const int64_t* prev_row, float* out, int n, int H, hipStream_t s, int bf16 = 0);
bool policy_bf16_store(const dc_dims* d);      // policy.hip: gate buffers, pre, hseq / hprev stored as bf16 for these dims
bool rnn_uses_persistent(int cell, int H, int flags);   // register-resident LSTM kernels: W_hh^T is not needed
int rnn_forward_layer(int cell, RnnStepArgs a, int max_len, hipStream_t s);
int rnn_backward_layer(int cell, RnnStepArgs a, int max_len, hipStream_t s);
// rnn_persist.hip (LSTM, H <= 128: all time steps in one launch, W_hh register-resident)
bool lstm_persist_supported(int H);
int lstm_forward_persist(RnnStepArgs a, int max_len, hipStream_t s);
int lstm_backward_persist(RnnStepArgs a, int max_len, hipStream_t s);
// rnn_persist_valu.hip (same contract, one sequence per workgroup on the packed-f32 VALU: the low-latency
// variant for batches of <= 2 sequences per CU; lstm_*_persist dispatch to it)
bool lstm_persist_use_valu(int n_seq, int flags);
int lstm_forward_valu(RnnStepArgs a, hipStream_t s);
int lstm_backward_valu(RnnStepArgs a, hipStream_t s);
// rnn_team.hip (GRU / LSTM with H = 256: all time steps in one launch, a sequence's W_hh spread over the registers
// of four workgroups that exchange the state every step)
bool rnn_team_supported(int cell, int H, int n_seq, int flags);
// rnn_step_bf16.hip: LSTM recurrence steps on the bf16 MFMA (DC_DIMS_BF16, H = 512 / 1024)
bool lstm_step_bf16_supported(int cell, int H, int flags, const void* Wb);
int lstm_forward_steps_bf16(RnnStepArgs a, int max_len, hipStream_t s);
int lstm_backward_steps_bf16(RnnStepArgs a, int max_len, hipStream_t s);
long long rnn_team_xbuf_bytes();   // size of DC_WS_TEAM_XBUF (H = 256)
// rnn_team512.hip: persistent bf16 LSTM-512 (teams of sixteen workgroups, 32-sequence MFMA tiles)
bool lstm_team512_supported(int cell, int H, int flags, const void* Wb);
int lstm_team512_forward(RnnStepArgs a, int max_len, hipStream_t s);
int lstm_team512_backward(RnnStepArgs a, int max_len, hipStream_t s);
long long lstm_team512_xbuf_bytes();   // size of DC_WS_TEAM_XBUF (H = 512)
// rnn_team_mfma.hip (LSTM-256, more than 128 sequences: a team advances four sequences together on the 4x4x1 MFMA)
bool lstm_team_mfma_supported(int cell, int H, int n_seq, int flags, bool backward);
int lstm_team_mfma_forward(int cell, RnnStepArgs a, int max_len, int n_teams, hipStream_t s);
int lstm_team_mfma_backward(int cell, RnnStepArgs a, int max_len, int n_teams, hipStream_t s);
// rnn_team8.hip (DC_DIMS_TEAM8: the same in teams of eight workgroups, two workgroups per CU)
bool rnn_team8_supported(int cell, int H, int n_seq, int flags);
int rnn_team8_forward(int cell, RnnStepArgs a, int max_len, hipStream_t s);
int rnn_team8_backward(int cell, RnnStepArgs a, int max_len, hipStream_t s);
long long rnn_team8_xbuf_bytes();
int rnn_team_forward(int cell, RnnStepArgs a, int max_len, hipStream_t s);
int rnn_team_backward(int cell, RnnStepArgs a, int max_len, hipStream_t s);
// adam.hip
int gradnorm_clip_adam(const int64_t* seg_off, const int32_t* seg_len, const int32_t* seg_gate, int n_seg, int max_seg_len,
                       float* param, float* grad, float* m, float* v, double* segsq, const int32_t* head_on,
                       const float* losses, float* norms_out, float* ctl, int32_t* seg_step, int32_t* status,
                       float max_norm, float vf_coef, double lr, double beta1, double beta2, float eps, hipStream_t s);

int dp_average_grads(const int64_t* seg_off, const int32_t* seg_len, const int32_t* seg_gate, int n_seg, int max_seg_len,
                     float* grad, const float* counts, float vf_coef, hipStream_t s);

}  // namespace dc
