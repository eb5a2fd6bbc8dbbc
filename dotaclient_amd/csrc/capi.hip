// extern "C" surface of libdotaclient_hip.so - see include/dotaclient_hip.h for the contract.
#include "../../include/dotaclient_hip.h"
#include "kernels.h"
#include <stdio.h>
#include <string.h>

namespace dc {

static thread_local char g_err[512] = "";

void set_error(const char* what, int code) {
    snprintf(g_err, sizeof(g_err), "%s (code %d%s%s)", what, code, code < 1000 ? ": " : "",
             code < 1000 ? hipGetErrorString((hipError_t)code) : "");
}

int launch_check(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(what, (int)e);
        return (int)e;
    }
    return 0;
}

int64_t workspace_layout(const dc_dims* d, int64_t* out);
long long emb_rows_of(const dc_dims* d);   // rows per unit of the type-major emb blocks (policy.hip)
int policy_forward(const dc_dims* d, const float* params, const int64_t* poff, const float* obs, const float* h0,
                   const float* c0, const int64_t* seq_off, const int32_t* seq_len, void* ws_base, float* hT, float* cT, const uint8_t* unit_mask,
                   hipStream_t s);
int policy_backward(const dc_dims* d, const float* params, const int64_t* poff, float* grads, int64_t total_floats,
                    const float* obs, const int64_t* seq_off, const int32_t* seq_len, void* ws_base, hipStream_t s);

}  // namespace dc

// hipGetLastError() is sticky per thread and shared with every other user of the HIP runtime in the process (PyTorch
// probes devices and pointer attributes and may leave e.g. hipErrorNoDevice / hipErrorInvalidValue behind): clear it
// on entry, so that launch_check reports only what THIS call enqueued.
#define DC_ENTER() (void)hipGetLastError()

extern "C" {

int dc_abi_version(void) { return DC_ABI_VERSION; }
const char* dc_last_error(void) { return dc::g_err; }

int dc_gae_scan(const float* rewards, const float* values, const int64_t* seq_off, const int32_t* seq_len,
                int n_seq, int max_len, double gamma, double lam, float* adv, float* ret, dc_stream_t stream) {
    DC_ENTER();
    return dc::gae_scan(rewards, values, seq_off, seq_len, n_seq, max_len, gamma, lam, adv, ret, (hipStream_t)stream);
}

int dc_discount(const float* x, int n, double gamma, float* y, dc_stream_t stream) {
    DC_ENTER();
    return dc::discount(x, n, gamma, y, (hipStream_t)stream);
}

int dc_advantage_returns(const float* rewards, const float* values, int L, double gamma, double lam, float* adv, float* ret,
                         dc_stream_t stream) {
    DC_ENTER();
    return dc::advantage_returns(rewards, values, L, gamma, lam, adv, ret, (hipStream_t)stream);
}

int dc_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                int a_kmajor, int b_kmajor, const float* bias, int relu, const float* aux, int ldaux,
                int accumulate, int splits, float* scratch, int64_t scratch_floats, dc_stream_t stream) {
    DC_ENTER();
    return dc::gemm_f32(A, B, C, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, bias, relu, aux, ldaux, accumulate,
                        splits, (hipStream_t)stream, dc::GemmScratch{scratch, (long long)scratch_floats});
}

int dc_gemm_x3(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
               int a_kmajor, int b_kmajor, const float* bias, int relu, const float* aux, int ldaux,
               int accumulate, int prec, float* scratch, int64_t scratch_floats, dc_stream_t stream) {
    DC_ENTER();
    using namespace dc;
    hipStream_t s = (hipStream_t)stream;
    X3Gemm g;
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.nbias = N; g.relu = relu; g.aux = aux; g.ldaux = ldaux;
    // prec: low byte 1 / 4 / 6; for 4 (two f16 pieces) bits 8..15 / 16..23 = log2 of A's / B's power-of-two pre-scale, signed bytes
    // (DC_GEMM_PREC_F16X2(la, lb), include/dotaclient_hip.h) - what policy.hip passes for activations (2^4), weights (2^8), gradients
    const int form = prec & 0xff;
    g.accumulate = accumulate; g.prec = form == 1 ? 1 : (form == 4 ? 4 : 6);
    g.tile128 = (prec >> 24) & 1;
    if (g.prec == 4) {
        g.sa = ldexpf(1.f, (int)(int8_t)((prec >> 8) & 0xff));
        g.sb = ldexpf(1.f, (int)(int8_t)((prec >> 16) & 0xff));
    }
    if (g.prec == 1) {      // bf16 storage of an operand / the output / the mask (DC_GEMM_PREC_BF16_STORE): same shapes and ld, 2-byte elements
        g.a_bf16 = (prec >> 8) & 1; g.b_bf16 = (prec >> 9) & 1; g.c_bf16 = (prec >> 10) & 1; g.aux_bf16 = (prec >> 11) & 1;
        if (g.b_bf16 && !(a_kmajor && b_kmajor)) { set_error("dc_gemm_x3: a bf16-stored B is built for the k-major form only", 1005); return 1005; }
    }
    if (a_kmajor && b_kmajor) {
        g.A = A; g.a_mode = X3_KMAJ; g.lda = lda; g.B = B; g.b_mode = X3_KMAJ; g.ldb = ldb;
        if (!g.c_bf16) g.scratch = GemmScratch{scratch, (long long)scratch_floats};
        return gemm_x3(g, s);
    }
    if (a_kmajor) { set_error("dc_gemm_x3: a_kmajor needs b_kmajor", 1007); return 1007; }
    const long long plane_floats = ((long long)3 * N * K + 1) / 2;
    if (scratch == nullptr || scratch_floats < plane_floats) { set_error("dc_gemm_x3: scratch too small for the weight planes", 1008); return 1008; }
    // B: [N][ldb] (x W^T) or [K][ldb] (dy W): planes [3][N][K] either way
    X3SplitJob job{B, reinterpret_cast<uint16_t*>(scratch), b_kmajor ? K : N, b_kmajor ? N : K, ldb, b_kmajor ? 1 : 0, b_kmajor ? K : N};
    if (int e = split_weight_planes(&job, 1, g.prec, s, g.sb)) return e;
    g.A = A; g.a_mode = X3_ROW; g.lda = lda;
    g.B = scratch; g.b_mode = X3_PLANES; g.ldb = K; g.b_plane = (long long)N * K; g.transposed_w = b_kmajor;
    return gemm_x3(g, s);
}

int64_t dc_workspace_layout(const dc_dims* dims, int64_t* offsets) { return dc::workspace_layout(dims, offsets); }

int dc_policy_forward(const dc_dims* dims, const float* params, const int64_t* poff_host, const float* obs,
                      const float* h0, const float* c0, const int64_t* seq_off, const int32_t* seq_len, void* ws,
                      float* hT, float* cT, const uint8_t* unit_mask, dc_stream_t stream) {
    DC_ENTER();
    return dc::policy_forward(dims, params, poff_host, obs, h0, c0, seq_off, seq_len, ws, hT, cT, unit_mask, (hipStream_t)stream);
}

static float* ws_f(const dc_dims* dims, const void* ws, int idx) {
    int64_t off[DC_WS_FIXED + DC_WS_PER_LAYER * DC_MAX_LAYERS];
    dc::workspace_layout(dims, off);
    return reinterpret_cast<float*>((char*)ws + off[idx]);
}

int dc_chunk_initial_state(const dc_dims* dims, const void* ws, const int64_t* prev_row, int n_chunks, float* h0, float* c0,
                           dc_stream_t stream) {
    DC_ENTER();
    if (dims->layers < 1 || dims->layers > DC_MAX_LAYERS) { dc::set_error("chunk_initial_state: layers out of range", 1020); return 1020; }
    for (int l = 0; l < dims->layers; ++l) {
        const int b = DC_WS_FIXED + l * DC_WS_PER_LAYER;
        if (int e = dc::rnn_gather_state(ws_f(dims, ws, b + DC_WSL_HSEQ), prev_row, h0 + (size_t)l * n_chunks * dims->hidden, n_chunks,
                                         dims->hidden, (hipStream_t)stream, dc::policy_bf16_store(dims) ? 1 : 0))
            return e;
        if (dims->cell == 1 && c0 != nullptr)
            if (int e = dc::rnn_gather_state(ws_f(dims, ws, b + DC_WSL_CSEQ), prev_row, c0 + (size_t)l * n_chunks * dims->hidden, n_chunks,
                                             dims->hidden, (hipStream_t)stream))
                return e;
    }
    return 0;
}

int dc_policy_single(const dc_dims* dims, const float* params, const int64_t* poff_host, const float* obs, const float* h0, const float* c0,
                     float* out, float* hT, float* cT, float* scratch, dc_stream_t stream) {
    DC_ENTER();
    return dc::policy_single(dims, params, poff_host, obs, h0, c0, out, hT, cT, scratch, (hipStream_t)stream);
}

int dc_select_logp(const dc_dims* dims, const void* ws, const uint8_t* act, const uint8_t* mask, float* logp_sel,
                   float* values, int32_t* argmax, dc_stream_t stream) {
    DC_ENTER();
    if (dims->rows <= 0) return 0;
    if (dims->flags & DC_DIMS_LAZY_TU)
        if (int e = dc::attn_logits_masked(ws_f(dims, ws, DC_WS_HEADOUT), ws_f(dims, ws, DC_WS_EMB), mask, ws_f(dims, ws, DC_WS_TU),
                                           dims->rows, dc::emb_rows_of(dims), (hipStream_t)stream))
            return e;
    return dc::select_logp(ws_f(dims, ws, DC_WS_HEADOUT), ws_f(dims, ws, DC_WS_TU), act, mask, logp_sel, values, argmax,
                           dims->rows, (hipStream_t)stream);
}

int dc_ppo_loss_fwd_bwd(const dc_dims* dims, void* ws, const uint8_t* act, const uint8_t* mask, const float* old_logp,
                        const float* adv, const float* ret, float* losses_out, int32_t* head_on, float e_clip,
                        float entropy_coef, float vf_coef, dc_stream_t stream) {
    DC_ENTER();
    if (dims->rows <= 0) { dc::set_error("ppo_loss: empty batch", 1030); return 1030; }
    if (dims->flags & DC_DIMS_LAZY_TU)
        if (int e = dc::attn_logits_masked(ws_f(dims, ws, DC_WS_HEADOUT), ws_f(dims, ws, DC_WS_EMB), mask, ws_f(dims, ws, DC_WS_TU),
                                           dims->rows, dc::emb_rows_of(dims), (hipStream_t)stream))
            return e;
    return dc::ppo_loss_fwd_bwd(ws_f(dims, ws, DC_WS_HEADOUT), ws_f(dims, ws, DC_WS_TU), act, mask, old_logp, adv, ret,
                                reinterpret_cast<double*>(ws_f(dims, ws, DC_WS_STATS)), ws_f(dims, ws, DC_WS_DHEADOUT),
                                ws_f(dims, ws, DC_WS_DTU), losses_out, head_on, dims->rows, e_clip, entropy_coef, vf_coef,
                                (hipStream_t)stream);
}

int dc_policy_backward(const dc_dims* dims, const float* params, const int64_t* poff_host, float* grads,
                       int64_t total_floats, const float* obs, const int64_t* seq_off, const int32_t* seq_len, void* ws,
                       dc_stream_t stream) {
    DC_ENTER();
    return dc::policy_backward(dims, params, poff_host, grads, total_floats, obs, seq_off, seq_len, ws, (hipStream_t)stream);
}

int dc_gradnorm_clip_adam(const int64_t* seg_off, const int32_t* seg_len, const int32_t* seg_gate, int n_seg,
                          int max_seg_len, float* params, float* grads, float* m, float* v, double* segsq,
                          const int32_t* head_on, const float* losses, float* norms_out, float* ctl,
                          int32_t* seg_step, int32_t* status, float max_norm, float vf_coef, double lr, double beta1,
                          double beta2, float eps, dc_stream_t stream) {
    DC_ENTER();
    return dc::gradnorm_clip_adam(seg_off, seg_len, seg_gate, n_seg, max_seg_len, params, grads, m, v, segsq, head_on,
                                  losses, norms_out, ctl, seg_step, status, max_norm, vf_coef, lr, beta1, beta2, eps,
                                  (hipStream_t)stream);
}

int dc_dp_average_grads(const int64_t* seg_off, const int32_t* seg_len, const int32_t* seg_gate, int n_seg,
                        int max_seg_len, float* grads, const float* counts, float vf_coef, dc_stream_t stream) {
    DC_ENTER();
    return dc::dp_average_grads(seg_off, seg_len, seg_gate, n_seg, max_seg_len, grads, counts, vf_coef, (hipStream_t)stream);
}

}  // extern "C"
