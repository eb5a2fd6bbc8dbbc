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
 *     immediately; nothing is allocated, freed or synchronised inside;
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

int dc_abi_version(void);
const char* dc_last_error(void);

/* Replaces optimizer.py:53-64 `discount` + `advantage_returns`, fused with the sub-reward sum of
 * optimizer.py:397 and the terminal-zero append of optimizer.py:417-420.
 *   rewards [rows,10] f32, values [rows] f32, seq_off [n_seq] i64, seq_len [n_seq] i32 (device),
 *   max_len = max(seq_len) (host), gamma/lam as the reference's python doubles (0.98, 0.97),
 *   adv/ret [rows] f32 out. */
int dc_gae_scan(const float* rewards, const float* values, const int64_t* seq_off, const int32_t* seq_len,
                int n_seq, int max_len, double gamma, double lam, float* adv, float* ret, dc_stream_t stream);

/* fp32 MFMA GEMM building block (every nn.Linear of policy.py:54-75 and its autograd products).
 *   C[M,N] (op)= A[M,K] * B[K,N] (+ bias[N]) ; a_kmajor: A stored [K][lda] else [M][lda];
 *   b_kmajor: B stored [K][ldb] else [N][ldb]; relu: max(0,.); aux/ldaux: zero where aux<=0;
 *   accumulate: C += ; splits: 0 = auto split-K. */
int dc_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                int a_kmajor, int b_kmajor, const float* bias, int relu, const float* aux, int ldaux,
                int accumulate, int splits, dc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
