// extern "C" surface of libdotaclient_hip.so - see include/dotaclient_hip.h for the contract.
#include "../../include/dotaclient_hip.h"
#include "common.h"
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

int gae_scan(const float*, const float*, const int64_t*, const int32_t*, int, int, double, double, float*, float*,
             hipStream_t);
int gemm_f32(const float*, const float*, float*, int, int, int, int, int, int, int, int, const float*, int,
             const float*, int, int, int, hipStream_t);

}  // namespace dc

extern "C" {

int dc_abi_version(void) { return 1; }
const char* dc_last_error(void) { return dc::g_err; }

int dc_gae_scan(const float* rewards, const float* values, const int64_t* seq_off, const int32_t* seq_len,
                int n_seq, int max_len, double gamma, double lam, float* adv, float* ret, dc_stream_t stream) {
    return dc::gae_scan(rewards, values, seq_off, seq_len, n_seq, max_len, gamma, lam, adv, ret, (hipStream_t)stream);
}

int dc_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                int a_kmajor, int b_kmajor, const float* bias, int relu, const float* aux, int ldaux,
                int accumulate, int splits, dc_stream_t stream) {
    return dc::gemm_f32(A, B, C, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, bias, relu, aux, ldaux, accumulate,
                        splits, (hipStream_t)stream);
}

}  // extern "C"
