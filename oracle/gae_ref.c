/* ORACLE (test infrastructure only - never linked into the product library).
 *
 * Plain-C restatement of /root/reference/optimizer.py:53-64 (discount, advantage_returns) with the
 * sub-reward sum of optimizer.py:397 and the terminal-zero append of optimizer.py:417-420.
 * The reference delegates the recurrence to scipy==1.2.0 `lfilter([1],[1,-g], x[::-1])[::-1]`
 * (docker/Dockerfile:18): a direct-form IIR evaluated in float64, y[t] = x[t] + g*y[t+1].
 * Parity status: PINNED by tests/test_oracle.py (tests/golden/gae_kat.npz from the real reference).
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC oracle/gae_ref.c -o oracle/_build/libgae_ref.so
 */
#include <stdint.h>

/* numpy float32 pairwise sum for n = 10: 8 lanes, tree combine, 2 leftovers sequential */
static float reward_sum10(const float* r) {
    float a = (r[0] + r[1]) + (r[2] + r[3]);
    float b = (r[4] + r[5]) + (r[6] + r[7]);
    float s = a + b;
    s = s + r[8];
    s = s + r[9];
    return s;
}

/* rewards [rows,10], values [rows]; sequence i occupies rows [off[i], off[i]+len[i]) */
void gae_ref(const float* rewards, const float* values, const int64_t* off, const int32_t* len, int n_seq,
             double gamma, double lam, float* adv, float* ret) {
    const float gf = (float)gamma;
    const double gl = gamma * lam;
    for (int s = 0; s < n_seq; ++s) {
        const int64_t b = off[s];
        const int L = len[s];
        double ya = 0.0, yr = 0.0;
        for (int t = L - 1; t >= 0; --t) {
            const float r = reward_sum10(rewards + (b + t) * 10);
            const float v1 = (t + 1 < L) ? values[b + t + 1] : 0.0f;
            const float gv = gf * v1;
            const float d = (r + gv) - values[b + t];
            const double pa = gl * ya, pr = gamma * yr;
            ya = (double)d + pa;
            yr = (double)r + pr;
            adv[b + t] = (float)ya;
            ret[b + t] = (float)yr;
        }
    }
}
