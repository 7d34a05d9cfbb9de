/*
 * dlrm_oracle.c — CPU restatement of the reference DLRM hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (dlrm_amd/) never does.  Plain scalar C, one thread; every function states which lines of
 * /root/reference it follows.  Arithmetic contracts (measured against torch 2.10 CPU kernels in the
 * survey, SURVEY.md §8c, and pinned by tests/golden/ fixtures generated from the live reference):
 *   - EmbeddingBag(sum): in-order sequential fp32 accumulation from +0.0, FMA chain when weighted;
 *   - sparse SGD on the uncoalesced COO gradient: per lookup, in input order, W = fma(-lr, g, W);
 *   - Linear / bmm: no fixed order in MKL -> accumulated here in double, compared with tolerance.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

/* ---- nn.EmbeddingBag(mode="sum") forward, one table ------------------------------------------
 * follows DLRM_Net.apply_emb (dlrm_s_pytorch.py:407-462): V[b,:] = sum_{i in bag b} w_i * E[idx_i,:],
 * bag b = [off[b], off[b+1]) with the last bag running to nnz; empty bag -> zeros.               */
API void oracle_emb_fwd(const float* E, int64_t rows, int D, const int64_t* idx, int64_t nnz,
                        const int64_t* off, int64_t B, const float* psw, float* out, int64_t out_ld) {
    (void)rows;
    for (int64_t b = 0; b < B; ++b) {
        const int64_t s = off[b], e = (b + 1 < B) ? off[b + 1] : nnz;
        float* o = out + b * out_ld;
        for (int d = 0; d < D; ++d) o[d] = 0.0f;
        for (int64_t i = s; i < e; ++i) {
            const float* r = E + idx[i] * (int64_t)D;
            if (psw) { const float w = psw[i]; for (int d = 0; d < D; ++d) o[d] = fmaf(w, r[d], o[d]); }
            else     { for (int d = 0; d < D; ++d) o[d] = o[d] + r[d]; }
        }
    }
}

/* ---- EmbeddingBag backward + torch.optim.SGD.step on the sparse gradient, one table ------------
 * follows autograd's embedding_bag sparse backward (COO indices = idx verbatim, values =
 * dV[bag(i)] * w_i) and `p.add_(grad, alpha=-lr)` (dlrm_s_pytorch.py:1613,1620; optimizer :1343).  */
API void oracle_emb_bwd_sgd(float* E, int64_t rows, int D, const int64_t* idx, int64_t nnz,
                            const int64_t* off, int64_t B, const float* psw, const float* dV,
                            int64_t dV_ld, float lr) {
    (void)rows;
    for (int64_t b = 0; b < B; ++b) {
        const int64_t s = off[b], e = (b + 1 < B) ? off[b + 1] : nnz;
        const float* g = dV + b * dV_ld;
        for (int64_t i = s; i < e; ++i) {
            float* r = E + idx[i] * (int64_t)D;
            for (int d = 0; d < D; ++d) {
                const float val = psw ? g[d] * psw[i] : g[d];
                r[d] = fmaf(-lr, val, r[d]);
            }
        }
    }
}

/* ---- row-wise sparse Adagrad, one table (optim/rwsadagrad.py:73-152, sparse branch :117-143) ----
 * coalesce (duplicates summed in input order), mom[r] += mean_d(g^2), W[r] -= clr*g/(sqrt(mom)+eps) */
typedef struct { int64_t row; int64_t pos; } rp_t;
static int rp_cmp(const void* a, const void* b) {
    const rp_t* x = (const rp_t*)a; const rp_t* y = (const rp_t*)b;
    if (x->row != y->row) return x->row < y->row ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}
API void oracle_emb_bwd_rowwise_adagrad(float* E, float* mom, int64_t rows, int D, const int64_t* idx,
                                        int64_t nnz, const int64_t* off, int64_t B, const float* psw,
                                        const float* dV, int64_t dV_ld, float clr, float eps) {
    (void)rows;
    if (nnz == 0) return;
    rp_t* rp = (rp_t*)malloc(sizeof(rp_t) * (size_t)nnz);
    int64_t* bag = (int64_t*)malloc(sizeof(int64_t) * (size_t)nnz);
    for (int64_t b = 0; b < B; ++b) {
        const int64_t s = off[b], e = (b + 1 < B) ? off[b + 1] : nnz;
        for (int64_t i = s; i < e; ++i) { rp[i].row = idx[i]; rp[i].pos = i; bag[i] = b; }
    }
    qsort(rp, (size_t)nnz, sizeof(rp_t), rp_cmp);
    float* g = (float*)malloc(sizeof(float) * (size_t)D);
    int64_t k = 0;
    while (k < nnz) {
        const int64_t r = rp[k].row;
        for (int d = 0; d < D; ++d) g[d] = 0.0f;
        int first = 1;
        while (k < nnz && rp[k].row == r) {
            const int64_t i = rp[k].pos;
            const float* gv = dV + bag[i] * dV_ld;
            for (int d = 0; d < D; ++d) {
                const float v = psw ? gv[d] * psw[i] : gv[d];
                g[d] = first ? v : g[d] + v;
            }
            first = 0; ++k;
        }
        float sq = 0.0f;
        for (int d = 0; d < D; ++d) sq += g[d] * g[d];
        mom[r] += sq / (float)D;
        const float denom = sqrtf(mom[r]) + eps;
        float* w = E + r * (int64_t)D;
        for (int d = 0; d < D; ++d) w[d] = fmaf(-clr, g[d] / denom, w[d]);
    }
    free(g); free(bag); free(rp);
}

/* ---- dot interaction (dlrm_s_pytorch.py:483-504) ------------------------------------------------
 * T[b] = [x; e_1; ...; e_{F-1}] (F x D), Z = T T^T, R[b] = [x | Z[i][j] for i in 0..F-1, j < i (+i)] */
API void oracle_interact_fwd(const float* feat, int64_t B, int F, int D, int self_interaction, float* R,
                             int64_t ldr) {
    for (int64_t b = 0; b < B; ++b) {
        const float* Tb = feat + b * (int64_t)F * D;
        float* r = R + b * ldr;
        for (int d = 0; d < D; ++d) r[d] = Tb[d];
        int p = D;
        for (int i = 0; i < F; ++i) {
            const int jmax = self_interaction ? i + 1 : i;
            for (int j = 0; j < jmax; ++j) {
                double acc = 0.0;
                for (int d = 0; d < D; ++d) acc += (double)Tb[i * D + d] * (double)Tb[j * D + d];
                r[p++] = (float)acc;
            }
        }
    }
}

API void oracle_interact_bwd(const float* feat, int64_t B, int F, int D, int self_interaction,
                             const float* dR, int64_t ldr, float* dfeat) {
    for (int64_t b = 0; b < B; ++b) {
        const float* Tb = feat + b * (int64_t)F * D;
        const float* g = dR + b * ldr;
        float* dT = dfeat + b * (int64_t)F * D;
        double* acc = (double*)calloc((size_t)F * D, sizeof(double));
        for (int d = 0; d < D; ++d) acc[d] = g[d];
        int p = D;
        for (int i = 0; i < F; ++i) {
            const int jmax = self_interaction ? i + 1 : i;
            for (int j = 0; j < jmax; ++j) {
                const double gz = g[p++];
                for (int d = 0; d < D; ++d) {
                    acc[i * D + d] += gz * Tb[j * D + d];
                    acc[j * D + d] += gz * Tb[i * D + d];
                }
            }
        }
        for (int k = 0; k < F * D; ++k) dT[k] = (float)acc[k];
        free(acc);
    }
}

/* ---- nn.Linear + ReLU / Sigmoid (dlrm_s_pytorch.py:216,238-241) ---------------------------------- */
static float act_f(float v, int act) { return act == 1 ? (v > 0.f ? v : 0.f) : act == 2 ? 1.f / (1.f + expf(-v)) : v; }
API void oracle_linear_fwd(const float* X, int64_t M, int K, const float* W, const float* bias, int N,
                           int act, float* Y) {
    for (int64_t m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double acc = bias ? bias[n] : 0.0;
            for (int k = 0; k < K; ++k) acc += (double)X[m * K + k] * (double)W[(int64_t)n * K + k];
            Y[m * N + n] = act_f((float)acc, act);
        }
}
/* given dY (grad wrt the ACTIVATED output Y): dZ = dY*act'(Y); dW = dZ^T X; db = colsum dZ; dX = dZ W */
API void oracle_linear_bwd(const float* X, int64_t M, int K, const float* W, int N, int act, const float* Y,
                           const float* dY, float* dX, float* dW, float* db) {
    double* dZ = (double*)malloc(sizeof(double) * (size_t)(M * N));
    for (int64_t m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            const float y = Y[m * N + n], g = dY[m * N + n];
            dZ[m * N + n] = act == 1 ? (y > 0.f ? g : 0.0) : act == 2 ? (double)g * ((1.0 - y) * y) : g;
        }
    for (int n = 0; n < N; ++n) {
        double s = 0.0;
        for (int64_t m = 0; m < M; ++m) s += dZ[m * N + n];
        db[n] = (float)s;
        for (int k = 0; k < K; ++k) {
            double a = 0.0;
            for (int64_t m = 0; m < M; ++m) a += dZ[m * N + n] * (double)X[m * K + k];
            dW[(int64_t)n * K + k] = (float)a;
        }
    }
    if (dX)
        for (int64_t m = 0; m < M; ++m)
            for (int k = 0; k < K; ++k) {
                double a = 0.0;
                for (int n = 0; n < N; ++n) a += dZ[m * N + n] * (double)W[(int64_t)n * K + k];
                dX[m * K + k] = (float)a;
            }
    free(dZ);
}

/* ---- BCELoss(mean) (dlrm_s_pytorch.py:386-393): torch clamps both logs at -100; backward divides by
 * max((1-p)*p, 1e-12) (aten binary_cross_entropy_backward).                                          */
API float oracle_bce(const float* p, const float* t, int64_t B, float* dp) {
    double s = 0.0;
    for (int64_t i = 0; i < B; ++i) {
        float lp = logf(p[i]); if (lp < -100.f) lp = -100.f;
        float l1 = log1pf(-p[i]); if (l1 < -100.f) l1 = -100.f;
        s += -(double)(t[i] * lp + (1.f - t[i]) * l1);
        if (dp) {
            float den = (1.f - p[i]) * p[i]; if (den < 1e-12f) den = 1e-12f;
            dp[i] = (p[i] - t[i]) / den / (float)B;
        }
    }
    return (float)(s / (double)B);
}
API float oracle_mse(const float* p, const float* t, int64_t B, float* dp) {
    double s = 0.0;
    for (int64_t i = 0; i < B; ++i) {
        const float d = p[i] - t[i];
        s += (double)d * d;
        if (dp) dp[i] = 2.f * d / (float)B;
    }
    return (float)(s / (double)B);
}
