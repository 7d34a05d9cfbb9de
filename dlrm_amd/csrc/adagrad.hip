// adagrad.hip — fused EmbeddingBag backward + row-wise sparse Adagrad (optim/rwsadagrad.py:117-143).
// Placeholder translation unit: the entry points exist so the C ABI is complete, but they refuse
// loudly until the kernels land (no silent fallback).
#include "common.h"

extern "C" int64_t dlrm_emb_adagrad_workspace_bytes(int T, const int64_t* nnz_host, const int64_t* rows_host) {
    (void)T; (void)nnz_host; (void)rows_host;
    return 0;
}

extern "C" int dlrm_emb_bwd_rowwise_adagrad(int T, int64_t B, int D, void* const* weight_host,
                                            void* const* state_host, const int64_t* rows_host,
                                            const void* const* indices_host, const void* const* offsets_host,
                                            const int64_t* nnz_host, const void* const* psw_host, int idx_bits,
                                            const float* dout, int64_t dout_ld, float lr, float eps,
                                            void* workspace, int64_t workspace_bytes, void* stream) {
    fprintf(stderr, "libdlrm_hip: dlrm_emb_bwd_rowwise_adagrad: not implemented in this build\n");
    return DLRM_E_MODE;
}
