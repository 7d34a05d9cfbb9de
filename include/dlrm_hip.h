/*
 * dlrm_hip.h — C ABI of libdlrm_hip.so: the MI355X (gfx950 / CDNA4) implementation of the
 * DLRM forward/backward hot path.
 *
 * The reference (facebookresearch/dlrm) has no FFI of its own: its hot path is a chain of
 * PyTorch operator calls inside `DLRM_Net` (dlrm_s_pytorch.py:207-612).  Each entry point
 * below replaces one of those operator call sites; the citation names the call site.
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions (every function):
 *   - extern "C", returns int: 0 = ok, <0 = argument error (DLRM_E_*), >0 = hipError_t.
 *   - all matrices are fp32 row-major with an explicit leading dimension in ELEMENTS.
 *   - `*_dev` / plain `float*` arguments are DEVICE pointers borrowed for the call.
 *   - `*_host` arguments are HOST arrays (of device pointers or sizes) read during the call.
 *   - `stream` is a hipStream_t (NULL = default stream).  Nothing allocates, nothing
 *     synchronises; work is enqueued on `stream` and the call returns.
 *   - no process-wide mutable state: every mode (MLP arithmetic, update mode, error block) is a
 *     per-call argument, so two models / threads / streams in one process never interact.
 *   - `err` (embedding entry points): NULL, or a DEVICE-VISIBLE int64[4] block (device memory, or pinned
 *     host memory the caller can poll without synchronising).  An embedding index outside [0, rows_t)
 *     is never dereferenced: the lookup is skipped (contributes zero / updates nothing) and, when
 *     err != NULL, reported as err = {1, table, index, rows_t}.  The reference raises on such an
 *     index (torch's EmbeddingBag bounds check); the Python host raises from the block.
 */
#ifndef DLRM_HIP_H
#define DLRM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLRM_E_ARG     (-1)  /* null pointer / non-positive size                    */
#define DLRM_E_ALIGN   (-2)  /* pointer or leading dimension violates alignment     */
#define DLRM_E_RANGE   (-3)  /* size exceeds a compiled limit (see message on stderr)*/
#define DLRM_E_MODE    (-4)  /* unknown mode / activation / index width             */

/* activation codes (dlrm_s_pytorch.py:238-241: ReLU everywhere, Sigmoid at sigmoid_layer) */
#define DLRM_ACT_NONE    0
#define DLRM_ACT_RELU    1
#define DLRM_ACT_SIGMOID 2

/* embedding-update modes */
#define DLRM_UPD_ATOMIC        0  /* fast: LDS pre-reduction for tiny tables + HW fp32 atomics */
#define DLRM_UPD_DETERMINISTIC 1  /* exact: per-row in-input-order FMA chain (bit-exact vs torch sparse SGD) */
#define DLRM_UPD_SORTED        2  /* fast: lookups radix-sorted by (table,row), plain read-modify-write per row run;
                                     needs workspace (dlrm_emb_bwd_workspace_bytes); atomics only where a run straddles chunks */

/* library / device introspection ------------------------------------------------------- */
int         dlrm_hip_abi_version(void);          /* bumps when a signature changes */
const char* dlrm_hip_build_info(void);           /* "gfx950 <date> <compiler>"     */
int         dlrm_hip_device_info(int device, int* cu_count, int* lds_bytes,
                                 int64_t* hbm_bytes, char* name, int name_len);

/* In-run calibration of the box (bench.py `box`; not on the training path).  The caller times the launch with HIP events on `stream`.
 *   dlrm_calib_mfma:     kind 0 = v_mfma_f32_32x32x2_f32, kind 1 = v_mfma_f32_32x32x16_bf16, back to back on every SIMD, no memory traffic, CONSTANT
 *                        operands; kinds 2 / 3 = the same instructions on eight rotating sets of pseudo-random operands (a GEMM's bit toggling);
 *                        *flop_out = FLOPs the launch executes (so rate = flop / time; implied clock = rate / (CUs * 256 [* 16] FLOP/clk)).
 *                        scratch: device float[1].  iters = 15000 is ~50 ms of fp32, ~3 ms of bf16 work on MI355X.
 *   dlrm_calib_hbm_copy: float4 copy (one float4 per thread) of `bytes` (multiple of 16) from src to dst; HBM rate = 2 * bytes / time. */
int dlrm_calib_mfma(int kind, int iters, float* scratch, double* flop_out, void* stream);
int dlrm_calib_hbm_copy(const void* src, void* dst, int64_t bytes, void* stream);
/* 512-byte rows at pseudo-random places of table[0 .. table_bytes), one per half-wave, eight in flight: the embedding kernels' access pattern
 * (bench.py's `box.hbm_gather_gbps`; zero-fill the buffer first).  *bytes_out = bytes read by the launch. */
int dlrm_calib_hbm_gather(const void* table, int64_t table_bytes, int64_t rows_to_read, uint32_t seed, float* scratch, double* bytes_out, void* stream);
/* CU-partitioned streams for composition experiments (tools/probes/cu_mask_probe.py; not used by the training path): a stream restricted to
 * CUs [first, first + count) of the current device (hipExtStreamCreateWithCUMask).  The caller destroys it. */
int dlrm_stream_create_cu_range(int first, int count, void** stream_out);
int dlrm_stream_destroy(void* stream);

/* ---------------------------------------------------------------------------------------
 * K1  EmbeddingBag(mode="sum") forward for ALL tables in one launch.
 * Replaces: the per-table `nn.EmbeddingBag.__call__` loop in DLRM_Net.apply_emb
 *           (dlrm_s_pytorch.py:407-462; operator created at :277).
 *   out[b*out_ld + t*D + d] = sum_{i in bag(t,b)} psw_t[i] * W_t[idx_t[i]*D + d]
 *   bag(t,b) = [off_t[b], off_t[b+1])  with off_t[B] := nnz[t]   (no trailing offset needed)
 * Summation is the in-order sequential fp32 sum from +0.0 per output column (FMA chain when
 * weighted) — the torch CPU kernel's order, so results are bit-identical.
 *   weight_host[t]  : device float*  [rows_host[t], D]
 *   indices_host[t] : device int32/int64* [nnz_host[t]]   (idx_bits = 32 | 64)
 *   offsets_host[t] : device int32/int64* [B]
 *   psw_host        : NULL, or array[T] whose entries are NULL or device float* [nnz_host[t]]
 *   out             : device float*, row b at out + b*out_ld, table t at column t*D
 *                     (out_ld = T*D gives the torch.cat(ly,1) layout; out_ld = (T+1)*D with
 *                      out = feat + D writes straight into the [B, 1+T, D] interaction buffer)
 * Empty bags produce zeros.  Out-of-range indices are skipped and reported through `err` (see conventions).
 */
int dlrm_emb_fwd(int T, int64_t B, int D,
                 const void* const* weight_host, const int64_t* rows_host,
                 const void* const* indices_host, const void* const* offsets_host,
                 const int64_t* nnz_host, const void* const* psw_host, int idx_bits,
                 float* out, int64_t out_ld, int64_t* err, void* stream);

/* K2+K3  fused EmbeddingBag backward + sparse SGD step, all tables, no gradient materialised.
 * Replaces: autograd `EmbeddingBagBackward` (sparse COO grad) followed by
 *           `torch.optim.SGD.step` on that grad (dlrm_s_pytorch.py:1613,1620; optimizer :1343-1369).
 *   for every lookup i of bag (t,b), in input order:  W_t[idx_t[i],:] = fma(-lr*psw, dout[b, t*D:(t+1)*D], W_t[idx_t[i],:])
 * mode = DLRM_UPD_ATOMIC: order of duplicate-row accumulation is unspecified (fp32 atomics).
 * mode = DLRM_UPD_DETERMINISTIC: bit-exact with the reference (one owner per row, input order).
 * mode = DLRM_UPD_SORTED: rows whose lookups fall into one chunk of the sorted list (all rows of large
 *        tables in practice) are bit-exact; hot rows are re-associated.  workspace: device scratch of
 *        dlrm_emb_bwd_workspace_bytes(...) bytes (may be NULL for the other modes).
 */
int64_t dlrm_emb_bwd_workspace_bytes(int T, const int64_t* nnz_host, const int64_t* rows_host);
/* Which sorter the sort-based updates (DLRM_UPD_SORTED, dlrm_emb_bwd_rowwise_adagrad) will use for these table shapes:
 *   1 = the segmented per-table radix sort of csrc/seg_sort.h for every launch group: plain kernels only, replayable inside a HIP graph;
 *   0 = at least one group (a table segment of more than 262144 lookups, or >= 2^39 rows) goes to the general sorter (rocPRIM),
 *       which must not be captured into a HIP graph on ROCm 7.2 (tools/probes/graph_sorted_probe.py). */
int     dlrm_emb_sort_kind(int T, const int64_t* nnz_host, const int64_t* rows_host);
/* The sort itself (one launch group, T <= 32): positions_out[j] = global lookup position (table-major) of the j-th entry in
 * (table, row) order, equal rows in input order; keys_out[j] = table << *row_bits_out | row; bag_out[p] (nullable) = bag of position p
 * (0xFFFFFFFF: skipped out-of-range lookup).  workspace as dlrm_emb_bwd_workspace_bytes.  For tests and tools. */
int     dlrm_emb_sort_lookups(int T, int64_t B, const int64_t* rows_host, const void* const* indices_host,
                              const void* const* offsets_host, const int64_t* nnz_host, int idx_bits, void* workspace,
                              int64_t workspace_bytes, uint32_t* positions_out, uint64_t* keys_out, uint32_t* bag_out,
                              int* row_bits_out, int64_t* err, void* stream);
int dlrm_emb_bwd_sgd(int T, int64_t B, int D,
                     void* const* weight_host, const int64_t* rows_host,
                     const void* const* indices_host, const void* const* offsets_host,
                     const int64_t* nnz_host, const void* const* psw_host, int idx_bits,
                     const float* dout, int64_t dout_ld, float lr, const float* lr_dev, int mode,
                     void* workspace, int64_t workspace_bytes, int64_t* err, void* stream);

/* Pooling weights (--weighted-pooling fixed | learned; dlrm_s_pytorch.py:289-293, 370-375, 425-428).
 * dlrm_pool_weights_gather: psw_out_host[t][i] = vw_host[t][indices_host[t][i]]  (`v_W_l[k].gather(0, indices)`); the result is
 *   the psw_host operand of dlrm_emb_fwd / dlrm_emb_bwd_*.
 * dlrm_emb_psw_grad (learned weights): dvw_host[t][r] = sum over lookups i of table t with index r of
 *   <dout[bag(i), t*D:(t+1)*D], W_t[r, :]> — EmbeddingBag's per_sample_weights gradient followed by the gather's scatter-add,
 *   one pass; dvw_host[t] (device float[rows_host[t]]) is overwritten.  Call it BEFORE the embedding update of the step. */
int dlrm_pool_weights_gather(int T, const int64_t* rows_host, const void* const* indices_host, const int64_t* nnz_host,
                             int idx_bits, const void* const* vw_host, void* const* psw_out_host, int64_t* err, void* stream);
int dlrm_emb_psw_grad(int T, int64_t B, int D, const void* const* weight_host, const int64_t* rows_host,
                      const void* const* indices_host, const void* const* offsets_host, const int64_t* nnz_host,
                      int idx_bits, const float* dout, int64_t dout_ld, void* const* dvw_host, void* stream);

/* K2 alone: the reference's sparse COO gradient, materialised (escape hatch: `--fused-emb-update=0`, or any
 * optimizer the fused kernels do not implement, e.g. torch.optim.Adagrad / the reference's --optimizer=adagrad).
 * Replaces: autograd EmbeddingBagBackward (dlrm_s_pytorch.py:1613).
 *   values_host[t] : device float* [nnz_host[t], D];  values_t[i,:] = psw_t[i] * dout[bag(i), t*D:(t+1)*D]
 * The COO indices are indices_host[t] verbatim (uncoalesced, input order) — the caller wraps both into
 * torch.sparse_coo_tensor and hands it to the optimizer as `.grad`. */
int dlrm_emb_bwd_coo(int T, int64_t B, int D, const void* const* offsets_host, const int64_t* nnz_host,
                     const void* const* psw_host, int idx_bits, const float* dout, int64_t dout_ld,
                     void* const* values_host, void* stream);

/* K4  fused EmbeddingBag backward + row-wise sparse Adagrad (optim/rwsadagrad.py:117-143).
 *   per table, per UNIQUE row r touched this step (duplicates summed first, in input order):
 *     g_r      = sum_i psw_i * dout[bag(i), :]
 *     state[r] += mean_d(g_r[d]^2)
 *     W[r,:]  -= lr * g_r / (sqrt(state[r]) + eps)
 * workspace: device scratch of at least dlrm_emb_adagrad_workspace_bytes(...) bytes.
 */
int64_t dlrm_emb_adagrad_workspace_bytes(int T, int D, const int64_t* nnz_host, const int64_t* rows_host);
int dlrm_emb_bwd_rowwise_adagrad(int T, int64_t B, int D,
                     void* const* weight_host, void* const* state_host, const int64_t* rows_host,
                     const void* const* indices_host, const void* const* offsets_host,
                     const int64_t* nnz_host, const void* const* psw_host, int idx_bits,
                     const float* dout, int64_t dout_ld, float lr, const float* lr_dev, float eps,
                     void* workspace, int64_t workspace_bytes, int64_t* err, void* stream);

/* ---------------------------------------------------------------------------------------
 * K6  dot interaction forward.
 * Replaces: torch.cat + torch.bmm + Z[:, li, lj] + torch.cat in DLRM_Net.interact_features
 *           (dlrm_s_pytorch.py:483-504).
 *   feature f of sample b is the D-vector at feat_host[f] + b*feat_ld_host[f]   (f = 0 is the
 *   bottom-MLP output x, f = 1..F-1 the pooled embeddings, in table order)
 *   R[b, 0:D]            = feature 0
 *   R[b, D + p(i,j)]     = <feature i, feature j>,  j < i (j <= i if self_interaction),
 *                          p enumerates pairs row-major: (1,0),(2,0),(2,1),(3,0)...  (:499-501)
 *   columns [D+P, ldr) of R are zero-filled (ldr may pad the row for alignment).
 * self_interaction is a mode: 0 = strictly lower triangle in the reference's order, 1 = with the diagonal
 * (--arch-interaction-itself), 2 = the strictly upper triangle in torch.triu_indices(F, F, 1) order, i.e. pairs
 * (0,1),(0,2)...(0,F-1),(1,2)... — what torchrec's InteractionArch emits (BASELINE.json configs[4]; same dot products,
 * permuted columns).  The same mode goes to dlrm_interact_bwd.
 */
int dlrm_interact_fwd(int64_t B, int F, int D,
                      const void* const* feat_host, const int64_t* feat_ld_host,
                      int self_interaction, float* R, int64_t ldr, void* stream);

/* K6 backward:  dfeat_i = sum_j (dZ[i,j] + dZ[j,i]) * feat_j  (+ dR[:,0:D] for i = 0).
 * dfeat_host/dfeat_ld_host address the gradient rows exactly like feat_host addresses inputs.
 * Backward only (both dlrm_interact_bwd and dlrm_interact_bwd_gather): self_interaction | DLRM_INTERACT_RELU_X says that feature 0 is
 * the output of a ReLU (the bottom tower's last layer, dlrm_s_pytorch.py:238-241) and asks for dfeat_0 * [feature 0 > 0] — the kernel
 * has x staged anyway, so the tower's first backward pass (dlrm_act_bwd over [B, D]) disappears; same values, bit for bit. */
#define DLRM_INTERACT_RELU_X 4
int dlrm_interact_bwd(int64_t B, int F, int D,
                      const void* const* feat_host, const int64_t* feat_ld_host,
                      int self_interaction, const float* dR, int64_t ldr,
                      void* const* dfeat_host, const int64_t* dfeat_ld_host, void* stream);

/* K1 + K6 fused for one-lookup-per-bag inputs (the Criteo data sets: offsets = arange, dlrm_data_pytorch.py:334-337):
 * features whose index_host[f] != NULL are NOT [B, D] matrices but EmbeddingBag tables — feature f of sample b is row
 * index_host[f][b] of the table at feat_host[f] (feat_ld_host[f] = D), fetched by the interaction kernel itself.  The
 * pooled-embedding buffer that dlrm_emb_fwd would write and dlrm_interact_* read back (2 x T x B x D x 4 bytes: 1.7 GB of the
 * forward's 2.4 GB at Criteo-Terabyte shapes) never exists; results are bit-identical to dlrm_emb_fwd + dlrm_interact_fwd.
 *   offsets_host[f] : the table's bag starts; the kernels verify offsets[b] == b (one lookup per bag) — a violation, or an
 *                     index outside [0, rows_host[f]), is reported through `err` ({1, table, index, rows}; rows = -1 marks a
 *                     bag-layout violation) and row 0 is read instead.
 * Available when dlrm_interact_gather_ok(F, D) (D = 128, F <= 27, tables of < 2^32 rows, 16-byte aligned operands); otherwise DLRM_E_MODE and the
 * caller runs the two kernels.  The backward writes dfeat rows exactly like dlrm_interact_bwd (the gradient of a gathered
 * feature is the dout operand of dlrm_emb_bwd_*). */
int dlrm_interact_gather_ok(int F, int D);
/* The proof the caller owes before taking that path: offsets_host[t][b] == b for all T tables and B bags ("one lookup per bag";
 * nnz == B alone does not prove it — EmbeddingBag accepts an empty bag next to a two-lookup bag, dlrm_s_pytorch.py:453-457).  Adds the
 * number of bags whose start differs from their number to *violations (DEVICE-VISIBLE int32, zeroed by the caller; pinned host memory
 * lets the caller read it after synchronising `stream`). */
int dlrm_offsets_are_iota(int T, int64_t B, const void* const* offsets_host, int idx_bits, int32_t* violations, void* stream);
int dlrm_interact_fwd_gather(int64_t B, int F, int D, const void* const* feat_host, const int64_t* feat_ld_host,
                             const void* const* index_host, const void* const* offsets_host, const int64_t* rows_host,
                             int idx_bits, int self_interaction, float* R, int64_t ldr, int64_t* err, void* stream);
int dlrm_interact_bwd_gather(int64_t B, int F, int D, const void* const* feat_host, const int64_t* feat_ld_host,
                             const void* const* index_host, const void* const* offsets_host, const int64_t* rows_host,
                             int idx_bits, int self_interaction, const float* dR, int64_t ldr,
                             void* const* dfeat_host, const int64_t* dfeat_ld_host, int64_t* err, void* stream);

/* ABI 16 — the choice between the fused path and the two kernels taken ON THE DEVICE.  Until round 6 the caller proved "one lookup per bag"
 * for every offsets tensor nobody vouched for by a device pass the HOST waited for (dlrm_offsets_are_iota into pinned memory), which ended the
 * host's run-ahead once per training step: 30-110 us per step on most boxes of the pool, 1.25 ms on a bad day (profiles/round6/proof_wait.md).
 * Now the caller enqueues BOTH implementations with a launch predicate and never waits:
 *   dlrm_offsets_iota_flags   as dlrm_offsets_are_iota, into *flag_dev (device int32, zeroed by the caller) and, when the count is not zero,
 *                             1 into *flag_host (pinned, zeroed by the caller; nullable) — the host may read that LATER, to remember the verdict
 *                             of a tensor object it will see again;
 *   dlrm_*_pred               the call without the suffix, whose workgroups return at once unless (*pred_flag != 0) == (pred_nonzero != 0)
 *                             (pred_flag == NULL: always run).  dlrm_interact_fwd_pred / _bwd_pred are the gather form when index_host != NULL
 *                             and the plain form otherwise; only the D = 128 LDS-DMA kernels take a predicate (else DLRM_E_MODE).
 * A step therefore runs  fused(pred_nonzero = 0)  +  dlrm_emb_fwd_pred + dlrm_interact_fwd_pred(plain)(pred_nonzero = 1)  — three launches that
 * return at once for a one-lookup-per-bag batch, the fused one for a ragged batch — and its results are those of the implementation that ran. */
int dlrm_offsets_iota_flags(int T, int64_t B, const void* const* offsets_host, int idx_bits, int32_t* flag_dev, int32_t* flag_host, void* stream);
int dlrm_emb_fwd_pred(int T, int64_t B, int D, const void* const* weight_host, const int64_t* rows_host,
                      const void* const* indices_host, const void* const* offsets_host, const int64_t* nnz_host,
                      const void* const* psw_host, int idx_bits, float* out, int64_t out_ld, int64_t* err,
                      const int32_t* pred_flag, int pred_nonzero, void* stream);
int dlrm_interact_fwd_pred(int64_t B, int F, int D, const void* const* feat_host, const int64_t* feat_ld_host,
                           const void* const* index_host, const void* const* offsets_host, const int64_t* rows_host,
                           int idx_bits, int self_interaction, float* R, int64_t ldr, int64_t* err,
                           const int32_t* pred_flag, int pred_nonzero, void* stream);
int dlrm_interact_bwd_pred(int64_t B, int F, int D, const void* const* feat_host, const int64_t* feat_ld_host,
                           const void* const* index_host, const void* const* offsets_host, const int64_t* rows_host,
                           int idx_bits, int self_interaction, const float* dR, int64_t ldr,
                           void* const* dfeat_host, const int64_t* dfeat_ld_host, int64_t* err,
                           const int32_t* pred_flag, int pred_nonzero, void* stream);

/* ABI 17 — the sparse SGD step of the rows ONE lookup of the batch touches, taken INSIDE the fused backward (BASELINE north_star: "fused
 * sparse SGD ... for the backward embedding update"; replaces EmbeddingBagBackward + torch.optim.SGD.step, dlrm_s_pytorch.py:1613,1620, for
 * those rows).  The fused backward (dlrm_interact_bwd_gather) has every gathered table row staged in LDS and computes its gradient row: for a
 * row that no other lookup of the batch names, W[idx] = fma(-lr, dfeat, W[idx]) can be written to the table at once — the gradient row is
 * neither written nor read back, the table row not read a second time (per such lookup 1 row of traffic instead of 4; 30 % of the
 * Criteo-Terabyte batch's lookups, nearly all lookups of its 8 large tables).  Rows named by several lookups keep the sorted update.  Results
 * are those of dlrm_emb_bwd_sgd(DLRM_UPD_SORTED), bit for bit (a single lookup's update is the same one fma per element).  Three calls:
 *   dlrm_emb_presort            the first half of dlrm_emb_bwd_sgd(DLRM_UPD_SORTED): lookups expanded and sorted by (table, row) into `workspace`
 *                               (dlrm_emb_bwd_workspace_bytes), plus single_mask[b] (device uint32[B]): bit t set = lookup (t, b) is in range and the
 *                               only lookup of the batch that names its row.  One launch group (T <= 32), nnz_host[t] == B for every table.
 *   dlrm_interact_bwd_gather_sgd  dlrm_interact_bwd_pred (gather form; feature 0 plain, features 1..F-1 = tables 0..F-2 in presort order, rows of
 *                               D floats) that applies the step for every set bit and writes no dfeat row for it.  The caller must not touch
 *                               the tables between the presort and this call.  Runs under the launch predicate like dlrm_interact_bwd_pred.
 *   dlrm_emb_bwd_sgd_presorted  the second half from that workspace: the sorted update with dout = the dfeat rows of the backward; with
 *                               skip_singles != 0 the single lookups are skipped WHEN the predicate holds ((*pred_flag != 0) == (pred_nonzero != 0),
 *                               or pred_flag == NULL) — the same predicate dlrm_interact_bwd_gather_sgd ran under; when it does not hold (a ragged
 *                               batch: the two-kernel backward wrote every dfeat row) all lookups are applied.  D = 128 (else DLRM_E_MODE).
 * The step size of both update calls must be the same (float lr or device scalar lr_dev, see LEARNING RATES). */
int dlrm_emb_presort(int T, int64_t B, const int64_t* rows_host, const void* const* indices_host, const void* const* offsets_host,
                     const int64_t* nnz_host, int idx_bits, void* workspace, int64_t workspace_bytes, uint32_t* single_mask,
                     int64_t* err, void* stream);
int dlrm_interact_bwd_gather_sgd(int64_t B, int F, int D, const void* const* feat_host, const int64_t* feat_ld_host,
                                 const void* const* index_host, const void* const* offsets_host, const int64_t* rows_host,
                                 int idx_bits, int self_interaction, const float* dR, int64_t ldr, void* const* dfeat_host,
                                 const int64_t* dfeat_ld_host, const uint32_t* single_mask, float lr, const float* lr_dev, int64_t* err,
                                 const int32_t* pred_flag, int pred_nonzero, void* stream);
int dlrm_emb_bwd_sgd_presorted(int T, int64_t B, int D, void* const* weight_host, const int64_t* rows_host, const int64_t* nnz_host,
                               const float* dout, int64_t dout_ld, float lr, const float* lr_dev, const void* workspace,
                               int64_t workspace_bytes, int skip_singles, const int32_t* pred_flag, int pred_nonzero, void* stream);

/* ---------------------------------------------------------------------------------------
 * K5  MLP layer = nn.Linear + activation  (dlrm_s_pytorch.py:216,238-241,405), fp32 MFMA.
 *   Y[M,N] = act(X[M,K] · W[N,K]^T + bias[N])
 */
/* Arithmetic of the three MLP GEMMs — the `arith` argument of every dlrm_linear_* call (per call: no global switch):
 *   DLRM_ARITH_F32    v_mfma_f32_32x32x2_f32: every product and sum in fp32 (157 TFLOP/s matrix peak).
 *   DLRM_ARITH_BF16X6 fp32 operands split EXACTLY into three bf16 terms inside the kernel (x = h + m + l), the six
 *                     products of order >= 2^-16 issued on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; dropped
 *                     terms total <= 2^-21 |a*b| in the worst case and 2^-24 |a*b| rms — the rms of ONE fp32 rounding (oracle/oracle.py product_bf16x6,
 *                     tests/test_oracle_golden.py).  Inputs, outputs and accumulators stay fp32.
 * Only operands that meet the fast-path preconditions (16-byte alignment, k % 16 == 0) use BF16X6; others run F32. */
#define DLRM_ARITH_F32    0
#define DLRM_ARITH_BF16X6 1
#define DLRM_ARITH_BF16   2   /* operands rounded to bf16 (nearest even) in the kernel, one bf16 MFMA per 16-k step, fp32 accumulate:
                                 the "bf16 MLP" of BASELINE.json configs[4]; NOT an fp32-class result (about 3 decimal digits per operand) */

/* relu_bits (nullable; act must be DLRM_ACT_RELU): device buffer of dlrm_relu_bits_bytes(M, N) bytes that receives one SIGN BIT
 * per output element (Y > 0) — the ReLU derivative the backward pass needs, 32x smaller than re-reading Y.  Layout: the
 * matrix is cut into 32-row x 64-column blocks, block (mb, nb) owns the 64 consecutive uint32 words starting at
 * (mb * ceil(N/64) + nb) * 64; bit 31 - (4*it + c) of word l is element (row 32*mb + 4*it + l/16, column 64*nb + 4*(l%16) + c),
 * it < 8, c < 4 (the lane geometry of the GEMM epilogues: lane l of a wave owns word l, shifts its 32 signs in with one
 * compare + add-with-carry each, and the block moves as ONE coalesced 4-byte access per lane). */
/* bf16-STORAGE tower (BASELINE.json configs[4] "bf16 MLP on MFMA"; torchrec_dlrm/dlrm_main.py:302,526 runs its MLPs reduced-precision):
 * activations and weights are kept as bf16 copies in HBM and v_mfma_f32_32x32x16_bf16 reads them without any conversion in the loop.
 *   dlrm_cast_bf16            dst[m, n] = bf16(src[m, n]) (nearest even), zero for N <= n < Npad        (inputs, weights)
 *   dlrm_cast_bf16_transposed dstT[c, r] = bf16(src[r, c]), zero for R <= r < Rpad                        (W^T for the data gradient)
 *   dlrm_gemm_bf16            C (fp32, nullable) and/or Cb (bf16, nullable) [M, N] = epilogue(A[M, K] . B[N, K]^T), fp32 accumulation:
 *                             forward  A = X,  B = W   : + bias, activation, ReLU sign bits OUT (relu_bits_out, nullable)
 *                             dgrad    A = dY, B = W^T : masked by the previous layer's ReLU sign bits (relu_bits_in, nullable)
 *                             needs K % 32 == 0, N % 4 == 0, lda/ldb % 8 == 0, 16-byte aligned rows; DLRM_E_ALIGN otherwise.
 * Results equal DLRM_ARITH_BF16 of dlrm_linear_fwd / _bwd_data bit for bit (same operand rounding, same accumulation order); the
 * weight gradient keeps reading the fp32 copies through dlrm_linear_bwd_weight(arith = DLRM_ARITH_BF16). */
int dlrm_cast_bf16(int64_t M, int N, int Npad, const float* src, int64_t lds, uint16_t* dst, int64_t ldd, void* stream);
int dlrm_cast_bf16_transposed(int R, int C, int Rpad, const float* src, int64_t lds, uint16_t* dstT, int64_t ldd, void* stream);
/* both copies of up to DLRM_CAST_MULTI_MAX tensors in ONE launch (the weights of a tower: W16 for the forward GEMMs, W^T16 for the data
 * gradients): tensor i = src[i] [R, C] -> dst[i] [R, Cpad] (nullable) and / or dstT[i] [C, Rpad] (nullable), padding zero-filled */
#define DLRM_CAST_MULTI_MAX 16
int dlrm_cast_bf16_multi(int n, const float* const* src, const int64_t* lds, const int* R, const int* C, uint16_t* const* dst, const int64_t* ldd,
                         const int* Cpad, uint16_t* const* dstT, const int64_t* lddT, const int* Rpad, void* stream);
int dlrm_gemm_bf16(int64_t M, int N, int K, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const float* bias, int act,
                   uint64_t* relu_bits_out, const uint64_t* relu_bits_in, const float* addend, int64_t ldadd, const float* addend2, int64_t ldadd2,
                   float* C, int64_t ldc, uint16_t* Cb, int64_t ldcb, void* stream);
/* (addend: nullable fp32 [M, N] added to the product AFTER bias / activation / mask — C = A.B^T + addend, the DCN-v2 cross layer's
 *  gradient sum g + dv.V without a separate add kernel; may alias C for an in-place accumulation; addend2: a second one (needs addend);
 *  only where the bf16-shaped kernel runs, DLRM_E_MODE otherwise) */
/* Weight gradient of a bf16 layer from bf16 operands as stored (no fp32 activation / gradient copy is read): dW[N, K] (+)= dZ[M, N]^T . X[M, K],
 * dbias[N] (+)= column sums of dZ (nullable).  Replaces AddmmBackward's weight / bias branch (dlrm_s_pytorch.py:1613) in the arithmetic of
 * dlrm_gemm_bf16.  Both operands are read k-strided through ds_read_b64_tr_b16 (csrc/gemm_bf16.hip), the batch is split into fp32 slabs in
 * `workspace` (dlrm_linear_bwd_weight_bf16_workspace_bytes) summed in a fixed order: deterministic.  DLRM_E_ALIGN when M % 64, N % 8, K % 8,
 * N < 64, K < 64 or unaligned rows: use dlrm_linear_bwd_weight on fp32 operands.  dW is [N, K_store], K_store <= K: the trailing K - K_store
 * columns of X are alignment padding (zeros) whose gradient is dropped (the 479 -> 480 interaction width). */
int64_t dlrm_linear_bwd_weight_bf16_workspace_bytes(int64_t M, int N, int K);
int dlrm_linear_bwd_weight_bf16(int64_t M, int N, int K, int K_store, const uint16_t* dZ, int64_t lddz, const uint16_t* X, int64_t ldx,
                                float* dW, int64_t lddw, float* dbias, int accumulate, void* workspace, int64_t workspace_bytes,
                                void* stream);

/* ---- arith "bf16x6" from PRE-SPLIT operands.  An fp32 tensor is held as THREE bf16 planes h, m, l (truncation split: x == h + m + l exactly,
 * each plane an [rows, ld] bf16 matrix, the planes `plane` ELEMENTS apart inside one allocation); a product is the six bf16 MFMAs per 16 k of
 * DLRM_ARITH_BF16X6 (h.h, h.m, m.h, m.m, h.l, l.h; fp32 accumulation) — the same values, in the same order, as dlrm_linear_fwd / _bwd_data
 * compute when they split fp32 operands inside their k-loops: results are BIT-IDENTICAL to those, i.e. the fp32 round-off class of the
 * reference's addmm (dlrm_s_pytorch.py:399-405, AddmmBackward :1613).  What changes is the cost: operands are split once by their producer
 * (dlrm_split_bf16x3, or the Cp output of the previous GEMM) instead of by every k-loop that reads them, and the k-loop is the bf16-shaped
 * four-phase pipeline of csrc/gemm_bf16.hip with no VALU work in it.
 *   dlrm_split_bf16x3            [M, N] fp32 -> planes of [M, Npad] (zero columns N..Npad-1; Npad, ldd, plane % 8 == 0)
 *   dlrm_split_bf16x3_transposed [R, C] fp32 -> planes of its transpose [C, Rpad] (W^T for the data gradient)
 *   dlrm_gemm_bf16x6             C (fp32, nullable) / Cp (planes of the result, nullable) = epilogue(A . B^T): bias, activation, ReLU sign bits
 *                                out / mask in as dlrm_gemm_bf16.  Preconditions (dlrm_gemm_bf16x6_supported; DLRM_E_ALIGN otherwise — there is
 *                                no other kernel that reads planes, the caller keeps fp32 storage): K % 16 == 0, N % 4 == 0, N >= 192, M >= 256,
 *                                lda, ldb % 8 == 0, plane strides % 8 == 0 and below 2^30 elements.
 *   dlrm_linear_bwd_weight_bf16x6  dW (+)= dZ^T . X, dbias (+)= column sums of dZ from the planes of dZ [M, N] and X [M, K], read k-strided;
 *                                preconditions and workspace of dlrm_linear_bwd_weight_bf16. */
int dlrm_split_bf16x3(int64_t M, int N, int Npad, const float* src, int64_t lds, uint16_t* dst, int64_t ldd, int64_t plane, void* stream);
int dlrm_split_bf16x3_transposed(int R, int C, int Rpad, const float* src, int64_t lds, uint16_t* dstT, int64_t ldd, int64_t plane, void* stream);
int dlrm_gemm_bf16x6_supported(int64_t M, int N, int K, int64_t lda, int64_t ldb);
int dlrm_gemm_bf16x6(int64_t M, int N, int K, const uint16_t* A, int64_t lda, int64_t planeA, const uint16_t* B, int64_t ldb, int64_t planeB,
                     const float* bias, int act, uint64_t* relu_bits_out, const uint64_t* relu_bits_in, float* C, int64_t ldc,
                     uint16_t* Cp, int64_t ldcp, int64_t planeC, void* stream);
int dlrm_linear_bwd_weight_bf16x6(int64_t M, int N, int K, int K_store, const uint16_t* dZ, int64_t lddz, int64_t planeZ, const uint16_t* X,
                                  int64_t ldx, int64_t planeX, float* dW, int64_t lddw, float* dbias, int accumulate, void* workspace,
                                  int64_t workspace_bytes, void* stream);

int64_t dlrm_relu_bits_bytes(int64_t M, int N);
int dlrm_linear_fwd(int64_t M, int N, int K,
                    const float* X, int64_t ldx, const float* W, int64_t ldw,
                    const float* bias, int act, float* Y, int64_t ldy, uint64_t* relu_bits, int arith, void* stream);

/* data gradient with the PREVIOUS layer's activation backward fused into the epilogue:
 *   dX[M,K] = (dY[M,N] · W[N,K]) ⊙ act'(Xact[M,K])        (Xact = this layer's input = previous
 *                                                           layer's activated output; xact_kind
 *                                                           ACT_NONE -> no mask, Xact may be NULL)
 * relu_bits (nullable, xact_kind == DLRM_ACT_RELU): the sign bits of Xact [M, K] that dlrm_linear_fwd of the previous layer
 * wrote; the epilogue then reads 256 B per 32 x 64 block instead of 8 KB of fp32 activations (Xact must still be passed:
 * shapes outside the fast path fall back to it).
 */
int dlrm_linear_bwd_data(int64_t M, int N, int K,
                         const float* dY, int64_t lddy, const float* W, int64_t ldw,
                         const float* Xact, int64_t ldxa, int xact_kind, const uint64_t* relu_bits,
                         float* dX, int64_t lddx, int arith, void* stream);

/* weight AND bias gradient:  dW[N,K] (+)= dY[M,N]^T · X[M,K],  dbias[N] (+)= column sums of dY
 * (reduction over the batch, split over workgroups; the bias gradient is taken from the dY fragments
 * the MFMAs consume, so it costs no extra pass over dY).  accumulate != 0 adds into dW/dbias, else
 * they are overwritten.  dbias may be NULL.
 * workspace: device scratch of dlrm_linear_bwd_weight_workspace_bytes(M,N,K) bytes (16-byte aligned) for the
 * split-K partial slabs, summed in a fixed order by a second kernel (deterministic dW).  NULL / too small:
 * the k-slices accumulate into dW with fp32 atomics instead (correct, slower, order-dependent rounding). */
int64_t dlrm_linear_bwd_weight_workspace_bytes(int64_t M, int N, int K);
int dlrm_linear_bwd_weight(int64_t M, int N, int K,
                           const float* dY, int64_t lddy, const float* X, int64_t ldx,
                           float* dW, int64_t lddw, float* dbias, int accumulate,
                           void* workspace, int64_t workspace_bytes, int arith, void* stream);

/* The same weight gradient when X carries alignment padding: X is [M, K] with columns K_store..K-1 zero (e.g. 13 dense
 * features padded to 16, 479 interaction outputs padded to 480), dW is the true [N, K_store] gradient.  Needs the
 * workspace of dlrm_linear_bwd_weight_workspace_bytes(M, N, K). */
int dlrm_linear_bwd_weight_padded(int64_t M, int N, int K, int K_store,
                           const float* dY, int64_t lddy, const float* X, int64_t ldx,
                           float* dW, int64_t lddw, float* dbias, int accumulate,
                           void* workspace, int64_t workspace_bytes, int arith, void* stream);
/* The whole backward of an N == 1 layer (the 256 -> 1 head of the top tower) in one pass over X:
 *   dz = dY * act'(Y) (Y == NULL or act == NONE: dY is dz), dW[K] (+)= dz^T X, dbias[1] (+)= sum dz, dX = (dz W) * xact'(X) (dX == NULL: none).
 * Bit-identical to dlrm_act_bwd + dlrm_linear_bwd_weight + dlrm_linear_bwd_data with N = 1 (same row partition, order and expressions);
 * workspace of dlrm_linear_bwd_weight_workspace_bytes(M, 1, K).  DLRM_E_MODE for shapes outside the fast path (K % 4, K > 1024, unaligned
 * rows): the caller then makes the three calls. */
int dlrm_linear_head_bwd(int64_t M, int K, const float* dY, int64_t lddy, const float* Y, int64_t ldy, int act,
                         const float* X, int64_t ldx, const float* W, int xact_kind, float* dX, int64_t lddx,
                         float* dW, float* dbias, int accumulate, void* workspace, int64_t workspace_bytes, void* stream);
/* dst[m, 0:K] = src[m, 0:K], dst[m, K:Kp] = 0   (builds those padded operands: no ATen fill/copy on the hot path) */
int dlrm_pad_cols(int64_t M, int K, int Kp, const float* src, int64_t ld_src, float* dst, int64_t ld_dst, void* stream);

/* ---------------------------------------------------------------------------------------
 * K5 for SMALL batches: a whole tower (nn.Sequential(Linear, act, ...), dlrm_s_pytorch.py:208-246, 399-405; native fp32 MFMA) in
 * four launches — forward, data-gradient chain, all weight / bias gradients (+ their slab sums) — instead of one GEMM launch per layer and direction.
 * A workgroup owns 16 batch rows whose activations stay in LDS between the layers; every workgroup streams all weights, so this is
 * for batches of a few thousand rows (Criteo-Kaggle: 2048), where the per-layer GEMMs are launch- and latency-bound.  csrc/tower.hip.
 *   widths[0..L]      widths[0] = the tower's input width, widths[l + 1] = outputs of layer l (all <= DLRM_TOWER_MAX_WIDTH)
 *   acts[l]           DLRM_ACT_* of layer l
 *   W_host[l]         [widths[l + 1], widths[l]] at pitch ldw_host[l]; bias_host[l] nullable
 *   Y_host[l]         [M, widths[l + 1]]: every layer's activated output is WRITTEN by dlrm_tower_fwd (the backward pass reads them)
 * dlrm_tower_bwd: dY [M, widths[L]] = gradient of the tower's output (last_act_applied != 0: already multiplied by the last
 *   activation's derivative — DLRM_INTERACT_RELU_X); writes dZ_host[l] [M, widths[l + 1]] = dL/dz of every layer and, when dX != NULL,
 *   the gradient of the tower's input.
 * dlrm_tower_wgrad: dW_host[l] [widths[l + 1], kstore[l]] = dZ_l^T . In_l (In_0 = the tower's input, In_l = Y_{l-1}; kstore[l] <=
 *   widths[l]: trailing alignment-padding columns of the input are dropped), db_host[l] (nullable) = column sums of dZ_l; both
 *   OVERWRITTEN.  64 x 64 tiles x batch slices into slabs (`workspace`, dlrm_tower_wgrad_workspace_bytes), then a second launch adds
 *   every tile's slabs in slice order (deterministic).
 * Unaligned operands (13 dense features: rows of 52 bytes) take element loads; nothing needs padding. */
#define DLRM_TOWER_MAX_LAYERS 8
#define DLRM_TOWER_MAX_WIDTH 512
int dlrm_tower_fwd(int64_t M, int L, const int* widths, const int* acts, const float* X, int64_t ldx, const void* const* W_host,
                   const int64_t* ldw_host, const void* const* bias_host, void* const* Y_host, const int64_t* ldy_host, void* stream);
int dlrm_tower_bwd(int64_t M, int L, const int* widths, const int* acts, const float* dY, int64_t lddy, int last_act_applied,
                   const void* const* W_host, const int64_t* ldw_host, const void* const* Y_host, const int64_t* ldy_host,
                   void* const* dZ_host, const int64_t* lddz_host, float* dX, int64_t lddx, void* stream);
int64_t dlrm_tower_wgrad_workspace_bytes(int64_t M, int L, const int* widths);
int dlrm_tower_wgrad(int64_t M, int L, const int* widths, const int* kstore, const void* const* dZ_host, const int64_t* lddz_host,
                     const void* const* In_host, const int64_t* ldin_host, void* const* dW_host, const int64_t* lddw_host,
                     void* const* db_host, void* workspace, int64_t workspace_bytes, void* stream);

/* activation backward for a layer whose dY does not come out of dlrm_linear_bwd_data (i.e. the last
 * layer of a tower):  dZ = dY ⊙ act'(Y);  optionally dbias[N] += column sums of dZ (dbias != NULL; the
 * caller zeroes it — normally NULL because dlrm_linear_bwd_weight produces the bias gradient). */
int dlrm_act_bwd(int64_t M, int N, const float* dY, int64_t lddy, const float* Y, int64_t ldy,
                 int act, float* dZ, int64_t lddz, float* dbias, void* stream);

/* ---------------------------------------------------------------------------------------
 * K7  BCELoss(reduction="mean") forward + backward (dlrm_s_pytorch.py:386-393,148-156).
 *   loss = -mean_b w_b*(t_b*max(log p_b,-100) + (1-t_b)*max(log(1-p_b),-100))
 *   dp_b = w_b*(p_b - t_b) / max(p_b*(1-p_b), 1e-12) / B * grad_scale       (torch's formulas)
 *   w_b  = weights[b] (1 if weights == NULL) * (t_b >= 1 ? w_pos : w_neg)
 * w_neg / w_pos are the two class weights of --loss-function=wbce (`loss_ws[T.long()]`, loss_fn_wrap :150-156: the
 * whole weighted mean in one pass); pass 1, 1 for plain BCE.  loss_out: device float[1].  dp may be NULL (forward only).
 * partials: device scratch, >= dlrm_loss_workspace_bytes(B) bytes.
 */
int64_t dlrm_loss_workspace_bytes(int64_t B);
int dlrm_bce_loss(int64_t B, const float* p, const float* target, const float* weights, float w_neg, float w_pos,
                  float grad_scale, float* loss_out, float* dp, void* partials, void* stream);
/* BCEWithLogitsLoss(reduction="mean") on raw logits — the loss of torchrec's DLRMTrain (BASELINE.json configs[4]; the over-arch
 * ends without a sigmoid, torchrec_dlrm/dlrm_main.py:650): loss = mean((1-t)*x + m + log(exp(-m) + exp(-x-m))), m = max(-x, 0)
 * (torch's formula), dlogits = (sigmoid(x) - t) / B * grad_scale. */
int dlrm_bce_logits_loss(int64_t B, const float* logits, const float* target, float grad_scale,
                         float* loss_out, float* dlogits, void* partials, void* stream);
/* MSELoss(mean): loss = mean((p-t)^2), dp = 2(p-t)/B*grad_scale */
int dlrm_mse_loss(int64_t B, const float* p, const float* target,
                  float grad_scale, float* loss_out, float* dp, void* partials, void* stream);

/* loss backward glue: y[i] = x[i] * scalar_dev[0] (dL/dp scaled by the upstream gradient of the scalar loss, which
 * autograd hands over as a device scalar — read on the device, no host synchronisation) */
int dlrm_scale_by_device_scalar(int64_t n, const float* x, const float* scalar_dev, float* y, void* stream);

/* LEARNING RATES (ABI 15).  Every update entry point takes `float lr` AND `const float* lr_dev`: lr_dev == NULL -> the step size is the by-value
 * lr, baked into the launch; lr_dev != NULL -> the kernel reads it from that device float WHEN IT RUNS and `lr` is ignored.  The second form is
 * for captured HIP graphs: the reference steps its LRPolicyScheduler every iteration (dlrm_s_pytorch.py:169-203, :1621), and a graph whose
 * update kernels read a device scalar follows the schedule without re-capture (dlrm_set_f32 / dlrm_graph_replay write the scalar).  Both forms
 * multiply with the same fp32 value: bit-identical results. */
/* dense SGD step over a flat parameter buffer: w -= lr * g   (torch.optim.SGD, no momentum) */
int dlrm_sgd_dense(int64_t n, float* w, const float* g, float lr, const float* lr_dev, void* stream);
/* dst_host[i][0] = values_host[i], i < n <= 16: the values travel in the kernarg (no host buffer outlives the call) */
int dlrm_set_f32(int n, float* const* dst_host, const float* values_host, void* stream);
/* the same step for `count` parameter tensors in ONE launch (w_host/g_host: host arrays of device pointers,
 * n_host: element counts) — the 16 weight/bias tensors of the two MLP towers (dlrm_s_pytorch.py:1620). */
int dlrm_sgd_dense_multi(int count, float* const* w_host, const float* const* g_host, const int64_t* n_host,
                         float lr, const float* lr_dev, void* stream);
/* dense Adagrad step, RWSAdagrad's dense branch (optim/rwsadagrad.py:145-148):
 *   sum += g*g;  w -= lr * g / (sqrt(sum) + eps)        (lr = the decayed clr of :115) */
int dlrm_adagrad_dense(int64_t n, float* w, float* sum, const float* g, float lr, const float* lr_dev, float eps, void* stream);

/* ---------------------------------------------------------------------------------------
 * Evaluation metrics of the inference pass, on the device (dlrm_s_pytorch.py:759-899: numpy accuracy :819-821,
 * scikit-learn recall / precision / f1 / accuracy / roc_auc / average_precision :828-847).
 *   scores, targets : device float[n]  (targets are 0/1 labels; label = target > 0.5)
 *   out             : device double[9] = { n, positives, TP, FP, FN, TN (predicted label = rint(score), i.e.
 *                     np.round), roc_auc (trapezoids over distinct thresholds == sklearn.metrics.roc_auc_score),
 *                     average_precision (== sklearn.metrics.average_precision_score),
 *                     count of rint(score) == target (the reference's A_test) }
 *                     roc_auc / average_precision are NaN when a class is absent (sklearn raises there).
 *   workspace       : device scratch of dlrm_binary_metrics_workspace_bytes(n) bytes, 16-byte aligned.
 * Sums are integer or fp64 in a fixed order: deterministic. */
int64_t dlrm_binary_metrics_workspace_bytes(int64_t n);
int dlrm_binary_metrics(int64_t n, const float* scores, const float* targets, double* out,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Synthetic ("random") input batches generated in HBM.  Replaces generate_dist_input_batch (uniform) and
 * generate_random_output_batch of the reference (dlrm_data_pytorch.py:899-960, 835-846):
 *   table t, bag b:  L = P if fixed else round(max(1, r * min(rows_t, P)));  indices = unique(round(r_j * (rows_t-1))), j < L
 *   offsets[t][b]  = running sum of the post-unique bag lengths;  nnz_dev[t] = total lookups of table t (device int64)
 * Uniforms come from Philox4x32-10 keyed by `seed` with counter (bag, table, draw): same distributions as the
 * reference, different stream (numpy's MT19937 sequence is not reproduced).
 *   offsets_host[t] : device int32/int64* [B]        indices_host[t] : device int32/int64*, capacity >= B * P
 *   workspace       : dlrm_gen_workspace_bytes(T, B) bytes (not needed for the one-hot case P == 1 && fixed)
 * dlrm_gen_uniform_dense fills x[n] with float32(U[0,1)) (dense features), rounded to 0/1 when round_values != 0
 * (targets with --round-targets). */
int64_t dlrm_gen_workspace_bytes(int T, int64_t B);
int dlrm_gen_uniform_bags(int T, int64_t B, const int64_t* rows_host, int num_indices_per_lookup, int fixed,
                          uint64_t seed, int idx_bits, void* const* offsets_host, void* const* indices_host,
                          int64_t* nnz_dev, void* workspace, int64_t workspace_bytes, void* stream);
int dlrm_gen_uniform_dense(int64_t n, float* x, int round_values, uint64_t seed, void* stream);

/* Criteo-Terabyte binary records -> model inputs on the device.  Replaces CriteoBinDataset.__getitem__ +
 * _transform_features (data_loader_terabyte.py:233-248, 74-93).  raw: device int32 [B, 40] = label | 13 dense | 26 ids.
 *   X[b, 0:13] = logf(float(dense) + 1);  indices[t*ld_idx + b] = id_t mod max_ind_range (if > 0; result in [0, m));
 *   offsets[t*ld_idx + b] = b;  target[b] = float(label).   indices/offsets: int32 or int64 [26, ld_idx >= B]. */
int dlrm_criteo_bin_transform(int64_t B, const int32_t* raw, int64_t max_ind_range, int idx_bits, float* X,
                              int64_t ldx, void* indices, void* offsets, int64_t ld_idx, float* target, void* stream);

/* ---------------------------------------------------------------------------------------
 * MLPerf-v2 multi-hot synthetic inputs (BASELINE.json configs[4]).  Replaces the reference's `Multihot`
 * (torchrec_dlrm/multi_hot.py:80-159), which expands every batch on the host CPU.
 * dlrm_multihot_gen_table: lookup table M [rows, hot] int32 in HBM; column 0 = the row id, columns 1.. = uniform
 *   randint(0, rows) (dist 0) or int32(pareto(0.25)) % rows (dist 1) (:86-104) from Philox4x32-10 (counter (row, column,
 *   table_id), key seed): the reference's distributions, a different stream than numpy's seed-0 MT19937.
 * dlrm_multihot_expand (:129-159, :115-127): ids = device int32/int64 [T, B] 1-hot ids, key(table)-major as in the KJT
 *   `_values` the reference reshapes (:135);  values[vbase_t + b*hot_t + j] = M_t[ids[t, b], j] with vbase_t = B * sum_{k<t} hot_k
 *   (== torch.cat of the per-table F.embedding results);  offsets_global [T*B + 1] int64 = the reference's cumulative
 *   offsets (nullable);  offsets_local [T, B] int32 = b * hot_t, the per-table bag starts dlrm_emb_* take together with
 *   indices_host[t] = values + vbase_t (nullable).  An id outside [0, rows_t) is reported through `err` (the reference's
 *   F.embedding raises) and expanded as id 0. */
int dlrm_multihot_gen_table(int table_id, int64_t rows, int hot, int dist, uint64_t seed, int32_t* table_out, void* stream);
int dlrm_multihot_expand(int T, int64_t B, const void* ids, int idx_bits, const void* const* tables_host,
                         const int64_t* rows_host, const int* hot_host, int32_t* values, int64_t* offsets_global,
                         int32_t* offsets_local, int64_t* err, void* stream);

/* ---------------------------------------------------------------------------------------
 * One replay of a captured whole-step HIP graph from one host call (dlrm_amd.graph.GraphedTrainStep; the reference loop body
 * dlrm_s_pytorch.py:1574-1621 captured once): optionally wait for the stream (the previous replay), copy n input tensors
 * (device to device, bytes_host[i] each, skipped where source == destination) into the graph's static buffers, write n_scalars
 * device floats (dlrm_set_f32: the learning rates the captured update kernels read through lr_dev; 0 = none changed), hipGraphLaunch.
 * graph_exec is the hipGraphExec_t (torch.cuda.CUDAGraph.raw_cuda_graph_exec()).  The only entry point that synchronises, and
 * only when asked to. */
int dlrm_graph_replay(int n, void* const* dst_host, const void* const* src_host, const int64_t* bytes_host,
                      int n_scalars, float* const* scalar_dst_host, const float* scalar_values_host, void* graph_exec,
                      int sync_first, void* stream);

/* ---------------------------------------------------------------------------------------
 * Strided block copy = torch.cat / torch.split along dim 1 without ATen:
 *   for k < nblk:  dst_host[k][m*dst_ld_host[k] + c] = src_host[k][m*src_ld_host[k] + c],  m < M, c < width_host[k]
 * Uses: the "cat" interaction R = cat([x] + ly, 1) (dlrm_s_pytorch.py:505-507) when the features do not already sit
 * in one buffer (distributed mode: the all-to-all receive blocks, extend_distributed.py:446-465), and its backward
 * (the same call with the roles swapped).  The dot interaction never needs it: dlrm_interact_* read the receive
 * buffer in place through their {ptr, stride} tables. */
int dlrm_copy_blocks(int64_t M, int nblk, const void* const* src_host, const int64_t* src_ld_host,
                     void* const* dst_host, const int64_t* dst_ld_host, const int* width_host, void* stream);

/* BCELoss(reduction="none") forward / backward — the per-sample loss the reference's `wbce` path weights and averages
 * (dlrm_s_pytorch.py:388-391, loss_fn_wrap :150-156).  loss[i] = -(t*max(log p,-100) + (1-t)*max(log(1-p),-100));
 * dp[i] = dloss[i] * (p-t) / max(p*(1-p), 1e-12).  (dlrm_bce_loss with `weights` is the fully fused weighted mean.) */
int dlrm_bce_elementwise(int64_t n, const float* p, const float* target, float* loss, void* stream);
int dlrm_bce_elementwise_bwd(int64_t n, const float* p, const float* target, const float* dloss, float* dp, void* stream);

/* DCN-v2 interaction (MLPerf-v2's default, torchrec_dlrm/dlrm_main.py:608-619; torchrec LowRankCrossNet, not in the tree):
 *   x_{l+1} = x_0 * (W_l (V_l x_l) + b_l) + x_l  on the flattened [B, F*D] feature buffer.  The two products of a layer are
 *   dlrm_linear_fwd calls (act none; V without bias); these are the elementwise halves, contiguous fp32 arrays of n elements
 *   (n % 4 == 0, 16-byte aligned):  dlrm_cross_fwd  out = x0 * u + xl;   dlrm_cross_bwd  du = g * x0,  dx0 (+)= g * u
 *   (accumulate != 0 adds into dx0);  dlrm_add  out = a + b  (the gradient reaching x_l through the V product joins g). */
int dlrm_cross_fwd(int64_t n, const float* x0, const float* u, const float* xl, float* out, uint16_t* out16 /* nullable: bf16(out) */, void* stream);
int dlrm_cross_bwd(int64_t n, const float* g, const float* x0, const float* u /* fp32 u ... */, const uint16_t* u16 /* ... or its bf16 copy: exactly one */,
                   float* du /* nullable */, uint16_t* du16 /* nullable: bf16(g * x0) */, float* dx0, int accumulate, void* stream);
/* The second product of a cross layer WITH its elementwise half in the GEMM epilogue (bf16-storage towers; replaces dlrm_gemm_bf16 + dlrm_cross_fwd,
 * whose fp32 u made a round trip through HBM):  u = A . B^T + bias (A = bf16 v_l [M, K], B = bf16 W_l [N, K], fp32 accumulation);
 * Ub (bf16 [M, N], nullable) = u — what dlrm_cross_bwd(u16) reads;  C (fp32) = fma(x0, u, xl) = x_{l+1};  Cb (bf16, nullable) = bf16(C), the next
 * layer's GEMM operand.  Same operation order as the two kernels (bit-identical x_{l+1}).  Preconditions of the bf16-shaped kernel (K % 64 == 0,
 * N >= 192, M >= 256, 16-byte aligned rows); DLRM_E_MODE otherwise — the caller keeps the two kernels. */
int dlrm_gemm_bf16_cross(int64_t M, int N, int K, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const float* bias,
                         const float* x0, int64_t ldx0, const float* xl, int64_t ldxl, uint16_t* Ub, int64_t ldub,
                         float* C, int64_t ldc, uint16_t* Cb, int64_t ldcb, void* stream);
int dlrm_add(int64_t n, const float* a, const float* b, float* out, void* stream);

/* torch.clamp(p, lo, hi) of the predictions and its backward (--loss-threshold, dlrm_s_pytorch.py:580-583, 607-610):
 * y = min(max(x, lo), hi);  dx = dy where lo <= x <= hi, else 0. */
int dlrm_clamp(int64_t n, const float* x, float lo, float hi, float* y, void* stream);
int dlrm_clamp_bwd(int64_t n, const float* x, float lo, float hi, const float* dy, float* dx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DLRM_HIP_H */
