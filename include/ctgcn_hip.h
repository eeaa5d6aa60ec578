/*
 * ctgcn_hip.h — C ABI of libctgcn_hip.so, the MI355X (gfx950) hot path of CTGCN.
 *
 * The reference (jhljx/CTGCN) is pure Python and has no FFI layer; its "operator API" for this
 * path is four call sites.  Each entry point below names the reference call site it replaces
 * (paths relative to the reference root).  A reference-side ctypes binding is shown in
 * INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - every function returns 0 on success or a negative CTGCN_E_* code, never throws, never
 *     allocates device memory (callers pass workspaces sized by ctgcn_workspace_bytes);
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); work is enqueued on it
 *     and the call returns without synchronising, except ctgcn_kcore_i32 which must read its
 *     termination counter back and therefore synchronises `stream` internally;
 *   - ctgcn_last_error() returns a thread-local, NUL-terminated description of the last failure;
 *   - CSR arrays: row_ptr int32[n_rows+1] (nnz < 2^31), col_idx int32[nnz], val float[nnz]; the per-entry
 *     arrays may be NULL when nnz == 0;
 *   - dense matrices are row-major fp32 with an explicit leading dimension (elements).
 */
#ifndef CTGCN_HIP_H
#define CTGCN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTGCN_ABI_VERSION 28

enum {
    CTGCN_OK = 0,
    CTGCN_E_INVALID = -1,   /* bad argument (null pointer, negative size, K out of range ...) */
    CTGCN_E_HIP = -2,       /* a HIP runtime call failed; see ctgcn_last_error()              */
    CTGCN_E_WORKSPACE = -3, /* workspace too small                                            */
    CTGCN_E_UNSUPPORTED = -4
};

/* flags of ctgcn_core_aggregate_f32 / ctgcn_core_aggregate_bwd_f32 */
enum {
    CTGCN_F_SELF_LOOP = 1,  /* slot 0 is (A + I): add X[row] once (helper.py:71-72)                     */
    CTGCN_F_RELU = 2,       /* apply ReLU to every emitted slot (layers.py:48)                          */
    CTGCN_F_NESTED = 4      /* edges tagged s belong to slots s..K-1 (nested k-cores).  Without it an   */
                            /* edge belongs to slot s only (arbitrary, non-nested adjacency list).      */
};

enum { /* `op` of ctgcn_workspace_bytes */
    CTGCN_OP_KCORE = 1,
    CTGCN_OP_INGEST = 2     /* pass the number of edge rows m as `nnz` */
};

#define CTGCN_MAX_SLOTS 255

/* arithmetic of the GRU matrix products (ctgcn_gru_seq_f32, ctgcn_gru_input_proj_f32) */
#define CTGCN_SPLIT_NONE 0
#define CTGCN_SPLIT_BF16X3 1
#define CTGCN_SPLIT_F16X2 2
#define CTGCN_ACT_NONE 0
#define CTGCN_ACT_SELU 1     /* torch.nn.functional.selu, layers.py:103-104 */

int ctgcn_abi_version(void);
const char *ctgcn_last_error(void);

/* Name of the device the library would run on and its compute-unit count (diagnostics). */
int ctgcn_device_info(char *name_host, size_t name_len, int *cu_count_host);

/*
 * out[n, d] = w[d, n]^T + bias[d]  — nn.Linear applied to one-hot node features (the sparse identity of reference
 * helper.py:161-172 fed to layers.py:95-106 MLP): X·W^T + b with X = I is a transpose of the weight.  w rows are ldw
 * floats apart, out rows ldo; bias may be NULL.
 */
int ctgcn_transpose_bias_f32(int64_t n, int32_t d, const float *w, int64_t ldw, const float *bias, float *out, int64_t ldo,
                             void *stream);

/*
 * Y = A·X  (accumulate == 0)   or   Y += A·X  (accumulate != 0).
 * Replaces one torch.sparse.mm(adj, x), layers.py:43 / layers.py:45.
 */
int ctgcn_spmm_csr_f32(int64_t n_rows, int32_t d, const int32_t *row_ptr, const int32_t *col_idx,
                       const float *val, const float *X, int64_t ldx, float *Y, int64_t ldy,
                       int accumulate, void *stream);

/*
 * The whole aggregation loop of CoreDiffusion.forward, layers.py:41-48, plus the
 * stack/transpose of layers.py:58, in ONE pass over the edge list:
 *     res_0 = A_0·X (+ X if SELF_LOOP);  res_j = res_{j-1} + A_j·X;  H[:, j, :] = relu?(res_j)
 * The K matrices are given as one CSR whose entries carry a slot tag and are sorted by
 * (row, slot, col):  NESTED: entry tagged s is present in A_s, A_{s+1}, ..., A_{K-1};
 * otherwise it is present in A_s only.  H is [n_rows, K, d] contiguous (the [batch, core, feat]
 * layout nn.GRU(batch_first=True) consumes at layers.py:59).   1 <= K <= CTGCN_MAX_SLOTS.
 * long_rows (optional, device int32[n_long]): the rows with more than long_threshold entries (hubs).  They are
 * skipped by the row-per-lane-group kernel and processed one 1024-thread block each, so that a 100k-entry row does
 * not serialise on 32 lanes.  n_long == 0 disables the split.
 * hub_split > 1 with a hub workspace (ctgcn_hub_workspace_bytes(n_long, hub_split, slots = K forward / 1 backward, d) bytes, 16-byte
 * aligned): a hub row of L entries is cut into min(hub_split, ceil(L / 8192)) pieces, one block each; a second small kernel adds the
 * pieces' partial sums in piece order and finishes the row (two passes, fixed order: results do not depend on scheduling).
 * hub_split <= 1 or a NULL workspace: one block per hub row.  hub_split <= 64.
 */
size_t ctgcn_hub_workspace_bytes(int32_t n_long, int32_t hub_split, int32_t slots, int32_t d);
/* entries per piece when a hub row is cut (the 8192 above): callers derive hub_split = ceil(longest row / this) from it */
int32_t ctgcn_hub_split_entries(void);
int ctgcn_core_aggregate_f32(int64_t n_rows, int32_t d, int32_t K, const int32_t *row_ptr,
                             const int32_t *col_idx, const float *val, const uint8_t *slot,
                             const float *X, int64_t ldx, float *H, uint32_t flags,
                             const int32_t *long_rows, int32_t n_long, int32_t long_threshold,
                             int32_t hub_split, void *hub_workspace, size_t hub_workspace_bytes, void *stream);

/*
 * Backward of ctgcn_core_aggregate_f32 w.r.t. X (what autograd derives for layers.py:41-48).
 * Step 1 (ctgcn_core_aggregate_bwd_prep_f32), elementwise over [n, K, d]:
 *     G_j = dH_j * [H_j > 0] (RELU) ; S_j = sum_{i>=j} G_i ; Z_j = NESTED ? sum_{i>=j} S_i : S_j
 *     S0 = S_0  ([n, d], the self-loop term; may be NULL when SELF_LOOP is not set)
 * Step 2 (ctgcn_core_aggregate_bwd_f32) on the CSR of the TRANSPOSED matrices (same arrays when
 * the adjacency is symmetric, which every matrix the reference loader builds is):
 *     dX[r] = S0[r] (SELF_LOOP) + sum_{e in row r} val[e] * Z[col[e], slot[e], :]
 */
int ctgcn_core_aggregate_bwd_prep_f32(int64_t n_rows, int32_t d, int32_t K, const float *dH,
                                      const float *H, float *Z, float *S0, uint32_t flags,
                                      void *stream);
int ctgcn_core_aggregate_bwd_f32(int64_t n_rows, int32_t d, int32_t K, const int32_t *row_ptr,
                                 const int32_t *col_idx, const float *val, const uint8_t *slot,
                                 const float *Z, const float *S0, float *dX, int64_t lddx,
                                 uint32_t flags, const int32_t *long_rows, int32_t n_long,
                                 int32_t long_threshold, int32_t hub_split, void *hub_workspace, size_t hub_workspace_bytes,
                                 void *stream);

/*
 * Edge rows (in file order) -> the snapshot's simple undirected weighted graph as symmetric, zero-diagonal CSR
 * with sorted columns.  Replaces the graph construction at utils.py:23-30 (get_nx_graph) and utils.py:35-58
 * (get_sp_adj_mat): every row sets weight({src,dst}) = w (1.0 when w == NULL); the LAST row naming an
 * unordered pair wins; rows with src == dst are dropped.  src/dst: int32[m] node indices in [0, n).
 * Outputs: row_ptr int32[n+1]; col_idx int32 / val float with capacity 2*m; *nnz_host = entries stored
 * (2 x distinct pairs).  workspace: ctgcn_workspace_bytes(CTGCN_OP_INGEST, n, m, 0, 0).  Synchronises `stream`.
 */
int ctgcn_edges_to_csr(int64_t n, int64_t m, const int32_t *src, const int32_t *dst, const float *w,
                       int32_t *row_ptr, int32_t *col_idx, float *val, int64_t *nnz_host,
                       void *workspace, size_t workspace_bytes, void *stream);

/*
 * Core number of every vertex of an undirected simple graph given as symmetric CSR structure
 * (self-loop entries, if any, are ignored).  Replaces networkx.core_number at
 * preprocessing/structure_generation.py:35.  Integer result, unique, bit-exact.
 * workspace: ctgcn_workspace_bytes(CTGCN_OP_KCORE, n, nnz, 0, 0) bytes.
 * level_cap <= 0: exact core numbers.  level_cap = L > 0: only levels 0..L-1 are peeled and every vertex whose core
 * number is >= L is reported as L — all a loader with max_core = L needs (helper.py:63 keeps k <= max_core, so the
 * levels above it are never told apart), at a fraction of the peel depth.
 * max_core_host (optional) receives max(core) (capped likewise).  Synchronises `stream`.
 */
int ctgcn_kcore_i32(int64_t n, const int32_t *row_ptr, const int32_t *col_idx, int32_t *core,
                    void *workspace, size_t workspace_bytes, int32_t level_cap, int32_t *max_core_host,
                    void *stream);

/*
 * level[e] = min(core[row(e)], core[col(e)]) for every CSR entry: entry e belongs to the k-core
 * subgraph A_k (structure_generation.py:48-53, networkx.k_core) iff level[e] >= k.
 * Also accumulates, for L in [0, hist_len): count[L] = #entries with min(level, hist_len-1) == L
 * and wsum[L] = sum of their weights (fp64) — what helper.py:74-75 needs to decide
 * `delta.sum() == 0`.  count/wsum must be zeroed by the caller; either may be NULL.
 */
int ctgcn_edge_levels_i32(int64_t n, const int32_t *row_ptr, const int32_t *col_idx,
                          const float *val, const int32_t *core, int32_t *level,
                          int64_t *count, double *wsum, int32_t hist_len, void *stream);

/*
 * Tag every entry with its slot (slot_of_level[min(level, table_len-1)], table of uint8 on the
 * device) and stably re-order the entries of each row by slot, producing the (row, slot, col)
 * order ctgcn_core_aggregate_f32 consumes.  Encodes helper.py:63-78 (file truncation, reversal,
 * skipped matrices) once the caller has built the table.  Outputs must not alias inputs.
 */
int ctgcn_slot_reorder(int64_t n, int32_t K, const int32_t *row_ptr, const int32_t *col_idx,
                       const float *val, const int32_t *level, const uint8_t *slot_of_level,
                       int32_t table_len, int32_t *col_out, float *val_out, uint8_t *slot_out,
                       void *stream);

/*
 * Recurrent part of a single-layer GRU over `steps` for `rows` independent sequences, hidden = 128,
 * fused with the reduction the reference applies to its output:
 *   reduce_sum != 0 :  out[rows,128]        = LayerNorm(sum_t h_t)   nn.GRU + .sum(dim=1) + LayerNorm, layers.py:59-62
 *   reduce_sum == 0 :  out[rows,steps,128]  = LayerNorm(h_t)         nn.GRU + LayerNorm, models.py:249-250
 * gi [rows, steps, 384] is the input projection x·W_ih^T + b_ih (+ b_hh for the r and z gates) in PyTorch's
 * gate order r,z,n — a plain GEMM the caller runs with its BLAS; w_hh [384,128] and b_hn [128] (the n-gate's
 * hidden bias, NULL = 0) are the module's weight_hh_l0 and bias_hh_l0[256:384].  ln_weight == NULL skips the
 * LayerNorm.  h_0 = 0.  split_bf16 selects the arithmetic of the product h_{t-1}·W_hh^T (all fp32-accurate, fp32 I/O):
 *   CTGCN_SPLIT_NONE   (0)  f32-input MFMA (an fmaf chain)
 *   CTGCN_SPLIT_BF16X3 (1)  operands split exactly into three bf16 terms, six partial products, fp32 accumulation
 *   CTGCN_SPLIT_F16X2  (2)  operands scaled per row by a power of two and split into two fp16 terms (22 bits), three
 *                           partial products, fp32 accumulation: half the matrix-core work of (1) and measured MORE
 *                           accurate than (0) and (1) (tools/probes/mfma_f16x2_probe.hip)
 * ld_out (reduce_sum only; 0 = dense 128): floats between output rows — lets a snapshot's embeddings land directly in
 *   column t of the [nodes, T, 128] input of the temporal GRU (models.py:248 stack + transpose without the copies).
 * gi_blocked != 0 (CTGCN_SPLIT_F16X2 only): gi is in the tile layout ctgcn_gru_input_proj_f32 writes for steps_blocked = steps
 *   ([node tile of 64][step][gate][16-column group][node in tile][16]; the buffer must cover ceil(rows/64)*64 rows).
 * gates_out (optional; requires reduce_sum == 0 and ln_weight == NULL): [rows, steps, 4, 128] receives r, z, n and
 * q = W_hn·h_{t-1} + b_hn for ctgcn_gru_seq_bwd_f32.
 * row_order / tile_mask / tile_base (all or none; reduce_sum, CTGCN_SPLIT_F16X2, plain gi layout, steps <= 64): gi was projected from
 * the COMPACT operand rows ctgcn_core_aggregate_split_f32 wrote under a row plan with tiles of 64 — only the steps that bring a new x
 * row exist: tile T starts at gi row tile_base[T], sequence p of it owns popcount(tile_mask[T]) consecutive rows, step t reads the row of
 * the last set bit <= t.  Sequence p is written to out row row_order[p].  Same arithmetic on the same numbers: bit-identical.
 */
int ctgcn_gru_seq_f32(int64_t rows, int32_t steps, int32_t hidden, const float *gi, const float *w_hh,
                      const float *b_hn, const float *ln_weight, const float *ln_bias, float ln_eps,
                      int reduce_sum, float *out, int64_t ld_out, float *gates_out, int split_bf16, int gi_blocked,
                      const int32_t *row_order, const uint32_t *tile_mask, const int32_t *tile_base, void *stream);

/*
 * Backward of the LayerNorm(128) behind the GRU (layers.py:61-62 norm(output.sum(dim=1)); models.py:250 norm(output)):
 *   x[r] = sum_{t < steps} h[r, t, :]  (steps = 1: h is [rows, 128]);  dx = dL/dx given dy = dL/dLayerNorm(x) (the mean / rstd are recomputed),
 *   partial [n_partial, 256]: per-block partial sums of (dgamma | dbeta), every row written; the caller adds them up (deterministic).
 *   ld_dy: floats between the rows of dy (0 = 128; larger: dy is a column of a [rows, T, 128] gradient, no copy needed).
 * One pass instead of the framework's three kernels + sum + forward recompute.  n_partial = number of blocks (<= 65535, e.g. 2048).
 *   dy_rows (optional, device int32[rows]): row p of h / dx belongs to row dy_rows[p] of dy (the GRU ran in a row-plan order).
 */
int ctgcn_layernorm_bwd_f32(int64_t rows, int32_t steps, int32_t hidden, const float *h, const float *dy, int64_t ld_dy, const float *gamma, float eps,
                            float *dx, float *partial, int32_t n_partial, const int32_t *dy_rows, void *stream);

/*
 * The same for nn.LSTM (rnn_type = 'LSTM'; layers.py:27-28, models.py:234-235): gi [rows, steps, 512] = x·W_ih^T + b_ih
 * + b_hh in PyTorch's gate order i,f,g,o; w_hh [512, 128]; h_0 = c_0 = 0.  Exact fp32 (f32-input MFMA).
 * gates_out (optional; reduce_sum == 0, no LayerNorm): [rows, steps, 5, 128] for ctgcn_lstm_seq_bwd_f32 (training's recompute pass).
 */
int ctgcn_lstm_seq_f32(int64_t rows, int32_t steps, int32_t hidden, const float *gi, const float *w_hh,
                       const float *ln_weight, const float *ln_bias, float ln_eps, int reduce_sum, float *out,
                       float *gates_out, void *stream);
/*
 * Backward of that recurrence (autograd of nn.LSTM).  gates [rows, steps, 5, 128] = i, f, g, o (after their activations) and c, as
 * ctgcn_lstm_seq_f32 writes them into gates_out (reduce_sum == 0, no LayerNorm: `out` is then the raw h sequence).  dh_seq
 * [rows, steps, 128] and / or dh_sum [rows, 128] (added at every step: the gradient of sum_t h_t).  d_gi [rows, steps, 512] = gradient
 * w.r.t. the gate pre-activations (= d of x·W_ih^T + b; d x, d W_ih, d W_hh are GEMMs over it), bias_partial [n_partial, 512]: per-block
 * column sums of d_gi (d b_ih = d b_hh = their sum).  Exact fp32 (f32-input MFMA).
 */
int ctgcn_lstm_seq_bwd_f32(int64_t rows, int32_t steps, int32_t hidden, const float *gates, const float *dh_seq, const float *dh_sum,
                           const float *w_hh, float *d_gi, float *bias_partial, int32_t n_partial, void *stream);

/*
 * layers.py:59-62 / models.py:249-250 in ONE kernel for d_in = hidden = 128 with both weight matrices resident in the register
 * file of the CU (four waves, one per SIMD, 512 registers each): the input projection gi is consumed from the MFMA accumulators
 * and never materialised - not in HBM, not in a workspace.  x [rows, steps, 128] with row-step stride ldx.
 *   reduce_sum != 0: out[rows, ld_out] = LayerNorm(sum_t h_t) (ld_out as in ctgcn_gru_seq_f32, 0 = dense 128)
 *   reduce_sum == 0: out[rows, steps, 128] = LayerNorm(h_t) per step (ld_out must be 0)
 * ln_weight / ln_bias NULL: no LayerNorm.  bias_gi [384] = b_ih (+ b_hh for the r and z gates), b_hn [128] = bias_hh_l0[256:384];
 * either may be NULL.  CTGCN_SPLIT_F16X2 arithmetic, operation for operation that of ctgcn_gru_input_proj_f32 followed by
 * ctgcn_gru_seq_f32: results are bit-identical to the kernel pair.  HBM traffic: x in + out (the pair: 7x that).
 * gates_out (optional; reduce_sum == 0, no LayerNorm): [rows, steps, 4, 128] as ctgcn_gru_seq_f32 writes them - the recompute pass of
 * training without the gi round trip.
 * step_offsets (optional, device int64[steps]) + ld_row: x of step t of sequence r is read at x + r ld_row + step_offsets[t] (floats;
 * offsets multiples of 4) instead of x + (r steps + t) ldx - the steps of a sequence may live in different buffers' regions (the temporal
 * GRU of models.py:249 on the receive buffer of a snapshot-parallel exchange, no stack / transpose copy).
 */
int ctgcn_gru_layer_f32(int64_t rows, int32_t steps, int32_t d_in, int32_t hidden, const float *x, int64_t ldx, const float *w_ih,
                        const float *w_hh, const float *bias_gi, const float *b_hn, const float *ln_weight, const float *ln_bias,
                        float ln_eps, int reduce_sum, float *out, int64_t ld_out, float *gates_out,
                        const int64_t *step_offsets, int64_t ld_row, void *stream);

/*
 * ctgcn_gru_layer_f32 (sum-over-steps form) on an input that ctgcn_core_aggregate_split_f32 already wrote as fp16 planes + row scales
 * (d = hidden = 128: `planes` = that call's workspace for n_rows = rows, K = steps).  The aggregation (HBM-bound) does the per-row
 * max / scale / split, this matrix-core-bound kernel only copies the planes into LDS: the whole CoreDiffusion layer of
 * layers.py:41-62 in two kernels, x [rows, steps, 128] never exists in fp32.  Bit-identical to ctgcn_core_aggregate_f32 +
 * ctgcn_gru_layer_f32.  Inference only.
 *   row_order / tile_mask (both or neither; steps <= 64): the row plan the aggregation call was given (see there).  Sequence p of the
 *   planes is written to out row row_order[p]; a step whose tile_mask bit is clear re-uses the x·W_ih products of the step before
 *   (its x row is the same row again: layers.py:41-48 with no entry of that slot or below in any of the tile's 16 rows) and costs the
 *   h·W_hh half only.  Same products in the same order: results are bit-identical to the call without a plan.
 */
int ctgcn_gru_layer_presplit_f32(int64_t rows, int32_t steps, int32_t hidden, const void *planes, const float *w_ih, const float *w_hh,
                                 const float *bias_gi, const float *b_hn, const float *ln_weight, const float *ln_bias, float ln_eps,
                                 float *out, int64_t ld_out, const int32_t *row_order, const uint32_t *tile_mask, void *stream);

/*
 * Training under the row plan (round 4): the backward of a CoreDiffusion layer with d_in = hidden = 128 (layers.py:41-62; what
 * embedding.py:347-348 runs every batch) in position order, chunk by chunk, with two intermediates in HBM instead of six.
 *
 * ctgcn_gru_layer_presplit_save_f32 — the recompute pass: ctgcn_gru_layer_presplit_f32's kernel on the sequences
 *   [first_row, first_row + rows) of the forward's planes (`planes`, plane_rows = n K rows per plane; first_row a multiple of 16;
 *   tile_mask, optional, already points at tile first_row / 16), writing the gates [rows, steps, 4, 128] (r, z, n, q = W_hn h + b_hn), the
 *   raw h sequence [rows, steps, 128] and the pre-LayerNorm sum over the steps [rows, 128] — no LayerNorm, no `out`.  Its gates are
 *   [rows, steps, 3, 128] = r, z, q (gate_count = 3 below): n is rebuilt from h_t, h_{t-1} and z, h_t = n + z (h_{t-1} - n).
 * ctgcn_gru_bwd_rec_f32 — backward recurrence + dW_hh: gates (gate_count = 4: r, z, n, q as ctgcn_gru_layer_f32 / ctgcn_gru_seq_f32 write them; 3: r, z, q), h_seq as above; exactly one of dh_sum [rows, 128] (gradient of sum_t h_t,
 *   added at every step) / dh_seq [rows, steps, 128]; d_gi [rows, steps, 384] receives, at every step whose tile_mask bit is set, the sum
 *   of (da_r, da_z, da_n) over that step and the steps after it that repeat its x (tile_mask NULL: every step);
 *   dw_partial [n_partial, 384, 128] / dbn_partial [n_partial, 128]: per-block sums of dW_hh and of the n-gate column of d b_hh
 *   (n_partial >= ctgcn_gru_bwd_blocks(rows), rows of the tables beyond that are not touched; accumulate != 0 adds to them: zero the
 *   tables once, accumulate over the chunks, one reduction at the end — deterministic).
 *   bf16 x 2 split arithmetic (three v_mfma_f32_16x16x32_bf16 per product).
 *   steps <= 64 for the three calls of this block (round 6; 32 before): tile_mask holds one word per 16-row tile up to 32 steps and two
 *   (steps 0-31, 32-63; tile t at [2 t], [2 t + 1]) beyond, as the forward's row plan does — core lists of 33-64 matrices (America-Air /
 *   Europe-Air depth) train under the plan too.
 * ctgcn_gru_bwd_in_f32 — dx = d_gi·W_ih, dW_ih, d b_ih over the fresh steps: x either as the forward's planes (x_planes, plane_rows,
 *   first_row as above) or fp32 rows (x, row-step stride ldx).  Output either dx [rows, steps, 128] or (Z != NULL) the aggregation
 *   backward's operands in matrix-row order: Z [n, steps, 128] and S0 [n, 128] (NULL without self loop) with the ReLU mask x > 0 and
 *   the suffix sums of ctgcn_core_aggregate_bwd_prep_f32 applied (nested: CTGCN_F_NESTED lists), written at row row_order[p]
 *   (row_order points at first_row; NULL: p).  Z is only defined at fresh steps — a row has no entry tagged with a step that
 *   repeats, ctgcn_core_aggregate_bwd_f32 never reads the others (symmetric lists).
 */
int32_t ctgcn_gru_bwd_blocks(int64_t rows);
int ctgcn_gru_layer_presplit_save_f32(int64_t rows, int32_t steps, int32_t hidden, const void *planes, int64_t plane_rows, int64_t first_row,
                                      const float *w_ih, const float *w_hh, const float *bias_gi, const float *b_hn, const uint32_t *tile_mask,
                                      float *gates_out, float *hseq_out, float *presum_out, void *stream);
int ctgcn_gru_bwd_rec_f32(int64_t rows, int32_t steps, int32_t hidden, const float *gates, int32_t gate_count, const float *h_seq, const float *dh_sum,
                          const float *dh_seq, const float *w_hh, const uint32_t *tile_mask, float *d_gi, float *dw_partial,
                          float *dbn_partial, int32_t n_partial, int32_t accumulate, void *stream);
int ctgcn_gru_bwd_in_f32(int64_t rows, int32_t steps, int32_t hidden, const float *d_gi, const float *w_ih, const uint32_t *tile_mask,
                         const void *x_planes, int64_t plane_rows, int64_t first_row, const float *x, int64_t ldx, float *dx, float *Z, float *S0,
                         const int32_t *row_order, int32_t nested, float *dw_partial, float *dbi_partial, int32_t n_partial,
                         int32_t accumulate, void *stream);

/*
 * One launch per kernel for a whole WINDOW of small snapshots (reference models.py:243-247 loops over the snapshots; at 6 828 - 87 036 nodes a
 * snapshot's aggregation is 25 - 200 us of kernel, mostly ramp, tail and dependent load chains): the width-128 CoreDiffusion layer of `groups`
 * snapshots that share the node set (n_rows each), every snapshot with its own graph, inputs, weights and outputs.
 *   ctgcn_core_aggregate_split_group_f32   = ctgcn_core_aggregate_split_f32(d = 128, n_out = 1, no hub rows) per group, one grid
 *   ctgcn_gru_layer_presplit_group_f32     = ctgcn_gru_layer_presplit_f32 per group, one persistent grid of <= one block per CU: a block serves
 *                                            one snapshot (its weights stay resident), blocks are dealt out in proportion to `work`
 * Same kernels' code per snapshot: bit-identical to the per-snapshot calls.  g: HOST array of `groups` descriptors; table: device memory,
 * 256-byte aligned, ctgcn_group_table_bytes(groups) bytes, that the call fills on `stream` before the launch (do not reuse it for
 * another call before that call's kernel has run — same stream: fine).  Hub rows (longer than the caller's long-row threshold) are not
 * handled here: windows that have any take the per-snapshot calls.
 *
 * How the table is filled (ABI 28; rounds 4-5 used hipMemcpyAsync from pageable memory: a host stall per call, not capturable): the descriptor
 * bytes travel as kernel arguments of a small writer kernel — asynchronous, stream-ordered, recordable into a hipGraph, no host-to-device copy.
 * `shadow` (all ctgcn_*_group_f32 calls; may be NULL): HOST memory of the table's size that the caller zero-fills once and keeps together with
 * `table`.  The call compares this call's descriptors with the shadow; when they are equal the device table is current and NOTHING is written
 * (the steady state of an inference loop over one window: same graphs, weights and buffers forward after forward); otherwise table and shadow
 * are rewritten.  A caller that passes a shadow promises that nothing else writes to `table`, and that it always uses the pair on the same
 * stream (the table's last writer and its readers must be stream-ordered).  One (table, shadow) pair per call site: two calls sharing a table
 * would rewrite it every time.
 */
typedef struct {
    const int32_t *row_ptr, *col_idx;
    const float *val;
    const uint8_t *slot;
    const float *X;
    int64_t ldx;
    int32_t K;
    uint32_t flags;
    const int32_t *row_order;      /* row plan (both or neither), as ctgcn_core_aggregate_split_f32 */
    const uint32_t *tile_mask;
    void *workspace;               /* d = 128, consumer = the GRU layer kernel: planes + row scales of the group,
                                      ctgcn_core_aggregate_split_workspace_bytes(n_rows, 128, K, 1, 0) bytes, 256-byte aligned (planes1 = NULL) */
    size_t workspace_bytes;
    /* consumer = the split GEMM (round 5: the 500-wide first layer, any d <= 512): the group's operand rows inside planes SHARED by the window —
     * pointers to the group's first row of plane 1 / plane 2 (row length = d rounded up to 64 halfs) and of the scales; with a row plan the
     * rows are compact (tile 64) and tile_base gives each tile's first row RELATIVE to the group's first row */
    void *planes1, *planes2;
    float *scales;
    const int32_t *tile_base;
} ctgcn_agg_split_group_t;
typedef struct {
    const void *planes;            /* the group's aggregation workspace */
    const float *w_ih, *w_hh, *bias_gi, *b_hn, *ln_weight, *ln_bias;
    float ln_eps;
    int32_t steps;                 /* K of the snapshot */
    float *out;
    int64_t ld_out;
    const int32_t *row_order;
    const uint32_t *tile_mask;
    int64_t work;                  /* relative cost (e.g. (position, step) rows that bring a new x + rows x steps); <= 0: equal shares */
} ctgcn_gru_layer_group_t;
typedef struct {
    const float *gi;               /* gate pre-activations of the group's (compact) operand rows, [rows, 384] */
    const float *w_hh, *b_hn, *ln_weight, *ln_bias;
    float ln_eps;
    int32_t steps;
    float *out;
    int64_t ld_out;
    const int32_t *row_order;      /* the aggregation's row plan (tile 64): all three or none */
    const uint32_t *tile_mask;
    const int32_t *tile_base;
    int64_t work;
} ctgcn_gru_seq_group_t;
size_t ctgcn_group_table_bytes(int32_t groups);
/* Diagnostic: descriptor tables written by the grouped calls of this process so far (current = 0), or found current through their shadow and
 * not written (current = 1).  A steady-state inference loop moves only the second counter (tests/test_gpu_group.py). */
uint64_t ctgcn_table_uploads(int current);
/* Round 5, the 500-wide first CoreDiffusion layer of a small window in one launch per kernel (reference models.py:243-247 loops over the
 * snapshots): Linear(I) = W^T + b of every snapshot; the aggregation into shared operand planes (ctgcn_agg_split_group_t, GEMM form);
 * ctgcn_linear_packed_group_f32 (one panel GEMM over all snapshots' rows, weights per snapshot); the recurrences. */
int ctgcn_transpose_bias_group_f32(int32_t groups, int64_t n, int32_t d, const float *const *w, int64_t ldw, const float *const *bias, float *const *out,
                                   int64_t ldo, void *table, size_t table_bytes, void *shadow, void *stream);
int ctgcn_gru_seq_group_f32(int32_t groups, int64_t rows, int32_t hidden, const ctgcn_gru_seq_group_t *g, void *table, size_t table_bytes, void *shadow,
                            void *stream);
int ctgcn_core_aggregate_split_group_f32(int32_t groups, int64_t n_rows, int32_t d, const ctgcn_agg_split_group_t *g, void *table, size_t table_bytes,
                                         void *shadow, void *stream);
int ctgcn_gru_layer_presplit_group_f32(int32_t groups, int64_t rows, int32_t hidden, const ctgcn_gru_layer_group_t *g, void *table, size_t table_bytes,
                                       void *shadow, void *stream);

/*
 * Dense  y[rows, n_out] = x[rows, k]·w[n_out, k]^T + bias  (bias [n_out] may be NULL) in fp32-accurate fp16x2 split arithmetic on the
 * matrix cores (CTGCN_SPLIT_F16X2: operand rows scaled by a power of two and written as two fp16 terms, three
 * v_mfma_f32_16x16x32_f16 per product, fp32 accumulation) - the GRU input projection for d_in != 128 (layers.py:59 with
 * input_size = hid_dim = 500) and nn.Linear on dense inputs (layers.py:95-106), which otherwise run as fp32 library GEMMs.
 * ldx / ldw / ldy: row strides in floats.  Any k >= 1 (rows that are only 4-byte aligned, e.g. k = 1737, are read with the same 16-byte
 * loads), any n_out (more than 512 columns: one launch per 512).  activation: CTGCN_ACT_NONE, or CTGCN_ACT_SELU applied to y in the
 * epilogue (the F.selu after each Linear of an 'N' MLP).
 * workspace: ctgcn_linear_workspace_bytes(rows, n_out, k) bytes, 256-byte aligned: ctgcn_split_planes_bytes(rows, k) bytes of x's operand
 * planes followed by ctgcn_pack_weight_bytes(n_out, k) bytes of w's packed operand.
 */
size_t ctgcn_linear_workspace_bytes(int64_t rows, int32_t n_out, int32_t k);
/*
 * The same product with the operands kept by the caller: an operand that does not change between calls — the weights of an inference run,
 * node features the reference builds once and feeds to every batch (train.py:72-76) — is prepared ONCE:
 *   ctgcn_split_rows_f32      x side: rows x k fp32 (row stride ldx) -> `planes`: plane 1 [rows, kp] | plane 2 | row scales, kp = k rounded up
 *                             to 64 (zero padded), ctgcn_split_planes_bytes(rows, k) bytes, 256-byte aligned
 *   ctgcn_pack_weight_f32     w side (round 5): n_out x k fp32 (row stride ldw) -> `packed`: the same split stored in matrix-core fragment order
 *                             ([column tile of 16][k slab of 32][plane][lane][8 halfs], column tiles padded to a multiple of 8 per chunk of 512
 *                             columns) | column scales; ctgcn_pack_weight_bytes(n_out, k) bytes, 256-byte aligned
 *   ctgcn_linear_packed_f32   y = x·w^T + bias (+ activation) from x's planes and w's packed operand (gemm_h2_panel_kernel: persistent blocks,
 *                             each 128-row panel of x read once over the full width)
 * Same split, same kernel as ctgcn_linear_f32: bit-identical results.
 */
size_t ctgcn_split_planes_bytes(int64_t rows, int32_t k);
size_t ctgcn_pack_weight_bytes(int32_t n_out, int32_t k);
int ctgcn_split_rows_f32(int64_t rows, int32_t k, const float *x, int64_t ldx, void *planes, size_t planes_bytes, void *stream);
int ctgcn_pack_weight_f32(int32_t n_out, int32_t k, const float *w, int64_t ldw, void *packed, size_t packed_bytes, void *stream);
int ctgcn_linear_packed_f32(int64_t rows, int32_t n_out, int32_t k, const void *x_planes, const void *w_packed, const float *bias,
                            int32_t activation, float *y, int64_t ldy, void *stream);
/* ctgcn_linear_packed_f32 whose output leaves as the NEXT layer's X operand (the hidden layers of the MLP, reference layers.py:95-106, in
 * inference): out_planes = what ctgcn_split_rows_f32(rows, n_out, y) would write for y = act(x w^T + bias) — per-row scale + two fp16 planes
 * [rows, n_out rounded up to 64], bit for bit — without y going to memory as fp32 and coming back.  n_out <= 512;
 * out_planes: ctgcn_split_planes_bytes(rows, n_out) bytes, 256-byte aligned. */
int ctgcn_linear_packed_chain_f32(int64_t rows, int32_t n_out, int32_t k, const void *x_planes, const void *w_packed, const float *bias, int32_t activation,
                                  void *out_planes, size_t out_planes_bytes, void *stream);
/* ctgcn_linear_packed_f32 for the operand rows of `groups` snapshots in ONE launch (the 500-wide first layer of a small window): planes1 /
 * planes2 / scales / y hold the rows of all groups one after the other, every group padded to whole panels of 128 rows (total_rows % 128 == 0;
 * rows a group does not fill are multiplied and written like any other: give them finite contents or ignore them); panel_group (DEVICE,
 * total_rows / 128 entries) names the group of every panel; w_packed[i] / bias[i] (host arrays of device pointers): the group's packed weight
 * and bias.  n_out <= 512.  table / shadow: as for the other grouped calls above (256-byte aligned device memory, >= 24 bytes per group —
 * ctgcn_group_table_bytes covers it; optional host shadow of the same size).  Same arithmetic per row as ctgcn_linear_packed_f32: bit-identical to the per-group calls. */
int ctgcn_linear_packed_group_f32(int32_t groups, int64_t total_rows, int32_t n_out, int32_t k, const void *planes1, const void *planes2,
                                  const float *scales, const int32_t *panel_group, const void *const *w_packed, const float *const *bias,
                                  int32_t activation, float *y, int64_t ldy, void *table, size_t table_bytes, void *shadow, void *stream);
int ctgcn_linear_f32(int64_t rows, int32_t n_out, int32_t k, const float *x, int64_t ldx, const float *w, int64_t ldw, const float *bias,
                     int32_t activation, float *y, int64_t ldy, void *workspace, size_t workspace_bytes, void *stream);

/*
 * CoreDiffusion aggregation (ctgcn_core_aggregate_f32: layers.py:41-48,58) whose only consumer is the GRU input projection of a
 * layer with d_in != 128 (layers.py:59, nn.GRU(input_size = hid_dim = 500, ...)): instead of the fp32 H [n_rows, K, d] the kernel
 * writes what ctgcn_linear_f32 would make of it - the per-row scales and the two fp16 planes of the n_rows*K operand rows
 * (row = node*K + core) - straight into that GEMM's workspace, bit-identical to aggregate + ctgcn_linear_f32, without the
 * write + read + write of the fp32 intermediate.  Inference only (nothing is kept for a backward pass).
 *   d % 4 == 0, d <= 512, X 16-byte aligned, ldx % 4 == 0.  The same planes serve both consumers (one split
 *   definition: scale 2^(e-14), leading term fp16(x/s), residual fp16(x/s - leading)); at d = 128 that is ctgcn_gru_layer_presplit_f32.
 *   workspace: ctgcn_core_aggregate_split_workspace_bytes(n_rows, d, K, n_out, n_long) bytes, 256-byte aligned; n_out is the
 *   width of the projection that follows (its weight planes share the workspace); n_long hub rows pass through an fp32
 *   scratch at the end of it.
 * ctgcn_linear_presplit_f32(rows = n_rows*K, n_out, k = d, w, ...) then runs the GEMM on that workspace.
 *   Row plan (row_order, tile_mask: both or neither; K <= 64.  K <= 32: one mask word per tile; 33 <= K <= 64 (America-Air max core 64,
 *   Europe-Air 33): two, tile_mask[2 T] = slots 0-31, tile_mask[2 T + 1] = slots 32-63 - for every consumer of the same plan).
 *   A node whose first stored entry is tagged f has H[v, 0] = ... = H[v, f-1] = relu(x_v) (only the self loop has arrived): rows the
 *   reference computes, stacks and multiplies by W_ih f times (layers.py:41-48,58-59).  With a plan, operand rows p K .. p K + K - 1
 *   belong to matrix row row_order[p] (a permutation that puts rows with equal repeat patterns next to each other, so that the 16
 *   sequences of a GRU tile share one), and slot j of position p is only WRITTEN when bit j of tile_mask[p / 16] is set (bit 0 always
 *   is; a clear bit j promises that slot j repeats slot j - 1 for all 16 positions of the tile).  The planes' layout does not change -
 *   skipped rows are holes nobody reads.
 *   GEMM consumer (d != 128) under a row plan: tiles of 64 positions (the row tile of ctgcn_gru_seq_f32) and tile_base int32[tiles]:
 *   the operand rows are COMPACT - tile T's written rows start at row tile_base[T], position p owns popcount(tile_mask[T])
 *   consecutive rows (one per set bit, in slot order) from tile_base[T] + (p % 64) popcount(tile_mask[T]); operand_rows = their total
 *   (incl. the padding of the last tile), which is the row count ctgcn_linear_presplit_f32 is then called with, and
 *   ctgcn_gru_seq_f32 takes the same plan.  tile_base NULL with d = 128, n_out = 1.
 *   hub_row_dest int32[n_long K] (required under a row plan when n_long > 0): operand row of slot j of hub row long_rows[i], or -1
 *   when the plan does not want that slot.
 */
size_t ctgcn_core_aggregate_split_workspace_bytes(int64_t n_rows, int32_t d, int32_t K, int32_t n_out, int32_t n_long);
int ctgcn_core_aggregate_split_f32(int64_t n_rows, int32_t d, int32_t K, const int32_t *row_ptr, const int32_t *col_idx,
                                   const float *val, const uint8_t *slot, const float *X, int64_t ldx, uint32_t flags,
                                   const int32_t *long_rows, int32_t n_long, int32_t long_threshold, int32_t n_out,
                                   const int32_t *row_order, const uint32_t *tile_mask, const int32_t *tile_base, int64_t operand_rows,
                                   const int32_t *hub_row_dest,
                                   int32_t hub_split, void *hub_workspace, size_t hub_workspace_bytes,
                                   void *workspace, size_t workspace_bytes, void *stream);
int ctgcn_linear_presplit_f32(int64_t rows, int32_t n_out, int32_t k, const float *w, int64_t ldw, const float *bias, float *y, int64_t ldy,
                              void *workspace, size_t workspace_bytes, void *stream);

/*
 * Backward of the recurrence above (autograd of nn.GRU, layers.py:59 / models.py:249).  Inputs: the saved gates and
 * the raw h sequence h_seq [rows, steps, 128]; the upstream gradient per step dh_seq [rows, steps, 128] and/or one
 * gradient dh_sum [rows, 128] added at every step (the .sum(dim=1) case).  Outputs, for the caller's GEMMs:
 *   d_gi  [rows, steps, 384] = dL/d(x·W_ih^T + b_ih)      -> dX = d_gi·W_ih, dW_ih = d_gi^T·x, db_ih = sum d_gi
 *   d_ghn [rows, steps, 128] = dL/d(W_hn·h + b_hn)         -> dW_hh = [d_gi_r, d_gi_z, d_ghn]^T·h_{t-1}, db_hh likewise
 * bias_partial (nullable) [n_partial, 512]: on return its rows sum to the bias gradients — columns 0..383 = sum over
 *   (row, step) of d_gi, columns 384..511 = of d_ghn (one row per launched block, the rest zeroed; n_partial >= the
 *   CU count uses the whole device).  Saves re-reading d_gi / d_ghn for db_ih / db_hh.
 * split_bf16: as in ctgcn_gru_seq_f32 (the product dGH·W_hh on the bf16 matrix cores, fp32-accurate).
 */
int ctgcn_gru_seq_bwd_f32(int64_t rows, int32_t steps, int32_t hidden, const float *gates, const float *h_seq,
                          const float *dh_seq, const float *dh_sum, const float *w_hh, float *d_gi, float *d_ghn,
                          float *bias_partial, int32_t n_partial, int split_bf16, void *stream);

/*
 * The GRU input projection  gi[rows, 384] = x[rows, 128]·w_ih^T + bias  (bias [384] may be NULL) for d_in = hidden = 128,
 * in fp32-accurate split arithmetic on the matrix cores: split_mode = CTGCN_SPLIT_BF16X3 or CTGCN_SPLIT_F16X2 (see
 * ctgcn_gru_seq_f32; error vs fp64 no larger than an fp32 fmaf chain's, measured).  Other widths: use a BLAS GEMM.
 * steps_blocked > 0 (CTGCN_SPLIT_F16X2 only): rows = nodes*steps_blocked rows of [nodes, steps, 128] sequences; gi is written
 * in the recurrence kernel's tile layout (see ctgcn_gru_seq_f32, gi_blocked) instead of [rows, 384].
 */
int ctgcn_gru_input_proj_f32(int64_t rows, int32_t d_in, int32_t hidden, const float *x, int64_t ldx,
                             const float *w_ih, const float *bias, float *gi, int split_mode, int32_t steps_blocked, void *stream);

/*
 * Gradient of the projection w.r.t. its input: d_x[rows, 128] = d_gi[rows, 384] · W_ih  (autograd of the F.linear inside
 * nn.GRU, reference layers.py:59), same split-bf16 arithmetic.  d_x rows are ldx floats apart.
 */
int ctgcn_gru_input_grad_f32(int64_t rows, int32_t d_in, int32_t hidden, const float *d_gi, const float *w_ih, float *d_x,
                             int64_t ldx, void *stream);

/*
 * Weight gradients of the GRU (autograd of nn.GRU's two F.linear, reference layers.py:59 / models.py:249):
 *   partial[p] (p < n_pairs), each [384, 128], sum over p = sum over r < rows of G[r, :]^T · X'[r, :]
 * G = gate columns [0,256) from g01 (row stride ldg01) and [256,384) from g2 (row stride ldg2): dW_ih takes both from
 * d_gi; dW_hh takes g2 = d_ghn.  X' = x (row stride ldx), or with shift_steps != 0 the rows shifted by one step inside
 * each `steps`-long sequence (X'[r] = r % steps ? x[r-1] : 0): h_{t-1} read straight from the h sequence.
 * accumulate != 0 adds to `partial` instead of overwriting it (one host-side sum after many calls).
 * rows = nodes * steps.  n_pairs: a positive multiple of 8 (half the CU count uses the whole device).  Split-bf16
 * arithmetic, deterministic.
 */
int ctgcn_gru_weight_grad_f32(int64_t rows, int32_t steps, int32_t hidden, const float *g01, int64_t ldg01, const float *g2,
                              int64_t ldg2, const float *x, int64_t ldx, int shift_steps, float *partial, int32_t n_pairs,
                              int accumulate, void *stream);

/* Rows one wave of persistent blocks covers (rows per block x compute units): callers that split `rows` into
 * chunks should use multiples of this so that every launch keeps all CUs equally busy. */
int64_t ctgcn_gru_row_granule(void);
/* Compute units the persistent kernels run one block on (the device's, or the number set with ctgcn_set_persistent_cus): the grouped
 * launches take at most this many snapshots per call (ctgcn_gru_layer_presplit_group_f32 gives every snapshot at least one block). */
int32_t ctgcn_compute_units(void);

/*
 * Random-walk corpus (preprocessing/random_walk.py:8-69).  ctgcn_row_cumsum_f32: cumw[e] = inclusive prefix sum of the
 * weights within each CSR row.  ctgcn_random_walk_pairs: `walk_time` walks of walk_length steps from EVERY node (walk
 * indices first_walk .. first_walk+walk_time-1 of the seed's stream); next hop drawn with probability proportional to
 * the edge weight (weighted != 0) or uniformly; a walk stops at a node without neighbours.  For every walk the
 * (L+1)·L/2 position pairs i < j are written to pair_src/pair_dst (capacity n·walk_time·(L+1)·L/2; pairs with equal
 * endpoints or beyond the end of a short walk are written as the self pair (0,0), which ctgcn_edges_to_csr drops) and
 * freq[a]++, freq[b]++ (int64[n], caller-zeroed) for every emitted pair — the reference's node_freq_arr.
 * Feeding the pairs to ctgcn_edges_to_csr (w = NULL) yields the reference's symmetric 0/1 walk_spadj.
 */
int ctgcn_row_cumsum_f32(int64_t n, const int32_t *row_ptr, const float *val, float *cumw, void *stream);
int ctgcn_random_walk_pairs(int64_t n, const int32_t *row_ptr, const int32_t *col_idx, const float *cumw,
                            int32_t walk_length, int32_t walk_time, int32_t first_walk, uint64_t seed, int weighted,
                            int32_t *pair_src, int32_t *pair_dst, int64_t *freq, void *stream);

/*
 * Index draws of the negative-sampling loss (metrics.py:62-93).  For batch node b (batch_nodes[b], int64): all of its
 * walk partners (row of the pair CSR) if there are at most `num`, else `num` of them uniformly without replacement;
 * written at node_out/pos_out[offsets[b] ...] where offsets = exclusive scan of min(deg, num) (caller computes it).
 * neg_out[num]: the table entries at `num` distinct uniformly drawn positions of neg_table (random.sample semantics).
 * scratch: int64[num].
 */
int ctgcn_neg_sampling_indices(int64_t batch, const int64_t *batch_nodes, const int32_t *pair_row_ptr,
                               const int32_t *pair_col, int32_t num, int64_t table_len, const int32_t *neg_table,
                               uint64_t seed, const int64_t *offsets, int64_t *node_out, int64_t *pos_out,
                               int64_t *neg_out, int64_t *scratch, void *stream);

/*
 * HOST function (no GPU work): write one snapshot's embedding [n, d] float32 (host pointer, leading dimension ld) as
 * the text file pandas produces for the reference's save_embedding, embedding.py:79-89
 * (pd.DataFrame(data, index=names).to_csv(path, sep=sep, header=True, index=True)), byte for byte: numpy float32
 * shortest-repr formatting, NaN as an empty field, minimal quoting of names.  names_blob_host holds the n
 * NUL-terminated node names, name_offsets_host[i] is the offset of name i.  threads <= 0: all host threads.
 */
int ctgcn_write_embedding_tsv(const char *path_host, int64_t n, int32_t d, const float *data_host, int64_t ld,
                              const char *names_blob_host, const int64_t *name_offsets_host, char sep,
                              int32_t threads);

size_t ctgcn_workspace_bytes(int op, int64_t n, int64_t nnz, int32_t d, int32_t K);

#ifdef __cplusplus
}
#endif
#endif /* CTGCN_HIP_H */
