/*
 * sagnn.h — C ABI of libsagnn.so: SelfGNN's per-time-interval graph propagation and
 * interval fusion as hand-written HIP kernels for MI355X (gfx950).
 *
 * The reference (LIU-YUXI/SA-GNN, TF1 graph mode) has no FFI: its seam is the Python
 * method Recommender.messagePropagate (model.py:80-92) and the loop around it in
 * Recommender.ours (model.py:118-155). Each entry point below names the reference
 * lines it replaces. The Python host in sa-gnn_amd/ binds these with ctypes
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - Plain C: pointers, sizes, strides in ELEMENTS. No torch / HIP types.
 *     `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - Every `d_*` / unprefixed tensor pointer is DEVICE memory owned by the caller.
 *     `h_*` pointers are HOST memory. The library never frees or retains caller memory
 *     except the two CSR device pointers kept inside a plan object.
 *   - All floating point is fp32, all indices int32 (reference: fp32 / int32 throughout).
 *   - Calls are asynchronous on `stream`; the library never synchronises except in
 *     sagnn_spmm_plan_create (one-off upload of plan metadata).
 *   - Return value: 0 = OK; negative = argument error (SAGNN_ERR_*); positive = hipError_t.
 *     No C++ exception crosses the ABI. sagnn_last_error() gives a thread-local message.
 *   - Feature rows must be 16-byte aligned: base pointers 16-B aligned, every ld a
 *     multiple of 4, d a multiple of 4 with 4 <= d <= 256.
 */
#ifndef SAGNN_H
#define SAGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAGNN_VERSION 10302 /* 1.3.0 */

enum {
  SAGNN_OK = 0,
  SAGNN_ERR_NULL = -1,      /* a required pointer is NULL */
  SAGNN_ERR_DIM = -2,       /* d / t / heads unsupported */
  SAGNN_ERR_ALIGN = -3,     /* pointer or leading dimension not 16-byte aligned */
  SAGNN_ERR_CSR = -4,       /* rowptr not monotone, rowptr[n]!=nnz, colidx out of range */
  SAGNN_ERR_ARG = -5,       /* any other inconsistent argument */
  SAGNN_ERR_WORKSPACE = -6, /* workspace missing or too small */
  SAGNN_ERR_NOMEM = -7      /* host allocation failed */
};

int sagnn_version(void);

/* Arithmetic engine of the GEMM-shaped fusion stages (LSTM gate product, the three dense layers of the
 * attention, their backward products), selected PER CALLING THREAD — there is no environment switch and no
 * process-wide state: a thread that never calls sagnn_set_engine runs SAGNN_ENGINE_F16X2.
 *   SAGNN_ENGINE_F16X2  the default: 16-bit matrix cores over two-piece split fp32 operands (see ARITHMETIC below)
 *   SAGNN_ENGINE_F32    v_mfma_f32_32x32x2_f32 kernels: an fp32 fmaf chain bit for bit (d in {32, 64}; the exact-fp32
 *                       reference the default engine is measured against)
 *   SAGNN_ENGINE_VALU   the plain VALU formulations (any d)
 * sagnn_set_engine returns SAGNN_OK or SAGNN_ERR_ARG; sagnn_get_engine the calling thread's engine. The SpMM is
 * plain fp32 under every engine. */
enum { SAGNN_ENGINE_F16X2 = 0, SAGNN_ENGINE_F32 = 1, SAGNN_ENGINE_VALU = 2 };
int sagnn_set_engine(int engine);
int sagnn_get_engine(void);
/* Number of tiles / chunks the default engine has re-evaluated in fp32 on the current device since the last reset,
 * because an operand left the window of the split (ARITHMETIC below): 0 on ordinary data. Synchronises the device
 * (a diagnostic: tests use it to pin which path produced a result). */
int sagnn_range_redo_count(int64_t* count, int reset);

/* Copies the calling thread's last error text (NUL-terminated, truncated to cap) and
 * returns its full length. */
size_t sagnn_last_error(char* buf, size_t cap);

/* ------------------------------------------------------------------------------------
 * Per-launch timing with HIP events recorded on the launch stream (bench.py's roofline
 * figure). sagnn_profile_enable(capacity) pre-creates `capacity` event pairs and turns
 * recording on (capacity 0 turns it off and frees them); every kernel launch the library
 * issues then takes one slot until they run out. sagnn_profile_read synchronises on the
 * recorded events, returns up to `cap` records in issue order and clears the log.
 * kind: 0 = SpMM row/chunk kernel (units_a = nnz, units_b = n_rows), 1 = SpMM fix-up,
 *       2 = LSTM, 3 = layer-norm, 4 = MHSA+mean (units_a = n, units_b = t).
 * -------------------------------------------------------------------------------- */
int sagnn_profile_enable(int capacity);
int sagnn_profile_read(float* ms, int32_t* kind, int64_t* units_a, int64_t* units_b, int cap,
                       int* n_out);

/* ------------------------------------------------------------------------------------
 * CSR validation (host). Replaces nothing in the reference: TF-CPU raised
 * InvalidArgument from GatherV2 on an out-of-range index (model.py:86); the kernels do
 * not bounds-check, so the host wrapper calls this once per adjacency at load time.
 * -------------------------------------------------------------------------------- */
int sagnn_csr_check_host(const int32_t* h_rowptr, const int32_t* h_colidx,
                         int64_t n_rows, int64_t n_src, int64_t nnz);

/* ------------------------------------------------------------------------------------
 * SpMM plan: per-adjacency metadata, built once (the reference bakes each adjacency into
 * the TF graph as a constant SparseTensor once, model.py:227-237).
 *
 * Rows are handled by degree class:
 *   deg <= short_thresh            one lane-group (d/4 lanes) per row, several rows per wave
 *   short < deg <= long_thresh     one whole wavefront per row
 *   deg > long_thresh              split into chunks of <= chunk_edges edges, one wavefront
 *                                  per chunk into a partial-sum workspace, then a fix-up
 *                                  pass adds the partials in chunk order (deterministic,
 *                                  no atomics)
 * Only the third class needs stored metadata (the chunk list).
 * -------------------------------------------------------------------------------- */
typedef struct sagnn_spmm_plan sagnn_spmm_plan;

typedef struct sagnn_spmm_tuning {
  int32_t short_thresh;  /* 0 = default */
  int32_t long_thresh;   /* 0 = default */
  int32_t chunk_edges;   /* 0 = default; rounded up to a multiple of 64 */
  int32_t reserved;
} sagnn_spmm_tuning;

typedef struct sagnn_spmm_plan_info {
  int64_t n_rows, n_src, nnz;
  int64_t n_long_rows;   /* rows with deg > long_thresh */
  int64_t n_chunks;      /* total chunks over all long rows */
  int32_t short_thresh, long_thresh, chunk_edges;
  int32_t max_degree;
  int32_t on_device;     /* 1 if chunk metadata was uploaded (d_rowptr given) */
  int32_t reserved;
} sagnn_spmm_plan_info;

/* h_rowptr: host copy of rowptr [n_rows+1] (read during the call only).
 * d_rowptr/d_colidx: device CSR, must outlive the plan. Pass both NULL to build a
 * host-only plan (no GPU touched) — used by the CPU test-suite to check the chunking. */
int sagnn_spmm_plan_create(const int32_t* h_rowptr, const int32_t* d_rowptr,
                           const int32_t* d_colidx, int64_t n_rows, int64_t n_src,
                           int64_t nnz, const sagnn_spmm_tuning* tuning /* nullable */,
                           sagnn_spmm_plan** plan_out);
int sagnn_spmm_plan_destroy(sagnn_spmm_plan* plan);
int sagnn_spmm_plan_get_info(const sagnn_spmm_plan* plan, sagnn_spmm_plan_info* info);
/* Copies the chunk list to host arrays of capacity `cap` entries each (cap >= n_chunks).
 * chunk i covers edges [e_begin[i], e_end[i]) of row rows[i]; chunks of one row are
 * consecutive and in edge order. */
int sagnn_spmm_plan_copy_chunks(const sagnn_spmm_plan* plan, int32_t* rows, int32_t* e_begin,
                                int32_t* e_end, int64_t cap);
/* Bytes of device workspace sagnn_spmm_f32 needs for feature width d (0 if no long rows). */
size_t sagnn_spmm_workspace_bytes(const sagnn_spmm_plan* plan, int d);

/* ------------------------------------------------------------------------------------
 * sagnn_spmm_f32 — replaces Recommender.messagePropagate (model.py:80-92) together with
 * the residual add (model.py:124-125) and the running tf.add_n (model.py:126-127):
 *
 *     s[r,:]   = sum over edges (r,c) of X[c,:]          GatherV2 + SegmentSum (:86-87);
 *                                                        rows without edges give 0 (:87-91)
 *     y[r,:]   = max(leaky*s, s) + residual[r,:]         Activate 'leakyRelu' (NNLayers.py:136)
 *                                                        + embs[-1]          (model.py:124)
 *     out[r,:]     = y            (if out     != NULL)
 *     acc_out[r,:] = acc_in + y   (if acc_out != NULL; acc_in NULL means 0)
 *
 * Edge values are ignored, as in the reference (model.py:84, :86). residual may be NULL.
 * acc_in may alias residual and may alias acc_out (in-place running sum). out/acc_out
 * must not alias X. At least one of out/acc_out is required.
 * -------------------------------------------------------------------------------- */
int sagnn_spmm_f32(const sagnn_spmm_plan* plan, const float* X, int64_t ldx, int d,
                   const float* residual, int64_t ldr, float leaky, float* out, int64_t ldo,
                   const float* acc_in, int64_t ld_acc_in, float* acc_out, int64_t ld_acc_out,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Extended epilogue for training (sagnn_spmm_ex_f32); zero-initialise and fill what is used.
 *   s = A·X;  y = max(leaky*s, s) + residual;  out = y;  acc_out = acc_in + y
 *   mask_out [n_rows, d/4] bytes: bit j of byte l is 1 iff the activation passed s[4l+j] through
 *       with slope 1 (tf.maximum(leaky*x, x) sends the gradient to its FIRST argument on ties,
 *       so x = 0 counts as slope `leaky`: reference Utils/NNLayers.py:136)
 *   out2 = v * (mask_in bit ? 1 : slope2), v = the accumulated value if acc_out is given, else y
 *       (what the next backward step gathers). */
typedef struct sagnn_spmm_epilogue {
  float leaky;
  const float* residual; int64_t ldr;
  float* out; int64_t ldo;
  const float* acc_in; int64_t ld_acc_in;
  float* acc_out; int64_t ld_acc_out;
  uint8_t* mask_out;
  const uint8_t* mask_in;
  float* out2; int64_t ldo2; float slope2;
  const float* acc_in2; int64_t ld_acc_in2;   /* optional second addend: acc_out = acc_in + acc_in2 + y */
} sagnn_spmm_epilogue;

int sagnn_spmm_ex_f32(const sagnn_spmm_plan* plan, const float* X, int64_t ldx, int d,
                      const sagnn_spmm_epilogue* epilogue, void* workspace, size_t workspace_bytes,
                      void* stream);
/* out[r, :] = g[r, :] * (mask bit ? 1 : slope), mask [n_rows, d/4] bytes as sagnn_spmm_ex_f32 records them: the seed of the
 * backward chain of the interval stack, for hosts that run the chain on row slices themselves (T < world sharding). */
int sagnn_mask_scale_f32(const float* g, int64_t ldg, const uint8_t* mask, float slope, float* out, int64_t ldo,
                         int64_t n_rows, int d, void* stream);

/* ------------------------------------------------------------------------------------
 * sagnn_gnn_interval_f32 — one iteration k of the loop model.py:118-129 (L layers, both
 * directions, simultaneous update, residuals, add_n), 2*L SpMM launches issued from C.
 *
 *   e_u^0 = u0, e_i^0 = i0
 *   e_u^{l+1} = leaky(A   e_i^l) + e_u^l        plan_user: rows = users, cols = items
 *   e_i^{l+1} = leaky(A^T e_u^l) + e_i^l        plan_item: rows = items, cols = users
 *   user_out = sum_{l=0..L} e_u^l ;  item_out = sum_{l=0..L} e_i^l
 *
 * scratch_u: [2, U, d] floats, scratch_i: [2, I, d] floats (ping-pong layer outputs; may be
 * NULL when n_layers <= 1). user_out/item_out are written with row strides ld_uo/ld_io so
 * the caller can target row k of an [N, T, d] slab directly (replaces tf.stack +
 * tf.transpose, model.py:131-134).
 * -------------------------------------------------------------------------------- */
int sagnn_gnn_interval_f32(const sagnn_spmm_plan* plan_user, const sagnn_spmm_plan* plan_item,
                           const float* u0, int64_t ld_u0, const float* i0, int64_t ld_i0,
                           int d, int n_layers, float leaky, float* scratch_u, float* scratch_i,
                           float* user_out, int64_t ld_uo, float* item_out, int64_t ld_io,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Training form of sagnn_gnn_interval_f32: additionally records the activation masks of every
 * layer, mask_u [n_layers, U, d/4] and mask_i [n_layers, I, d/4] bytes (both or neither). */
int sagnn_gnn_interval_ex_f32(const sagnn_spmm_plan* plan_user, const sagnn_spmm_plan* plan_item,
                              const float* u0, int64_t ld_u0, const float* i0, int64_t ld_i0, int d,
                              int n_layers, float leaky, float* scratch_u, float* scratch_i,
                              float* user_out, int64_t ld_uo, float* item_out, int64_t ld_io,
                              uint8_t* mask_u, uint8_t* mask_i, void* workspace,
                              size_t workspace_bytes, void* stream);

/* Backward of the interval stack (what tf.gradients builds for model.py:118-129): given
 * G_u = dL/d user_out, G_i = dL/d item_out and the recorded masks, writes dL/d u0 and dL/d i0.
 * Same SpMM kernel with the roles of the two adjacencies swapped (the reference already holds
 * both, model.py:234-236). scratch_u: [4, U, d] floats, scratch_i: [4, I, d] floats.
 * ADJOINT CONTRACT: plan_user (rows = users) must be the exact transpose, multiplicities included,
 * of the item-side pattern the forward pass used, and plan_item that of the user-side pattern.
 * For matrices without duplicated stored entries these are the forward plans themselves; with
 * duplicates (forward counts one twice, DataHandler.transpose merges it: DataHandler.py:9-11) the
 * caller passes plans of the exact transposes — the library cannot tell the two cases apart from
 * the handles, so the host side checks it (sa-gnn_amd/graph.py interval_pair, ops.gnn_interval_bwd). */
int sagnn_gnn_interval_bwd_f32(const sagnn_spmm_plan* plan_user, const sagnn_spmm_plan* plan_item,
                               const float* G_u, int64_t ld_gu, const float* G_i, int64_t ld_gi, int d,
                               int n_layers, float leaky, const uint8_t* mask_u, const uint8_t* mask_i,
                               float* scratch_u, float* scratch_i, float* grad_u0, int64_t ld_du,
                               float* grad_i0, int64_t ld_di, void* workspace, size_t workspace_bytes,
                               void* stream);

/* ------------------------------------------------------------------------------------
 * The WHOLE loop over k of model.py:118-129 in one launch per layer. The reference's loop body is independent per
 * interval, so a layer of the stack is 2 T independent SpMMs (T with rows = users, T with rows = items); on
 * dataset-sized graphs (tens of thousands of rows) each is a launch of a few microseconds and the stack is bound by
 * its 2 T L launches (+ as many fix-ups). A batch object ties the T interval plans together: a block of the batched
 * kernel finds (direction, interval, row block) by division, every per-interval operand is a SLAB — interval k's
 * [N, d] matrix starts `slab` elements after interval k-1's, rows `ld` apart — so u0 [T, U, d] is (ld = d,
 * slab = U d) and a column of the [N, T, d] tensor the fusion reads is (ld = T d, slab = d). One row launch and (if
 * any interval has long rows) ONE fix-up launch per layer: Amazon-shaped T = 5, L = 3: 60 launches -> 6.
 *
 * sagnn_spmm_batch_create keeps the plans' device CSR pointers: the plans must outlive the batch. Every interval must
 * have the same user / item counts. Workspace: sagnn_spmm_batch_workspace_bytes (partial sums of all long-row chunks).
 * sagnn_gnn_stack_f32: sagnn_gnn_interval_ex_f32 for all intervals; scratch_u [2, T, U, d], scratch_i [2, T, I, d]
 *   (NULL when n_layers <= 1); mask_u [T, n_layers, U, d/4] / mask_i [T, n_layers, I, d/4] bytes (both or neither).
 * sagnn_gnn_stack_bwd_f32: sagnn_gnn_interval_bwd_f32 for all intervals; scratch_u [4, T, U, d], scratch_i [4, T, I, d];
 *   `batch` must tie the ADJOINT plans (the adjoint contract above, per interval).
 * -------------------------------------------------------------------------------- */
typedef struct sagnn_spmm_batch sagnn_spmm_batch;
int sagnn_spmm_batch_create(const sagnn_spmm_plan* const* plans_user, const sagnn_spmm_plan* const* plans_item,
                            int n_intervals, sagnn_spmm_batch** batch_out);
int sagnn_spmm_batch_destroy(sagnn_spmm_batch* batch);
size_t sagnn_spmm_batch_workspace_bytes(const sagnn_spmm_batch* batch, int d);
int sagnn_gnn_stack_f32(const sagnn_spmm_batch* batch, const float* u0, int64_t ld_u0, int64_t slab_u0,
                        const float* i0, int64_t ld_i0, int64_t slab_i0, int d, int n_layers, float leaky,
                        float* scratch_u, float* scratch_i, float* user_out, int64_t ld_uo, int64_t slab_uo,
                        float* item_out, int64_t ld_io, int64_t slab_io, uint8_t* mask_u, uint8_t* mask_i,
                        void* workspace, size_t workspace_bytes, void* stream);
int sagnn_gnn_stack_bwd_f32(const sagnn_spmm_batch* batch, const float* G_u, int64_t ld_gu, int64_t slab_gu,
                            const float* G_i, int64_t ld_gi, int64_t slab_gi, int d, int n_layers, float leaky,
                            const uint8_t* mask_u, const uint8_t* mask_i, float* scratch_u, float* scratch_i,
                            float* grad_u0, int64_t ld_du, int64_t slab_du, float* grad_i0, int64_t ld_di,
                            int64_t slab_di, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Interval fusion (model.py:135-155). x[node, interval, :] is read at
 * x + node*ld_n + interval*ld_t (elements): [n, t, d] storage is ld_t = d, ld_n >= t*d (what
 * tf.stack + tf.transpose produce, model.py:131-134); [t, n, d] storage is ld_n = d,
 * ld_t >= n*d (what the interval-sharded exchange delivers). Outputs h / y are [n, t, d] with
 * node stride ld_h / ld_y and the t*d block of a node dense.
 *
 * sagnn_lstm_fwd_f32 — dynamic_rnn(MultiRNNCell([DropoutWrapper(BasicLSTMCell(d))]))
 *   (model.py:135-146) at keep probability 1: TF 1.14 BasicLSTMCell, kernel W [2d, 4d]
 *   row-major (rows 0..d-1 multiply x_t, rows d..2d-1 multiply h), bias b [4d], gate
 *   order i, j, f, o;  c' = c*sigmoid(f + forget_bias) + sigmoid(i)*tanh(j);
 *   h' = tanh(c')*sigmoid(o); zero initial state. Writes h for every step: [n, t, d].
 *   drop_scale (nullable) [n, t, d] multiplies the EMITTED h only (DropoutWrapper's
 *   output_keep_prob; the recurrent state is not dropped).
 *
 * sagnn_layernorm_td_f32 — tf.contrib.layers.layer_norm defaults (model.py:152-153):
 *   mean/variance over (t, d) jointly per node, gamma/beta [d], eps = 1e-12 inside rsqrt.
 *
 * sagnn_mhsa_mean_f32 — MultiHeadSelfAttention.attention (Utils/attention.py:55-78) with
 *   ScaledDotProductAttention (:35-45), then tf.reduce_mean(axis=1) (model.py:154-155):
 *   Q/K/V = x@W+b (W [d, d] row-major, in x out), heads of d_k = d/heads,
 *   scores = exp(Q K^T / sqrt(d_k)) (no max subtraction), attn = scores/(rowsum + 1e-8),
 *   context = attn V, mean over the t query positions -> out [n, d].
 *
 * sagnn_interval_fusion_f32 — the three stages back to back on `stream` with the LSTM
 *   output kept in a caller-provided workspace (sagnn_interval_fusion_workspace_bytes) and
 *   normalised in place; only out [n, d] is a result.
 *
 * ARITHMETIC of the GEMM-shaped stages (the gate product [x_t | h] W and the three dense layers)
 *   for d in {32, 64, 128}, 16 heads, under the default engine: fp32 in, fp32 out, evaluated on the 16-bit
 *   matrix cores over SPLIT operands: two round-to-nearest f16 pieces per value (v = v1 + v2'/4096), three piece
 *   products, fp32 accumulation. The split represents v to 2^-23 |v| inside a window — |v| < 32768 at the top, and an
 *   ABSOLUTE floor of 2^-37 at the bottom — so what is guaranteed is:
 *     - every operand goes through a range check on its way to the matrix cores: a tile that holds |v| >= 32768, or
 *       an aligned 4-element segment that is non-zero but below 2^-18 as a whole (an input row of 1e-9), is
 *       re-evaluated in the kernel with fp32 fmaf chains. Such inputs get exactly an fp32 evaluation.
 *     - on the fast path each operand element is within max(2^-23 |v|, 2^-37) of its fp32 value, i.e. within
 *       2^-19 of its segment's largest element at worst and 2^-23 for segments above 2^-14 (every layer-normed row,
 *       every embedding sum): as close to the float64 product as an fp32 fmaf chain for such rows, not bit-identical
 *       to one.
 *     - gradients have no natural scale, so the attention-backward tail (sagnn_attn_bwd_tail_f32) scales every row
 *       of dQ|dK|dV by an exact power of two before the split: dy is accurate per ROW (relative to that row's own
 *       largest gradient, from 1e-38 to 1e38), dW / db relative to the sum of the magnitudes of their terms.
 *   sagnn_set_engine(SAGNN_ENGINE_F32) selects the exact-fp32 kernels instead. Results are deterministic run to run
 *   under every engine.
 * -------------------------------------------------------------------------------- */
int sagnn_lstm_fwd_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, const float* W,
                       const float* b, float forget_bias, const float* drop_scale, float* h,
                       int64_t ld_h, void* stream);
/* The same recurrence continued from a given state: h_init [n, d] (row stride ld_hi) and c_init
 * [n, d] (both NULL = zero state, i.e. sagnn_lstm_fwd_f32), c_final [n, d] receives the cell state
 * after the last step (NULL = not wanted). A sequence cut into consecutive calls is bit-identical
 * to one call; the multi-GPU pipeline uses this to run the steps of the intervals that have
 * already arrived while the last exchange round is still in flight. */
int sagnn_lstm_fwd_state_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d,
                             const float* W, const float* b, float forget_bias, const float* drop_scale,
                             const float* h_init, int64_t ld_hi, const float* c_init, float* h, int64_t ld_h,
                             float* c_final, void* stream);
int sagnn_layernorm_td_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d,
                           const float* gamma, const float* beta, float eps, float* y,
                           int64_t ld_y, void* stream);
int sagnn_mhsa_mean_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads,
                        const float* Wq, const float* bq, const float* Wk, const float* bk,
                        const float* Wv, const float* bv, float* out, int64_t ld_out,
                        void* stream);
/* "Wide" attention for any d that is a multiple of 32 (e.g. the MovieLens configuration, d = 128,
 * whose three [d, d] weights and Q|K|V tiles do not fit LDS together): Q|K|V by MFMA products
 * (sagnn_dense_nn_f32's kernel) into caller scratch [n, t, 3d], then a per-node attention kernel.
 * sagnn_interval_fusion_f32 and the Python wrappers route here for d other than 32 / 64 / 128 (d = 128, 16 heads
 * runs the split-operand kernels: attention over two column halves, the LSTM as one launch per step). */
size_t sagnn_mhsa_wide_workspace_bytes(int64_t n, int t, int d);
int sagnn_mhsa_mean_wide_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads,
                             const float* Wq, const float* bq, const float* Wk, const float* bk, const float* Wv,
                             const float* bv, float* out, int64_t ld_out, void* workspace, size_t workspace_bytes,
                             void* stream);
/* layer_norm over (T, d) + attention + mean without the LSTM (model.py:152-155): what the
 * training forward calls after sagnn_lstm_fwd_train_f32. Workspace (bytes from
 * sagnn_ln_mhsa_mean_workspace_bytes, 0 on the fused matrix-core path) holds the normalised
 * tensor where the normalisation cannot ride on the attention kernel's operand. */
size_t sagnn_ln_mhsa_mean_workspace_bytes(int64_t n, int t, int d, int heads);
int sagnn_ln_mhsa_mean_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads,
                           const float* ln_gamma, const float* ln_beta, float ln_eps, const float* Wq,
                           const float* bq, const float* Wk, const float* bk, const float* Wv, const float* bv,
                           float* out, int64_t ld_out, void* workspace, size_t workspace_bytes, void* stream);
int sagnn_interval_fusion_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads,
                              const float* lstm_W, const float* lstm_b, float forget_bias,
                              const float* ln_gamma, const float* ln_beta, float ln_eps,
                              const float* Wq, const float* bq, const float* Wk, const float* bk,
                              const float* Wv, const float* bv, float* out, int64_t ld_out,
                              void* workspace, size_t workspace_bytes, void* stream);
size_t sagnn_interval_fusion_workspace_bytes(int64_t n, int t, int d);

/* ------------------------------------------------------------------------------------
 * Backward of the interval fusion: the gradients tf.gradients derives for
 * model.py:135-155. The host (sa-gnn_amd/autograd.py) sequences these entries with the dense
 * products below. Any d that is a multiple of 32 trains through the per-step entries and the dense products
 * (d = 128 is checked against the float64 oracle); the fused entries — attention-backward front and tail, the
 * one-launch BPTT — say which d they cover through their *_supported queries. d_k a power of two.
 *
 * sagnn_lstm_fwd_train_f32 — sagnn_lstm_fwd_f32 that also stores the gate activations
 *   gates [n, t, 4d] = sigmoid(i) | tanh(j) | sigmoid(f + forget_bias) | sigmoid(o) and the cell
 *   state cell [n, t, d].
 * sagnn_attn_bwd_f32 — qkv [n, t, 3d] (Q | K | V rows, as x@W+b produced them) is overwritten
 *   with dQ | dK | dV given g_out = dL/d(mean-over-queries context) [n, d].
 * sagnn_layernorm_td_bwd_f32 — dy -> dh (may alias dy), dgamma/dbeta accumulated with atomics
 *   (zero them first).
 * sagnn_lstm_bwd_step_f32 — step ts of BPTT: dh_ext [n, t, d] (gradient arriving at the emitted
 *   h, scaled by drop_scale if given), dh_rec [n, d] (recurrent gradient, NULL at the last step),
 *   dc_in [n, d] (NULL at the last step) -> dgates [n, 4d] (pre-activation gradients, i|j|f|o)
 *   and dc_out [n, d].
 * sagnn_lstm_bwd_f32 — the whole BPTT in one launch (d in {32, 64}; sagnn_lstm_bwd_supported):
 *   x as given to the forward, h/gates/cell as sagnn_lstm_fwd_train_f32 stored them (h un-dropped,
 *   [n, t, d] dense), dh_ext [n, t, d] with row stride ld_dhe -> dx [n, t, d] dense, and
 *   dW [2d, 4d] / db [4d] ACCUMULATED with float atomics (zero them first). Gate gradients stay
 *   on chip: per 32-row chunk and step, d[x|h] = dG W^T and dW += [x|h]^T dG run as MFMA tiles.
 * -------------------------------------------------------------------------------- */
int sagnn_lstm_fwd_train_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d,
                             const float* W, const float* b, float forget_bias, const float* drop_scale,
                             float* h, int64_t ld_h, float* gates, float* cell, void* stream);
int sagnn_attn_bwd_f32(float* qkv, const float* g_out, int64_t ld_g, int64_t n, int t, int d, int heads,
                       void* stream);
int sagnn_layernorm_td_bwd_f32(const float* h, int64_t ld_h, const float* dy, int64_t ld_dy, int64_t n, int t,
                               int d, const float* gamma, float eps, float* dh, int64_t ld_dh, float* dgamma,
                               float* dbeta, void* stream);
int sagnn_lstm_bwd_step_f32(const float* gates, const float* cell, const float* dh_ext, int64_t ld_dhe,
                            const float* drop_scale, const float* dh_rec, int64_t ld_dhr, const float* dc_in,
                            float* dgates, float* dc_out, int64_t n, int t, int d, int ts, void* stream);
/* Front of the attention backward pass in one launch (sagnn_attn_bwd_front_supported: d in {32, 64},
 * d_k in {2, 4}, t in {1..6, 8}; on the default engine with 16 heads also t in {12, 16}, and d = 128 with t <= 6): y = layer_norm(x) when apply_ln (else x), Q|K|V = y W + b, attention
 * backward given g_out = dL/d(mean context) [n, d] -> dqkv [n*t, 3d] (dQ | dK | dV rows) and, when
 * y_out is not NULL, y [n*t, d]. Replaces layernorm_td + dense_nn + attn_bwd of the recompute path. */
int sagnn_attn_bwd_front_supported(int d, int t, int heads);
int sagnn_attn_bwd_front_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads,
                             const float* ln_gamma, const float* ln_beta, float ln_eps, int apply_ln,
                             const float* Wq, const float* bq, const float* Wk, const float* bk, const float* Wv,
                             const float* bv, const float* g_out, int64_t ld_g, float* dqkv, float* y_out,
                             void* stream);
/* Tail of the attention backward pass in one pass over dQ|dK|dV (d in {32, 64}): y [rows, d] is
 * OVERWRITTEN with dy = dqkv Wqkv^T; dWqkv [d, 3d] += y^T dqkv and dbqkv [3d] += column sums of dqkv
 * are accumulated with float atomics (zero them first). Wqkv = [Wq | Wk | Wv] as [d, 3d]. Replaces
 * sagnn_dense_tn_f32 + sagnn_dense_nn_f32 on the same operands (dqkv read once instead of twice). */
int sagnn_attn_bwd_tail_supported(int d);
int sagnn_attn_bwd_tail_f32(float* y, const float* dqkv, int64_t rows, int d, const float* Wqkv, float* dWqkv,
                            float* dbqkv, void* stream);
int sagnn_lstm_bwd_supported(int d);
int sagnn_lstm_bwd_f32(const float* x, int64_t ld_n, int64_t ld_t, const float* h, const float* gates,
                       const float* cell, const float* dh_ext, int64_t ld_dhe, const float* drop_scale,
                       const float* W, float* dx, float* dW, float* db, int64_t n, int t, int d, void* stream);
/* sagnn_lstm_bwd_f32 with the weight gradient as a SECOND PASS on the default engine: the BPTT launch leaves out its
 * dW product (fp32 MFMA: the larger part of its time) and stores the gate gradients time-major into the caller's
 * scratch ([t, n, 4d] floats = sagnn_lstm_bwd_workspace_bytes, 16-byte aligned); one pass over them then takes
 * dW += [x_s | h_{s-1}]^T dG_s on the 16-bit matrix cores over split operands, gate-gradient rows scaled by exact
 * powers of two as in the attention tail (ARITHMETIC above). Same arguments and results otherwise; with
 * workspace == NULL, or under SAGNN_ENGINE_F32 / SAGNN_ENGINE_VALU, the call IS sagnn_lstm_bwd_f32. */
size_t sagnn_lstm_bwd_workspace_bytes(int64_t n, int t, int d);
int sagnn_lstm_bwd_ws_f32(const float* x, int64_t ld_n, int64_t ld_t, const float* h, const float* gates,
                          const float* cell, const float* dh_ext, int64_t ld_dhe, const float* drop_scale,
                          const float* W, float* dx, float* dW, float* db, int64_t n, int t, int d, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Prediction head (reference model.py:156-173). The masked sum of item /
 * position embeddings (model.py:161-162) is sagnn_spmm_f32 on a per-batch CSR (row = batch slot,
 * columns = the unmasked sequence entries, leaky = 1); layer_norm and the length-1 MHSA are
 * sagnn_layernorm_td_f32 / sagnn_mhsa_mean_f32 with t = 1; what is left:
 *   sagnn_leaky_add_f32:  out = max(leaky*a, a) + b      (model.py:166; b nullable)
 *   sagnn_pair_score_f32: preds[e] = <U[uids[e]], I[iids[e]]> + <leaky(S[locs[e]]), A[iids[e]]>
 *                         (model.py:169-173; S/A/locs NULL drops the second term)
 * -------------------------------------------------------------------------------- */
int sagnn_leaky_add_f32(const float* a, const float* b, float* out, float leaky, int64_t count, void* stream);
int sagnn_pair_score_f32(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* S, int64_t lds,
                         const float* A, int64_t lda, const int32_t* uids, const int32_t* iids,
                         const int32_t* locs, float leaky, float* out, int64_t n_pairs, int d, void* stream);

/* ------------------------------------------------------------------------------------
 * Training-side operators (reference model.py:169-205, 241-250). All scatter
 * outputs (dU, dI, dS, dA, dX, dY, dF, dV, dw3, db3, loss) ACCUMULATE with float atomics: zero
 * them first. Dense gradient rows are [rows, d] contiguous.
 *   sagnn_pair_score_bwd_f32      backward of sagnn_pair_score_f32 given g [n_pairs]
 *   sagnn_prod_leaky_sum_f32      s[e] = sum_j leaky(X[uids[e]][j] * Y[iids[e]][j])   (model.py:191,199)
 *   sagnn_prod_leaky_sum_bwd_f32  its backward
 *   sagnn_meta_features_f32       m[e] = [F[u]*V[u] | F[u] | V[u]], u = uids[e]      (model.py:179)
 *   sagnn_meta_features_bwd_f32   its backward
 *   sagnn_leaky_f32               backward == 0: out = max(leaky*a, a); else out = g * slope(a)
 *   sagnn_rowdot_sigmoid_f32      w[e] = sigmoid(<A[e, :k], w3> + b3)                 (model.py:182)
 *   sagnn_rowdot_sigmoid_bwd_f32  dA[e, :k] = dz*w3, dw3 += sum dz*A[e], db3 += sum dz, dz = dw*w*(1-w)
 *   sagnn_hinge_f32               loss += scale * sum max(0, 1 - S*(pos - neg)), S = wp*sp - wn*sn
 *                                 (S = 1 when wp is NULL: model.py:244; weighted: model.py:196,202);
 *                                 writes d(loss)/d pos, neg, wp, wn (each nullable)
 * -------------------------------------------------------------------------------- */
int sagnn_pair_score_bwd_f32(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* S, int64_t lds,
                             const float* A, int64_t lda, const int32_t* uids, const int32_t* iids,
                             const int32_t* locs, float leaky, const float* g, float* dU, float* dI, float* dS,
                             float* dA, int64_t n_pairs, int d, void* stream);
int sagnn_prod_leaky_sum_f32(const float* X, int64_t ldx, const float* Y, int64_t ldy, const int32_t* uids,
                             const int32_t* iids, float leaky, float* out, int64_t n_pairs, int d, void* stream);
int sagnn_prod_leaky_sum_bwd_f32(const float* X, int64_t ldx, const float* Y, int64_t ldy, const int32_t* uids,
                                 const int32_t* iids, float leaky, const float* g, float* dX, float* dY,
                                 int64_t n_pairs, int d, void* stream);
int sagnn_meta_features_f32(const float* F, int64_t ldf, const float* V, int64_t ldv, const int32_t* uids,
                            float* out, int64_t n, int d, void* stream);
int sagnn_meta_features_bwd_f32(const float* F, int64_t ldf, const float* V, int64_t ldv, const int32_t* uids,
                                const float* dm, float* dF, float* dV, int64_t n, int d, void* stream);
int sagnn_leaky_f32(const float* a, const float* g, float* out, float leaky, int64_t count, int backward,
                    void* stream);
int sagnn_rowdot_sigmoid_f32(const float* A, int64_t lda, const float* w3, const float* b3, float* out, int64_t n,
                             int k, void* stream);
int sagnn_rowdot_sigmoid_bwd_f32(const float* A, int64_t lda, const float* w3, const float* w, const float* dw,
                                 float* dA, int64_t ldda, float* dw3, float* db3, int64_t n, int k, void* stream);
int sagnn_hinge_f32(const float* pos, const float* neg, const float* wp, const float* wn, const float* sp,
                    const float* sn, float scale, float* loss, float* dpos, float* dneg, float* dwp, float* dwn,
                    int64_t n, void* stream);

/* out[i] = a[i] * b[i] (dropout scaling of the emitted LSTM output, model.py:139). */
int sagnn_mul_f32(const float* a, const float* b, float* out, int64_t count, void* stream);

/* One tf.train.AdamOptimizer step (model.py:248-250) with the L2 term of regLoss
 * (args.reg * Regularize(), model.py:245, Utils/NNLayers.py:159-175) folded into the gradient:
 *   g' = g + 2*l2*p;  m = b1*m + (1-b1)*g';  v = b2*v + (1-b2)*g'^2
 *   p -= lr*sqrt(1 - b2^step)/(1 - b1^step) * m / (sqrt(v) + eps)        (step counts from 1)
 * The caller applies the staircase decay lr = lr0 * decay^floor(step/decay_step) (model.py:249). */
int sagnn_adam_step_f32(float* param, const float* grad, float* m, float* v, int64_t count, float lr,
                        float beta1, float beta2, float eps, float l2, int64_t step, void* stream);

/* The same step for n_tensors parameter tensors in ONE launch (one per 48 tensors): host arrays of
 * device pointers / element counts / per-tensor l2. grads[i] == NULL means a zero gradient: TF's
 * minimize() differentiates loss + reg*Regularize() (model.py:245-250), so a registered tensor that
 * no forward op reads (timeEmbed, the dead [d,d] weights of model.py:81) still receives 2*l2*p and
 * decays under Adam. All pointers 16-byte aligned. */
int sagnn_adam_multi_f32(int n_tensors, float* const* params, const float* const* grads, float* const* m,
                         float* const* v, const int64_t* counts, const float* l2, float lr, float beta1,
                         float beta2, float eps, int64_t step, void* stream);

/* ------------------------------------------------------------------------------------
 * Dense products on the matrix cores (exact fp32), n rows huge, W small:
 *   sagnn_dense_nn_f32:  Y[n, dout] (+)= X[n, din] @ W[din, dout] + bias      (bias nullable;
 *       accumulate != 0 adds into Y). Replaces `inp @ W` of NNLayers.FC (Utils/NNLayers.py:108)
 *       and tf.layers.dense (Utils/attention.py:66-72) outside the fused kernels, and the
 *       input-gradient products of the backward pass.
 *   sagnn_dense_tn_f32:  dW[din, dout] += X[n, din]^T @ G[n, dout];  db[dout] += column sums of G
 *       (db nullable). The weight-gradient products; accumulates with float atomics, so zero
 *       dW/db first and expect run-to-run differences in the last bits.
 *   sagnn_dense_tn_seg_f32: the same sums over n_seg row SEGMENTS of seg_rows rows each, row i of segment s
 *       at X + s * x_seg + i * ldx and G + s * g_seg + i * ldg (strides in floats, multiples of 4;
 *       seg_rows * n_seg < 2^31): the weight gradient of a whole BPTT in one launch — x [n, t, d] in any
 *       node / interval strides against the stored gate gradients [t, n, 4d] — instead of one product per step.
 * din, dout: any multiples of 32. A W block that fits LDS stays resident there and the tall operand streams
 * past it; larger blocks (the d = 128 LSTM: [512, 256]) and the segmented form run a tiled GEMM
 * (dense_gemm.hip: 128 x 128 / 64 x 128 block tiles, split-K with atomics for the transposed form).
 * -------------------------------------------------------------------------------- */
int sagnn_dense_nn_f32(const float* X, int64_t ldx, int64_t n, int din, int dout, const float* W,
                       const float* bias, float* Y, int64_t ldy, int accumulate, void* stream);
int sagnn_dense_tn_f32(const float* X, int64_t ldx, const float* G, int64_t ldg, int64_t n, int din,
                       int dout, float* dW, float* db, void* stream);
int sagnn_dense_tn_seg_f32(const float* X, int64_t ldx, int64_t x_seg, const float* G, int64_t ldg, int64_t g_seg,
                           int64_t seg_rows, int n_seg, int din, int dout, float* dW, float* db, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAGNN_H */
