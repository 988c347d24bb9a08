/*
 * tfrec_amd -- C ABI of the MI355X-native matrix-factorization ranking path.
 *
 * One shared library (libtfrec_amd.so), plain pointers and sizes, no C++ or
 * framework types.  Every entry point returns 0 on success and a negative
 * TFREC_E* code on failure; tfrec_last_error() then holds a message for the
 * calling thread.  Nothing throws across this boundary.
 *
 * Conventions
 *   - "d_" parameters are device pointers on the context's GPU, "h_" are host
 *     pointers.  Tables are row-major float32: P [U,d], Q [I,d], b [I].
 *   - The caller owns every buffer it passes.  The context owns only its own
 *     workspace (grown on demand, or up front with tfrec_ctx_reserve) and its
 *     HIP stream handle (borrowed when one is passed in).
 *   - Work is enqueued on the context's stream and is asynchronous unless a host
 *     output is produced; tfrec_ctx_synchronize() waits and reports the sticky
 *     device-side error flags (e.g. an index out of range in a batch).
 *   - A context is not re-entrant: one thread at a time per context.
 *
 * Each declaration names the reference interface it replaces (file:line in the
 * reference tree).
 */
#ifndef TFREC_AMD_H
#define TFREC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFREC_ABI_VERSION 1

#define TFREC_OK 0
#define TFREC_EINVAL (-1)   /* bad argument (shape, alignment, null pointer)            */
#define TFREC_EHIP (-2)     /* a HIP runtime call failed                                */
#define TFREC_ENOMEM (-3)   /* device or host allocation failed                         */
#define TFREC_EINDEX (-4)   /* a user/item index in a batch was out of range            */
#define TFREC_EUNSUP (-5)   /* unsupported configuration (e.g. d too large, k too large) */

/* device-side sticky flags reported by tfrec_ctx_error_flags() */
#define TFREC_FLAG_BAD_INDEX 1u

typedef struct tfrec_ctx tfrec_ctx;

/* ---- library / context ------------------------------------------------------------------ */
/* Replaces nothing in the reference: errors there are Python exceptions
 * (evaluator/backend/cpp/apt_evaluate.pyx:21-22, utils/tools.py:21). */
const char *tfrec_last_error(void);
int tfrec_abi_version(void);

/* The reference's device context is the tf.Session made in run_model.py:14-16.
 * hip_stream may be NULL (the context then creates and owns a stream) or a hipStream_t
 * owned by the caller (e.g. torch's current stream). */
int tfrec_ctx_create(int device, void *hip_stream, tfrec_ctx **out);
int tfrec_ctx_destroy(tfrec_ctx *ctx);
int tfrec_ctx_set_stream(tfrec_ctx *ctx, void *hip_stream);
void *tfrec_ctx_stream(tfrec_ctx *ctx);
int tfrec_ctx_synchronize(tfrec_ctx *ctx);                    /* waits; TFREC_EINDEX if a flag is set */
int tfrec_ctx_error_flags(tfrec_ctx *ctx, uint32_t *h_flags, int clear);
/* Pre-size the workspace for training batches up to max_batch triples at factor size d and
 * evaluation of up to max_eval_users users at top-k k, so that later calls allocate nothing. */
int tfrec_ctx_reserve(tfrec_ctx *ctx, int64_t max_batch, int d, int64_t max_eval_users, int k);
int64_t tfrec_ctx_workspace_bytes(tfrec_ctx *ctx);
/* Phase timing (no counterpart in the reference beyond its wall-clock `timer` decorator,
 * utils/tools.py:27-36): when enabled, every training/evaluation call records HIP events on
 * the context's stream between its kernels.  tfrec_ctx_phase_times waits for the last call
 * and returns the duration in milliseconds of each phase of that call; tfrec_ctx_phase_name
 * names phase i ("index", "user_tile", "user_fixup", "item_tile", "item_fixup", ...). */
int tfrec_ctx_set_profiling(tfrec_ctx *ctx, int enable);
int tfrec_ctx_phase_times(tfrec_ctx *ctx, float *h_ms, int max_phases, int *n_phases);
const char *tfrec_ctx_phase_name(tfrec_ctx *ctx, int i);

/* ---- device memory helpers (for callers without their own allocator) --------------------- */
int tfrec_dev_malloc(tfrec_ctx *ctx, size_t bytes, void **d_ptr);
int tfrec_dev_free(tfrec_ctx *ctx, void *d_ptr);
int tfrec_memcpy_h2d(tfrec_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);   /* synchronous */
int tfrec_memcpy_d2h(tfrec_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);   /* synchronous */
int tfrec_memset_zero(tfrec_ctx *ctx, void *d_dst, size_t bytes);

/* ---- parameters --------------------------------------------------------------------------- */
/* model/base/MatrixFactorization.py:17-21: tables ~ U(lo, hi) (reference: -0.05, 0.05), bias 0.
 * Counter-based generator: element n gets the n-th value of the stream named by seed. */
int tfrec_fill_uniform(tfrec_ctx *ctx, float *d_x, int64_t n, float lo, float hi, uint64_t seed);
/* The same stream starting at element first_index: a shard of a table gets exactly the values
 * the whole table would get. */
int tfrec_fill_uniform_at(tfrec_ctx *ctx, float *d_x, int64_t n, float lo, float hi, uint64_t seed, int64_t first_index);
int tfrec_fill_const(tfrec_ctx *ctx, float *d_x, int64_t n, float value);

/* model/base/MatrixFactorization.py:26-32 (predict): out[t] = <P[users[t]], Q[items[t]]> + b[items[t]] */
int tfrec_mf_predict(tfrec_ctx *ctx, const float *d_P, const float *d_Q, const float *d_b,
                     int64_t U, int64_t I, int d,
                     const int32_t *d_users, const int32_t *d_items, int64_t n, float *d_out);

/* Stable sort of n (key, value) pairs of 32-bit words by the low key_bits bits of the key (higher
 * key bits must be zero); not in place.  No counterpart in the reference, whose duplicate row ids
 * are summed by TensorFlow's gather gradient (model/BPR.py:36-40 + tf.train optimizers): this is
 * the index step every training entry point below runs first, exposed so callers and tests can
 * use and check it on its own. */
int tfrec_sort_pairs_u32(tfrec_ctx *ctx, const uint32_t *d_keys_in, const uint32_t *d_vals_in,
                         uint32_t *d_keys_out, uint32_t *d_vals_out, int64_t n, int key_bits);

/* ---- epoch pipeline on the device ------------------------------------------------------------ */
/* One training epoch's triples without the host: replaces BPR.get_training_data (model/BPR.py:50-62),
 * i.e. utils/tools.py:39-49 random_choice(all_items, size=len(pos), exclusion=pos) per user, and the
 * shuffle of data/dataiterator.py:52-54.  For every stored interaction t (CSR order, rows sorted
 * ascending) writes its user, its item and n_neg items drawn uniformly, with replacement, from the
 * items NOT in the user's row (rejection against the row).  The random numbers are a counter-based
 * hash of (seed, stream, t * n_neg + c, attempt) -- not numpy's stream, which cannot be drawn in
 * parallel -- so the output is a pure function of the arguments.  first_interaction is the position
 * of this matrix's first stored interaction in the whole training set (0 for the whole matrix): a
 * shard of user rows then draws exactly what the whole matrix draws for those rows.  d_users_out
 * (row numbers of THIS matrix) / d_pos_out may be null.  A user who holds every item raises the
 * bad-index flag (TFREC_EINDEX at synchronize). */
int tfrec_sample_negatives(tfrec_ctx *ctx, const int64_t *d_indptr, const int32_t *d_indices,
                           int64_t U, int64_t I, int64_t nnz, int n_neg, uint64_t seed, uint64_t stream,
                           int64_t first_interaction,
                           int32_t *d_users_out, int32_t *d_pos_out, int32_t *d_neg_out);
/* A permutation of 0..n-1 that is a pure function of (seed, stream, n): stable sort by 64-bit hash
 * keys, equal keys in storage order (the shuffle of data/dataiterator.py:52-54). */
int tfrec_shuffle_perm(tfrec_ctx *ctx, int64_t n, uint64_t seed, uint64_t stream, int32_t *d_perm_out);
/* One whole epoch of BPR triples, drawn and shuffled (model/BPR.py:50-62 + data/dataiterator.py:52-54 in one call):
 * exactly what tfrec_sample_negatives(n_neg = 1, stream = draw_stream), tfrec_shuffle_perm(nnz, seed, shuffle_stream)
 * and three tfrec_gather_rows_i32 produce, with the draws kept as one 16-byte record per interaction in between, so
 * that the shuffle moves every interaction with one read instead of three.  Outputs: int32 [nnz] each. */
int tfrec_epoch_triples(tfrec_ctx *ctx, const int64_t *d_indptr, const int32_t *d_indices,
                        int64_t U, int64_t I, int64_t nnz, uint64_t seed, uint64_t draw_stream, uint64_t shuffle_stream,
                        int64_t first_interaction,
                        int32_t *d_users_out, int32_t *d_pos_out, int32_t *d_neg_out);
/* dst[r, :] = src[perm[r], :] for rows of `width` int32 (applies a shuffle to index arrays). */
int tfrec_gather_rows_i32(tfrec_ctx *ctx, const int32_t *d_src, int64_t n_src, const int32_t *d_perm,
                          int64_t n, int width, int32_t *d_dst);

/* ---- training steps ------------------------------------------------------------------------ */
/* One BPR-MF minibatch step with plain SGD: the graph of model/BPR.py:36-48 (identical in
 * model/DNSBPR.py:35-47) executed by sess.run(self.update) at model/BPR.py:72.
 *   loss = sum_t -log_sigmoid(y(u,i)-y(u,j)+1e-7) + reg * l2(gathered rows and biases)
 * All gradients are taken at the pre-step parameters; a row that occurs n times in the batch
 * receives the sum of its n per-occurrence gradients (duplicates reduced without atomics, in a
 * fixed order: results are bitwise reproducible).  d_loss, if not NULL, receives two doubles:
 * the rank loss and the un-scaled l2 term. */
int tfrec_bpr_sgd_step(tfrec_ctx *ctx, float *d_P, float *d_Q, float *d_b,
                       int64_t U, int64_t I, int d,
                       const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B,
                       float lr, float reg, double *d_loss);

/* tfrec_bpr_sgd_step for callers that cannot allocate.  No counterpart in the reference: it is for code under
 * HIP graph capture, tight serving or training loops and harnesses that forbid hipMalloc.  Same arguments, same
 * kernels, same results bit for bit; but every byte of device scratch (sorted index arrays, sort counters, staged
 * rows, partial sums) comes from one array of 16 MiB that the library holds on each GPU it runs on until the
 * process ends, so the call allocates and frees nothing, on the device or on the host.  A batch whose scratch
 * does not fit (about 40,000 triples at d=64, 23,000 at d=128, 12,000 at d=256) is rejected with TFREC_EUNSUP
 * before anything is launched.  All calls on one GPU share that GPU's array: calls that overlap there (two
 * contexts or streams on one device) race; calls in stream order do not, and nothing needs clearing between
 * them.  Phase timing is off during the call.  The call borrows the context's workspace fields while it runs: the
 * context must not be used from another host thread until it returns (contexts are not re-entrant anyway). */
int tfrec_bpr_sgd_step_noalloc(tfrec_ctx *ctx, float *d_P, float *d_Q, float *d_b,
                               int64_t U, int64_t I, int d,
                               const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B,
                               float lr, float reg, double *d_loss);

/* The same step with its index work (range check, two radix sorts) taken off the critical path.
 * tfrec_batch_prepare sorts batch `slot` (0 or 1) on the context's auxiliary stream, beside
 * whatever the main stream is running; tfrec_bpr_sgd_step_prepared then runs the row passes of
 * that batch on the main stream and consumes the slot.  Typical loop: prepare(0, batch 0); for
 * s: prepare((s+1)&1, batch s+1); step_prepared(s&1).  The index arrays of a batch must stay
 * untouched until the step that consumes it has been enqueued.  Results are identical to
 * tfrec_bpr_sgd_step on the same batch. */
int tfrec_batch_prepare(tfrec_ctx *ctx, int slot, int64_t U, int64_t I,
                        const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B);
/* K consecutive batches (index arrays of K*B triples) sorted in ONE pass: the fixed costs of the
 * index phase are paid once per K steps.  The next K calls of tfrec_bpr_sgd_step_prepared on that
 * slot consume the batches in order. */
int tfrec_batch_prepare_many(tfrec_ctx *ctx, int slot, int64_t U, int64_t I,
                             const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B, int K);
int tfrec_bpr_sgd_step_prepared(tfrec_ctx *ctx, int slot, float *d_P, float *d_Q, float *d_b,
                                int64_t U, int64_t I, int d, float lr, float reg, double *d_loss);

/* ---- versioned user table -------------------------------------------------------------------
 * The steps above update P in place, yet the item half of a step needs the PRE-step user rows
 * (model/BPR.py:47-48: every gradient at the pre-step parameters), so they stage a copy of every
 * triple's user row: one row written and two read per triple, 45 % of the bytes a step moves.  A
 * tfrec_utable avoids the copies: it pairs the caller's table d_P with a second row buffer and one
 * version byte per user (both owned by the library).  A row's current value is in d_P when its byte is
 * 0, in the second buffer when it is 1; a step reads the current rows, writes the updated rows into
 * the other buffer, lets the item half read the old rows where they still are, and flips the bytes of
 * the batch's users at the end.  Same arithmetic in the same order: after tfrec_utable_flatten d_P holds
 * bit for bit what the plain steps leave there.  No counterpart in the reference (TensorFlow keeps its
 * variables behind the session too, run_model.py:14-16).
 *   tfrec_utable_create   wraps d_P [U, d] (caller-owned; its current contents are the table)
 *   tfrec_utable_flatten  copies the rows that live in the second buffer back: d_P is then complete
 *                         and current again (stream-ordered).  REQUIRED before anything else reads d_P
 *                         (evaluation, prediction, DNS selection, checkpoints, the plain steps).
 *   tfrec_utable_destroy  flattens, waits, frees.
 * The *_v steps take the table in place of (d_P, U, d) and are otherwise the entry points of the same
 * name without the suffix. */
typedef struct tfrec_utable tfrec_utable;
int tfrec_utable_create(tfrec_ctx *ctx, float *d_P, int64_t U, int d, tfrec_utable **out);
int tfrec_utable_flatten(tfrec_ctx *ctx, tfrec_utable *t);
int tfrec_utable_destroy(tfrec_ctx *ctx, tfrec_utable *t);
int tfrec_bpr_sgd_step_v(tfrec_ctx *ctx, tfrec_utable *t, float *d_Q, float *d_b, int64_t I,
                         const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B,
                         float lr, float reg, double *d_loss);
int tfrec_bpr_sgd_step_prepared_v(tfrec_ctx *ctx, int slot, tfrec_utable *t, float *d_Q, float *d_b, int64_t I,
                                  float lr, float reg, double *d_loss);

/* The two halves of a step on row-sharded tables (no counterpart in the reference, which is a
 * single process, run_model.py:14-16; the arithmetic is that of model/BPR.py:36-48).
 * tfrec_bpr_grad_step: like tfrec_bpr_sgd_step on P (complete update: the caller routes every
 * triple to the GPU that owns its user), but d_Qc/d_bc hold FETCHED COPIES of the Ic distinct item
 * rows the batch touches (items are addressed by their slot in that copy); on return each row of
 * d_Qc/d_bc holds the summed gradient of that item over this GPU's triples (including the l2
 * term of every occurrence) instead of an updated row.
 * tfrec_rows_sgd_apply: the owner side: Q[ids[t]] -= lr * grad_rows[t], b[ids[t]] -= lr *
 * grad_b[t], rows that share an id summed first, in the order they appear (so summing the
 * per-GPU sums in rank order is reproducible). */
int tfrec_bpr_grad_step(tfrec_ctx *ctx, float *d_P, float *d_Qc, float *d_bc,
                        int64_t U, int64_t Ic, int d,
                        const int32_t *d_users, const int32_t *d_pos_slot, const int32_t *d_neg_slot, int64_t B,
                        float lr, float reg, double *d_loss);
int tfrec_rows_sgd_apply(tfrec_ctx *ctx, float *d_Q, float *d_b, int64_t I, int d,
                         const int32_t *d_ids, const float *d_grad_rows, const float *d_grad_b, int64_t n, float lr);
/* tfrec_bpr_grad_step on a batch whose index arrays (local user rows, item slots) were sorted ahead with
 * tfrec_batch_prepare(slot, U, Ic, ...): the sort then runs beside the all-to-all that fetches the rows. */
int tfrec_bpr_grad_step_prepared(tfrec_ctx *ctx, int slot, float *d_P, float *d_Qc, float *d_bc,
                                 int64_t U, int64_t Ic, int d, float lr, float reg, double *d_loss);
/* The same on a versioned user table (see tfrec_utable): the user shard's pre-step rows are not staged. */
int tfrec_bpr_grad_step_prepared_v(tfrec_ctx *ctx, int slot, tfrec_utable *t, float *d_Qc, float *d_bc, int64_t Ic,
                                   float lr, float reg, double *d_loss);
/* The requester side of the lookup: the distinct items of a batch (d_pos / d_neg, global ids below I), ascending --
 * hence grouped by owner when owner r holds items [r * items_per_owner, (r + 1) * items_per_owner) -- in d_uniq
 * (room for min(2B, I) ids), the slot of every triple's items in that list (d_pos_slot, d_neg_slot), the number of
 * distinct items per owner (d_counts, int64 [n_owners]) and in all (d_n_uniq, optional).  Marks over the item
 * range and a prefix sum, no sort; everything stays on the device.
 * tfrec_rows_gather: the owner side: out_rows[t] = Q[ids[t]], out_b[t] = b[ids[t]] (ids local to the shard). */
int tfrec_items_dedup(tfrec_ctx *ctx, const int32_t *d_pos, const int32_t *d_neg, int64_t B, int64_t I,
                      int64_t items_per_owner, int n_owners, int32_t *d_uniq, int32_t *d_pos_slot, int32_t *d_neg_slot,
                      int64_t *d_counts, int64_t *d_n_uniq);
int tfrec_rows_gather(tfrec_ctx *ctx, const float *d_Q, const float *d_b, int64_t I, int d,
                      const int32_t *d_ids, int64_t n, float *d_out_rows, float *d_out_b);

/* The step of a GPU that holds a shard of P and a full copy of Q, b (small item tables): P is
 * updated in place; d_G = [I*d | I] floats is zeroed and receives, for every item the batch
 * touches, the summed gradient of its row and bias (regularisation included).  The caller sums
 * d_G over the GPUs (all-reduce) and applies tfrec_axpy(Qb, G, -lr) to every copy: together that
 * is sess.run(self.update) of model/BPR.py:72 on the union of the GPUs' batches.
 * slot -1 sorts the given index arrays now; slot 0/1 consumes a batch prepared with
 * tfrec_batch_prepare (the index arrays are then ignored and may be null). */
int tfrec_bpr_dense_grad_step(tfrec_ctx *ctx, int slot, float *d_P, const float *d_Q, const float *d_b, float *d_G,
                              int64_t U, int64_t I, int d,
                              const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B,
                              float lr, float reg, double *d_loss);
/* The same on a versioned user table (see tfrec_utable above). */
int tfrec_bpr_dense_grad_step_v(tfrec_ctx *ctx, int slot, tfrec_utable *t, const float *d_Q, const float *d_b, float *d_G,
                                int64_t I, const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B,
                                float lr, float reg, double *d_loss);
/* The same with the item half run in n_chunks pieces (1..32), one range of ceil(I / n_chunks) item ids after the
 * other: chunk_done[c] (HIP events the caller created on this device) is recorded on the context's stream when
 * rows [c * ceil(I / n_chunks), ...) of d_G are complete, so that the caller can start summing those rows over the
 * GPUs on another stream while the next ranges are still being computed.  The bias part of d_G is complete with
 * the last event.  Sums of an item are formed in a different association than in the unchunked call (last bits). */
int tfrec_bpr_dense_grad_step_chunked(tfrec_ctx *ctx, int slot, tfrec_utable *t, const float *d_Q, const float *d_b, float *d_G,
                                      int64_t I, const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B,
                                      float lr, float reg, double *d_loss, int n_chunks, void **chunk_done);
/* y[i] = fma(alpha, x[i], y[i]) for i < n: the plain SGD update (model/BPR.py:48) of a replicated
 * table from a summed gradient table, alpha = -learning rate. */
int tfrec_axpy(tfrec_ctx *ctx, float *d_y, const float *d_x, float alpha, int64_t n);

/* One APR minibatch step: both session calls of model/APR.py:163-164.
 *   1) delta <- eps * l2_normalize(d L_bpr / d Theta)   (model/APR.py:116-132; rows one by one,
 *      the bias gradient as one vector over all items)
 *   2) sparse Adagrad (accumulators start at 0.1 in the reference) on
 *      L + reg_adv * L(Theta+delta) + 2*reg*mean(p^2+q_i^2+q_j^2+b_i^2+b_j^2)  (model/APR.py:86-114,134-136)
 * d_dP/d_dQ/d_db are the delta tables of model/APR.py:55-60 (same shapes as P/Q/b); only the
 * rows of the batch are written and read.  d_loss (nullable) receives three doubles:
 * L, L_adv and the mean-square term. */
int tfrec_apr_adagrad_step(tfrec_ctx *ctx, float *d_P, float *d_Q, float *d_b,
                           float *d_accP, float *d_accQ, float *d_accb,
                           float *d_dP, float *d_dQ, float *d_db,
                           int64_t U, int64_t I, int d,
                           const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B,
                           float lr, float reg, float reg_adv, float eps, double *d_loss);

/* The APR step of a GPU that holds a shard of P (with its accumulator and delta table) and full
 * copies of Q, b, their accumulators and deltas.  Four calls and two all-reduces make one step:
 *   1) tfrec_apr_dense_delta_grads   delta_P rows of this GPU's users (model/APR.py:116-132); d_G =
 *      [I*d | I] zeroed and filled with the raw sums of d L_bpr / d(Q, b) over this GPU's triples
 *      -> all-reduce d_G
 *   2) tfrec_apr_delta_from_grads    delta_Q = eps * rows of G normalised one by one, delta_b = eps *
 *      the bias part of G normalised as one vector (model/APR.py:124-132; rows without gradient
 *      become zero, as the reference's full-table assignment makes them)
 *   3) tfrec_apr_dense_adagrad_grads Adagrad step on this GPU's rows of P (model/APR.py:134-136) and
 *      d_G = summed gradient of the optimised loss (model/APR.py:104-114) w.r.t. Q, b over this
 *      GPU's triples -> all-reduce d_G.  B_total = triples of the whole step over all GPUs (the
 *      mean-square term divides by B_total*d).  d_loss (optional): this GPU's share of the three
 *      loss terms, to be summed over the GPUs.  Must directly follow call 1 for the same batch on
 *      this context: it reuses that call's sorted index.
 *   4) tfrec_adagrad_apply           acc += g*g; w -= lr*g/sqrt(acc) wherever g != 0, on the flat
 *      [Q | b], [accQ | accb] buffers (the sparse Adagrad of the reference never visits entries
 *      without gradient). */
int tfrec_apr_dense_delta_grads(tfrec_ctx *ctx, const float *d_P, const float *d_Q, const float *d_b, float *d_dP, float *d_G,
                                int64_t U, int64_t I, int d,
                                const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B, float eps);
int tfrec_apr_delta_from_grads(tfrec_ctx *ctx, const float *d_G, float *d_dQ, float *d_db, int64_t I, int d, float eps);
int tfrec_apr_dense_adagrad_grads(tfrec_ctx *ctx, float *d_P, const float *d_Q, const float *d_b, float *d_accP,
                                  const float *d_dP, const float *d_dQ, const float *d_db, float *d_G,
                                  int64_t U, int64_t I, int d, int64_t B, int64_t B_total,
                                  float lr, float reg, float reg_adv, double *d_loss);
int tfrec_adagrad_apply(tfrec_ctx *ctx, float *d_w, float *d_acc, const float *d_g, float lr, int64_t n);

/* DNS-BPR candidate selection, model/DNSBPR.py:74-79: for positive t with user d_users[t] and
 * candidates d_cand[t*C .. t*C+C), out_neg[t] = the candidate with the largest score
 * <P[u],Q[c]> + b[c]; the first maximum wins.  Only the C sampled rows are read (the reference
 * scores all items per user, model/DNSBPR.py:68-70, and then indexes the pool). */
int tfrec_dns_select(tfrec_ctx *ctx, const float *d_P, const float *d_Q, const float *d_b,
                     int64_t U, int64_t I, int d,
                     const int32_t *d_users, const int32_t *d_cand, int64_t n_pos, int C,
                     int32_t *d_out_neg);

/* ---- evaluation ---------------------------------------------------------------------------- */
#define TFREC_SCORE_F32 0    /* exact float32: score = fma chain over f ascending, starting at b[i] */
#define TFREC_SCORE_BF16 1   /* operands rounded to bfloat16, float32 accumulation (MFMA)            */

/* Scoring + top-k in factor form, never forming the dense [users, items] matrix that
 * model/BPR.py:76-80 (_predict) builds and evaluator/backend/cpp/evaluate.py:25-27 masks:
 * for each listed user, score items [item_lo, item_hi) of Q/b (rows item_lo.. of the full
 * tables), skip the user's train items (CSR over all users, sorted rows; may be NULL) and
 * keep the k best under the total order (score descending, item id ascending).
 * Outputs [n,k]: ids are GLOBAL item ids; unused slots (fewer than k candidates in the shard)
 * hold id -1 and score -inf.  This is one item shard's work in a multi-GPU evaluation. */
int tfrec_score_topk(tfrec_ctx *ctx, const float *d_P, const float *d_Q, const float *d_b,
                     int64_t I, int d, int64_t item_lo, int64_t item_hi,
                     const int32_t *d_eval_users, int64_t n,
                     const int64_t *d_train_indptr, const int32_t *d_train_indices,
                     int k, int score_mode, int32_t *d_out_ids, float *d_out_scores);

/* The same for a GPU that holds only its own item shard (rows item_id_lo .. item_id_lo +
 * n_shard_items of the full tables, stored from row 0) and receives the evaluated users' factor
 * rows packed in list order: row r of d_P_rows belongs to user d_eval_users[r] (global id, used
 * for the train CSR).  Reported ids are global. */
int tfrec_score_topk_shard(tfrec_ctx *ctx, const float *d_P_rows, const float *d_Q_shard, const float *d_b_shard,
                           int64_t n_shard_items, int64_t item_id_lo, int d,
                           const int32_t *d_eval_users, int64_t n,
                           const int64_t *d_train_indptr, const int32_t *d_train_indices,
                           int k, int score_mode, int32_t *d_out_ids, float *d_out_scores);

/* Merge G per-shard top-k lists [G,n,k] into the global top-k [n,k] under the same total
 * order (the all-gather merge of an item-sharded evaluation). */
int tfrec_topk_merge(tfrec_ctx *ctx, const int32_t *d_ids, const float *d_scores, int G, int64_t n, int k,
                     int32_t *d_out_ids, float *d_out_scores);

/* If a user has fewer than k rankable (non-train) items the reference's ranking continues into
 * its -inf masked train items (evaluator/backend/cpp/evaluate.py:27 + tools.h:17); this fills
 * ids still -1 with that user's train items in ascending id order. */
int tfrec_topk_fill_masked(tfrec_ctx *ctx, int32_t *d_ids, const int32_t *d_eval_users, int64_t n, int k,
                           const int64_t *d_train_indptr, const int32_t *d_train_indices);

/* Replaces c_top_k_array_index (evaluator/backend/cpp/tools.h:22-30) on a dense row-major
 * [rows, cols] float32 matrix already on the device: ids of the k best of every row, total
 * order (score descending, id ascending).  If mask CSR is given, row r's listed columns are
 * treated as -inf (evaluator/backend/cpp/evaluate.py:25-27) without touching the matrix;
 * d_mask_rows maps matrix row -> CSR row (NULL = identity). */
int tfrec_topk_dense(tfrec_ctx *ctx, const float *d_ratings, int64_t rows, int64_t cols, int k,
                     const int32_t *d_mask_rows, const int64_t *d_mask_indptr, const int32_t *d_mask_indices,
                     int32_t *d_out_ids, float *d_out_scores);

/* Replaces evaluate (evaluator/backend/cpp/evaluate.h:13-93) and the five curves of
 * evaluator/backend/cpp/metrics.h:11-100: from rank lists [n,k] and the ground truth CSR
 * (row of user d_eval_users[r]; sorted rows) compute Precision/Recall/MAP/NDCG/MRR @1..k per
 * user and their mean over the n users.  h_mean [5,k] float32 (fully overwritten; no
 * pre-zeroing contract).  d_per_user [n,5,k] is optional (NULL to skip).
 * mean_mode 0: deterministic float64 tree sum on the device; 1: per-user rows are brought to
 * the host and summed sequentially in float32 in user order, which is bit-for-bit what
 * evaluate.h:40-92 does. */
int tfrec_rank_metrics(tfrec_ctx *ctx, const int32_t *d_ids, const int32_t *d_eval_users, int64_t n, int k,
                       const int64_t *d_test_indptr, const int32_t *d_test_indices,
                       int mean_mode, float *d_per_user, float *h_mean);
/* Same, but leaves the [5,k] float64 SUMS over the n users on the device (no host sync), for
 * callers that reduce across GPUs before dividing. */
int tfrec_rank_metrics_sums(tfrec_ctx *ctx, const int32_t *d_ids, const int32_t *d_eval_users, int64_t n, int k,
                            const int64_t *d_test_indptr, const int32_t *d_test_indices,
                            float *d_per_user, double *d_sums);

/* All of the above in one call for one GPU: what FoldOutEvaluator.evaluate
 * (evaluator/evaluator.py:29-32) returns, from the factors. */
int tfrec_eval_fullrank(tfrec_ctx *ctx, const float *d_P, const float *d_Q, const float *d_b,
                        int64_t I, int d, const int32_t *d_eval_users, int64_t n,
                        const int64_t *d_train_indptr, const int32_t *d_train_indices,
                        const int64_t *d_test_indptr, const int32_t *d_test_indices,
                        int k, int score_mode, int mean_mode,
                        int32_t *d_out_ids /* [n,k] or NULL */, float *h_mean /* [5,k] */);

#ifdef __cplusplus
}
#endif
#endif /* TFREC_AMD_H */
