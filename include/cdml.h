/* cdml.h -- C ABI of libcdml_hip.so: the MI355X (gfx950) triplet-embedding hot
 * path of the CDML video recommender.
 *
 * This is the drop-in boundary for ONE path of the reference
 * (geekieo/collaborative-deep-metric-learning, citations are file:line in it):
 *   sampler + feature gather   inputs.py:102-166, parse_data.py:292-320
 *   embedding tower (VNet)     models.py:19-62
 *   triplet hinge loss         losses.py:20-49
 *   backward + optimizer step  train.py:105-146
 *
 * Conventions
 *   - plain C: pointers and sizes only, no C++/torch types.  Every pointer is a
 *     DEVICE pointer (HBM) unless its name ends in _host.
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library
 *     allocates nothing and keeps no state between calls.
 *   - every entry point only ENQUEUES work on `stream` (a hipStream_t passed as
 *     void*); it never synchronises, so a sequence of calls can be captured into
 *     a hipGraph.
 *   - return value: CDML_OK or a negative cdml_status; cdml_last_error() gives
 *     the message for the calling thread.  No exception or exit() crosses the
 *     boundary (the reference swallows Python exceptions, train.py:331-332; the
 *     Python facade raises instead).
 *   - matrices are row-major fp32 with an explicit leading dimension (elements).
 *     GEMM operands must have leading dimensions that are multiples of 4 and
 *     16-byte aligned bases (CDML_E_ALIGN otherwise); padded columns must hold
 *     zeros where noted.
 *   - row order of a batch follows the reference: rows 3i,3i+1,3i+2 are the
 *     anchor, positive, negative of triplet i (train.py:313,128).
 */
#ifndef CDML_H_
#define CDML_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *cdml_stream_t; /* hipStream_t */

enum cdml_status {
  CDML_OK = 0,
  CDML_E_BADARG = -1,      /* null pointer, non-positive size, bad enum */
  CDML_E_HIP = -2,         /* a HIP runtime call / launch failed */
  CDML_E_ALIGN = -3,       /* base pointer or leading dimension misaligned */
  CDML_E_UNSUPPORTED = -4  /* shape outside what the kernels are built for */
};

/* ABI version (major*1000+minor) and the message of the last failure on this
 * thread ("" if none). */
int cdml_version(void);
const char *cdml_last_error(void);
/* "CDML_BUILD_ID=<16 hex digits>": sha256[:16] over the sources this library was compiled from (csrc/ .hip and .h
 * files by name, then this header) -- compiled in by __graft_entry__.build(); the host side
 * (_lib.load_library) compares it with the tree it runs from and refuses a library built from other sources.
 * The reference has no counterpart (pure Python, nothing prebuilt). */
const char *cdml_build_id(void);

/* ---- synthetic catalogue (bench/test data, imitation_data.py:41-53 shape) --
 * table[r][j] for r in [row0,row0+n_rows), j < feature_size: U[0,1) on the fp32
 * grid from Philox4x32-10 (ctr=(j>>2,row_lo,row_hi,TABLE_TAG), key=seed); columns
 * feature_size..row_stride-1 are zeroed.  Global row id r is stored at local row
 * r-row0. */
int cdml_fill_uniform_table(float *table, int64_t row0, int64_t n_rows,
                            int feature_size, int64_t row_stride, uint64_t seed,
                            cdml_stream_t stream);

/* ---- sampler: replaces MPTripletPipe.subprocess (inputs.py:102-142) and
 * yield_negative_index / combine_cowatch_neg (parse_data.py:292-320) ----------
 * pairs int32[n_pairs][2] co-watch (anchor,positive) video ids, read as one
 * sequential stream with wrap-around: slot s of step t is pair
 * (t*batch_global + s) mod n_pairs.  This call produces slots
 * [slot0, slot0+batch).  The step is `step` plus, if step_dev is non-NULL, the
 * counter read from device memory at execution (graph replay; `step` is then an
 * offset, e.g. 1 to sample the NEXT step while the optimizer has not advanced yet).
 * Uniform mode: negative ~ U{0..n_rows-1} from the counter-based stream
 * specified in oracle/sampler.py, redrawn while it equals the anchor or the
 * positive (inputs.py:125-127).  idx_out int32[batch][3] = (a,p,n). */
int cdml_sample_uniform(const int32_t *pairs, int64_t n_pairs, int64_t n_rows,
                        uint64_t seed, uint64_t step, const uint64_t *step_dev,
                        int batch, int64_t slot0, int64_t batch_global,
                        int32_t *idx_out, cdml_stream_t stream);

/* In-batch mode (build-defined, no reference counterpart): rows_out
 * int32[2*batch] = a0,p0,a1,p1,...; the negative of triplet i is the positive of
 * triplet (i+shift) mod batch with shift = 1 + U{0..batch-2} drawn per step;
 * shift_out int32[1]. */
int cdml_sample_inbatch(const int32_t *pairs, int64_t n_pairs, uint64_t seed,
                        uint64_t step, const uint64_t *step_dev, int batch,
                        int64_t slot0, int64_t batch_global, int32_t *rows_out,
                        int32_t *shift_out, cdml_stream_t stream);

/* *step_dev += 1 (enqueue; lets a captured graph advance the sampler). */
int cdml_step_advance(uint64_t *step_dev, cdml_stream_t stream);

/* ---- feature gather: replaces FEATURES[np.asarray(idx)] (inputs.py:158), the
 * reshape (train.py:313) and the feed_dict host->device copy (train.py:318),
 * fused with the tower's input tf.nn.l2_normalize (models.py:58) -------------
 * x_out[r][0:F] = table[idx[r]-row0][0:F] * (normalize ? rsqrt(max(sum sq,
 * 1e-12)) : 1), columns F..out_stride-1 zeroed.  table rows have stride
 * row_stride (>= F, multiple of 4; 16-B aligned base).  inv_norm_out may be
 * NULL.  idx must lie in [row0, row0+n_rows) -- out-of-range ids are clamped
 * and flagged in *oob_flag (int32, may be NULL); idx == -1 marks a padding slot
 * (fixed-capacity exchange): that output row is left untouched, nothing flagged.
 * `normalize` is a flag word: bit 0 = l2-normalise; bit 1 = idx == -1 means a
 * MISSING row (a request that found no slot in the exchange): the output row is
 * filled with all-ones words (NaN as fp32, NaN as bf16 pairs) so that the loss of
 * the same step shows it. */
int cdml_gather_rows(const float *table, int64_t row0, int64_t n_rows,
                     int64_t row_stride, const int32_t *idx, int n_idx, int F,
                     int normalize, float *x_out, int64_t out_stride,
                     float *inv_norm_out, int32_t *oob_flag,
                     cdml_stream_t stream);

/* Row exchange over a row-sharded catalogue (multi-GPU; the reference is single-GPU,
 * train.py:341-342 -- this fills its "distributed arguments" TODO).  Fixed capacity: every
 * rank sends each peer `capacity` id slots and receives `capacity` row slots, so the step path
 * needs no host-side counts (enqueue-only, hipGraph-capturable).
 * cdml_route_rows: request r (global id ids[r], owner = id / rows_per_shard) goes to slot
 * owner*capacity + k of send_ids (int32[world*capacity], unused slots = -1), k counting the
 * owner's requests in ascending r; slot_out[r] = that slot (-1 if the owner's segment is full).
 * *overflow_flag |= 1 on a full segment, |= 2 on an id outside the catalogue (check it off the
 * critical path; a flagged step fetched wrong rows).
 * cdml_scatter_rows: dst[slot[r]] = src[r] (slot -1 dropped): the way back, for row gradients. */
int cdml_route_rows(const int32_t *ids, int n, int64_t rows_per_shard, int world,
                    int capacity, int32_t *send_ids, int32_t *slot_out,
                    int32_t *overflow_flag, cdml_stream_t stream);
int cdml_scatter_rows(const float *src, int64_t ld_src, const int32_t *slot, int n,
                      int width, float *dst, int64_t ld_dst, cdml_stream_t stream);

/* Persistent fused sampler+gather for n_steps consecutive training steps (step, step+1, ...;
 * the sampler is counter-based, so later steps' triplets are known now and one launch can
 * fetch several steps' rows): samples slots [slot0,slot0+batch) of every step exactly as
 * cdml_sample_uniform (mode 0) / cdml_sample_inbatch (mode 1).  A block walks chunks of 8
 * rows; the chunk's ids are computed once, staged in LDS (and written to idx_out) while the
 * previous chunk's row loads are in flight; each wave gathers + l2-normalises two rows at a
 * time with whole-128-B-line loads.  Step s writes idx_out + s*idx_step_stride (int32[batch][3]
 * in mode 0, int32[2*batch] in mode 1), shift_out[s] (mode 1) and x_out + s*x_step_stride
 * (3*batch or 2*batch rows of out_stride floats).  n_steps = 1: the strides are unused.
 * oob_flag (may be NULL): bit 0 is OR-ed in when a pair id lies outside [0, n_rows) -- the
 * reference raises IndexError there (inputs.py:158); the row load itself clamps the id, as
 * cdml_gather_rows does. */
int cdml_sample_gather(int mode, const int32_t *pairs, int64_t n_pairs,
                       uint64_t seed, uint64_t step, const uint64_t *step_dev,
                       int batch, int64_t slot0, int64_t batch_global,
                       const float *table, int64_t n_rows, int64_t row_stride,
                       int F, int32_t *idx_out, int32_t *shift_out,
                       float *x_out, int64_t out_stride, int n_steps,
                       int64_t x_step_stride, int64_t idx_step_stride,
                       int32_t *oob_flag, cdml_stream_t stream);

/* ---- tower pieces: VNet.create_model (models.py:46-62) ----------------------
 * y = x * rsqrt(max(sum(x^2), 1e-12)) per row (tf.nn.l2_normalize, models.py:58,
 * 61).  inv_out[M] may be NULL. */
int cdml_l2norm_fwd(const float *x, int64_t ldx, int M, int N, float *y,
                    int64_t ldy, float *inv_out, cdml_stream_t stream);

/* dz = d/dz [ z*rsqrt(max(sum z^2,eps)) ] applied to g, optionally followed by
 * the leaky-relu derivative of the layer that produced z (z is the
 * post-activation; slope alpha where z <= 0).  lrelu_alpha < 0 disables it. */
int cdml_l2norm_bwd(const float *z, int64_t ldz, const float *g, int64_t ldg,
                    int M, int N, float lrelu_alpha, float *dz, int64_t lddz,
                    cdml_stream_t stream);

/* y[M][N] = leaky_relu(x[M][K] @ W[K][N] + b[N], alpha): slim.fully_connected
 * with tf.nn.leaky_relu (models.py:19-30,59-60).  W is [in,out] like slim.
 * K % 32 == 0 and N % 64 == 0 (pad with zeros); any M >= 1. */
int cdml_fc_lrelu_fwd(const float *x, int64_t ldx, const float *W, int64_t ldw,
                      const float *b, float alpha, int M, int K, int N,
                      float *y, int64_t ldy, cdml_stream_t stream);

/* dx[M][K] = (dy[M][N] @ W[K][N]^T) * lrelu'(x_post[M][K]) -- the data gradient
 * of an FC layer chained with the leaky-relu derivative of the PREVIOUS layer
 * (x_post = that layer's post-activation output = this layer's input).
 * x_post NULL => no mask.  N % 32 == 0, K % 64 == 0. */
int cdml_fc_bwd_data(const float *dy, int64_t lddy, const float *W, int64_t ldw,
                     const float *x_post, int64_t ldxp, float alpha, int M,
                     int K, int N, float *dx, int64_t lddx,
                     cdml_stream_t stream);

/* dW[K][N] = x[M][K]^T @ dy[M][N], db[N] = column sums of dy (db may be NULL).
 * Deterministic split-K: workspace must hold cdml_fc_bwd_weight_workspace()
 * bytes.  K % 64 == 0, N % 64 == 0; any M >= 1. */
size_t cdml_fc_bwd_weight_workspace(int M, int K, int N);
int cdml_fc_bwd_weight(const float *x, int64_t ldx, const float *dy,
                       int64_t lddy, int M, int K, int N, float *dW,
                       int64_t lddw, float *db, void *workspace,
                       size_t workspace_bytes, cdml_stream_t stream);

/* Both weight gradients of the two-layer tower in ONE launch (train.py:141):
 *   dW1[K1][N1] = x1[M][K1]^T . dy1[M][N1],  db1[N1] = column sums of dy1   (nullable)
 *   dW2[K2][N2] = x2[M][K2]^T . dy2[M][N2],  db2[N2] likewise
 * as a stream-K product: the tile-iterations of both outputs are dealt evenly to 2 blocks per
 * CU, so no CU idles while another still has a whole tile to do (480 + 80 tiles on 512 slots
 * at the step's shapes); tiles computed in parts are summed in a fixed order by a second small
 * launch (deterministic, no atomics).  K1, N1, K2, N2 multiples of 128, M >= 256; otherwise
 * CDML_E_UNSUPPORTED and cdml_fc_bwd_weight2_workspace returns 0: call cdml_fc_bwd_weight per
 * layer.  workspace: cdml_fc_bwd_weight2_workspace bytes, 16-B aligned. */
size_t cdml_fc_bwd_weight2_workspace(int M, int K1, int N1, int K2, int N2);
int cdml_fc_bwd_weight2(const float *x1, int64_t ldx1, const float *dy1, int64_t lddy1,
                        int K1, int N1, float *dW1, int64_t lddw1, float *db1,
                        const float *x2, int64_t ldx2, const float *dy2, int64_t lddy2,
                        int K2, int N2, float *dW2, int64_t lddw2, float *db2, int M,
                        void *workspace, size_t workspace_bytes, cdml_stream_t stream);

/* ---- loss: HingeLoss.calculate_loss (losses.py:20-49) fused with its
 * gradient (train.py:141) --------------------------------------------------
 * e[3B][ld] rows a,p,n per triplet.  pos/neg/hinge: float[B] (the reference's
 * pos_dist / neg_dist / hinge_dist, shape [B,1]).  stats float[4] =
 * {hinge_loss (mean), mean_pos_dist, mean_neg_dist, active fraction}.
 * de (may be NULL) = d hinge_loss / d e, [3B][ldde]. */
int cdml_triplet_hinge(const float *e, int64_t lde, int B, int D, float margin,
                       float *pos, float *neg, float *hinge, float *stats,
                       float *de, int64_t ldde, cdml_stream_t stream);

/* In-batch variant: e[2B][ld] rows a_i,p_i; rows int32[2B] video ids;
 * shift int32[1] (device).  Triplet i = (a_i, p_i, p_{(i+shift)%B}); it is
 * masked (hinge 0, no gradient, still counted in the mean) when the negative's
 * video id equals a_i's or p_i's.  valid_out uint8[B] may be NULL. */
int cdml_triplet_hinge_inbatch(const float *e, int64_t lde, const int32_t *rows,
                               const int32_t *shift, int B, int D, float margin,
                               float *pos, float *neg, float *hinge,
                               uint8_t *valid_out, float *stats, float *de,
                               int64_t ldde, cdml_stream_t stream);

/* Fused tail of the tower for the training step: tf.nn.l2_normalize of the output layer
 * (models.py:61) -> HingeLoss.calculate_loss (losses.py:32-38) -> d loss / d e -> l2-normalise
 * backward -> leaky-relu' of the output layer (train.py:141), one launch, one wave per triplet
 * slot (= cdml_l2norm_fwd + cdml_triplet_hinge{,_inbatch} + cdml_l2norm_bwd, same bits).
 *   mode 0: z[3B][ldz] rows a,p,n per triplet (rows/shift unused).
 *   mode 1: z[2B][ldz] rows a_i,p_i, in-batch negatives as cdml_triplet_hinge_inbatch.
 * Outputs: e (unit rows, "l2_norm"), pos/neg/hinge float[B], valid_out uint8[B] (mode 1,
 * nullable), dz2 = d loss / d (pre-activation of the output layer), optionally also as bf16
 * (dz2_bf16 nullable, round-to-nearest-even).  stats (nullable) float[8]: [0..3] as
 * cdml_triplet_hinge (a second small launch).  var_ws (nullable, cdml_vnet_tail_workspace bytes;
 * needs stats): stats[4] = calc_var (train.py:67-71) of the [B,3,D] triplet tensor: mean over
 * everything of (t - mean over [batch, role])^2. */
/* ticket words of cdml_adam_step's advance_step: uint32[CDML_TICKET_WORDS], zero before the first call */
#define CDML_TICKET_WORDS 128
size_t cdml_vnet_tail_workspace(int B, int D);
int cdml_vnet_tail(int mode, const float *z, int64_t ldz, const int32_t *rows,
                   const int32_t *shift, int B, int D, float margin, float lrelu_alpha,
                   float *e, int64_t lde, float *pos, float *neg, float *hinge,
                   uint8_t *valid_out, float *dz2, int64_t lddz2, uint16_t *dz2_bf16,
                   int64_t ldbf, float *stats, float *var_ws, cdml_stream_t stream);

/* ---- semi-hard negative mining (BASELINE config 2; build-defined, no reference
 * counterpart -- spec: oracle/tower.py semihard_select) -----------------------
 * e[2B][lde]: row 2i = anchor i, 2i+1 = positive i; rows int32[2B] video ids.
 * S[B][ldS] = dot products of every anchor with every embedded row (one
 * cdml_fc_bwd_data call: dy = anchors (ld 2*lde), W = e, no mask).
 * neg_row_out[i] = the closest other-video row farther than the positive, else
 * the farthest other-video row, else -1.  sqn_scratch: float[2B]. */
int cdml_semihard_select(const float *S, int64_t ldS, const float *e,
                         int64_t lde, const int32_t *rows, int B, int D,
                         float *sqn_scratch, int32_t *neg_row_out,
                         cdml_stream_t stream);

/* The same selection with the score product on the bf16 matrix cores (fp32 operands as three exact bf16 planes, six
 * plane products: precision "f32x3") and the selection as that product's EPILOGUE: S is never written (537 MB at
 * B = 8192).  Scratch owned by the caller: e_planes = bf16 [2B][ldp] (planes `plane` >= D apart, ldp >= 2 plane + D),
 * sqn float[2B], dp float[B] (receives d(anchor, positive)), workspace of cdml_semihard_mine_x3_workspace(B) bytes.
 * 2B % 256 == 0, D % 64 == 0.  Same rule and tie-breaks as cdml_semihard_select; the distances differ from its by
 * rounding, so a candidate within ~1e-6 of d_p may fall on either side (the test's tolerance). */
size_t cdml_semihard_mine_x3_workspace(int B);
int cdml_semihard_mine_x3(const float *e, int64_t lde, const int32_t *rows, int B, int D,
                          uint16_t *e_planes, int64_t ldp, int64_t plane, float *sqn, float *dp,
                          void *workspace, size_t workspace_bytes, int32_t *neg_row_out,
                          cdml_stream_t stream);
/* The same from the output layer's UN-normalised rows z (round 6): the prep launch normalises them first (cdml_l2norm_fwd's
 * arithmetic, models.py:61) and WRITES e (rows 0 .. 2B-1) -- one launch and a round trip of the embedded rows fewer in
 * BASELINE config 2's step. */
int cdml_semihard_mine_x3_z(const float *z, int64_t ldz, float *e, int64_t lde, const int32_t *rows, int B, int D,
                            uint16_t *e_planes, int64_t ldp, int64_t plane, float *sqn, float *dp,
                            void *workspace, size_t workspace_bytes, int32_t *neg_row_out,
                            cdml_stream_t stream);

/* Hinge loss + gradient over triplets (row 2i, row 2i+1, row neg_row[i]);
 * neg_row[i] = -1 masks a triplet (hinge 0, still counted in the mean).  Rows
 * mined by several anchors accumulate their gradients in ascending triplet
 * order (deterministic).  scale_scratch: B 32-bit words of scratch (the forward launch leaves every triplet's key
 * for the gradient launch there: the mined row if the triplet is active, -1 if not). */
int cdml_triplet_hinge_indexed(const float *e, int64_t lde,
                               const int32_t *neg_row, int B, int D, float margin,
                               float *pos, float *neg, float *hinge, float *stats,
                               float *scale_scratch, float *de, int64_t ldde,
                               cdml_stream_t stream);
/* The same with the rest of the step's tail folded into the gradient launch (round 6; BASELINE config 2's step):
 * z (pre-normalisation output rows, e = l2norm(z)) given, every row's finished gradient de goes straight through
 * cdml_l2norm_bwd's arithmetic -- dz2 = l2norm-backward(z, de) times leaky-relu'(z) (lrelu_alpha < 0: none) -- and,
 * dz2_bf16 given, its bf16 copy (plane_bf = 0) or its three exact bf16 planes hi | mid | lo, plane_bf elements apart
 * (what cdml_split_f32_bf16x3 writes).  Equal to the separate launches: de bit for bit, dz2 to a few ulp (which
 * multiply-adds become fmas differs between kernels), the planes exactly those of the dz2 written here
 * (losses.py:32-38, models.py:61, train.py:141).  z = NULL: cdml_triplet_hinge_indexed. */
int cdml_triplet_hinge_indexed_tail(const float *e, int64_t lde, const int32_t *neg_row, int B, int D,
                                    float margin, float *pos, float *neg, float *hinge, float *stats,
                                    float *scale_scratch, float *de, int64_t ldde, const float *z,
                                    int64_t ldz, float lrelu_alpha, float *dz2, int64_t lddz,
                                    uint16_t *dz2_bf16, int64_t ldbf, int64_t plane_bf,
                                    cdml_stream_t stream);

/* ---- evaluation metric: Evaluation.mean_dist / mean_cos_dist (evaluate.py:57-90)
 * e[n_rows][lde] embeddings; pairs int32[P][2] row indices (must be < n_rows).
 * sqdist[P] = sum (a-b)^2, dot[P] = sum a*b; means float[4] (may be NULL):
 * means[1] = mean sqdist (= mean_dist), means[2] = mean dot (= mean_cos_dist). */
int cdml_pair_dist(const float *e, int64_t lde, int n_rows, const int32_t *pairs,
                   int P, int D, float *sqdist, float *dot, float *means,
                   cdml_stream_t stream);

/* ---- co-watch graph statistics of the ETL (parse_data.py:221-289).
 * pairs: int32 [n_pairs][2] row ids (8-B aligned).  self_pair_flag: device int32 the caller
 * zeroes; set to 1 if some pair has a == p (get_cowatch_graph raises RuntimeError there,
 * parse_data.py:244-246 -- the host side raises).  Counts (n_edges_out, out_count) are written
 * to DEVICE int64s.  workspace: cdml_cowatch_workspace(n_pairs) bytes, 256-B aligned.
 *   cdml_cowatch_graph : the distinct UNDIRECTED edges (min, max) in ascending order into
 *     edges_out [<= n_pairs][2] with their multiplicities in counts_out (get_cowatch_graph).
 *   cdml_cowatch_select: unique == 0 -> every pair whose edge was seen >= threshold times, all
 *     occurrences, input order kept (select_cowatch's default; threshold <= 1 keeps everything);
 *     unique != 0 -> each qualifying edge once as (min, max) in ascending edge order (the
 *     reference orients and orders these at random). */
size_t cdml_cowatch_workspace(int64_t n_pairs);
int cdml_cowatch_graph(const int32_t *pairs, int64_t n_pairs, int32_t *edges_out,
                       int32_t *counts_out, int64_t *n_edges_out, int32_t *self_pair_flag,
                       void *workspace, size_t workspace_bytes, cdml_stream_t stream);
int cdml_cowatch_select(const int32_t *pairs, int64_t n_pairs, int threshold, int unique,
                        int32_t *out_pairs, int64_t *out_count, int32_t *self_pair_flag,
                        void *workspace, size_t workspace_bytes, cdml_stream_t stream);

/* ---- exact kNN export (faiss_knn.py:82-131 `calc_knn`: squared-L2 distances D
 * and neighbour ids I over the l2-normalised embeddings, nearest first, the
 * query itself included; the reference asks faiss HNSW, this is the brute-force
 * answer HNSW approximates).  The inner products of a query block with a
 * catalogue block come from cdml_fc_bwd_data(dy=queries, W=catalogue block,
 * x_post=NULL); cdml_knn_merge folds one such [nq][nb] block into the running
 * per-query lists best_d/best_i [nq][CDML_KNN_LIST] (sorted by (distance, id);
 * unused entries = +inf / INT32_MAX): d = q_sq[r] + b_sq[c] - 2*score, id = col0+c,
 * ids >= n_valid (padding rows of the catalogue) are skipped.  first != 0 starts
 * the lists.  k <= CDML_KNN_LIST; nb, lds multiples of 4. */
#define CDML_KNN_LIST 128
int cdml_knn_list_capacity(void);
int cdml_row_sqnorm(const float *x, int64_t ldx, int n_rows, int D, float *out,
                    cdml_stream_t stream);
int cdml_knn_merge(const float *scores, int64_t lds, int nq, int nb, int col0,
                   int n_valid, const float *q_sq, const float *b_sq, int k,
                   float *best_d, int32_t *best_i, int first, cdml_stream_t stream);

/* The same export WITHOUT a score matrix (round 6).  cdml_knn_filter_x3: the query x catalogue-block inner products on
 * the plane kernels (fp32 operands as three bf16 planes [rows][hi D | mid D | lo D], six plane products per fp32 product)
 * whose epilogue appends every element with d = |q|^2 + |b|^2 - 2 q.b <= tau[query] as a (float d, int32 id) pair to
 * slot atomic(cnt[query])++ of the query's `cap`-slot list `cand` (8 B per slot; slots past the capacity are dropped
 * and counted).  tau = the query's current k-th best distance (from a first block of the catalogue through
 * cdml_knn_merge); n_cols (a multiple of 256) catalogue rows starting at row col0; rows >= n_valid are padding.
 * cdml_knn_merge_list: every query's candidates into its sorted list (as cdml_knn_merge keeps it); cnt back to 0;
 * *overflow = 1 if a list was longer than cap (redo the search with the score-block form: nothing is silently lost). */
int cdml_knn_filter_x3(const uint16_t *Q, int64_t ldq, int64_t plane_q, const uint16_t *Bk, int64_t ldb,
                       int64_t plane_b, int nq, int n_cols, int D, const float *q_sq, const float *b_sq,
                       const float *tau, int col0, int n_valid, int32_t *cnt, void *cand, int cap,
                       cdml_stream_t stream);
int cdml_knn_merge_list(const void *cand, int32_t *cnt, int cap, int nq, int k, float *best_d, int32_t *best_i,
                        int32_t *overflow, cdml_stream_t stream);

/* cdml_adam_step on a weight matrix W[K][N] (row-major, contiguous: ld = N) that also writes the
 * bf16 operand copies the config-4 GEMMs read -- W^T as bf16 [N][ldt] (wt_bf16, nullable) and W as
 * bf16 [K][ldc] (wc_bf16, nullable), round-to-nearest-even of the UPDATED weights: the optimizer
 * step of train.py:146 and cdml_transpose_to_bf16 / cdml_cast_f32_bf16 in one pass over the
 * weights (bit-equal to the separate calls).  K, N multiples of 64.  t / t_dev / lr_dev as
 * cdml_adam_step.  bias_w / bias_g / bias_m / bias_v (nullable together; bias_n elements): the
 * layer's bias vector takes the same update in the same launch.  advance_step != 0: the last
 * block to finish does *t_dev += 1, as in cdml_adam_step (tickets: CDML_TICKET_WORDS zeroed words). */
int cdml_adam_matrix_bf16(float *w, const float *g, float *m, float *v, int K, int N, float lr,
                          const float *lr_dev, float beta1, float beta2, float eps, int64_t t,
                          uint64_t *t_dev, uint16_t *wt_bf16, int64_t ldt,
                          uint16_t *wc_bf16, int64_t ldc, float *bias_w, const float *bias_g,
                          float *bias_m, float *bias_v, int bias_n, int advance_step,
                          uint32_t *tickets, cdml_stream_t stream);

/* ---- reduced-precision tower (BASELINE config 4: fp16 catalogue + bf16 MFMA
 * projection; build-defined precision with its own tolerance, never the default).
 * bf16 / fp16 buffers are passed as uint16_t*.  Same layers as the fp32 entry
 * points (models.py:59-60, train.py:141), fp32 accumulation and master weights. --
 * C[M][N] = epilogue(A[M][K] . B[N][K]^T), both operands k-contiguous bf16.
 *   epilogue 0: C bf16 = lrelu(acc + bias[n], alpha)          (FC forward, hidden layer)
 *            1: C f32  = lrelu(acc + bias[n], alpha)          (FC forward, output layer)
 *            2: C bf16 = acc * (aux[m][n] > 0 ? 1 : alpha)    (data gradient * lrelu'; aux NULL = none)
 *            3: C f32  = acc, deterministic split-K (workspace of cdml_gemm_bf16_workspace bytes)
 * N % 128 == 0, K % 64 == 0, any M >= 1.  Epilogue 1 uses the workspace too when given one
 * (a narrow output layer is split over K and finished by the slab combine); without it the
 * one-pass kernel runs.  Results of the two routes differ in summation order only.
 * Epilogue 2 at K == 256, N % 256 == 0 and M * N >= 2^22 (the output layer's data gradient,
 * an HBM-bound product) runs on a streaming kernel: B held in registers, A by 32-row chunks,
 * bit-equal to the tiled kernels.
 *            4: epilogue 0 that ALSO writes the sign bitmask of C through `aux` (an OUTPUT here):
 *               uint8 [M][ldaux bytes], bit j of byte b of row m = (C[m][8b+j] > 0)
 *            5: epilogue 2 reading that bitmask (aux, ldaux in bytes) instead of the bf16 values --
 *               leaky-relu' needs one bit per element: 16 MB instead of 252 MB per step at config 4
 * Epilogues 4 / 5 exist on the 256x256 kernel / the K == 256 streaming kernel only:
 * cdml_gemm_bf16_epilogue_supported says whether a shape takes them (else use 0 / 2). */
size_t cdml_gemm_bf16_workspace(int M, int N, int K);
int cdml_gemm_bf16_epilogue_supported(int epilogue, int M, int N, int K, int64_t lda,
                                      int64_t ldb, int64_t ldc, int64_t ldaux);
int cdml_gemm_bf16_nt(int epilogue, const uint16_t *A, int64_t lda,
                      const uint16_t *B, int64_t ldb, int M, int N, int K, void *C,
                      int64_t ldc, const float *bias, const uint16_t *aux,
                      int64_t ldaux, float alpha, void *workspace,
                      size_t workspace_bytes, cdml_stream_t stream);

/* Weight-gradient form, k-strided operands: C[M][N] f32 = sum_k A[k][M] * B[k][N]
 * (dW = x^T . dy with x [rows][M], dy [rows][N] bf16 as the forward/backward wrote
 * them: no transposed copies).  Needs M, N % 256 == 0 and K % 128 == 0
 * (cdml_gemm_bf16_tn_supported); other shapes: cdml_transpose_to_bf16 +
 * cdml_gemm_bf16_nt.  colsum (nullable): colsum[n] = sum_k B[k][n], the bias
 * gradient, accumulated from the same LDS tiles.  Deterministic split-K and
 * column-sum partials through `workspace` (cdml_gemm_bf16_tn_workspace bytes). */
int cdml_gemm_bf16_tn_supported(int M, int N, int K, int64_t lda, int64_t ldb);
size_t cdml_gemm_bf16_tn_workspace(int M, int N, int K);
int cdml_gemm_bf16_tn(const uint16_t *A, int64_t lda, const uint16_t *B, int64_t ldb,
                      int M, int N, int K, float *C, int64_t ldc, float *colsum,
                      void *workspace, size_t workspace_bytes, cdml_stream_t stream);

/* BOTH weight gradients of the two-layer tower in one launch (train.py:141): C1 = A1^T B1 and
 * C2 = A2^T B2 over the same K (the batch rows), as cdml_gemm_bf16_tn each; colsum1 / colsum2
 * (nullable) = column sums of B1 / B2 (the bias gradients).  One block per CU takes an equal run
 * of the joint (tile, k) iteration space; the pieces are written as tile-local fp32 partials
 * and a fix-up pass adds them in k order (bit-reproducible).  The narrow second-layer product,
 * bound by streaming its activations from HBM, runs beside the first layer's MFMA-bound one.
 * cdml_gemm_bf16_tn2_workspace: bytes needed; 0 = shapes not taken (M, N % 256, K % 128). */
size_t cdml_gemm_bf16_tn2_workspace(int M1, int N1, int M2, int N2, int K);
int cdml_gemm_bf16_tn2(const uint16_t *A1, int64_t lda1, const uint16_t *B1, int64_t ldb1,
                       int M1, int N1, float *C1, int64_t ldc1, float *colsum1,
                       const uint16_t *A2, int64_t lda2, const uint16_t *B2, int64_t ldb2,
                       int M2, int N2, float *C2, int64_t ldc2, float *colsum2, int K,
                       void *workspace, size_t workspace_bytes, cdml_stream_t stream);

/* ---- fp32 products on the bf16 matrix cores (precision "f32x3"; csrc/gemm_bf16x3.hip) ----
 * Replaces the same reference lines as the fp32 GEMMs above (models.py:59-60, train.py:141): an fp32 value is
 * exactly hi + mid + lo with three bf16 values, a bf16 x bf16 product is exact in the MFMA's fp32 accumulator, and
 * the `products` = 6 plane products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid miss a product by < 2^-26 of it
 * (3: the first three, 16-bit operands).  Operands are three bf16 planes side by side; K, M, N are the fp32
 * problem's.  cdml_split_f32_bf16x3: dst[r][p*plane + c] = plane p of src[r][c] (transpose != 0:
 * dst[c][p*plane + r]).  _nt: C = epilogue(A . B^T), A[M][hi K|mid K|lo K] (plane stride plane_a along k), B[N][..];
 * epilogue 1: fp32 C = lrelu(. + bias); 3: fp32 C; 6: C = the three planes (bf16, plane_c apart) of
 * lrelu(. + bias); 7: C = planes of (. * (aux > 0 ? 1 : alpha)), aux = bf16 [M][ldaux]; 8: 6 with bias[ROW] (a layer
 * computed transposed).  colsum (nullable; M % 256): colsum[n] = sum_k B[n][k].  N % 256, K % 64.
 * _tn: C[M][N] fp32 = sum_k A[k][M] B[k][N], A[K][hi M|mid M|lo M] (plane stride along the columns), B[K][..];
 * C = lrelu(. + bias) when bias is given (a forward layer whose activations are stored transposed);
 * colsum[n] = sum_k B[k][n] on request.  M, N % 256, K % 128.  Workspace: cdml_gemm_bf16x3_workspace(tn, ...). */
int cdml_split_f32_bf16x3(const float *src, int64_t ld_src, int rows, int cols, uint16_t *dst,
                          int64_t ld_dst, int64_t plane, int transpose, cdml_stream_t stream);
/* cdml_sample_gather (inputs.py:102-166, models.py:58) writing every l2-normalised row as its three planes:
 * x_out_planes = bf16 [rows][out_stride], out_stride = 3 planes of out_stride / 3 >= F columns. */
int cdml_sample_gather_x3(int mode, const int32_t *pairs, int64_t n_pairs, uint64_t seed,
                          uint64_t step, const uint64_t *step_dev, int batch, int64_t slot0,
                          int64_t batch_global, const float *table, int64_t n_rows,
                          int64_t row_stride, int F, int32_t *idx_out, int32_t *shift_out,
                          uint16_t *x_out_planes, int64_t out_stride, int n_steps, int64_t x_step_stride,
                          int64_t idx_step_stride, int32_t *oob_flag, cdml_stream_t stream);
/* The last step of the row exchange on the split-fp32 path (round 5; the build's data-parallel extension, no reference row):
 * x_out_planes[r] = the three bf16 planes of the fp32 row src[idx[r]] AS STORED -- request order and operand form in one pass
 * (flags bit 1: idx -1 = a request that found no slot -> NaN planes, else left untouched). */
int cdml_gather_rows_x3(const float *src, int64_t n_rows, int64_t row_stride, const int32_t *idx, int n_idx, int F,
                        int flags, uint16_t *x_out_planes, int64_t out_stride, int32_t *oob_flag,
                        cdml_stream_t stream);
/* cdml_sample_gather_x3 that ALSO writes every step's rows k8-interleaved (round 5): x_ki = bf16
 * [3 planes][rows per step / 8][out_stride / 3][8 rows] per step, steps ki_step_stride elements apart -- the operand layout of
 * cdml_gemm_bf16x3_tnk (the first layer's weight gradient contracts over these rows).  rows per step % 8 == 0,
 * out_stride / 3 a multiple of 256. */
int cdml_sample_gather_x3k(int mode, const int32_t *pairs, int64_t n_pairs, uint64_t seed,
                           uint64_t step, const uint64_t *step_dev, int batch, int64_t slot0,
                           int64_t batch_global, const float *table, int64_t n_rows,
                           int64_t row_stride, int F, int32_t *idx_out, int32_t *shift_out,
                           uint16_t *x_out_planes, int64_t out_stride, int n_steps, int64_t x_step_stride,
                           int64_t idx_step_stride, int32_t *oob_flag, uint16_t *x_ki, int64_t ki_step_stride,
                           cdml_stream_t stream);
/* cdml_vnet_tail (models.py:61, losses.py:32-38, train.py:141) writing dz2 also as its three planes:
 * dz2_planes = bf16 [rows][ldbf], plane p at columns p * plane_bf (plane_bf >= D, ldbf >= 2 plane_bf + D). */
int cdml_vnet_tail_planes(int mode, const float *z, int64_t ldz, const int32_t *rows,
                          const int32_t *shift, int B, int D, float margin, float lrelu_alpha,
                          float *e, int64_t lde, float *pos, float *neg, float *hinge,
                          uint8_t *valid_out, float *dz2, int64_t lddz2, uint16_t *dz2_planes,
                          int64_t ldbf, int64_t plane_bf, float *stats, float *var_ws,
                          cdml_stream_t stream);
/* cdml_adam_matrix_bf16 (train.py:146) writing the operand copies as planes: wt_planes = W^T as
 * [N][hi K | mid K | lo K] (plane stride plane_t), wc_planes = W as [K][hi N | mid N | lo N] (plane_c). */
int cdml_adam_matrix_planes(float *w, const float *g, float *m, float *v, int K, int N, float lr,
                            const float *lr_dev, float beta1, float beta2, float eps, int64_t t,
                            uint64_t *t_dev, uint16_t *wt_planes, int64_t ldt, int64_t plane_t,
                            uint16_t *wc_planes, int64_t ldc, int64_t plane_c, float *bias_w,
                            const float *bias_g, float *bias_m, float *bias_v, int bias_n,
                            int advance_step, uint32_t *tickets, cdml_stream_t stream);
/* The narrow forward layer (N = 256) of cdml_gemm_bf16x3_nt splits its contraction into slabs.  Their length follows K
 * and the row-tile class of the call (round 6): 60 K-tile steps where 120-step slabs would give at most 64 tiles (small
 * batches: the reference's own B = 1 024), else 120 -- within a class the partition depends on K alone (a batch whole or
 * in row blocks: the same bits).  cdml_x3_slab_steps(steps) pins ONE length (a multiple of 6, >= 12) for the calling
 * thread's later calls whatever their size -- catalogue inference pins 120, so an embedding does not depend on the
 * chunk it was computed in; 0 = back to the rule.  Returns the previous pin.  Needs no GPU. */
int cdml_x3_slab_steps(int steps);

/* cdml_gemm_bf16x3_nt epilogues: 1 fp32 C = lrelu(. + bias); 3 fp32 C; 6 C = the three planes of lrelu(. + bias);
 * 7 C = planes of (. times (aux > 0 ? 1 : alpha)), aux = bf16 values [M][ldaux]; 8 = 6 with the bias indexed by the
 * output row; 9 = 6 that ALSO writes the sign bitmask of its result to aux (as uint8 [M][ldaux BYTES], bit j of byte b of
 * row r = C[r][8 b + j] > 0); 10 = 7 reading that bitmask (leaky-relu' of the hidden layer, models.py:59 / train.py:141,
 * as one bit per element instead of a 2-byte value).  The narrow forward layer (N == 256) splits its contraction into
 * slabs of 60 K-tile steps when given the workspace -- a partition that depends on K alone. */
size_t cdml_gemm_bf16x3_workspace(int tn, int M, int N, int K, int products);
/* The weight gradients (train.py:141) on k8-INTERLEAVED operands, round 5: an operand whose contraction runs over the batch
 * rows stored as bf16 [3 planes][rows / 8][columns][8 rows], so that a fragment of the k-strided product is one aligned
 * 16-B LDS read instead of two transposed ones.  cdml_interleave8_bf16x3 converts row-major planes (src [rows][ld_src],
 * plane p at columns p * plane_src; rows, cols % 8 == 0); cdml_gemm_bf16x3_tnk is cdml_gemm_bf16x3_tn on such operands
 * (A: ma columns per row group, the product takes columns [a_col0, a_col0 + M); B likewise; six products; the same
 * workspace, split-K, slab sum and colsum; bit-identical results). */
int cdml_interleave8_bf16x3(const uint16_t *src, int64_t ld_src, int64_t plane_src, int rows, int cols,
                            uint16_t *dst, cdml_stream_t stream);
int cdml_gemm_bf16x3_tnk(const uint16_t *A, int ma, int a_col0, const uint16_t *B, int nb, int b_col0,
                         int M, int N, int K, float *C, int64_t ldc, float *colsum, void *workspace,
                         size_t workspace_bytes, cdml_stream_t stream);
int cdml_gemm_bf16x3_nt(int epilogue, const uint16_t *A, int64_t lda, int64_t plane_a, const uint16_t *B,
                        int64_t ldb, int64_t plane_b, int M, int N, int K, int products, void *C,
                        int64_t ldc, int64_t plane_c, const float *bias, const uint16_t *aux,
                        int64_t ldaux, float alpha, float *colsum, void *workspace,
                        size_t workspace_bytes, cdml_stream_t stream);
int cdml_gemm_bf16x3_tn(const uint16_t *A, int64_t lda, int64_t plane_a, const uint16_t *B, int64_t ldb,
                        int64_t plane_b, int M, int N, int K, int products, float *C, int64_t ldc,
                        const float *bias, float alpha, float *colsum, void *workspace,
                        size_t workspace_bytes, cdml_stream_t stream);

/* ---- the same five products (models.py:59-60, train.py:141) on TWO fp16 planes per fp32 operand: precision "f16x2"
 * (round 6; csrc/gemm_f16x2_256.hip).  A tensor x is held as the fp16 planes hi | lo of x * 2^s -- hi = fp16(x 2^s), lo =
 * fp16(x 2^s - hi), s a per-tensor power of two chosen by the caller (engine_f16x2.py keeps them) -- and a product is the
 * THREE plane products hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_f16, accumulated in fp32 and multiplied by out_scale =
 * 2^-(sa + sb): half the matrix work of cdml_gemm_bf16x3_*, 22 of the 24 significant bits in the operands, fp16's exponent
 * range (a value beyond it saturates at +-65504).  Error against fp64 at the tower's shapes: at or below the fp32-MFMA
 * kernels' own (tests/test_gpu_f16x2.py; the probe that decided it: profiles/r06_f16x2_probe.txt).
 * cdml_split_f32_f16x2: dst = the two planes of src * scale, layout and `transpose` as cdml_split_f32_bf16x3 (ld_dst >=
 * plane + columns).  cdml_gemm_f16x2_nt: epilogues 1, 3, 6, 7, 9, 10 of cdml_gemm_bf16x3_nt; plane outputs are the fp16
 * planes of (result * c_scale).  cdml_gemm_f16x2_tn: C = out_scale * A^T B, colsum[n] = colsum_scale * sum_k B[k][n].
 * The narrow layer's K-slabs partition K exactly as cdml_gemm_bf16x3_nt's do (cdml_x3_slab_steps applies). */
int cdml_split_f32_f16x2(const float *src, int64_t ld_src, int rows, int cols, uint16_t *dst, int64_t ld_dst,
                         int64_t plane, int transpose, float scale, cdml_stream_t stream);
size_t cdml_gemm_f16x2_workspace(int tn, int M, int N, int K);
int cdml_gemm_f16x2_nt(int epilogue, const uint16_t *A, int64_t lda, int64_t plane_a, const uint16_t *B,
                       int64_t ldb, int64_t plane_b, int M, int N, int K, void *C, int64_t ldc, int64_t plane_c,
                       const float *bias, const uint16_t *aux, int64_t ldaux, float alpha, float out_scale,
                       float c_scale, void *workspace, size_t workspace_bytes, cdml_stream_t stream);
int cdml_gemm_f16x2_tn(const uint16_t *A, int64_t lda, int64_t plane_a, const uint16_t *B, int64_t ldb,
                       int64_t plane_b, int M, int N, int K, float *C, int64_t ldc, float out_scale, float *colsum,
                       float colsum_scale, void *workspace, size_t workspace_bytes, cdml_stream_t stream);

/* The producers of "f16x2" plane tensors other than the GEMM epilogues: the fused sampler + gather (inputs.py:125-158)
 * writing each l2-normalised row as the two fp16 planes of x_hat * CDML_F16X2_X_SCALE (|x_hat| <= 1: a constant of the
 * format; out_stride = 2 planes of out_stride / 2 >= F columns); cdml_vnet_tail writing dz2's planes times `scale`
 * (dz2_planes fp16 [rows][ldbf], ldbf >= plane_h + D); cdml_adam_matrix_bf16 / cdml_lars_matrix / cdml_momentum_matrix
 * (train.py:146, :354, :115) writing the weight copies as the planes of W * scale (wt = W^T [N][hi K | lo K], wc = W
 * [K][hi N | lo N]).  A value beyond fp16's range saturates at +-65504. */
#define CDML_F16X2_X_SCALE 16384.0f
int cdml_sample_gather_h2(int mode, const int32_t *pairs, int64_t n_pairs, uint64_t seed,
                          uint64_t step, const uint64_t *step_dev, int batch, int64_t slot0,
                          int64_t batch_global, const float *table, int64_t n_rows,
                          int64_t row_stride, int F, int32_t *idx_out, int32_t *shift_out,
                          uint16_t *x_out_planes, int64_t out_stride, int n_steps, int64_t x_step_stride,
                          int64_t idx_step_stride, int32_t *oob_flag, cdml_stream_t stream);
int cdml_vnet_tail_h2(int mode, const float *z, int64_t ldz, const int32_t *rows,
                      const int32_t *shift, int B, int D, float margin, float lrelu_alpha,
                      float *e, int64_t lde, float *pos, float *neg, float *hinge,
                      uint8_t *valid_out, float *dz2, int64_t lddz2, uint16_t *dz2_planes,
                      int64_t ldbf, int64_t plane_h, float scale, float *stats, float *var_ws,
                      cdml_stream_t stream);
/* cdml_semihard_mine_x3 / _z (z nullable) and cdml_knn_filter_x3 with their score products on two fp16 planes per row --
 * the rows times `scale` (unit rows: 2^14), three plane products on the fp16 MFMA; e_planes = fp16 [2B][ldp >= plane + D];
 * cdml_knn_filter_h2: Q / Bk from cdml_split_f32_f16x2 at scales sq / sb, out_scale = 1 / (sq sb). */
int cdml_semihard_mine_h2(const float *z, int64_t ldz, float *e, int64_t lde, const int32_t *rows, int B, int D,
                          uint16_t *e_planes, int64_t ldp, int64_t plane, float scale, float *sqn_scratch,
                          float *dp_scratch, void *workspace, size_t workspace_bytes, int32_t *neg_row_out,
                          cdml_stream_t stream);
int cdml_knn_filter_h2(const uint16_t *Q, int64_t ldq, int64_t plane_q, const uint16_t *Bk, int64_t ldb,
                       int64_t plane_b, int nq, int n_cols, int D, float out_scale, const float *q_sq,
                       const float *b_sq, const float *tau, int col0, int n_valid, int32_t *cnt, void *cand,
                       int cap, cdml_stream_t stream);
int cdml_lars_matrix_h2(float *w, const float *g, float *acc, const int64_t *seg_offsets, const int64_t *seg_sizes,
                        int n_seg, int seg_matrix, int seg_bias, int K, int N, float lr, const float *lr_dev,
                        float momentum, float weight_decay, float eeta, float eps, const float *scratch,
                        float *norms_out, uint16_t *wt_planes, int64_t ldt, int64_t plane_t, uint16_t *wc_planes,
                        int64_t ldc, int64_t plane_c, float scale, uint64_t *step_dev_advance, uint32_t *tickets,
                        cdml_stream_t stream);
int cdml_momentum_matrix_h2(float *w, const float *g, float *acc, int K, int N, float lr, const float *lr_dev,
                            float momentum, int use_nesterov, uint16_t *wt_planes, int64_t ldt, int64_t plane_t,
                            uint16_t *wc_planes, int64_t ldc, int64_t plane_c, float scale, float *bias_w,
                            const float *bias_g, float *bias_acc, int bias_n, uint64_t *step_dev_advance,
                            uint32_t *tickets, cdml_stream_t stream);
int cdml_adam_matrix_h2(float *w, const float *g, float *m, float *v, int K, int N, float lr,
                        const float *lr_dev, float beta1, float beta2, float eps, int64_t t,
                        uint64_t *t_dev, uint16_t *wt_planes, int64_t ldt, int64_t plane_t,
                        uint16_t *wc_planes, int64_t ldc, int64_t plane_c, float scale, float *bias_w,
                        const float *bias_g, float *bias_m, float *bias_v, int bias_n,
                        int advance_step, uint32_t *tickets, cdml_stream_t stream);

/* dst[c][r] = bf16(src[r][c]) (src fp32 or bf16): k-contiguous copies of weights
 * and of activations for the weight-gradient GEMMs (contraction over batch rows). */
int cdml_transpose_to_bf16(int src_is_f32, const void *src, int64_t ld_src,
                           int rows, int cols, uint16_t *dst, int64_t ld_dst,
                           cdml_stream_t stream);
int cdml_cast_f32_bf16(const float *src, int64_t ld_src, int rows, int cols,
                       uint16_t *dst, int64_t ld_dst, cdml_stream_t stream);

/* out[c] = sum_r src[r][c] (bias gradients), two fixed-order stages;
 * workspace: cdml_colsum_workspace_floats(rows, cols) floats. */
size_t cdml_colsum_workspace_floats(int rows, int cols);
int cdml_colsum(int src_is_bf16, const void *src, int64_t ld, int rows, int cols,
                float *out, float *workspace, cdml_stream_t stream);

/* fp16 catalogue: the Philox table of cdml_fill_uniform_table rounded to half, and
 * the gather of inputs.py:158 + models.py:58 reading fp16 rows and writing
 * l2-normalised (fp32 arithmetic) bf16 rows. */
int cdml_fill_uniform_table_f16(uint16_t *table, int64_t row0, int64_t n_rows,
                                int feature_size, int64_t row_stride,
                                uint64_t seed, cdml_stream_t stream);
int cdml_gather_rows_f16(const uint16_t *table, int64_t row0, int64_t n_rows,
                         int64_t row_stride, const int32_t *idx, int n_idx, int F,
                         uint16_t *x_out_bf16, int64_t out_stride,
                         int32_t *oob_flag, cdml_stream_t stream);
/* cdml_sample_gather on the fp16 catalogue: same sampler, same staging, same n_steps semantics;
 * rows are read as fp16 and written l2-normalised (fp32 arithmetic) as bf16. */
int cdml_sample_gather_f16(int mode, const int32_t *pairs, int64_t n_pairs, uint64_t seed,
                           uint64_t step, const uint64_t *step_dev, int batch,
                           int64_t slot0, int64_t batch_global, const uint16_t *table,
                           int64_t n_rows, int64_t row_stride, int F, int32_t *idx_out,
                           int32_t *shift_out, uint16_t *x_out_bf16, int64_t out_stride,
                           int n_steps, int64_t x_step_stride, int64_t idx_step_stride,
                           int32_t *oob_flag, cdml_stream_t stream);

/* ---- fusion towers MultiplyNet / MlpNet / ResNet (models.py:65-157): the
 * elementwise pieces between their FC layers; [M][N] fp32 views, N % 4 == 0. ----
 * mode 0: out = a*b (tf.multiply, models.py:90,117,148); 1: out = a*b + a + b
 * (ResNet's first residual sum, models.py:150); 2: out = a + b (models.py:152,154). */
int cdml_ew_combine(int mode, const float *a, int64_t lda, const float *b,
                    int64_t ldb, int M, int N, float *out, int64_t ldo,
                    cdml_stream_t stream);
/* Gradient of modes 0 (residual=0) / 1 (residual=1) wrt a and b, each multiplied
 * by the leaky-relu derivative of the FC layer that produced it (a, b are
 * post-activations): da = g*(b+res)*lrelu'(a), db = g*(a+res)*lrelu'(b). */
int cdml_ew_fusion_bwd(int residual, const float *g, int64_t ldg, const float *a,
                       int64_t lda, const float *b, int64_t ldb, int M, int N,
                       float alpha, float *da, int64_t ldda, float *db,
                       int64_t lddb, cdml_stream_t stream);
/* out = g * lrelu'(y), y = post-activation (LeakyReluGrad). */
int cdml_lrelu_bwd(const float *g, int64_t ldg, const float *y, int64_t ldy, int M,
                   int N, float alpha, float *out, int64_t ldo,
                   cdml_stream_t stream);

/* ---- optimizers (train.py:108-125,146) --------------------------------------
 * Adam, TensorFlow form (epsilon outside the bias correction):
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m=b1*m+(1-b1)*g; v=b2*v+(1-b2)*g*g;
 *   w -= lr_t*m/(sqrt(v)+eps).            t = 1-based step.  n elements.
 * lr_dev (nullable) overrides lr, and *t_dev (nullable) is ADDED to t, both read
 * from device memory at execution time (t_dev = the sampler's 0-based step
 * counter with t = 1), so a captured hipGraph follows the step counter and the
 * staircase learning-rate schedule (train.py:108-113) without re-capture.
 * advance_step != 0: the last block to finish stores *t_dev + 1 -- the global_step
 * increment of apply_gradients (train.py:146) without a launch of its own; needs t_dev and
 * tickets = uint32[CDML_TICKET_WORDS], zero before the first call (left zero). */
int cdml_adam_step(float *w, const float *g, float *m, float *v, int64_t n,
                   float lr, const float *lr_dev, float beta1, float beta2,
                   float eps, int64_t t, uint64_t *t_dev, int advance_step,
                   uint32_t *tickets, cdml_stream_t stream);

/* build_graph's gradient options (train.py:133-145; both off in the reference's own run,
 * train.py:221-222), applied to ONE variable in place before its optimizer step:
 *   g <- g + l2_scale*w    slim.l2_regularizer of a weight matrix (models.py:28: 1e-8*|W|^2/2)
 *                          times regularization_penalty; pass 0 for biases
 *   g <- g * clip_norm / max(|g|_2, clip_norm)      tf.clip_by_norm (clip_norm <= 0: off)
 * scratch: float[cdml_lars_scratch_floats()].  norms_out (nullable) float[2] = {|g|_2 after the
 * regulariser and before clipping, |w|_2^2 (for the reg_loss summary)}. */
int cdml_grad_prepare(float *g, const float *w, int64_t n, float l2_scale,
                      float clip_norm, float *scratch, float *norms_out,
                      cdml_stream_t stream);

/* tf.train.MomentumOptimizer(lr, momentum=0.9, use_nesterov=True) (train.py:115-116), TF's
 * ApplyMomentum: acc = acc*momentum + g;  w -= nesterov ? g*lr + acc*momentum*lr : acc*lr. */
int cdml_momentum_step(float *w, const float *g, float *acc, int64_t n, float lr,
                       const float *lr_dev, float momentum, int use_nesterov,
                       cdml_stream_t stream);

/* Trainable catalogue rows (north_star: "the catalogue feature table and its Adam states
 * shard row-wise"; the reference keeps the features frozen, train.py:265, so this is
 * build-defined and off by default; spec oracle/table.py).  grad_xhat[r] = dLoss/d x_hat of
 * the r-th gathered row (= cdml_fc_bwd_data(dz1, W1, NULL)), idx[r] its GLOBAL row id; rows
 * outside [row0, row0+n_rows) are skipped (other shards').  Per touched row, once: the
 * gradients of all batch rows that gathered it are summed in ascending r (deterministic),
 * taken through the l2norm backward of models.py:58 with the raw row, and a lazy-Adam update
 * (the arithmetic of cdml_adam_step; m/v/x of untouched rows are not read) is applied to
 * table, m_table, v_table (same shape and stride).  head: int32[n_rows] scratch that must be
 * all -1 on entry and is -1 again on exit; next: int32[n_idx] scratch.  grad_scale
 * multiplies the summed row gradient (1/world in a data-parallel run, where every rank's
 * grad_xhat carries the mean over its LOCAL batch and the dense gradients are averaged). */
int cdml_table_adam_rows(float *table, int64_t row0, int64_t n_rows, int64_t row_stride,
                         int F, const int32_t *idx, int n_idx, const float *grad_xhat,
                         int64_t ldg, float *m_table, float *v_table, int32_t *head,
                         int32_t *next, float grad_scale, float lr, const float *lr_dev,
                         float beta1, float beta2, float eps, int64_t t,
                         const uint64_t *t_dev, cdml_stream_t stream);

/* LARS (tf.contrib.opt.LARSOptimizer, train.py:354), one variable of n
 * elements: trust = eeta*|w|/(|g|+wd*|w|+eps) (1 if |w|==0 or |g|==0);
 * acc = momentum*acc + lr*trust*(g+wd*w); w -= acc.
 * scratch: cdml_lars_scratch_floats() floats (norm partials; deterministic). */
size_t cdml_lars_scratch_floats(void);
int cdml_lars_step(float *w, const float *g, float *acc, int64_t n, float lr,
                   const float *lr_dev, float momentum, float weight_decay,
                   float eeta, float eps, float *scratch, cdml_stream_t stream);

/* The same update for EVERY variable of a flat parameter buffer in two launches (the training
 * step's form; the reference applies LARS to weights and biases alike, train.py:354): the
 * variables are n_seg (<= 8) contiguous segments, seg_offsets[k] = sum of the sizes before k,
 * sizes multiples of 4 floats (host arrays).  One trust ratio per segment; 16-B accesses.
 * scratch: cdml_lars_multi_scratch_floats() floats.  norms_out (nullable): float[2*n_seg] =
 * {|w|, |g|} per segment.  step_dev_advance (nullable): *step_dev_advance += 1 by the last
 * block of the update (train.py:146 apply_gradients(global_step=...)); needs `tickets`
 * (CDML_TICKET_WORDS zeroed uint32, as cdml_adam_step). */
size_t cdml_lars_multi_scratch_floats(void);
int cdml_lars_multi(float *w, const float *g, float *acc, const int64_t *seg_offsets,
                    const int64_t *seg_sizes, int n_seg, float lr, const float *lr_dev,
                    float momentum, float weight_decay, float eeta, float eps,
                    float *scratch, float *norms_out, uint64_t *step_dev_advance,
                    uint32_t *tickets, cdml_stream_t stream);

/* LARS / momentum with the GEMMs' operand copies written by the update itself -- what cdml_adam_matrix_bf16 /
 * cdml_adam_matrix_planes are for Adam, for the optimizer the reference actually runs (train.py:354: LARSOptimizer;
 * train.py:115-116: MomentumOptimizer).  planes = 1: bf16 copies (config 4); planes = 3: the three bf16 planes
 * hi | mid | lo of the new weights (precision "f32x3"; plane strides as in cdml_adam_matrix_planes).
 *   cdml_lars_multi_norms: launch 1 of cdml_lars_multi alone -- per-block partial sums of |w|^2, |g|^2 of every segment
 *     into scratch (cdml_lars_multi_scratch_floats() floats);
 *   cdml_lars_matrix: the update of ONE weight matrix (segment seg_matrix of the same segment arrays, K x N row-major)
 *     and of its bias vector (segment seg_bias, or -1), each with its own trust ratio reduced from those partials;
 *     w / g / acc are the FLAT buffers; norms_out (nullable) float[2*n_seg] gets (|w|, |g|) of the two variables;
 *     step_dev_advance (nullable, with tickets): global_step += 1 by the last block -- pass it on the last launch.
 *   cdml_momentum_matrix: ApplyMomentum on a matrix (w / g / acc point AT the matrix) + its bias vector.
 * wt_copy = W^T [N][>= K] (nullable), wc_copy = W [K][>= N] (nullable).  Element arithmetic = cdml_lars_multi /
 * cdml_momentum_step bit for bit. */
int cdml_lars_multi_norms(const float *w, const float *g, const int64_t *seg_offsets, const int64_t *seg_sizes,
                          int n_seg, float *scratch, cdml_stream_t stream);
int cdml_lars_matrix(float *w, const float *g, float *acc, const int64_t *seg_offsets, const int64_t *seg_sizes,
                     int n_seg, int seg_matrix, int seg_bias, int K, int N, float lr, const float *lr_dev,
                     float momentum, float weight_decay, float eeta, float eps, const float *scratch,
                     float *norms_out, uint16_t *wt_copy, int64_t ldt, int64_t plane_t, uint16_t *wc_copy,
                     int64_t ldc, int64_t plane_c, int planes, uint64_t *step_dev_advance, uint32_t *tickets,
                     cdml_stream_t stream);
int cdml_momentum_matrix(float *w, const float *g, float *acc, int K, int N, float lr, const float *lr_dev,
                         float momentum, int use_nesterov, uint16_t *wt_copy, int64_t ldt, int64_t plane_t,
                         uint16_t *wc_copy, int64_t ldc, int64_t plane_c, int planes, float *bias_w,
                         const float *bias_g, float *bias_acc, int bias_n, uint64_t *step_dev_advance,
                         uint32_t *tickets, cdml_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CDML_H_ */
