/*
 * tgpose.h -- C ABI of libtgpose_hip.so: the MI355X (gfx950) kernels behind the TG-Pose
 * point-cloud forward path and its Chamfer distance.
 *
 * Conventions (all entry points):
 *   - plain pointers are DEVICE pointers (fp32 row-major, int32 indices) unless noted;
 *   - every buffer, including scratch, is owned and sized by the caller; nothing is allocated,
 *     nothing synchronises; work is enqueued on `stream` (a hipStream_t passed as void*);
 *   - `ld*` arguments are row strides in ELEMENTS, so operators read/write column slices of
 *     wider buffers (the 1286-channel concat buffer) in place;
 *   - return value: 0 = enqueued; > 0 = hipError_t from the launch; < 0 = TGP_E* argument error
 *     (nothing was enqueued).
 *
 * Each declaration cites the reference interface it replaces (paths under the TG-Pose tree).
 * The reference has exactly one native binding on this path -- the pybind module `chamfer_3D`
 * (losses/chamfer3D/chamfer_cuda.cpp:30-33) -- and otherwise composes torch ops in
 * network/fs_net_repo/gcn3d.py; the operator-level entry points below are what a native binding
 * of those functions would export.  INTEGRATION.md shows the Python-side stubs.
 */
#ifndef TGPOSE_H
#define TGPOSE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGP_ABI_VERSION 7
#define TGP_EINVAL (-1)       /* null pointer / non-positive size / misaligned stride */
#define TGP_EUNSUPPORTED (-2) /* shape outside what the kernels are built for */

typedef void *tgp_stream_t; /* hipStream_t */

int tgp_version(void);
/* Node census of a captured hipGraph (hipGraph_t): counts[0..3] = kernel, memcpy, memset, other nodes.  Host-only, nothing is
 * launched.  No reference counterpart (the reference never captures); the step capture of this repo refuses a graph that holds
 * a MEMSET node -- the mark of an ATen multi-block reduction inside the capture ("captured step"). */
int tgp_graph_node_counts(void *hip_graph, int *counts);
/* largest point count per object / neighbour count the kNN kernels accept */
int tgp_knn_max_points(void);
int tgp_knn_max_k(void);

/* ---- geometry ---------------------------------------------------------------------------- */

/* PoseNet9D.py:48  `points - points.mean(dim=1, keepdim=True)`.
 * points (B,n,3) -> xyz_c (B,n,3), mean (B,3).  The column sums follow ATen's cascade order so
 * that the centred cloud is bit-identical to the CPU path (the kNN order depends on it). */
int tgp_center(const float *points, int B, int n, float *xyz_c, float *mean, tgp_stream_t stream);
/* (ABI 6) tgp_center that also clears zero_words 32-bit words at `zero` (the forward's zero-initialised scratch: max keys, range
 * flags, magnitude words, tickets) in the same launch; every later launch of the forward is ordered behind it. */
int tgp_center_zero(const float *points, int B, int n, float *xyz_c, float *mean, void *zero, int64_t zero_words, tgp_stream_t stream);

/* ---- graph construction ------------------------------------------------------------------- */

/* gcn3d.py:14-23 get_neighbor_index for 3-d coordinates.  xyz (B,n,3) packed -> idx (B,n,k):
 * the k+1 nearest by fl(fl(-2<a,b> + |b|^2) + |a|^2) ordered by (distance, index), rank 0 dropped. */
int tgp_knn_xyz(const float *xyz, int B, int n, int k, int32_t *idx, tgp_stream_t stream);

/* Bytes of scratch tgp_knn_feat needs for (B,n,d). */
int64_t tgp_knn_feat_workspace_bytes(int B, int n, int d);

/* gcn3d.py:14-23 get_neighbor_index in feature space (mode 'RF-F', gcn3d.py:149,201-206).
 * feat (B,n,d) with row stride ld; d a multiple of 32, d <= 480. */
int tgp_knn_feat(const float *feat, int ld, int B, int n, int d, int k, int32_t *idx, void *workspace,
                 int64_t workspace_bytes, tgp_stream_t stream);
/* (ABI 5) the same with the kernel form chosen by the caller -- 0: the library's choice; 1: 32-row blocks, one workgroup per CU
 * (v_mfma_f32_32x32x2_f32); 2: 16-row blocks, two workgroups per CU whose distance and selection phases overlap
 * (v_mfma_f32_16x16x4_f32, the same ascending-k chain); 3 (ABI 7): the 16-row arithmetic on producer / consumer waves of one 512-thread
 * workgroup per CU over a double-buffered LDS image (half of the waves multiply block j + 1 while the other half select block j):
 * the library's choice where a workgroup gets at least four row blocks to walk.  Identical index lists; a measurement / test handle. */
int tgp_knn_feat_form(const float *feat, int ld, int B, int n, int d, int k, int32_t *idx, void *workspace, int64_t workspace_bytes,
                      int form, tgp_stream_t stream);
/* (ABI 6) tgp_knn_feat that also leaves, beside each list, the unit directions from the point to its selected neighbours in the
 * level's coordinates xyz (B, n, 3) (gcn3d.py:48-58 get_neighbor_direction_norm): dirs (B, n, k) float4 = (x, y, z, 0), 16-byte
 * aligned -- what tgp_gconv_hs_fwd_dirs takes, so the graph convolution that walks the list needs no direction launch.  The fused
 * kernel's shapes only (d = 128 / 256, n <= 1056): TGP_EUNSUPPORTED otherwise, nothing launched. */
int tgp_knn_feat_dirs(const float *feat, int ld, int B, int n, int d, int k, int32_t *idx, void *workspace,
                      int64_t workspace_bytes, const float *xyz, float *dirs, tgp_stream_t stream);

/* gcn3d.py:26-35 get_nearest_index: for each of n target points the nearest of m source points
 * by fl(fl(|s|^2 + |t|^2) - 2<t,s>), lowest index on ties.  idx (B,n). */
int tgp_nn1(const float *target, const float *source, int B, int n, int m, int32_t *idx, tgp_stream_t stream);
/* (ABI 6) tgp_nn1 of the same targets against two clouds in one launch: the two nearest-coarse-point look-ups of the up-sampling
 * (FaceRecon.py:71-77).  Same results as two tgp_nn1 calls. */
int tgp_nn1_pair(const float *target, const float *source1, const float *source2, int B, int n, int m1, int m2, int32_t *idx1,
                 int32_t *idx2, tgp_stream_t stream);
/* (ABI 6) tgp_nn1_pair whose launch also writes tgp_fill_tail's columns for the targets (target = the centred cloud xyz_c):
 * feat[b, i, col0 ...] = one-hot(obj_id[b]) | target[b, i] | 0 ... up to ld.  feat NULL = tgp_nn1_pair. */
int tgp_nn1_pair_tail(const float *target, const float *source1, const float *source2, int B, int n, int m1, int m2, int32_t *idx1,
                      int32_t *idx2, const float *obj_id, int n_cls, float *feat, int ld, int col0, tgp_stream_t stream);

/* ---- graph convolution --------------------------------------------------------------------- */

/* F.normalize(directions, dim=0) (gcn3d.py:100,165).  directions (3,SC) -> out (3,SC). */
int tgp_normalize_dirs(const float *directions, int SC, float *out, tgp_stream_t stream);
/* its backward (round 3): out (3, SC) = d loss / d directions from grad (3, SC) = d loss / d normalised directions */
int tgp_normalize_dirs_bwd(const float *directions, const float *grad, int SC, float *out, tgp_stream_t stream);

/* gcn3d.py:91-106 HSlayer_surface.graph_conv.  xyz (B,n,3), idx (B,n,k), sdn (3,S*C) unit support
 * directions -> out (B,n,C) row stride ldo:  mean_s max_j relu(<dir_j, sdn[:, s*C+c]>).
 * xyz_pad != 0 (ldo >= C + 4): the row's columns C..C+3 also receive the point (x, y, z, 0), so that the caller's next GEMM can
 * take HSlayer_surface's STE convolution of xyz (gcn3d.py:79,87) as four more K columns (ABI 4). */
int tgp_gconv_surface_fwd(const float *xyz, const int32_t *idx, const float *sdn, int B, int n, int k, int S,
                          int C, float *out, int ldo, int xyz_pad, tgp_stream_t stream);

/* gcn3d.py:157-180 HS_layer.graph_conv after the dense projection.  proj (B*n, (S+1)*C) row stride
 * ldp holds [centre | support] = feature_map @ weights + bias; idx is the feature-space graph.
 * out = centre + mean_s max_j relu(<dir_j, sdn>) * support[idx_j].
 * dirs_ws: scratch of B*n*k*4 floats (16-byte aligned) for the unit neighbour directions (gcn3d.py:48-58), or NULL;
 * with it, clouds whose support table fits LDS in channel slices take the LDS-staged kernel (same results). */
int tgp_gconv_hs_fwd(const float *xyz, const int32_t *idx, const float *proj, int ldp, const float *sdn, int B,
                     int n, int k, int S, int C, float *out, int ldo, float *dirs_ws, tgp_stream_t stream);
/* (ABI 6) tgp_gconv_hs_fwd with the unit neighbour directions supplied (tgp_knn_feat_dirs): the LDS-staged kernel alone.
 * TGP_EUNSUPPORTED (nothing launched) where that kernel does not serve the shape: the caller uses tgp_gconv_hs_fwd. */
int tgp_gconv_hs_fwd_dirs(const float *xyz, const int32_t *idx, const float *proj, int ldp, const float *sdn, int B,
                          int n, int k, int S, int C, float *out, int ldo, const float *dirs, tgp_stream_t stream);

/* gcn3d.py:210-217 get_ORL_global: g[b,c] = mean_i max_j feat[b, idx[b,i,j], c].
 * partial: scratch of tgp_orl_partial_floats(B,n,C) floats.  out (B,C). */
int64_t tgp_orl_partial_floats(int B, int n, int C);
int tgp_orl_global(const float *feat, int ldf, const int32_t *idx, int B, int n, int k, int C, float *partial,
                   float *out, tgp_stream_t stream);

/* get_ORL_global followed by the global half of ORL_forward's conv2 (gcn3d.py:108-112,182-186):
 * g as above (optionally stored to g_out, may be NULL), rb[b,:] = g[b,:] @ W2^T with w2t = W2 transposed (C,C). */
int tgp_orl_rowbias(const float *feat, int ldf, const int32_t *idx, int B, int n, int k, int C, float *partial,
                    const float *w2t, float *g_out, float *rb, tgp_stream_t stream);
/* (ABI 5) the same, and feat -- the A operand of the layer's last GEMM (gcn3d.py:108-112: conv2's feature half) -- also written as
 * blocked fp16 planes (tgp_gemm_args.A_planes: `kts` K-tiles per row block, K-tile c / 16 for channel c) with the per-row-block
 * magnitudes in amax (zero-filled by the caller; may be NULL).  xyz_tile (may be NULL; HSlayer_surface): points (B*n, 3) written as
 * K-tile C / 16 = (x, y, z, 0, ...), the STE convolution's extra K columns.  TGP_EUNSUPPORTED where the LDS-staged form does not
 * serve the shape (the caller falls back to tgp_orl_rowbias and an fp32 operand). */
int tgp_orl_rowbias_planes(const float *feat, int ldf, const int32_t *idx, int B, int n, int k, int C, float *partial,
                           const float *w2t, float *g_out, float *rb, void *planes, int kts, uint32_t *amax, const float *xyz_tile,
                           tgp_stream_t stream);
/* (ABI 6) tgp_orl_rowbias / tgp_orl_rowbias_planes (planes may be NULL) as ONE launch: the pooling kernel finishes the mean over
 * points and the projection itself -- each of an object's C / 16 workgroups adds its 16-channel slice of g @ W2^T to `contrib`
 * (B * (C / 16) * C floats of scratch) and the last one to arrive (tickets: B ints, zero on entry, zero again on return) sums the
 * slices in chunk order, so rb does not depend on the arrival order.  rb agrees with tgp_orl_rowbias to rounding (another summation
 * order), g_out bit for bit.  TGP_EUNSUPPORTED (nothing launched) where the LDS-staged form does not serve the shape. */
int tgp_orl_rowbias_fused(const float *feat, int ldf, const int32_t *idx, int B, int n, int k, int C, const float *w2t,
                          float *g_out, float *rb, void *planes, int kts, uint32_t *amax, const float *xyz_tile, float *contrib,
                          int32_t *tickets, tgp_stream_t stream);

/* gcn3d.py:219-245 Pool_layer.forward with the random subsample (randperm, :242) supplied by the
 * host: out_f[b,m,:] = max_{j<kpool} feat[b, idx[b, sample[m], j], :], out_xyz[b,m] = xyz[b, sample[m]].
 * idx has row stride ldi (>= kpool) so the k=4 list may be the prefix of a longer one. */
int tgp_pool_fwd(const float *xyz, const float *feat, int ldf, const int32_t *idx, int ldi, const int32_t *sample,
                 int B, int n, int n_out, int kpool, int C, float *out_xyz, float *out_f, int ldo,
                 tgp_stream_t stream);
/* (ABI 5) the same, the pooled features also written as blocked fp16 planes (tgp_gemm_args.A_planes; kts >= C / 16 K-tiles per row
 * block) with their per-row-block magnitude words (amax, zero-filled by the caller; may be NULL): they are the A operand of the next
 * layer's projection GEMM.  C / 4 a divisor of 64, else TGP_EUNSUPPORTED. */
int tgp_pool_fwd_planes(const float *xyz, const float *feat, int ldf, const int32_t *idx, int ldi, const int32_t *sample, int B, int n,
                        int n_out, int kpool, int C, float *out_xyz, float *out_f, int ldo, void *planes, int kts, uint32_t *amax,
                        tgp_stream_t stream);

/* gcn3d.py:38-46 indexing_neighbor_new with one neighbour (FaceRecon.py:69-73 nearest up-sampling):
 * dst[b,i,0:C] = src[b, idx[b,i], 0:C]. */
int tgp_gather_rows(const float *src, int lds, const int32_t *idx, int B, int n_src, int n_out, int C, float *dst,
                    int ldd, tgp_stream_t stream);

/* FaceRecon.py:49-54,75-77 and PoseNet9D.py:63: the tail columns of the concat buffer:
 * one_hot(obj_id) (n_cls columns at col0), then the centred xyz (3), then zero padding up to ld. */
int tgp_fill_tail(const float *obj_id, const float *xyz_c, int B, int n, int n_cls, float *feat, int ld, int col0,
                  tgp_stream_t stream);

/* ---- dense per-point layers ------------------------------------------------------------------ */

typedef struct tgp_gemm_args {
    /* C[m, n] = act( (sum_k A[m,k] * W[n,k] + bias[n] + rowbias[m / rows_per_obj, n]
     *                 + res1[m,n] + res2[m,n]) * scale[n] + shift[n] )
     * A (M,K) row stride lda; W (N,K) row stride ldw (Conv1d/Linear weight layout, FaceRecon.py:93-110,
     * PoseR.py:16-19); K, lda, ldw multiples of 4; every optional pointer may be NULL. */
    const float *A; int lda;
    const float *W; int ldw;
    float *C; int ldc;                 /* may be NULL when only colmax is wanted */
    int M, N, K;
    const float *bias;                 /* [N] */
    const float *rowbias; int ldrb;    /* [(M / rows_per_obj), N] */
    int rows_per_obj;
    const float *res1; int ldr1;
    const float *res2; int ldr2;
    const float *scale;                /* [N] eval-mode BatchNorm fold: gamma / sqrt(var + eps) */
    const float *shift;                /* [N] beta - mean * scale */
    int act;                           /* 0 none, 1 leaky-relu with `slope` (0 -> ReLU) */
    float slope;
    uint32_t *colmax_keys; int ldcm;   /* [(M / rows_per_obj), cm_cols] order-preserving keys, zero-filled by the
                                          caller; decoded by tgp_colmax_decode (torch.max(x, 2), PoseR.py:30) */
    /* Column ranges let ONE launch serve several layers that read the same rows (conv_5 and the three
     * head conv1 layers all read `feat`): per-column slope, colmax only for the first cm_cols columns,
     * C stored only for columns >= c_col0 (at C[m * ldc + n - c_col0]). */
    const float *slope_vec;            /* [N] per-column leaky slope, overrides `slope`; may be NULL */
    int cm_cols;                       /* 0 = all N columns */
    int c_col0;
    /* batch > 1: `batch` independent problems of the same shape in one launch; operand b starts at
     * A + b*batch_stride_a etc. (element strides; vec = bias/scale/shift/slope_vec). */
    int batch;
    int64_t batch_stride_a, batch_stride_w, batch_stride_c, batch_stride_vec, batch_stride_colmax;
    /* Optional: W pre-split into three bf16 terms by tgp_split_bf16 ([batch*N][ldws/16][3][16], ldws % 16 == 0).  When
     * given, launches that are large enough for the tile kernels run on the bf16 matrix cores with the
     * 3-term operand split (six MFMA terms, fp32 accumulate): fp32-level accuracy at 2.67x fewer MFMA cycles.
     * W itself is still required (small launches and the skinny kernel read it). */
    const uint16_t *W_split; int ldws;
    /* 0: W_split holds the three bf16 planes of tgp_split_bf16 (any fp32-range operands);
     * 1: the two fp16 planes of tgp_split_f16 ([batch*N][ldws/16][2][16]) -- three MFMA terms instead of six, products
     *    accurate to ~2^-22; requires |A|, |W| < 65504 (an out-of-range operand yields inf/NaN, never a wrong finite
     *    number) */
    int w_split_kind;
    /* Backward GEMMs on the fp16 split (w_split_kind 1 only; all three optional).  Gradients lie far below fp16's range, so
     * the caller picks a power of two s with max|A| * s < 65504 (tgp_absmax_scale): A is multiplied by *a_scale before it is
     * split, the accumulated sums by *c_scale (= 1 / s, or 1 when a later pass unscales) before the epilogue.  Both are
     * device pointers, so the scale of a step is chosen on the device and the step stays capturable in a graph.
     * ksplit_chunk > 0: split-K -- the `batch` problems are consecutive K-chunks of ONE problem: problem b reads
     * A[:, b*chunk : b*chunk + K], the W_split K-tiles from b*chunk on, and writes C + b*batch_stride_c (partial sums the
     * caller adds up, tgp_sum_slabs); chunk % 16 == 0, operands zero-padded so that every chunk holds K columns. */
    const float *a_scale, *c_scale;
    int ksplit_chunk;
    /* Gathered residuals (tile kernels only, M > 32): C[m, n] gains gres1[gidx1[m], n] + gres2[gidx2[m], n] before the
     * per-object bias and the BatchNorm fold.  This is how a layer over `feat` = [fine | up(coarse1) | up(coarse2)] (nearest
     * upsampling, FaceRecon.py:70-75) runs factored: W_coarse x coarse features is computed once per COARSE point (4x / 16x
     * fewer rows) and each point fetches the rows of its nearest coarse points, instead of multiplying upsampled copies.
     * gidx: int32 per output row, absolute row of gres; excludes res1 / res2; rows_per_obj >= 64 with rowbias / colmax. */
    const float *gres1; int ldg1; const int32_t *gidx1;
    const float *gres2; int ldg2; const int32_t *gidx2;
    /* 0: the library picks the epilogue form (LDS-staged 16-byte form when every operand is 16-byte addressable).
     * 1: the register-direct form (4-byte accesses; same arithmetic in the same order, bit-identical results; slower --
     *    a per-call A/B handle for tests, not a tuning knob; not available with gathered residuals). */
    int epilogue;
    /* Predicate (may be NULL): a device int; while it is 0 the launch does nothing (every workgroup returns at once).  For
     * repair launches whose condition is raised on the device -- tgp_heads_fused's overflow flag -- without a host read.
     */
    const int *pred;
    /* (ABI 4) The launch covers rows [row_base, row_base + M) of a batch whose objects are rows_per_obj rows each: the
     * per-object bias and the max over an object's points address object (row + row_base) / rows_per_obj.  A / C / residual /
     * gather-index pointers are those of the launch's first row, as always.  0 for whole-batch launches.  (The eval forward
     * hands the last rows of the heads' layers to the tile kernels beside the fused kernel: engine.wide_gemm_factored.) */
    int row_base;
    /* (ABI 5) Operands and results as BLOCKED fp16 planes.  A row-major fp32 matrix X (rows, K) in this form is
     * tgp_planes_bytes(rows, K) bytes: for each block of 32 rows and each 16-wide K-tile a 2 KB chunk
     *     [plane: hi | lo][h = (k % 16) / 8][r = row % 32][8 fp16],      x = hi + lo to ~2^-23 (the split of tgp_split_f16),
     * chunks ordered [row block][K-tile], `kt` K-tiles per row block (>= ceil(K / 16); rows and columns past the end are zero).
     * One plane of a chunk is 1 KB in the lane order of v_mfma_f32_32x32x16_f16's operand, so the kernels stage it by LDS-DMA and
     * never convert in their K loop.
     *   A_planes + W_planes (both or neither): the launch runs on the pre-split kernel (csrc/gemm_pp.hip); batch == 1, no
     *     a_scale / ksplit_chunk; every epilogue operand 16-byte addressable.  Results are bit-identical to the launch without
     *     them (same products, same order).  a_amax (may be NULL = unguarded): per 32-row block of A the bits of its largest
     *     magnitude, as tgp_planes_split / a producing launch's c_amax wrote them: a tile whose blocks hold a magnitude
     *     >= 65504 (or a NaN), or nothing >= 2^-4, is computed in exact fp32 from A / W, which are still required then.
     *   C_planes: the result (the columns >= c_col0) is ALSO written as planes, column c_col0 landing at plane column cp_col0
     *     (% 16 == 0), N - c_col0 a multiple of 16; c_amax (zero-filled by the caller once per tensor) receives the per-block
     *     magnitudes.  Tile kernels with the 16-byte epilogue only (the launch is refused otherwise).  C may be NULL.
     * pp_config: 0 = the library picks the tile shape; 1 .. 7 force one (tests, tuning scripts). */
    const void *A_planes; int a_kt; const uint32_t *a_amax;
    const void *W_planes; int w_kt;
    void *C_planes; int c_kt; int cp_col0; uint32_t *c_amax;
    int pp_config;
    /* (ABI 5) Planes-only tensors.  With C_planes, C and colmax_keys may both be NULL: the result exists as planes only.  A launch
     * whose A_planes come without the fp32 operand (A == NULL) cannot recompute a tile that its magnitude words put outside fp16's
     * comfortable range: it computes the tile on the split anyway and sets *range_flag = 1 (device int, zeroed by the caller;
     * required then) -- the caller follows the chain with its fp32 form predicated on that flag (`pred`), as tgp_heads_fused's
     * callers do.  A chain of layers whose activations feed only the next GEMM (the decoder, FaceRecon.py:112-117) then writes and
     * reads 4 bytes per element instead of 8 + 4. */
    int *range_flag;
    /* (ABI 6) The per-object vector layers (M <= 32, the weight-streaming kernel; refused elsewhere), FaceRecon.py:145-165:
     *   a_keys != 0: A holds the order-preserving max keys a colmax_keys launch left (uint32), decoded as they are loaded -- no
     *     decode launch; a_wrap > 0 (% 4 == 0): column k of the operand is key k % a_wrap, i.e. cat((max, max), 1)
     *     (FaceRecon.py:148) is never built.
     *   C_sigmoid != NULL: 1 / (1 + exp(-v)) of every stored value is also written there (row stride ldc): the PH predictor's
     *     nn.Sigmoid (FaceRecon.py:156,162) without a launch of its own. */
    int a_keys; int a_wrap;
    float *C_sigmoid;
} tgp_gemm_args;

/* W (rows, K) fp32, row stride ld -> out[rows][ldo/16][3][16] bf16: per 16-wide K-tile the hi, mid and lo terms
 * (x = hi+mid+lo to 2^-24) stored back to back; columns K..ldo-1 zero.  3*rows*ldo elements.  Done once per
 * weight version. */
int tgp_split_bf16(const float *W, int rows, int K, int ld, uint16_t *out, int ldo, tgp_stream_t stream);
/* Same for the two-term fp16 split: out[rows][ldo/16][2][16] (hi, lo), 2*rows*ldo elements. */
int tgp_split_f16(const float *W, int rows, int K, int ld, uint16_t *out, int ldo, tgp_stream_t stream);

/* (ABI 5) bytes of the blocked fp16 planes of a (rows, K) matrix (tgp_gemm_args.A_planes), and the split itself: X (rows, K) fp32
 * with row stride ld -> out (kts >= ceil(K / 16) K-tiles per row block; padding rows / columns zero); amax (may be NULL; ceil(rows /
 * 32) words, zero-filled by the caller) receives atomicMax of the bits of |x| per row block.  Weights at pack time; activations
 * whose producer does not write planes itself. */
int64_t tgp_planes_bytes(int64_t rows, int K);
int tgp_planes_split(const float *X, int rows, int K, int ld, void *out, int kts, uint32_t *amax, tgp_stream_t stream);
/* the same into a column range of a wider planes buffer: X's K columns land at plane columns col0 .. col0 + K - 1 (col0 % 16 == 0;
 * only those ceil(K / 16) K-tiles of each row block are written, the last one zero padded) */
int tgp_planes_split_cols(const float *X, int rows, int K, int ld, void *out, int kts, int col0, uint32_t *amax, tgp_stream_t stream);
/* tgp_gather_rows and tgp_planes_split in one pass: dst[b][p][0..C) = src[b][idx[b][p]][0..C) (dst may be NULL) and the planes of
 * the gathered rows' first K columns (columns K.. of the planes zero); kts * 16 >= C when dst is given. */
int tgp_planes_gather(const float *src, int lds, const int32_t *idx, int B, int n_src, int n_out, int K, int C, float *dst, int ldd,
                      void *out, int kts, uint32_t *amax, tgp_stream_t stream);

/* nn.Conv1d(kernel 1) / nn.Linear on channel-last rows with the fused epilogue above. */
int tgp_gemm_f32(const tgp_gemm_args *args, tgp_stream_t stream);

/* keys (rows, N) row stride ldk -> out[r, n] (row stride ldo) and, if out2 != NULL, a second copy
 * (FaceRecon.py:145-147 concatenates the pooled vector with itself). */
int tgp_colmax_decode(const uint32_t *keys, int ldk, int rows, int N, float *out, int ldo, float *out2,
                      tgp_stream_t stream);

/* PoseNet9D.py:50 feat_global.max(2): out[b,c] = max_i x[b,i,c], x (B,n,C) row stride ld. */
int tgp_colmax(const float *x, int ld, int B, int n, int C, float *out, tgp_stream_t stream);

/* FaceRecon.py:156,162 nn.Sigmoid on a contiguous vector. */
int tgp_sigmoid(const float *x, float *y, int64_t count, tgp_stream_t stream);

/* PoseNet9D.py:57-66: green/red (B,4) -> unit axes p_* (B,3) and confidences f_* (B); ts (B,6) + mean
 * -> Pred_T (B,3), Pred_s (B,3).  ldg / ldr / ldt: row strides of green / red / ts (ABI 4: the batched head tail writes the three
 * heads' outputs as rows of one (3, B, 8) buffer, read here in place). */
int tgp_head_post(const float *green, const float *red, const float *ts, int ldg, int ldr, int ldt, const float *mean, int B,
                  float *p_green, float *p_red, float *f_green, float *f_red, float *pred_T, float *pred_s, tgp_stream_t stream);

/* (ABI 6) The three pose heads after their max over points as ONE launch (PoseR.py:37-43, PoseTs.py:43-49 in eval mode, then
 * PoseNet9D.py:57-66): keys2 (3, B, 256) the order-preserving max keys of conv2's output (heads rot_green, rot_red, ts);
 * w3t (3, 256, 256) conv3's weights TRANSPOSED (input channel major); b3 / scale3 / shift3 (3, 256) bias and BatchNorm fold;
 * w4 (3, 8, 256) conv4 with its 4 / 4 / 6 rows zero-padded to 8, b4 (3, 8); outputs as tgp_head_post's; raw (may be NULL)
 * receives conv4's (3, B, 8) outputs. */
int tgp_pose_tail(const uint32_t *keys2, const float *w3t, const float *b3, const float *scale3, const float *shift3,
                  const float *w4, const float *b4, const float *mean, int B, float *p_green, float *p_red, float *f_green,
                  float *f_red, float *pred_T, float *pred_s, float *raw, tgp_stream_t stream);

/* (ABI 6) A per-point layer with n_out <= 4 outputs whose rows leave in another order -- the decoder's last conv
 * (FaceRecon.py:117) behind the row sort of the factored layers: out[obj, map[obj, i], j] = bias[j] + sum_k x[obj, i, k] W[j, k]
 * for the `rows` rows of x (row stride ld, K % 4 == 0), objects of rows_per_obj rows; map (int64, relative to the object; NULL =
 * identity); out (rows, n_out) contiguous.  TGP_EUNSUPPORTED for other n_out. */
int tgp_rows_out(const float *x, int ld, int64_t rows, int K, const float *W, int ldw, const float *bias, int n_out,
                 const int64_t *map, int rows_per_obj, float *out, tgp_stream_t stream);
/* (ABI 7) the same, predicated: the launch returns at once while *pred == 0 (pred may be NULL: always runs) -- the last step of a repair
 * chain (tgp_gemm_args.pred) whose result replaces tgp_dec_fused's rows */
int tgp_rows_out_pred(const float *x, int ld, int64_t rows, int K, const float *W, int ldw, const float *bias, int n_out,
                      const int64_t *map, int rows_per_obj, float *out, const int *pred, tgp_stream_t stream);

/* PoseNet9D.py:71 recon + mean, in place on recon (B,n,3). */
int tgp_add_mean(float *recon, const float *mean, int B, int n, tgp_stream_t stream);

/* ---- training-mode BatchNorm / dropout ------------------------------------------------------- */

/* nn.BatchNorm1d in train mode on channel-last rows x (rows, C) with row stride ld: per-channel batch mean and
 * BIASED variance (deterministic: one pass of shifted sums merged in chunk order when the rows are 16-byte addressable, two
 * passes otherwise).  workspace: tgp_bn_workspace_floats(rows, C) floats. */
int64_t tgp_bn_workspace_floats(int64_t rows, int C);
int tgp_bn_stats(const float *x, int ld, int64_t rows, int C, float *mean, float *var, float *workspace,
                 tgp_stream_t stream);
/* tgp_bn_stats that also moves nn.BatchNorm1d's buffers (ABI 4): run_mean / run_var (C) <- (1 - momentum) x running + momentum x
 * batch value, the variance with the unbiased factor rows / (rows - 1) (torch's `running.mul_(1 - m).add_(batch, alpha = m)`);
 * batches (one int64, may be NULL) += 1 (num_batches_tracked).  run_mean == run_var == NULL: statistics only. */
int tgp_bn_stats_running(const float *x, int ld, int64_t rows, int C, float *mean, float *var, float *workspace, float *run_mean,
                         float *run_var, float momentum, int64_t *batches, tgp_stream_t stream);

/* y = (x - mean) / sqrt(var + eps) * gamma + beta, then leaky-relu (act = 1; per-column slope_vec overrides slope);
 * out may alias x or be NULL; colmax_keys as in tgp_gemm_args (max over each object's rows_per_obj rows for the
 * first cm_cols columns). */
int tgp_bn_apply(const float *x, int ld, int64_t rows, int C, const float *mean, const float *var,
                 const float *gamma, const float *beta, float eps, int act, float slope, const float *slope_vec,
                 float *out, int ldo, uint32_t *colmax_keys, int ldcm, int cm_cols, int rows_per_obj,
                 tgp_stream_t stream);

/* nn.Dropout(p) in train mode with the keep mask (uint8 0/1) supplied by the host: y = keep ? x / (1 - p) : 0. */
int tgp_dropout_apply(const float *x, const uint8_t *keep, float p, int64_t count, float *y, tgp_stream_t stream);

/* ---- Chamfer distance ------------------------------------------------------------------------ */

/* chamfer_3D.forward (losses/chamfer3D/chamfer_cuda.cpp:17-19,31; kernels chamfer3D.cu:12-152).
 * xyz1 (B,n,3), xyz2 (B,m,3) -> dist1 (B,n), idx1 (B,n) nearest in xyz2; dist2 (B,m), idx2 (B,m)
 * nearest in xyz1.  Squared distance (dx*dx + dy*dy) + dz*dz in fp32, lowest index on ties. */
int tgp_chamfer_fwd(const float *xyz1, const float *xyz2, int B, int n, int m, float *dist1, float *dist2,
                    int32_t *idx1, int32_t *idx2, tgp_stream_t stream);

/* chamfer_3D.backward (chamfer_cuda.cpp:22-27,32; chamfer3D.cu:155-195): ACCUMULATES into
 * gradxyz1 (B,n,3) / gradxyz2 (B,m,3) (the caller zero-fills, dist_chamfer_3D.py:56-60).
 * Deterministic: contributions are summed in the serial order of the reference's CPU loop
 * (tools/pyTorchChamferDistance/chamfer_distance.cpp:140-175), no atomics. */
int tgp_chamfer_bwd(const float *xyz1, const float *xyz2, int B, int n, int m, const float *graddist1,
                    const float *graddist2, const int32_t *idx1, const int32_t *idx2, float *gradxyz1,
                    float *gradxyz2, tgp_stream_t stream);

/* ---- density-aware Chamfer loss (config 3: forward + Chamfer loss) ------------------------------ */

/* calc_dcd after the Chamfer search (losses/TDA_loss_sym_recon.py:427-445): per object
 * loss = mean_i(1 - exp(-alpha d1_i) w1_i) + 0.5 mean_j(1 - exp(-alpha d2_j) w2_j),
 * w1_i = frac_21 / (count1[idx1_i]^n_lambda + 1e-6) with count1 = bincount(idx1) (w2 likewise, frac_12).
 * loss (B); w1 (B,n), w2 (B,m) are optional outputs kept for the backward pass. n, m <= 4096. */
int tgp_dcd_fwd(const float *dist1, const float *dist2, const int32_t *idx1, const int32_t *idx2, int B, int n,
                int m, float alpha, float n_lambda, int non_reg, float *loss, float *w1, float *w2,
                tgp_stream_t stream);

/* Gradient of the above w.r.t. the distances (the weights are detached, :433,438): graddist1 (B,n), graddist2 (B,m)
 * from gloss (B); they feed tgp_chamfer_bwd. */
int tgp_dcd_bwd(const float *dist1, const float *dist2, const float *w1, const float *w2, const float *gloss,
                int B, int n, int m, float alpha, float *graddist1, float *graddist2, tgp_stream_t stream);

/* TDA_loss.R_DCD pose normalisation (:326-339): R from the predicted axes p_g, p_r (B,3) and confidences f_g, f_r
 * (B) -- for objects with sym[b*sym_ld] == 1 the green axis is paired with column 0 of the true rotation gR (B,3,3)
 * -- then out[b,i] = (R^T (points[b,i] - p_t[b])) * p_s[b].  R_out (B,3,3) optional. */
int tgp_canonicalize(const float *points, const float *gR, const float *p_g, const float *f_g, const float *p_r,
                     const float *f_r, const float *p_t, const float *p_s, const float *sym, int sym_ld, int B,
                     int n, float *out, float *R_out, tgp_stream_t stream);

/* Pose assembly of the evaluater: generate_RT([p_green, p_red], [f_green, f_red], T, mode='vec', sym)
 * (evaluater/RT_TDA_Evaluater.py:94; tools/rot_utils.py:95-98 to_R_matrices).  rt (B,4,4); sym may be NULL. */
int tgp_generate_rt(const float *p_green, const float *p_red, const float *f_green, const float *f_red,
                    const float *T, const float *sym, int sym_ld, int B, float *rt, tgp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Backward pass of the dense per-point layers (what torch autograd runs for nn.Conv1d(k=1) / nn.Linear /
 * nn.BatchNorm1d(train) / F.relu / F.leaky_relu / torch.max in loss.backward(), trainer/RL_TDA.py:205-224).
 * ------------------------------------------------------------------------------------------------------------------ */

/* C (N, K) (+)= A^T B over the rows: C[n][k] = sum_m A[m][n] * B[m][k]  (weight gradient dW = dx^T a).
 * workspace: tgp_gemm_tn_workspace_floats(rows, N, K) floats.  Row slices are summed in slice order (deterministic). */
int64_t tgp_gemm_tn_workspace_floats(int64_t rows, int N, int K);
int tgp_gemm_tn_f32(const float *A, int lda, const float *B, int ldb, int64_t rows, int N, int K, float *C, int ldc,
                    int accumulate, float *workspace, tgp_stream_t stream);

/* out[c] (+)= sum over rows of dy[r][c] (bias gradient).  workspace: tgp_bw_workspace_floats(rows, C). */
int64_t tgp_bw_workspace_floats(int64_t rows, int C);
int tgp_colsum(const float *dy, int lddy, int64_t rows, int C, float *out, int accumulate, float *workspace, tgp_stream_t stream);

/* y = act(BN_train(x)) backward: dz = dy * act'(z); dgamma = sum dz*xhat; dbeta = sum dz;
 * dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)).  x is the raw layer output the forward normalised, mean / var
 * its batch statistics (tgp_bn_stats).  dx may alias dy.  workspace: tgp_bw_workspace_floats(rows, C). */
int tgp_bn_bwd(const float *dy, int lddy, const float *x, int ld, int64_t rows, int C, const float *mean, const float *var,
               float eps, const float *gamma, const float *beta, int act, float slope, const float *slope_vec, float *dx,
               int lddx, float *dgamma, float *dbeta, float *workspace, uint32_t *absmax_bits, tgp_stream_t stream);
/* absmax_bits (round 3; NULL = not wanted): tgp_bn_bwd_absmax_words(rows, C) words, one per workgroup of the apply pass, that
 * receive the bit patterns of max |dx| -- the number the next layer's backward needs for its fp16 scale
 * (tgp_absmax_scale_from_bits over these words instead of tgp_absmax_scale's pass over dx).  Collected by the
 * 16-byte form only (C % 4 == 0, strides % 4 == 0, 16-byte aligned operands): TGP_EUNSUPPORTED otherwise. */
int64_t tgp_bn_bwd_absmax_words(int64_t rows, int C);
int tgp_absmax_scale_from_bits(const uint32_t *bits, int n, float target, float *out, tgp_stream_t stream);

/* The same for a layer followed by a max over each object's points: dpool (objects, C) is the gradient of the pooled
 * vector, argrow (objects, C) the winning global row from tgp_colmax_arg; dx (objects*rows_per_obj, C) is dense. */
int tgp_bn_bwd_pooled(const float *dpool, int ldp, const int *argrow, int lda, const float *x, int ld, int objects,
                      int rows_per_obj, int C, const float *mean, const float *var, float eps, const float *gamma,
                      const float *beta, int act, float slope, const float *slope_vec, float *dx, int lddx, float *dgamma,
                      float *dbeta, tgp_stream_t stream);

/* out[o][c] = max over the object's n rows of act(BN(x)) (mean == NULL: of x), argrow[o][c] = the winning global row
 * (first on ties, as torch.max). */
int tgp_colmax_arg(const float *x, int ld, int objects, int n, int C, const float *mean, const float *var, float eps,
                   const float *gamma, const float *beta, int act, float slope, const float *slope_vec, float *out, int ldo,
                   int *argrow, int lda, tgp_stream_t stream);

/* out[o][c] = sum over object o's n rows of dy (objects*n, C): gradient of a per-object bias broadcast over the object's points
 * (gcn3d.py:108-112 ORL_forward's global half; FaceRecon.py:165); fixed summation order. */
int tgp_colsum_objects(const float *dy, int ld, int objects, int n, int C, float *out, int ldo, tgp_stream_t stream);

/* Backward of tgp_colmax_arg without BatchNorm (PoseNet9D.py:50 feat_global = feat.max over points): dx (objects*rows_per_obj, C)
 * = dpool[o][c] on the recorded winning row, 0 elsewhere; written densely, deterministic. */
int tgp_colmax_bwd(const float *dpool, int ldp, const int *argrow, int lda, int objects, int rows_per_obj, int C, float *dx, int lddx,
                   tgp_stream_t stream);

/* dst (cols, rows) = src (rows, cols)^T. */
int tgp_transpose(const float *src, int ld_src, int rows, int cols, float *dst, int ld_dst, tgp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Backward pass of the graph layers (the `_bwd` entry points of SURVEY.md section 8b, seam #2).  Scatters use hardware
 * float atomics (as the reference's CUDA autograd does for index / max backward); d* outputs that are scattered into
 * must be zero-initialised (or hold a gradient to accumulate onto) by the caller.
 * ------------------------------------------------------------------------------------------------------------------ */

/* HSlayer_surface.graph_conv (gcn3d.py:91-106): dg (B,n,C) -> dsdn (3, S*C), the gradient w.r.t. the unit support
 * directions (F.normalize's own backward stays with the caller).  workspace: tgp_gconv_bwd_workspace_floats(B, n, C). */
int64_t tgp_gconv_bwd_workspace_floats(int B, int n, int C);
int tgp_gconv_surface_bwd(const float *xyz, const int32_t *idx, const float *sdn, const float *dg, int ldg, int B, int n, int k,
                          int S, int C, float *dsdn, float *workspace, tgp_stream_t stream);

/* HS_layer.graph_conv (gcn3d.py:157-180): proj (B,n,>=8C) = [centre | 7 support blocks] as in tgp_gconv_hs_fwd;
 * dproj (same layout) += d centre, d support (scatter to the arg-max neighbour); dsdn (3, S*C) as above. */
int tgp_gconv_hs_bwd(const float *xyz, const int32_t *idx, const float *proj, int ldp, const float *sdn, const float *dg, int ldg,
                     int B, int n, int k, int S, int C, float *dproj, int lddp, float *dsdn, float *workspace, tgp_stream_t stream);

/* y[b][p][c] = max_j src[b][idx[b][p][j]][c], p < n_rows (get_ORL_global's and Pool_layer's neighbour max):
 * dsrc[b][arg][c] += dy.  per_object != 0: dy is (B, C) and every row receives dy[b][c] * scale (the mean over the
 * points that follows in get_ORL_global: scale = 1 / n); else dy is (B*n_rows, C). */
int tgp_nbrmax_bwd(const float *src, int ld_src, const int32_t *idx, int B, int n_src, int n_rows, int k, int C, const float *dy,
                   int lddy, int per_object, float scale, float *dsrc, int ld_dsrc, tgp_stream_t stream);

/* tgp_gather_rows backward: dsrc[b][idx[b][p]][:] += dy[b][p][:]. */
int tgp_gather_rows_bwd(const float *dy, int lddy, const int32_t *idx, int B, int n_src, int n_out, int C, float *dsrc,
                        int ld_dsrc, tgp_stream_t stream);

/* Differentiable half of TDA_loss.R_DCD's canonicalisation (losses/TDA_loss_sym_recon.py:334-337):
 * out[b][i] = (R[b]^T (points[b][i] - t[b])) * s[b]; R (B,3,3) row-major, t, s (B,3).  The backward returns dpoints (may be
 * NULL), dR, dt, ds (per-object sums over the n points, fixed order). */
int tgp_pose_transform_fwd(const float *points, const float *R, const float *t, const float *s, int B, int n, float *out,
                           tgp_stream_t stream);
int tgp_pose_transform_bwd(const float *points, const float *R, const float *t, const float *s, const float *dout, int B, int n,
                           float *dpoints, float *dR, float *dt, float *ds, tgp_stream_t stream);

/* ---- backward GEMMs on the fp16 operand split: scale selection, scaled transposes, K-split partial sums ----------------- */

/* out = {s, 1/s, max|x|} on the device: s = 2^k, the largest power of two with max|x| * s <= target (s = 1 for an all-zero
 * tensor).  x (rows, cols) row stride ld; workspace: 2048 uint32. */
int tgp_absmax_scale(const float *x, int ld, int64_t rows, int cols, float target, uint32_t *workspace, float *out,
                     tgp_stream_t stream);
/* dst (cols, rows_pad) = (src (rows, cols) * *scale)^T as fp32 (scale may be NULL), columns rows.. zero; rows_pad % 4 == 0,
 * dst 16-byte aligned */
int tgp_transpose_scaled(const float *src, int ld_src, int rows, int cols, const float *scale, float *dst, int rows_pad,
                         tgp_stream_t stream);
/* the same, written as the fp16 hi / lo planes of tgp_split_f16: dst [cols][rows_pad / 16][2][16]; rows_pad % 16 == 0 */
int tgp_transpose_split_f16(const float *src, int ld_src, int rows, int cols, const float *scale, uint16_t *dst, int rows_pad,
                            tgp_stream_t stream);
/* out[i] (+)= *scale * sum_z parts[z * n + i], z ascending (scale may be NULL) */
/* (round 3) W (rows, cols) -> dst_f32 (cols, rows_pad) = W^T zero padded and dst_split = its fp16 hi / lo planes
 * [cols][rows_pad / 16][2][16], rows_pad % 16 == 0: the weight operands of the backward's dx GEMM in one pass */
int tgp_transpose_both(const float *src, int ld_src, int rows, int cols, float *dst_f32, uint16_t *dst_split, int rows_pad,
                       tgp_stream_t stream);
int tgp_sum_slabs(const float *parts, int Z, int64_t n, const float *scale, float *out, int accumulate, tgp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * The regression terms of the training loss either side of Chamfer (losses/TDA_loss_sym_recon.py, losses/consistency_loss.py).
 * The reference loops over the batch in Python with one host synchronisation per object and term (`if sym[i, 0] == 1`);
 * these entries make the symmetry tests on the device, one launch per bundle, so the loss is capturable in a HIP graph.
 * sym: (B, sym_ld) int32, column 0 = symmetric about y, column 1.. = reflection flags (datasets' sym_info).
 * ------------------------------------------------------------------------------------------------------------------ */

/* TDA_loss.forward's eight small terms (:42-86 via cal_loss_Rot1 :223, cal_cosine_dis :243, cal_loss_Rot2 :227,
 * cal_cosine_dis_sym :247, cal_rot_regular_angle :265, cal_loss_Tran :285, cal_loss_Size :288, cal_loss_R_con :205), UNWEIGHTED:
 * out[0..7] = Rot1, Rot1_cos, Rot2, Rot2_cos, Rot_regular, Tran, Size, R_con; out[8] = number of objects with sym0 != 1.
 * rot1, rot2, tran, size and their targets are (B,3); f1, f2 (B) the predicted confidences.  kind 0 = nn.L1Loss,
 * 1 = nn.SmoothL1Loss(beta) (FLAGS.fsnet_loss_type, :20-35). */
int tgp_pose_terms_fwd(const float *rot1, const float *rot2, const float *f1, const float *f2, const float *tran, const float *size,
                       const float *g_rot1, const float *g_rot2, const float *g_tran, const float *g_size, const int32_t *sym,
                       int sym_ld, int B, int kind, float beta, float *out, tgp_stream_t stream);
/* gradient of sum_k gw[k] out[k] (gw: 8 floats on the device) w.r.t. the six predictions; fwd_out = the forward's out */
int tgp_pose_terms_bwd(const float *rot1, const float *rot2, const float *f1, const float *f2, const float *tran, const float *size,
                       const float *g_rot1, const float *g_rot2, const float *g_tran, const float *g_size, const int32_t *sym,
                       int sym_ld, int B, int kind, float beta, const float *fwd_out, const float *gw, float *d_rot1, float *d_rot2,
                       float *d_f1, float *d_f2, float *d_tran, float *d_size, tgp_stream_t stream);

/* prop_sym_matching_loss (losses/consistency_loss.py:19-48 = TDA_loss_sym_recon.py:120-148): L1 between the reconstruction
 * PC_re (B,N,3) and the cloud PC (B,N,3) mapped by the object's symmetry in the ground-truth frame (half turn about y /
 * mirror in z / identity, :171-203), mean over B*N*3 -> loss[0].  workspace: tgp_sym_recon_workspace_floats(B, N) floats.
 * The backward writes dPC and / or dPC_re (either may be NULL) for the upstream gradient gloss[0]. */
int64_t tgp_sym_recon_workspace_floats(int B, int N);
int tgp_sym_recon_fwd(const float *PC, const float *PC_re, const float *gt_R, const float *gt_t, const int32_t *sym, int sym_ld,
                      int sym_cols, int B, int N, float *workspace, float *loss, tgp_stream_t stream);
int tgp_sym_recon_bwd(const float *PC, const float *PC_re, const float *gt_R, const float *gt_t, const int32_t *sym, int sym_ld,
                      int sym_cols, int B, int N, const float *gloss, float *dPC, float *dPC_re, tgp_stream_t stream);

/* ph_loss_fn / omega (TDA_loss_sym_recon.py:292-322): out[0] = mean(|a - b| * w), a, b, wsrc (B,D), w = 1 on rows with
 * sum(wsrc) > 0.  The reference's host-side has_nan_or_inf branch (|a - a| instead) is folded in: NaN when a holds a
 * NaN / Inf, 0 when only b does; out[1] = 1 when the plain mean was taken.  rows: 4*B floats kept for the backward, which
 * writes da (B,D) for the upstream gradient gout[0]. */
int tgp_rowl1_fwd(const float *a, const float *b, const float *wsrc, int B, int D, float *rows, float *out, tgp_stream_t stream);
int tgp_rowl1_bwd(const float *a, const float *b, const float *rows, const float *fwd_out, const float *gout, int B, int D, float *da,
                  tgp_stream_t stream);

/* feat_consistency_loss (losses/consistency_loss.py:11-16, unweighted): loss[0] = 2 - 2 sum_b <x1_b/|x1_b|, x2_b/|x2_b|> / B,
 * x1, x2 (B,C), norms clamped at 1e-12 as F.normalize.  rows: 3*B floats kept for the backward (d1 / d2 may be NULL). */
int tgp_feat_consistency_fwd(const float *x1, const float *x2, int B, int C, float *rows, float *loss, tgp_stream_t stream);
int tgp_feat_consistency_bwd(const float *x1, const float *x2, const float *rows, const float *gloss, int B, int C, float *d1, float *d2,
                             tgp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Pairwise metrics of the NOCS pose evaluation (evaluation/eval_utils_v1.py): every (prediction, ground truth) pair of a
 * result set in one launch, double precision as the numpy original.  RT: (P,4,4) row-major, scales: (P,3).
 * ------------------------------------------------------------------------------------------------------------------ */

/* compute_3d_iou_new (:829-887): the reference's IoU of the two posed boxes -- its amax / amin run over axis 0 of the [3, 8]
 * corner array (:842-845), so the overlap is taken over eight per-corner (min, max) coordinate ranges; reproduced as written.
 * The fourth RT row is honoured (transform_coordinates_3d divides by it, :1011).  symmetric[t] != 0 (bottle / bowl / can,
 * mug with a hidden handle) -> the maximum over 20 rotations of box 1 about its own y axis. */
int tgp_iou3d_pairs(const double *RT1, const double *RT2, const double *scales1, const double *scales2, const int *symmetric,
                    int P, double *iou, tgp_stream_t stream);

/* compute_RT_degree_cm_symmetry (:890-963): out[t] = {rotation error in degrees, translation error in cm}.
 * mode[t]: 0 general, 1 symmetric about y (angle between the y axes), 2 symmetric under a half turn about y. */
int tgp_rt_error_pairs(const double *RT1, const double *RT2, const int *mode, int P, double *out, tgp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Input side of the evaluation loader (evaluation/load_data_eval.py): depth image + detection
 * mask + box -> the (n_pts, 3) camera-frame cloud PoseNet9D.forward is fed.  Byte / integer work, HBM-bound.
 * ------------------------------------------------------------------------------------------------------------------ */

/* One detection per workgroup, replacing load_data_eval.py:302-355 (ROI resampling of depth / mask / pixel grid by
 * crop_resize_by_warp_affine with cv2.INTER_NEAREST, tools/dataset_utils.py:80-136; _depth_to_pcl :451-462; /1000; the cut
 * of points within a quarter of the extent's diagonal of point number 25, :341-355).
 *   depth   (I,H,W) uint16 millimetres          masks   the images' (H,W,n_i) byte masks (Mask-RCNN 'pred_masks'), any packing
 *   mask_off[d]  byte offset of detection d's channel within masks; mask_stride[d] = n_i (bytes between neighbouring pixels,
 *                < 128; H*W < 2^24: offsets inside one image's masks are 31-bit)
 *   det_img[d]   image of detection d           window[d] = {cmin+cmax, rmin+rmax, s}: get_bbox's window (tools.eval_utils),
 *                                                s = min(max(rmax-rmin, cmax-cmin), max(H,W)) (:309-316)
 *   camk    (I,4) fx, fy, cx, cy (float32, as the reference's intrinsics :158-161); fx, fy ordinary focal lengths (normal
 *           floats far from overflow / underflow: the quotients are correctly rounded under that assumption)
 *   roi_size     FLAGS.img_size; a power of two in [64, 256] (the fixed-point walk is then exact and a pixel index fits 16 bits)
 *   recs    (D, roi_size^2) uint32 scratch: on return entries [0, counts[d][2]) are detection d's cloud in ROI row-major order
 *           as records (ROI pixel index << 16) | depth; a record and the detection's window / intrinsics determine its point,
 *           which tgp_cloud_select / tgp_cloud_sample materialise (only the sampled points are ever written as floats)
 *   counts  (D,3): depth-valid ROI pixels (:332), valid points (:336), points kept by the cut; the last is -1 when there are
 *           fewer than 26 valid points (the reference raises IndexError at :350). */
int tgp_roi_cloud(const uint16_t *depth, const uint8_t *masks, const int64_t *mask_off, const int *mask_stride, const int *det_img,
                  const int *window, const float *camk, int D, int H, int W, int roi_size, uint32_t *recs, int *counts,
                  tgp_stream_t stream);

/* The same kernel for the TRAINING loader's device half (datasets/load_data.py:235-290, 395-407):
 *   tables    (D, 2, roi_size) int32 or NULL.  The training window comes from aug_bbox_DZI (tools/dataset_utils.py:24-61): a
 *             real-valued centre and scale, for which OpenCV's 10-bit fixed-point walk has no integer closed form.  The host
 *             evaluates the walk once per detection in double (as cv2.warpAffine does): tables[d][0][x] = source column of ROI
 *             column x, tables[d][1][y] = source row of ROI row y (rot = 0, so the map is separable).  With tables, `window` is
 *             not read (may be NULL).
 *   mask_val  (D) int32 or NULL: v > 0 -> a pixel is inside when its mask byte EQUALS v (the ground-truth instance mask
 *             `mask == inst_id`, :245-247, with mask_stride 1 and mask_off = the image's offset); 0 / NULL -> any non-zero byte.
 *   cut_frac  the outlier cut keeps the points farther than cut_frac x the extent's diagonal from point number 25: 0.25 in the
 *             evaluation loader (load_data_eval.py:352), 0.15 in the training loader (load_data.py:283); float32 product.
 * tgp_roi_cloud = tgp_roi_cloud_ex(..., NULL, NULL, 0.25f). */
int tgp_roi_cloud_ex(const uint16_t *depth, const uint8_t *masks, const int64_t *mask_off, const int *mask_stride, const int *det_img,
                     const int *window, const float *camk, int D, int H, int W, int roi_size, uint32_t *recs, int *counts,
                     const int *tables, const int *mask_val, float cut_frac, tgp_stream_t stream);
/* tgp_cloud_select with the detections' source-pixel tables (NULL: the window's closed form). */
int tgp_cloud_select_ex(const uint32_t *recs, const int32_t *sel, const int *det_img, const int *window, const float *camk, int D,
                        int roi_size, int n_pts, float *out, const int *tables, tgp_stream_t stream);

/* _sample_points (:404-417) as a gather with a host-drawn selection: out[d][i] = point(recs[d][sel[d][i]]), out (D,n_pts,3);
 * det_img / window / camk as given to tgp_roi_cloud.  An index outside [0, roi_size^2) produces NaNs, never a fault.
 * With sel[d] = 0..n-1 it materialises a cloud's first n points. */
int tgp_cloud_select(const uint32_t *recs, const int32_t *sel, const int *det_img, const int *window, const float *camk, int D,
                     int roi_size, int n_pts, float *out, tgp_stream_t stream);

/* The same resampling drawn on the device (no read-back of counts): the first n_pts elements of a keyed pseudo-random
 * permutation of each cloud (4-round Feistel bijection, cycle-walked); clouds with at most n_pts points are tiled exactly
 * as :411-412.  Not the draw np.random would make -- a documented deviation for throughput runs.  Rows of detections
 * whose count is <= 0 are NaN. */
int tgp_cloud_sample(const uint32_t *recs, const int *counts, const int *det_img, const int *window, const float *camk, int D,
                     int roi_size, int n_pts, uint64_t seed, float *out, tgp_stream_t stream);

/* Row order of the factored wide layers (this repo's engine; no reference counterpart): per object, the n fine rows sorted by
 * key = near2 * n1 + near1 with ties in point order -- torch.argsort(key, stable=True) -- and the sorted parents as global
 * row numbers: order / order64 (B,n) the permutation (int32 for tgp_gather_rows, int64 for torch scatter), near1_out = sorted
 * near1 + b*n1, near2_out = sorted near2 + b*n2.  n <= 2048, n1*n2 <= 2^21; near1 in [0,n1), near2 in [0,n2). */
int tgp_sort_by_parent(const int32_t *near1, const int32_t *near2, int B, int n, int n1, int n2, int32_t *order, int64_t *order64,
                       int32_t *near1_out, int32_t *near2_out, tgp_stream_t stream);

/* The pose heads' conv1 -> BatchNorm(eval) -> ReLU -> conv2 -> BatchNorm(eval) -> ReLU -> max over each object's points as one
 * kernel (PoseR.py:26-36 Rot_green / Rot_red, PoseTs.py:31-42 Pose_Ts; eval mode), on the factored form of conv1 (this repo's
 * engine): conv1(feat)[m] = W_fine . fine[m] + p1[idx1[m]] + p2[idx2[m]].  The (M, heads * 1024) activation is never written.
 *   fine (M, ldf) fp32, K = its live columns (256 < K <= 272 = ldf's first 17 K-tiles);
 *   wa_planes (ABI 7) = the heads' conv1 weights over fine, (heads * 1024) rows x 272 columns, as BLOCKED fp16 planes
 *   (tgp_planes_split with kts = 17: the layout of tgp_gemm_args.W_planes; one 32-channel block is 34 contiguous KB);
 *   p1 / p2: rows of per-coarse-point products, columns head * 1024 + channel at the pointers given, row strides ldp1 / ldp2;
 *   idx1 / idx2 (M) rows of p1 / p2 per point;
 *   w2p (ABI 7) = tgp_heads_pack_w2(conv2 weights, conv1 bias and BatchNorm fold): tgp_heads_w2_bytes(heads) bytes;
 *   bias2 / scale2 / shift2 (heads * 256); keys (heads, B, 256) order-preserving keys of the maxima (tgp_colmax_decode), zeroed by
 *   the caller; M = B * rows_per_obj. */
typedef struct tgp_heads_fused_args {
    const float *fine; int ldf; int K;
    const void *wa_planes;
    const float *p1; int ldp1; const int32_t *idx1;
    const float *p2; int ldp2; const int32_t *idx2;
    const void *w2p;
    const float *bias2; const float *scale2; const float *shift2;
    uint32_t *keys;
    int M; int rows_per_obj; int B; int heads;
    /* fp16 range: a wave whose fine features or conv1 activations leave fp16's range (a magnitude that rounds to infinity, a NaN: its
     * conv2 sums are then non-finite), or whose fine features all lie under 2^-4, writes no keys for its 32 points and sets
     * *overflow = 1 (device int, zeroed by the caller; NULL: such waves write whatever keys their sums give -- NaN keys are loud).
     * The caller then runs the two-launch form predicated on the flag (tgp_gemm_args.pred), which recomputes every key in guarded
     * arithmetic. */
    int *overflow;
    /* (ABI 4) rows > 0: only the first `rows` rows (a multiple of 128) of every head are processed here; the caller supplies the
     * keys of rows [rows, M) through tgp_gemm_f32 (row_base = rows), merging into the same key buffer.  The grid is
     * heads x 128-point workgroups, one per CU and round. */
    int rows;
    /* (ABI 5) fine_planes (may be NULL): the same features as blocked fp16 planes (tgp_gemm_args.A_planes' layout, fine_kt >= 17
     * K-tiles per row block, columns K.. zero; fine_amax: the per-row-block magnitude words, may be NULL): the kernel then loads its
     * points' operand fragments as 1 KB runs instead of splitting fp32 rows -- same bits.  `fine` is still required (the repair). */
    const void *fine_planes; int fine_kt; const uint32_t *fine_amax;
} tgp_heads_fused_args;
int tgp_heads_fused(const tgp_heads_fused_args *args, tgp_stream_t stream);
/* conv -> BatchNorm(eval) -> LeakyReLU -> max over each object's points of a factored layer whose activation only feeds the max
 * (FaceRecon.py:76-77 conv_5 and the `feat.max(1)` that follows, on the factored form of this repo's engine): operands as for
 * tgp_heads_fused with one "head" of C channels (C % 32 == 0; wa_planes / bias / scale / shift / p1 / p2 point at its first channel),
 * slope = the LeakyReLU slope, 0 <= slope <= 1; keys (B, C) row stride ldk zeroed by the caller; p1_rows / p2_rows: rows of p1 / p2
 * (p1_rows * ldp1 and p2_rows * ldp2 below 2^30: rows are addressed with 32-bit byte offsets); overflow as in tgp_heads_fused (the
 * caller's repair is tgp_gemm_f32 with `pred`).  A NaN or an infinity among an object's activations gives that (object, channel) a
 * NaN's key. */
typedef struct tgp_conv_max_fused_args {
    const float *fine; int ldf; int K;
    const void *wa_planes;     /* (ABI 7) the layer's (C, K) weight as blocked fp16 planes with 17 K-tiles: tgp_planes_split, the layout of tgp_gemm_args.W_planes */
    const float *p1; int ldp1; int p1_rows; const int32_t *idx1;
    const float *p2; int ldp2; int p2_rows; const int32_t *idx2;
    const float *bias; const float *scale; const float *shift; float slope;
    uint32_t *keys; int ldk;
    int M; int rows_per_obj; int C;
    int *overflow;
    const void *fine_planes; int fine_kt; const uint32_t *fine_amax;      /* (ABI 5) as in tgp_heads_fused_args */
} tgp_conv_max_fused_args;
int tgp_conv_max_fused(const tgp_conv_max_fused_args *args, tgp_stream_t stream);
/* (ABI 7) w2 (heads, 256, 1024) fp32 and the heads' conv1 bias / BatchNorm scale / shift (heads * 1024 each) -> the kernel's operand:
 * per (head, block of 32 conv1 channels) 33 KB = conv2's fp16 hi / lo planes in MFMA fragment order (K permuted to the order the
 * conv1 accumulators leave the channels in) + the block's bias | scale | shift.  out: tgp_heads_w2_bytes(heads) bytes, 16-byte aligned.
 * w2 == NULL: only the vectors are rewritten, in place (the BatchNorm fold follows the running statistics; the weights do not). */
int64_t tgp_heads_w2_bytes(int heads);
int tgp_heads_pack_w2(const float *w2, const float *bias1, const float *scale1, const float *shift1, int heads, void *out,
                      tgp_stream_t stream);

/* (ABI 7) An HS layer's last GEMM and the NEXT layer's projection GEMM as one kernel (gcn3d.py:87-89 / 108-112 / 147-155 followed by
 * gcn3d.py:170), csrc/hs_chain.hip, for the two shapes of Face_Enc without a pooling in between: (K1 in 129..144, N1 = 128, N2 = 1152)
 * and (K1 in 241..256, N1 = 256, N2 = 2304); other shapes: -1 from tgp_hs_chain_pack_bytes, an argument error from the launches.
 *   c1 = act(bn(A W1^T + rowbias[object] + res1 + res2)) (M, N1), the tile kernel's epilogue order; c2 = c1 W2^T + bias2 (M, N2).
 *   a_planes / a_kt / a_amax: A (M, K1) as blocked fp16 planes + magnitude words (tgp_gemm_args.A_planes' layout; a_kt >= ceil(K1 / 16));
 *   units = tgp_hs_chain_pack(W1 (N1, K1), W2 (N2, N1)); rowbias (M / rows_per_obj, N1) or NULL; res1 / res2 (M, N1) or NULL; scale1 /
 *   shift1 (N1) both or neither; relu != 0: ReLU; c1 also as planes (c1_planes, c1_kt K-tiles per row block, first K-tile c1_kt0) with
 *   magnitude words c1_amax (all may be NULL);
 *   flag: device int, zeroed by the caller, raised when the operand or the intermediate leaves the fp16 split's range (>= 65504, NaN, or
 *   a 32-row block wholly below 2^-4): c1 / c2 are then not to be trusted and the caller's two tgp_gemm_f32 launches follow with
 *   tgp_gemm_args.pred = flag.  While the flag stays 0 both results equal those two launches' on the same planes bit for bit. */
typedef struct tgp_hs_chain_args {
    const void *a_planes; int a_kt; const uint32_t *a_amax;
    int M; int K1; int N1; int N2;
    const void *units;
    const float *rowbias; int ldrb; int rows_per_obj;
    const float *res1; int ldr1;
    const float *res2; int ldr2;
    const float *scale1; const float *shift1;
    int relu;
    float *c1; int ldc1;
    void *c1_planes; int c1_kt; int c1_kt0; uint32_t *c1_amax;
    const float *bias2;
    float *c2; int ldc2;
    int *flag;
} tgp_hs_chain_args;
int64_t tgp_hs_chain_pack_bytes(int K1, int N1, int N2);
int tgp_hs_chain_pack(const float *w1, int ld1, int K1, int N1, const float *w2, int ld2, int N2, void *out, tgp_stream_t stream);
int tgp_hs_chain(const tgp_hs_chain_args *args, tgp_stream_t stream);

/* (ABI 7) C = A W^T (+ bias; NULL: none) for the projection shapes of the HS layers (gcn3d.py:170: K = 128 or 256, N % 128 == 0, e.g.
 * N = 9 x width) and the coarse products of the factored wide layers (K = 512),
 * csrc/hs_chain.hip hs_proj_kernel: the operand from its planes (a_planes / a_kt / a_amax as tgp_gemm_args.A_planes), a workgroup keeps
 * its 128 rows' fragments for all columns and streams the weights (units = tgp_proj_pack(W (N, K))) once.  Same bits as tgp_gemm_f32 on
 * the same planes, except in a 128-row tile that the fp16 range rule (a magnitude >= 65504 / NaN, or nothing >= 2^-4) sends to the exact
 * path: computed here from a (M, K, row stride lda) and w (N, K, ldw) by fp32 fma chains in ascending k.  a / w may be NULL when a_amax is. */
typedef struct tgp_proj_planes_args {
    const void *a_planes; int a_kt; const uint32_t *a_amax;
    const float *a; int lda;
    int M; int K; int N;
    const void *units;
    const float *w; int ldw;
    const float *bias;
    float *c; int ldc;
} tgp_proj_planes_args;
int64_t tgp_proj_pack_bytes(int K, int N);
int tgp_proj_pack(const float *w, int ld, int K, int N, void *out, tgp_stream_t stream);
int tgp_proj_planes(const tgp_proj_planes_args *args, tgp_stream_t stream);

/* (ABI 7) The decoder behind its first conv as ONE kernel (FaceRecon.py:105-117 Face_Dec in eval mode: conv 512 -> 512, 512 -> 256,
 * 256 -> 128, each + BatchNorm + ReLU, then conv 128 -> 3), csrc/dec_fused.hip: a wave owns 32 points for the whole chain, no activation
 * between the layers leaves its registers.
 *   h1_planes: the first conv's activation (M, 512) as blocked fp16 planes (tgp_gemm_args.C_planes of that launch, h1_kt >= 32 K-tiles
 *   per row block); h1_amax: its per-32-row-block magnitude words (may be NULL);
 *   units = tgp_dec_pack(W2 (512, 512), W3 (256, 512), W4 (128, 256)) (tgp_dec_pack_bytes() bytes, 16-byte aligned): the three weights
 *   as fp16 hi / lo MFMA fragments in staging order, W3 / W4 with K permuted to the order the accumulators hold the channels in;
 *   vec[l][0 / 1 / 2]: bias / BatchNorm scale / shift of layer l + 2 (512, 256, 128 floats; read at every launch: they follow a refold);
 *   w5 (3, 128) and b5 (3): the last conv; map (may be NULL) as tgp_rows_out's: out row of row i = (i / rows_per_obj) * rows_per_obj + map[i];
 *   out (M, 3);
 *   flag: device int (zeroed by the caller; required): raised when the operand's magnitude words or any sum of the chain leave fp16's
 *   range (or the operand's block lies wholly under 2^-4): `out` is then not to be trusted and the caller's predicated fp32 chain
 *   (tgp_gemm_args.pred) redoes the layers.  Layer 2's sums equal tgp_gemm_f32's on the same planes bit for bit; layers 3, 4 and the
 *   last conv add the same products in another order (agreement to rounding). */
typedef struct tgp_dec_fused_args {
    const void *h1_planes; int h1_kt; const uint32_t *h1_amax;
    const void *units;
    const float *vec[3][3];
    const float *w5; const float *b5;
    const int64_t *map; int rows_per_obj;
    float *out;
    int *flag;
    int M;
} tgp_dec_fused_args;
int tgp_dec_fused(const tgp_dec_fused_args *args, tgp_stream_t stream);
int64_t tgp_dec_pack_bytes(void);
/* h1_permuted: the operand's planes hold the channels in accumulator order (tgp_dec_l1's output) instead of the natural one
 * (tgp_gemm_args.C_planes): W2's K order is then permuted like W3's / W4's. */
int tgp_dec_pack(const float *w2, const float *w3, const float *w4, int h1_permuted, void *out, tgp_stream_t stream);
/* (ABI 7) The decoder's FIRST conv (FaceRecon.py:112, factored form: W_fine . fine[m] + p1[idx1[m]] + p2[idx2[m]] + rowbias[object],
 * BatchNorm(eval) + ReLU; 512 channels) in the fused heads kernel's style, writing the activation as the operand tgp_dec_fused loads:
 *   fine_planes (M rows, fine_kt >= 17 K-tiles) / fine_amax: the points' features as blocked planes; wa_planes: the conv's weights over
 *   fine (512, 272) as blocked planes with 17 K-tiles; p1 / p2 / idx1 / idx2 as tgp_heads_fused (pointers at the conv's first channel);
 *   bias / scale / shift (512); rowbias (may be NULL): (objects, >= 512) per-object bias, row stride ldrb, objects of rows_per_obj rows;
 *   h1_planes (M rows x 512, h1_kt >= 32 K-tiles) receives fp16 hi / lo planes whose K order is the ACCUMULATOR order (slot 8 h + t of
 *   K-tile 2 b + s = channel 32 b + 16 s + 8 (t >> 2) + 4 h + (t & 3)): consumed by tgp_dec_fused with tgp_dec_pack(h1_permuted = 1), not by
 *   tgp_gemm_f32; h1_amax: per-32-row-block magnitude words (zeroed by the caller; may be NULL); flag as tgp_dec_fused's.  Values equal
 *   the tile kernel's for the same layer bit for bit. */
typedef struct tgp_dec_l1_args {
    const void *fine_planes; int fine_kt; const uint32_t *fine_amax;
    const void *wa_planes;
    const float *p1; int ldp1; const int32_t *idx1;
    const float *p2; int ldp2; const int32_t *idx2;
    const float *bias; const float *scale; const float *shift;
    const float *rowbias; int ldrb; int rows_per_obj;
    void *h1_planes; int h1_kt; uint32_t *h1_amax;
    int *flag;
    int M;
} tgp_dec_l1_args;
int tgp_dec_l1(const tgp_dec_l1_args *args, tgp_stream_t stream);

/* ---- the factored wide layers in TRAINING (ABI 4; this repo's engine, no reference counterpart: the reference multiplies the
 * up-sampled copies, FaceRecon.py:70-77) -----------------------------------------------------------------------------------------
 * The layers that read the concat buffer run as  W_fine x fine[i] + P1[near1(i)] + P2[near2(i)]  (tgp_gemm_args.gres1 / gres2); the
 * backward of the two fetches is  d P[r] = sum of d Y[i] over the points i whose nearest coarse point is r.
 * tgp_child_lists inverts near (B, n) (parent of every point: ids in [0, R), or b*R + that when global_ids) into CSR child lists: ptr (B*R + 1) int32,
 * idx (B*n) int32 global rows b*n + i, children in point order.  R <= 4096, n <= 8192.
 * tgp_segsum_rows: out[r][:C] = sum of g[idx[k]][:C] over k in [ptr[r], ptr[r+1]) in list order -- no atomics, bit-repeatable.
 * C, ldg, ldo multiples of 4 and g, out 16-byte aligned (16-byte accesses), or multiples of 2 and 8-byte aligned (8-byte accesses). */
int tgp_child_lists(const int32_t *near, int B, int n, int R, int global_ids, int32_t *ptr, int32_t *idx, tgp_stream_t stream);
int tgp_segsum_rows(const float *g, int ldg, int C, const int32_t *ptr, const int32_t *idx, int R, float *out, int ldo,
                    tgp_stream_t stream);

/* ---- scatter-free backward of the graph layers (ABI 4; csrc/graph_bwd.hip) ---------------------------------------------------------
 * Same gradients as tgp_nbrmax_bwd / tgp_gconv_hs_bwd (the reference's autograd: max backward + index_add, gcn3d.py:157-180,210-245)
 * without float atomics: the neighbour lists are inverted once (tgp_reverse_graph), pass 1 stores every (row, channel)'s winning slot,
 * pass 2 sums per SOURCE row over its reverse list.  Outputs are written densely (no zero fill by the caller) and are bit-repeatable.
 * tgp_reverse_graph: idx (B, n_rows, k) int32 ids in [0, n_src) -> rptr (B*n_src + 1) int32, rent (B*n_rows*k) int32 = (row << 6) | slot,
 * each list ascending.  k <= 64, n_src <= 8192, (n_rows*k + 2*n_src) * 4 <= 150 KB (one workgroup sorts an object in LDS), else
 * TGP_EUNSUPPORTED.
 * tgp_nbrmax_bwd_gather: arg_ws B*n_rows*C bytes; C % 4 == 0, C / 4 a divisor of 256, strides % 4 == 0, 16-byte aligned, else
 * TGP_EUNSUPPORTED.  tgp_gconv_hs_bwd_gather: workspace as tgp_gconv_bwd_workspace_floats; arg_ws B*n*7C bytes, contrib_ws B*n*7C
 * floats; C in {128, 256, 512}, k <= 63, 16-byte aligned operands, else TGP_EUNSUPPORTED; have_slots: arg_ws was filled by
 * tgp_gconv_hs_fwd_slots (below) and the slot pass is skipped. */
int tgp_reverse_graph(const int32_t *idx, int B, int n_rows, int k, int n_src, int32_t *rptr, int32_t *rent, tgp_stream_t stream);
int tgp_nbrmax_bwd_gather(const float *src, int ld_src, const int32_t *idx, const int32_t *rptr, const int32_t *rent, int B, int n_src,
                          int n_rows, int k, int C, const float *dy, int lddy, int per_object, float scale, uint8_t *arg_ws,
                          float *dsrc, int ld_dsrc, tgp_stream_t stream);
int tgp_gconv_hs_bwd_gather(const float *xyz, const int32_t *idx, const int32_t *rptr, const int32_t *rent, const float *proj, int ldp,
                            const float *sdn, const float *dg, int ldg, int B, int n, int k, int S, int C, float *dproj, int lddp,
                            float *dsdn, float *workspace, uint8_t *arg_ws, float *contrib_ws, int have_slots, tgp_stream_t stream);
/* tgp_gconv_hs_fwd (same output, bit for bit) that also records the slots (B*n*7C bytes) for tgp_gconv_hs_bwd_gather(have_slots = 1):
 * the training forward then needs no slot pass in the backward.  Same shape limits as tgp_gconv_hs_bwd_gather. */
int tgp_gconv_hs_fwd_slots(const float *xyz, const int32_t *idx, const float *proj, int ldp, const float *sdn, int B, int n, int k, int S,
                           int C, float *out, int ldo, uint8_t *slots, tgp_stream_t stream);

/* ---- the rotation R_DCD canonicalises with, and its gradient (ABI 4; csrc/poserot.hip) --------------------------------------------
 * losses/TDA_loss_sym_recon.py:327-333 (get_vertical_rot_vec_in_batch :370-395 for the symmetric and the general case,
 * get_rot_mat_y_first :351-360) in one launch.  gR0 (B, 3) = g_R[:, :, 0]; p_g, p_r (B, 3) predicted axes; f_g, f_r (B) their
 * confidences; sym0 (B) = sym[:, 0] as float.  R (B, 3, 3) row-major; J (B, 9, 8) = d R / d (p_g, f_g, p_r, f_r), evaluated by
 * forward-mode differentiation of the same formulas.  tgp_pose_rotation_bwd: din (B, 8) = J^T dR per object. */
int tgp_pose_rotation_fwd(const float *gR0, const float *p_g, const float *f_g, const float *p_r, const float *f_r, const float *sym0,
                          int B, float *R, float *J, tgp_stream_t stream);
int tgp_pose_rotation_bwd(const float *dR, const float *J, int B, float *din, tgp_stream_t stream);
/* backward of tgp_head_post (PoseNet9D.py:57-66): dgreen (B, 4), dred (B, 4), dts (B, 6) contiguous, from the gradients of the six
 * outputs (contiguous; NULL = zero). */
int tgp_head_post_bwd(const float *green, const float *red, int ldg, int ldr, int B, const float *g_pg, const float *g_pr,
                      const float *g_fg, const float *g_fr, const float *g_T, const float *g_s, float *dgreen, float *dred, float *dts,
                      tgp_stream_t stream);

/* ---- dW = dy^T x on the fp16 split without transposed copies (ABI 4; csrc/gemm_tn_split.hip) --------------------------------------
 * a = dy (rows, N), b = x (rows, K), both row-major; scale = tgp_absmax_scale(a) ({s, 1/s, ...} on the device).  The reduction over
 * the rows is cut into Z chunks of `chunk` rows (Z * chunk >= rows); parts (Z, N, K) receives their partial products, which
 * tgp_sum_slabs adds in order and unscales.  N, K, lda, ldb multiples of 4 and 16-byte aligned operands, else TGP_EUNSUPPORTED
 * (the caller then takes the transposed-copy form: tgp_transpose_scaled + tgp_transpose_split_f16 + tgp_gemm_f32). */
int tgp_gemm_tn_split(const float *a, int lda, const float *b, int ldb, int rows, int N, int K, const float *scale, int Z, int chunk,
                      float *parts, tgp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TGPOSE_H */
