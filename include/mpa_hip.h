/*
 * mpa_hip.h — C ABI of libmpa_hip.so, the MI355X (gfx950) native operator library for the
 * multi-part-assembly training hot path.
 *
 * Every entry point takes raw DEVICE pointers, plain sizes and a HIP stream handle
 * (`void* stream` == hipStream_t; NULL = the legacy default stream), returns 0 on success or a
 * negative MPA_E* code, never allocates or frees user-visible memory, never synchronises the
 * device, and keeps no mutable global state besides a thread-local last-error string
 * (mpa_last_error).  All launches are asynchronous on `stream`.
 *
 * Each block below cites the reference interface (Wuziyi616/multi_part_assembly, file:line) that
 * the entry point replaces; INTEGRATION.md shows the reference-side binding.
 */
#ifndef MPA_HIP_H_
#define MPA_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPA_OK 0
#define MPA_EINVAL (-1)  /* bad argument (null pointer, negative size, size overflow) */
#define MPA_ELAUNCH (-2) /* hipGetLastError() reported a launch failure */

/* ABI version of this header; bumped whenever a signature changes. */
#define MPA_ABI_VERSION 8
int mpa_abi_version(void);

/* Thread-local, NUL-terminated description of the last failure on this thread ("" if none). */
const char* mpa_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Chamfer distance — replaces the `chamfer_cuda` extension module
 *   chamfer_forward : multi_part_assembly/utils/chamfer/cuda/chamfer.cpp:21
 *                     -> ChamferForward   chamfer_kernel.cu:116-168 (kernel :32-95)
 *   chamfer_backward: multi_part_assembly/utils/chamfer/cuda/chamfer.cpp:22
 *                     -> ChamferBackward  chamfer_kernel.cu:224-289 (kernel :175-210)
 *
 * Layout: xyz1 [batch, n1, 3], xyz2 [batch, n2, 3] contiguous fp32 (AoS), as CHECK_INPUT
 * (chamfer_kernel.cu:20-22) demands.  Outputs dist1 [batch, n1], dist2 [batch, n2] fp32 and
 * idx1, idx2 int64 — the dtypes ChamferForward allocates (:129-132).
 *
 * Semantics (bit-exact contract, see DESIGN.md "Chamfer arithmetic"):
 *   d(p,q) = ((px-qx)*(px-qx) + (py-qy)*(py-qy)) + (pz-qz)*(pz-qz), every operation rounded to
 *   fp32, no fused multiply-add; dist1[b,i] = min_j d(xyz1[b,i], xyz2[b,j]) and idx1[b,i] the LOWEST
 *   j attaining it (strict `<` scan in index order, chamfer_kernel.cu:82); a query with no
 *   candidate below 1e32 (n2 == 0, NaN/huge input) gets dist = 1e32f, idx = -1 (:60-61).
 *   dist2/idx2: the same with the roles of the clouds swapped.
 *
 * Three searches stand behind the call, with identical results (tests/test_chamfer_gpu.py, tests/test_gate_gpu.py):
 *   - the exhaustive scan: n1 * n2 pair evaluations per sample and direction (what the reference does);
 *   - the matrix-core gated search (csrc/gate_nn.hip) for clouds of a few hundred to a few thousand points — the per-part
 *     call of rot_points_cd_loss, [B*P, N, 3] against itself (utils/loss.py:113-138): one bf16 matrix instruction per
 *     32 x 32 pairs bounds every pair, the pinned fp32 arithmetic answers from the few candidates that can win.  No
 *     workspace.  Chosen when min(n1, n2) >= 192 and the grid-pruned search below is not;
 *   - an exact grid-pruned search (csrc/grid_nn.hip) for large clouds — the whole-shape call of shape_cd_loss,
 *     [32, 20000, 3] against itself (utils/loss.py:173-199), is 2.56e10 pair evaluations exhaustively and a few
 *     dozen candidates per query pruned.  It needs scratch memory, which the CALLER provides (the library never
 *     allocates): `workspace` = mpa_chamfer_workspace() bytes, 16-byte aligned, contents irrelevant before and
 *     after the call (0 bytes for the sizes the exhaustive scan answers anyway; mpa_chamfer_workspace_variant sizes
 *     a pinned search of mpa_chamfer_forward_variant).  With workspace == NULL (or too small) every size is answered
 *     by the exhaustive scan.
 *   The pruned search is chosen when min(n1, n2) >= 512 and n1 * n2 >= 9e6.  Samples that hold a non-finite
 *   coordinate, or one beyond 1e15 in magnitude, are always answered by the exhaustive scan.
 * ---------------------------------------------------------------------------------------------- */
int mpa_chamfer_workspace(int64_t batch, int64_t n1, int64_t n2, int64_t* bytes);
int mpa_chamfer_workspace_variant(int64_t batch, int64_t n1, int64_t n2, int variant, int64_t* bytes);
int mpa_chamfer_forward(const float* xyz1, const float* xyz2, int64_t batch, int64_t n1, int64_t n2,
                        float* dist1, int64_t* idx1, float* dist2, int64_t* idx2, void* workspace,
                        int64_t workspace_bytes, void* stream);

/* Diagnostic twin of mpa_chamfer_forward that pins the search (all bit-identical in their results):
 * 0 = direct compare/select per pair, 1 = fused-form gate + exact recheck (fastest on tie-free clouds),
 * 2 = exact chunk-minimum scan (the exhaustive default: insensitive to duplicated points), 3 = the grid-pruned
 * search at ANY size (needs the workspace), 4 = the matrix-core gated search (any n1, n2 in 1 .. 32768; no workspace),
 * -1 = by size as mpa_chamfer_forward does. */
int mpa_chamfer_forward_variant(const float* xyz1, const float* xyz2, int64_t batch, int64_t n1,
                                int64_t n2, float* dist1, int64_t* idx1, float* dist2,
                                int64_t* idx2, int variant, void* workspace, int64_t workspace_bytes,
                                void* stream);

/*
 * grad_xyz1 [batch, n1, 3] and grad_xyz2 [batch, n2, 3] are OVERWRITTEN (the library zero-fills
 * them itself, as ChamferBackward does with at::zeros, chamfer_kernel.cu:252-253):
 *   grad_xyz1[b,i]        += 2*grad_dist1[b,i] * (xyz1[b,i] - xyz2[b,idx1[b,i]])
 *   grad_xyz2[b,idx1[b,i]] -= the same vector                 (and symmetrically for dist2/idx2)
 * The scatter half uses fp32 hardware atomics, so the summation ORDER of colliding contributions
 * is unspecified — exactly the reference's behaviour (chamfer_kernel.cu:203-208).
 * Indices outside [0, n) (the -1 of an empty search) contribute nothing.
 */
int mpa_chamfer_backward(const float* grad_dist1, const float* grad_dist2, const float* xyz1,
                         const float* xyz2, const int64_t* idx1, const int64_t* idx2, int64_t batch,
                         int64_t n1, int64_t n2, float* grad_xyz1, float* grad_xyz2, void* stream);

/* Double-precision twins (the reference dispatches float and double, chamfer_kernel.cu:145,156,
 * and its own gradcheck runs in double, utils/chamfer/test_chamfer.py:92-101).  Same contract with
 * every fp32 replaced by fp64 (initial distance 1e32 as a double). */
int mpa_chamfer_forward_f64(const double* xyz1, const double* xyz2, int64_t batch, int64_t n1,
                            int64_t n2, double* dist1, int64_t* idx1, double* dist2, int64_t* idx2,
                            void* stream);
int mpa_chamfer_backward_f64(const double* grad_dist1, const double* grad_dist2, const double* xyz1,
                             const double* xyz2, const int64_t* idx1, const int64_t* idx2,
                             int64_t batch, int64_t n1, int64_t n2, double* grad_xyz1,
                             double* grad_xyz2, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Rigid pose application — replaces the tensor-op chain of
 *   rot_pc / transform_pc : multi_part_assembly/utils/transforms.py:199-244
 *   qrot / qtransform     : multi_part_assembly/utils/transforms.py:75-109
 *   (pytorch3d.transforms.quaternion_apply underneath, transforms.py:87)
 *
 * pc [num_parts, num_points, 3] fp32, quat [num_parts, 4] (real part first, NOT normalised here),
 * trans [num_parts, 3] or NULL (rotation only), mask [num_parts] or NULL: where mask[m] == 0 every
 * point of part m is replaced by (fill, fill, fill) BEFORE the transform — the padded-part fill of
 * shape_cd_loss (utils/loss.py:173-175).  out [num_parts, num_points, 3].
 * Arithmetic: the two Hamilton products of quaternion_apply evaluated term by term, left to right,
 * no FMA — bit-identical to the reference CPU path.
 * ---------------------------------------------------------------------------------------------- */
int mpa_pose_apply_forward(const float* pc, const float* quat, const float* trans,
                           const float* mask, float fill, int64_t num_parts, int64_t num_points,
                           float* out, void* stream);

/* Backward of the above: given grad_out [num_parts, num_points, 3] writes grad_quat [num_parts, 4],
 * grad_trans [num_parts, 3] (skipped if NULL) and grad_pc [num_parts, num_points, 3] (skipped if
 * NULL; zero for masked parts).  Deterministic (fixed reduction tree, no atomics). */
int mpa_pose_apply_backward(const float* grad_out, const float* pc, const float* quat,
                            const float* mask, float fill, int64_t num_parts, int64_t num_points,
                            float* grad_quat, float* grad_trans, float* grad_pc, void* stream);

/* Rotation3D's constructor rule (multi_part_assembly/utils/rotation.py:115-126): quaternions [count, 4] whose
 * norm is <= 0.5 (zero padding) are replaced by the identity (1,0,0,0); keep [count] receives 1 where the input
 * was kept (the gradient gate).  One launch instead of norm + compare + where. */
int mpa_quat_sanitize(const float* quat, int64_t count, float* out, float* keep, void* stream);

/* The weighting of the loss terms for one stochastic sample (multi_part_assembly/models/modules/base_model.py:348-387:
 * `loss = sum_k w_k * mean_b term_k[b]`): terms [K, B] row-major, weights [K] -> means [K] (what the reference logs per
 * term) and loss [1], in ONE launch instead of mean + dot; fixed reduction order.  1 <= K <= 64.
 * Backward: grad_terms [K, B] = (grad_loss[0] * w_k + grad_means[k]) / B; grad_loss and grad_means may each be NULL. */
int mpa_loss_reduce_forward(const float* terms, const float* weights, int64_t K, int64_t B, float* means, float* loss,
                            void* stream);
int mpa_loss_reduce_backward(const float* grad_loss, const float* grad_means, const float* weights, int64_t K, int64_t B,
                             float* grad_terms, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused geometric-assembly loss — replaces, for the geometric datasets, the loss half of
 *   BaseModel._calc_loss : multi_part_assembly/models/modules/base_model.py:240-314
 * i.e. trans_l2_loss, rot_cosine_loss, rot_points_l2_loss, rot_points_cd_loss, shape_cd_loss
 * (utils/loss.py:22-35,59-202) together with their rot_pc / transform_pc / chamfer_distance calls and
 * the autograd graph behind them.  Same values as composing mpa_pose_apply_* and mpa_chamfer_*,
 * without touching padded slots: padded target parts enter the whole-shape search as ONE
 * representative point (all their points coincide after the 1e3 fill), padded query points are
 * skipped (the loss multiplies their distances by zero).
 *
 * part_pcs [B,P,N,3]; valids [B,P] (1/0); quat_* [B,P,4] real-first, already passed through
 * Rotation3D's zero-quaternion rule; trans_* [B,P,3].  P <= 64.
 * losses [5,B]: 0 trans_loss, 1 rot_pt_cd_loss, 2 transform_pt_cd_loss, 3 rot_loss, 4 rot_pt_l2_loss
 * (names of base_model.py:283-298), `training` selects the whole-shape normalisation of
 * utils/loss.py:185-198.  The caller provides the workspaces sized by mpa_assembly_loss_workspace
 * and keeps them untouched until the backward call.  The four transformed clouds are the first
 * 4*B*P*N*3 floats of float_ws: rot(pred), rot(gt), transform(pred), transform(gt), each [B,P,N,3];
 * with fill_pad_points != 0 the padded parts of the last two are filled completely (the `ret_pts`
 * outputs of shape_cd_loss), otherwise only their first point is defined.
 * ---------------------------------------------------------------------------------------------- */
int mpa_assembly_loss_workspace(int64_t B, int64_t P, int64_t N, int64_t* float_elems,
                                int64_t* int_elems);
int mpa_assembly_loss_forward(const float* part_pcs, const float* valids, const float* quat_pred,
                              const float* trans_pred, const float* quat_gt, const float* trans_gt,
                              int64_t B, int64_t P, int64_t N, int training, int fill_pad_points,
                              float* float_ws, int32_t* int_ws, float* losses, void* stream);
/* Profiling twin: identical launches; `events` (may be NULL) is a host array of 7 hipEvent_t recorded on
 * `stream`: [0] start, [1] after pose kernel, [2] after per-part Chamfer, [3] after the whole-shape Chamfer
 * phase, [4] after finalize, [5]/[6] right before/after the whole-shape search kernel itself — so a benchmark can
 * time the dominant kernel inside its timed region.  Individual entries may be NULL (e.g. only [5] and [6] set: two
 * records per call instead of seven). */
int mpa_assembly_loss_forward_timed(const float* part_pcs, const float* valids, const float* quat_pred,
                                    const float* trans_pred, const float* quat_gt, const float* trans_gt,
                                    int64_t B, int64_t P, int64_t N, int training, int fill_pad_points,
                                    float* float_ws, int32_t* int_ws, float* losses, void* const* events,
                                    void* stream);
/* The spatial structure behind both Chamfer searches of the loss.  Every cloud the loss searches — rot_pc / transform_pc of
 * part_pcs under the predicted and the ground-truth pose (utils/loss.py:127-129,177-183) — is a rigid image of the same
 * source points, so ONE balanced k-d ordering of each valid part's N points (leaves of 32) serves them all: it depends on
 * part_pcs and valids only, not on any pose.  mpa_assembly_order writes it (order: mpa_assembly_order_elems floats =
 * [B*P][Npad] x (local x, y, z, original index), Npad = the power of two >= max(N, 32); 0 floats and a no-op when
 * N > 2048, where the loss keeps its grid search); mpa_assembly_loss_forward_ordered is mpa_assembly_loss_forward_timed
 * taking that ordering, so a training step orders its batch once and evaluates the loss as often as the model asks
 * (3 GNN iterations, min-of-N samples).  order == NULL: computed into the workspace by the call itself.  The ordering
 * steers speed only — results are the exhaustive scan's, bit for bit, for any permutation.
 * `search` picks the searches behind the two Chamfer terms (identical results): 0 exhaustive scans, 1 exhaustive per-part
 * scan + grid-pruned whole-shape search (rounds 1-4; the default), 2 k-d leaves for both, 3 "auto" = leaves for the per-part
 * term and, per sample on the device, grid or leaves for the whole-shape term; -1 = MPA_SHAPE_SEARCH (brute | grid | leaf
 * | auto) or the default.  The leaf structure wins where parts are many and small; only modes 2 / 3 use `order`. */
int mpa_assembly_order_elems(int64_t B, int64_t P, int64_t N, int64_t* float_elems);
int mpa_assembly_order(const float* part_pcs, const float* valids, int64_t B, int64_t P, int64_t N, float* order,
                       void* stream);
int mpa_assembly_loss_forward_ordered(const float* part_pcs, const float* valids, const float* quat_pred,
                                      const float* trans_pred, const float* quat_gt, const float* trans_gt,
                                      int64_t B, int64_t P, int64_t N, int training, int fill_pad_points,
                                      const float* order, int search, float* float_ws, int32_t* int_ws, float* losses,
                                      void* const* events, void* stream);
/* grad_losses [5,B] = d(objective)/d(losses); writes grad_quat [B,P,4] and grad_trans [B,P,3] of the
 * PREDICTED pose.  Deterministic (no atomics). */
int mpa_assembly_loss_backward(const float* grad_losses, const float* part_pcs, const float* valids,
                               const float* quat_pred, const float* trans_pred, const float* quat_gt,
                               const float* trans_gt, int64_t B, int64_t P, int64_t N, int training,
                               const float* float_ws, const int32_t* int_ws, float* grad_quat,
                               float* grad_trans, void* stream);

/* ------------------------------------------------------------------------------------------------
 * PointNet part encoder — replaces
 *   PointNet.forward           : multi_part_assembly/models/modules/encoder/pointnet.py:29-41
 *   _extract_part_feats        : multi_part_assembly/models/pn_transformer/network.py:59-68
 *                                (boolean-mask compaction + scatter; here: mask in, zeros out)
 * 5 x [1x1 conv (no bias) -> BatchNorm1d -> ReLU (none after the last)], widths 3-64-64-64-128-F,
 * max over the N points of every part.  F must be 64, 128 or 256 (the shipped configs use 128 / 256),
 * N <= 32768 points per part (the reference samples 1000 per part and feeds P*N = 20000 to B-Global's shape encoder).
 *
 * points [M,N,3]; valids [M] (1/0): padded parts are skipped everywhere (they do not enter the
 * BatchNorm statistics) and get feat = 0.  conv_w[l] = [C_l, C_{l-1}] row-major (the Conv1d weight
 * with its trailing 1 dropped), bn_w / bn_b / running_mean / running_var [C_l], l = 0..4 — HOST
 * arrays of 5 DEVICE pointers each.  training != 0: batch statistics (biased variance) and running
 * statistics updated in place with `momentum` (unbiased variance), else running statistics.
 * feat [M,F].  Workspaces sized by mpa_pointnet_workspace must stay untouched until backward.
 * A call in which NO part is valid returns zero features, leaves the running statistics untouched and its backward
 * writes zero gradients (the reference's BatchNorm would refuse the empty batch).
 * ---------------------------------------------------------------------------------------------- */
int mpa_pointnet_workspace(int64_t M, int64_t N, int64_t F, int64_t* float_elems, int64_t* int_elems);
int mpa_pointnet_forward(const float* points, const float* valids, const float* const* conv_w,
                         const float* const* bn_w, const float* const* bn_b,
                         float* const* running_mean, float* const* running_var, int training,
                         float momentum, float eps, int64_t M, int64_t N, int64_t F, float* float_ws,
                         int32_t* int_ws, float* feat, void* stream);
/* Training-mode backward: grad_feat [M,F] -> grad_conv_w[l] [C_l, C_{l-1}], grad_bn_w[l], grad_bn_b[l]
 * (host arrays of 5 device pointers; every buffer is overwritten).  No gradient w.r.t. the points.
 * Deterministic: two-stage reductions, no atomics. */
int mpa_pointnet_backward(const float* grad_feat, const float* points, const float* valids,
                          const float* const* conv_w, const float* const* bn_w, int64_t M, int64_t N,
                          int64_t F, float* float_ws, const int32_t* int_ws, float* const* grad_conv_w,
                          float* const* grad_bn_w, float* const* grad_bn_b, void* stream);

/* bf16 PERFORMANCE VARIANT of the PointNet encoder (csrc/pointnet_bf16.hip) — separately named, never the default,
 * outside every parity claim (the reference's own precision switch: AMP runs the encoder's convolutions in half
 * precision and keeps the Chamfer loss in fp32, multi_part_assembly/utils/chamfer/chamfer.py:14).  Same arguments
 * and masking as mpa_pointnet_forward / _backward above; the convolution outputs are STORED in bf16, the GEMMs are
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation, BatchNorm statistics / affine / every reduction and all returned
 * gradients are fp32 and deterministic.  One workspace (`bytes` from mpa_pointnet_workspace_bf16, 256-byte aligned)
 * that must stay untouched between forward and backward. */
int mpa_pointnet_workspace_bf16(int64_t M, int64_t N, int64_t F, int64_t* bytes);
int mpa_pointnet_forward_bf16(const float* points, const float* valids, const float* const* conv_w,
                              const float* const* bn_w, const float* const* bn_b, float* const* running_mean,
                              float* const* running_var, int training, float momentum, float eps, int64_t M,
                              int64_t N, int64_t F, void* ws, float* feat, void* stream);
int mpa_pointnet_backward_bf16(const float* grad_feat, const float* points, const float* valids,
                               const float* const* conv_w, const float* const* bn_w, int64_t M, int64_t N, int64_t F,
                               void* ws, float* const* grad_conv_w, float* const* grad_bn_w, float* const* grad_bn_b,
                               void* stream);

/* ------------------------------------------------------------------------------------------------
 * DGCNN part encoder, whole forward / backward — replaces
 *   DGCNN.forward        : multi_part_assembly/models/modules/encoder/dgcnn.py:73-109 (global_feat = True)
 *   _extract_part_feats  : multi_part_assembly/models/dgl/network.py:90-99 (boolean-mask compaction + scatter; here:
 *                          mask in, zeros out, the valid parts are counted and compacted on the device)
 * 4 x [kNN (k = 20) -> Conv2d 1x1 over [x_j - x_i ; x_i] -> BatchNorm2d -> LeakyReLU 0.2 -> max_k], widths
 * 3-64-64-128-256, concatenation (512) -> Conv1d 1x1 -> BatchNorm1d -> LeakyReLU 0.2 -> [max ; mean] over the N
 * points -> Linear(2F -> F).  20 <= N <= 1024 points per part, F = 64, 128 or 256.
 *
 * points [M,N,3]; valids [M] (1/0): padded parts are skipped everywhere (they do not enter the BatchNorm statistics)
 * and get feat = 0.  conv_w[l]: the Conv2d / Conv1d weights with their trailing 1s dropped, [64,6] [64,128] [128,128]
 * [256,256] [F,512]; bn_w / bn_b / running_mean / running_var [C_l], l = 0..4 — HOST arrays of 5 DEVICE pointers;
 * fc_w [F,2F], fc_b [F].  training != 0: batch statistics (over all edges of the valid parts), running statistics
 * updated in place; else running statistics.  feat [M,F].
 * kNN indices: arithmetic and tie rule pinned as for mpa_knn_exact below (index-exact against oracle/knn_ref.c).
 * `ws` (mpa_dgcnn_workspace bytes, 256-byte aligned) carries everything backward needs and must stay untouched until
 * then.  `events` (nullable): host array of 8 hipEvent_t, [2l] / [2l+1] recorded on `stream` right before / after the
 * kNN kernels of stage l — lets a benchmark time them inside its timed region.
 * backward (training mode): grad_feat [M,F] -> grad_conv_w[l], grad_bn_w[l], grad_bn_b[l] (host arrays of 5 device
 * pointers), grad_fc_w, grad_fc_b (all overwritten) and, if non-NULL, grad_points [M,N,3] (through the edge features;
 * the neighbour choice itself is not differentiable).  Deterministic: no cross-wave atomics on floats.
 * ---------------------------------------------------------------------------------------------- */
int mpa_dgcnn_workspace(int64_t M, int64_t N, int64_t F, int64_t* bytes);
int mpa_dgcnn_forward(const float* points, const float* valids, const float* const* conv_w,
                      const float* const* bn_w, const float* const* bn_b, float* const* running_mean,
                      float* const* running_var, const float* fc_w, const float* fc_b, int training, float momentum,
                      float eps, int64_t M, int64_t N, int64_t F, void* ws, float* feat, void* const* events,
                      void* stream);
/* The same forward with caller-supplied kNN graphs, and the read-out of the graphs a forward built.  `graphs`: HOST array
 * of 4 DEVICE pointers, one per EdgeConv stage; a non-NULL entry [nv*N, 20] int32 (indices inside each cloud, rows of
 * the nv VALID parts in order) replaces that stage's search, a NULL entry is searched as usual.  The reference rebuilds
 * every graph from the features (dgcnn.py:84-96); this entry point exists so that a parity test can hold the graphs
 * fixed to the reference's own `knn` outputs and attribute what remains of a difference.  mpa_dgcnn_export_graph copies
 * stage `stage`'s (0..3) lists out of a forward's workspace: idx [M*N, 20] int32, rows past the valid parts = -1. */
int mpa_dgcnn_forward_graphs(const float* points, const float* valids, const float* const* conv_w,
                             const float* const* bn_w, const float* const* bn_b, float* const* running_mean,
                             float* const* running_var, const float* fc_w, const float* fc_b, int training,
                             float momentum, float eps, int64_t M, int64_t N, int64_t F, void* ws, float* feat,
                             const int32_t* const* graphs, void* stream);
int mpa_dgcnn_export_graph(const void* ws, int64_t M, int64_t N, int64_t F, int64_t stage, int32_t* idx,
                           void* stream);
int mpa_dgcnn_backward(const float* grad_feat, const float* const* conv_w, const float* const* bn_w,
                       const float* fc_w, int64_t M, int64_t N, int64_t F, void* ws, float* const* grad_conv_w,
                       float* const* grad_bn_w, float* const* grad_bn_b, float* grad_fc_w, float* grad_fc_b,
                       float* grad_points, void* stream);

/* kNN graph with the pinned arithmetic, as used inside mpa_dgcnn_forward — replaces `knn`
 * (multi_part_assembly/models/modules/encoder/dgcnn.py:8-15) for n clouds of N points (20 <= N <= 1024), k = 20.
 * x [n*N, ld] row-major point features (16-byte aligned), the first C columns are used (C = 3: ld = 4 with a zero pad
 * column; C = 64 / 128: any ld >= C, a multiple of 4).  idx [n*N, 20] int32, best first.
 *   C = 3:    dot = fma(x2,y2, fma(x1,y1, x0*y0)), |x|^2 = (x0*x0 + x1*x1) + x2*x2 — the reference's CPU arithmetic,
 *             bit for bit (torch matmul + torch.sum on the fixture cloud);
 *   C >= 64:  dot = fmaf chain over k in the order 0, C/2, 1, C/2+1, ... (the matrix-core chain), |x|^2 likewise;
 *   score = (-|x_j|^2 + 2 dot) - |x_i|^2, every operation rounded to fp32; neighbours = the 20 best by (score
 *   descending, index ascending).  On the reference's own stage inputs this selects exactly the reference's neighbour
 *   sets (tests/golden/dgcnn_graphs.npz).
 * C >= 64 runs as a shortlist search (csrc/dg_knn_fast.h): bf16 matrix-core Gram tiles with a proven error bound select
 * ~21 of the N candidates per point, only those get the pinned fp32 score — same indices as the exhaustive scan, bit for
 * bit.  `ws`: mpa_knn_exact_workspace(n, N) bytes of scratch, 256-byte aligned. */
int mpa_knn_exact_workspace(int64_t n, int64_t N, int64_t* bytes);
int mpa_knn_exact(const float* x, int64_t ld, int64_t n, int64_t N, int64_t C, void* ws, int32_t* idx,
                  void* stream);

/* ------------------------------------------------------------------------------------------------
 * One MLP layer of the graph networks — replaces the Conv1d(k=1) + BatchNorm1d + ReLU layers of
 *   MLP3 / MLP4 / MLP5   : multi_part_assembly/models/dgl/modules.py:5-58, models/rgl_net/modules.py:5-30
 * and the Linear + ReLU layers of
 *   RelationNet          : multi_part_assembly/models/dgl/modules.py:61-73
 * which DGL / RGL-NET run over the B*P*P part pairs (edge MLP, relation weights) and the B*P parts (node MLP) in every
 * GNN iteration (models/dgl/network.py:121-152).
 * x [R, K] row-major (leading dimension ldx >= K), w [N, K] (the Conv1d weight with its trailing 1 dropped / the Linear
 * weight), bias [N] or NULL; gamma == NULL: out = act(x w^T + bias); else out = act(BatchNorm(x w^T + bias)) with
 * batch statistics over the R rows (training != 0; running statistics updated in place) or the running statistics.
 * act = ReLU if relu != 0.  K and N multiples of 64.  out [R, N].  `ws` (mpa_mlp_layer_workspace bytes, 256-byte
 * aligned) carries the pre-normalisation values and the statistics to backward, which takes the forward's `out`
 * (for the ReLU mask) and overwrites grad_x [R, K] (if non-NULL), grad_w [N, K], grad_b [N] (if non-NULL),
 * grad_gamma / grad_beta [N] (BatchNorm layers).  Exact-fp32 matrix-core GEMMs, fixed-order reductions: deterministic.
 * ---------------------------------------------------------------------------------------------- */
int mpa_mlp_layer_workspace(int64_t R, int64_t K, int64_t N, int64_t* bytes);
int mpa_mlp_layer_forward(const float* x, int64_t ldx, const float* w, const float* bias, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, int training, float momentum,
                          float eps, int relu, int64_t R, int64_t K, int64_t N, void* ws, float* out, void* stream);
int mpa_mlp_layer_backward(const float* grad_out, const float* x, int64_t ldx, const float* w, const float* gamma,
                           const float* out, int relu, int64_t R, int64_t K, int64_t N, void* ws, float* grad_x,
                           float* grad_w, float* grad_b, float* grad_gamma, float* grad_beta, void* stream);

/* ------------------------------------------------------------------------------------------------
 * First layer of the P x P edge MLP without the pair tensor — replaces, for the edge MLP's first Conv1d(k=1) + BatchNorm1d
 * + ReLU, the reference's
 *   torch.cat([part_feats.unsqueeze(2).repeat(..), part_feats.unsqueeze(1).repeat(..)], dim=-1) -> MLP3.conv1 / bn1 / relu
 *   (multi_part_assembly/models/dgl/network.py:135-152, models/dgl/modules.py:5-31; RGL-NET: models/rgl_net/network.py:70-88).
 * The layer's input row (s, i, j) is [a[s, i] ; b[s, j]], so x w^T = (a Wa^T + bias)[s, i] + (b Wb^T)[s, j] with
 * Wa | Wb the column halves of w [N, 2F]: two GEMMs over the B*P part rows instead of one over the B*P*P pair rows, the
 * BatchNorm statistics taken over all B*P*P rows as the reference's BatchNorm1d does.
 * a, b [B*P, F] row-major, w [N, 2F], bias [N] or NULL, gamma / beta / running_* [N] (the layer has a BatchNorm);
 * out [B*P*P, N], row (s, i, j) at (s*P + i)*P + j.  F and N multiples of 64.  `ws`: mpa_pair_layer_workspace bytes,
 * 256-byte aligned, carried from forward to backward, which overwrites grad_a / grad_b [B*P, F] (each if non-NULL;
 * grad_b == grad_a — one tensor in both roles, as the networks call it — receives the sum of the two),
 * grad_w [N, 2F], grad_bias [N] (if non-NULL), grad_gamma / grad_beta [N].  Fixed-order reductions: deterministic.
 * ---------------------------------------------------------------------------------------------- */
int mpa_pair_layer_workspace(int64_t B, int64_t P, int64_t F, int64_t N, int64_t* bytes);
int mpa_pair_layer_forward(const float* a, const float* b, const float* w, const float* bias, const float* gamma,
                           const float* beta, float* running_mean, float* running_var, int training, float momentum,
                           float eps, int relu, int64_t B, int64_t P, int64_t F, int64_t N, void* ws, float* out,
                           void* stream);
int mpa_pair_layer_backward(const float* grad_out, const float* a, const float* b, const float* w, const float* gamma,
                            const float* out, int relu, int64_t B, int64_t P, int64_t F, int64_t N, void* ws, float* grad_a,
                            float* grad_b, float* grad_w, float* grad_bias, float* grad_gamma, float* grad_beta,
                            void* stream);

/* ------------------------------------------------------------------------------------------------
 * The small per-iteration pieces of the graph networks between the MLP layers — replace the library element-wise /
 * reduction / 1-column GEMM launches behind
 *   PoseEncoder.mlp1 + ReLU (7 -> 256)                  : multi_part_assembly/models/dgl/modules.py:76-86
 *   RelationNet.mlp3 + sigmoid, times the valid matrix  : models/dgl/modules.py:61-73, models/dgl/network.py:121-133
 *   the relation-weighted mean of the edge features     : models/dgl/network.py:135-152
 *   the [part i ; part j] pair rows fed to the edge MLP / relation net : models/dgl/network.py:121-125, 135-141
 * narrow_linear_relu: out [R, N] = relu(x [R, K] w[N, K]^T + bias), K <= 16; backward takes the forward's `out` (ReLU
 *   mask) and scratch `ws` (mpa_narrow_linear_relu_workspace floats) and overwrites grad_x [R, K] (if non-NULL),
 *   grad_w [N, K], grad_b [N] (if non-NULL).
 * relation_head: out [R] = sigmoid(h [R, K] . w [K] + bias[0]) * mask [R] (mask NULL = ones), K a multiple of 4; `ws`
 *   (mpa_relation_head_workspace floats) carries the sigmoids to backward, which overwrites grad_h [R, K] (if non-NULL),
 *   grad_w [K], grad_b [1] (if non-NULL).
 * relation_mean: out [G, C] = sum_j edge [G, P, C] rel [G, P] / (sum_j rel [G, P] + 1e-6), P <= 64; backward overwrites
 *   grad_edge [G, P, C] and grad_rel [G, P] (each if non-NULL).
 * pair_rows: out [S, P, P, 2F] = [a [S, P, F] of part i ; b [S, P, F] of part j] for every pair (i, j) (swap != 0: [b of
 *   part j ; a of part i]) — the input rows of the edge MLP and the relation net; F a multiple of 4; backward overwrites
 *   grad_a = sum over j of a's F columns and grad_b = sum over i of b's (each if non-NULL).
 * Fixed-order reductions, no atomics: deterministic.
 * ---------------------------------------------------------------------------------------------- */
int mpa_pair_rows_forward(const float* a, const float* b, int64_t S, int64_t P, int64_t F, int swap, float* out,
                          void* stream);
int mpa_pair_rows_backward(const float* grad_out, int64_t S, int64_t P, int64_t F, int swap, float* grad_a, float* grad_b,
                           void* stream);
int mpa_narrow_linear_relu_forward(const float* x, const float* w, const float* bias, int64_t R, int64_t K, int64_t N,
                                   float* out, void* stream);
int mpa_narrow_linear_relu_workspace(int64_t R, int64_t K, int64_t N, int64_t* float_elems);
int mpa_narrow_linear_relu_backward(const float* grad_out, const float* out, const float* x, const float* w, int64_t R,
                                    int64_t K, int64_t N, float* ws, float* grad_x, float* grad_w, float* grad_b,
                                    void* stream);
int mpa_relation_head_workspace(int64_t R, int64_t K, int64_t* float_elems);
int mpa_relation_head_forward(const float* h, const float* w, const float* bias, const float* mask, int64_t R, int64_t K,
                              float* ws, float* out, void* stream);
int mpa_relation_head_backward(const float* grad_out, const float* h, const float* w, const float* mask, int64_t R,
                               int64_t K, float* ws, float* grad_h, float* grad_w, float* grad_b, void* stream);
int mpa_relation_mean_forward(const float* edge, const float* rel, int64_t G, int64_t P, int64_t C, float* out,
                              void* stream);
int mpa_relation_mean_backward(const float* grad_out, const float* edge, const float* rel, const float* out, int64_t G,
                               int64_t P, int64_t C, float* grad_edge, float* grad_rel, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Recurrent half of a single-layer (bi)directional GRU — replaces the per-step library launches behind
 *   RNNWrapper(nn.GRU(batch_first, bidirectional)) : multi_part_assembly/models/modules/rnn.py:6-46
 *   RGLNet.forward                                 : multi_part_assembly/models/rgl_net/network.py:118-127
 * The input projections gi[d][b][t] = W_ih x + b_ih [D,B,T,3H] (gate order r, z, n as torch.nn.GRU) are one GEMM done by
 * the caller; this runs the T sequential steps of D directions in ONE launch:
 *   r = sigmoid(gi_r + W_hr h + b_hr), z = sigmoid(gi_z + W_hz h + b_hz), n = tanh(gi_n + r (W_hn h + b_hn)),
 *   h' = (1 - z) n + z h,   out [D,B,T,H] = h' of every step.
 * h0 [D,B,H], whh [D,3H,H], bhh [D,3H].  H = 128 or 256, B <= 64, D = 1 or 2.  `ws` (mpa_gru_workspace floats)
 * (8-byte aligned) carries the gates to backward and the tagged words the blocks exchange once per step.  backward: grad_out [D,B,T,H] -> grad_gi [D,B,T,3H], grad_whh, grad_bhh (overwritten);
 * deterministic (partials summed in block order, no float atomics).  Sequences of different lengths: run the padded
 * batch (valid steps first) and mask the outputs — the reverse direction on the per-sample reversed valid prefix.
 * ---------------------------------------------------------------------------------------------- */
int mpa_gru_workspace(int64_t D, int64_t B, int64_t T, int64_t H, int64_t* float_elems);
/* *ok = 1 iff the shapes are instantiated AND the current device can hold the whole grid of both kernels at once (their
 * blocks poll for each other's per-step words; forward / backward return an error instead of stalling when it cannot). */
int mpa_gru_resident(int64_t D, int64_t B, int64_t H, int* ok);
/* `status` (device memory, one int32 the caller keeps at 0; NULL: the kernel traps instead): raised to 1 by a launch whose
 * blocks could not all wait for each other — a block that was never dispatched because another stream's kernels (a
 * collective beside the step, say) held its CU makes the waiting blocks give up after a few seconds (MPA_GRU_POLL_BUDGET
 * polls of ~1 us; default 2^22).  The launch then ends on its own with undefined outputs; the caller reads the word when
 * it next synchronises (gru.py raises a RuntimeError from it) — no trap, no hang, the HIP context stays usable. */
int mpa_gru_forward(const float* gi, const float* h0, const float* whh, const float* bhh, int64_t D, int64_t B, int64_t T,
                    int64_t H, float* ws, float* out, int32_t* status, void* stream);
int mpa_gru_backward(const float* grad_out, const float* h0, const float* whh, const float* out, int64_t D, int64_t B,
                     int64_t T, int64_t H, float* ws, float* grad_gi, float* grad_whh, float* grad_bhh, int32_t* status,
                     void* stream);
/* Test support: `blocks` workgroups that each hold `lds_bytes` of LDS and do nothing for `usec` microseconds — CUs no other
 * stream can use meanwhile (how tests/test_gru_gpu.py provokes the co-residency failure above). */
int mpa_debug_occupy(int64_t blocks, int64_t lds_bytes, int64_t usec, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Part-relation transformer encoder — replaces
 *   TransformerEncoder.forward : multi_part_assembly/models/pn_transformer/transformer.py:63-79
 *   (nn.TransformerEncoder of pre-LN nn.TransformerEncoderLayer, ReLU FFN, batch_first,
 *    src_key_padding_mask = ~valid, final LayerNorm; transformer.py:20-39)
 * tokens [B,P,D] (P <= 64 parts), valid [B*P] (a part is real iff its entry == 1, the reference's
 * `part_valids == 1`; every other part is masked as a KEY; its own row is still computed, as upstream).  D and FF multiples of 64, head dim D/H <= 64, L <= 16 layers.
 * params: HOST array of 12*L + 2 DEVICE pointers, per layer in nn.TransformerEncoderLayer's
 * named_parameters() order — self_attn.in_proj_weight [3D,D], in_proj_bias [3D], out_proj.weight [D,D],
 * out_proj.bias [D], linear1.weight [FF,D], linear1.bias [FF], linear2.weight [D,FF], linear2.bias [D],
 * norm1.weight, norm1.bias, norm2.weight, norm2.bias [D] — then the final norm.weight, norm.bias [D].
 * dropout_p in [0,1) applies to the 4 dropout sites of every layer (attention probabilities, attention
 * output, FFN hidden, FFN output) with a counter-based generator keyed by (seed, site, element): backward
 * must receive the same seed.  seed_dev (nullable): DEVICE address of the seed, read by the kernels instead of
 * `seed` — lets a captured HIP graph draw fresh masks on every replay (the host updates the word between replays).
 * dropout_p = 0 for evaluation.  ws (mpa_transformer_workspace floats) carries the
 * saved activations from forward to backward.  out [B,P,D].
 * backward: grad_out [B,P,D] -> grad_tokens [B,P,D] and grad_params (same layout as params; every buffer
 * is overwritten).  Deterministic: fixed-order reductions, no atomics.
 * ---------------------------------------------------------------------------------------------- */
int mpa_transformer_workspace(int64_t B, int64_t P, int64_t D, int64_t H, int64_t FF, int64_t L,
                              int64_t* float_elems);
int mpa_transformer_forward(const float* tokens, const float* valid, const float* const* params, int64_t B,
                            int64_t P, int64_t D, int64_t H, int64_t FF, int64_t L, float dropout_p,
                            uint64_t seed, const uint64_t* seed_dev, float* ws, float* out, void* stream);
int mpa_transformer_backward(const float* grad_out, const float* valid, const float* const* params, int64_t B,
                             int64_t P, int64_t D, int64_t H, int64_t FF, int64_t L, float dropout_p,
                             uint64_t seed, const uint64_t* seed_dev, float* ws, float* grad_tokens,
                             float* const* grad_params, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pose head — replaces
 *   PoseRegressor.forward : multi_part_assembly/models/modules/regressor.py:50-68
 *   (Linear F-256, LeakyReLU 0.2, Linear 256-128, LeakyReLU 0.2, rot_head 128-4 + F.normalize,
 *    trans_head 128-3; regressor.py:33-48)
 * x [M,F] (1 <= F <= 4096: widths that are not a multiple of 64 — labels / noise appended to the features — are
 * zero-padded inside the workspace; M = B*P tokens).  params: HOST array of 8 DEVICE pointers — fc_layers.0.weight
 * [256,F], .bias, fc_layers.2.weight [128,256], .bias, rot_head.weight [4,128], .bias, trans_head.weight
 * [3,128], .bias.  rot [M,4] unit quaternions (x / max(|x|, 1e-12)), trans [M,3].
 * ws (mpa_pose_head_workspace floats) carries activations to backward, which overwrites grad_x [M,F] and
 * the 8 grad_params buffers (fc_layers.0.weight must be unchanged between the two calls).
 * ---------------------------------------------------------------------------------------------- */
int mpa_pose_head_workspace(int64_t M, int64_t F, int64_t* float_elems);
int mpa_pose_head_forward(const float* x, const float* const* params, int64_t M, int64_t F, float* ws,
                          float* rot, float* trans, void* stream);
int mpa_pose_head_backward(const float* grad_rot, const float* grad_trans, const float* x,
                           const float* const* params, int64_t M, int64_t F, float* ws, float* grad_x,
                           float* const* grad_params, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused optimiser step — replaces torch.optim.Adam / AdamW as configured by
 *   BaseModel.configure_optimizers : multi_part_assembly/models/modules/base_model.py:389-406
 * One streaming pass over flat, 16-byte-aligned fp32 buffers of `numel` elements (parameters,
 * gradients, first and second moments).  `step` is the 1-based step count (bias correction),
 * `grad_scale` multiplies the gradient first (1/world_size of the data-parallel mean),
 * `decoupled_weight_decay` selects AdamW (p *= 1 - lr*wd) over Adam's L2 form (g += wd*p).
 * `decay_mask` (nullable = decay everything): [numel] 1/0 per element; the reference's AdamW parameter groups
 * exempt biases and normalisation weights (multi_part_assembly/utils/utils.py:90-125 filter_wd_parameters).
 * ---------------------------------------------------------------------------------------------- */
int mpa_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                  float lr, float beta1, float beta2, float eps, float weight_decay,
                  int decoupled_weight_decay, int64_t step, float grad_scale, const float* decay_mask,
                  void* stream);

/* Graph-capturable twin: `hyper` is an 8-float DEVICE buffer
 *   [0] lr  [1] 1-beta1^step  [2] sqrt(1-beta2^step)  [3] grad_scale  [4] step count (int32 bits)
 *   [5] clip coefficient (1 = no clipping; written by mpa_grad_clip_coef)  [6] clipped norm (info)  [7] unused.
 * Every call first ADVANCES the step count on the device and recomputes [1], [2] from it, then applies the update:
 * the launch arguments stay constant across replays of a captured step and the host uploads nothing per step (it
 * writes [0] / [3] only when the schedule or the world size changes, [4] when a checkpoint is loaded). */
int mpa_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                      float* hyper, float beta1, float beta2, float eps, float weight_decay,
                      int decoupled_weight_decay, const float* decay_mask, void* stream);

/* Global-norm gradient clipping — replaces Lightning's `gradient_clip_val` (scripts/train.py:90 passes
 * cfg.optimizer.clip_grad; algorithm "norm" = torch.nn.utils.clip_grad_norm_): coef[0] = min(1, max_norm /
 * (|scale * grad|_2 + 1e-6)), coef[1] = that norm; `scale` = *grad_scale_dev if non-NULL else grad_scale (the
 * 1/world of the data-parallel mean that the optimiser applies later).  Point `coef` at hyper + 5 of
 * mpa_adam_step_dev to fold the clipping into the update.  Deterministic two-stage sum in double; `ws` =
 * mpa_grad_clip_workspace bytes. */
int mpa_grad_clip_workspace(int64_t* bytes);
int mpa_grad_clip_coef(const float* grad, int64_t numel, float max_norm, const float* grad_scale_dev,
                       float grad_scale, void* ws, float* coef, void* stream);

/* ---- GT <-> prediction matching of equivalent parts (semantic datasets) --------------------------------------
 * Replaces BaseModel._linear_sum_assignment / _match_parts (multi_part_assembly/models/modules/base_model.py:
 * 150-238: p x p Chamfer cost matrix on n = 100 sub-sampled points, scipy.optimize.linear_sum_assignment on the
 * host, GT poses permuted inside every group) with device code for all groups of a batch, no host round trip.
 *
 * mpa_linear_sum_assignment: `problems` square cost matrices cost [problems, ld, ld] (float32, the top-left
 *   sizes[i] x sizes[i] block of each is the problem, ld <= 64) -> col4row [problems, ld] (column assigned to each
 *   row, -1 past the size).  scipy's shortest-augmenting-path algorithm (rectangular_lsap.cpp, scipy 1.15.3)
 *   step by step in float64: same assignment as scipy, ties included.
 * mpa_match_parts: match_ids [B,P] int32 (0 = unique or padded, g >= 1 = group g; G = number of group slots
 *   considered, groups with id > G are left unmatched), sample_idx [B,G,n] int32 point indices (the reference
 *   draws torch.randperm(N)[:n] per group), poses as [B,P,3] / [B,P,4] (w,x,y,z).  Writes new_trans / new_quat
 *   (the GT poses after rearrangement), perm [B,P] (source slot of every slot), and leaves the cost matrices
 *   [B,G,P,P] and assignments [B,G,P] in the two workspaces. */
int mpa_linear_sum_assignment(const float* cost, const int32_t* sizes, int64_t problems, int64_t ld,
                              int32_t* col4row, void* stream);
int mpa_match_parts(const float* part_pcs, const float* pred_trans, const float* pred_quat, const float* gt_trans,
                    const float* gt_quat, const int32_t* match_ids, const int32_t* sample_idx, int64_t B, int64_t P,
                    int64_t N, int64_t G, int64_t n, float* cost_ws, int32_t* col4row_ws, float* new_trans,
                    float* new_quat, int32_t* perm, void* stream);

/* ---- batch producer (device side) -------------------------------------------------------------------------
 * Replaces the per-part numpy work of GeometryPartDataset.__getitem__ (multi_part_assembly/datasets/
 * geometry_data.py:74-107,133-146) for a whole batch: raw [M,N,3] float64 sampled points (M = B*max_num_part
 * slots), rot [M,9] float64 row-major rotation matrices, perm [M,N] int32 point orders, valids [M] ->
 * part_pcs [M,N,3] float32 = ((rot @ (p - centroid))[perm]) and part_trans [M,3] float32 = centroid; padded
 * slots are zero-filled.  float64 arithmetic like the reference, fixed summation order. */
int mpa_part_batch_transform(const double* raw, const double* rot, const int32_t* perm, const float* valids,
                             int64_t M, int64_t N, float* part_pcs, float* part_trans, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MPA_HIP_H_ */
