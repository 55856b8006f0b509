/*
 * p2pb_hip.h -- C ABI of libp2pb_hip.so, the MI355X (gfx950) implementation of the P2P-Bridge hot path.
 *
 * This is the drop-in boundary: every entry point replaces one launcher of the reference's CUDA
 * extensions (the closest thing the reference has to a C interface are the raw-pointer launcher
 * prototypes in its .cuh files; the pybind layer above them only allocates outputs and checks dtypes).
 * Reference paths are relative to the reference repository root;
 *   PN2 = third_party/openpoints/cpp/pointnet2_batch/src.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM), fp32 / int32, contiguous, layouts exactly as the
 *     reference's tensors: point tensors channel-major [B,C,N], coords [B,3,N], metric clouds [B,N,3];
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream). Every kernel is enqueued
 *     on it; nothing synchronises, allocates or touches the host, so a sequence of calls is
 *     hipGraph-capturable (the reference launches vox/devox/FPS/metrics on the legacy default
 *     stream: PN2/vox_gpu.cu:125, PN2/trilinear_devox_gpu.cu:175, PN2/pvcnn_sampling_gpu.cu:189);
 *   - outputs are fully written by the callee (no pre-zeroing needed) unless stated;
 *   - workspaces are caller-provided (sizes via the *_ws_bytes helpers);
 *   - return value: 0 on success, a hipError_t value on launch failure, negative on invalid
 *     arguments (P2PB_EINVAL). The reference exit(-1)s (PN2/cuda_utils.cuh:30-39) or returns 0/1.
 */
#ifndef P2PB_HIP_H
#define P2PB_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2PB_EINVAL (-22)

/* ABI version of THIS header: bumped whenever an entry point is added, removed or changes meaning (6: round 6 -- since 1:
 * p2pb_debug_gn_finisher removed, flag bit 5 of p2pb_conv3d_k3_forward_sparse, arithmetic code 3 in
 * p2pb_set_split_terms_thread, p2pb_group_sub_stats*, p2pb_se_gate_*, p2pb_conv3d_k3_wgrad_occ*, the *_amax / *_adjoint packs).
 * A binding must compare p2pb_version() with the P2PB_ABI_VERSION it was written against and refuse a mismatch
 * (p2p_bridge_amd/_lib.py does): a stale library behind P2PB_LIB_PATH otherwise fails late, or silently differently. */
#define P2PB_ABI_VERSION 7

/* library / device info --------------------------------------------------------------------- */
int p2pb_version(void);            /* == P2PB_ABI_VERSION of the header the library was built from */
const char *p2pb_target_arch(void); /* "gfx950" */
/* Arithmetic of the split-operand matrix kernels (conv3d_k3 *_forward, pointwise_conv *_forward with >= 128 channels).
 * The reference's layers are cuDNN / cuBLAS convolutions in fp32, which on its Ampere+ targets run as TF32
 * (torch.backends.cudnn.allow_tf32 defaults to True; train.py:221 also sets float32_matmul_precision("high")).
 * gfx950 has no TF32: an fp32 operand is split into 16-bit terms and a product is a sum of EXACT matrix products of the
 * terms, accumulated in fp32 (operands, accumulation and results stay fp32):
 *   16 (default) fp16 pair h0 + h1 of the SCALED operand, h1g0 + h0g1 + h0g0 -- <= 3 * 2^-22 relative per product inside
 *                fp16's range: activations are multiplied by 4 and saturate at |x| = 16376, below |x| = 2^-5 the
 *                representation error is an absolute 2^-27; weights get a per-tensor power-of-two scale at pack time;
 *    6           bf16 terms x0 + x1 + x2: x2y0 + x1y1 + x0y2 + x1y0 + x0y1 + x0y0 -- within a quarter ulp of fp32 at any
 *                magnitude (use it for operands without a usable scale: gradients).
 * Process-wide; takes effect at the next launch (a captured graph keeps what it captured). The *_pack_weights_split
 * functions pack for the arithmetic selected when they are called: re-pack after a switch.
 * -> 0, or P2PB_EINVAL for anything but 6 or 16. */
int p2pb_set_split_terms(int terms);
/* the same for the CALLING THREAD only (0 clears it): a temporary switch, e.g. around one backward pass, that other
 * threads' launches and weight packs do not see; p2pb_get_split_terms returns what a launch from this thread would use */
int p2pb_set_split_terms_thread(int terms);
int p2pb_get_split_terms(void);
/* Deterministic mode (process-wide, default off): the scatter-add backward passes (p2pb_trilinear_devoxelize_backward,
 * p2pb_grouping_backward, p2pb_three_nn_interpolate_backward -- the adjoints of PN2/trilinear_devox_gpu.cu:111,
 * pvcnn_grouping_gpu.cu:51, pvcnn_neighbor_interpolate_gpu.cu:101, whose float atomicAdd order is arbitrary in the reference
 * too) accumulate in a fixed order -- one wave per workgroup over rows held in LDS -- so that a training run is bit-reproducible;
 * slower. A row that does not fit 128 KB of LDS is refused (P2PB_EINVAL) instead of falling back to global atomics.
 * HARDWARE ASSUMPTION (gfx950, the only architecture this library is built for -- p2pb_target_arch): the 64 lanes of ONE
 * ds_add_f32 instruction that hit the same LDS address are served in ascending lane order. The ISA manual does not promise
 * it; tests/test_train_gpu.py::test_deterministic_mode_makes_training_bit_reproducible (two whole training runs, bit for bit) is what holds it on this part. */
int p2pb_set_deterministic(int on);
int p2pb_get_deterministic(void);
/* Test / debug hook: the kernel form the most recent p2pb_pointwise_conv* launch with these (cin, cout, npos) took
 * (0 exact-fp32 unaligned, 1 wide exact-fp32, 2 wide f16x3, 3 split 128-channel, 4 split 256-channel, 5 ping-pong,
 * 6 gathered wide f16x3), or -1 if none; *launches (may be NULL) = how many such launches since the last reset.
 * cin < 0 resets the table. Host-side bookkeeping only (under hipGraph capture: recorded at capture time). */
int p2pb_debug_pointwise_form(int cin, int cout, int npos, unsigned long long *launches);

/* The GroupNorm(+AdaGN) that FOLLOWS a layer, handed to the layer's own launch: arm, then call a statistics-producing pointwise
 * entry point (p2pb_pointwise_conv_forward / _conv_pool_forward / _conv_pool_gather with stats_part != NULL) from the same
 * thread. scale / shift / chmean (f32[b, cout]; chmean may be NULL) are then filled in stream order exactly as
 * p2pb_gn_affine_params(b, cout, groups, nslots, count_per_channel, stats_part, gamma, beta, style, style_stride, eps, ...)
 * called right after that launch would fill them (same bits): the entry point puts the gn_affine launch behind the producer
 * (the forms that ran it in the producing kernel's last workgroup measured slower and left in round 5). Replaces the reference's separate
 * nn.GroupNorm / AdaGN pass after each 1x1 convolution (models/pvcnn.py:162-205, models/modules.py:341-358).
 * Arming twice without a consuming launch in between is an error (P2PB_EINVAL); p2pb_gn_finisher_armed(): 1 while one waits. */
int p2pb_gn_finisher_arm(int groups, double count_per_channel, const float *gamma, const float *beta, const float *style,
                         int style_stride, float eps, float *scale, float *shift, float *chmean);
int p2pb_gn_finisher_armed(void);
void p2pb_gn_finisher_disarm(void); /* drop an armed finisher whose producing launch did not happen (error paths) */

/* Voxelization.forward normalisation (models/pvcnn.py:215-228): centre on the mean, divide by
 * 2*max-norm (+eps), +0.5, *r, clamp [0,r-1]; also the half-to-even rounded int voxel coords.
 * The reference does this with torch reductions; the build fixes the summation order
 * (DESIGN.md "voxel_coords") so CPU oracle and HIP agree bit-for-bit.
 *   coords f32[b,3,n] -> norm f32[b,3,n], vox i32[b,3,n] */
int p2pb_voxel_coords(int b, int n, int r, int normalize, float eps, const float *coords, float *norm,
                      int *vox, void *stream);

/* avg_voxelize forward: replaces avg_voxelize() PN2/vox.cuh:5-6 (kernels PN2/vox_gpu.cu:18,50),
 * called from avg_voxelize_forward PN2/vox.cpp:17.
 *   coords i32[b,3,n], feat f32[b,c,n] -> ind i32[b,n], cnt i32[b,r^3], out f32[b,c,r^3]
 * Deterministic: each voxel sums its points in ascending point index (the reference's float
 * atomicAdd order is arbitrary). ws: p2pb_avg_voxelize_ws_bytes(b,n,r) bytes. */
size_t p2pb_avg_voxelize_ws_bytes(int b, int n, int r);
int p2pb_avg_voxelize_forward(int b, int c, int n, int r, const int *coords, const float *feat, int *ind,
                              int *cnt, float *out, void *ws, void *stream);
/* replaces avg_voxelize_grad() PN2/vox.cuh:7-8 (kernel PN2/vox_gpu.cu:92) */
int p2pb_avg_voxelize_backward(int b, int c, int n, int r3, const int *ind, const int *cnt,
                               const float *grad_y, float *grad_x, void *stream);

/* trilinear devoxelize: replaces trilinear_devoxelize() / _grad() PN2/trilinear_devox.cuh:5-11
 * (kernels PN2/trilinear_devox_gpu.cu:21,123). inds i32[b,8,n] / wgts f32[b,8,n] are written only
 * when is_training != 0 (may be NULL otherwise).
 *   coords f32[b,3,n] (voxel units), feat f32[b,c,r^3] -> outs f32[b,c,n] */
int p2pb_trilinear_devoxelize_forward(int b, int c, int n, int r, int is_training, const float *coords,
                                      const float *feat, int *inds, float *wgts, float *outs, void *stream);
/* grad_x f32[b,c,r^3] is zero-filled by the callee, then scatter-added (fp32 atomics) */
int p2pb_trilinear_devoxelize_backward(int b, int c, int n, int r3, const int *inds, const float *wgts,
                                       const float *grad_y, float *grad_x, void *stream);

/* ball query: replaces ball_query() PN2/pvcnn_ball_query.cuh:4-6 (kernel PN2/pvcnn_ball_query_gpu.cu:19).
 * r2 = radius*radius (float product, PN2/pvcnn_ball_query.cpp:25).
 *   centers f32[b,3,m], points f32[b,3,n] -> idx i32[b,m,u]: first u points (ascending index) with
 *   d2 < r2, remaining slots padded with the first hit; all zero when nothing is in range. */
int p2pb_ball_query(int b, int n, int m, float r2, int u, const float *centers, const float *points, int *idx,
                    void *stream);

/* grouping: replaces grouping() / grouping_grad() PN2/pvcnn_grouping.cuh:4-7
 * (kernels PN2/pvcnn_grouping_gpu.cu:18,62).  feat f32[b,c,n], idx i32[b,m,u] -> out f32[b,c,m,u] */
int p2pb_grouping_forward(int b, int c, int n, int m, int u, const float *feat, const int *idx, float *out,
                          void *stream);
int p2pb_grouping_backward(int b, int c, int n, int m, int u, const float *grad_y, const int *idx,
                           float *grad_x, void *stream);
/* the same with sample b of grad_y at grad_y + b * gy_pitch floats (gy_pitch >= c*m*u): a channel slice of a wider tensor -- one
 * part of the gradient of a concatenation (models/pvcnn.py:126) -- is read in place (build addition; training) */
int p2pb_grouping_backward_pitched(int b, int c, int n, int m, int u, const float *grad_y, long gy_pitch, const int *idx,
                                   float *grad_x, void *stream);

/* fused set-abstraction operand (BallQuery.forward, models/pvcnn.py:116-126, inference): out f32[b,3+c,m,u] =
 * [coords[:, idx] - centers (3 ch) | feat[:, idx] (c ch)] in one pass */
int p2pb_group_concat(int b, int c, int n, int m, int u, const float *coords, const float *centers,
                      const float *feat, const int *idx, float *out, void *stream);

/* gather: replaces gather_features() / _grad() PN2/pvcnn_sampling.cuh:4-7
 * (kernels PN2/pvcnn_sampling_gpu.cu:17,55).  feat f32[b,c,n], idx i32[b,m] -> out f32[b,c,m] */
int p2pb_gather_features_forward(int b, int c, int n, int m, const float *feat, const int *idx, float *out,
                                 void *stream);
int p2pb_gather_features_backward(int b, int c, int n, int m, const float *grad_y, const int *idx,
                                  float *grad_x, void *stream);

/* furthest point sampling: replaces furthest_point_sampling() PN2/pvcnn_sampling.cuh:8-9
 * (kernel PN2/pvcnn_sampling_gpu.cu:92). Starts at index 0; ties resolve exactly like the
 * reference's 512-thread block: (d desc, k mod 512 asc, k asc).
 *   coords f32[b,3,n] -> idx i32[b,m].  dist_ws f32[b,n] is scratch (running min distances),
 *   required (non-NULL) only when n > 16384; initialised by the callee. */
int p2pb_furthest_point_sampling(int b, int n, int m, const float *coords, float *dist_ws, int *idx,
                                 void *stream);

/* the same sampling for LARGE clouds (object merge, denoise_object.py:112: 3N patch points -> N; the 50000-point
 * room patches of BASELINE configs 4-5): 64 workgroups share a cloud, points and running distances stay in registers,
 * one global all-gather per round. Same indices as p2pb_furthest_point_sampling. 16384 < n <= 524288, any b (launched
 * four clouds at a time). ws: p2pb_fps_coop_ws_bytes(b, n) bytes; after the call the b ints at ws + b*1024 are
 * per-cloud flags (1 = the cooperative kernel lost a peer; the single-workgroup kernel then recomputed that cloud on
 * the device, so idx is valid either way). P2PB_EINVAL when the device cannot hold 64 workgroups at once. */
size_t p2pb_fps_coop_ws_bytes(int b, int n);
int p2pb_furthest_point_sampling_coop(int b, int n, int m, const float *coords, void *ws, int *idx, void *stream);
/* The same indices for large clouds from ONE workgroup per cloud with exact pruning (sampling.hip: a 16^3 grid over the
 * cloud; a round only revisits the cells whose bounding box is closer to the new sample than their current maximum --
 * ~n/j points in round j instead of n). Any n >= 1; ws: p2pb_fps_grid_ws_bytes(b, n) bytes, 16-byte aligned. */
size_t p2pb_fps_grid_ws_bytes(int b, int n);
int p2pb_furthest_point_sampling_grid(int b, int n, int m, const float *coords, void *ws, int *idx, void *stream);

/* point <-> triangle-mesh squared distances (P2M metric): replace pytorch3d._C.point_face_dist_forward /
 * face_point_dist_forward as called by metrics/p2m.py:66,131 for one (mesh, cloud) pair. points f32[np,3],
 * tris f32[nt,3,3] (faces as vertex triples); triangles of area < min_triangle_area count as their edges.
 *   p2pb_point_face_dist -> dist f32[np], idx i32[np] (closest triangle of every point, first minimum)
 *   p2pb_face_point_dist -> dist f32[nt], idx i32[nt] (closest point of every triangle) */
int p2pb_point_face_dist(int np, int nt, const float *points, const float *tris, float min_triangle_area, float *dist,
                         int *idx, void *stream);
int p2pb_face_point_dist(int np, int nt, const float *points, const float *tris, float min_triangle_area, float *dist,
                         int *idx, void *stream);

/* exact K nearest neighbours with patch-sized K: replaces pytorch3d.ops.knn_points(seeds, cloud, K, return_nn=True)
 * as called by denoise_object.py:91 (pytorch3d: pip dependency, not vendored; contract = the K smallest squared
 * distances per query, ascending, with indices). Ties: ascending point index.
 *   query f32[b,s,3], points f32[b,n,3] (point-major) -> dist2 f32[b,s,k], idx i32[b,s,k], nn f32[b,s,k,3]
 *   (any output may be NULL); 1 <= k <= min(n, 4096); ws: p2pb_knn_points_ws_bytes(b,s,n) bytes of scratch. */
size_t p2pb_knn_points_ws_bytes(int b, int s, int n);
int p2pb_knn_points(int b, int s, int n, int k, const float *query, const float *points, float *dist2, int *idx,
                    float *nn, void *ws, void *stream);

/* 3-NN inverse-squared-distance interpolation: replaces three_nearest_neighbors_interpolate() /
 * _grad() PN2/pvcnn_neighbor_interpolate.cuh:4-14 (kernels PN2/pvcnn_neighbor_interpolate_gpu.cu:20,96,154).
 *   points f32[b,3,n], centers f32[b,3,m], cfeat f32[b,c,m] -> idx i32[b,3,n], w f32[b,3,n], out f32[b,c,n] */
int p2pb_three_nn_interpolate_forward(int b, int c, int m, int n, const float *points, const float *centers,
                                      const float *cfeat, int *idx, float *w, float *out, void *stream);
/* the same op in two launches (search: coordinates only; interpolation: features), for callers that
 * overlap the geometry pipeline with the feature path */
int p2pb_three_nn(int b, int m, int n, const float *points, const float *centers, int *idx, float *w, void *stream);
int p2pb_three_interpolate(int b, int c, int m, int n, const float *cfeat, const int *idx, const float *w,
                           float *out, void *stream);
/* p2pb_three_nn through a uniform 16^3 grid over the centres (exact: same idx / w, ties by ascending index);
 * m >= 3 (the centre records sit in LDS up to 8192 centres and stay in global memory / L2 above), ws:
 * p2pb_three_nn_cells_ws_bytes(b, m) bytes of scratch (16-byte aligned) */
size_t p2pb_three_nn_cells_ws_bytes(int b, int m);
int p2pb_three_nn_cells(int b, int m, int n, const float *points, const float *centers, int *idx, float *w, void *ws,
                        void *stream);
int p2pb_three_nn_interpolate_backward(int b, int c, int n, int m, const float *grad_y, const int *idx,
                                       const float *w, float *grad_x, void *stream);
/* (gy_pitch >= c*n: as p2pb_grouping_backward_pitched; the slice of models/pvcnn.py:219's concatenation) */
int p2pb_three_nn_interpolate_backward_pitched(int b, int c, int n, int m, const float *grad_y, long gy_pitch, const int *idx,
                                               const float *w, float *grad_x, void *stream);

/* chamfer_3D: replaces chamfer_cuda_forward/backward (metrics/chamfer3D/chamfer3D.cu:135,176;
 * kernels :12,:155). xyz are POINT-major f32[b,n,3] / f32[b,m,3]. Lowest index wins distance ties.
 * backward ACCUMULATES into gradxyz1/2 (caller zero-fills, metrics/chamfer3D/dist_chamfer_3D.py:77-83). */
int p2pb_chamfer_forward(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1, float *dist2,
                         int *idx1, int *idx2, void *stream);
/* the same with scratch (p2pb_chamfer_ws_bytes(b, n, m) bytes): small batches split the target cloud over more
 * workgroups and combine through 64-bit atomicMin keys (distance bits, index) -- identical results */
size_t p2pb_chamfer_ws_bytes(int b, int n, int m);
int p2pb_chamfer_forward_ws(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1, float *dist2,
                            int *idx1, int *idx2, void *ws, void *stream);
int p2pb_chamfer_backward(int b, int n, int m, const float *xyz1, const float *xyz2, float *gradxyz1,
                          float *gradxyz2, const float *graddist1, const float *graddist2, const int *idx1,
                          const int *idx2, void *stream);

/* PyTorchEMD: replaces ApproxMatchForward / MatchCostForward / MatchCostBackward
 * (metrics/PyTorchEMD/cuda/emd_kernel.cu:177,264,377; kernels :33,:211,:300,:347).
 *   xyz1 f32[b,n,3], xyz2 f32[b,m,3] -> match f32[b,m,n]; temp = the reference's scratch, 2(n+m) floats per cloud
 *   (emd_kernel.cu:34). p2pb_approxmatch_temp_floats(b,n,m) = that plus the partial sums of the chunked launches the _ws
 *   entry point uses when the batch alone cannot fill the chip */
size_t p2pb_approxmatch_temp_floats(int b, int n, int m);
int p2pb_approxmatch_forward(int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                             float *temp, void *stream);
/* explicit scratch size: temp_floats >= p2pb_approxmatch_temp_floats() enables the chunked kernels (small batches of large
 * clouds); the plain entry point above keeps the reference's contract (temp = 2 (n + m) b floats, emd_kernel.cu:34) */
int p2pb_approxmatch_forward_ws(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                                size_t temp_floats, void *stream);
int p2pb_matchcost_forward(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                           float *cost, void *stream);
int p2pb_matchcost_backward(int b, int n, int m, const float *grad_cost, const float *xyz1, const float *xyz2,
                            const float *match, float *grad1, float *grad2, void *stream);

/* emd_assignment (auction): replaces emd_cuda_forward / emd_cuda_backward
 * (metrics/emd_assignment/emd_assignment/emd_cuda.cu:228,305; kernels :23-226,:284). Buffers as
 * allocated by emd_module.py:43-54; assignment/assignment_inv must be -1-filled, price and
 * max_increments zero-filled by the caller, as the reference's Python does.
 * Returns 1 on success (the reference's convention), -1 on n != m / n % 128 / b > 512 (:236-249).
 * A round is Bid (one wave per unassigned bidder) / GetMax / Assign; the first ten rounds are three launches each, the
 * latency-bound rest is ONE launch (one workgroup per cloud, n <= 8192; csrc/emd.hip auction_persist_kernel). The assignment is
 * schedule dependent in the reference as well (racing atomicMax / Assign): the contract is the invariants + transport cost. */
int p2pb_auction_forward(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *assignment,
                         float *price, int *assignment_inv, int *bid, float *bid_increments,
                         float *max_increments, int *unass_idx, int *unass_cnt, int *unass_cnt_sum, int *cnt_tmp,
                         int *max_idx, float eps, int iters, void *stream);
int p2pb_auction_backward(int b, int n, const float *xyz1, const float *xyz2, float *gradxyz,
                          const float *graddist, const int *idx, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused voxel-branch kernels. These have no counterpart in the reference's extension modules: they
 * replace the torch.nn / cuDNN calls between avg_voxelize and trilinear_devoxelize inside PVConv
 * (models/pvcnn.py:265-286,318-324: Conv3d, AdaGN, Swish, Conv3d, AdaGN, SE3d) for inference.
 * ------------------------------------------------------------------------------------------- */

/* 3x3x3, stride 1, pad 1 convolution (nn.Conv3d of models/pvcnn.py:266-282) on the fp32 matrix cores.
 * Weights are pre-packed once: w f32[cout,cin,3,3,3] -> wt_packed f32[p2pb_conv3d_k3_packed_floats()]. */
size_t p2pb_conv3d_k3_packed_floats(int cout, int cin);
int p2pb_conv3d_k3_pack_weights(int cout, int cin, const float *w, float *wt_packed, void *stream);
/* The split pack: every weight as 16-bit terms for the split-operand form of the same convolution, in the arithmetic
 * selected by p2pb_set_split_terms WHEN THE PACK IS MADE -- 16: an fp16 pair of w * S_w (S_w = the power of two that
 * brings max |w| into [2^13, 2^14), found on the device; 1 / (4 S_w) is stored behind the pack), three MFMA products per
 * fp32 product; 6: three bf16 terms, six products, dropped terms < 2^-26 |x*w|. fp32 accumulate either way
 * (conv3d.hip). Selected by flags bit 2 below. */
size_t p2pb_conv3d_k3_split_packed_bytes(int cout, int cin);
int p2pb_conv3d_k3_pack_weights_split(int cout, int cin, const float *w, void *wt_split, void *stream);
/* out[b,cout,r,r,r] = conv(xf(in[b,cin,r,r,r])) + bias;  xf(x) = x when in_scale == NULL, else
 * x*in_scale[b,ci] + in_shift[b,ci] followed by Swish when in_swish != 0 (the preceding AdaGN+Swish,
 * folded). stats_part (optional, f32[p2pb_conv3d_k3_stats_floats()]) receives per-(b, slot, cout)
 * {sum, sum of squares} of the output for the GroupNorm that follows. r in {4,8,16,32}. */
size_t p2pb_conv3d_k3_stats_floats(int b, int cout, int r);
int p2pb_conv3d_k3_forward(int b, int cin, int cout, int r, const float *in, const float *wt_packed,
                           const float *bias, const float *in_scale, const float *in_shift, int in_swish,
                           float *out, float *stats_part, void *stream);

/* Sparse-aware form (exact): in_sub f32[b,cin] is subtracted from the transformed operand, out_class
 * f32[b,27,cout] replaces bias per boundary class (low / interior / high along d,h,w) -- see conv3d.hip:
 * conv(x) = conv(x - a) + conv(a). flags bit 0: skip all-zero operand tiles; bit 1: compact 4x8x8 bricks;
 * bit 2: wt_packed is the split pack (bf16x6 arithmetic), else the fp32 pack (exact-fp32 MFMA);
 * bit 3: voxel-major grids, in f32[b,r,r,r,cin] / out f32[b,r,r,r,cout] (the fused voxel branch's layout: a
 * voxel's channels are contiguous for the staging loads, the stores, voxelize and devoxelize). */
int p2pb_conv3d_k3_forward_ex(int b, int cin, int cout, int r, const float *in, const void *wt_packed,
                              const float *bias, const float *out_class, const float *in_scale,
                              const float *in_shift, int in_swish, const float *in_sub, int flags, float *out,
                              float *stats_part, void *stream);
/* List-driven sparse form. p2pb_conv3d_brick_lists derives, from the voxel occupancy cnt i32[b,r^3] of
 * avg_voxelize, the compacted lists of (sample*NBRICK + brick) with / without MFMA work for the first
 * (halo 1) and second (halo 2) convolution of a PVConv: lists i32[4][b*NBRICK] = {active0, inactive0,
 * active1, inactive1}, counts i32[4]; flags_ws = b*NBRICK*2 bytes scratch; NBRICK = 128 (r=32) / 16 (r=16).
 * p2pb_conv3d_k3_forward_sparse runs the MFMA kernel on the active pairs only and writes the (exactly
 * known) constants + statistics of the inactive bricks. flags bit 5 (32): the inactive bricks' STATISTICS only -- their
 * outputs are left unwritten, for a caller that reads `out` inside the active bricks alone (a PVConv's second
 * convolution: the devoxelisation's corners lie within one voxel of an occupied voxel). */
int p2pb_conv3d_brick_lists(int b, int r, const int *cnt, unsigned char *flags_ws, int *lists, int *counts,
                            void *stream);
int p2pb_conv3d_k3_forward_sparse(int b, int cin, int cout, int r, const float *in, const void *wt_packed,
                                  const float *bias, const float *out_class, const float *in_scale,
                                  const float *in_shift, int in_swish, const float *in_sub, int flags /* bits 2, 3, 4, 5 */,
                                  const int *active_list, const int *active_count, const int *inactive_list,
                                  const int *inactive_count, float *out, float *stats_part, void *stream);
/* Compact form (voxel-level sparsity inside the bricks; conv3d.hip). p2pb_conv3d_active_lists derives, from the voxel
 * occupancy cnt i32[b,r^3], per (sample, 4x8x8 brick) the sorted local ids of the voxels in D1 = dilate(occupied,1)
 * (set 0) and D2 = dilate(D1,1) (set 1), followed by the ids outside the set: lists u8[2][b][NBRICK][256], counts
 * i32[2][b][NBRICK]; r in {8,16,32}. p2pb_conv3d_k3_forward_compact computes only the listed outputs of every brick
 * (a first convolution with set 0; a second one in far-field form -- in_sub / out_class -- with set 1) and writes the
 * known constants (+ their exact statistics) elsewhere. Voxel-major tensors, split weight pack. flags bit 5 (32, as in
 * p2pb_conv3d_k3_forward_sparse; ABI 6): the constants of the UNLISTED voxels are left unwritten (statistics still exact) --
 * for a caller that reads `out` at listed voxels alone (a PVConv's second convolution: read by the devoxelisation only). */
int p2pb_conv3d_active_lists(int b, int r, const int *cnt, unsigned char *lists, int *counts, void *stream);
int p2pb_conv3d_k3_forward_compact(int b, int cin, int cout, int r, const float *in, const void *wt_split,
                                   const float *bias, const float *out_class, const float *in_scale,
                                   const float *in_shift, int in_swish, const float *in_sub,
                                   const unsigned char *alist, const int *acount, float *out, float *stats_part,
                                   int flags /* bit 5 */, void *stream);
/* Pre-split operand grids ("S format", round 3; conv3d.hip PreStage). The split kernels' staging phase -- load, folded
 * norm + Swish, fp16-pair split, LDS write, repeated for every brick halo and channel block that touches a voxel -- can be
 * done ONCE per element by the operand's producer: S = u32x4[b][r^3][ceil(cin/16)][2 planes][2 khalf], 8 fp16 per entry
 * (h0 | h1 of 4 x value), 4 bytes per (voxel, channel) with the channels zero-padded to a multiple of 16. The f16x3
 * kernels then stage with LDS-DMA alone. Bit-identical outputs. Producers: p2pb_avg_voxelize_cl_gather_split (a first
 * convolution's operand straight from the voxeliser) and p2pb_conv3d_presplit (y f32[b,nvox,c] voxel-major -> S, applying
 * swish?(y*in_scale + in_shift) - in_sub; in_scale == NULL: the plain split). Consumers: p2pb_conv3d_k3_forward_ex /
 * _sparse with flags bit 4 (16; needs bits 2 and 3, no in_scale / in_sub) and
 * p2pb_conv3d_k3_forward_compact_pre. f16x3 arithmetic only (P2PB_EINVAL under bf16x6). */
int p2pb_conv3d_presplit(int b, int c, long nvox, const float *y, const float *in_scale, const float *in_shift,
                         int in_swish, const float *in_sub, void *out_split, void *stream);
int p2pb_conv3d_k3_forward_compact_pre(int b, int cin, int cout, int r, const void *in_split, const void *wt_split,
                                       const float *bias, const float *out_class, const unsigned char *alist,
                                       const int *acount, float *out, float *stats_part, int flags /* bit 5 */,
                                       void *stream);
/* a[b,cin] = xf(prev_bias[cin]) (the operand's far-field constant) and k_out[b,27,cout] = conv(a) + bias per
 * boundary class, for p2pb_conv3d_k3_forward_ex */
int p2pb_conv3d_k3_far_field(int b, int cin, int cout, const float *prev_bias, const float *in_scale,
                             const float *in_shift, int in_swish, const float *wt_packed, const float *bias,
                             float *a, float *k_out, float *tap_ws /* f32[b,27,cout] scratch */, void *stream);
/* The same with the GroupNorm(+AdaGN) BETWEEN the two convolutions of a PVConv (models/pvcnn.py:267-268, modules.py:341-358)
 * folded into the launch: part f32[b,nslots,cin,2] = the first convolution's statistics partials and the norm's parameters as
 * in p2pb_gn_affine_params; scale / shift f32[b,cin] are OUTPUTS (the values and bits p2pb_gn_affine_params writes) for the
 * kernels that stage the second convolution's operand. One launch instead of two on the sampler's dependent chain. */
int p2pb_conv3d_k3_far_field_gn(int b, int cin, int cout, const float *prev_bias, const float *part, int nslots,
                                double count_per_channel, int groups, const float *gamma, const float *beta,
                                const float *style, int style_stride, float eps, int in_swish, const float *wt_packed,
                                const float *bias, float *scale, float *shift, float *a, float *k_out, float *tap_ws,
                                void *stream);

/* GroupNorm (+AdaGN style, models/modules.py:341-358) folded into a per-(sample, channel) affine:
 * AdaGN(GN(x)) == x*scale + shift. part f32[b,nslots,c,2] partial {sum,sumsq}; gamma/beta f32[c] or
 * NULL; style = b rows of (factor[c] | bias[c]) with a row pitch of style_stride floats, or NULL; chmean
 * (optional) = per-channel mean of the transformed output (SE3d squeeze, models/modules.py:377-378). */
int p2pb_gn_affine_params(int b, int c, int groups, int nslots, double count_per_channel, const float *part,
                          const float *gamma, const float *beta, const float *style, int style_stride, float eps,
                          float *scale, float *shift, float *chmean, void *stream);
/* the same, additionally returning the group moments mean_rstd f32[b,groups,2] = {mean, 1/sqrt(var+eps)} (may be NULL)
 * for the training backward pass (p2pb_norm_act_backward) */
int p2pb_gn_affine_params_ex(int b, int c, int groups, int nslots, double count_per_channel, const float *part,
                             const float *gamma, const float *beta, const float *style, int style_stride, float eps,
                             float *scale, float *shift, float *chmean, float *mean_rstd, void *stream);
/* Training backward of y = act(GroupNorm(x) * gamma + beta [* factor + bias]) (torch.nn.GroupNorm + AdaGN,
 * models/modules.py:341-358, + Swish :14-19) given the folded affine (scale, shift) and mean_rstd of the forward pass:
 * x, gy f32[b,c,npos] -> dx f32[b,c,npos], dgamma / dbeta f32[c] (NULL to skip), dstyle f32[b,2c] = (d factor | d bias)
 * (required iff style != NULL). ws: 2*b*c + 2*b*groups floats. Deterministic. */
int p2pb_norm_act_backward(int b, int c, int groups, int npos, const float *x, const float *gy, const float *scale,
                           const float *shift, const float *mean_rstd, const float *gamma, const float *beta,
                           const float *style, int style_stride, int swish, float *dx, float *dgamma, float *dbeta,
                           float *dstyle, float *ws, void *stream);
/* The same with the layer's neighbours in a training step folded in (each used to be 1-3 elementwise launches per layer):
 *   gmean f32[b,c] | NULL    gradient of the per-channel MEAN of the activation-free output (SE3d's squeeze input,
 *                            models/modules.py:377-378, which the forward pass takes from p2pb_gn_affine_params_ex's chmean):
 *                            gy_eff = gy + gmean / npos. Requires swish == 0 and drop_p == 0.
 *   drop_p, seed, salt       nn.Dropout(drop_p) behind the Swish (models/pvcnn.py:268-272): the keep mask is a counter-based hash
 *                            of (element index, seed[0], seed[1] read from DEVICE memory, salt); pass the forward pass's values.
 *   residual, rgate          the forward pass added residual f32[b,c,npos] * rgate f32[b,c] (PVConv: devoxelised grid * SE gate,
 *                            models/pvcnn.py:322-326): dres f32[b,c,npos] = gy * rgate, drgate f32[b,c] = sum_p gy * residual.
 *                            All four or none (a residual without a gate needs no kernel: its gradient is gy).
 *   gy_pitch                 floats between two samples of gy (0 = c * npos): a channel slice of a wider tensor -- one part of a
 *                            concatenation's gradient -- is read in place instead of through a contiguous copy.
 * Everything NULL / 0 = p2pb_norm_act_backward (same bits). */
int p2pb_norm_act_backward_ex(int b, int c, int groups, int npos, const float *x, const float *gy, const float *scale,
                              const float *shift, const float *mean_rstd, const float *gamma, const float *beta,
                              const float *style, int style_stride, int swish, long gy_pitch, const float *gmean,
                              const float *residual, const float *rgate, float drop_p, const unsigned *seed, unsigned salt, float *dx, float *dgamma,
                              float *dbeta, float *dstyle, float *dres, float *drgate, float *ws, void *stream);
/* Forward of the folded norm in train(): y = drop(act(x * scale[b,c] + shift[b,c])) + residual * rgate[b,c]; residual, rgate
 * NULL = none (rgate alone is ignored), drop_p == 0 = no dropout (then p2pb_affine_act's bits). seed: two 32-bit words in
 * DEVICE memory (drawn once per forward pass by the host framework's generator, so that a captured step replays with fresh
 * masks), salt: a per-layer constant. */
int p2pb_affine_act_train(int b, int c, int npos, const float *x, const float *scale, const float *shift, int swish,
                          const float *residual, const float *rgate, float drop_p, const unsigned *seed, unsigned salt,
                          float *y, void *stream);
/* SE3d gate (models/modules.py:362-378) folded into the devoxelisation affine: gate = sigmoid(w2 relu(w1 chmean)),
 * aff_a = scale*gate, aff_b = shift*gate. w1 f32[hidden,c], w2 f32[c,hidden] (nn.Linear layouts, no bias). */
int p2pb_se_gate_affine(int b, int c, int hidden, const float *chmean, const float *w1, const float *w2,
                        const float *scale, const float *shift, float *aff_a, float *aff_b, void *stream);
/* The tail of a PVConv's voxel branch in one launch (models/pvcnn.py:283-286 AdaGN + SE3d behind the second Conv3d, and the
 * norm of the point branch's SharedMLP, :162-205): part2 f32[b,nslots2,c,2] + its norm (as p2pb_gn_affine_params) ->
 * aff_a, aff_b f32[b,c] = (scale, shift) x the SE3d gate (hidden == 0: no gate) -- what p2pb_gn_affine_params(..., chmean)
 * followed by p2pb_se_gate_affine returns, same bits; partp f32[b,nslotsp,cp,2] (NULL: none) + its norm -> scale_p, shift_p
 * f32[b,cp]. Three launches of the dependent chain become one. */
int p2pb_pvconv_tail(int b, int c, int hidden, const float *part2, int nslots2, double count2, int groups2,
                     const float *gamma2, const float *beta2, const float *style2, int style_stride2, float eps2,
                     const float *w1, const float *w2, float *aff_a, float *aff_b, int cp, const float *partp, int nslotsp,
                     double countp, int groupsp, const float *gammap, const float *betap, const float *stylep,
                     int style_stridep, float epsp, float *scale_p, float *shift_p, void *stream);

/* Voxel-major forms for the fused branch (grid f32[b,r,r,r,c]; conv flags bit 3): same values as
 * p2pb_avg_voxelize_forward / p2pb_trilinear_devoxelize_affine, coalesced on both sides (voxelize.hip).
 * feat_t: f32[b,n,c] scratch (the point-major copy of feat); aff_a/aff_b may both be NULL. With add:
 * outs += swish(add*add_scale[b,c] + add_shift[b,c]), PVConv's point branch (models/pvcnn.py:286,325). */
int p2pb_avg_voxelize_cl_forward(int b, int c, int n, int r, const int *coords, const float *feat, int *ind, int *cnt,
                                 float *out, float *feat_t, void *ws, void *stream);
/* the same in two halves: the coordinate-only sort (run once per (level, resolution), on the geometry stream) and
 * the feature gather that consumes its cnt / ws */
int p2pb_voxel_sort(int b, int n, int r, const int *coords, int *ind, int *cnt, void *ws, void *stream);
int p2pb_avg_voxelize_cl_gather(int b, int c, int n, int r, const float *feat, const int *cnt, const void *ws,
                                float *out, float *feat_t, void *stream);
/* the same straight into the pre-split operand format of the voxel convolutions (S format above):
 * out_split = b * r^3 * ceil(c/16) * 64 bytes */
int p2pb_avg_voxelize_cl_gather_split(int b, int c, int n, int r, const float *feat, const int *cnt, const void *ws,
                                      void *out_split, float *feat_t, void *stream);
int p2pb_trilinear_devoxelize_cl_affine(int b, int c, int n, int r, const float *coords, const float *grid,
                                        const float *aff_a, const float *aff_b, const float *add /* f32[b,c,n] or NULL */,
                                        const float *add_scale, const float *add_shift, float *outs, void *stream);
/* trilinear devoxelize of feat*aff_a[b,c] + aff_b[b,c] (AdaGN + SE gate folded), inference only */
int p2pb_trilinear_devoxelize_affine(int b, int c, int n, int r, const float *coords, const float *feat,
                                     const float *aff_a, const float *aff_b, float *outs, void *stream);

/* Set abstraction with the first 1x1 convolution applied before the grouping (linear: W[xyz[idx]-centre; f[idx]]
 * = Z[:,idx] - Cx[:,centre], Z = W[xyz;f]+bias on the n points, Cx = W_xyz centre): out[b,c,j,k] = z[b,c,idx[b,j,k]]
 * - cx[b,c,j] (cx may be NULL), plus the {sum, sumsq} partials f32[p2pb_group_sub_stats_floats()] = [b,nslots,c,2] of
 * the GroupNorm that follows. Replaces grouping + concat + the (3+C)-channel GEMM of models/pvcnn.py:117-126,408. */
/* Feature propagation likewise (interpolation is linear too): out[b,c,j] = sum_k w_k cz[b,c,idx[b,k,j]] + add[b,c,j]
 * (+ bias[c]), cz = W_g g on the m coarse points, add = W_s skip + bias on the n fine ones (either may be NULL);
 * stats_part f32[p2pb_group_sub_stats_floats(b,c,n,1)]. Replaces interpolate + concat + the first GEMM of
 * models/pvcnn.py:457-461. */
int p2pb_three_interpolate_add(int b, int c, int m, int n, const float *cz, const int *idx, const float *w,
                               const float *add, const float *bias, float *out, float *stats_part,
                               float *ws /* f32[b*m*c], or NULL if cz is point-major f32[b,m,c] */, void *stream);
size_t p2pb_group_sub_stats_floats(int b, int c, int m, int u);
/* The statistics of the grouped tensor alone (two-layer set abstractions: the consuming GEMM, p2pb_pointwise_conv_pool_gather,
 * gathers its operand itself): zt f32[b,n,c], cxt f32[b,m,c] | NULL point-major, idx i32[b,m,u] ->
 * stats_part f32[b, p2pb_group_sub_stats_slots(m,u), c, 2] ({sum, sum of squares} of z[idx] - cx per slot of 128 positions). */
int p2pb_group_sub_stats_slots(int m, int u);
int p2pb_group_sub_stats(int b, int c, int n, int m, int u, const float *zt, const float *cxt, const int *idx,
                         float *stats_part, void *stream);
int p2pb_group_sub(int b, int c, int n, int m, int u, const float *z, const float *cx, const int *idx, float *out,
                   float *stats_part, float *ws /* f32[b*(n+m)*c], or NULL if z, cx are point-major */, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused shared point MLPs (SharedMLP models/pvcnn.py:162-205 = k=1 Conv1d/Conv2d -> AdaGN|GroupNorm ->
 * Swish, chained; set-abstraction neighbour max :414; Pnet2Stage max-pool :923,930). Inference only.
 * ------------------------------------------------------------------------------------------- */
size_t p2pb_pointwise_packed_floats(int cout, int cin);
int p2pb_pointwise_pack_weights(int cout, int cin, const float *w /* [cout][cin] */, float *wp, void *stream);
size_t p2pb_pointwise_stats_floats(int b, int cout, int npos);
/* out[b,cout,npos] = bias[cout] (+ bias_b[b,cout]) + W * xf(in[b,cin,npos]); xf / stats_part as in
 * p2pb_conv3d_k3_forward (stats_part f32[b, ceil(npos/256)*4, cout, 2]). bias, bias_b may be NULL. */
int p2pb_pointwise_conv_forward(int b, int cin, int cout, int npos, const float *in, const void *wp,
                                const float *bias, const float *bias_b, const float *in_scale,
                                const float *in_shift, int in_swish, int flags, float *out, float *stats_part,
                                void *stream);
/* flags bit 5: point-major output, out f32[b,npos,cout] (what p2pb_group_sub / p2pb_three_interpolate_add gather
 * whole rows from); stats_part must be NULL.
 * flags bit 2: wp is the split pack below and the GEMM runs in the split-operand form (p2pb_set_split_terms; bf16x6: three bf16 terms per fp32
 * operand, six MFMA products, fp32 accumulate -- see the conv3d split pack); needs npos % 4 == 0 and
 * 16-byte aligned in/out. Meant for the matrix-bound layers (wide channel counts). */
size_t p2pb_pointwise_split_packed_bytes(int cout, int cin);
int p2pb_pointwise_pack_weights_split(int cout, int cin, const float *w /* [cout][cin] */, void *wp, void *stream);
/* flags bit 7 (with bit 2, f16x3 arithmetic, 16-byte rows, no in_fold / out_acc): a NARROW layer (cin or cout < 128) on
 * the split pack -- the register-tiled kernel of the fp32 pack with its products on the 16-bit matrix pipe (three MFMAs of
 * K = 16 instead of eight exact-fp32 ones of K = 2 per 32x32x16 block); outputs, statistics and minmax in the layout of the
 * fp32-pack form (p2pb_pointwise_minmax_floats takes the same flags). */
/* The same GEMM with the max-pool that follows the layer (set abstraction: max over the pool_u = 4..64
 * neighbours, models/pvcnn.py:414; Pnet2Stage: pool_u = 0, max over all positions, :923,930) prepared in the
 * epilogue: minmax receives {min, max} of the raw output per pooling group (pool_u > 0: f32[b,cout,npos/pool_u,2];
 * pool_u == 0: per-wave partials f32[b, nslots, cout, 2], nslots from p2pb_pointwise_minmax_floats) and p2pb_minmax_act turns them into the
 * pooled activations once the norm parameters are known (the activations are quasi-convex, so the max sits at
 * the min or the max of the pre-activation). out may be NULL: the layer output is then never written.
 * Needs npos % 4 == 0 and 16-byte aligned in/out (p2pb_pointwise_pool_supported), else P2PB_EINVAL. */
int p2pb_pointwise_pool_supported(int npos, int pool_u);
size_t p2pb_pointwise_minmax_floats(int b, int cout, int npos, int pool_u, int flags);
int p2pb_pointwise_conv_pool_forward(int b, int cin, int cout, int npos, const float *in, const void *wp,
                                     const float *bias, const float *bias_b, const float *in_scale,
                                     const float *in_shift, int in_swish, int flags, float *out, float *stats_part,
                                     int pool_u, float *minmax, void *stream);
/* The same for the grouped tensor of a set abstraction WITHOUT building it: operand[ci, (mi, ui)] = zt[b, idx[b,mi,ui], ci]
 * - cxt[b, mi, ci] gathered on load from the point-major rows zt f32[b,n,cin] / cxt f32[b,m,cin] (or NULL) -- exactly what
 * p2pb_group_sub would have written as f32[b,cin,m*u] (call it with out = NULL for the statistics in_scale / in_shift are
 * folded from). Split pack, f16x3 arithmetic, cin % 8 == 0; minmax f32[b,cout,m,2]; the output itself is never stored. */
int p2pb_pointwise_conv_pool_gather(int b, int cin, int cout, int n, int m, int u, const float *zt, const float *cxt,
                                    const int *idx, const void *wp_split, const float *bias, const float *in_scale,
                                    const float *in_shift, int in_swish, float *stats_part, float *minmax, void *stream);
/* nn.Linear on a handful of rows (the per-evaluation Linears: AdaGN styles models/modules.py:337-345, time embedding
 * models/unet_pvc.py:108-112, the global embedding's per-sample bias models/pvcnn.py:926): out[b,co] = bias[co] + sum_ci
 * w[co,ci] x[b,ci]. x f32[b,cin] with row pitch x_stride (floats), w f32[cout,cin] with row pitch w_stride (a column slice of
 * a wider matrix is fine), bias f32[cout] or NULL, out f32[b,cout] with row pitch out_stride. cin % 4 == 0, 16-byte aligned
 * rows. No scratch memory (safe in concurrently replayed hipGraphs, unlike a BLAS call with a per-stream workspace). */
int p2pb_linear_rows(int b, int cin, int cout, const float *x, long x_stride, const float *w, long w_stride,
                     const float *bias, float *out, long out_stride, void *stream);
/* y = max(act(scale*min+shift), act(scale*max+shift)); nslots == 0: minmax f32[b,c,m,2] -> y f32[b,c,m];
 * nslots > 0: minmax f32[b,nslots,c,2] (reduced over nslots first) -> y f32[b,c] */
int p2pb_minmax_act(int b, int c, int m, int nslots, const float *minmax, const float *scale, const float *shift,
                    int swish, float *y, void *stream);
/* the global max-pool form (minmax f32[b,nslots,c,2], nslots >= 1 -> y f32[b,c]) with the GroupNorm in front of the activation
 * folded in (Pnet2Stage, models/pvcnn.py:905-932: MyGroupNorm -> Swish -> max over the points): part f32[b,nslots_st,c,2] = the
 * producing GEMM's statistics partials + the norm's parameters; scale / shift f32[b,c] are OUTPUTS (p2pb_gn_affine_params'
 * values and bits: the next GEMM applies them to the same tensor on load). c <= 4096. */
int p2pb_minmax_act_pool_gn(int b, int c, int nslots, const float *minmax, const float *part, int nslots_st,
                            double count_per_channel, int groups, const float *gamma, const float *beta, const float *style,
                            int style_stride, float eps, int swish, float *scale, float *shift, float *y, void *stream);
/* y[b,c,p] = act(x*scale[b,c] + shift[b,c]) (+ residual[b,c,p]); act = Swish when swish != 0 */
int p2pb_affine_act(int b, int c, int npos, const float *x, const float *scale, const float *shift, int swish,
                    const float *residual, float *y, void *stream);
/* y[b,c,m] = max_{k<u} act(x[b,c,m,k]*scale + shift), u a power of two <= 64; u == 0: y[b,c] = max over m */
int p2pb_affine_act_max(int b, int c, int m, int u, const float *x, const float *scale, const float *shift,
                        int swish, float *y, void *stream);

/* ---- LinearAttention core (models/modules.py:165-194: `global_att`, models/unet_pvc.py:124-125,234-244) ----
 * replaces k.softmax(-1) + the two torch.einsum calls (:186-188) between the to_qkv and to_out 1x1 convolutions
 * (which run on p2pb_pointwise_conv_forward). qkv f32[b, 3*heads*dim_head, n] in the reference's channel order
 * (q | k | v) x heads x dim_head; out f32[b, heads*dim_head, n]; dim_head must be 32 (the reference's only value).
 * ctx (optional, f32[b, heads, 32, 32]) receives softmax(k) v^T for the backward pass.
 * backward: grad_out f32[b, heads*32, n] -> grad_qkv f32[b, 3*heads*32, n]. */
int p2pb_linear_attention_forward(int b, int heads, int dim_head, int n, const float *qkv, float *out, float *ctx,
                                  void *stream);
int p2pb_linear_attention_backward(int b, int heads, int dim_head, int n, const float *qkv, const float *ctx,
                                   const float *grad_out, float *grad_qkv, void *stream);

/* ---- training: the squeeze-excite gate (csrc/normact.hip) ---------------------------------------------------------------
 * SE3d of the reference (models/modules.py:362-378): gate = sigmoid(W2 relu(W1 mean)); mean f32[b,c] = per-channel mean of the
 * voxel grid, w1 f32[hidden,c], w2 f32[c,hidden] (the two bias-free nn.Linear weights), c <= 1024, hidden <= 128.
 * forward -> hid f32[b,hidden] (post-ReLU, kept for backward), gate f32[b,c];
 * backward (dgate f32[b,c]) -> dmean f32[b,c], dw1 f32[hidden,c], dw2 f32[c,hidden]; ws: b * (c + hidden) floats.
 * Replaces the two cuBLAS GEMMs + ReLU + sigmoid and their autograd of the eager module; deterministic. */
int p2pb_se_gate_forward(int b, int c, int hidden, const float *mean, const float *w1, const float *w2, float *hid, float *gate,
                         void *stream);
int p2pb_se_gate_backward(int b, int c, int hidden, const float *mean, const float *w1, const float *w2, const float *hid,
                          const float *gate, const float *dgate, float *dmean, float *dw1, float *dw2, float *ws, void *stream);

/* Max over the last axis with its arg-max, and the backward that writes the whole input gradient (csrc/normact.hip, ABI 6):
 * the neighbour max of a set abstraction (models/pvcnn.py:414) and Pnet2Stage's global max-pools (:923,930) in train().
 *   x f32[rows,u] -> y f32[rows], idx i32[rows] (first position of the maximum; a NaN in the row is returned);
 *   gy f32[rows], idx -> gx f32[rows,u] = gy at idx, 0 elsewhere (no pre-zeroing). */
int p2pb_row_max_forward(long rows, int u, const float *x, float *y, int *idx, void *stream);
int p2pb_row_max_backward(long rows, int u, const float *gy, const int *idx, float *gx, void *stream);

/* ---- training: weight gradients of the dense layers (csrc/wgrad.hip) --------------------------------------
 * What cuDNN / cuBLAS compute in the reference's backward pass for nn.Conv3d(k 3, pad 1) (models/pvcnn.py:265-282)
 * and the k = 1 Conv1d / Conv2d layers (models/pvcnn.py:162-205, 803-823): exact-fp32 MFMA GEMMs over the voxel /
 * position index, split over K with a deterministic reduction (ws = scratch for the partials).
 *   x f32[b,cin,r,r,r], dy f32[b,cout,r,r,r]  ->  dw f32[cout,cin,3,3,3], db f32[cout] (db may be NULL); r in {4,8,16,32}
 *   x f32[b,cin,npos],  dy f32[b,cout,npos]   ->  dw f32[cout,cin],       db f32[cout] (db may be NULL)
 * (data gradients: p2pb_conv3d_k3_forward_ex on dy with the point-reflected, channel-swapped weight;
 *  p2pb_pointwise_conv_forward with the transposed weight)
 * math: 0 = bf16x3 split operands (default: torch's "high" float32 matmul precision, which the reference selects in
 * train.py:221; 16 significand bits, the class of its TF32 kernels), 1 = bf16x6 (fp32-faithful), 2 = exact-fp32 MFMA.
 * Size limit of the bf16 forms: they use 32-bit buffer offsets, so a convolution batch with b * max(cin, cout) * r^3 * 4 >= 2 GiB
 * runs form 2 instead (both functions below apply the same rule; nothing is refused). */
size_t p2pb_conv3d_k3_wgrad_ws_floats(int b, int cin, int cout, int r, int math);
int p2pb_conv3d_k3_wgrad(int b, int cin, int cout, int r, const float *x, const float *dy, float *dw, float *db,
                         float *ws, int math, void *stream);
/* The same weight gradient for a PVConv's FIRST convolution, whose operand x is zero outside the occupied voxels: K runs over
 * the occupied voxels only (cnt i32[b, r^3] = avg_voxelize's counts; n = points per cloud, an upper bound of their number).
 * Exact fp32 products, deterministic. ws: p2pb_conv3d_k3_wgrad_occ_ws_floats(b, cin, cout, r, n) floats. db may be NULL. */
size_t p2pb_conv3d_k3_wgrad_occ_ws_floats(int b, int cin, int cout, int r, int n);
int p2pb_conv3d_k3_wgrad_occ(int b, int cin, int cout, int r, int n, const float *x, const float *dy, const int *cnt, float *dw,
                             float *db, float *ws, void *stream);
size_t p2pb_pointwise_wgrad_ws_floats(int b, int cin, int cout, int npos, int math);
int p2pb_pointwise_wgrad(int b, int cin, int cout, int npos, const float *x, const float *dy, float *dw, float *db,
                         float *ws, int math, void *stream);

/* ---- room pipeline (denoise_room.py) ---------------------------------------------------------
 * exact radius query = sklearn.neighbors.KDTree.query_radius as used at denoise_room.py:459-464: for every patch
 * centre the indices of ALL points with |p - c|^2 <= r^2 (fp32, fma(dz,dz,fma(dy,dy,dx*dx))), ascending. Ragged, so two
 * passes: p2pb_radius_count -> counts i32[s]; the caller prefix-sums them into offsets i64[s] (exclusive) and calls
 * p2pb_radius_fill -> out i32[total]. centers f32[s,3], points f32[n,3] (point-major). */
int p2pb_radius_count(int s, int n, const float *centers, const float *points, float radius, int *counts, void *stream);
int p2pb_radius_fill(int s, int n, const float *centers, const float *points, float radius, const long long *offsets,
                     int *out, void *stream);
/* running-mean merge of overlapping patch predictions (update_prediction_noisy_batches, denoise_room.py:263-289):
 * pred f32[npatch,k,3], idx i32[npatch,k] (room point of each patch point), cuts i32[npatch] (only the first cuts[p]
 * entries of patch p count). sums f64[n,3] / counts i32[n] are accumulators zeroed by the caller once per room; any
 * number of batches add into them; p2pb_merge_finish writes sums / counts (or `original` where counts == 0). */
int p2pb_merge_accumulate(int npatch, int k, const float *pred, const int *idx, const int *cuts, double *sums,
                          int *counts, void *stream);
int p2pb_merge_finish(int n, const double *sums, const int *counts, const float *original, float *out, void *stream);

/* ---- packs of the ADJOINT (data-gradient) operator, read straight from the forward layer's weight ----
 * dX of a layer is the same convolution kernel on dY with the transposed weight (and the taps point-reflected for the
 * 3x3x3 convolution). These pack variants take the FORWARD weight -- w_forward f32[cin][cout][27] resp. f32[cin][cout],
 * where cout / cin are the adjoint operator's (cout = the layer's input channels) -- and produce the pack the forward
 * kernels take; same sizes and arithmetic selection as the non-adjoint functions (p2pb_*_packed_bytes / _floats).
 * Replaces the flipped / transposed weight copies an autograd backward would make (the reference gets dX from
 * cuDNN / cuBLAS: models/pvcnn.py:265-282 Conv3d, :162-205 SharedMLP). -> 0 or P2PB_EINVAL */
int p2pb_conv3d_k3_pack_weights_split_adjoint(int cout, int cin, const float *w_forward, void *wt_split, void *stream);
int p2pb_pointwise_pack_weights_adjoint(int cout, int cin, const float *w_forward, float *wp, void *stream);
int p2pb_pointwise_pack_weights_split_adjoint(int cout, int cin, const float *w_forward, void *wp, void *stream);

/* ---- optimiser tail of a training step: clip_grad_norm_ + Adam / AdamW in three launches (csrc/optim.hip) ----
 * replaces torch.nn.utils.clip_grad_norm_(model.parameters(), clip) + torch.optim.AdamW/Adam.step() of the reference's
 * loop (train.py:127-133, models/model_loader.py:13-33); same arithmetic, per element, as torch's _single_tensor_adam.
 *   table   device array of n_tensors entries {float *param, *grad, *exp_avg, *exp_avg_sq; int64 numel}
 *           (p2pb_optim_entry_bytes() each)
 *   chunks  device int32[nchunks][2] = (tensor index, chunk index within the tensor), p2pb_optim_chunk() elements per chunk
 *   partial device f64[nchunks] scratch
 *   ctl     device f64[8]: [0] updates applied so far (in/out), [1] learning rate (in: the host writes it, so a captured
 *           graph follows a scheduler), [2] gradient norm before clipping (out), [3] clip coefficient applied (out),
 *           [4] 1.0 when this update was skipped (out), [5], [6] bias corrections (scratch)
 *   max_norm <= 0: no clipping (gradients are not rewritten). decoupled: AdamW (p *= 1 - lr wd) / Adam (g += wd p).
 *   skip_nonfinite: a non-finite gradient norm leaves parameters, moments and the step count untouched (what the
 *           GradScaler of the reference's loop does to optimizer.step()).
 *   amax    NULL, or device u32[ntensors]: bits of max |param| per tensor AFTER this update (0 for an empty / all-zero
 *           tensor) -- what p2pb_*_pack_weights_split_amax take, so that the next forward's weight packs need no reduction
 *           launch of their own
 * -> 0 or a hipError_t code */
size_t p2pb_optim_entry_bytes(void);
int p2pb_optim_chunk(void);
int p2pb_optim_clip_adam_step(int nchunks, const void *table, const int *chunks, double *partial, double *ctl, double max_norm,
                              double beta1, double beta2, double eps, double weight_decay, int decoupled, int skip_nonfinite,
                              unsigned *amax, int ntensors, void *stream);
/* p2pb_conv3d_k3_pack_weights_split / p2pb_pointwise_pack_weights_split with max |w| (float bits, device) supplied by the
 * caller instead of reduced in front of the pack: the same pack for the same value. -> 0 or P2PB_EINVAL */
int p2pb_conv3d_k3_pack_weights_split_amax(int cout, int cin, const float *w, void *wt_split, const unsigned *amax_bits,
                                           void *stream);
int p2pb_pointwise_pack_weights_split_amax(int cout, int cin, const float *w, void *wp, const unsigned *amax_bits, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* P2PB_HIP_H */
