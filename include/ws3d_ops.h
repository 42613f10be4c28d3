/*
 * ws3d_ops.h -- C ABI of libws3d_hip.so: the MI355X (gfx950) replacement for the
 * three CUDA extension modules of hlesmqh/WS3D's PointNet++ / roipool3d / iou3d
 * hot path.
 *
 * The reference has no C ABI: its boundary is three pybind11 modules
 * (pointnet2_cuda, iou3d_cuda, roipool3d_cuda) whose functions take pre-allocated
 * at::Tensor objects.  Every entry point below replaces one of those functions
 * (cited as file:line relative to the reference checkout), with the tensors
 * flattened to plain device pointers + sizes + a stream.  INTEGRATION.md shows the
 * Python (ctypes) binding a reference maintainer adds; ws3d_amd/compat.py is that
 * binding, exporting modules with the reference's exact function names.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to C-contiguous row-major arrays; data are
 *     float32, indices int32 (the reference's dtypes), unless stated otherwise;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All work
 *     is enqueued asynchronously on it; no entry point synchronises the device,
 *     allocates, or frees (scratch comes from caller-provided workspaces sized by
 *     the *_workspace_bytes() queries);
 *   - return value: 0 on success, a negative WS3D_E_* code on failure (never
 *     exit()s, unlike the reference: sampling_gpu.cu:39-43, iou3d.cpp:13-21);
 *     ws3d_last_error() returns a thread-local message for the last failure;
 *   - arithmetic contract (bit-exact index outputs): DESIGN.md section 4.
 */
#ifndef WS3D_OPS_H
#define WS3D_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WS3D_OK 0
#define WS3D_E_INVALID (-1)   /* bad size / null pointer                        */
#define WS3D_E_LAUNCH (-2)    /* hipGetLastError() != hipSuccess after a launch */
#define WS3D_E_WORKSPACE (-3) /* workspace missing or too small                 */
#define WS3D_E_UNSUPPORTED (-4)

#if defined(__GNUC__)
#define WS3D_API __attribute__((visibility("default")))
#else
#define WS3D_API
#endif

typedef void *ws3d_stream_t;

/* bumped whenever an entry point is added or a signature changes (5: ws3d_three_nn_wq, ws3d_roipool3d_ws / ws3d_roipool3d_workspace_bytes, ws3d_furthest_point_sampling_nested_chain, ws3d_ball_query_pairs2; 4: ws3d_pgather_gemm3_compact, ws3d_qinterp_gemm; 3: ws3d_topk_sorted_ws / ws3d_topk_workspace_bytes; 2: launch gates
 * of the SharedMLP kernels, ws3d_sa_mlp3_pool_lists, ws3d_ball_query_pairs, ws3d_three_nn_w; 1: rounds 1-2); ws3d_amd/_lib.py refuses a library whose version differs from the header it was written against */
#define WS3D_ABI_VERSION 6
WS3D_API int ws3d_abi_version(void);
/* Launch-geometry knobs of the persistent kernels (round 6): key 0 = workgroups of ws3d_chain_mlp3, 1 = of ws3d_mlp2_rows, 2 = of
 * ws3d_sa_mlp3_pool_compact, 3 = of ws3d_qinterp_gemm, 4 = of ws3d_compact_mlp_pair kinds 2 / 1 (along x), 5 = waves per 64-centre tile of ws3d_ball_query_pairs2 launches that do not fill the chip (8; anything else: 16); value 0 restores the built-in choice, a negative value only reads.  Returns the
 * previous value (WS3D_E_INVALID for an unknown key).  Speed only: results do not depend on these. Process-wide, not thread-safe. */
WS3D_API int ws3d_tune(int key, int value);
/* Squared-distance convention this library was BUILT with (csrc/common.h WS3D_DIST_MODE; the reference spells
 * dx*dx + dy*dy + dz*dz, sampling_gpu.cu:133 / ball_query_gpu.cu:33 / interpolate_gpu.cu:36, and nvcc's contraction of it
 * cannot be captured without a CUDA device): 0 = fma(dz,dz,fma(dx,dx,dy*dy)) [default], 1 = no contraction,
 * 2 = fma(dz,dz,fma(dy,dy,dx*dx)).  libws3d_hip_dm1.so / _dm2.so are the other builds.                          */
WS3D_API int ws3d_dist_mode(void);
WS3D_API const char *ws3d_last_error(void);
/* name/CU count/LDS bytes of the current device; any pointer may be NULL */
WS3D_API int ws3d_device_info(char *name, int name_len, int *cu_count, int *lds_bytes_per_block);

/* ---------------------------------------------------------------- pointnet2_cuda */

/* furthest_point_sampling_wrapper(b,n,m,xyz,temp,idx)   sampling.cpp:36-46 ->
 * furthest_point_sampling_kernel sampling_gpu.cu:93-253.
 * xyz (b,n,3); temp (b,n) in/out scratch, pre-filled by the caller (1e10,
 * pointnet2_utils.py:26) -- may be NULL (then 1e10 is assumed and nothing is written
 * back); idx (b,m) out.  idx[.,0] = 0.  Tie-break identical to the reference launch
 * geometry (block = opt_n_threads(n), cuda_utils.h:10-14).                          */
WS3D_API int ws3d_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int32_t *idx,
                                 ws3d_stream_t stream);

/* Fused a1+a2 of SURVEY section 8a: FPS that also emits the sampled coordinates
 * new_xyz (b,m,3) = xyz[idx]  (replaces furthest_point_sample + gather_operation +
 * transpose at pointnet2_modules.py:30-35).  new_xyz may be NULL.                  */
WS3D_API int ws3d_furthest_point_sampling_gather(int b, int n, int m, const float *xyz, float *temp,
                                        int32_t *idx, float *new_xyz, ws3d_stream_t stream);

/* FPS + gather of a cloud that is ALREADY in sampling order -- level l+1 of a set-abstraction stack samples the centres
 * level l selected, in the order it selected them (pointnet2_msg.py:56-70) -- where greedy selection returns idx = 0..m-1
 * unless two points tie for a maximum.  Same results as ws3d_furthest_point_sampling_gather for ANY input: the nesting is
 * verified in parallel (every point's running minimum against the picked point's, strictly), and a scene that fails the
 * check is sampled by a literal restatement of sampling_gpu.cu:93-209.  n <= 4096 (else WS3D_E_UNSUPPORTED), m <= n;
 * idx (b,m) and new_xyz (b,m,3) are both required (they double as scratch).  No reference counterpart.               */
WS3D_API int ws3d_furthest_point_sampling_nested(int b, int n, int m, const float *xyz, int32_t *idx, float *new_xyz,
                                        ws3d_stream_t stream);

/* ws3d_furthest_point_sampling_nested for a CHAIN of levels in four launches: level 0 samples m[0] of the n points of xyz (verified in
 * parallel as above), level l > 0 samples m[l] of level l-1's new_xyz (m non-increasing; host array of `levels` <= 5 counts, host arrays of
 * `levels` device pointers idx[l] (b, m[l]) / new_xyz[l] (b, m[l], 3)).  When level 0 verifies, the levels below it are verified with it
 * (their check is a subset of level 0's inequalities on the same floats): one kernel writes their prefixes; a scene that does not verify
 * takes the literal restatement of sampling_gpu.cu:93-209 level by level.  Same outputs as `levels` calls of the entry above.  (round 5) */
WS3D_API int ws3d_furthest_point_sampling_nested_chain(int b, int n, int levels, const int *m, const float *xyz, int32_t *const *idx,
                                                       float *const *new_xyz, ws3d_stream_t stream);


/* gather_points_wrapper(b,c,n,npoints,points,idx,out)   sampling.cpp:11-20 ->
 * sampling_gpu.cu:8-37.  points (b,c,n), idx (b,npoints) -> out (b,c,npoints).     */
WS3D_API int ws3d_gather_points(int b, int c, int n, int npoints, const float *points, const int32_t *idx,
                       float *out, ws3d_stream_t stream);

/* gather_points_grad_wrapper   sampling.cpp:23-33 -> sampling_gpu.cu:46-76.
 * grad_points (b,c,n) must be pre-zeroed by the caller (pointnet2_utils.py:67).    */
WS3D_API int ws3d_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                            const int32_t *idx, float *grad_points, ws3d_stream_t stream);

/* ball_query_wrapper(b,n,m,radius,nsample,new_xyz,xyz,idx)   ball_query.cpp:14-25 ->
 * ball_query_gpu.cu:9-67.  new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample),
 * pre-zeroed by the caller (pointnet2_utils.py:218); rows without a hit are left
 * untouched.  sorted: NULL, or the output of ws3d_sort_points_x for this xyz.       */
WS3D_API int ws3d_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                    const float *xyz, int32_t *idx, const void *sorted, ws3d_stream_t stream);
/* ws3d_ball_query that also writes the rows of centres WITHOUT a hit (as zeros: what the reference's callers see in their
 * zero-initialised idx tensor, pointnet2_utils.py:201), so that idx need not be cleared first.  ws3d extension.           */
WS3D_API int ws3d_ball_query_fill(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                         const float *xyz, int32_t *idx, const void *sorted, ws3d_stream_t stream);
/* ws3d_ball_query_fill + ws3d_compact_pairs in ONE launch: the lists idx (b, m, nsample) and their distinct (centre, source) pairs
 * rowc / rowsrc (b * m * nsample ints each, the first *total valid; *total must be ZERO on entry; compact rows of a centre
 * contiguous, centres in arrival order of their 64-centre workgroups).  Needs the fine-grid buffer of ws3d_sort_points_grid and
 * nsample <= 64 (the kernel that works with one wave per centre), else WS3D_E_UNSUPPORTED.  ws3d extension.                   */
WS3D_API int ws3d_ball_query_pairs(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int32_t *idx,
                          const void *sorted_grid, int32_t *rowc, int32_t *rowsrc, int32_t *total, ws3d_stream_t stream);

/* ws3d_ball_query_pairs for the TWO scales of a set-abstraction level in one launch: same xyz / new_xyz / fine-grid copy, per scale a
 * radius, an nsample (<= 64) and the four outputs of ws3d_ball_query_pairs (every `total` zero on entry).  Same lists and pair tables as
 * two calls; WS3D_E_UNSUPPORTED where the one-wave-per-centre kernel does not cover the call (then call per scale).  (round 5) */
WS3D_API int ws3d_ball_query_pairs2(int b, int n, int m, const float *new_xyz, const float *xyz, const void *sorted_grid,
                                    float radius0, int nsample0, int32_t *idx0, int32_t *rowc0, int32_t *rowsrc0, int32_t *total0,
                                    float radius1, int nsample1, int32_t *idx1, int32_t *rowc1, int32_t *rowsrc1, int32_t *total1,
                                    ws3d_stream_t stream);


/* Optional accelerator for ws3d_ball_query / ws3d_query_and_group (no reference counterpart):
 * a per-scene copy of xyz counting-sorted into uniform x cells (float4 {x,y,z,index} x n, a
 * 16-byte header and a cell-start table per scene; opaque to the caller).  With it each centre
 * scans only the cells overlapping |x - cx| < radius instead of all n points; results are
 * bit-identical.  ws3d_sorted_points_bytes() returns the buffer size, or 0 when the shape is not
 * supported (n > 16384): then pass sorted = NULL.  One binning serves every radius queried
 * against the same xyz.                                                                        */
WS3D_API size_t ws3d_sorted_points_bytes(int b, int n);
WS3D_API int ws3d_sort_points_x(int b, int n, const float *xyz, void *sorted, ws3d_stream_t stream);
/* Same buffer size and layout, but the points are binned into a near-square (x, z) grid (~2 points per
 * cell) instead of x slabs: for ws3d_three_nn ONLY (its search then visits square rings of cells around
 * the query, ~20 candidates instead of an x slab of ~100); not accepted by the ball-query entries.   */
WS3D_API int ws3d_sort_points_xz(int b, int n, const float *xyz, void *sorted, ws3d_stream_t stream);
/* Same buffer size, FINE (x, z) grid (up to 32768 near-square cells, ~0.4 m on a KITTI scene, 16-bit cell offsets): for
 * ws3d_ball_query / ws3d_query_and_group[_nlc] -- a centre scans, per grid row overlapping |z - cz| < r, the cells
 * overlapping |x - cx| < r instead of an x slab spanning all z (~5-30 candidates instead of 50-500 on lidar scenes);
 * results bit-identical to the full scan.  Preferred over ws3d_sort_points_x for the ball-query entries; NOT accepted by
 * ws3d_three_nn.                                                                                                   */
WS3D_API int ws3d_sort_points_grid(int b, int n, const float *xyz, void *sorted, ws3d_stream_t stream);
/* Several binning jobs in ONE launch (round 5): job i bins the (b, n[i], 3) cloud xyz[i] into sorted[i] (ws3d_sorted_points_bytes(b, n[i])
 * bytes each) -- kind[i] 0: as ws3d_sort_points_grid, 1: as ws3d_sort_points_xz; at most 8 jobs, 0 < n[i] <= 16384.  The arrays are host
 * arrays read before the call returns.  Same buffers as one call per job (the levels of a network and both flavours: 6 launches -> 1). */
WS3D_API int ws3d_sort_points_jobs(int b, int njobs, const int *n, const int *kind, const float *const *xyz, void *const *sorted,
                                   ws3d_stream_t stream);

/* group_points_wrapper(b,c,n,npoints,nsample,points,idx,out)   group_points.cpp:25-36
 * -> group_points_gpu.cu:47-86.  points (b,c,n), idx (b,npoints,nsample) ->
 * out (b,c,npoints,nsample).                                                        */
WS3D_API int ws3d_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                      const int32_t *idx, float *out, ws3d_stream_t stream);

/* group_points_grad_wrapper   group_points.cpp:11-22 -> group_points_gpu.cu:8-44.   */
WS3D_API int ws3d_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                           const int32_t *idx, float *grad_points, ws3d_stream_t stream);

/* Fused QueryAndGroup.forward (pointnet2_utils.py:241-264 = ball_query + 2x
 * grouping_operation + centre subtraction + cat).  xyz (b,n,3), new_xyz (b,m,3),
 * features (b,c,n) or NULL (c = 0) -> out (b, (use_xyz?3:0)+c, m, nsample) with
 * channel order [dx,dy,dz, features...].  idx_out (b,m,nsample) may be NULL; when
 * given it receives exactly what ws3d_ball_query would write into a zeroed idx.     */
WS3D_API int ws3d_query_and_group(int b, int n, int m, int c, float radius, int nsample, int use_xyz,
                         const float *xyz, const float *new_xyz, const float *features,
                         int32_t *idx_out, float *out, const void *sorted, ws3d_stream_t stream);

/* three_nn_wrapper(b,n,m,unknown,known,dist2,idx)   interpolate.cpp:14-23 ->
 * interpolate_gpu.cu:9-67.  unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3)
 * SQUARED distances, idx (b,n,3).  sorted_known: NULL (full scan) or the x-binned copy of
 * `known` from ws3d_sort_points_x / ws3d_sort_points_xz(b, m, known, ...) -- same result, the search then only visits
 * the known points whose x is closer than the running third-best distance.            */
WS3D_API int ws3d_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                  int32_t *idx, const void *sorted_known, ws3d_stream_t stream);

/* dist2 (rows,3) SQUARED distances from ws3d_three_nn -> weight (rows,3): the FP module's normalised
 * inverse-distance weights (pointnet2_modules.py:139-142) in one launch.  ws3d extension.            */
WS3D_API int ws3d_three_nn_weights(long rows, const float *dist2, float *weight, ws3d_stream_t stream);
/* ws3d_three_nn with ws3d_three_nn_weights folded into the search kernel's epilogue (same dist2 / idx, weight (b,n,3) as the
 * two-launch form, bit for bit): what the FP modules of ws3d_amd/fastpath.py call.                              */
WS3D_API int ws3d_three_nn_w(int b, int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx, float *weight,
                             const void *sorted_known, ws3d_stream_t stream);
/* ws3d_three_nn_w with its QUERIES taken in cell order: sorted_unknown = a binned copy of `unknown` (ws3d_sort_points_x / _grid / _xz
 * of the same tensor, any flavour: only its n float4 {x, y, z, bits(index)} per scene are read) or NULL.  Lane i searches for the
 * i-th binned point and writes row bits(index): the rows of dist2 / idx / weight are those of ws3d_three_nn_w bit for bit (a
 * query's result does not depend on the lane that runs it; interpolate_gpu.cu:9-52), the lanes of a wave walk cells of similar
 * density.  Applies to the binned search (sorted_known given, 3 <= m <= 16384); 0 < n <= 16384 when sorted_unknown is given. */
WS3D_API int ws3d_three_nn_wq(int b, int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx, float *weight,
                              const void *sorted_known, const void *sorted_unknown, ws3d_stream_t stream);
/* Several ws3d_three_nn_wq searches in ONE launch (round 5): job k takes the (b, n[k], 3) queries unknown[k] against the binned known set
 * sorted_known[k] (ws3d_sort_points_xz / _x of a (b, m[k], 3) cloud, 3 <= m[k] <= 4096) and writes dist2[k], idx[k], weight[k] (b, n[k], 3);
 * sorted_unknown (NULL, or per job NULL / a binned copy of the queries: cell order).  At most 4 jobs; host arrays, read before the call
 * returns.  Same rows as one call per job (the FP modules of a network: 3 launches -> 1).                                              */
WS3D_API int ws3d_three_nn_jobs(int b, int njobs, const int *n, const int *m, const float *const *unknown, const void *const *sorted_known,
                                float *const *dist2, int32_t *const *idx, float *const *weight, const void *const *sorted_unknown,
                                ws3d_stream_t stream);

/* The SA module's pool over nsample (pointnet2_modules.py:50, F.max_pool2d(kernel_size=[1, nsample]))
 * with the position of the maximum kept for the backward pass.  x (rows, nsample) -- the contiguous
 * (B,C,npoint,nsample) activation with rows = B*C*npoint -- -> out (rows), arg (rows) u8.  Window
 * scan rule `v > best || isnan(v)`: first position of the maximum, NaN propagates.  nsample <= 255.
 * _grad: grad_x (rows, nsample) = grad_out at arg, 0 elsewhere (every element written).
 * ws3d extension (replaces the library max-pool kernels in the training step).                  */
WS3D_API int ws3d_pool_nsample(long rows, int nsample, const float *x, float *out, uint8_t *arg, ws3d_stream_t stream);
WS3D_API int ws3d_pool_nsample_grad(long rows, int nsample, const float *grad_out, const uint8_t *arg, float *grad_x,
                                    ws3d_stream_t stream);

/* Last SharedMLP layer of a set-abstraction scale fused with the pool over nsample (ws3d extension, replaces
 * addmm + ws3d_rowmax_rows): out[g, 0:o] = act(max over the nsample rows of group g of x_rows . wt + bias).
 * x_rows (rows, k) row-major, wt (k, o) = W^T, rows % 64 == 0, o % 64 == 0, k % 4 == 0, nsample 16 or 32;
 * out (rows/nsample, o) with row stride out_stride.  fp32 matrix cores, fp32 accumulate.
 * WS3D_E_UNSUPPORTED for other shapes (the caller keeps the two-launch path).                       */
WS3D_API int ws3d_gemm_pool(long rows, int nsample, int k_dim, int o_dim, const float *x_rows, const float *wt,
                            const float *bias, int relu, float *out, int out_stride, const int32_t *gate, long gate_limit, ws3d_stream_t stream);

/* First SharedMLP layer of a set-abstraction scale with the grouping fused into the GEMM's A operand (no reference
 * counterpart; replaces QueryAndGroup -> Conv2d(1x1)+BN+ReLU, pointnet2_utils.py:241-264 + pytorch_utils.py:20-32, on
 * channels-last tensors): out (b*m*nsample, o) = relu?([feats[nbr] | xyz[nbr] - new_xyz] @ wt + bias) with feats (b,n,c),
 * xyz (b,n,3), new_xyz (b,m,3), nbr (b,m,nsample) int32 (ws3d_ball_query), wt (c+3, o) row-major with the three xyz rows
 * LAST, on the fp32 matrix cores.  c % 4 == 0, o % 64 == 0, b*m*nsample % 64 == 0, else WS3D_E_UNSUPPORTED.        */
WS3D_API int ws3d_gather_gemm(int b, int n, int m, int nsample, int c_feat, int o_dim, const float *feats, const float *xyz,
                     const float *new_xyz, const int32_t *nbr, const float *wt, const float *bias, int relu, float *out,
                     ws3d_stream_t stream);

/* The first TWO SharedMLP layers of a set-abstraction scale in one kernel (no reference counterpart): layer 1 as
 * ws3d_gather_gemm, its activation tile kept on chip, layer 2 (o1 -> o2) multiplied out of it:
 * out (b*m*nsample, o2) = relu2?(relu1?([f[nbr] | xyz[nbr] - centre] @ w1t + b1) @ w2t + b2).  w1t (c_feat+3, o1) with the
 * three coordinate rows LAST, w2t (o1, o2), both row-major.  o1 in {64, 128, 256}, o2 % 4 == 0, c_feat % 4 == 0,
 * rows % 64 == 0, else WS3D_E_UNSUPPORTED.                                                                             */
WS3D_API int ws3d_gather_gemm2(int b, int n, int m, int nsample, int c_feat, int o1, int o2, const float *feats, const float *xyz,
                      const float *new_xyz, const int32_t *nbr, const float *w1t, const float *b1, int relu1, const float *w2t,
                      const float *b2, int relu2, float *out, ws3d_stream_t stream);

/* Layer 1 of a set-abstraction SharedMLP without its per-pair product: W [f_j ; x_j - c] = W_f f_j + W_x (x_j - c), and
 * P = feats @ W_f (b * n rows, o1 columns at row stride p_stride: ONE GEMM over the points of a scene, by the caller) replaces the
 * product over the b * m * nsample (centre, sample) pairs.  Per pair: gather P's row of the neighbour, add the three-term xyz
 * product (w1x (3, o1): the x, y, z rows of W^T; centred coordinates, as pointnet2_utils.py:255-258) + b1, ReLU.
 *   ws3d_pgather_gemm2: ... then layer 2 (w2t (o1, o2)) as in ws3d_gather_gemm2 -> out (rows, o2); o1 in {64, 128}, m * nsample % 64
 *   ws3d_pgather_rows:  layer 1 alone -> out (rows, o1); o1 % 4, P / w1x / b1 / out 16-byte aligned
 * Same function as ws3d_gather_gemm(2) up to fp32 summation order.  ws3d extension, used by ws3d_amd/fastpath.py.          */
WS3D_API int ws3d_pgather_gemm2(int b, int n, int m, int nsample, int o1, int o2, const float *pmat, int p_stride, const float *xyz,
                       const float *new_xyz, const int32_t *nbr, const float *w1x, const float *b1, int relu1,
                       const float *w2t, const float *b2, int relu2, float *out, const int32_t *gate, long gate_limit, ws3d_stream_t stream);
WS3D_API int ws3d_pgather_rows(int b, int n, int m, int nsample, int o1, const float *pmat, int p_stride, const float *xyz, const float *new_xyz,
                      const int32_t *nbr, const float *w1x, const float *b1, int relu1, float *out, ws3d_stream_t stream);

/* First layer of a feature-propagation module without its product over the interpolated channels: interpolation is linear,
 * so Q = known_feats @ W_a (b * m rows, o columns, by the caller: a product over the KNOWN points) is interpolated instead of the
 * features:  out[r, :] = relu?( w0 Q[i0] + w1 Q[i1] + w2 Q[i2] + lin[r, :] )  with lin = skip @ W_b + bias (rows, o) from the
 * caller, or, lin == NULL and c1 <= 4 skip channels, skip (rows, c1) @ wb (c1, o) + bias evaluated inside.  idx / weight
 * (b, n, 3) as three_interpolate's (interpolate.cpp:37-56).  o % 4 == 0, float pointers 16-byte aligned.  Same function as
 * ws3d_interp_gemm up to fp32 summation order.  ws3d extension, used by ws3d_amd/fastpath.py.                              */
WS3D_API int ws3d_qinterp_rows(int b, int n, int m, int o, const float *q, const int32_t *idx, const float *weight, const float *lin,
                      const float *skip, int c1, const float *wb, const float *bias, int relu, float *out, ws3d_stream_t stream);

/* Launch gates (device-side dispatch between the compact and the dense form of a SharedMLP; ABI version 2).  Both forms are exact;
 * which one is faster depends on how full the ball-query lists are (break-even ~55 % distinct rows), a number that only exists
 * on the device and changes per batch -- a captured hipGraph cannot branch on the host.  So the caller launches BOTH forms and
 * every kernel of a form reads the pair total in its prologue:
 *   compact kernels (ws3d_pgather_gemm2_compact, ws3d_gemm_pool_compact, ws3d_pgather_gemm3_compact, ws3d_sa_mlp3_pool_compact): `limit` -- run iff *total <= limit
 *                  (limit < 0: always);
 *   dense kernels   (ws3d_pgather_gemm2, ws3d_gemm_pool, ws3d_sa_mlp3_pool_lists): `gate`, `gate_limit` -- run iff *gate > gate_limit
 *                  (gate == NULL: always), gate = the `total` word of ws3d_compact_pairs.
 * The form that does not run costs one launch whose workgroups return at once.  Exactly one of the two writes the result; with
 * the same limit on both sides the decision is consistent whatever the total.                                                 */

/* Compact (centre, sample) pairs.  A ball-query list holds its hits in ascending order, padded with the first one
 * (ball_query_gpu.cu:29-44); a padded row repeats row 0 of its centre in every SharedMLP layer and cannot change the maximum over
 * nsample, so the layers may run over the DISTINCT pairs only -- bit-identical results.
 *   ws3d_compact_pairs_count: cnt[c] = number of distinct neighbours of centre c (centres = b * m rows of nbr (centres, nsample))
 *   ws3d_compact_pairs_rows:  with incl = the INCLUSIVE prefix sum of cnt (by the caller): rowc[t] / rowsrc[t] = centre / source
 *                             point of compact row t, *total = incl[centres - 1]; rowc, rowsrc hold centres * nsample ints
 *   ws3d_pgather_gemm2_compact: ws3d_pgather_gemm2 over the compact rows (o1 in {64, 128, 256}) -> out (total rows, o2); launched for max_rows rows,
 *                             workgroups beyond *total return at once
 *   ws3d_gemm_pool_compact:   last layer + ReLU + max over each centre's rows by integer atomic max into out[centre, 0:o_dim]
 *                             (row stride out_stride), which the caller ZEROES first; the layer must end in a ReLU
 *   ws3d_pgather_gemm3_compact: the two above in ONE kernel (o1 in {64, 128}, o3 % 128 == 0, layer 2's tile kept in LDS: WS3D_E_UNSUPPORTED
 *                             beyond 159 KB): out[centre, 0:o3] by integer atomic max, bit-identical to the two-kernel form
 * ws3d extensions, used by ws3d_amd/fastpath.py.                                                                          */
 /* ws3d_compact_pairs: count + placement in ONE launch (*total must be ZERO on entry; the order of the compact rows is then arrival
 *  order of the 256-centre workgroups -- undefined, and irrelevant to every consumer above) */
WS3D_API int ws3d_compact_pairs(long centres, int nsample, const int32_t *nbr, int32_t *rowc, int32_t *rowsrc, int32_t *total, ws3d_stream_t stream);
WS3D_API int ws3d_compact_pairs_count(long centres, int nsample, const int32_t *nbr, int32_t *cnt, ws3d_stream_t stream);
WS3D_API int ws3d_compact_pairs_rows(long centres, int nsample, const int32_t *nbr, const int32_t *cnt, const int32_t *incl, int32_t *rowc,
                            int32_t *rowsrc, int32_t *total, ws3d_stream_t stream);
WS3D_API int ws3d_pgather_gemm2_compact(int b, int n, int m, long max_rows, int o1, int o2, const float *pmat, int p_stride, const float *xyz,
                               const float *new_xyz, const int32_t *rowc, const int32_t *rowsrc, const int32_t *total, const float *w1x,
                               const float *b1, int relu1, const float *w2t, const float *b2, int relu2, float *out, long limit, ws3d_stream_t stream);
WS3D_API int ws3d_pgather_gemm3_compact(int b, int n, int m, long max_rows, int o1, int o2, int o3, const float *pmat, int p_stride, const float *xyz,
                               const float *new_xyz, const int32_t *rowc, const int32_t *rowsrc, const int32_t *total, const float *w1x,
                               const float *b1, int relu1, const float *w2t, const float *b2, int relu2, const float *w3t, const float *b3,
                               float *out, int out_stride, long limit, ws3d_stream_t stream);
WS3D_API int ws3d_gemm_pool_compact(long max_rows, int k_dim, int o_dim, const float *x_rows, const int32_t *rowc, const int32_t *total,
                           const float *wt, const float *bias, float *out, int out_stride, long limit, ws3d_stream_t stream);
/* The two scales of a set-abstraction level in ONE launch (round 5).  Each block restates the arguments of the single-scale entries:
 * kind 3 = ws3d_pgather_gemm3_compact on both blocks, kind 2 = ws3d_pgather_gemm2_compact (its rows go to `mid`, (max_rows, o2)),
 * kind 1 = ws3d_gemm_pool_compact (x_rows = mid, k = o2, o = o3, wt = w3t, bias = b3).  Both blocks must name the same o1 (kinds 3, 2);
 * pmat already points at the scale's first column.  Same results as the two single-scale launches.                                 */
typedef struct ws3d_compact_mlp_args {
    int b, n, m;
    long max_rows;
    int o1, o2, o3;
    const float *pmat;
    int p_stride;
    const float *xyz, *new_xyz;
    const int32_t *rowc, *rowsrc, *total;
    const float *w1x, *b1;
    int relu1;
    const float *w2t, *b2;
    int relu2;
    const float *w3t, *b3;
    float *mid;
    float *out;
    int out_stride;
    long limit;
} ws3d_compact_mlp_args;
WS3D_API int ws3d_compact_mlp_pair(int kind, const ws3d_compact_mlp_args *scale0, const ws3d_compact_mlp_args *scale1, ws3d_stream_t stream);
/* The same function as kind 3 (three layers + pool of one or two scales over their compact rows; scale1 may be NULL) on the round-6
 * kernel (csrc/chain_mlp.hip, ABI 6): a wave owns 32 rows from the gather to the pooled atomics, the activations stay in registers
 * between the layers (transposed products + v_permlane32_swap), the weights of both scales are resident in LDS, 32-row tiles are handed
 * out by ticket counters.  blob0 / blob1: the scales' weights packed by ws3d_chain_mlp3_pack (once per weight set,
 * ws3d_chain_mlp3_blob_floats(o2) floats, 16-byte aligned); ticket: ws3d_chain_mlp3_ticket_ints() int32, ZERO on entry, consumed.
 * Covered: o1 = 64, o2 <= 96, o3 = 128, 16-byte aligned P rows (SA2 of the Stage-1 network: pointnet2_msg.py's MLPS[1]); anything else
 * WS3D_E_UNSUPPORTED, nothing launched.  workgroups: 0 = one per compute unit.  Pooled rows bit-identical to
 * ws3d_pgather_gemm3_compact's (the same fmaf chains in the same k order). */
WS3D_API int ws3d_chain_mlp3_ticket_ints(void);
WS3D_API size_t ws3d_chain_mlp3_blob_floats(int o2);
WS3D_API int ws3d_chain_mlp3_pack(const ws3d_compact_mlp_args *scale, float *blob, ws3d_stream_t stream);
WS3D_API int ws3d_chain_mlp3(const ws3d_compact_mlp_args *scale0, const ws3d_compact_mlp_args *scale1, const float *blob0, const float *blob1,
                             int32_t *ticket, int workgroups, ws3d_stream_t stream);

/* ws3d_sa_mlp3_pool over compact pairs (see ws3d_compact_pairs_*): the first level's three-layer SharedMLP (16-16-32 or 32-32-64,
 * every layer with bias + ReLU) on rows [x_j - c, f_j] built here from xyz (b, n, 3), new_xyz (b, m, 3) and the ONE feature
 * channel feat (b, n); max over each centre's rows by integer atomic max into out[centre, 0:c3] (row stride out_stride), which
 * the caller ZEROES.  Other widths return WS3D_E_UNSUPPORTED.  ws3d extension, used by ws3d_amd/fastpath.py.               */
WS3D_API int ws3d_sa_mlp3_pool_compact(int b, int n, int m, long max_rows, int c1, int c2, int c3, const float *xyz, const float *new_xyz,
                              const float *feat, const int32_t *rowc, const int32_t *rowsrc, const int32_t *total, const float *w1t,
                              const float *b1, const float *w2t, const float *b2, const float *w3t, const float *b3, float *out,
                              int out_stride, long limit, ws3d_stream_t stream);

/* The same three layers + pool over ALL rows of the neighbour lists nbr (b, m, nsample), rows built here like above (no grouped
 * tensor), the pool over registers and STORED (no atomics, out need not be zeroed): replaces ws3d_query_and_group_nlc +
 * ws3d_sa_mlp3_pool, bit-identical to it and to the compact form.  nsample 16 | 32, widths 16-16-32 | 32-32-64.              */
WS3D_API int ws3d_sa_mlp3_pool_lists(int b, int n, int m, int nsample, int c1, int c2, int c3, const float *xyz, const float *new_xyz,
                            const float *feat, const int32_t *nbr, const float *w1t, const float *b1, const float *w2t, const float *b2,
                            const float *w3t, const float *b3, int relu3, float *out, int out_stride, const int32_t *gate, long gate_limit,
                            ws3d_stream_t stream);

/* First layer of a feature-propagation module with three_interpolate and the skip concatenation fused into the GEMM's A
 * operand (no reference counterpart; replaces three_interpolate -> torch.cat -> Conv2d(1x1)+BN+ReLU,
 * pointnet2_modules.py:138-155, on channels-last tensors): out (b*n, o) = relu?([w0 f[i0] + w1 f[i1] + w2 f[i2] | u] @ wt +
 * bias) with f = known_feats (b,m,c2), u = unknown_feats (b,n,c1) or NULL (c1 = 0), idx / weight (b,n,3) from ws3d_three_nn
 * + ws3d_three_nn_weights, wt (c2+c1, o) row-major.  c2 % 4 == 0, o % 64 == 0, b*n % 64 == 0, else WS3D_E_UNSUPPORTED. */
WS3D_API int ws3d_interp_gemm(int b, int n, int m, int c2, int c1, int o_dim, const float *known_feats, const float *unknown_feats,
                     const int32_t *idx, const float *weight, const float *wt, const float *bias, int relu, float *out,
                     ws3d_stream_t stream);

/* Weight gradient of a 1x1 convolution on channels-first tensors (the Conv1d / Conv2d of every SharedMLP
 * block, pytorch_utils.py:35-101): grad_w (o, c) = sum_b sum_l grad_out[b, o, l] * x[b, c, l].  fp32 matrix
 * cores, the (scene, l-range) slices of the sum are added in a fixed order: bit-reproducible (the library's
 * split-K kernels use atomics).  workspace: ws3d_conv1x1_wgrad_workspace_bytes(b, o, c, l) bytes.  ws3d extension. */
WS3D_API size_t ws3d_conv1x1_wgrad_workspace_bytes(int b, int o, int c, long l);
WS3D_API int ws3d_conv1x1_wgrad(int b, int o, int c, long l, const float *grad_out, const float *x, float *grad_w,
                                void *workspace, size_t workspace_bytes, ws3d_stream_t stream);

/* Training-mode BatchNorm (+ ReLU) of a channels-first activation x (b, c, l) -- the norm + activation
 * of every conv -> BatchNorm -> ReLU block of the reference (pytorch_utils.py:35-101, nn.BatchNorm1d/2d
 * in train() mode followed by nn.ReLU).  fwd: batch statistics per channel over b*l values (biased
 * variance for the normalisation, unbiased for running_var), y = relu(((x-mean)*invstd)*gamma+beta),
 * running_mean/var (nullable) updated with `momentum`, *num_batches_tracked (nullable, one int64 on
 * the device) incremented, save_mean/save_invstd (c) kept for bwd.
 * bwd: dx (b,c,l), dgamma (c), dbeta (c) from dy; the ReLU mask is re-derived from x.  relu = 0: plain
 * BatchNorm.  Reductions run in a fixed order (fp64 partials): bit-reproducible run to run.
 * workspace: ws3d_bn_workspace_bytes(b, c, l) bytes, 8-byte aligned.  ws3d extension (replaces the
 * library BatchNorm + ReLU kernels in the training step).                                        */
WS3D_API size_t ws3d_bn_workspace_bytes(int b, int c, long l);
WS3D_API int ws3d_bn_relu_train_fwd(int b, int c, long l, const float *x, const float *gamma, const float *beta, float eps,
                                    float momentum, int relu, float *running_mean, float *running_var,
                                    int64_t *num_batches_tracked, float *y, float *save_mean, float *save_invstd,
                                    void *workspace, size_t workspace_bytes, ws3d_stream_t stream);
WS3D_API int ws3d_bn_relu_train_bwd(int b, int c, long l, const float *x, const float *dy, const float *gamma,
                                    const float *beta, const float *save_mean, const float *save_invstd, int relu, float *dx,
                                    float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes, ws3d_stream_t stream);

/* three_interpolate_wrapper(b,c,m,n,points,idx,weight,out)   interpolate.cpp:26-39 ->
 * interpolate_gpu.cu:77-117.  points (b,c,m), idx/weight (b,n,3) -> out (b,c,n).    */
WS3D_API int ws3d_three_interpolate(int b, int c, int m, int n, const float *points, const int32_t *idx,
                           const float *weight, float *out, ws3d_stream_t stream);

/* three_interpolate_grad_wrapper(b,c,n,m,grad_out,idx,weight,grad_points)
 * interpolate.cpp:41-53 -> interpolate_gpu.cu:120-160.  grad_points pre-zeroed.     */
WS3D_API int ws3d_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                const int32_t *idx, const float *weight, float *grad_points,
                                ws3d_stream_t stream);

/* Deterministic backward.  Same results as the *_grad entry points above up to the
 * summation order, which is FIXED here: every gradient element is the sum of its contributions in
 * ascending slot order (slot = m*nsample+s, or point*3+k), i.e. bit-identical to the sequential
 * loop `for slot: dst[idx[slot]] += v[slot]` and to itself from run to run (the reference's float
 * atomicAdd scatter is neither).  grad_points is fully written (no pre-zeroing needed).
 * workspace: ws3d_scatter_workspace_bytes(b, c, n_targets, slots_per_scene) bytes of device memory,
 * n_targets = n (group/gather) or m (three_interpolate), slots_per_scene = npoints*nsample or n*3.
 * gather_points_grad == group_points_grad_det with nsample = 1.                                  */
WS3D_API size_t ws3d_scatter_workspace_bytes(int b, int c, int n_targets, long slots_per_scene);
WS3D_API int ws3d_group_points_grad_det(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                        const int32_t *idx, float *grad_points, void *workspace,
                                        size_t workspace_bytes, ws3d_stream_t stream);
WS3D_API int ws3d_three_interpolate_grad_det(int b, int c, int n, int m, const float *grad_out, const int32_t *idx,
                                             const float *weight, float *grad_points, void *workspace,
                                             size_t workspace_bytes, ws3d_stream_t stream);

/* y[b,o,l] = relu?(y[b,o,l] + bias[o]) in place, one pass: the epilogue of the SharedMLP
 * (1x1 conv + eval-mode BatchNorm folded into one GEMM, pytorch_utils.py:20-101).  ws3d extension. */
WS3D_API int ws3d_bias_act_inplace(int b, int o_ch, long l, int relu, float *y, const float *bias,
                                   ws3d_stream_t stream);

/* out[b,o,m] = relu?(max_s y[b,o,m,s] + bias[o]) (bias may be NULL): the set-abstraction pool
 * F.max_pool2d(new_features, kernel_size=[1, nsample]) (pointnet2_modules.py:51-58) fused with
 * the last SharedMLP layer's bias + ReLU, which commute with the max exactly.  ws3d extension. */
WS3D_API int ws3d_rowmax_bias_act(int b, int o_ch, long m, int s, int relu, const float *y, const float *bias,
                                  float *out, ws3d_stream_t stream);

/* Channels-last ("nlc") variants for a row-major SharedMLP chain (ws3d extension, ws3d_amd/fastpath.py):
 * features are (b, points, C) instead of the reference's (b, C, points); idx / xyz arguments and
 * every index result are identical to the channels-first entry points.
 *   ws3d_query_and_group_nlc : out (b, m, nsample, 3*use_xyz + c), row = [dx, dy, dz, features...]
 *   ws3d_three_interpolate_nlc: out row p = sum_k weight[p,k] * feats[idx[p,k], :], row stride out_stride
 *   ws3d_rowmax_rows          : out row r = max over rows r*ns .. r*ns+ns-1 of y (rows, o_ch)        */
WS3D_API int ws3d_query_and_group_nlc(int b, int n, int m, int c, float radius, int nsample, int use_xyz,
                                      const float *xyz, const float *new_xyz, const float *features_nlc,
                                      int32_t *idx_out, float *out_nlc, const void *sorted, ws3d_stream_t stream);
WS3D_API int ws3d_three_interpolate_nlc(int b, int c, int m, int n, const float *feats_nlc, const int32_t *idx,
                                        const float *weight, float *out_nlc, int out_stride, ws3d_stream_t stream);
WS3D_API int ws3d_rowmax_rows(long rows_out, int ns, int o_ch, const float *y, float *out, int out_stride,
                              ws3d_stream_t stream);

/* First SA level fused: x_rows4 (rows, 4) grouped columns [dx,dy,dz,f] -> three pointwise layers
 * (W^T row-major (in, out), BN folded; ReLU after layers 1, 2 and, if relu3, 3) -> max over each
 * group of nsample consecutive rows -> out (rows/nsample, c3) with row stride out_stride.  Widths
 * (c1,c2,c3) x nsample in {(16,16,32),(32,32,64)} x {16,32}; WS3D_E_UNSUPPORTED otherwise (the
 * caller then runs the GEMM chain).  ws3d extension, used by ws3d_amd/fastpath.py.               */
WS3D_API int ws3d_sa_mlp3_pool(long rows, int nsample, int c1, int c2, int c3, const float *x_rows4,
                               const float *w1t, const float *b1, const float *w2t, const float *b2,
                               const float *w3t, const float *b3, int relu3, float *out, int out_stride,
                               ws3d_stream_t stream);

/* Two pointwise layers on rows in one kernel, 128 -> 128 -> o2 (o2 <= 64; the RPN's classification and regression heads):
 * out (rows, o2) = relu2?(relu1?(x @ w1t + b1) @ w2t + b2), w1t (128, 128) and w2t (128, o2) row-major, biases may be NULL.
 * The (rows, 128) activation between the layers stays in registers (fp32 matrix cores).  rows % 32 == 0; other shapes
 * return WS3D_E_UNSUPPORTED (the caller runs two GEMMs).  ticket: one device int, ZERO on entry (consumed: it counts the
 * 32-row tiles handed out), or NULL for a static split of the rows over the workgroups -- slower when other streams hold
 * compute units.  ws3d extension, used by ws3d_amd/fastpath.py.                                                        */
WS3D_API int ws3d_mlp2_rows(long rows, int k_dim, int o1, int o2, const float *x_rows, const float *w1t, const float *b1, int relu1,
                   const float *w2t, const float *b2, int relu2, float *out, int *ticket, ws3d_stream_t stream);

/* Both layers of a two-layer feature-propagation module in one kernel (round 4): the first layer's rows are built in the A operand of the
 * second layer's product exactly as ws3d_qinterp_rows builds them,
 *   x = relu1?( w0 Q[i0] + w1 Q[i1] + w2 Q[i2] + (lin (b*n, c)  |  skip (b*n, c1 <= 4) @ wb (c1, c) + b1) ),   out (b*n, o) = relu2?( x @ w2t (c, o) + b2 )
 * q (b, m, c) = known_feats @ W_a; idx / weight (b, n, 3).  b*n % 64, c % 16, o % 128; otherwise WS3D_E_UNSUPPORTED (the caller runs
 * ws3d_qinterp_rows + a GEMM).  ws3d extension, used by ws3d_amd/fastpath.py.                                                          */
WS3D_API int ws3d_qinterp_gemm(int b, int n, int m, int c, int o_dim, const float *q, const int32_t *idx, const float *weight, const float *lin,
                      const float *skip, int c1, const float *wb, const float *b1, int relu1, const float *w2t, const float *b2, int relu2,
                      float *out, ws3d_stream_t stream);

/* -------------------------------------------------------------------- iou3d_cuda */

/* boxes_overlap_bev_gpu(boxes_a,boxes_b,ans)   iou3d.cpp:31-50 -> iou3d_kernel.cu:
 * 108-234,354-363.  boxes (N,5) [x1,y1,x2,y2,ry] -> ans (num_a,num_b) overlap area. */
WS3D_API int ws3d_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                           float *ans, ws3d_stream_t stream);

/* boxes_iou_bev_gpu   iou3d.cpp:52-71 -> iou3d_kernel.cu:214-248,365-371.           */
WS3D_API int ws3d_boxes_iou_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                       float *ans, ws3d_stream_t stream);

/* The K12/K13 mask alone (iou3d_kernel.cu:250-292 / 306-348): mask (n, ceil(n/64))
 * uint64, bit t of word c of row i = iou(box_i, box_{64c+t}) > thresh (diagonal
 * block: only t > i%64).  Words of blocks c < i/64 (never read by the sweep,
 * iou3d.cpp:108) are written as 0 unless full_grid != 0, in which case they are
 * computed like the reference grid does.  normal != 0 selects the axis-aligned IoU.  */
WS3D_API int ws3d_nms_mask(int boxes_num, const float *boxes, float thresh, int normal, int full_grid,
                  uint64_t *mask, ws3d_stream_t stream);

/* nms_gpu / nms_normal_gpu(boxes, keep, thresh) -> num_to_keep   iou3d.cpp:73-170.
 * boxes (n,5) already score-sorted (iou3d_utils.py:67-69).  The reference copies the
 * mask to the host and sweeps there; here mask kernel + greedy sweep both run on the
 * device: keep (n) int64 DEVICE, num_keep (1) int32 DEVICE -- no D2H, no sync.
 * max_keep > 0 stops the sweep after that many survivors (the first max_keep entries of
 * the reference's keep list; its callers slice [:RPN_POST_NMS_TOP_N]); <= 0 = all.
 * workspace: ws3d_nms_workspace_bytes(n) bytes of device memory.                     */
WS3D_API size_t ws3d_nms_workspace_bytes(int boxes_num);
WS3D_API int ws3d_nms(int boxes_num, const float *boxes, float thresh, int normal, int max_keep,
                      void *workspace, size_t workspace_bytes, int64_t *keep, int32_t *num_keep,
                      ws3d_stream_t stream);

/* The same for a batch of scenes in ONE launch pair: boxes (batch,n,5), keep (batch,n),
 * num_keep (batch); workspace >= batch * ws3d_nms_workspace_bytes(n).                      */
WS3D_API int ws3d_nms_batched(int batch, int boxes_num, const float *boxes, float thresh, int normal,
                              int max_keep, void *workspace, size_t workspace_bytes, int64_t *keep,
                              int32_t *num_keep, ws3d_stream_t stream);

/* ------------------------------------------------ Stage-1 proposal stage */

/* Greedy radius NMS of centre proposals, replacing the reference's Python loops
 * (generate_box_dataset.py:127-140, tools/eval_auto.py:270-284: one host sync per candidate).
 * centers (batch,n,2) = (x,z) per scene, sorted by descending score; candidate i is kept iff
 * distance_2 (lib/utils/distance.py:3, fp32) to every kept centre is > radius.  Outputs and
 * workspace as ws3d_nms_batched (workspace >= batch * ws3d_nms_workspace_bytes(n)).          */
WS3D_API int ws3d_radius_nms_batched(int batch, int n, const float *centers, float radius, int max_keep,
                                     void *workspace, size_t workspace_bytes, int64_t *keep,
                                     int32_t *num_keep, ws3d_stream_t stream);

/* Sorted top-k per scene (ws3d extension): scores (b, n), n <= 16384 -> the k largest in
 * descending order, out_scores (b, k) and out_idx (b, k) int64; equal scores in ascending index
 * order, NaN first (torch.topk's convention).                                                       */
WS3D_API int ws3d_topk_sorted(int b, int n, int k, const float *scores, float *out_scores, int64_t *out_idx,
                              ws3d_stream_t stream);
/* The same result with the sort of a scene spread over several workgroups (2048-key segments sorted side by side, then ranked
 * against each other): workspace >= ws3d_topk_workspace_bytes(b, n) bytes of device memory, 16-byte aligned; with a NULL / short
 * workspace, or n <= 2048 (ws3d_topk_workspace_bytes returns 0), it IS ws3d_topk_sorted.  At 16384 scores x 8 scenes: 107 -> ~25 us. */
WS3D_API size_t ws3d_topk_workspace_bytes(int b, int n);
WS3D_API int ws3d_topk_sorted_ws(int b, int n, int k, const float *scores, float *out_scores, int64_t *out_idx, void *workspace,
                                 size_t workspace_bytes, ws3d_stream_t stream);
/* The same two sorts over sigmoid(logits[i]) = 1 / (1 + exp(-x)) in fp32, torch.sigmoid's expression, evaluated while the keys are
 * loaded (round 5): the proposal stage then needs no score tensor and no sigmoid launch; out_scores holds the sigmoid values. */
WS3D_API int ws3d_topk_sorted_sigmoid(int b, int n, int k, const float *logits, float *out_scores, int64_t *out_idx, ws3d_stream_t stream);
WS3D_API int ws3d_topk_sorted_sigmoid_ws(int b, int n, int k, const float *logits, float *out_scores, int64_t *out_idx, void *workspace,
                                         size_t workspace_bytes, ws3d_stream_t stream);

/* Proposal decode (ws3d extension): xyz (b,n,3), rpn_reg (b,n,4*bins) -> boxes (b,n,7)
 * = [x + dx, y + h/2, z + dz, h, w, l, ry] with (dx, dz) = decode_center_target
 * (lib/utils/bbox_transform.py:24-61) and ry the per-index synthetic heading of
 * ws3d_amd.stage1.synthetic_orientation; bit-identical to the torch composition.          */
WS3D_API int ws3d_decode_center_boxes(int b, int n, int bins, float loc_scope, float loc_bin_size, float h, float w,
                                      float l, const float *xyz, const float *rpn_reg, float *boxes,
                                      ws3d_stream_t stream);

/* Glue of the on-device proposal stage (no reference counterpart: the reference does these steps in Python around its ops,
 * kitti_utils.py:134-160, roipool3d_utils.py:19, generate_box_dataset.py:92-140).
 * ws3d_gather_boxes_bev: box (b,n,7), order (b,top) int64 -> box_sorted (b,top,7) = box[order] and bev (b,top,5) =
 * [x - l/2, z - w/2, x + l/2, z + w/2, ry] (boxes3d_to_bev).                                                        */
WS3D_API int ws3d_gather_boxes_bev(int b, int n, int top, const float *box, const int64_t *order, float *box_sorted, float *bev,
                          ws3d_stream_t stream);
/* ws3d_decode_center_boxes + ws3d_gather_boxes_bev in one launch (round 5): only the `top` points that order (b, top) names are decoded
 * (same arithmetic, bit-identical rows), written in that order with their BEV rectangles; no (b, n, 7) tensor of every point's box. */
WS3D_API int ws3d_decode_gather_boxes_bev(int b, int n, int top, int bins, float loc_scope, float loc_bin_size, float h, float w, float l,
                                          const float *xyz, const float *rpn_reg, const int64_t *order, float *box_sorted, float *bev,
                                          ws3d_stream_t stream);
/* ws3d_select_proposals: keep (b,keep_stride) int64 / num (b) int32 from ws3d_nms_batched on score-sorted boxes -> the
 * first min(num, k) survivors as boxes_out (b,k,7) and scores_out (b,k), zero-padded (row * 0 for the padding, as the
 * torch composition does), count (b) int64, and (optional) pooled_boxes (b,k,7) = enlarge_box3d(boxes_out, extra_width). */
WS3D_API int ws3d_select_proposals(int b, int top, int keep_stride, int k, const float *box_sorted, const float *scores_sorted,
                          const int64_t *keep, const int32_t *num, float extra_width, float *boxes_out, float *scores_out,
                          int64_t *count, float *pooled_boxes, ws3d_stream_t stream);
/* ... and additionally packed (b, k, 8) rows = box + score (or NULL): what the multi-GPU gather sends (ws3d_amd/dist.py), written here
 * instead of by a concatenation launch (round 5). */
WS3D_API int ws3d_select_proposals_packed(int b, int top, int keep_stride, int k, const float *box_sorted, const float *scores_sorted,
                                          const int64_t *keep, const int32_t *num, float extra_width, float *boxes_out, float *scores_out,
                                          int64_t *count, float *pooled_boxes, float *packed, ws3d_stream_t stream);
/* ... the packed rows written straight into the multi-GPU exchange's send buffer (round 6, ABI 6): scene i's k rows of 8 floats at
 * send + i * send_stride, the scene's count as the float behind them (send_stride >= k * 8 + 1 floats; counts < 2^24 are exact), so
 * that the step's one all-gather (ws3d_amd/dist.py ProposalExchange; replaces nn.DataParallel's gather, tools/train_rpn.py:175-176)
 * sends a buffer no other launch packs. */
WS3D_API int ws3d_select_proposals_send(int b, int top, int keep_stride, int k, const float *box_sorted, const float *scores_sorted,
                                        const int64_t *keep, const int32_t *num, float extra_width, float *boxes_out, float *scores_out,
                                        int64_t *count, float *pooled_boxes, float *send, long send_stride, ws3d_stream_t stream);
/* The step's prologue in one launch (round 5): the (rows, c) input rows, c >= 3, split into xyz (rows, 3) and feats (rows, c - 3), and
 * clear_bytes bytes at `clear` zeroed (16-byte aligned and a multiple of 16; 0: nothing) -- the pass's zero-initialised scratch. */
WS3D_API int ws3d_split_points_clear(long rows, int c, const float *pc, float *xyz, float *feats, void *clear, size_t clear_bytes,
                                     ws3d_stream_t stream);

/* ---------------------------------------------------------------- roipool3d_cuda */

/* forward(xyz,boxes3d,pts_feature,pooled_features,pooled_empty_flag)
 * roipool3d.cpp:48-79 -> roipool3dLauncher roipool3d_kernel.cu:209-237 (K14-K16);
 * forward_slow (roipool3d.cpp:15-44) has identical results.
 * xyz (B,N,3), boxes3d (B,M,7) [x,y_bottom,z,h,w,l,ry] (already enlarged by the
 * caller, roipool3d_utils.py:19), pts_feature (B,N,C) -> pooled_features
 * (B,M,S,3+C) and pooled_empty_flag (B,M) int32, BOTH pre-zeroed by the caller
 * (roipool3d_utils.py:21-23): rows of empty boxes are left untouched.
 * pts_idx (B,M,S) int32 may be NULL; when given it receives the selected point
 * indices (the reference's internal pts_idx scratch; zeros for empty boxes).
 * No B*N*M scratch, no allocation: selection and copy are fused per box.             */
WS3D_API int ws3d_roipool3d(int batch_size, int pts_num, int boxes_num, int feature_in_len,
                   int sampled_pts_num, const float *xyz, const float *boxes3d,
                   const float *pts_feature, float *pooled_features, int32_t *pooled_empty_flag,
                   int32_t *pts_idx, ws3d_stream_t stream);
/* Same result, but EVERY element of pooled_features / pooled_empty_flag (/ pts_idx) is written: no
 * pre-zeroing by the caller (the wrapper's zero-fill is a 214 MB memset per 8 scenes at the C3 shapes;
 * here only the rows of empty boxes are zeroed, by the workgroup that found the box empty).        */
WS3D_API int ws3d_roipool3d_fill(int batch_size, int pts_num, int boxes_num, int feature_in_len,
                   int sampled_pts_num, const float *xyz, const float *boxes3d,
                   const float *pts_feature, float *pooled_features, int32_t *pooled_empty_flag,
                   int32_t *pts_idx, ws3d_stream_t stream);

/* ws3d_roipool3d / ws3d_roipool3d_fill (fill != 0) with a caller-provided scratch buffer: for scenes of
 * ws3d_roipool3d_workspace_bytes(batch, pts_num) > 0 bytes (16384 <= pts_num <= 262144) the scene is counting-sorted once into an
 * (x, z) grid inside `workspace` and every box tests only the cells under its footprint (roipool3d.hip: roi_bin_kernel +
 * roipool3d_binned_kernel; the first-S-by-point-index rule of roipool3d_kernel.cu:97-160 is kept by a radix select on the index).
 * Same outputs as ws3d_roipool3d bit for bit; workspace NULL / too small / not applicable: the scanning kernels run.
 * 16-byte aligned, contents undefined on return; nothing is allocated or synchronised.  ws3d extension (round 5, ABI 5). */
WS3D_API size_t ws3d_roipool3d_workspace_bytes(int batch_size, int pts_num);
WS3D_API int ws3d_roipool3d_ws(int batch_size, int pts_num, int boxes_num, int feature_in_len, int sampled_pts_num, const float *xyz,
                               const float *boxes3d, const float *pts_feature, float *pooled_features, int32_t *pooled_empty_flag,
                               int32_t *pts_idx, int fill, void *workspace, size_t workspace_bytes, ws3d_stream_t stream);


/* Device twin of pts_in_boxes3d_cpu (roipool3d.cpp:97-124): pts (N,3), boxes3d (M,7)
 * -> flag (M,N) int64 in {0,1}.                                                      */
WS3D_API int ws3d_pts_in_boxes3d(int boxes_num, int pts_num, const float *pts, const float *boxes3d,
                        int64_t *flag, ws3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WS3D_OPS_H */
