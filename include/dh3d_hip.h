/*
 * dh3d_hip.h -- C ABI of libdh3d_hip.so, the MI355X (gfx950) implementation of the DH3D
 * point-cloud feature-extraction hot path.
 *
 * Contract (every entry point):
 *   - plain device pointers + sizes, no framework types; `stream` is a hipStream_t passed as void*;
 *   - returns an int status (DH3D_OK = 0); never throws, never allocates or frees device memory,
 *     never synchronises the device; all work is enqueued on `stream` (graph-capturable);
 *   - the caller owns every buffer including scratch; gradient outputs of the reference's operators are zeroed by
 *     the library (hipMemsetAsync on `stream`), as the reference kernels do with cudaMemset.  The ACCUMULATORS of the
 *     training step's internal kernels (statistics partials and scatter targets of dh3d_bn_colstats / dh3d_bn_bwd_sums /
 *     dh3d_interp_bn_* / dh3d_netvlad_commuted_*, marked "zeroed by the CALLER" below) are not: a step takes all of them
 *     from one arena that it clears with ONE fill (each hipMemsetAsync is a ~4 us launch of its own; there were ~30);
 *   - float32 + int32 (the six flex operators of section A also for double: *_f64); re-entrant, no global mutable
 *     state.
 *
 * Section A are drop-ins for the reference's TF custom ops, in the reference's tensor layouts
 * (user_ops: channels-first [B,C,N]; tf_ops: channels-last [B,N,C]).  Section B are the fused
 * point-major ([B,N,C]) kernels the dh3d_amd model path is built from; they compute the same
 * functions (see each comment) with layouts chosen for gfx950.
 *
 * Reference citations are file:line in the upstream DH3D tree.
 */
#ifndef DH3D_HIP_H_
#define DH3D_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DH3D_OK 0
#define DH3D_ERR_INVALID_ARGUMENT 1 /* what TF reports as errors::InvalidArgument */
#define DH3D_ERR_UNSUPPORTED 2      /* shape outside what the kernels implement    */
#define DH3D_ERR_LAUNCH 3           /* hipGetLastError() != success after a launch  */

#define DH3D_ACT_NONE 0
#define DH3D_ACT_RELU 1
#define DH3D_ACT_SIGMOID 2

/* Library / build identification. */
int dh3d_version(void);                 /* 100*major + minor */
/* Bumped whenever the MEANING of an existing entry point changes under an unchanged signature (which a linker cannot
 * see).  History: 1 = rounds 1-2; 2 = the accumulators of dh3d_bn_colstats / dh3d_bn_bwd_sums / dh3d_interp_bn_colstats /
 * dh3d_interp_bn_bwd_sums / dh3d_interp_bn_bwd_apply are no longer zeroed by the library ("zeroed by the CALLER");
 * 3 (round 4) = dh3d_knn_grid takes the three outputs of dh3d_spatial_sort_cells (sorted, gbox, cells), serves any
 * N <= 16384 and hands clouds the sort flags as crowded (cells[4106]) to the pruned scan; 4 (round 5) = the cell table
 * describes a grid whose 12 code bits are dealt to the axes by extent (cells[4103..4105] = 2^(nb+2)/extent, cells[4107] =
 * the bit schedule; see dh3d_spatial_sort_cells) and dh3d_knn_grid reads that header.
 * A binding compares it with the DH3D_ABI_VERSION it was written against and refuses to run on a mismatch
 * (dh3d_amd/_lib.py does). */
#define DH3D_ABI_VERSION 4
int dh3d_abi_version(void);
const char *dh3d_arch(void);            /* "gfx950" */
/* First 16 hex digits of sha256 over (basename, content) of dh3d_amd/csrc/{*.hip, *.h, Makefile} and this header, sorted by
 * path as the Makefile's $(sort ...) orders them -- baked in at build time so that a binding (dh3d_amd/_lib.py
 * tree_source_hash, tests/test_abi.py) can tell a library built from THIS tree from a stale one that travelled with it. */
const char *dh3d_source_hash(void);
const char *dh3d_status_string(int st); /* static string */

/* Staging copy on the given stream by a KERNEL (not a copy engine): `src` / `dst` are device pointers or PINNED host
 * pointers (device-addressable: hipHostMalloc / torch pin_memory), 16-byte aligned.  For the serving loop around the hot
 * path (the reference feeds numpy arrays and saves numpy arrays: localdesc_extract.py:106-138, globaldesc_extract.py:84-100):
 * the batch goes host -> slot buffer and the descriptors slot buffer -> host on the SAME compute queue as the step, so no
 * engine-to-engine hand-over sits on the step's chain.  device_to_device != 0: both are device buffers (the launch is
 * sized for HBM instead of for the host link).  No reference counterpart (TF's feed_dict / fetches). */
int dh3d_stage_copy(const void *src, void *dst, size_t bytes, int device_to_device, void *stream);

/* ===================================================================================== *
 * A. Drop-in operators (reference layouts)
 * ===================================================================================== */

/* KnnBruteforce -- replaces KnnBruteforceFunctor<GPUDevice,float,int>
 * (user_ops/kernels/knn_bruteforce_kernel_gpu.cu.cc:163-228; op user_ops/ops/knn_bruteforce.cc:11-35).
 * positions [B,Dp,N] -> nn [B,N,K] int32, dist [B,N,K] (Euclidean, ascending; self at rank 0).
 * Bit-exact ids incl. the CUB tie order.  Dp must be 3, 1 <= K <= 64.  N > 8192 (unsupported
 * upstream) is accepted with the ladder continued as (1024, ceil(N/1024)). */
int dh3d_knn_bruteforce(const float *positions, int B, int Dp, int N, int K, int32_t *nn,
                        float *dist, void *stream);

/* Same function on point-major coordinates xyz [B,N,3] (the layout the model holds clouds in,
 * core/model.py:146); saves the transpose of core/model.py:157. */
int dh3d_knn_bruteforce_xyz(const float *xyz, int B, int N, int K, int32_t *nn, float *dist,
                            void *stream);

/* FlexConv -- replaces FlexConvFunctor<GPUDevice,float> (flex_conv_kernel_gpu.cu.cc:392-441;
 * op user_ops/ops/flex_conv.cc:25-83).  Argument order is the functor's:
 * features [B,Din,N], theta [Dp,Din,Dout], bias [Din,Dout], neighborhood [B,K,N] int32,
 * positions [B,Dp,N] -> output [B,Dout,N].  Centre point = point n (gpu.cu.cc:77-79). */
int dh3d_flex_conv_fwd(const float *features, const float *theta, const float *bias,
                       const int32_t *neighborhood, const float *positions, int B, int N, int K,
                       int Dp, int Din, int Dout, float *output, void *stream);

/* FlexConvGrad -- replaces FlexConvGrad<GPUDevice,float> (flex_conv_kernel_gpu.cu.cc:446-548).
 * Centre point = rank-0 neighbour (gpu.cu.cc:196-202,314).  Zeroes the three outputs first. */
int dh3d_flex_conv_bwd(const float *features, const float *theta, const float *bias,
                       const int32_t *neighborhood, const float *positions, const float *topdiff,
                       int B, int N, int K, int Dp, int Din, int Dout, float *grad_features,
                       float *grad_theta, float *grad_bias, void *stream);

/* FlexPool -- replaces FlexPoolFunctor<GPUDevice,float> (flex_pool_kernel_gpu.cu.cc:100-125).
 * features [B,D,N], neighborhood [B,K,N] -> output [B,D,N], argmax [B,D,N] (global point id). */
int dh3d_flex_pool_fwd(const float *features, const int32_t *neighborhood, int B, int N, int K,
                       int D, float *output, int32_t *argmax, void *stream);

/* FlexPoolGrad -- replaces FlexPoolGrad<GPUDevice,float> (flex_pool_kernel_gpu.cu.cc:131-157). */
int dh3d_flex_pool_bwd(const float *topdiff, const int32_t *argmax, int B, int N, int D,
                       float *grad_features, void *stream);

/* ConvPointset -- replaces ConvPointsetFunctor<GPUDevice,float>
 * (conv_pointset_kernel_gpu.cu.cc:354-402).  features [B,Din,N], theta [Din,Dout], bias [Dout],
 * neighborhood [B,K,N] -> output [B,Dout,N]; bias added once (the CPU functor's rule,
 * conv_pointset_kernel.cc:60). */
int dh3d_conv_pointset_fwd(const float *features, const float *theta, const float *bias,
                           const int32_t *neighborhood, int B, int N, int K, int Din, int Dout,
                           float *output, void *stream);

/* ConvPointsetGrad -- replaces ConvPointsetGrad<GPUDevice,float> (gpu.cu.cc:408-503). */
int dh3d_conv_pointset_bwd(const float *features, const float *theta,
                           const int32_t *neighborhood, const float *topdiff, int B, int N, int K,
                           int Din, int Dout, float *grad_features, float *grad_theta,
                           float *grad_bias, void *stream);

/* The same six operators for double (the reference registers float AND double kernels, flex_conv_op.cc:97-106 /
 * flex_pool_op.cc / conv_pointset_op.cc, and its gradient tests run in double, test_flex_convolution.py:93-115): the
 * reference formulation of csrc/flex_generic.hip instantiated for double, any shape. */
int dh3d_flex_conv_fwd_f64(const double *features, const double *theta, const double *bias, const int32_t *neighborhood,
                           const double *positions, int B, int N, int K, int Dp, int Din, int Dout, double *output,
                           void *stream);
int dh3d_flex_conv_bwd_f64(const double *features, const double *theta, const double *bias, const int32_t *neighborhood,
                           const double *positions, const double *topdiff, int B, int N, int K, int Dp, int Din, int Dout,
                           double *grad_features, double *grad_theta, double *grad_bias, void *stream);
int dh3d_flex_pool_fwd_f64(const double *features, const int32_t *neighborhood, int B, int N, int K, int D, double *output,
                           int32_t *argmax, void *stream);
int dh3d_flex_pool_bwd_f64(const double *topdiff, const int32_t *argmax, int B, int N, int D, double *grad_features,
                           void *stream);
int dh3d_conv_pointset_fwd_f64(const double *features, const double *theta, const double *bias,
                               const int32_t *neighborhood, int B, int N, int K, int Din, int Dout, double *output,
                               void *stream);
int dh3d_conv_pointset_bwd_f64(const double *features, const double *theta, const int32_t *neighborhood,
                               const double *topdiff, int B, int N, int K, int Din, int Dout, double *grad_features,
                               double *grad_theta, double *grad_bias, void *stream);

/* FarthestPointSample -- replaces farthestpointsamplingLauncher (tf_ops/sampling/tf_sampling.cpp:94,
 * tf_sampling_g.cu:105-170,203-205).  inp [B,N,3] -> out [B,m] int32, first pick 0, bit-exact tie
 * order.  `temp` is the reference's scratch (tf_sampling.cpp:115 allocates [32,N]): for N <= 16384 the
 * running min-distances live in registers and it may be NULL; above that it must hold B*N floats. */
int dh3d_farthest_point_sample(int B, int N, int m, const float *inp, float *temp, int32_t *out,
                               void *stream);

/* Same op with the rounding of d = (x2-x1)^2+(y2-y1)^2+(z2-z1)^2 (tf_sampling_g.cu:141) selectable:
 * contract = 1: fma(dz,dz,fma(dx,dx,dy*dy)) -- the LLVM/NVVM contraction, what the kernels above compute;
 * contract = 0: (dx*dx+dy*dy)+dz*dz -- an nvcc -fmad=false build.  Which one a given reference binary
 * used cannot be checked without nvcc (parity unpinned); an integrator who can check picks here.
 * Any N; temp [B*N] floats is REQUIRED (the running min-distances, as upstream). */
int dh3d_farthest_point_sample_mode(int B, int N, int m, const float *inp, float *temp, int32_t *out,
                                    int contract, void *stream);

/* GroupPoint / GroupPointGrad -- replace groupPointLauncher / groupPointGradLauncher
 * (tf_ops/grouping/tf_grouping.cpp:208-274, tf_grouping_g.cu:94-132).
 * points [b,n,c], idx [b,m,nsample] -> out [b,m,nsample,c]. */
int dh3d_group_point_fwd(int b, int n, int c, int m, int nsample, const float *points,
                         const int32_t *idx, float *out, void *stream);
int dh3d_group_point_bwd(int b, int n, int c, int m, int nsample, const float *grad_out,
                         const int32_t *idx, float *grad_points, void *stream);

/* ThreeNN -- replaces threenn_cpu (tf_ops/interpolation/tf_interpolate.cpp:60-103).
 * xyz1 [b,n,3], xyz2 [b,m,3] -> dist [b,n,3] (SQUARED, ascending), idx [b,n,3]. */
int dh3d_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist,
                  int32_t *idx, void *stream);

/* ThreeInterpolate / Grad -- replace threeinterpolate_cpu / threeinterpolate_grad_cpu
 * (tf_interpolate.cpp:107-153).  points [b,m,c], idx/weight [b,n,3] -> out [b,n,c]. */
int dh3d_three_interpolate_fwd(int b, int m, int c, int n, const float *points,
                               const int32_t *idx, const float *weight, float *out, void *stream);
int dh3d_three_interpolate_bwd(int b, int n, int c, int m, const float *grad_out,
                               const int32_t *idx, const float *weight, float *grad_points,
                               void *stream);

/* ---- A'. The same operators on the fast kernels (reference layouts, caller-provided workspace) ----
 * FlexConv / FlexConvGrad / FlexPool with the signatures above + (workspace, workspace_bytes): the channels-first
 * tensors are transposed through LDS tiles into the workspace, the fused point-major MFMA kernels of section B run
 * (flex_conv: the bf16x6 pipeline for K = 8, Dout = 64, Din in {32, 64}; the exact-f32 MFMA kernel for the other
 * DH3D shapes; for ANY other Din, Dout that are multiples of four the same factorisation in two launches -- S =
 * [S0|Sx|Sy|Sz] materialised in the workspace, then S @ [bias; theta] on the GEMM kernels: 0.07 ms at 48 -> 96,
 * 8 x 8192, K = 8, against ~3 ms of the reference formulation), and the result is transposed back.  Same function as the section-A entry (forward centres on
 * point n, backward on the rank-0 neighbour); summation order differs (factorised form), within the reference's own
 * CPU-vs-GPU tolerance.  *_workspace_bytes returns 0 when the shape is not served -- use the section-A entry then.
 * The backward is  dWcat = S^T dOut,  dS = dOut Wcat^T  on the f32 MFMA pipe + an atomics scatter of dS over the
 * neighbour lists (the reference's feature gradient is atomics too, flex_conv_kernel_gpu.cu.cc:250-385). */
size_t dh3d_flex_conv_fwd_workspace_bytes(int B, int N, int K, int Dp, int Din, int Dout);
int dh3d_flex_conv_fwd_ws(const float *features, const float *theta, const float *bias,
                          const int32_t *neighborhood, const float *positions, int B, int N, int K, int Dp,
                          int Din, int Dout, float *output, void *workspace, size_t workspace_bytes, void *stream);
size_t dh3d_flex_conv_bwd_workspace_bytes(int B, int N, int K, int Dp, int Din, int Dout);
int dh3d_flex_conv_bwd_ws(const float *features, const float *theta, const float *bias,
                          const int32_t *neighborhood, const float *positions, const float *topdiff, int B, int N,
                          int K, int Dp, int Din, int Dout, float *grad_features, float *grad_theta,
                          float *grad_bias, void *workspace, size_t workspace_bytes, void *stream);
size_t dh3d_flex_pool_fwd_workspace_bytes(int B, int N, int K, int D);
int dh3d_flex_pool_fwd_ws(const float *features, const int32_t *neighborhood, int B, int N, int K, int D,
                          float *output, int32_t *argmax, void *workspace, size_t workspace_bytes, void *stream);

/* ===================================================================================== *
 * B. Fused point-major kernels (model path)
 *    Activations are [B,N,C] (C contiguous).  Neighbourhoods are [B,N,K] (the kNN op's native
 *    output, knn_bruteforce_op.cc:44-49).  Every kernel can apply the per-channel epilogue
 *        y = act( scale[c] * (x + pre_bias[c]) + shift[c] )
 *    which folds feature_bias (core/layers.py:330-331), inference BatchNorm
 *    (core/tf_utils.py:60-63) and the activation; scale/shift/pre_bias may be NULL.
 * ===================================================================================== */

typedef struct dh3d_epilogue {
  const float *pre_bias; /* [C] or NULL */
  const float *scale;    /* [C] or NULL */
  const float *shift;    /* [C] or NULL */
  int act;               /* DH3D_ACT_* */
} dh3d_epilogue;

/* Packs a row-major weight W [Kd, Dout] (Kd%8==0, Dout%32==0) into the MFMA fragment order the
 * GEMM kernels stream: out[(nb*KB+kb)*256 + lane*4 + t] = W[kb*8 + 4*(lane>>5) + t][nb*32 + (lane&31)].
 * For flex_conv, W is the stacked [bias; theta_x; theta_y; theta_z] of shape [4*Din, Dout]. */
int dh3d_pack_weight(const float *W, int Kd, int Dout, float *packed, void *stream);
/* Stacks flex_conv parameters theta [3,Din,Dout], bias [Din,Dout] into packed [4*Din,Dout]. */
int dh3d_pack_flex_weight(const float *theta, const float *bias, int Din, int Dout, float *packed,
                          void *stream);

/* Spatial (Morton) ordering of each cloud -- a preprocessing step with no reference counterpart that
 * changes no result; it lets kNN and FPS skip work exactly (csrc/spatial.hip).
 * xyz [B,N,3] -> sorted [B,N,4] records (x, y, z, bits(original index)) in Morton order, and
 * gbox [B, ceil(N/64), 8] = (min xyz, 0, max xyz, 0) of every 64 consecutive records.  N <= 16384. */
int dh3d_spatial_sort(const float *xyz, int B, int N, float *sorted, float *gbox, void *stream);

/* KnnBruteforce on an ordered cloud: identical outputs to dh3d_knn_bruteforce_xyz (ids are ORIGINAL point
 * indices, rows are in original query order); candidate groups whose box is provably too far are skipped. */
int dh3d_knn_sorted(const float *sorted, const float *gbox, int B, int N, int K, int32_t *nn, float *dist,
                    void *stream);

/* dh3d_spatial_sort that also writes the CELL TABLE of a 4096-cell grid over each cloud's bounding box: cells
 * [B, DH3D_CELL_INTS] int32.  The sort key is an 18-bit code; its top 12 bits are the GRID code, and since ABI 4 those 12
 * bits are dealt to the axes BY EXTENT -- one at a time to the axis whose cells are the widest so far (at most 6 per axis;
 * cell widths within 25 % of each other: z first, then y, then x), so axis a gets nb[a] grid bits (a cube: 4 + 4 + 4 = the
 * plain z-y-x Morton code of ABI <= 3; a 60 x 60 x 8 slab: 5 + 5 + 2) and the grid is 2^nb[0] x 2^nb[1] x 2^nb[2] cells.
 *   [0..4095]     first sorted position of every cell, cells ordered by their 12-bit code (an empty cell's entry = the
 *                 next cell's); [4096] = N
 *   [4100..4102]  as floats: the grid origin (x, y, z) = the bounding box's minimum
 *   [4103..4105]  as floats: the quantisation scale per axis, 2^(nb[a] + 2) / extent[a] (a cell = 4 quantisation steps; the
 *                 two sub-cell bits per axis are the key's low 6 bits, z y x z y x)
 *   [4106]        1 when the cloud's points crowd into fewer than 0.6 x the cells a uniform cloud of N points would occupy
 *                 (street scenes, clusters: dh3d_knn_grid then serves this cloud with the pruned scan of
 *                 dh3d_knn_sorted), else 0
 *   [4107]        the bit SCHEDULE: 12 x 2 bits, field s (bits 2s, 2s+1) = the axis (0 x, 1 y, 2 z) that grid-code bit
 *                 11 - s belongs to -- field 0 is the code's MOST significant bit; within an axis the bits appear in the
 *                 code from that axis' most significant cell-coordinate bit downwards.  0x186186 = z y x z y x ... (cube).
 * To decode cell code c into per-axis cell coordinates: walk s = 0..11, append bit (c >> (11 - s)) & 1 to the coordinate of
 * axis (sched >> 2s) & 3.  A table written by an ABI <= 3 sort (no schedule, scale 64 / extent) must not be handed to an
 * ABI 4 dh3d_knn_grid. */
#define DH3D_CELL_INTS 4112
int dh3d_spatial_sort_cells(const float *xyz, int B, int N, float *sorted, float *gbox, int32_t *cells, void *stream);

/* KnnBruteforce on that grid (cell-list search: every query scans the cells its K-th distance reaches, candidates pooled
 * per query in LDS, 4 lanes per query; csrc/knn.hip knn_grid_kernel).  Same outputs as dh3d_knn_bruteforce_xyz bit for
 * bit -- ids in the reference's (distance, CUB rank) order, IEEE distances, original point order.  K <= 8, any
 * N <= 16384 (small sets search a coarser grid: 2^D consecutive cells of the same table).  sorted / gbox / cells are the
 * three outputs of dh3d_spatial_sort_cells: a query the two cell passes cannot serve (no K-th distance after 27
 * cells, a search ball wider than two cells in x: sparse corners, outliers) restarts over the 64-point groups whose box
 * (gbox) its search ball meets; a cloud the sort flagged as crowded (cells[4106]) is served by dh3d_knn_sorted's
 * kernels instead, from the same launch sequence (same results either way). */
int dh3d_knn_grid(const float *sorted, const float *gbox, const int32_t *cells, int B, int N, int K, int32_t *nn, float *dist,
                  void *stream);


/* ThreeNN on ordered clouds: identical dist / idx to dh3d_three_nn (original indexing on both sides) from the
 * dh3d_spatial_sort outputs of the query cloud (sorted1 [b,n,4], gbox1) and of the candidate set (sorted2 [b,m,4],
 * gbox2; the boxes are not needed by the current kernel and may be NULL).  Every candidate is still visited; the
 * ordering makes the 3-deep insertion rare (a wave's queries are a compact region and its scan starts at the matching
 * place of the candidates' order). */
int dh3d_three_nn_sorted(int b, int n, int m, const float *sorted1, const float *gbox1, const float *sorted2,
                         const float *gbox2, float *dist, int32_t *idx, void *stream);

/* FarthestPointSample on an ordered cloud: identical outputs to dh3d_farthest_point_sample (original
 * indices); per round only the 64-point groups the new sample can affect are re-evaluated.  N <= 12288. */
int dh3d_fps_sorted(const float *sorted, const float *gbox, int B, int N, int m, int32_t *out, void *stream);
/* the same + xyz_out [B,m,3] = the sampled coordinates (group_point of the cloud by `out`, core/tf_utils.py:92-95) */
int dh3d_fps_sorted_xyz(const float *sorted, const float *gbox, int B, int N, int m, int32_t *out, float *xyz_out,
                        void *stream);
/* the same + the SAMPLED SET IN MORTON ORDER out of the same launch (round 6): sorted_s [B,m,4] records (x, y, z, bits(pick
 * rank)), gbox_s [B, ceil(m/64), 8], and -- when `cells`, the cloud's table from dh3d_spatial_sort_cells, is given (else both
 * NULL) -- cells_s [B, DH3D_CELL_INTS]: the subset's cell table on the CLOUD's grid (same origin / scales / schedule header,
 * its own crowded verdict).  The picks are a subset of a sorted cloud, so a stable compaction of the picked positions is their
 * spatial order: these are valid inputs for dh3d_three_nn_sorted (candidate side) and dh3d_knn_grid / dh3d_knn_sorted on
 * the sampled set, without dh3d_spatial_sort_cells(xyz_out) on the chain behind the sampling.  N <= 8192. */
int dh3d_fps_sorted_ordered(const float *sorted, const float *gbox, const int32_t *cells, int B, int N, int m, int32_t *out,
                            float *xyz_out, float *sorted_s, float *gbox_s, int32_t *cells_s, void *stream);
/* Same with the cloud itself (xyz [B,N,3], what dh3d_spatial_sort was given): clouds of up to 16384 points (above
 * 12288 the by-index coordinate table no longer fits the LDS and the kernel reads winners from xyz).  xyz_out may be NULL. */
int dh3d_fps_sorted_cloud(const float *sorted, const float *gbox, const float *xyz, int B, int N, int m, int32_t *out,
                          float *xyz_out, void *stream);

/* flex_conv forward (same function as dh3d_flex_conv_fwd, Dp = 3) in the factorised form
 *   out[n,:] = [S0 | Sx | Sy | Sz][n,:] @ [bias; theta_x; theta_y; theta_z],
 *   S0[n,i] = sum_k f[nk,i],  Sd[n,i] = sum_k (p[nk,d]-p[n,d]) f[nk,i]
 * as one kernel: K-neighbour gather-reduce into LDS, then an exact-f32 MFMA GEMM, then epilogue.
 * features [B,N,Din], xyz [B,N,3], nbr [B,N,K], wpacked from dh3d_pack_flex_weight.
 * Din in {32,64,128}, Dout % 32 == 0, Dout <= 256. */
int dh3d_flex_conv_pm_fwd(const float *features, const float *xyz, const int32_t *nbr,
                          const float *wpacked, int B, int N, int K, int Din, int Dout,
                          const dh3d_epilogue *ep, float *out, void *stream);

/* dh3d_flex_conv_pm_fwd with one more linear layer applied to the finished output tile before it leaves the chip:
 * out [B,N,Dout] as above AND out2 [B,N,Dpost] = out @ Wpost (wpost_packed = dh3d_pack_weight of [Dout, Dpost], no bias /
 * activation).  The global step uses it for NetVLAD's cluster logits on the sampled rows (core/backbones.py:213-216,
 * commuted through the up-sampling).  Din == 128, Dout == 256, K == 8, Dpost == 64. */
int dh3d_flex_conv_pm_post_fwd(const float *features, const float *xyz, const int32_t *nbr, const float *wpacked, int B,
                               int N, int K, int Din, int Dout, const dh3d_epilogue *ep, float *out,
                               const float *wpost_packed, int Dpost, float *out2, void *stream);
/* The same operator on 32-point tiles with the tile GEMM on the bf16 matrix pipe at f32 accuracy (six bf16 products per
 * f32 product, like dh3d_flex_conv_pm_x6_fwd; csrc/flex_tx6.hip): the sampled levels' layers, whose f32-MFMA GEMM phase
 * sits at the f32 pipe's floor for one tile per CU.  wpacked_x3 from dh3d_pack_flex_weight_x3.  wpost_packed (may be
 * NULL; then Dpost / out2 are ignored) as in dh3d_flex_conv_pm_post_fwd (Din == 128, Dout == 256, Dpost == 64).
 * (Din, Dout, K) in {(64,128,8), (128,128,8), (128,256,8), (128,128,12)}. */
int dh3d_flex_conv_pm_tile_x6_fwd(const float *features, const float *xyz, const int32_t *nbr, const void *wpacked_x3,
                                  int B, int N, int K, int Din, int Dout, const dh3d_epilogue *ep, float *out,
                                  const float *wpost_packed, int Dpost, float *out2, void *stream);
/* Same with group_point fused in: `features` is the [B, Nsrc, Din] map of the level above and point j of this level
 * (xyz / nbr / out are [B, N, ...]) is its row remap[b*N + j] (remap = the farthest-point-sampling picks).  Equal to
 * dh3d_flex_conv_pm_fwd(group_point(features, remap), ...) bit for bit. */
int dh3d_flex_conv_pm_gather_fwd(const float *features, const int32_t *remap, int Nsrc, const float *xyz,
                                 const int32_t *nbr, const float *wpacked, int B, int N, int K, int Din, int Dout,
                                 const dh3d_epilogue *ep, float *out, void *stream);

/* flex_conv of the full-resolution layers on the bf16 matrix pipe with f32 accuracy (three-way bf16 split of
 * both operands, six products; see csrc/flex_x6.hip).  Same operands and result as dh3d_flex_conv_pm_fwd up
 * to f32 summation order; K == 8, (Din, Dout) in {(32,64), (64,64)}.  The weight comes from
 * dh3d_pack_flex_weight_x3 ( [bias; theta_x; theta_y; theta_z] as three bf16 planes, 6*4*Din*Dout bytes ). */
int dh3d_pack_flex_weight_x3(const float *theta, const float *bias, int Din, int Dout, void *packed, void *stream);
int dh3d_flex_conv_pm_x6_fwd(const float *features, const float *xyz, const int32_t *nbr, const void *wpacked_x3,
                             int B, int N, int K, int Din, int Dout, const dh3d_epilogue *ep, float *out,
                             void *stream);

/* Same kernel with a placement hint: reserve_cus_per_xcd CUs of every XCD are known to be held by another stream's
 * kernel with a large LDS allocation (the model's farthest-point sampling: one CU per cloud for the whole step); the
 * persistent launch leaves that many workgroups per XCD out so that all of its workgroups are resident at once
 * instead of the last ones running as a second wave.  Speed only -- results are identical for any value in [0, 31]. */
int dh3d_flex_conv_pm_x6_fwd_r(const float *features, const float *xyz, const int32_t *nbr, const void *wpacked_x3,
                               int B, int N, int K, int Din, int Dout, const dh3d_epilogue *ep,
                               int reserve_cus_per_xcd, float *out, void *stream);

/* flex_pool forward, point-major: out[n,c] = max_k f[nbr[n,k],c], argmax may be NULL. C % 4 == 0. */
int dh3d_flex_pool_pm_fwd(const float *features, const int32_t *nbr, int B, int N, int K, int C,
                          float *out, int32_t *argmax, void *stream);

/* Flex_Avg forward (core/layers.py:342-436 = flex_conv with theta 0, bias eye), point-major:
 * out[n,c] = scale * sum_k f[nbr[n,k],c]  (backbones.py:80-82 passes scale = 1/knn).  C % 4 == 0. */
int dh3d_flex_avg_pm_fwd(const float *features, const int32_t *nbr, int B, int N, int K, int C,
                         float scale, float *out, void *stream);

/* conv_pointset forward on coordinates (Din = 3), point-major, + epilogue:
 * xyz [B,N,3], theta [3,Dout], bias [Dout] -> out [B,N,Dout].  Dout % 4 == 0, Dout <= 128. */
int dh3d_conv_pointset_pm_fwd(const float *xyz, const int32_t *nbr, const float *theta,
                              const float *bias, int B, int N, int K, int Dout,
                              const dh3d_epilogue *ep, float *out, void *stream);

/* S[n] = sum_k (xyz[nbr[n,k]] - xyz[nbr[n,0]]) as [B*N, 4] floats (x, y, z, 0): the one 3-vector conv_pointset on
 * coordinates is linear in (conv_pointset_kernel.cc:46-64, Din = 3) -- out = theta^T S + bias, grad_theta = S^T grad_out
 * (ConvPointsetGrad, conv_pointset_kernel_gpu.cu.cc:157-347, as one GEMM).  K == 8. */
int dh3d_pointset_sum_pm(const float *xyz, const int32_t *nbr, int B, int N, int K, float *S, void *stream);

/* conv_pointset on the coordinates (Din = 3) -> epilogue (bias once, BatchNorm, activation) -> flex_pool over the same
 * neighbourhoods, fused (core/backbones.py:107-110; conv_pointset_kernel.cc:46-64 + flex_pool_kernel.cc:41-57): the
 * [B,N,Dout] map between the two is never materialised.  out [B,N,Dout] = max_k act(bn(conv[nbr[n,k]])).  K == 8,
 * Dout in {32, 64, 128}; scratch: B*N*4 floats (one 3-vector per point, the sum of its neighbour offsets). */
int dh3d_conv_pointset_pool_pm_fwd(const float *xyz, const int32_t *nbr, const float *theta, const float *bias, int B,
                                   int N, int K, int Dout, const dh3d_epilogue *ep, float *scratch, float *out,
                                   void *stream);

/* Per-point linear layer (the 1x1 Conv2D of core/tf_utils.py:99-109):
 *   out[r,:] = epilogue( [x1[r,:] | x2[r,:]] @ W + b ) (+ residual[r,:] after the activation)
 * x1 [R,C1], x2 [R,C2] or NULL (fuses the concat of core/backbones.py:98-100), wpacked from
 * dh3d_pack_weight with Kd = C1+C2; b folded into ep->pre_bias.  C1,C2 % 8 == 0, Dout % 32 == 0. */
int dh3d_linear_pm_fwd(const float *x1, int C1, const float *x2, int C2, const float *wpacked,
                       int R, int Dout, const dh3d_epilogue *ep, const float *residual, float *out,
                       void *stream);

/* Squeeze-excite residual block (core/backbones.py:45-55), one kernel:
 *   g = sigmoid(relu(pool @ W1 + b1) @ W2 + b2);  out = relu(x + x*g)
 * x, pool [R,C]; W1 [C,C/4], W2 [C/4,C] row-major (unpacked).  C in {64,128}. */
int dh3d_se_res_pm_fwd(const float *x, const float *pool, const float *W1, const float *b1,
                       const float *W2, const float *b2, int R, int C, float *out, void *stream);
/* The same block on the matrix pipe: w1packed = dh3d_pack_weight of W1 zero-padded to [C,32] columns, b1pad = b1
 * zero-padded to 32, w2packed = dh3d_pack_weight of W2 zero-padded to [32,C] rows. */
int dh3d_se_res_pm_packed_fwd(const float *x, const float *pool, const float *w1packed, const float *b1pad,
                              const float *w2packed, const float *b2, int R, int C, float *out, void *stream);
/* the same block on flex_pool(x, nbr) (core/backbones.py:76-79) in one launch: the pooled rows are formed while staging,
 * the [B*N, C] pooled map is never written.  x [B*N, C], nbr [B, N, K] int32; C = 64 or 128. */
int dh3d_se_res_pool_pm_packed_fwd(const float *x, const int32_t *nbr, int B, int N, int K, const float *w1packed,
                                   const float *b1pad, const float *w2packed, const float *b2, int C, float *out,
                                   void *stream);
/* ... followed by a 1x1 conv C -> C (+ bias / BatchNorm / activation `ep`) on the block's output in the same launch:
 * out and out2 [B*N, C] are both stored.  C = Dout = 64 (stage 1 -> before_stage2_conv1d, core/backbones.py:115-117) or
 * C = Dout = 128 (stage 2's SE block -> the upper block of its commuted concat conv on the sampled rows). */
int dh3d_se_res_pool_conv_pm_fwd(const float *x, const int32_t *nbr, int B, int N, int K, const float *w1packed,
                                 const float *b1pad, const float *w2packed, const float *b2, int C, float *out,
                                 const float *wconv_packed, const dh3d_epilogue *ep, int Dout, float *out2, void *stream);

/* dh3d_se_res_pool_conv_pm_fwd for C = Dout = 64 with two more 1x1 convs 64 -> 128 riding in the launch, on the bf16 pipe at
 * f32 accuracy: out_a = act_a(bn_a(y @ Wa)) on the block's output y (out), out_b = act_b(bn_b(z @ Wb)) on z = the 64 -> 64
 * conv's output (out2).  wa_x3 / wb_x3 = dh3d_pack_weight_x3 of [64, 128]; act NONE or RELU.  out may be NULL (y is not
 * stored).  The local step: stage 1's SE block + before_stage2_conv1d + local_stage1_shortcut + the lower block of stage 2's
 * commuted concat conv (core/backbones.py:115-123) -- one launch instead of three over the same tiles. */
int dh3d_se_res_pool_conv_tails_pm_fwd(const float *x, const int32_t *nbr, int B, int N, int K, const float *w1packed,
                                       const float *b1pad, const float *w2packed, const float *b2, float *out,
                                       const float *wconv_packed, const dh3d_epilogue *ep, float *out2, const void *wa_x3,
                                       const dh3d_epilogue *ep_a, float *out_a, const void *wb_x3,
                                       const dh3d_epilogue *ep_b, float *out_b, void *stream);

/* three_nn + inverse-distance weights + three_interpolate (core/backbones.py:90-96) fused:
 * weight = (1/max(d,1e-10)) / sum(1/max(d,1e-10)).  idx/dist from dh3d_three_nn.
 * points [b,m,c] -> out [b,n,c]; c % 4 == 0. */
int dh3d_three_interpolate_idw_fwd(int b, int m, int c, int n, const float *points,
                                   const int32_t *idx, const float *dist, float *out, void *stream);

/* Row-wise L2 normalisation x / sqrt(max(sum x^2, eps)) (tf.nn.l2_normalize, core/model.py:177,205)
 * with optional prefix columns copied in front (the xyz of 'xyz_feat', core/model.py:181):
 * out[r,:] = [prefix[r,:P] | normalised x[r,:C]]. */
int dh3d_l2norm_concat_fwd(const float *x, int R, int C, float eps, const float *prefix, int P,
                           float *out, void *stream);

/* Point-wise attention / detector head (core/backbones.py:132-173): a chain of per-point linear
 * layers ending in a 1-channel logit + sigmoid, with the last wide hidden layer never written to
 * memory:  att[r] = sigmoid( relu(bn(h[r,:] @ W + b)) . w_fc + b_fc ),  h [R,C], W packed [C,H]. */
int dh3d_mlp_head_pm_fwd(const float *h, int R, int C, const float *wpacked, int H,
                         const dh3d_epilogue *ep, const float *w_fc, float b_fc, float *att,
                         void *stream);

/* dh3d_linear_pm_fwd on the bf16 matrix pipe at f32 accuracy (bf16x6, csrc/dense_x6.hip) for the wide 1x1 convs:
 * Dout in {128, 256}, C1 and C2 multiples of 32, weight from dh3d_pack_weight_x3.  Same result up to f32
 * summation order. */
int dh3d_linear_pm_x6_fwd(const float *x1, int C1, const float *x2, int C2, const void *wpacked_x3, int R, int Dout,
                          const dh3d_epilogue *ep, const float *residual, float *out, void *stream);
/* The same GEMM with its x1 half up-sampled on the fly: x1[b,j,:] = three_interpolate(points [B,m,C1], idx [B,n,3],
 * inverse-distance weights of dist [B,n,3]) (core/backbones.py:91-95 + tf_interpolate.cpp:107-127) -- the
 * interpolated [B,n,C1] tensor is never written.  Bit-identical to dh3d_three_interpolate_idw_fwd followed by
 * dh3d_linear_pm_x6_fwd. */
int dh3d_upsample_linear_pm_x6_fwd(const float *points, const int32_t *idx, const float *dist, int B, int n, int m,
                                   int C1, const float *x2, int C2, const void *wpacked_x3, int Dout,
                                   const dh3d_epilogue *ep, const float *residual, float *out, void *stream);
/* the same with the local path's last step fused into the store (core/model.py:177-181): out_cat [B*n, 3+128] =
 * [prefix [B*n,3] | l2_normalize(y, l2_eps)], Dout == 128; y itself is not written */
int dh3d_upsample_linear_l2cat_pm_x6_fwd(const float *points, const int32_t *idx, const float *dist, int B, int n,
                                         int m, int C1, const float *x2, int C2, const void *wpacked_x3, int Dout,
                                         const dh3d_epilogue *ep, const float *residual, const float *prefix,
                                         float l2_eps, float *out_cat, void *stream);
/* out = act(BN([upsample(points) | x2] W)) + act_sc(BN_sc(x3 W_sc)): the concat conv with the local backbone's shortcut
 * conv (core/backbones.py:123) in one kernel; wpacked_x3 = dh3d_pack_weight_x3 of [W; W_sc] ([C1+C2+C3, Dout]),
 * Dout == 128; prefix != NULL: out is [B*n, 3+128] = [prefix | l2_normalize(sum, l2_eps)] */
int dh3d_upsample_linear_shortcut_pm_x6_fwd(const float *points, const int32_t *idx, const float *dist, int B, int n,
                                            int m, int C1, const float *x2, int C2, const float *x3, int C3,
                                            const void *wpacked_x3, int Dout, const dh3d_epilogue *ep,
                                            const dh3d_epilogue *ep_shortcut, const float *prefix, float l2_eps,
                                            float *out, void *stream);

/* The same head with the GEMM on the bf16 matrix pipe at f32 accuracy ("bf16x6": every f32 operand is split
 * exactly into three bf16 chunks, six chunk products are accumulated in f32; error <= 2^-23 per product, see
 * csrc/dense_x6.hip).  wpacked_x3 from dh3d_pack_weight_x3 (3 * Kd * Dout * 2 bytes).  C % 16 == 0. */
int dh3d_pack_weight_x3(const float *W, int Kd, int Dout, void *packed, void *stream);
int dh3d_mlp_head_pm_x6_fwd(const float *h, int R, int C, const void *wpacked_x3, int H,
                            const dh3d_epilogue *ep, const float *w_fc, float b_fc, float *att, void *stream);

/* The same head on an up-sampled input with the wide conv commuted through the (linear, weights sum to 1) 3-NN
 * interpolation of core/backbones.py:91-95:  att = sigmoid(w_fc . act(BN(interp(H) + pre_bias)) + b_fc),  H = x_coarse W
 * computed on the m coarse rows, layout [Hd/256][B*m][256] (one dh3d_linear_pm_x6_fwd per 256-column weight slice);
 * idx / dist [B*n,3] from dh3d_three_nn.  8x fewer GEMM flops than running the head on the up-sampled rows. */
int dh3d_interp_head_fwd(const float *H, int Hd, const int32_t *idx, const float *dist, int B, int n, int m,
                         const dh3d_epilogue *ep, const float *w_fc, float b_fc, float *att, void *stream);
/* Same head with the fine points walked in the Morton order of `order` (the dh3d_spatial_sort records [B,n,4] of the
 * fine cloud; NULL = index order) and the distinct coarse rows of 128 consecutive points staged in LDS, one 256-channel
 * slice at a time: ~10x less L2 traffic.  m <= 1024.  Equal to dh3d_interp_head_fwd up to the summation order of the
 * row dot. */
int dh3d_interp_head_sorted_fwd(const float *H, int Hd, const int32_t *idx, const float *dist, const float *order, int B,
                                int n, int m, const dh3d_epilogue *ep, const float *w_fc, float b_fc, float *att,
                                void *stream);
/* the same with the fc bias in device memory (a trainable parameter); row_major != 0: H is [B*m][Hd] (one GEMM's
 * output) instead of the 256-column slices */
int dh3d_interp_head_sorted_fwd_dev(const float *H, int Hd, int row_major, const int32_t *idx, const float *dist,
                                    const float *order, int B, int n, int m, const dh3d_epilogue *ep, const float *w_fc,
                                    const float *b_fc_dev, float *att, void *stream);
/* The tail of a concat conv commuted through the up-sampling (C = 128):
 *   out[n] = act(BN(interp3(coarse_w)[n] + partial[n] + pre_bias)) + residual[n]     (or [prefix | l2_normalize(.)])
 * coarse_w [B,M,C] = coarse rows already multiplied by the conv's upper weight block, partial [B,N,C] = the lower block
 * applied to the full-resolution input (may be NULL), idx/dist [B,N,3] from three_nn (inverse-distance weights as
 * core/backbones.py:92-95), residual / prefix ([B,N,3]) may be NULL. */
int dh3d_interp_combine_fwd(const float *coarse_w, const int32_t *idx, const float *dist, const float *partial, int B,
                            int N, int M, int C, const dh3d_epilogue *ep, const float *residual, const float *prefix,
                            float l2_eps, float *out, void *stream);
/* The whole tail of the local-descriptor forward behind the sampled level in one launch (core/backbones.py:89-100,
 * 117-123; core/model.py:177-181) -- dh3d_interp_combine_fwd with its `partial` and `residual` operands computed on the
 * fly instead of read back from memory:
 *   out[n] = [ prefix[n] | l2_normalize( relu(BN_c(interp3(coarse_w)[n] + x2[n] W_lower + b_c)) + relu(BN_s(x1[n] W_s + b_s)) ) ]
 * x1, x2 [B,N,64]; wpacked_x3_* = dh3d_pack_weight_x3 of the [64,128] shortcut weight / of the concat conv's lower block
 * (f32-accurate bf16x6 products); ep_shortcut / ep_concat: bias + folded BatchNorm, activation ReLU; coarse_w [B,M,128];
 * idx / dist [B,N,3] from three_nn; prefix [B,N,3]; out [B,N,131].  N % 32 == 0.
 * prefix == NULL: out [B,N,128] = the sum before the normalisation (the global path's local features; l2_eps unused). */
int dh3d_local_tail_fused_fwd(const float *x1, const float *x2, const void *wpacked_x3_shortcut,
                              const void *wpacked_x3_lower, const dh3d_epilogue *ep_shortcut,
                              const dh3d_epilogue *ep_concat, const float *coarse_w, const int32_t *idx, const float *dist,
                              const float *prefix, float l2_eps, int B, int N, int M, float *out, void *stream);

/* a layer wider than 256 as `slices` column slices of 256 in one launch: out [slices][R][256] = x1 @ W[:, 256 j ..],
 * wpacked_x3 = the dh3d_pack_weight_x3 images of the slices back to back (the H operand of dh3d_interp_head_fwd) */
int dh3d_linear_slices_pm_x6_fwd(const float *x1, int C1, const void *wpacked_x3, int R, int slices, float *out,
                                 void *stream);

/* Attention-weighted NetVLAD aggregation (core/backbones.py:202-262), stage 1+2:
 *   xn = l2norm(x); a = softmax(bn(xn @ Wc)) * att;  vlad[b,d,c] = sum_n a[n,c] xn[n,d] - (sum_n a[n,c]) W2[d,c]
 *   then intra-normalise over d per cluster and L2-normalise the flattened [D*Cl] vector (d-major).
 * x [B,N,D], att [B,N], wc_packed = pack(Wc [D,Cl]), bn as (scale,shift) [Cl], W2 [D,Cl]
 * -> vlad [B, D*Cl].  D = 256, Cl = 64.  workspace: dh3d_netvlad_workspace_bytes(B,N,D,Cl). */
size_t dh3d_netvlad_workspace_bytes(int B, int N, int D, int Cl);
int dh3d_netvlad_aggregate_fwd(const float *x, const float *att, const float *wc_packed,
                               const float *bn_scale, const float *bn_shift, const float *W2, int B,
                               int N, int D, int Cl, void *workspace, size_t workspace_bytes,
                               float *vlad, void *stream);

/* NetVLAD projection + context gating (core/backbones.py:262-320):
 *   h = bn1(vlad @ Wh);  out = h * sigmoid(bn2(h @ Wg));  optional final L2 normalise (model.py:205).
 * vlad [B,Kd], Wh [Kd,O], Wg [O,O] row-major; bn as (scale,shift) [O]. O = 256.
 * workspace: dh3d_netvlad_head_workspace_bytes(B,Kd,O). */
size_t dh3d_netvlad_head_workspace_bytes(int B, int Kd, int O);
int dh3d_netvlad_head_fwd(const float *vlad, const float *Wh, const float *bn1_scale,
                          const float *bn1_shift, const float *Wg, const float *bn2_scale,
                          const float *bn2_shift, int B, int Kd, int O, float l2_eps,
                          void *workspace, size_t workspace_bytes, float *out, void *stream);

/* ===================================================================================== *
 * C. Backward / training kernels (point-major)
 * ===================================================================================== */

/* flex_conv backward, factorised (csrc/flex_bwd.hip): features [B,N,Din], xyz [B,N,3], nbr [B,N,K], theta [3,Din,Dout],
 * bias [Din,Dout], grad_out [B,N,Dout] -> grad_features [B,N,Din] (may be NULL: weights only), grad_theta, grad_bias.
 * center_rank0: offsets relative to the rank-0 neighbour (the reference backward's rule) instead of the point itself.
 * Zeroes its outputs.  workspace: dh3d_flex_conv_pm_bwd_workspace_bytes.  Din % 4 == 0, Dout % 4 == 0. */
size_t dh3d_flex_conv_pm_bwd_workspace_bytes(int B, int N, int Din, int Dout);
int dh3d_flex_conv_pm_bwd(const float *features, const float *xyz, const int32_t *nbr, const float *theta,
                          const float *bias, const float *grad_out, int B, int N, int K, int Din, int Dout,
                          int center_rank0, void *workspace, size_t workspace_bytes, float *grad_features,
                          float *grad_theta, float *grad_bias, void *stream);

/* f32 GEMMs of the backward passes (csrc/gemm.hip), all row-major; f32-accurate: the exact-f32 matrix pipe for small
 * products, the bf16 pipe with a three-way split of BOTH operands (six products, ~2^-23 relative) from 2^26
 * multiply-adds when K % 4 == 0 (environment DH3D_GEMM_F32=1 keeps everything on the exact-f32 kernel):
 *   tn: C[M,N] (+)= A[K,M]^T B[K,N]  (weight gradients: reduction over rows, split over workgroups + f32 atomics)
 *   nn: C[M,N] (+)= A[M,K]   B[K,N] (+ colbias[N], may be NULL; not with accumulate)  (linear layers of the
 *       training step, input gradients with W^T materialised)
 * accumulate = 0 overwrites C.  M, N (tn) / K, N (nn) multiples of 4. */
/* 1 when the product's reduction is split over workgroups (C is then zeroed by the call and the partials added with f32
 * atomics; with accumulate = 1 they are added onto the caller's C -- a caller holding zeroed memory saves the fill). */
int dh3d_gemm_is_split(int ta /* 1: tn form */, int M, int N, int K, int batch);
int dh3d_gemm_tn_f32(const float *A, const float *B, int K, int M, int N, int accumulate, float *C, void *stream);
int dh3d_gemm_nn_f32(const float *A, const float *B, const float *colbias, int M, int K, int N, int accumulate,
                     float *C, void *stream);
/* [Bt,R,C] -> [Bt,C,R] of 32-bit elements; out[c] (+)= sum_r x[r,c]. */
int dh3d_transpose32(const void *in, int Bt, int R, int C, void *out, void *stream);
int dh3d_colsum_f32(const float *x, long long R, int C, int accumulate, float *out, void *stream);

/* Training-mode BatchNorm in two HBM-bound passes per direction and the attention head's element-wise passes
 * (csrc/train.hip; tensorpack BatchNorm / slim batch_norm with is_training -- core/tf_utils.py:60-63,
 * core/backbones.py:145-173,218-223,271-274).  x [R,C] row-major; mask (may be NULL): one byte per cloud of
 * rows_per_cloud rows, 0 = padding cloud, excluded.  Sums are f64 (f32 per thread, f64 atomics across workgroups).
 *   colstats       : sum[c] = sum_r x, sumsq[c] = sum_r x^2                         (zeroes its outputs)
 *   scale_shift_act: y = act(x*scale[c] + shift[c])   (relu = 0/1; y may alias x)
 *   row_logit_sigmoid: att[r] = sigmoid(sum_c relu(h*scale+shift)[r,c] w[c] + b)    (attention head, backbones.py:170-173)
 *   bn_colstats    : sum / sumsq [C] f64 += column sums of x over the live rows (zeroed by the CALLER)
 *   bn_bwd_sums    : (S1, S2, S3 zeroed by the CALLER) S1 = sum dz, S2 = sum dz*xhat with xhat = (x-mean)*rstd, dz = dy masked by relu(xhat*gamma+beta) > 0;
 *                    dy == NULL: dy[r,c] = rowscale[r]*colvec[c] (rank one) and S3[c] = sum_r rowscale[r]*y[r,c]
 *   bn_finalize    : (sum, sumsq, count) -> mean, rstd, scale = gamma*rstd, shift = beta - mean*scale; running buffers
 *                    <- momentum*old + (1-momentum)*batch statistic (untouched when count == 0)
 *   bn_bwd_finalize: (S1, S2, count) -> k2, k3 of bn_bwd_apply
 *   bn_bwd_apply   : dx = scale*dz - k2[c] - k3[c]*x  == gamma*rstd*(dz - S1/n - xhat*S2/n)   (dx may alias x) */
/* bn_bwd_finalize from per-cloud partial sums part [nk][P][C] f64 (nk = 2: S1, S2; 3: + S3): their sums over P give k2 / k3
 * and, as float32, grads [nk][C] (dbeta, dgamma, d w_fc).  dh3d_sigmoid_bwd: dlogit = datt*att*(1-att) (0 on rows of
 * masked clouds) and sum[0] += sum dlogit (sum zeroed by the CALLER). */
int dh3d_bn_finalize_parts(const double *part /* [2][P][C]: sum | sumsq per cloud */, int P, const double *count,
                           const float *gamma, const float *beta, float eps, float momentum, int unbiased, float *run_mean,
                           float *run_var, int C, float *mean, float *rstd, float *scale, float *shift, void *stream);
int dh3d_bn_bwd_finalize_parts(const double *part, int nk, int P, const double *count, const float *mean,
                               const float *rstd, const float *gamma, int C, float *k2, float *k3, float *grads,
                               void *stream);
int dh3d_sigmoid_bwd(const float *datt, const float *att, const unsigned char *mask, int rows_per_cloud, long long n,
                     float *dlogit, float *sum, void *stream);
int dh3d_bn_colstats(const float *x, long long R, int C, const unsigned char *mask, int rows_per_cloud, double *sum,
                     double *sumsq, void *stream);
int dh3d_scale_shift_act(const float *x, long long R, int C, const float *scale, const float *shift, int relu,
                         float *y, void *stream);
/* The same pass with `residual` [R,C] (or NULL) added AFTER the activation: y = act(x*scale+shift) + residual -- the
 * shortcut sum behind the last BatchNorm of the local backbone in training mode (core/backbones.py:123). */
int dh3d_scale_shift_act_res(const float *x, long long R, int C, const float *scale, const float *shift, int relu,
                             const float *residual, float *y, void *stream);

/* Element-wise / scatter passes of the LOCAL backbone's training step (stage 1-2: basic_config / detection_config,
 * core/model.py:212-246; dh3d_amd/training.py LocalTrainer), point-major rows:
 *   dh3d_flex_pool_pm_bwd : FlexPoolGrad (flex_pool_kernel_gpu.cu.cc:65-93) -- din[cloud(n) + argmax[n,c], c] += dout[n,c]
 *                           with f32 atomics (as the reference); din [B*N, C] zeroed by the CALLER;
 *   dh3d_se_gate_fwd/_bwd : y = relu(x + x * sigmoid(z)) and its gradients (se_res_bottleneck's tail, backbones.py:52-55);
 *   dh3d_relu_fwd/_bwd    : y = relu(x);  dx = y > 0 ? dy : 0.   Element counts are multiples of 4. */
/* pairwise_dist (core/tf_utils.py:125-136) of the local losses: out[b,i,j] = sum_d (A[b,i,d] - Bm[b,j,d])^2 with the
 * differences formed as upstream; A [B,n,D], Bm [B,m,D] -> out [B,n,m]. */
int dh3d_pairwise_sqdist(const float *A, const float *Bm, int B, int n, int m, int D, float *out, void *stream);
int dh3d_flex_pool_pm_bwd(const float *dout, const int32_t *argmax, int B, int N, int C, float *din, void *stream);
int dh3d_se_gate_fwd(const float *x, const float *z, long long n, float *y, void *stream);
int dh3d_se_gate_bwd(const float *x, const float *z, const float *dy, long long n, float *dx, float *dz, void *stream);
int dh3d_relu_fwd(const float *x, long long n, float *y, void *stream);
int dh3d_relu_bwd(const float *y, const float *dy, long long n, float *dx, void *stream);
int dh3d_row_logit_sigmoid(const float *h, long long R, int C, const float *scale, const float *shift, const float *w,
                           const float *b /* device scalar */, float *att, void *stream);
int dh3d_bn_bwd_sums(const float *x, const float *dy, const float *rowscale, const float *colvec, long long R, int C,
                     const float *mean, const float *rstd, const float *gamma, const float *beta, int relu,
                     const unsigned char *mask, int rows_per_cloud, double *S1, double *S2, double *S3, void *stream);
int dh3d_bn_bwd_apply(const float *x, const float *dy, const float *rowscale, const float *colvec, long long R, int C,
                      const float *scale, const float *shift, const float *k2, const float *k3, int relu,
                      const unsigned char *mask, int rows_per_cloud, float *dx, void *stream);
int dh3d_bn_finalize(const double *sum, const double *sumsq, const double *count /* device scalar */,
                     const float *gamma, const float *beta, float eps, float momentum,
                     int unbiased /* moving variance <- Bessel-corrected batch variance (fused_batch_norm) */,
                     float *run_mean, float *run_var, int C, float *mean, float *rstd, float *scale, float *shift,
                     void *stream);
int dh3d_bn_bwd_finalize(const double *S1, const double *S2, const double *count, const float *mean, const float *rstd,
                         const float *gamma, int C, float *k2, float *k3, void *stream);

/* NetVLAD soft assignment rows (core/backbones.py:214-238; Cl = 64, one wave per row):
 *   a[r,:] = softmax(s[r,:]*scale + shift) * att[r];   backward: da -> dz (gradient at the BatchNorm output), datt.
 * asum (may be NULL) [R / rows_per_cloud, 64]: the per-cloud column sums of a (backbones.py:241) from the same pass
 * (rows_per_cloud % 64 == 0).  l2norm_rows_bwd: backward of xn = x * rsqrt(max(sum x^2, eps)) (tf.nn.l2_normalize).
 * idw_weights: three_interpolate's inverse-distance weights from three_nn's distances (backbones.py:92-95).
 * context_gate: y = v * sigmoid(g) (backbones.py:271-277) and its backward. */
int dh3d_netvlad_assign_rows(const float *s, long long R, int Cl, const float *scale, const float *shift,
                             const float *att, float *a, float *asum, long long rows_per_cloud, void *stream);
int dh3d_netvlad_assign_rows_bwd(const float *s, long long R, int Cl, const float *scale, const float *shift,
                                 const float *att, const float *da, float *dz, float *datt, void *stream);
int dh3d_l2norm_rows_bwd(const float *x, const float *dxn, long long R, int C, float eps, float *dx, void *stream);
int dh3d_idw_weights(const float *dist, long long R, float *w, void *stream);
int dh3d_context_gate_fwd(const float *v, const float *g, long long n, float *y, void *stream);
int dh3d_context_gate_bwd(const float *v, const float *g, const float *dy, long long n, float *dv, float *dg,
                          void *stream);
/* Training-mode attention head with its convolution commuted through the up-sampling (csrc/interp_train.hip): the
 * pre-activation h = three_interpolate(G) of globalatt_block (core/backbones.py:89-100,156-173), G = coarse @ W + b as
 * 256-column slices [Hd/256][B*m][256] (row_major = 0) or as the GEMM's own [B*m][Hd] output (row_major = 1; dG likewise),
 * is never materialised.  idx / dist [B,n,3] = three_nn of the fine points, order =
 * dh3d_spatial_sort records [B,n,4] of the fine cloud (NULL: index order), mask [B] bytes (NULL: all clouds live).
 *   colstats : per-cloud partials part [2][B][Hd] f64 (zeroed by the CALLER) of sum / sumsq of h over the live rows; their
 *              sums over B -> dh3d_bn_finalize
 *   forward  : dh3d_interp_head_sorted_fwd with the folded batch statistics as its epilogue
 *   bwd_sums : per-cloud partials part [3][B][Hd] f64 (zeroed by the CALLER) of S1, S2, S3 of dh3d_bn_bwd_sums for dy = dlogit x w_fc (dlogit [B*n]
 *              by original point index)
 *   bwd_apply: dG [Hd/256][B*m][256] = interp^T(scale dz - k2 - k3 h) (zeroed by the CALLER, f32 atomics), from which
 *              dW = coarse^T dG and dcoarse = dG W^T are GEMMs on B*m rows.  Hd <= 1024, m <= 1024. */
int dh3d_interp_bn_colstats(const float *G, int Hd, int row_major, const int32_t *idx, const float *dist,
                            const float *order, int B, int n, int m, const unsigned char *mask,
                            double *part /* [2][B][Hd] */, void *stream);
int dh3d_interp_bn_bwd_sums(const float *G, int Hd, int row_major, const int32_t *idx, const float *dist,
                            const float *order, int B, int n, int m, const unsigned char *mask, const float *dlogit, const float *w_fc,
                            const float *mean, const float *rstd, const float *gamma, const float *beta,
                            double *part /* [3][B][Hd] */, void *stream);
int dh3d_interp_bn_bwd_apply(const float *G, int Hd, int row_major, const int32_t *idx, const float *dist,
                             const float *order, int B, int n, int m, const unsigned char *mask, const float *dlogit, const float *w_fc,
                             const float *scale, const float *shift, const float *k2, const float *k3, float *dG,
                             void *stream);
/* three_interpolate's backward (threeinterpolate_grad_cpu, tf_interpolate.cpp:131-153) on the Morton order of the fine
 * cloud (order = dh3d_spatial_sort records [b,n,4], NULL: index order): the rows of a 128-point block are scattered onto
 * the <= 56 coarse rows it touches by an MFMA product in LDS, then added to grad_points [b,m,c] (zeroed by the call) with
 * one f32 atomic per (block, row, channel).  c == 256, m <= 1024; same result as dh3d_three_interpolate_bwd up to the
 * summation order. */
int dh3d_three_interpolate_bwd_sorted(int b, int n, int c, int m, const float *grad_out, const int32_t *idx,
                                      const float *weight, const float *order, float *grad_points, void *stream);
/* NetVLAD's assignment (core/backbones.py:202-256) in the training step with its rows commuted through
 * three_interpolate (core/backbones.py:89-100), csrc/netvlad_train.hip: c [B*m,256] are the sampled rows, cw = c Wc and
 * E = c dV^T GEMMs on them; s / p / dz [B*n,64], rinv / att / datt / t2 / q [B*n] live on the fine points (by original
 * point index); idx / dist = dh3d_three_nn of the fine points [B,n,3], order = dh3d_spatial_sort records of the fine
 * clouds [B,n,4] (NULL: index order), mask [B] bytes (NULL: every cloud).  m <= 1024.
 *   fwd_stats : s = rinv * interp(cw), rinv = rsqrt(max(|interp(c)|^2, 1e-12)), part [2][B][64] f64 = per-cloud column
 *               sums / sums of squares of s (zeroed by the CALLER)
 *   fwd_assign: p = softmax(s*scale + shift), asum [B,64] = sum_n p*att, Ap [B*m,64] = interp^T(p*att*rinv) (both zeroed
 *               by the CALLER; V[b] = Ap[b]^T c[b])
 *   bwd_sums  : da = rinv*interp(E) + dasum; datt = sum_k da p (0 for masked clouds); dz = softmax backward of da*att;
 *               t2 = sum_k p att interp(E); part [2][B][64] f64 (zeroed by the CALLER) = per-cloud sums of dz and dz*(s - mean)*rstd
 *   bwd_apply : ds = k1*dz - k2 - k3*s (dh3d_bn_bwd_finalize); q = rinv^2 sum_k ds s + rinv^3 t2; dcw [B*m,64] =
 *               interp^T(rinv*ds) (zeroed by the CALLER)
 * dh3d_interp_scatter_scaled: dc [B*m,256] += interp^T(-q * interp(c)) (dc NOT zeroed: it holds Ap dV + dcw Wc^T). */
int dh3d_netvlad_commuted_fwd_stats(const float *c, const float *cw, const int32_t *idx, const float *dist,
                                    const float *order, int B, int n, int m, const unsigned char *mask, float *s,
                                    float *rinv, double *part, void *stream);
int dh3d_netvlad_commuted_fwd_assign(const float *s, const float *rinv, const float *att, const float *scale,
                                     const float *shift, const int32_t *idx, const float *dist, const float *order, int B,
                                     int n, int m, const unsigned char *mask, float *p, float *asum, float *Ap,
                                     void *stream);
int dh3d_netvlad_commuted_bwd_sums(const float *E, const float *p, const float *s, const float *att, const float *rinv,
                                   const float *dasum, const float *mean, const float *rstd, const int32_t *idx,
                                   const float *dist, const float *order, int B, int n, int m, const unsigned char *mask,
                                   float *dz, float *datt, float *t2, double *part, void *stream);
int dh3d_netvlad_commuted_bwd_apply(const float *dz, const float *s, const float *rinv, const float *t2, const float *k1,
                                    const float *k2, const float *k3, const int32_t *idx, const float *dist,
                                    const float *order, int B, int n, int m, const unsigned char *mask, float *q,
                                    float *dcw, void *stream);
int dh3d_interp_scatter_scaled(const float *c, const float *q, const int32_t *idx, const float *dist, const float *order,
                               int B, int n, int m, const unsigned char *mask, float *dc, void *stream);
/* NetVLAD between the VLAD contraction and the hidden projection (core/backbones.py:241-262) for the training step:
 * out[b, d*64 + c] = l2norm_all( intra_norm_d( V[b,c,d] - asum[b,c] * W2[d,c] ) ), one workgroup per cloud, and its
 * gradients (dW2 zeroed by the call, f32 atomics over the clouds).  D == 256, Cl == 64. */
int dh3d_vlad_normalize_fwd(const float *V, const float *asum, const float *W2, int B, int D, int Cl, float eps,
                            float *out, float *inv_c, float *inv_t, void *stream);
int dh3d_vlad_normalize_bwd(const float *V, const float *asum, const float *W2, const float *grad_out, int B, int D,
                            int Cl, float eps, float *dV, float *dasum, float *dW2, void *stream);
/* lazy quadruplet loss (core/losses.py:137-200) and its gradient in one launch: desc [B*(2+P+Ng), 256] role-ordered
 * (queries | positives | negatives | other negatives); loss[0] is zeroed by the call; grad has desc's shape. */
int dh3d_quadruplet_loss(const float *desc, int B, int P, int Ng, int D, float margin, float margin2, float *loss,
                         float *grad, void *stream);
/* training-mode BatchNorm (+ReLU) of a short tensor (R <= 64 rows, e.g. the [clouds, 256] activations behind NetVLAD) in
 * one launch per direction; stats [4,C] = mean, rstd, scale, shift of the forward; mask [R] bytes or NULL. */
int dh3d_bn_small_fwd(const float *x, int R, int C, const float *gamma, const float *beta, float eps, float momentum,
                      int unbiased, int relu, const unsigned char *mask, float *run_mean, float *run_var, float *stats,
                      float *y, void *stream);
int dh3d_bn_small_bwd(const float *x, const float *dy, int R, int C, const float *gamma, const float *stats, int relu,
                      const unsigned char *mask, float *dx, float *dgamma, float *dbeta, void *stream);
/* batched GEMMs: `batch` independent products on operands stored back to back; colbias [batch, N] (nn only). */
int dh3d_gemm_tn_f32_batched(const float *A, const float *B, int batch, int K, int M, int N, float *C, void *stream);
int dh3d_gemm_nn_f32_batched(const float *A, const float *B, const float *colbias, int batch, int M, int K, int N,
                             float *C, void *stream);

/* The global-descriptor tail with the up-sampling commuted through BOTH consumers of the up-sampled map (attention MLP
 * and NetVLAD; csrc/dense_x6.hip VladTail): dh3d_global_tail_fwd walks the fine points once (Morton order of `order`,
 * coarse rows staged in LDS) from H = the 256-column slices of coarse @ W_att (dh3d_linear_slices_pm_x6_fwd), coarse
 * [B,m,256] and cw = coarse @ cluster_weights [B,m,64], and produces att [B,n] (may be NULL) and
 * accum = [ apart B*m*64 | asum B*64 | V B*64*256 ] floats (zeroed by the call): apart = A' (softmax * attention
 * scattered onto the coarse rows, f32 atomics), asum its per-cluster sums, V[b] = apart[b]^T coarse[b].
 * dh3d_netvlad_tail_fwd(V, asum, ...) finishes (subtract asum*W2, intra-normalise, project, gate).  Same function as
 * three_interpolate -> attention head -> dh3d_netvlad_fused_fwd, reassociated; m <= 1024. */
int dh3d_global_tail_fwd(const float *H, int Hd, const float *coarse, const float *cw, const int32_t *idx,
                         const float *dist, const float *order, int B, int n, int m, const dh3d_epilogue *ep,
                         const float *w_fc, float b_fc, const float *cl_scale, const float *cl_shift, float *att,
                         float *accum, void *stream);
size_t dh3d_netvlad_tail_workspace_bytes(int B, int D, int Cl, int O);
int dh3d_netvlad_tail_fwd(const float *V, const float *asum, const float *W2, const float *Wh, const float *bn1_scale,
                          const float *bn1_shift, const float *Wg, const float *bn2_scale, const float *bn2_shift, int B,
                          int D, int Cl, int O, float l2_eps, void *workspace, size_t workspace_bytes, float *out,
                          void *stream);
/* The two-call form the model uses since round 4: dh3d_global_walk_fwd = the walk alone, accum = [ apart B*m*64 | asum B*64 ]
 * floats (zero_accum != 0: cleared by the call; 0: zeroed by the CALLER, e.g. by a fill issued beside the sampling chain),
 * and dh3d_netvlad_tail_assign_fwd, which forms V = apart^T coarse inside its finalize kernel (no batched GEMM launch, a
 * fixed summation order) and then projects and gates as dh3d_netvlad_tail_fwd.  m <= 1024; workspace:
 * dh3d_netvlad_tail_workspace_bytes. */
int dh3d_global_walk_fwd(const float *H, int Hd, const float *coarse, const float *cw, const int32_t *idx, const float *dist,
                         const float *order, int B, int n, int m, const dh3d_epilogue *ep, const float *w_fc, float b_fc,
                         const float *cl_scale, const float *cl_shift, float *att, float *accum, int zero_accum, void *stream);
/* Round 6: the walk's slot tables built AHEAD of it.  Inside the walk the table of a 128-point block (bitmap of the coarse
 * rows its points touch -> prefix popcounts -> slots; three dependent global round trips + five barriers) was 17 of the
 * launch's ~95 us at 32 x 4096.  dh3d_walk_plan builds every block's table from the three_nn result (idx, dist [B,n,3], order =
 * dh3d_spatial_sort records of the fine cloud, may be NULL) into `plan` (dh3d_walk_plan_bytes(B, n) bytes, opaque:
 * per block the slot table and the slot-major lists of the references to every staged row, which the planned walk's
 * NetVLAD scatter follows instead of forming a selection matrix on the matrix pipe) -- launched behind dh3d_three_nn_*, off the critical chain -- and dh3d_global_walk_planned_fwd is
 * dh3d_global_walk_fwd reading it (plan == NULL: the table is built inside the walk as before).  Same results bit for bit
 * up to the order of the f32 atomics (as dh3d_global_walk_fwd).  m <= 1024. */
size_t dh3d_walk_plan_bytes(int B, int n);
int dh3d_walk_plan(const int32_t *idx, const float *dist, const float *order, int B, int n, int m, void *plan, void *stream);
int dh3d_global_walk_planned_fwd(const float *H, int Hd, const float *coarse, const float *cw, const int32_t *idx,
                                 const float *dist, const float *order, const void *plan, int B, int n, int m,
                                 const dh3d_epilogue *ep, const float *w_fc, float b_fc, const float *cl_scale,
                                 const float *cl_shift, float *att, float *accum, int zero_accum, void *stream);
int dh3d_netvlad_tail_assign_fwd(const float *apart, const float *coarse, const float *asum, int m, const float *W2,
                                 const float *Wh, const float *bn1_scale, const float *bn1_shift, const float *Wg,
                                 const float *bn2_scale, const float *bn2_shift, int B, int D, int Cl, int O, float l2_eps,
                                 void *workspace, size_t workspace_bytes, float *out, void *stream);

/* NetVLAD aggregation + projection + gating in one call (what the model runs): dh3d_netvlad_aggregate_fwd followed by
 * dh3d_netvlad_head_fwd, without the separate whole-vector L2-normalisation kernel (its factor is applied to the
 * projected vector).  Wg may be NULL (no gating).  workspace: dh3d_netvlad_fused_workspace_bytes. */
size_t dh3d_netvlad_fused_workspace_bytes(int B, int N, int D, int Cl, int O);
int dh3d_netvlad_fused_fwd(const float *x, const float *att, const float *wc_packed, const float *bn_scale,
                           const float *bn_shift, const float *W2, const float *Wh, const float *bn1_scale,
                           const float *bn1_shift, const float *Wg, const float *bn2_scale, const float *bn2_shift, int B,
                           int N, int D, int Cl, int O, float l2_eps, void *workspace, size_t workspace_bytes, float *out,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DH3D_HIP_H_ */
