/*
 * srfdet3d.h -- C ABI of libsrfdet3d_hip.so, the MI355X (gfx950) implementation of the SRFDet3D hot path.
 *
 * The reference (gopi-erabati/SRFDet3D) has no native code and no FFI of its own: its Python reaches the
 * operators below through third-party CUDA wheels (mmcv-full 1.7.0, mmdet 2.28.2, mmdet3d 1.0.0rc6,
 * spconv).  Each entry point cites the reference call site whose operator it replaces; the Python binding a
 * maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions:
 *   - plain C types only; every pointer is a DEVICE pointer borrowed from the caller unless marked "host";
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised here;
 *   - the library never allocates, frees or retains memory: outputs and workspaces come from the caller,
 *     sized with the *_workspace_bytes / *_capacity helpers (host-side, pure functions);
 *   - data-dependent sizes (voxel count, active-site count) are written to a device int the caller reads
 *     back when it needs the value on the host;
 *   - return value: 0 on success, negative on error (srf_error_string); no exceptions, no exit();
 *   - re-entrant, no global mutable state.
 */
#ifndef SRFDET3D_H
#define SRFDET3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *srf_stream_t;

#define SRF_OK 0
#define SRF_EINVAL (-1)       /* bad argument (null pointer, non-positive size, unsupported shape) */
#define SRF_EWORKSPACE (-2)   /* workspace smaller than *_workspace_bytes() */
#define SRF_EUNSUPPORTED (-3) /* channel count / kernel size outside the compiled set */
#define SRF_EHIP_BASE (-1000) /* -(1000 + hipError_t) */

int srf_abi_version(void);
/* 0 = the production library; 1 = the developer build (-DSRF_DEV: kernels for timing ablations whose outputs are wrong by
 * design).  A loader must refuse flavour 1 as the product library. */
int srf_build_flavour(void);
const char *srf_error_string(int code);
/* Number of HIP devices visible, or a negative error.  Used by the Python loader to fail loudly. */
int srf_device_count(void);

/* ---- bitmap-rank rulebooks -------------------------------------------------------------------------------------
 * The same indice-pair generation (spconv SubMConv3d / SparseConv3d, sparse_encoder_custom.py:73-107, :123-134, :182-201)
 * for active sets whose rows are kept sorted by (b, y, x, z): one occupancy bit per cell of the level's grid plus an
 * exclusive popcount prefix per 32-bit word give "active? which row?" without a hash table, and the prefix scan of a
 * strided conv's output bitmap emits the new active set already sorted (a canonical order: independent of scheduling).
 * cell = ((b*H + y)*W + x)*D + z.  shape = host {D,H,W}.  bitmap: srf_bitmap_words() uint32; prefix: as many ints. */
size_t srf_bitmap_words(const int *shape, int batch);
/* pair_counts may be NULL (the counts are bookkeeping, no kernel needs them).  ints a pair_counts buffer of the
 * srf_bitmap_* entry points must hold otherwise: the K counts come first, the rest is scratch
 * (replicated counters: hundreds of workgroups adding to one address would serialise in L2) */
size_t srf_bitmap_pair_count_ints(void);
size_t srf_bitmap_workspace_bytes(size_t words);
/* marks `indices` (A x 4 (b,z,y,x), distinct; rows with b < 0 are padding and are skipped by every srf_bitmap_* entry
 * point and by srf_densify), ranks them; order[r] = original row of sorted row r and
 * sorted_indices[r] = its coordinate (both optional, pass NULL for rows that are already sorted) */
int srf_bitmap_build(const int *indices, int A, const int *shape, int batch, void *bitmap, int *prefix, int *order,
                     int *sorted_indices, void *workspace, size_t workspace_bytes, srf_stream_t stream);
/* the same for a capacity-sized set with padding rows: the slots of `order` / `sorted_indices` past the live count are written as
 * 0 / (-1,-1,-1,-1) by the call itself (same launch as the bitmap clear), so the caller passes uninitialised buffers */
int srf_bitmap_build_padded(const int *indices, int A, const int *shape, int batch, void *bitmap, int *prefix, int *order,
                            int *sorted_indices, void *workspace, size_t workspace_bytes, srf_stream_t stream);
int srf_bitmap_rulebook_subm(const int *sorted_indices, int A, const int *shape, int batch, const int *ksize,
                             const void *bitmap, const int *prefix, int *nbr /* K x A */, int *pair_counts /* K */,
                             srf_stream_t stream);
/* phase 1 of a strided conv: bitmap + prefix of the OUTPUT level (sized for the output shape), the sorted output
 * coordinates (at most out_capacity rows are written; use srf_strided_max_outputs) and their number */
int srf_bitmap_strided_outputs(const int *indices, int A, const int *shape, int batch, const int *ksize, const int *stride,
                               const int *pad, void *out_bitmap, int *out_prefix, int *out_indices, int out_capacity,
                               int *num_out, void *workspace, size_t workspace_bytes, srf_stream_t stream);
/* fixed-shape form: ALL out_capacity rows of out_indices are written, those past *num_out as (-1,-1,-1,-1) */
int srf_bitmap_strided_outputs_static(const int *indices, int A, const int *shape, int batch, const int *ksize, const int *stride,
                                      const int *pad, void *out_bitmap, int *out_prefix, int *out_indices, int out_capacity,
                                      int *num_out, void *workspace, size_t workspace_bytes, srf_stream_t stream);
/* phase 2: nbr (K rows of nbr_stride ints) from the INPUT level's bitmap + prefix.  The number of outputs is read on
 * the device (num_out, as written by phase 1), so this launch need not wait for the host; max_out bounds the grid (the
 * out_capacity of phase 1).  Columns >= *num_out are left untouched, or set to -1 when fill_tail != 0 (static-shape
 * levels: rows >= *num_out are padding whose coordinates are -1 and which every kernel here skips).  in_rows = rows of
 * the input level's arrays: ranks beyond it (possible only after a capacity overflow upstream) become -1. */
int srf_bitmap_strided_pairs(const int *out_indices, const int *num_out, int max_out, const int *shape, int batch,
                             const int *ksize, const int *stride, const int *pad, const void *in_bitmap, const int *in_prefix,
                             int in_rows, int *nbr, int nbr_stride, int fill_tail, int *pair_counts, srf_stream_t stream);

/* SparseConvTensor.dense() + the (N, C, D, H, W) -> (N, C * D, H, W) view of sparse_encoder_custom.py:144-147 as ONE pass that writes
 * the BEV map channels-last (out: (B, H, W, C * D) floats, channel c * D + z): every pixel is written once, from the level's bitmap
 * (active cells: the feature row of their rank; others: zeros) -- no zero fill, no scatter, no NCHW -> NHWC transpose.
 * feats (A x C) sorted by (b, y, x, z); bitmap / prefix as built for this level; (C * D) % 4 == 0. */
int srf_densify_bev(const float *feats, int A, int C, const void *bitmap, const int *prefix, int B, int D, int H, int W, float *out,
                    srf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * f-4  the step before the path: the CPU transforms of the reference's test pipeline between the decoded
 * sensor data and SRFDet.forward (configs/nus/srfdet_voxel_nusc_LC.py:253-283), on the device.
 *
 * srf_points_filter: PointsRangeFilter (mmdet3d points.in_range_3d: x > x_min && y > y_min && z > z_min &&
 * x < x_max && y < y_max && z < z_max) and, when close_radius > 0, the remove_close of
 * LoadPointsFromMultiSweeps (points with |x| < r && |y| < r dropped).  pc_range: host[6] or NULL (no range
 * test).  The kept points are copied in their original order to out_points (room for n rows); out_index
 * (may be NULL) receives their source rows; *num_out (device int) their count.
 * workspace: srf_points_filter_workspace_bytes(n).
 *
 * srf_image_prepare: V decoded views (V, H, W, 3) uint8 -> (V, 3, Hp, Wp) float32 = NormalizeMultiviewImage
 * ((x - mean) * (1 / std) per channel in float32, after a BGR<->RGB swap when to_rgb != 0) + PadMultiViewImage
 * (zeros below / right of the image; Hp >= H, Wp >= W, Wp % 4 == 0; the caller rounds up to size_divisor) of
 * mmdet3d_plugin/datasets/pipelines/transform_3d.py:7-93 + the HWC -> CHW transpose and stack of
 * DefaultFormatBundle3D.  mean / std: host[3].
 * ------------------------------------------------------------------------------------------------------- */
size_t srf_points_filter_workspace_bytes(int n);
int srf_points_filter(const float *points, int n, int nf, const float *pc_range /*host[6] or NULL*/,
                      float close_radius, float *out_points, int *out_index, int *num_out, void *workspace,
                      srf_stream_t stream);
int srf_image_prepare(const unsigned char *images, int V, int H, int W, const float *mean /*host[3]*/,
                      const float *std /*host[3]*/, int to_rgb, int Hp, int Wp, float *out, srf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K2  dynamic voxelization.
 * Replaces mmcv.ops.Voxelization(max_num_points=-1).forward as called from SRFDet.voxelize,
 * mmdet3d_plugin/models/detectors/srfdet.py:233-247.
 * coors: (n,3) int32 rows (z,y,x), (-1,-1,-1) for points outside the range.  grid = (gx,gy,gz).
 * ------------------------------------------------------------------------------------------------------- */
int srf_dynamic_voxelize(const float *points, int n, int nf, const float *voxel_size /*host[3]*/,
                         const float *pc_range /*host[6]*/, const int *grid /*host[3] x,y,z*/, int *coors,
                         srf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K1  hard voxelization (+ a3 HardSimpleVFE mean, fused).
 * Replaces mmcv.ops.Voxelization(max_num_points>0, deterministic=True).forward as called from
 * SRFDet.voxelize, srfdet.py:218-232 (module built at srfdet.py:58), and the HardSimpleVFE of
 * configs/nus/srfdet_voxel_nusc_L.py:40.
 * Outputs in first-seen voxel order: voxels (rows x max_points x nf, unused slots zero), coors (rows x 3,
 * z,y,x), num (rows), voxel_num (1 int).  rows = min(n, max_voxels) must be allocated; only the first
 * *voxel_num rows are written.  mean (rows x mean_features) may be NULL.
 * ------------------------------------------------------------------------------------------------------- */
size_t srf_hard_voxelize_workspace_bytes(int n, int max_points);
int srf_hard_voxelize(const float *points, int n, int nf, const float *voxel_size /*host[3]*/,
                      const float *pc_range /*host[6]*/, const int *grid /*host[3]*/, int max_points, int max_voxels,
                      float *voxels, int *coors, int *num, int *voxel_num, float *mean, int mean_features,
                      void *workspace, size_t workspace_bytes, srf_stream_t stream);
/* The fixed-shape form of K1 for a frame replayed from a hipGraph (one sample, n >= 1): ALL rows = min(n, max_voxels) rows of every
 * output are written -- rows >= *voxel_num are padding: zeros, num 0, coordinates -1 -- and the coordinates are (rows x 4)
 * (batch_index, z, y, x), the layout SparseEncoderCustom takes (the F.pad of srfdet.py:228-229 folded in). */
int srf_hard_voxelize_static(const float *points, int n, int nf, const float *voxel_size /*host[3]*/,
                             const float *pc_range /*host[6]*/, const int *grid /*host[3]*/, int max_points, int max_voxels,
                             float *voxels, int *coors4, int *num, int *voxel_num, float *mean, int mean_features, int batch_index,
                             void *workspace, size_t workspace_bytes, srf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K3  DynamicScatter, split in its two halves so that a caller can reuse the point->voxel map.
 * Replaces mmcv.ops.DynamicScatter.forward as called from
 * mmdet3d_plugin/models/voxel_encoders/voxel_encoder.py:189 (mean) and :232 (max), and the dense canvas of
 * map_voxel_center_to_point (:118-158).
 * srf_voxel_unique: coors (n,4) int32 (b,z,y,x); rows with any negative entry are dropped.  Outputs, all
 *   caller-allocated with n rows: out_coors (n x 4, first *num_voxels rows valid, sorted lexicographically),
 *   point2voxel (n; -1 for dropped points), counts (n; points per voxel), offsets (n; start of each voxel's
 *   list in `order`), order (n; point indices grouped by voxel, ascending within a voxel), num_voxels (1 int).
 * srf_scatter_reduce: out (rows x C), rows >= *num_voxels; mode 0 = mean (sum in point order / count),
 *   1 = max.
 * ------------------------------------------------------------------------------------------------------- */
size_t srf_voxel_unique_workspace_bytes(int n, const int *grid_zyx /*host[3] D,H,W*/, int batch);
int srf_voxel_unique(const int *coors, int n, const int *grid_zyx /*host[3]*/, int batch, int *out_coors,
                     int *point2voxel, int *counts, int *offsets, int *order, int *num_voxels, void *workspace,
                     size_t workspace_bytes, srf_stream_t stream);
int srf_scatter_reduce(const float *feats, const int *order, const int *offsets, const int *counts,
                       const int *num_voxels, int rows, int C, int mode, float *out, srf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K4  sparse-conv rulebooks.
 * Replaces spconv's indice-pair generation behind SparseConvTensor / SubMConv3d / SparseConv3d as used by
 * SparseEncoderCustom, mmdet3d_plugin/models/middle_encoders/sparse_encoder_custom.py:123-134 (layers built
 * at :73-107, :182-201).
 * A coordinate table maps (b,z,y,x) -> row.  The rulebook is output-stationary:
 * nbr[k * nbr_stride + o] = input row feeding output row o through kernel offset k, or -1;
 * k = (kz*KH + ky)*KW + kx.  pair_counts[k] = number of valid pairs of offset k.
 * Strided convs number their output rows in first-seen order over (input row, k) ascending.
 * ------------------------------------------------------------------------------------------------------- */
int srf_coord_table_capacity(int max_rows); /* power of two >= 2*max_rows */
size_t srf_coord_table_bytes(int capacity);
int srf_coord_table_build(const int *indices /* A x 4 */, int A, const int *shape /*host[3] D,H,W*/, int batch,
                          void *table, int capacity, srf_stream_t stream);
int srf_rulebook_subm(const int *indices, int A, const int *shape /*host[3]*/, const int *ksize /*host[3]*/,
                      const void *table, int capacity, int *nbr /* K x A */, int *pair_counts /* K */,
                      srf_stream_t stream);
/* bound on the outputs of a strided conv: A * prod(ceil(k/s)), clipped to the output volume */
int srf_strided_max_outputs(int A, int batch, const int *shape, const int *ksize, const int *stride, const int *pad);
size_t srf_rulebook_strided_workspace_bytes(int A, const int *ksize /*host[3]*/, int out_capacity);
/* phase 1: discover and number the outputs; builds the coordinate table of the output level */
int srf_rulebook_strided_outputs(const int *indices, int A, const int *shape, int batch, const int *ksize,
                                 const int *stride, const int *pad, int *out_indices /* max_out x 4 */,
                                 int *num_out /* 1 int */, void *out_table, int out_capacity, void *workspace,
                                 size_t workspace_bytes, srf_stream_t stream);
/* phase 2: fill nbr (K x num_out_host, caller-allocated after reading num_out) from the same workspace */
int srf_rulebook_strided_pairs(int A, const int *ksize, const void *out_table, int out_capacity,
                               const void *workspace, int num_out_host, int *nbr, int *pair_counts,
                               srf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K5  sparse convolution forward with fused eval-BatchNorm / residual / ReLU epilogue.
 * Replaces spconv SubMConv3d/SparseConv3d.forward (+ the BN1d and ReLU modules chained by mmdet3d's
 * make_sparse_convmodule / SparseBasicBlock) on the path sparse_encoder_custom.py:125-134.
 * out[o] = sum_k W[k]^T in[nbr[k][o]] accumulated as an f32 fma chain (k ascending, c ascending);
 * then y = fma(x, alpha, beta) if alpha, y += residual[o] if residual, y = max(y,0) if relu.
 * W: (K, Cin, Cout) row-major.  Supported Cout: 16 (any Cin <= 512) and 32, 64, 128 (Cin a multiple of 4).
 * rows_dev (may be NULL): device int; output rows >= *rows_dev are padding of a static-shape level and are skipped
 * (their tiles return at once), so a hipGraph can launch the layer at its capacity without paying for it.
 * ------------------------------------------------------------------------------------------------------- */
int srf_spconv_fwd(const float *in, int A_in, int Cin, const float *W, int K, const int *nbr, int nbr_stride,
                   int A_out, int Cout, const float *alpha, const float *beta, const float *residual, int relu,
                   float *out, const int *rows_dev, srf_stream_t stream);

/* Gradients of the sparse convolution, for the configurations that train the LiDAR branch (the reference does so in every
 * L-only config: tools/train.py:221-234 freezes it only under `freeze_lidar_components`); they stand where spconv's
 * indice_conv_backward runs.
 * srf_spconv_transpose_rulebook: nbrT (K x A_in) with nbrT[k][i] = o <=> nbr[k][o] = i, -1 elsewhere.
 * srf_spconv_bwd_data: d_in[i] = sum_k W[k] d_out[nbrT[k][i]]; W_T is (K, Cout, Cin) row-major (each offset's matrix
 *   transposed).  It is srf_spconv_fwd with the roles swapped, so Cin must be one of its output widths (16, 32, 64, 128).
 *   A submanifold layer may pass nbrT = its own nbr with W_T's offsets reversed (the rulebook is symmetric).
 * srf_spconv_bwd_weight: d_W[k][ci][co] = sum_o in[nbr[k][o]][ci] d_out[o][co], (K, Cin, Cout); Cin, Cout <= 128.  Row
 *   ranges are combined with float atomics: reproducible to rounding, not bitwise. */
int srf_spconv_transpose_rulebook(const int *nbr, int nbr_stride, int K, int A_out, int *nbrT, int A_in, srf_stream_t stream);
int srf_spconv_bwd_data(const float *grad_out, int A_out, int Cout, const float *W_T, int K, const int *nbrT, int nbrT_stride,
                        int A_in, int Cin, float *grad_in, srf_stream_t stream);
int srf_spconv_bwd_weight(const float *in, int A_in, int Cin, const float *grad_out, int A_out, int Cout, const int *nbr,
                          int nbr_stride, int K, float *grad_W, srf_stream_t stream);

/* Fast path of K5 for constant weights: re-lay W once (srf_spconv_pack_weights -> packed, of
 * srf_spconv_packed_weight_bytes bytes) into the LDS operand image of the kernel, then call srf_spconv_fwd_packed with
 * the same remaining arguments.  Results are bit-identical to srf_spconv_fwd.  Cout in {32, 64, 128}, Cin % 4 == 0. */
size_t srf_spconv_packed_weight_bytes(int K, int Cin, int Cout);
int srf_spconv_pack_weights(const float *W, int K, int Cin, int Cout, float *packed, srf_stream_t stream);
int srf_spconv_fwd_packed(const float *in, int A_in, int Cin, const float *W_packed, int K, const int *nbr,
                          int nbr_stride, int A_out, int Cout, const float *alpha, const float *beta,
                          const float *residual, int relu, float *out, const int *rows_dev, const int *tiles,
                          srf_stream_t stream);

/* Work-balanced row ranges of a rulebook for srf_spconv_fwd_packed (optional, `tiles` may be NULL there): the output
 * rows are cut into srf_spconv_tiles_count(A_out) consecutive ranges of equal cost (a row costs its pairs +
 * srf_spconv_tiles_row_cost(); range t = the rows whose exclusive cost prefix lies in [t*P/T, (t+1)*P/T)),
 * tiles[t]..tiles[t+1], so that the workgroups of the 64- / 128-channel kernels (one range each) finish together
 * although the neighbour count per row varies over the sweep.  Built once per rulebook -- the SubM layers of a level share one (sparse_encoder_custom.py:125-134) --
 * from nbr alone; A_out / nbr_stride / rows_dev as in srf_spconv_fwd, and the same A_out must be passed to both calls.
 * workspace: srf_spconv_tiles_workspace_bytes(A_out) bytes; tiles: srf_spconv_tiles_count(A_out) + 1 ints. */
int srf_spconv_tiles_count(int A_out);
int srf_spconv_tiles_row_cost(void);
size_t srf_spconv_tiles_workspace_bytes(int A_out);
int srf_spconv_tiles_build(const int *nbr, int nbr_stride, int K, int A_out, const int *rows_dev, void *workspace,
                           int *tiles, srf_stream_t stream);
/* Row order for the 32-output-channel layers of srf_spconv_fwd_packed (optional; passed as its `tiles` argument when Cout == 32,
 * Cin in {16, 32}, K == 27): rows with the same set of kernel offsets side by side, so that a wave, which multiplies every offset any of
 * its 32 rows has, skips more of them.  plan: srf_spconv_order_ints(A_out, K) ints = [order: A_pad][sorted rulebook: K x A_pad], A_pad =
 * A_out rounded up to 1024; order[pos] = the row at position pos (-1: padding).  One launch, no workspace; rows_dev as for the
 * convolution.  Outputs are bit-identical with and without a plan (every row is its own fma chain).  No reference counterpart. */
size_t srf_spconv_order_ints(int A_out, int K);
int srf_spconv_order_build(const int *nbr, int nbr_stride, int K, int A_out, const int *rows_dev, int *plan, srf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K6  SparseConvTensor.dense() (+ the view to (B, C*D, H, W), which is a no-op on this layout).
 * Replaces sparse_encoder_custom.py:135-138.  out: (B, C, D, H, W) contiguous; zero_fill != 0 clears it first.
 * ------------------------------------------------------------------------------------------------------- */
int srf_densify(const float *feats, const int *indices, int A, int C, int B, int D, int H, int W, float *out,
                int zero_fill, srf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K7  RoIAlign(avg, aligned) behind mmdet SingleRoIExtractor.
 * Replaces pooler(feats[:4], rois) at mmdet3d_plugin/models/sparse_heads/srfdet_head.py:1685, :2548, :2626
 * (config configs/nus/srfdet_voxel_nusc_LC.py:169-178).
 * Feature maps are described by element strides so that NCHW and channels-last tensors both work.
 * out element (r, c, bin) is written at out[r*out_stride_r + c*out_stride_c + bin*out_stride_bin];
 * accumulate != 0 adds into out instead of overwriting it (camera sum, srfdet_head.py:2561).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct {
    const float *data;
    int N, H, W;
    int64_t stride_n, stride_c, stride_h, stride_w; /* in elements */
    float spatial_scale;                            /* 1 / featmap stride */
} srf_featmap;

int srf_roi_extract(const srf_featmap *levels /*host[num_levels]*/, int num_levels, int C, const float *rois /* R x 5 */,
                    int R, int pooled, int sampling_ratio, float finest_scale, float *out, int64_t out_stride_r,
                    int64_t out_stride_c, int64_t out_stride_bin, int accumulate, int *levels_out /* R or NULL */,
                    srf_stream_t stream);
/* srf_roi_extract_sum: output row r (R rows) = the sum over s = 0 .. n_sum - 1, added in that order, of the gathers of the RoIs
 * rois[s * R + r] (n_sum * R RoIs, group-major): the per-camera image RoIs of a proposal summed over the cameras
 * (srfdet_head.py:2543-2562: pooler over all n_cam * R RoIs, then .sum over the camera axis) in one launch, without the
 * (n_cam * R, S, C) intermediate.  Same sampling arithmetic as srf_roi_extract; out is addressed through the three strides, so it may
 * be a channel slice of the fused (R, S, 2 C) operand of `output_fused_proj` (:2255-2329). */
int srf_roi_extract_sum(const srf_featmap *levels /*host[num_levels]*/, int num_levels, int C, const float *rois /* (n_sum R) x 5 */,
                        int R, int n_sum, int pooled, int sampling_ratio, float finest_scale, float *out, int64_t out_stride_r,
                        int64_t out_stride_c, int64_t out_stride_bin, srf_stream_t stream);


/* Backward of srf_roi_extract with respect to the feature maps (training of the image branch; RoIs carry no
 * gradient, as in mmcv's roi_align_backward).  grad_levels[i] has the layout of levels[i] and must be zeroed by
 * the caller; grad_out has the strides given. */
int srf_roi_extract_bwd(const srf_featmap *levels /*host*/, float *const *grad_levels /*host array of device ptrs*/,
                        int num_levels, int C, const float *rois, int R, int pooled, int sampling_ratio,
                        float finest_scale, const float *grad_out, int64_t out_stride_r, int64_t out_stride_c,
                        int64_t out_stride_bin, srf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * a11-a13  proposal box -> RoI geometry, fused.
 * Replaces the torch op chains of points_feats_sampling_bboxes_roi (srfdet_head.py:1638-1683 / :2579-2624)
 * and img_feats_sampling_bboxes_roi (:2435-2528), including boxes3d_to_corners3d
 * (mmdet3d_plugin/core/bbox/util.py:84-176).
 * boxes: (B, P, box_dim >= 8).  mutate_centres != 0 overwrites boxes[..., :3] with metres in place, as
 * srfdet_head.py:1646 does.  rois_bev: (B*P, 5).  rois_img: (n_cam*B*P, 5), cam-major, batch id b + cam*B
 * (the reference's indexing, :2520-2528); lidar2img: (B, n_cam, 4, 4) row-major.  Either output may be NULL.
 * ------------------------------------------------------------------------------------------------------- */
int srf_box_rois(float *boxes, int B, int P, int box_dim, const float *pc_range /*host[6]*/,
                 const float *voxel_size /*host[3]*/, int mutate_centres, float *rois_bev, const float *lidar2img,
                 int n_cam, float *rois_img, srf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * a14-a17  dense arithmetic of a decoder stage (SingleSRFDetHeadLiDAR.forward / SingleSRFDetHead.forward,
 * srfdet_head.py:1484-1525 == :2281-2322; DynamicConv :2633-2693; fused projection :2255-2264).
 *
 * srf_linear: Y (M x N, row stride ldy) = chain(X (M x K, ldx) . W^T (W is N x K, ldw) + bias) where chain is
 *   [LayerNorm(ln1, eps1)] -> [ReLU if relu1] -> [+ residual (M x N, ldr)] -> [LayerNorm(ln2, eps2)] -> [ReLU if relu2];
 *   null pointers skip a step.  Replaces nn.Linear + nn.LayerNorm + ReLU (+ the residual adds) call chains, e.g.
 *   out_layer -> norm3 -> ReLU -> (+prop) -> norm2 (srfdet_head.py:2689-2691, :1502-1503).  K % 4 == 0; K >= 2048 is
 *   split over K and needs the workspace (srf_linear_workspace_bytes); LayerNorm steps need N <= 1024.
 * srf_self_attention: qkv (P x 3E) = [q | k | v] rows after in_proj -> out (P x E), heads H, head dim E/H in {16, 32};
 *   replaces the attention core of nn.MultiheadAttention(batch 1) at srfdet_head.py:1489.
 * srf_dynconv_mid: feats (R x S x C) bin-major RoI features, params (R x 2*C*D) from dynamic_layer ->
 *   relu(LN_C(relu(LN_D(feats W1)) W2)) (R x S x C); srfdet_head.py:2671-2686.  (C, D) in {(128,32), (256,64)}, S <= 64.
 * srf_apply_deltas: srfdet_head.py:1534-1625; weights6 / pc_range are host arrays.
 * ------------------------------------------------------------------------------------------------------- */
size_t srf_linear_workspace_bytes(int M, int N, int K);
int srf_linear(const float *X, int M, int K, int ldx, const float *W, int N, int ldw, const float *bias,
               const float *ln1_g, const float *ln1_b, float eps1, int relu1, const float *residual, int ldr,
               const float *ln2_g, const float *ln2_b, float eps2, int relu2, float *Y, int ldy, void *workspace,
               size_t workspace_bytes, srf_stream_t stream);
int srf_self_attention(const float *qkv, int P, int E, int H, float *out, srf_stream_t stream);
/* the same for B samples of P proposals each (rows sample-major): one launch instead of the per-sample loop of a batched head */
int srf_self_attention_batched(const float *qkv, int B, int P, int E, int H, float *out, srf_stream_t stream);
int srf_dynconv_mid(const float *feats, const float *params, int R, int S, int C, int D, const float *g1,
                    const float *b1, float eps1, const float *g2, const float *b2, float eps2, float *out,
                    srf_stream_t stream);
/* srf_channel_affine: y[n][c][:] = x[n][c][:] * scale + shift (+ residual[n][c][:]), optionally clamped at 0, in one
 * pass.  per_sample == 0: scale/shift are [C] -- eval-mode BatchNorm2d (+ ReLU) after the dense convolutions of
 * SECONDCustom / FPN / VoVNet (second_custom.py:41-63, vovnet.py:116-153).  per_sample == 1: scale/shift are [N*C] -- the
 * eSE channel gate, with the OSA block's identity input as `residual` (vovnet.py:165-177, :225-228).  shift and
 * residual may be NULL.  x, y: NCHW f32 with plane size HW and their own batch strides (in floats), so y may be a
 * channel slice of a wider tensor; y == x is allowed; residual is contiguous (N, C, HW). */
int srf_channel_affine(const float *x, int N, int C, int HW, long long x_batch_stride, const float *scale,
                       const float *shift, int per_sample, const float *residual, int relu, float *y,
                       long long y_batch_stride, srf_stream_t stream);

/* srf_upsample_add: out = lateral + nearest_upsample(top), the top-down step of mmdet's FPN (`laterals[i-1] +=
 * F.interpolate(laterals[i], size=laterals[i-1].shape[2:], mode='nearest')`; the necks of configs/nus/srfdet_voxel_nusc_LC.py)
 * in one pass.  lateral, out: (NC, H, W) f32 contiguous (out == lateral allowed), top: (NC, Ht, Wt); source index
 * floor(dst * Ht / H) as F.interpolate.  W % 4 == 0, else SRF_EUNSUPPORTED. */
int srf_upsample_add(const float *lateral, const float *top, int NC, int H, int W, int Ht, int Wt, float *out,
                     srf_stream_t stream);

/* srf_dwconv3x3s2: depthwise 3x3 convolution, stride 2, padding 1 (w: (C, 9), no bias) followed by y = y * scale[c] +
 * shift[c] (eval BatchNorm2d; either may be NULL) and an optional ReLU: the stair of the dynamic proposal generator,
 * ConvModule(C, C, 3, stride=2, padding=1, groups=C, norm BN2d) built at srfdet_head.py:265-320 and run at :525-536.
 * x: (N, C, H, W) f32 contiguous -> y: (N, C, (H - 1) / 2 + 1, (W - 1) / 2 + 1). */
int srf_dwconv3x3s2(const float *x, int N, int C, int H, int W, const float *w, const float *scale, const float *shift,
                    int relu, float *y, srf_stream_t stream);

/* srf_nchw_to_nhwc: y[n][p][c] = x[n][c][p] (p = h * W + w): the channels-last copy of a feature level that
 * srf_roi_extract gathers from (the reference feeds NCHW levels to mmcv RoIAlign, srfdet_head.py:1685 / :2548; this
 * library's gather wants one contiguous C-run per tap).  HW % 4 == 0, else SRF_EUNSUPPORTED. */
int srf_nchw_to_nhwc(const float *x, int N, int C, int HW, float *y, srf_stream_t stream);

/* srf_ese_gate: gate[n][c] = relu6(sum_k W[c][k] * mean[n][k] + bias[c] + 3) / 6 -- the channel gate of VoVNet's eSE
 * module (mmdet3d_plugin/models/backbones/vovnet.py, eSEModule: Hsigmoid(fc(avg_pool(x)))) on the (N, C) global averages;
 * W: (C, C) row-major = the 1x1 conv weight, bias (C) or NULL.  N <= 8, C % 4 == 0. */
int srf_ese_gate(const float *mean, int N, int C, const float *W, const float *bias, float *gate, srf_stream_t stream);

/* srf_maxpool3s2_ceil: nn.MaxPool2d(kernel_size=3, stride=2, ceil_mode=True) on (NC, H, W) f32 -> (NC, Ho, Wo),
 * Ho = ceil((H - 3) / 2) + 1 (the last window must start inside the input), likewise Wo: the stage pooling of the
 * VoVNet image backbone (mmdet3d_plugin/models/backbones/vovnet.py, `_make_layer` of every stage but the first). */
int srf_maxpool3s2_ceil(const float *x, int NC, int H, int W, float *y, srf_stream_t stream);

/* srf_conv1x1: 1x1 convolution over the channel concatenation of n_src (<= 8) NCHW f32 tensors of the same N and
 * H*W, followed by y = y * scale[co] + shift[co] (scale may be NULL: bias only; both NULL: none) and an optional ReLU --
 * the `concat` layer of VoVNet's OSA blocks (vovnet.py:180-230: torch.cat + conv1x1 + BN + ReLU) without the
 * concatenated tensor, and the FPN lateral convolutions.  srcs / src_channels are HOST arrays (device pointers,
 * channel counts, each a multiple of 32); W_packed comes from srf_conv1x1_pack_weights(W (Cout x K row-major, K = sum
 * of the channel counts in concat order)); Cout % 128 == 0 and HW % 4 == 0, else SRF_EUNSUPPORTED.  out: (N, Cout, HW). */
size_t srf_conv1x1_packed_weight_bytes(int Cout, int K);
int srf_conv1x1_pack_weights(const float *W, int Cout, int K, float *packed, srf_stream_t stream);
int srf_conv1x1(const float *const *srcs, const int *src_channels, int n_src, int N, int HW, const float *W_packed, int Cout,
                const float *scale, const float *shift, int relu, float *out, srf_stream_t stream);

/* ---- channels-last (NHWC) dense convolutions on the f32 MFMA (csrc/conv.hip) --------------------------------------
 * Activations are (N, H, W, ld) f32, `ld` floats per pixel >= the channels used: x / y point at the first channel of
 * the slice a layer reads / writes inside its buffer, so the OSA concatenation of VoVNet (vovnet.py:222) is a set
 * of slices of one buffer and never a copy.
 *
 * srf_wino3x3: Conv2d(Cin, Cout, 3, stride 1, padding 1, bias folded into shift) as Winograd F(2x2, 3x3), followed by
 * y = y * scale[co] + shift[co] (either may be NULL) and an optional ReLU: the 3x3 layers of VoVNet's OSA blocks
 * (vovnet.py:116-133, :180-230), the image FPN outputs, `img_convs` (srfdet_head.py:404-416), SECONDCustom
 * (second_custom.py:41-63) and the BEV FPN.  U_packed comes from srf_wino3x3_pack_weights(W (Cout, Cin, 3, 3)).
 * Cin % 8 == 0, x 16-byte aligned, x_ld % 4 == 0, 4 H W x_ld < 2^30, else SRF_EUNSUPPORTED. */
size_t srf_wino3x3_packed_weight_bytes(int Cout, int Cin);
int srf_wino3x3_pack_weights(const float *W, int Cout, int Cin, float *packed, srf_stream_t stream);
int srf_wino3x3(const float *x, int N, int H, int W, int Cin, long long x_ld, const float *U_packed, int Cout,
                const float *scale, const float *shift, int relu, float *y, long long y_ld, srf_stream_t stream);

/* srf_wino43: the same layer as srf_wino3x3 as Winograd F(4x4, 3x3) (csrc/wino43.hip): 36 products per 16 outputs instead of
 * 16 per 4, i.e. direct FLOPs / 4 on the MFMA.  Two kernels per call: the input transform V = B^T d B of every 6 x 6 patch into
 * `workspace` (srf_wino43_workspace_bytes), then the 36 batched products with the output transform, scale / shift / ReLU as the
 * epilogue.  U_packed comes from srf_wino43_pack_weights(W (Cout, Cin, 3, 3)) (U = G g G^T computed in double, rounded once).
 * Replaces the same reference call sites as srf_wino3x3 (vovnet.py:116-133, :180-216; srfdet_head.py:404-416;
 * second_custom.py:41-63).  f32 result within ~2e-5 of the map's maximum of the direct convolution (Cin <= 1024).
 * Cin % 8 == 0, Cout % 4 == 0, x / y / scale / shift 16-byte aligned, x_ld % 4 == 0, y_ld % 4 == 0, else SRF_EUNSUPPORTED. */
size_t srf_wino43_packed_weight_bytes(int Cout, int Cin);
int srf_wino43_pack_weights(const float *W, int Cout, int Cin, float *packed, srf_stream_t stream);
size_t srf_wino43_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int srf_wino43(const float *x, int N, int H, int W, int Cin, long long x_ld, const float *U_packed, int Cout, const float *scale,
               const float *shift, int relu, float *y, long long y_ld, void *workspace, size_t workspace_bytes, srf_stream_t stream);

/* The two kernels of srf_wino43 as separate calls, for callers that time them apart (bench.py's roofline: the transform is
 * HBM-bound, the multiply MFMA-bound) or order other work between them.  Same arguments and limits; SRF_EUNSUPPORTED when the
 * layer's transformed input exceeds one workspace slab (srf_wino43 then runs it in several). */
int srf_wino43_transform(const float *x, int N, int H, int W, int Cin, long long x_ld, int Cout, void *workspace, size_t workspace_bytes,
                         srf_stream_t stream);
int srf_wino43_multiply(const void *workspace, size_t workspace_bytes, int N, int H, int W, int Cin, const float *U_packed, int Cout,
                        const float *scale, const float *shift, int relu, float *y, long long y_ld, srf_stream_t stream);

/* srf_conv1x1_nhwc: Conv2d(K, Cout, 1) on channels-last activations = the GEMM y[p][co] = sum_k x[p][k] W[co][k] over
 * M = N * H * W pixels, then y = y * scale[co] + shift[co] (either may be NULL) and an optional ReLU: the `concat` layer of
 * VoVNet's OSA blocks read straight from the block's concat buffer (vovnet.py:222-223) and the FPN lateral convolutions.
 * W_packed comes from srf_conv1x1_nhwc_pack_weights(W (Cout, K) row-major).  K % 32 == 0, x 16-byte aligned, x_ld % 4 == 0,
 * else SRF_EUNSUPPORTED. */
size_t srf_conv1x1_nhwc_packed_weight_bytes(int Cout, int K);
int srf_conv1x1_nhwc_pack_weights(const float *W, int Cout, int K, float *packed, srf_stream_t stream);
int srf_conv1x1_nhwc(const float *x, long long M, int K, long long x_ld, const float *W_packed, int Cout, const float *scale,
                     const float *shift, int relu, float *y, long long y_ld, srf_stream_t stream);
/* srf_conv1x1_nhwc_direct*: the same three operations (plain, _topdown, _pooled) on a GEMM kernel whose operands go L2 ->
 * registers without LDS staging or barriers (csrc/gemm_direct.hip; 128 x 128 tiles only: meant for launches of >= ~1000 tiles,
 * where it runs at 122-135 TFLOP/s against 110-117 for srf_conv1x1_nhwc).  W_packed comes from
 * srf_conv1x1_nhwc_direct_pack_weights (a different operand order than srf_conv1x1_nhwc_pack_weights).  Same arguments, limits
 * and -- output for output -- the same fma chain, i.e. identical bits in y; the pooled mean adds its block sums in a different
 * (fixed) order.  Workspace of the pooled form: srf_conv1x1_nhwc_pooled_workspace_bytes. */
size_t srf_conv1x1_nhwc_direct_packed_weight_bytes(int Cout, int K);
int srf_conv1x1_nhwc_direct_pack_weights(const float *W, int Cout, int K, float *packed, srf_stream_t stream);
int srf_conv1x1_nhwc_direct(const float *x, long long M, int K, long long x_ld, const float *W_packed, int Cout, const float *scale,
                            const float *shift, int relu, float *y, long long y_ld, srf_stream_t stream);
int srf_conv1x1_nhwc_direct_topdown(const float *x, int N, int H, int W, int K, long long x_ld, const float *W_packed, int Cout,
                                    const float *scale, const float *shift, int relu, const float *top, int Ht, int Wt, long long top_ld,
                                    float *y, long long y_ld, srf_stream_t stream);
int srf_conv1x1_nhwc_direct_pooled(const float *x, int N, long long HW, int K, long long x_ld, const float *W_packed, int Cout,
                                   const float *scale, const float *shift, int relu, float *y, long long y_ld, float *mean, void *workspace,
                                   size_t workspace_bytes, srf_stream_t stream);
/* srf_conv1x1_nhwc_split*: the same three operations as an f32 GEMM computed on the bf16 MFMA (csrc/gemm_split.hip).  Every f32
 * operand is the exact sum of three bf16 values (x = xh + xm + xl); of the nine exact partial products of a b the six of relative
 * magnitude >= 2^-16 are accumulated in f32 on v_mfma_f32_32x32x16_bf16, the three dropped ones are together below 2^-23 |a b| (one
 * f32 rounding of the product).  Error against float64 = that of the f32 fma chain (measured 2.0-2.8e-7 of sum |a b| against 1.7-3.0e-7);
 * NOT bit-identical to srf_conv1x1_nhwc (another summation order), deterministic.  189-199 TFLOP/s f32-equivalent where the f32 MFMA
 * kernels reach 122-135.  W_packed (bf16 planes, 6 bytes per weight) comes from srf_conv1x1_nhwc_split_pack_weights; x stays f32 and
 * is split while it is staged.  Same arguments and limits as the _direct forms; workspace of the pooled form:
 * srf_conv1x1_nhwc_pooled_workspace_bytes. */
size_t srf_conv1x1_nhwc_split_packed_weight_bytes(int Cout, int K);
int srf_conv1x1_nhwc_split_pack_weights(const float *W, int Cout, int K, void *packed, srf_stream_t stream);
int srf_conv1x1_nhwc_split(const float *x, long long M, int K, long long x_ld, const void *W_packed, int Cout, const float *scale,
                           const float *shift, int relu, float *y, long long y_ld, srf_stream_t stream);
int srf_conv1x1_nhwc_split_topdown(const float *x, int N, int H, int W, int K, long long x_ld, const void *W_packed, int Cout,
                                   const float *scale, const float *shift, int relu, const float *top, int Ht, int Wt, long long top_ld,
                                   float *y, long long y_ld, srf_stream_t stream);
int srf_conv1x1_nhwc_split_pooled(const float *x, int N, long long HW, int K, long long x_ld, const void *W_packed, int Cout,
                                  const float *scale, const float *shift, int relu, float *y, long long y_ld, float *mean,
                                  void *workspace, size_t workspace_bytes, srf_stream_t stream);
/* srf_conv_gemm_nhwc_split: srf_conv_gemm_nhwc (Conv2d with stride / padding as an implicit-im2col GEMM: VoVNet stem_3, SECONDCustom's
 * stride-2 layers, the BEV FPN extras) on the split GEMM kernel.  W_packed = srf_conv1x1_nhwc_split_pack_weights of the weight reordered
 * to (Cout, kh * kw * Cin), tap index slowest.  Cin % 32 == 0, the input below 2 GB, else SRF_EUNSUPPORTED. */
int srf_conv_gemm_nhwc_split(const float *x, int N, int H, int W, int Cin, long long x_ld, const void *W_packed, int Cout, int kh, int kw,
                             int stride, int pad, const float *scale, const float *shift, int relu, float *y, long long y_ld,
                             srf_stream_t stream);



/* srf_conv_wgrad_nhwc (training, config 4): the WEIGHT gradient of a stride-1 Conv2d on channels-last tensors,
 *   dW[co][ci][ky][kx] = sum over (n, y, x) of g[n][y][x][co] * x[n][y + ky - pad][x + kx - pad][ci],
 * ksize 1 (padding 0) or 3 (padding 1) -- what torch.autograd computes through aten::convolution_backward for the trainable layers of
 * tools/train.py:220-234 (VoVNet stages 4-5, vovnet.py:354-374; the image FPN; the head's img_convs, srfdet_head.py:404-416), there on
 * MIOpen's float-atomic split-K kernels.  Here: an f32 GEMM over the pixels on the bf16 MFMA through the exact three-way split of both
 * operands (csrc/wgrad.hip), the pixel ranges added in a fixed order: deterministic.  g: (N, H, W, g_ld) with Cout channels, x:
 * (N, H, W, x_ld) with Cin channels, dW: (Cout, Cin, ksize, ksize) contiguous; workspace: srf_conv_wgrad_workspace_bytes bytes.
 * Cin % 4 == Cout % 4 == 0, W >= 32, tensors below 2 GB, 16-byte aligned bases, g_ld % 4 == x_ld % 4 == 0; else SRF_EUNSUPPORTED. */
size_t srf_conv_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int ksize);
int srf_conv_wgrad_nhwc(const float *g, long long g_ld, const float *x, long long x_ld, int N, int H, int W, int Cin, int Cout, int ksize,
                        void *workspace, size_t workspace_bytes, float *dW, srf_stream_t stream);

/* srf_conv1x1_nhwc_topdown: an FPN lateral convolution with the top-down step in its epilogue (mmdet FPN.forward:
 * `laterals[i - 1] += F.interpolate(laterals[i], size=..., mode="nearest")`, necks of configs/nus/srfdet_voxel_nusc_LC.py:55-64
 * and :67-76): y[n][py][px][co] = act(conv) + top[n][floor(py Ht / H)][floor(px Wt / W)][co].  x rows are the pixels of an
 * (N, H, W) map; top is an (N, Ht, Wt, >= Cout) channels-last slice with top_ld floats per pixel.  Same bits as
 * srf_conv1x1_nhwc followed by srf_nhwc_upsample_add. */
int srf_conv1x1_nhwc_topdown(const float *x, int N, int H, int W, int K, long long x_ld, const float *W_packed, int Cout,
                             const float *scale, const float *shift, int relu, const float *top, int Ht, int Wt, long long top_ld,
                             float *y, long long y_ld, srf_stream_t stream);
/* srf_conv1x1_nhwc_pooled: the same convolution on N images of HW pixels (rows n HW .. (n + 1) HW - 1), which also returns
 * mean[n][co] = the mean over the image's pixels of the stored outputs: VoVNet's eSE average pool (reference
 * mmdet3d_plugin/models/backbones/vovnet.py:165-177 eSEModule.avg_pool on the output of vovnet.py:223 `concat`) without a
 * second pass over the map.  Deterministic (per-block sums added in a fixed order).  workspace: _workspace_bytes(N, HW, Cout). */
size_t srf_conv1x1_nhwc_pooled_workspace_bytes(int N, long long HW, int Cout);
int srf_conv1x1_nhwc_pooled(const float *x, int N, long long HW, int K, long long x_ld, const float *W_packed, int Cout,
                            const float *scale, const float *shift, int relu, float *y, long long y_ld, float *mean, void *workspace,
                            size_t workspace_bytes, srf_stream_t stream);

/* srf_conv_gemm_nhwc: Conv2d(Cin, Cout, (kh, kw), stride, padding) on channels-last activations as the GEMM of
 * srf_conv1x1_nhwc with an implicit im2col (rows = output pixels, k = (tap, input channel)): the stride-2 3x3 layers of the
 * path (VoVNet stem_3, vovnet.py stem; the first layer of SECONDCustom's later blocks, second_custom.py:41-52; the extra levels
 * of the BEV FPN) -- deterministic k-ordered fma chains where MIOpen's channels-last choice is an atomic split-K kernel.
 * W_packed = srf_conv1x1_nhwc_pack_weights(W reordered to (Cout, kh * kw * Cin), tap slowest).  Cin % 32 == 0, the input
 * tensor below 2 GiB.  y: (N, Ho, Wo, y_ld), Ho = (H + 2 pad - kh) / stride + 1.
 * srf_stem_conv_nchw: Conv2d(Cin <= 4, 64, 3, stride 2, padding 1) + scale / shift + ReLU from NCHW images to channels-last
 * (VoVNet stem_1 on the 6 camera views); Wt is the (64, Cin, 3, 3) weight. */
int srf_conv_gemm_nhwc(const float *x, int N, int H, int W, int Cin, long long x_ld, const float *W_packed, int Cout, int kh,
                       int kw, int stride, int pad, const float *scale, const float *shift, int relu, float *y, long long y_ld,
                       srf_stream_t stream);
int srf_stem_conv_nchw(const float *x, int N, int Cin, int H, int W, const float *Wt, int Cout, const float *scale,
                       const float *shift, int relu, float *y, long long y_ld, srf_stream_t stream);

/* ---- streaming layers of the channels-last camera branch (csrc/nhwc.hip); all take (N, H, W, ld) channel slices, C % 4 == 0,
 * 16-byte aligned pointers, ld % 4 == 0 ---------------------------------------------------------------------------------
 * srf_nhwc_affine: y = x * scale[(per_sample & 1 ? n : 0)][c] + shift[(per_sample & 2 ? n : 0)][c] (+ residual), optional ReLU; scale / shift / residual
 *   may be NULL; in place allowed: eval BatchNorm2d + ReLU behind a library convolution, and the eSE gate multiply + OSA
 *   identity add of VoVNet (vovnet.py:165-177, :225-228).  HW = pixels per sample.
 * srf_nhwc_colmean: mean[n][c] over the HW pixels (AdaptiveAvgPool2d(1) of the eSE module), deterministic two-level sum;
 *   C <= 1024.
 * srf_nhwc_maxpool3s2_ceil: MaxPool2d(3, stride 2, ceil_mode=True) (the VoVNet stage pooling) -> (N, Ho, Wo, C).
 * srf_nhwc_upsample_add: y = lat + nearest-upsampled top (the FPN top-down step; F.interpolate 'nearest' index rule).
 * srf_nhwc_dwconv3x3s2: depthwise Conv2d(C, C, 3, stride 2, padding 1, groups=C) + scale / shift (+ ReLU): the stair of the
 *   proposal generator on the camera levels (srfdet_head.py:265-320, :525-536); w is (C, 3, 3). */
int srf_nhwc_affine(const float *x, long long x_ld, int N, long long HW, int C, const float *scale, int per_sample,
                    const float *shift, const float *residual, long long r_ld, int relu, float *y, long long y_ld,
                    srf_stream_t stream);
size_t srf_nhwc_colmean_workspace_bytes(int N, int C);
int srf_nhwc_colmean(const float *x, long long x_ld, int N, long long HW, int C, float *mean, void *workspace,
                     size_t workspace_bytes, srf_stream_t stream);
/* out[n][c] = sum over the pixels of a[n][p][c] * b[n][p][c] (training: the pixel sum behind the gradient of the eSE gate, vovnet.py:165-177);
 * deterministic; workspace: srf_nhwc_colmean_workspace_bytes(N, C). */
int srf_nhwc_colsum_prod(const float *a, long long a_ld, const float *b, long long b_ld, int N, long long HW, int C, float *out /*N x C*/,
                         void *workspace, size_t workspace_bytes, srf_stream_t stream);
int srf_nhwc_maxpool3s2_ceil(const float *x, long long x_ld, int N, int H, int W, int C, float *y, long long y_ld,
                             srf_stream_t stream);
int srf_nhwc_upsample_add(const float *lat, long long l_ld, const float *top, long long t_ld, int N, int H, int W, int Ht, int Wt,
                          int C, float *y, long long y_ld, srf_stream_t stream);
int srf_nhwc_dwconv3x3s2(const float *x, long long x_ld, int N, int H, int W, int C, const float *w, const float *scale,
                         const float *shift, int relu, float *y, long long y_ld, srf_stream_t stream);
/* The proposal generator (srfdet_head.py:496-561) without its ~35 small torch launches:
 * srf_nhwc_dwconv3x3s2_cat: one step of the stair (:525-536), x = cat([level, conv(x)], 1): the convolution writes `y` and the
 *   pyramid level `side` (N, Ho, Wo, Cs) is copied to `side_out`, both slices of one concat buffer (row stride y_ld).
 * srf_nhwc_pool_sum: out[b][yo * Wo + xo] = sum over the n_cam images of sample b and the C channels of x at the `nearest`
 *   source pixel of (yo, xo) (F.interpolate to (Ho, Wo) + the camera sum + the channel sum, :537 / :548-552; Ho = H, Wo = W,
 *   n_cam = 1 for the BEV pyramid); out rows have out_ld >= Ho * Wo floats, the tail zero-filled (K padding of fc1).
 * srf_dpg_mix: softmax over the E experts of the logits wl (B, E * P) (with wi: of (wl + wi) / 2), the expert-weighted sums of
 *   the proposal boxes (E * P, D) and features (E * P, C), and the sigmoid of the box centres (:957) -> boxes (B, P, D),
 *   feats (B, P, C). */
int srf_nhwc_dwconv3x3s2_cat(const float *x, long long x_ld, int N, int H, int W, int C, const float *w, const float *scale,
                             const float *shift, int relu, float *y, long long y_ld, const float *side, long long side_ld, int Cs,
                             float *side_out, srf_stream_t stream);
/* Training (config 4, tools/train.py:220-234): the backward pass of y = relu(z * s[c] + t[c]) -- a convolution's eval-mode BatchNorm
 * (norm_eval=True, vovnet.py:371) + ReLU, which the forward pass runs as the convolution's epilogue -- over channels-last (M x C)
 * tensors in one pass: gz = (relu ? gy * [y > 0] : gy) * s, sums[0..C) = sum_p gu, sums[C..2C) = sum_p gu * y (gu = the masked gy);
 * deterministic (per-block partials added in block order).  C % 4 == 0, C <= 1024. */
size_t srf_nhwc_affine_relu_bwd_workspace_bytes(long long M, int C);
int srf_nhwc_affine_relu_bwd(const float *gy, long long gy_ld, const float *y, long long y_ld, long long M, int C, const float *scale /*or NULL*/,
                             int relu, float *gz, long long gz_ld, float *sums /*2 C*/, void *workspace, size_t workspace_bytes,
                             srf_stream_t stream);
/* ... with a second gradient of the same output, gy2 (M x C, row stride gy2_ld; NULL: none): gy + gy2 takes the place of gy (one rounding,
 * the add autograd makes as a pass of its own when an output has two consumers -- an OSA layer feeds the next layer AND the block's
 * concat convolution, vovnet.py:208-230). */
int srf_nhwc_affine_relu_bwd2(const float *gy, long long gy_ld, const float *gy2 /*or NULL*/, long long gy2_ld, const float *y, long long y_ld,
                              long long M, int C, const float *scale /*or NULL*/, int relu, float *gz, long long gz_ld, float *sums /*2 C*/,
                              void *workspace, size_t workspace_bytes, srf_stream_t stream);
/* The per-channel arithmetic around an eval-mode BatchNorm2d under training (norm_eval=True, vovnet.py:371; tools/train.py:220-234), C values:
 * srf_bn_eval_fold: out[0..C) = s = gamma / sqrt(var + eps), out[C..2C) = t0 = beta - mean s, out[2C..3C) = inv = 1 / sqrt(var + eps);
 * srf_bn_eval_grads: from sums = srf_nhwc_affine_relu_bwd's [sum gu, sum gu y] and fold = srf_bn_eval_fold's output:
 *   out[0..C) = d gamma = ((s != 0 ? (sum gu y - t0 sum gu) / s : 0) - mean sum gu) inv,  out[C..2C) = d beta = sum gu. */
int srf_bn_eval_fold(const float *gamma, const float *beta, const float *mean, const float *var, float eps, int C, float *out /*3 C*/,
                     srf_stream_t stream);
int srf_bn_eval_grads(const float *sums /*2 C*/, const float *fold /*3 C*/, const float *mean, int C, float *out /*2 C*/, srf_stream_t stream);
/* srf_ese_apply: the end of VoVNet's eSEModule (vovnet.py:165-177) applied to an OSA block's concat output (:225-228) in one launch:
 * gate[n][c] = hsigmoid(fc(mean[n])) (the bits of srf_ese_gate) and y = x * gate (+ residual = the block's identity input), the bits of
 * srf_nhwc_affine with a per-sample scale.  x / residual / y: (N * HW) pixel rows of x_ld / r_ld / y_ld floats; W (C x C), bias (C) the
 * fc = 1x1 conv; gate_out (N x C) optional.  C % 64 == 0, C <= 1024. */
int srf_ese_apply(const float *x, long long x_ld, int N, long long HW, int C, const float *mean, const float *W, const float *bias,
                  const float *residual /*or NULL*/, long long r_ld, float *y, long long y_ld, float *gate_out /*or NULL*/,
                  srf_stream_t stream);
int srf_nhwc_pool_sum(const float *x, long long x_ld, int B, int n_cam, int H, int W, int C, int Ho, int Wo, float *out, int out_ld,
                      srf_stream_t stream);
int srf_dpg_mix(const float *wl, const float *wi /*or NULL*/, int B, int E, int P, const float *boxes_w, int D, const float *feats_w,
                int C, float *boxes, float *feats, srf_stream_t stream);

/* srf_stage_tail: the row-local remainder of a stage in one launch (srfdet_head.py:1506-1520): FFN + residual + norm3,
 * classification tower + class_logits, regression tower + bboxes_delta + apply_deltas.  obj_in (R x C) is norm2's
 * output; outputs obj_out (R x C), logits (R x ncls), pred (R x Dd).  cls_/reg_ arrays are HOST arrays of n_cls / n_reg
 * device pointers (weights (C x C), LayerNorm gamma/beta) and host eps values.  C == 128, F in {128,...,512}.
 * workspace (srf_stage_tail_workspace_bytes(R, C, F), may be NULL): with it the FFN runs as its own launch, one workgroup per
 * (32 rows, 128 hidden units), and the tail adds the slices in a fixed order (deterministic; the sum of F / 128 partial chains
 * instead of one chain over F, i.e. equal to the in-line FFN to rounding, not bit for bit); without it everything is one launch. */
size_t srf_stage_tail_workspace_bytes(int R, int C, int F);
int srf_stage_tail(const float *obj_in, int R, int C, int F, const float *w1, const float *b1, const float *w2,
                   const float *b2, const float *n3_g, const float *n3_b, float n3_eps, int n_cls,
                   const float *const *cls_w, const float *const *cls_g, const float *const *cls_b, const float *cls_eps,
                   int n_reg, const float *const *reg_w, const float *const *reg_g, const float *const *reg_b,
                   const float *reg_eps, const float *wl, const float *bl, int ncls, const float *wd, const float *bd,
                   int Dd, const float *boxes_m, const float *weights6, const float *pc_range, float scale_clamp,
                   float *obj_out, float *logits, float *pred, void *workspace, size_t workspace_bytes, srf_stream_t stream);
int srf_apply_deltas(const float *deltas, const float *boxes, int R, int Dd, const float *weights6 /*host[6]*/,
                     const float *pc_range /*host[6]*/, float scale_clamp, float *out, srf_stream_t stream);
/* srf_decode_boxes: the tensors the reference hands to box3d_multiclass_nms (srfdet_head.py:1246-1271, after the
 * centres-to-metres step that ends `forward`, :1002-1006) from the last stage's outputs in one launch: logits (R, ncls) ->
 * scores = sigmoid; pred (R, Dd) [centres normalised to pc_range, log sizes, sin, cos, (vx, vy)] -> boxes (R, Dd - 1)
 * [x, y, bottom-centre z in metres, w, l, h, yaw, (vx, vy)]. */
int srf_decode_boxes(const float *logits, const float *pred, int R, int ncls, int Dd, const float *pc_range /*host[6]*/,
                     float *scores, float *boxes, srf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K8  rotated BEV NMS.
 * Replaces mmcv nms_rotated behind mmdet3d box3d_multiclass_nms, called at srfdet_head.py:1288-1293.
 * boxes: (n,5) [cx, cy, w, h, angle(rad)] already sorted by descending score; keep[i] = 1 if box i survives
 * greedy suppression at IoU > iou_threshold.  n <= 4096.
 * ------------------------------------------------------------------------------------------------------- */
/* K9 pairwise rotated BEV IoU (n x m), boxes (cx, cy, w, h, angle): mmcv box_iou_rotated under mmdet3d
 * BboxOverlaps3D, used by OTAssignerSRFDet (mmdet3d_plugin/core/bbox/assigners/ota_srfdet.py:148-150). */
/* srf_host_pack: the one vector a frame's host side reads back -- packed detections a (na floats), [survivors, candidates] b (nb ints),
 * live row counts of the sparse levels c (nc ints) -> out (na + nb + nc floats; the integers are < 2^24, exact).  One launch. */
int srf_host_pack(const float *a, int na, const int *b, int nb, const int *c, int nc, float *out, srf_stream_t stream);
int srf_box_iou_rotated(const float *boxes_a, int n, const float *boxes_b, int m, float *iou, srf_stream_t stream);
size_t srf_nms_rotated_workspace_bytes(int n);
int srf_nms_rotated(const float *boxes, int n, float iou_threshold, int *keep, void *workspace, size_t workspace_bytes,
                    srf_stream_t stream);
/* the same over the first *n_dev (device int, <= n) boxes only; keep[i] = 0 for the rest.  A fixed-shape form for
 * hipGraph replay: n is the capacity, the number of candidates above the score threshold is decided on the device. */
int srf_nms_rotated_counted(const float *boxes_xywhr, int n, const int *n_dev, float iou_threshold, int *keep,
                            void *workspace, size_t workspace_bytes, srf_stream_t stream);
/* class-aware: cls (n, int64) -- a box only suppresses boxes of its own class, i.e. the per-class loop of mmdet3d
 * box3d_multiclass_nms (mmdet3d/core/post_processing/box3d_nms.py as called at srfdet_head.py:1276-1293) in one pass on the
 * boxes' own coordinates (no class offsets added to x).  n_dev may be NULL (all n boxes are candidates). */
int srf_nms_rotated_classes(const float *boxes_xywhr, const long long *cls, int n, const int *n_dev, float iou_threshold,
                            int *keep, void *workspace, size_t workspace_bytes, srf_stream_t stream);

/* The fixed-shape multi-class selection around srf_nms_rotated_classes: the static form of mmdet3d box3d_multiclass_nms as
 * called at srfdet_head.py:1276-1293, without a host read-back (hipGraph replay).
 * srf_nms_select: boxes (n, D >= 7) [x, y, z, w, l, h, yaw, ...], scores (n, C) -> the L best (box, class) pairs above
 * score_thr in descending score: cand (L, D), top_s (L), cls (L, int64), bev (L, 5) = [x, y, w, l, yaw] (with cls the
 * operands of srf_nms_rotated_classes), *m (device int) = number of pairs above the threshold -- if it exceeds L the caller redoes the frame on the dynamic
 * path.  n * C <= 16384, L <= n * C.
 * srf_nms_finish: keep (L) from srf_nms_rotated_classes -> survivors first, class-major, descending score inside a class
 * (the order of the reference's per-class loop): out_boxes (L, D), out_scores (L), out_labels (L, int64), *kept. */
int srf_nms_select(const float *boxes, const float *scores, int n, int C, int D, float score_thr, int L, float *cand,
                   float *top_s, long long *cls, float *bev, int *m, srf_stream_t stream);
int srf_nms_finish(const float *cand, const float *top_s, const long long *cls, const int *keep, int L, int D,
                   float *out_boxes, float *out_scores, long long *out_labels, int *kept,
                   float *packed /* (L, D+2) rows [box, score, label] for one D2H copy, or NULL */,
                   const int *m /* srf_nms_select's count */, int *counts /* [kept, m] or NULL */, srf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SRFDET3D_H */
