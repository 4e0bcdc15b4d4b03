// roi.hip -- K7 multi-level RoIAlign gather and the fused proposal-box -> RoI geometry (gfx950).
//
// Reference call sites: pooler(feats[:4], rois) at mmdet3d_plugin/models/sparse_heads/srfdet_head.py:1685, :2548,
// :2626 (mmdet SingleRoIExtractor + mmcv RoIAlign, semantics in SURVEY.md Appendix B.5), and the box geometry of
// points_feats_sampling_bboxes_roi (:1638-1683) / img_feats_sampling_bboxes_roi (:2435-2528) with
// boxes3d_to_corners3d (mmdet3d_plugin/core/bbox/util.py:84-176).
//
// The gather is HBM/L2-bound: one workgroup per (RoI, output row of 7 bins).  The 28 bilinear sample points of the
// row are resolved once into LDS (4 tap offsets + 4 weights each); then every thread owns a channel and walks the
// samples, so a channels-last map is read 512 B at a time per tap.  Feature maps are addressed through element
// strides, so NCHW tensors work too (uncoalesced).  Arithmetic follows oracle/srf_oracle.c operation by operation
// (no fma contraction) so that the two agree exactly.
#include "common.hpp"

#define SRF_MAX_LEVELS 4
#define SRF_MAX_POOLED 8
#define SRF_MAX_SR 4

#define SRF_ROI_NC 4   /* channels per thread whose running sums srf_roi_extract_k keeps in registers (C <= 4 x 128) */

struct RoiLevels {
    srf_featmap lv[SRF_MAX_LEVELS];
    int num;
};

__device__ __forceinline__ int srf_roi_level(const float *b, int num_levels, float finest_scale)
{
    float area = __fmul_rn(__fsub_rn(b[3], b[1]), __fsub_rn(b[4], b[2]));
    float scale = sqrtf(area);
    float t = floorf(log2f(__fadd_rn(__fdiv_rn(scale, finest_scale), 1e-6f)));
    if (!(t == t)) return 0;
    return t < 0.0f ? 0 : (t > (float)(num_levels - 1) ? num_levels - 1 : (int)t);
}

// n_sum > 1 (srf_roi_extract_sum): output row r is the SUM over s = 0 .. n_sum - 1 of the gathers of the RoIs s * R + r, added in
// that order -- the per-camera image RoIs of a proposal (srfdet_head.py:2543-2562 gathers all n_cam * R RoIs and then sums the
// cameras): one launch and no (n_cam * R, S, C) intermediate instead of gather -> sum.  n_sum == 1 is the plain gather.
__global__ __launch_bounds__(128) void srf_roi_extract_k(RoiLevels L, int C, const float *__restrict__ rois, int R,
                                                       int pooled, int sr, float finest_scale, float *__restrict__ out,
                                                       long long so_r, long long so_c, long long so_b, int accumulate,
                                                       int *__restrict__ levels_out, int n_sum)
{
    __shared__ long long s_off[SRF_MAX_POOLED * SRF_MAX_SR * SRF_MAX_SR][4];
    __shared__ float s_w[SRF_MAX_POOLED * SRF_MAX_SR * SRF_MAX_SR][4];
    __shared__ int s_lvl;
    const int r_out = blockIdx.x, ph = blockIdx.y;
    // The running sums over the n_sum RoIs live in registers (up to SRF_ROI_NC channels per thread) and are written once; a RoI
    // with no sample inside its map -- a proposal is seen by one or two of the six cameras -- is skipped as a whole (it adds +0).
    // Before: every camera added its result into the output element in global memory, 7 read-modify-writes per thread and camera,
    // and walked all 28 samples of the invisible ones too: 61 -> 42 us per stage for 200 proposals x 6 cameras.  (Tried on top and
    // dropped: the pooled size and sampling ratio as template parameters with the bin / sample loops unrolled and every tap loaded
    // unconditionally -- 73 us: the 112 64-bit offsets of a thread no longer fit in registers.)
    float run[SRF_ROI_NC][SRF_MAX_POOLED];
#pragma unroll
    for (int i = 0; i < SRF_ROI_NC; ++i)
#pragma unroll
        for (int j = 0; j < SRF_MAX_POOLED; ++j) run[i][j] = 0.0f;
    const bool in_regs = C <= SRF_ROI_NC * (int)blockDim.x && pooled <= SRF_MAX_POOLED;
    for (int sidx = 0; sidx < n_sum; ++sidx) {
    const int r = sidx * R + r_out;
    if (sidx > 0) __syncthreads();   // the previous RoI's sample table is no longer read
    const float *b = rois + (size_t)r * 5;
    if (threadIdx.x == 0) {
        int l = srf_roi_level(b, L.num, finest_scale);
        s_lvl = l;
        if (levels_out && ph == 0) levels_out[r] = l;
    }
    __syncthreads();
    const srf_featmap f = L.lv[s_lvl];
    const int n = (int)b[0];
    const int nsamp = pooled * sr * sr;
    const bool valid_n = n >= 0 && n < f.N;
    bool any_w = false;
    if (threadIdx.x < nsamp) {
        const int pw = threadIdx.x / (sr * sr), iy = (threadIdx.x / sr) % sr, ix = threadIdx.x % sr;
        const float x1 = __fsub_rn(__fmul_rn(b[1], f.spatial_scale), 0.5f), y1 = __fsub_rn(__fmul_rn(b[2], f.spatial_scale), 0.5f);
        const float x2 = __fsub_rn(__fmul_rn(b[3], f.spatial_scale), 0.5f), y2 = __fsub_rn(__fmul_rn(b[4], f.spatial_scale), 0.5f);
        const float bin_h = __fdiv_rn(__fsub_rn(y2, y1), (float)pooled), bin_w = __fdiv_rn(__fsub_rn(x2, x1), (float)pooled);
        float y = __fadd_rn(__fadd_rn(y1, __fmul_rn((float)ph, bin_h)),
                            __fdiv_rn(__fmul_rn(__fadd_rn((float)iy, 0.5f), bin_h), (float)sr));
        float x = __fadd_rn(__fadd_rn(x1, __fmul_rn((float)pw, bin_w)),
                            __fdiv_rn(__fmul_rn(__fadd_rn((float)ix, 0.5f), bin_w), (float)sr));
        float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
        long long o1 = 0, o2 = 0, o3 = 0, o4 = 0;
        const int H = f.H, W = f.W;
        if (valid_n && !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) && (y == y) && (x == x)) {
            if (y <= 0.0f) y = 0.0f;
            if (x <= 0.0f) x = 0.0f;
            int y_low = (int)y, x_low = (int)x, y_high, x_high;
            if (y_low >= H - 1) {
                y_high = y_low = H - 1;
                y = (float)y_low;
            } else
                y_high = y_low + 1;
            if (x_low >= W - 1) {
                x_high = x_low = W - 1;
                x = (float)x_low;
            } else
                x_high = x_low + 1;
            const float ly = __fsub_rn(y, (float)y_low), lx = __fsub_rn(x, (float)x_low);
            const float hy = __fsub_rn(1.0f, ly), hx = __fsub_rn(1.0f, lx);
            w1 = __fmul_rn(hy, hx);
            w2 = __fmul_rn(hy, lx);
            w3 = __fmul_rn(ly, hx);
            w4 = __fmul_rn(ly, lx);
            const long long base = (long long)n * f.stride_n;
            o1 = base + y_low * f.stride_h + x_low * f.stride_w;
            o2 = base + y_low * f.stride_h + x_high * f.stride_w;
            o3 = base + y_high * f.stride_h + x_low * f.stride_w;
            o4 = base + y_high * f.stride_h + x_high * f.stride_w;
        }
        s_off[threadIdx.x][0] = o1;
        s_off[threadIdx.x][1] = o2;
        s_off[threadIdx.x][2] = o3;
        s_off[threadIdx.x][3] = o4;
        s_w[threadIdx.x][0] = w1;
        s_w[threadIdx.x][1] = w2;
        s_w[threadIdx.x][2] = w3;
        s_w[threadIdx.x][3] = w4;
        any_w = w1 != 0.f || w2 != 0.f || w3 != 0.f || w4 != 0.f;
    }
    const int live = __syncthreads_or(any_w ? 1 : 0);
    if (in_regs && !live) continue;   // block-uniform: nothing of this RoI lies inside its map
    const float count = (float)(sr * sr);
    int ci = 0;
    for (int c = threadIdx.x; c < C; c += blockDim.x, ++ci) {
        const float *plane = f.data + (long long)c * f.stride_c;
        for (int pw = 0; pw < pooled; ++pw) {
            float acc = 0.0f;
            // The taps are loaded UNCONDITIONALLY, four samples (16 taps) at a time: a sample outside its map has offset 0 (a valid
            // address) and its value is dropped by the select below, exactly the `v = 0` of the branch this replaces.  With the
            // loads under `if (any weight != 0)` hipcc waited for each sample's four taps before issuing the next sample's
            // (s_waitcnt vmcnt(0) per sample): 28 round trips in a row per bin row, the whole of this launch-sized kernel.
            const int q0 = pw * sr * sr, nq = sr * sr;
            int s = 0;
            for (; s + 4 <= nq; s += 4) {
                float p[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t) p[u][t] = plane[s_off[q0 + s + u][t]];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + s + u;
                    float v = __fmul_rn(s_w[q][0], p[u][0]);
                    v = __fadd_rn(v, __fmul_rn(s_w[q][1], p[u][1]));
                    v = __fadd_rn(v, __fmul_rn(s_w[q][2], p[u][2]));
                    v = __fadd_rn(v, __fmul_rn(s_w[q][3], p[u][3]));
                    const bool nz = s_w[q][0] != 0.f || s_w[q][1] != 0.f || s_w[q][2] != 0.f || s_w[q][3] != 0.f;
                    acc = __fadd_rn(acc, nz ? v : 0.0f);
                }
            }
            for (; s < nq; ++s) {
                const int q = q0 + s;
                float p[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) p[t] = plane[s_off[q][t]];
                float v = __fmul_rn(s_w[q][0], p[0]);
                v = __fadd_rn(v, __fmul_rn(s_w[q][1], p[1]));
                v = __fadd_rn(v, __fmul_rn(s_w[q][2], p[2]));
                v = __fadd_rn(v, __fmul_rn(s_w[q][3], p[3]));
                const bool nz = s_w[q][0] != 0.f || s_w[q][1] != 0.f || s_w[q][2] != 0.f || s_w[q][3] != 0.f;
                acc = __fadd_rn(acc, nz ? v : 0.0f);
            }
            const float res = __fdiv_rn(acc, count);
            if (in_regs) {
#pragma unroll
                for (int i = 0; i < SRF_ROI_NC; ++i)
#pragma unroll
                    for (int j = 0; j < SRF_MAX_POOLED; ++j)
                        if (i == ci && j == pw) run[i][j] = __fadd_rn(run[i][j], res);   // (constant indices: the array stays in registers)
            } else {
                float *dst = out + (long long)r_out * so_r + (long long)c * so_c + (long long)(ph * pooled + pw) * so_b;
                // the running sum over the n_sum RoIs lives in the output element itself: written by this thread only, in order
                *dst = (accumulate || sidx > 0) ? __fadd_rn(*dst, res) : res;
            }
        }
    }
    }
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < SRF_ROI_NC; ++i) {
            const int c = threadIdx.x + i * (int)blockDim.x;
            if (c >= C) break;
#pragma unroll
            for (int j = 0; j < SRF_MAX_POOLED; ++j) {
                if (j >= pooled) break;
                float *dst = out + (long long)r_out * so_r + (long long)c * so_c + (long long)(ph * pooled + j) * so_b;
                *dst = accumulate ? __fadd_rn(*dst, run[i][j]) : run[i][j];
            }
        }
    }
}

extern "C" int srf_roi_extract(const srf_featmap *levels, int num_levels, int C, const float *rois, int R, int pooled,
                               int sampling_ratio, float finest_scale, float *out, int64_t out_stride_r,
                               int64_t out_stride_c, int64_t out_stride_bin, int accumulate, int *levels_out,
                               srf_stream_t stream)
{
    if (!levels || num_levels <= 0 || num_levels > SRF_MAX_LEVELS || C <= 0 || R < 0 || pooled <= 0 ||
        pooled > SRF_MAX_POOLED || sampling_ratio <= 0 || sampling_ratio > SRF_MAX_SR || !(finest_scale > 0.0f))
        return SRF_EINVAL;
    if (pooled * sampling_ratio * sampling_ratio > 128) return SRF_EINVAL;
    if (R == 0) return SRF_OK;
    if (!rois || !out) return SRF_EINVAL;
    RoiLevels L;
    L.num = num_levels;
    for (int i = 0; i < num_levels; ++i) {
        if (!levels[i].data || levels[i].N <= 0 || levels[i].H <= 0 || levels[i].W <= 0) return SRF_EINVAL;
        L.lv[i] = levels[i];
    }
    for (int i = num_levels; i < SRF_MAX_LEVELS; ++i) L.lv[i] = levels[0];
    hipLaunchKernelGGL(srf_roi_extract_k, dim3(R, pooled), dim3(128), 0, (hipStream_t)stream, L, C, rois, R, pooled,
                       sampling_ratio, finest_scale, out, (long long)out_stride_r, (long long)out_stride_c,
                       (long long)out_stride_bin, accumulate, levels_out, 1);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_roi_extract_sum(const srf_featmap *levels, int num_levels, int C, const float *rois, int R, int n_sum, int pooled,
                                   int sampling_ratio, float finest_scale, float *out, int64_t out_stride_r, int64_t out_stride_c,
                                   int64_t out_stride_bin, srf_stream_t stream)
{
    if (!levels || num_levels <= 0 || num_levels > SRF_MAX_LEVELS || C <= 0 || R < 0 || n_sum <= 0 || pooled <= 0 ||
        pooled > SRF_MAX_POOLED || sampling_ratio <= 0 || sampling_ratio > SRF_MAX_SR || !(finest_scale > 0.0f))
        return SRF_EINVAL;
    if (pooled * sampling_ratio * sampling_ratio > 128) return SRF_EINVAL;
    if (R == 0) return SRF_OK;
    if (!rois || !out) return SRF_EINVAL;
    RoiLevels L;
    L.num = num_levels;
    for (int i = 0; i < num_levels; ++i) {
        if (!levels[i].data || levels[i].N <= 0 || levels[i].H <= 0 || levels[i].W <= 0) return SRF_EINVAL;
        L.lv[i] = levels[i];
    }
    for (int i = num_levels; i < SRF_MAX_LEVELS; ++i) L.lv[i] = levels[0];
    hipLaunchKernelGGL(srf_roi_extract_k, dim3(R, pooled), dim3(128), 0, (hipStream_t)stream, L, C, rois, R, pooled,
                       sampling_ratio, finest_scale, out, (long long)out_stride_r, (long long)out_stride_c,
                       (long long)out_stride_bin, 0, (int *)nullptr, n_sum);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// backward of the gather w.r.t. the feature maps (training of the image branch: tools/train.py freezes the LiDAR
// branch but VoVNet stages 3-5, the image FPN and img_convs receive gradients through RoIAlign).  Same geometry as
// the forward kernel; every tap adds weight * grad / count into its map with a float atomic (the order of the adds,
// and therefore the last bits of the gradient, is not deterministic -- the same holds for the reference's mmcv op).
// RoIs carry no gradient, as in mmcv.
// ---------------------------------------------------------------------------------------------------------------------
struct RoiGradLevels {
    float *grad[SRF_MAX_LEVELS];
};

__global__ __launch_bounds__(128) void srf_roi_extract_bwd_k(RoiLevels L, RoiGradLevels G, int C, const float *__restrict__ rois,
                                                           int R, int pooled, int sr, float finest_scale,
                                                           const float *__restrict__ gout, long long so_r, long long so_c,
                                                           long long so_b)
{
    __shared__ long long s_off[SRF_MAX_POOLED * SRF_MAX_SR * SRF_MAX_SR][4];
    __shared__ float s_w[SRF_MAX_POOLED * SRF_MAX_SR * SRF_MAX_SR][4];
    __shared__ int s_lvl;
    const int r = blockIdx.x, ph = blockIdx.y;
    const float *b = rois + (size_t)r * 5;
    if (threadIdx.x == 0) s_lvl = srf_roi_level(b, L.num, finest_scale);
    __syncthreads();
    const srf_featmap f = L.lv[s_lvl];
    float *gmap = G.grad[s_lvl];
    const int n = (int)b[0];
    const int nsamp = pooled * sr * sr;
    const bool valid_n = n >= 0 && n < f.N;
    if (threadIdx.x < nsamp) {
        const int pw = threadIdx.x / (sr * sr), iy = (threadIdx.x / sr) % sr, ix = threadIdx.x % sr;
        const float x1 = __fsub_rn(__fmul_rn(b[1], f.spatial_scale), 0.5f), y1 = __fsub_rn(__fmul_rn(b[2], f.spatial_scale), 0.5f);
        const float x2 = __fsub_rn(__fmul_rn(b[3], f.spatial_scale), 0.5f), y2 = __fsub_rn(__fmul_rn(b[4], f.spatial_scale), 0.5f);
        const float bin_h = __fdiv_rn(__fsub_rn(y2, y1), (float)pooled), bin_w = __fdiv_rn(__fsub_rn(x2, x1), (float)pooled);
        float y = __fadd_rn(__fadd_rn(y1, __fmul_rn((float)ph, bin_h)),
                            __fdiv_rn(__fmul_rn(__fadd_rn((float)iy, 0.5f), bin_h), (float)sr));
        float x = __fadd_rn(__fadd_rn(x1, __fmul_rn((float)pw, bin_w)),
                            __fdiv_rn(__fmul_rn(__fadd_rn((float)ix, 0.5f), bin_w), (float)sr));
        float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
        long long o1 = 0, o2 = 0, o3 = 0, o4 = 0;
        const int H = f.H, W = f.W;
        if (valid_n && !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) && (y == y) && (x == x)) {
            if (y <= 0.0f) y = 0.0f;
            if (x <= 0.0f) x = 0.0f;
            int y_low = (int)y, x_low = (int)x, y_high, x_high;
            if (y_low >= H - 1) {
                y_high = y_low = H - 1;
                y = (float)y_low;
            } else
                y_high = y_low + 1;
            if (x_low >= W - 1) {
                x_high = x_low = W - 1;
                x = (float)x_low;
            } else
                x_high = x_low + 1;
            const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.0f - ly, hx = 1.0f - lx;
            w1 = hy * hx;
            w2 = hy * lx;
            w3 = ly * hx;
            w4 = ly * lx;
            const long long base = (long long)n * f.stride_n;
            o1 = base + y_low * f.stride_h + x_low * f.stride_w;
            o2 = base + y_low * f.stride_h + x_high * f.stride_w;
            o3 = base + y_high * f.stride_h + x_low * f.stride_w;
            o4 = base + y_high * f.stride_h + x_high * f.stride_w;
        }
        s_off[threadIdx.x][0] = o1;
        s_off[threadIdx.x][1] = o2;
        s_off[threadIdx.x][2] = o3;
        s_off[threadIdx.x][3] = o4;
        s_w[threadIdx.x][0] = w1;
        s_w[threadIdx.x][1] = w2;
        s_w[threadIdx.x][2] = w3;
        s_w[threadIdx.x][3] = w4;
    }
    __syncthreads();
    const float inv = 1.0f / (float)(sr * sr);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float *plane = gmap + (long long)c * f.stride_c;
        for (int pw = 0; pw < pooled; ++pw) {
            const float g = gout[(long long)r * so_r + (long long)c * so_c + (long long)(ph * pooled + pw) * so_b] * inv;
            for (int s = 0; s < sr * sr; ++s) {
                const int q = pw * sr * sr + s;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (s_w[q][t] != 0.f) atomicAdd(plane + s_off[q][t], s_w[q][t] * g);
            }
        }
    }
}

extern "C" int srf_roi_extract_bwd(const srf_featmap *levels, float *const *grad_levels, int num_levels, int C,
                                   const float *rois, int R, int pooled, int sampling_ratio, float finest_scale,
                                   const float *grad_out, int64_t out_stride_r, int64_t out_stride_c, int64_t out_stride_bin,
                                   srf_stream_t stream)
{
    if (!levels || !grad_levels || num_levels <= 0 || num_levels > SRF_MAX_LEVELS || C <= 0 || R < 0 || pooled <= 0 ||
        pooled > SRF_MAX_POOLED || sampling_ratio <= 0 || sampling_ratio > SRF_MAX_SR || !(finest_scale > 0.0f))
        return SRF_EINVAL;
    if (pooled * sampling_ratio * sampling_ratio > 128) return SRF_EINVAL;
    if (R == 0) return SRF_OK;
    if (!rois || !grad_out) return SRF_EINVAL;
    RoiLevels L;
    RoiGradLevels G;
    L.num = num_levels;
    for (int i = 0; i < SRF_MAX_LEVELS; ++i) {
        const int j = i < num_levels ? i : 0;
        if (!grad_levels[j] || levels[j].N <= 0) return SRF_EINVAL;
        L.lv[i] = levels[j];
        G.grad[i] = grad_levels[j];
    }
    hipLaunchKernelGGL(srf_roi_extract_bwd_k, dim3(R, pooled), dim3(128), 0, (hipStream_t)stream, L, G, C, rois, R, pooled,
                       sampling_ratio, finest_scale, grad_out, (long long)out_stride_r, (long long)out_stride_c,
                       (long long)out_stride_bin);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// proposal box -> BEV RoI and per-camera image RoIs, one thread per (sample, proposal)
// ---------------------------------------------------------------------------------------------------------------------
struct BoxGeom {
    float lo[3], ext[3], vs[2];
};

__global__ __launch_bounds__(128) void srf_box_rois_k(float *__restrict__ boxes, int B, int P, int box_dim, BoxGeom g,
                                                    int mutate, float *__restrict__ rois_bev,
                                                    const float *__restrict__ lidar2img, int n_cam,
                                                    float *__restrict__ rois_img)
{
    const int t = blockIdx.x * 128 + threadIdx.x;
    if (t >= B * P) return;
    const int bi = t / P;
    float *bx = boxes + (size_t)t * box_dim;
    // centres to metres (srfdet_head.py:1646 / :2444): multiply, then add
    float cx = __fadd_rn(__fmul_rn(bx[0], g.ext[0]), g.lo[0]);
    float cy = __fadd_rn(__fmul_rn(bx[1], g.ext[1]), g.lo[1]);
    float cz = __fadd_rn(__fmul_rn(bx[2], g.ext[2]), g.lo[2]);
    if (mutate) {
        bx[0] = cx;
        bx[1] = cy;
        bx[2] = cz;
    }
    const float w = expf(bx[3]), l = expf(bx[4]), h = expf(bx[5]);
    const float ry = atan2f(bx[6], bx[7]);
    const float cs = cosf(ry), sn = sinf(ry);
    const float hw = __fdiv_rn(w, 2.0f), hl = __fdiv_rn(l, 2.0f), hh = __fdiv_rn(h, 2.0f);
    // corner order of util.py:126-141 (bottom_center=False)
    const float sx[8] = {1.f, -1.f, -1.f, 1.f, 1.f, -1.f, -1.f, 1.f};
    const float sy[8] = {-1.f, -1.f, 1.f, 1.f, -1.f, -1.f, 1.f, 1.f};
    const float sz[8] = {-1.f, -1.f, -1.f, -1.f, 1.f, 1.f, 1.f, 1.f};
    float X[8], Y[8], Z[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float x0 = sx[c] * hw, y0 = sy[c] * hl, z0 = sz[c] * hh;
        // [x0 y0 z0] @ [[cs,-sn,0],[sn,cs,0],[0,0,1]]  (util.py:146-159)
        const float xr = __fadd_rn(__fmul_rn(x0, cs), __fmul_rn(y0, sn));
        const float yr = __fadd_rn(__fmul_rn(x0, -sn), __fmul_rn(y0, cs));
        X[c] = __fadd_rn(cx, xr);
        Y[c] = __fadd_rn(cy, yr);
        Z[c] = __fadd_rn(cz, z0);
    }
    if (rois_bev) {
        float x1 = INFINITY, y1 = INFINITY, x2 = -INFINITY, y2 = -INFINITY;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float px = __fdiv_rn(__fsub_rn(X[c], g.lo[0]), g.vs[0]);
            const float py = __fdiv_rn(__fsub_rn(Y[c], g.lo[1]), g.vs[1]);
            x1 = fminf(x1, px);
            x2 = fmaxf(x2, px);
            y1 = fminf(y1, py);
            y2 = fmaxf(y2, py);
        }
        float *r = rois_bev + (size_t)t * 5;
        r[0] = (float)bi;
        r[1] = x1;
        r[2] = y1;
        r[3] = x2;
        r[4] = y2;
    }
    if (rois_img) {
        for (int cam = 0; cam < n_cam; ++cam) {
            const float *M = lidar2img + ((size_t)bi * n_cam + cam) * 16;
            float x1 = INFINITY, y1 = INFINITY, x2 = -INFINITY, y2 = -INFINITY;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float pu = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(M[0], X[c]), __fmul_rn(M[1], Y[c])), __fmul_rn(M[2], Z[c])), M[3]);
                const float pv = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(M[4], X[c]), __fmul_rn(M[5], Y[c])), __fmul_rn(M[6], Z[c])), M[7]);
                const float pz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(M[8], X[c]), __fmul_rn(M[9], Y[c])), __fmul_rn(M[10], Z[c])), M[11]);
                const float d = fmaxf(pz, 1e-5f);
                const float u = __fdiv_rn(pu, d), v = __fdiv_rn(pv, d);
                x1 = fminf(x1, u);
                x2 = fmaxf(x2, u);
                y1 = fminf(y1, v);
                y2 = fmaxf(y2, v);
            }
            // cam-major rows, batch id b + cam*B (srfdet_head.py:2520-2528)
            float *r = rois_img + ((size_t)cam * B * P + t) * 5;
            r[0] = (float)(bi + cam * B);
            r[1] = x1;
            r[2] = y1;
            r[3] = x2;
            r[4] = y2;
        }
    }
}

extern "C" int srf_box_rois(float *boxes, int B, int P, int box_dim, const float *pc_range, const float *voxel_size,
                            int mutate_centres, float *rois_bev, const float *lidar2img, int n_cam, float *rois_img,
                            srf_stream_t stream)
{
    if (B <= 0 || P < 0 || box_dim < 8 || !pc_range || !voxel_size) return SRF_EINVAL;
    if (rois_img && (!lidar2img || n_cam <= 0)) return SRF_EINVAL;
    if (P == 0) return SRF_OK;
    if (!boxes) return SRF_EINVAL;
    BoxGeom g;
    for (int j = 0; j < 3; ++j) {
        g.lo[j] = pc_range[j];
        g.ext[j] = pc_range[3 + j] - pc_range[j];
    }
    g.vs[0] = voxel_size[0];
    g.vs[1] = voxel_size[1];
    hipLaunchKernelGGL(srf_box_rois_k, dim3(srf_ceil_div(B * P, 128)), dim3(128), 0, (hipStream_t)stream, boxes, B, P,
                       box_dim, g, mutate_centres, rois_bev, lidar2img, n_cam, rois_img);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}
