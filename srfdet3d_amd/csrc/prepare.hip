// prepare.hip -- the step before the path: what the reference's test pipeline does on the CPU between the
// decoded sensor data and SRFDet.forward, on the device (gfx950).
//
// Reference call sites: the `test_pipeline` of configs/nus/srfdet_voxel_nusc_LC.py:253-283 --
//   PointsRangeFilter (mmdet3d 1.0.0rc6, points.in_range_3d: strict inequalities on x, y, z; order kept) and the
//   `remove_close` of LoadPointsFromMultiSweeps (|x| < r and |y| < r dropped),
//   NormalizeMultiviewImage / PadMultiViewImage of mmdet3d_plugin/datasets/pipelines/transform_3d.py:7-93
//   ((x - mean) * (1 / std) in float32 after an optional BGR->RGB swap; zero padding at the bottom / right up to a
//   multiple of size_divisor), and the HWC -> CHW transpose + stack of DefaultFormatBundle3D.
// Both are pure streaming work: the point filter is a flag + exclusive scan + ordered copy (bit-exact index work), the
// image kernel reads 1 byte and writes 4 per value (6 x 900 x 1600 x 3 u8 = 26 MB in, 107 MB out per nuScenes frame).
#include "common.hpp"

// ---------------------------------------------------------------------------------------------------------------------
// points: keep point i iff inside the open range box and not within the close radius; order preserved
// ---------------------------------------------------------------------------------------------------------------------
struct PfKeep {
    const float *p;
    int nf;
    float lo[3], hi[3];
    float radius;
    int use_range;
    __device__ int operator()(int i) const
    {
        const float x = p[(size_t)i * nf], y = p[(size_t)i * nf + 1], z = p[(size_t)i * nf + 2];
        bool keep = true;
        if (use_range) keep = x > lo[0] && y > lo[1] && z > lo[2] && x < hi[0] && y < hi[1] && z < hi[2];
        if (radius > 0.0f) keep = keep && !(fabsf(x) < radius && fabsf(y) < radius);
        return keep ? 1 : 0;
    }
};

struct PfCopy {
    const float *p;
    float *out;
    int *index;
    int nf;
    __device__ void operator()(int i, int v, int prefix) const
    {
        if (!v) return;
        for (int c = 0; c < nf; ++c) out[(size_t)prefix * nf + c] = p[(size_t)i * nf + c];
        if (index) index[prefix] = i;
    }
};

extern "C" size_t srf_points_filter_workspace_bytes(int n) { return n < 0 ? 0 : ((size_t)srf_scan_blocks(n) + 2) * sizeof(int); }

extern "C" int srf_points_filter(const float *points, int n, int nf, const float *pc_range, float close_radius, float *out_points,
                                 int *out_index, int *num_out, void *workspace, srf_stream_t stream)
{
    if (n < 0 || nf < 3 || !num_out) return SRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        SRF_HIP_TRY(srf_fill_bytes(num_out, 0, sizeof(int), st));
        return SRF_OK;
    }
    if (!points || !out_points || !workspace) return SRF_EINVAL;
    PfKeep keep;
    keep.p = points;
    keep.nf = nf;
    keep.radius = close_radius;
    keep.use_range = pc_range != nullptr;
    for (int d = 0; d < 3; ++d) {
        keep.lo[d] = pc_range ? pc_range[d] : 0.0f;
        keep.hi[d] = pc_range ? pc_range[3 + d] : 0.0f;
    }
    PfCopy copy{points, out_points, out_index, nf};
    return srf_device_scan(n, keep, copy, (int *)workspace, num_out, -1, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// images: V views of H x W x 3 bytes (as decoded) -> (V, 3, Hp, Wp) float32, normalised, zero-padded
// ---------------------------------------------------------------------------------------------------------------------
// One thread = 4 consecutive output pixels of one (view, row): the 12 source bytes are contiguous, the three planes get
// one 16-byte store each.  Rows / columns of the padding are written as zeros by the same grid.
__global__ __launch_bounds__(256) void srf_image_prepare_k(const unsigned char *__restrict__ src, int V, int H, int W, int Hp, int Wp,
                                                         float m0, float m1, float m2, float i0, float i1, float i2, int swap_rb,
                                                         float *__restrict__ dst)
{
    const int wq = Wp >> 2;  // Wp % 4 == 0 (size_divisor is a multiple of 4)
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)V * Hp * wq;
    if (t >= total) return;
    const int xq = (int)(t % wq);
    const long long rest = t / wq;
    const int y = (int)(rest % Hp), v = (int)(rest / Hp);
    const int x0 = xq * 4;
    float r[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) r[c][j] = 0.0f;
    if (y < H && x0 < W) {
        const unsigned char *s = src + (((size_t)v * H + y) * W + x0) * 3;
        const float mean[3] = {m0, m1, m2}, inv[3] = {i0, i1, i2};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (x0 + j < W) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int sc = swap_rb ? 2 - c : c;  // cv2.COLOR_BGR2RGB before the normalisation
                    r[c][j] = __fmul_rn(__fsub_rn((float)s[j * 3 + sc], mean[c]), inv[c]);
                }
            }
        }
    }
    const size_t plane = (size_t)Hp * Wp;
    float *d = dst + (size_t)v * 3 * plane + (size_t)y * Wp + x0;
#pragma unroll
    for (int c = 0; c < 3; ++c) *reinterpret_cast<float4 *>(d + c * plane) = make_float4(r[c][0], r[c][1], r[c][2], r[c][3]);
}

extern "C" int srf_image_prepare(const unsigned char *images, int V, int H, int W, const float *mean, const float *std, int to_rgb,
                                 int Hp, int Wp, float *out, srf_stream_t stream)
{
    if (V < 0 || H <= 0 || W <= 0 || Hp < H || Wp < W || (Wp & 3) || !mean || !std) return SRF_EINVAL;
    if (V == 0) return SRF_OK;
    if (!images || !out) return SRF_EINVAL;
    float inv[3];
    for (int c = 0; c < 3; ++c) {
        if (!(std[c] > 0.0f)) return SRF_EINVAL;
        inv[c] = (float)(1.0 / (double)std[c]);  // mmcv.imnormalize: stdinv = 1 / np.float64(std), applied to float32 pixels
    }
    const long long total = (long long)V * Hp * (Wp >> 2);
    hipLaunchKernelGGL(srf_image_prepare_k, dim3(srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, images, V, H, W, Hp, Wp,
                       mean[0], mean[1], mean[2], inv[0], inv[1], inv[2], to_rgb ? 1 : 0, out);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}
