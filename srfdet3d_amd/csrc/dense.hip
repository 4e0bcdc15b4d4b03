// dense.hip -- elementwise companions of the dense (MIOpen) convolutions of the path.
//
// srf_channel_affine: y = max(x * scale + shift (+ residual), 0) over an NCHW tensor in ONE pass; scale / shift are
// per channel (BatchNorm) or per (sample, channel) (the eSE gate of vovnet.py:136-150, with the OSA identity add as
// the residual).  The BatchNorm case: -- the eval-mode
// BatchNorm2d + ReLU that follows every convolution of SECONDCustom (second_custom.py:41-63), FPN and VoVNet
// (vovnet.py:39-56), which torch runs as two kernels (MIOpen batch-norm, then a clamp).  HBM-bound: 8 bytes per
// element.  The output may be a channel slice of a wider tensor (its own batch stride), so an OSA block's branch can be
// written straight into the concatenation buffer.
#include "common.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool VEC>
__global__ __launch_bounds__(256) void srf_channel_affine_k(const float *__restrict__ x, int C, int HW, long long x_sn,
                                                          const float *__restrict__ scale, const float *__restrict__ shift,
                                                          int per_sample, const float *__restrict__ residual, int relu,
                                                          float *__restrict__ y, long long y_sn)
{
    const int plane = blockIdx.y;  // n * C + c
    const int n = plane / C, c = plane - n * C;
    const int pi = per_sample ? plane : c;
    const float a = scale[pi], b = shift ? shift[pi] : 0.f;
    const float *px = x + (size_t)n * x_sn + (size_t)c * HW;
    const float *pr = residual ? residual + (size_t)plane * HW : nullptr;
    float *py = y + (size_t)n * y_sn + (size_t)c * HW;
    if (VEC) {
        const int nv = HW >> 2;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < nv; i += gridDim.x * 256) {
            f32x4 v = reinterpret_cast<const f32x4 *>(px)[i];
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (pr) r = reinterpret_cast<const f32x4 *>(pr)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = shift ? __fmaf_rn(v[j], a, b) : __fmul_rn(v[j], a);
                if (pr) t = __fadd_rn(t, r[j]);
                v[j] = relu ? (t > 0.f ? t : 0.f) : t;
            }
            reinterpret_cast<f32x4 *>(py)[i] = v;
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
            float t = shift ? __fmaf_rn(px[i], a, b) : __fmul_rn(px[i], a);
            if (pr) t = __fadd_rn(t, pr[i]);
            py[i] = relu ? (t > 0.f ? t : 0.f) : t;
        }
    }
}

extern "C" int srf_channel_affine(const float *x, int N, int C, int HW, long long x_batch_stride, const float *scale,
                                  const float *shift, int per_sample, const float *residual, int relu, float *y,
                                  long long y_batch_stride, srf_stream_t stream)
{
    if (N < 0 || C <= 0 || HW <= 0 || x_batch_stride < (long long)C * HW || y_batch_stride < (long long)C * HW) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!x || !y || !scale) return SRF_EINVAL;
    if ((long long)N * C > 65535) return SRF_EUNSUPPORTED;
    const bool vec = (HW % 4 == 0) && (x_batch_stride % 4 == 0) && (y_batch_stride % 4 == 0) && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) % 16 == 0);
    const int per_block = vec ? 256 * 4 * 4 : 256 * 4;
    int gx = srf_ceil_div(HW, per_block);
    if (gx < 1) gx = 1;
    dim3 grid(gx, N * C);
    if (vec)
        hipLaunchKernelGGL(srf_channel_affine_k<true>, grid, dim3(256), 0, (hipStream_t)stream, x, C, HW, x_batch_stride, scale,
                           shift, per_sample, residual, relu, y, y_batch_stride);
    else
        hipLaunchKernelGGL(srf_channel_affine_k<false>, grid, dim3(256), 0, (hipStream_t)stream, x, C, HW, x_batch_stride, scale,
                           shift, per_sample, residual, relu, y, y_batch_stride);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}
