// nhwc.hip -- the streaming (HBM-bound) layers of the channels-last camera branch: everything around the two MFMA
// kernels of conv.hip.  Activations are (N, H, W, ld) f32 channel slices of pixel-major buffers (see conv.hip); every
// kernel here moves float4 = 4 consecutive channels per thread with the channel quad fastest, so a wave reads and writes
// whole 128-byte runs.
//
// Reference call sites: eval BatchNorm2d + ReLU behind the two stride-2 stem convolutions, the eSE gate multiply and
// the OSA identity add (mmdet3d_plugin/models/backbones/vovnet.py:165-177, :225-228), the stage pooling
// (MaxPool2d(3, 2, ceil_mode=True), vovnet.py `_OSA_stage`), the global average pool of the eSE module, the FPN top-down
// step (mmdet FPN: lateral + nearest-upsampled coarser level), and the depthwise stride-2 stair of the proposal
// generator (srfdet_head.py:265-320, :525-536).
#include "common.hpp"

typedef float f32x4n __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------------
// y[p][c] = x[p][c] * scale[(per_sample ? n : 0)][c] + shift[c] (+ residual[p][c]), optional ReLU.  In place allowed.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void srf_nhwc_affine_k(const float *__restrict__ x, long long x_ld, long long M, long long HW, int Cq,
                                                         const float *__restrict__ scale, int per_sample, const float *__restrict__ shift,
                                                         const float *__restrict__ res, long long r_ld, int relu, float *__restrict__ y,
                                                         long long y_ld)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * Cq) return;
    const long long p = t / Cq;
    const int cq = (int)(t - p * Cq);
    f32x4n v = *reinterpret_cast<const f32x4n *>(x + p * x_ld + cq * 4);
    if (scale) {
        const long long n = (per_sample & 1) ? p / HW : 0;
        v *= *reinterpret_cast<const f32x4n *>(scale + n * Cq * 4 + cq * 4);
    }
    if (shift) v += *reinterpret_cast<const f32x4n *>(shift + ((per_sample & 2) ? (p / HW) * Cq * 4 : 0) + cq * 4);
    if (res) v += *reinterpret_cast<const f32x4n *>(res + p * r_ld + cq * 4);
    if (relu) {
        v[0] = fmaxf(v[0], 0.f);
        v[1] = fmaxf(v[1], 0.f);
        v[2] = fmaxf(v[2], 0.f);
        v[3] = fmaxf(v[3], 0.f);
    }
    *reinterpret_cast<f32x4n *>(y + p * y_ld + cq * 4) = v;
}

extern "C" int srf_nhwc_affine(const float *x, long long x_ld, int N, long long HW, int C, const float *scale, int per_sample,
                               const float *shift, const float *residual, long long r_ld, int relu, float *y, long long y_ld,
                               srf_stream_t stream)
{
    if (N < 0 || HW < 0 || C <= 0 || x_ld < C || y_ld < C || (residual && r_ld < C)) return SRF_EINVAL;
    if (N == 0 || HW == 0) return SRF_OK;
    if (!x || !y) return SRF_EINVAL;
    if ((C & 3) || (x_ld & 3) || (y_ld & 3) || (r_ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)residual & 15) ||
        ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15))
        return SRF_EUNSUPPORTED;
    const long long M = (long long)N * HW, total = M * (C / 4);
    hipLaunchKernelGGL(srf_nhwc_affine_k, dim3(srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_ld, M, HW, C / 4, scale,
                       per_sample, shift, residual, r_ld, relu, y, y_ld);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// srf_ese_apply: the end of VoVNet's eSE module (vovnet.py:165-177, :225-228) as ONE launch: gate = hsigmoid(fc(mean)) and
// y = x * gate (+ identity).  The gate GEMV used to be its own launch between the pooled convolution and this pass -- 14
// microsecond-sized launches per frame on the camera graph's critical path, each costing its launch latency.  Here a workgroup
// owns (64 channels, a range of pixel rows, an image): its four waves first compute their 64 gates exactly as
// srf_linear_gemv_k does (one wave per output: float4 fma chains over k = 4 lane + 256 j, xor-shuffle sum, + bias, relu6(v + 3) / 6
// -- the same bits), ~3 us hidden behind the other workgroups' streaming, then stream their rows (multiply, then add: the
// two roundings of srf_nhwc_affine_k).  C % 64 == 0, C <= 1024.  Measured on the LC frame: slower than the two launches it replaces
// (29.27 -> 28.93 frames/s): the 64-channel slices stream worse than the linear pass; kept as an opt-in (SRF_ESE_FUSED=1), tested.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void srf_ese_apply_k(const float *__restrict__ x, long long x_ld, long long HW, int C,
                                                       const float *__restrict__ mean, const float *__restrict__ W,
                                                       const float *__restrict__ bias, const float *__restrict__ res, long long r_ld,
                                                       float *__restrict__ y, long long y_ld, float *__restrict__ gate_out,
                                                       long long rows_per_part)
{
    __shared__ float s_gate[64];
    const int slice = blockIdx.x, n = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const float *mv = mean + (size_t)n * C;
        f32x4n xv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane * 4 + 256 * i;
            xv[i] = k < C ? *reinterpret_cast<const f32x4n *>(mv + k) : f32x4n{0.f, 0.f, 0.f, 0.f};
        }
        for (int j = 0; j < 16; ++j) {
            const int co = slice * 64 + wave * 16 + j;
            const float *w = W + (size_t)co * C;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = lane * 4 + 256 * i;
                if (k < C) {
                    const f32x4n wv = *reinterpret_cast<const f32x4n *>(w + k);
                    acc = __fmaf_rn(xv[i][3], wv[3], __fmaf_rn(xv[i][2], wv[2], __fmaf_rn(xv[i][1], wv[1], __fmaf_rn(xv[i][0], wv[0], acc))));
                }
            }
            float v = acc;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
            if (lane == 0) {
                v += bias ? bias[co] : 0.f;
                v = __fadd_rn(v, 3.0f);
                v = v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
                v = __fdiv_rn(v, 6.0f);
                s_gate[wave * 16 + j] = v;
                if (gate_out && blockIdx.y == 0) gate_out[(size_t)n * C + co] = v;
            }
        }
    }
    __syncthreads();
    const int q = tid & 15, rl = tid >> 4;
    const f32x4n g = *reinterpret_cast<const f32x4n *>(s_gate + q * 4);
    const long long r0 = (long long)blockIdx.y * rows_per_part;
    const long long r1 = r0 + rows_per_part < HW ? r0 + rows_per_part : HW;
    const int ch = slice * 64 + q * 4;
    for (long long r = r0 + rl; r < r1; r += 16) {
        const long long p = (long long)n * HW + r;
        f32x4n v = *reinterpret_cast<const f32x4n *>(x + p * x_ld + ch);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = __fmul_rn(v[c], g[c]);
        if (res) {
            const f32x4n rv = *reinterpret_cast<const f32x4n *>(res + p * r_ld + ch);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = __fadd_rn(v[c], rv[c]);
        }
        *reinterpret_cast<f32x4n *>(y + p * y_ld + ch) = v;
    }
}

extern "C" int srf_ese_apply(const float *x, long long x_ld, int N, long long HW, int C, const float *mean, const float *W, const float *bias,
                             const float *residual, long long r_ld, float *y, long long y_ld, float *gate_out, srf_stream_t stream)
{
    if (N < 0 || HW < 0 || C <= 0 || x_ld < C || y_ld < C || (residual && r_ld < C)) return SRF_EINVAL;
    if (N == 0 || HW == 0) return SRF_OK;
    if (!x || !y || !mean || !W) return SRF_EINVAL;
    if ((C & 63) || C > 1024 || N > 65535 || (x_ld & 3) || (y_ld & 3) || (r_ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15) ||
        ((uintptr_t)residual & 15) || ((uintptr_t)mean & 15) || ((uintptr_t)W & 15))
        return SRF_EUNSUPPORTED;
    // row ranges: enough workgroups to fill the chip a few times over, at least 256 rows each (the gate prologue is ~3 us)
    long long parts = (HW + 255) / 256;
    const long long want = 4096 / ((long long)(C / 64) * N) + 1;
    parts = parts > want ? want : parts;
    parts = parts < 1 ? 1 : (parts > 65535 ? 65535 : parts);
    const long long rows_per_part = (HW + parts - 1) / parts;
    hipLaunchKernelGGL(srf_ese_apply_k, dim3(C / 64, (unsigned)parts, N), dim3(256), 0, (hipStream_t)stream, x, x_ld, HW, C, mean, W, bias,
                       residual, r_ld, y, y_ld, gate_out, rows_per_part);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// mean over the pixels of every (sample, channel): AdaptiveAvgPool2d(1) of the eSE module.  Deterministic two-level sum:
// level 1: grid (P chunks of pixels, N); a workgroup = 256 threads = (C / 4 channel quads) x (256 / (C / 4) pixel lanes)
// when C <= 1024; each thread sums its pixels in order, the pixel lanes are combined through LDS in a fixed order;
// level 2: one thread per (n, channel) adds the P partial sums in order and divides.
// ---------------------------------------------------------------------------------------------------------------------
#define CM_CHUNKS 64

__global__ __launch_bounds__(256) void srf_nhwc_colsum_k(const float *__restrict__ x, long long x_ld, long long HW, int Cq,
                                                         float *__restrict__ partial, const float *__restrict__ x2 = nullptr,
                                                         long long x2_ld = 0)
{
    __shared__ f32x4n s_p[256];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int lanes = 256 / Cq;  // pixel lanes (>= 1); threads past lanes * Cq idle
    const int cq = threadIdx.x % Cq, pl = threadIdx.x / Cq;
    const long long per = (HW + CM_CHUNKS - 1) / CM_CHUNKS;
    const long long p0 = chunk * per;
    long long p1 = p0 + per;
    if (p1 > HW) p1 = HW;
    f32x4n acc = {0.f, 0.f, 0.f, 0.f};
    const float *xn = x + (long long)n * HW * x_ld + cq * 4;
    if (pl < lanes) {
        if (x2) {   // column sums of the product x * x2 (the eSE gate's gradient: sum over the pixels of g * out)
            const float *x2n = x2 + (long long)n * HW * x2_ld + cq * 4;
            for (long long p = p0 + pl; p < p1; p += lanes)
                acc += *reinterpret_cast<const f32x4n *>(xn + p * x_ld) * *reinterpret_cast<const f32x4n *>(x2n + p * x2_ld);
        } else
            for (long long p = p0 + pl; p < p1; p += lanes) acc += *reinterpret_cast<const f32x4n *>(xn + p * x_ld);
    }
    s_p[threadIdx.x] = acc;
    __syncthreads();
    if (pl == 0) {
        for (int l = 1; l < lanes; ++l) acc += s_p[l * Cq + cq];
        *reinterpret_cast<f32x4n *>(partial + (((long long)n * CM_CHUNKS + chunk) * Cq + cq) * 4) = acc;
    }
}

__global__ __launch_bounds__(256) void srf_nhwc_colmean_finish_k(const float *__restrict__ partial, int N, int C, float inv,
                                                                 float *__restrict__ mean)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= N * C) return;
    const int n = t / C, c = t - n * C;
    float s = 0.f;
    for (int k = 0; k < CM_CHUNKS; ++k) s += partial[((long long)n * CM_CHUNKS + k) * C + c];
    mean[t] = s * inv;
}

extern "C" size_t srf_nhwc_colmean_workspace_bytes(int N, int C) { return (N <= 0 || C <= 0) ? 0 : (size_t)N * CM_CHUNKS * C * 4; }

extern "C" int srf_nhwc_colmean(const float *x, long long x_ld, int N, long long HW, int C, float *mean, void *workspace,
                                size_t workspace_bytes, srf_stream_t stream)
{
    if (N < 0 || HW <= 0 || C <= 0 || x_ld < C) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!x || !mean || !workspace) return SRF_EINVAL;
    const int Cq = C / 4;
    if ((C & 3) || Cq > 256 || (x_ld & 3) || ((uintptr_t)x & 15) || N > 65535) return SRF_EUNSUPPORTED;
    if (workspace_bytes < srf_nhwc_colmean_workspace_bytes(N, C)) return SRF_EWORKSPACE;
    hipLaunchKernelGGL(srf_nhwc_colsum_k, dim3(CM_CHUNKS, N), dim3(256), 0, (hipStream_t)stream, x, x_ld, HW, Cq, (float *)workspace);
    hipLaunchKernelGGL(srf_nhwc_colmean_finish_k, dim3(srf_ceil_div((long long)N * C, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)workspace, N, C, 1.0f / (float)HW, mean);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// per-image column sums of a PRODUCT: out[n][c] = sum over the pixels of a[n][p][c] * b[n][p][c] (two-level, fixed order: deterministic);
// the pixel sum the gradient of VoVNet's eSE gate needs (vovnet.py:165-177: d gate[n][c] = sum_p g * x).  Workspace as srf_nhwc_colmean's.
extern "C" int srf_nhwc_colsum_prod(const float *a, long long a_ld, const float *b, long long b_ld, int N, long long HW, int C, float *out,
                                    void *workspace, size_t workspace_bytes, srf_stream_t stream)
{
    if (N < 0 || HW <= 0 || C <= 0 || a_ld < C || b_ld < C) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!a || !b || !out || !workspace) return SRF_EINVAL;
    const int Cq = C / 4;
    if ((C & 3) || Cq > 256 || (a_ld & 3) || (b_ld & 3) || ((uintptr_t)a & 15) || ((uintptr_t)b & 15) || N > 65535) return SRF_EUNSUPPORTED;
    if (workspace_bytes < srf_nhwc_colmean_workspace_bytes(N, C)) return SRF_EWORKSPACE;
    hipLaunchKernelGGL(srf_nhwc_colsum_k, dim3(CM_CHUNKS, N), dim3(256), 0, (hipStream_t)stream, a, a_ld, HW, Cq, (float *)workspace, b, b_ld);
    hipLaunchKernelGGL(srf_nhwc_colmean_finish_k, dim3(srf_ceil_div((long long)N * C, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)workspace, N, C, 1.0f, out);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// MaxPool2d(kernel 3, stride 2, ceil_mode=True, no padding): windows that reach past the bottom / right edge are clipped.
// ---------------------------------------------------------------------------------------------------------------------
// One thread = a 2 x 2 block of outputs x 4 channels: its 5 x 5 input pixels are loaded once (25 float4 instead of the 36 that
// four independent windows read: the kernel is bound by the loads it issues, not by HBM) and reduced as three-tap maxima
// along x, then along y -- max is exact, the result is the same whatever the order.
__global__ __launch_bounds__(256) void srf_nhwc_maxpool3s2_k(const float *__restrict__ x, long long x_ld, int N, int H, int W, int Cq, int Ho,
                                                             int Wo, float *__restrict__ y, long long y_ld)
{
    const int Ho2 = (Ho + 1) >> 1, Wo2 = (Wo + 1) >> 1;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)N * Ho2 * Wo2 * Cq;
    if (t >= total) return;
    const int cq = (int)(t % Cq);
    long long r = t / Cq;
    const int xb = (int)(r % Wo2);
    r /= Wo2;
    const int yb = (int)(r % Ho2), n = (int)(r / Ho2);
    const float NEG = -__builtin_inff();
    const f32x4n neg = {NEG, NEG, NEG, NEG};
    const int yi0 = 4 * yb, xi0 = 4 * xb;
    const float *base = x + ((long long)n * H * W) * x_ld + cq * 4;
    f32x4n h[5][2];
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
        const int yi = yi0 + dy;
        f32x4n v[5];
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
            const int xi = xi0 + dx;
            v[dx] = (yi < H && xi < W) ? *reinterpret_cast<const f32x4n *>(base + ((long long)yi * W + xi) * x_ld) : neg;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            h[dy][0][c] = fmaxf(fmaxf(v[0][c], v[1][c]), v[2][c]);
            h[dy][1][c] = fmaxf(fmaxf(v[2][c], v[3][c]), v[4][c]);
        }
    }
#pragma unroll
    for (int oy = 0; oy < 2; ++oy) {
        const int yo = 2 * yb + oy;
        if (yo >= Ho) break;
#pragma unroll
        for (int ox = 0; ox < 2; ++ox) {
            const int xo = 2 * xb + ox;
            if (xo >= Wo) continue;
            f32x4n m;
#pragma unroll
            for (int c = 0; c < 4; ++c) m[c] = fmaxf(fmaxf(h[2 * oy][ox][c], h[2 * oy + 1][ox][c]), h[2 * oy + 2][ox][c]);
            *reinterpret_cast<f32x4n *>(y + (((long long)n * Ho + yo) * Wo + xo) * y_ld + cq * 4) = m;
        }
    }
}

static inline int srf_pool_out(int H)
{
    int Ho = (H - 3 + 1) / 2 + 1;  // ceil((H - 3) / 2) + 1
    if (H < 3) Ho = 1;
    if ((Ho - 1) * 2 >= H) --Ho;   // the last window must start inside the input
    return Ho;
}

extern "C" int srf_nhwc_maxpool3s2_ceil(const float *x, long long x_ld, int N, int H, int W, int C, float *y, long long y_ld,
                                        srf_stream_t stream)
{
    if (N < 0 || H < 1 || W < 1 || C <= 0 || x_ld < C || y_ld < C) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!x || !y) return SRF_EINVAL;
    if ((C & 3) || (x_ld & 3) || (y_ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return SRF_EUNSUPPORTED;
    const int Ho = srf_pool_out(H), Wo = srf_pool_out(W);
    const long long total = (long long)N * ((Ho + 1) / 2) * ((Wo + 1) / 2) * (C / 4);
    hipLaunchKernelGGL(srf_nhwc_maxpool3s2_k, dim3(srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_ld, N, H, W, C / 4, Ho, Wo,
                       y, y_ld);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// FPN top-down step: y[n][yy][xx][c] = lat[n][yy][xx][c] + top[n][floor(yy Ht / H)][floor(xx Wt / W)][c] (F.interpolate
// mode='nearest' index rule: src = floor(dst * in / out), computed in float like torch: min(int(dst * scale), in - 1)).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void srf_nhwc_upsample_add_k(const float *__restrict__ lat, long long l_ld, const float *__restrict__ top,
                                                               long long t_ld, int N, int H, int W, int Ht, int Wt, int Cq, float sy, float sx,
                                                               float *__restrict__ y, long long y_ld)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)N * H * W * Cq;
    if (t >= total) return;
    const int cq = (int)(t % Cq);
    long long r = t / Cq;
    const int xx = (int)(r % W);
    r /= W;
    const int yy = (int)(r % H), n = (int)(r / H);
    int ys = (int)floorf((float)yy * sy), xs = (int)floorf((float)xx * sx);
    if (ys > Ht - 1) ys = Ht - 1;
    if (xs > Wt - 1) xs = Wt - 1;
    const long long p = ((long long)n * H + yy) * W + xx;
    f32x4n v = *reinterpret_cast<const f32x4n *>(lat + p * l_ld + cq * 4);
    v += *reinterpret_cast<const f32x4n *>(top + (((long long)n * Ht + ys) * Wt + xs) * t_ld + cq * 4);
    *reinterpret_cast<f32x4n *>(y + p * y_ld + cq * 4) = v;
}

extern "C" int srf_nhwc_upsample_add(const float *lat, long long l_ld, const float *top, long long t_ld, int N, int H, int W, int Ht, int Wt,
                                     int C, float *y, long long y_ld, srf_stream_t stream)
{
    if (N < 0 || H < 1 || W < 1 || Ht < 1 || Wt < 1 || C <= 0 || l_ld < C || t_ld < C || y_ld < C) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!lat || !top || !y) return SRF_EINVAL;
    if ((C & 3) || (l_ld & 3) || (t_ld & 3) || (y_ld & 3) || ((uintptr_t)lat & 15) || ((uintptr_t)top & 15) || ((uintptr_t)y & 15))
        return SRF_EUNSUPPORTED;
    const long long total = (long long)N * H * W * (C / 4);
    hipLaunchKernelGGL(srf_nhwc_upsample_add_k, dim3(srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, lat, l_ld, top, t_ld, N, H, W,
                       Ht, Wt, C / 4, (float)Ht / (float)H, (float)Wt / (float)W, y, y_ld);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Depthwise 3x3 / stride 2 / padding 1 convolution + eval BatchNorm (+ ReLU), channels-last: the stair of the proposal
// generator on the camera levels.  Taps summed in (ky, kx) order like srf_dwconv3x3s2_k; weights w (C, 3, 3).
// ---------------------------------------------------------------------------------------------------------------------
// `side` (N, Ho, Wo, Csq * 4) is copied next to the result (srf_nhwc_dwconv3x3s2_cat): the threads past the convolution's own
// work items move one float4 each.  A convolution thread owns a 2 x 2 block of outputs x 4 channels: its 5 x 5 input pixels
// are loaded once (25 float4 instead of the 36 four independent windows read -- on the finest camera level the kernel is
// bound by the loads it issues); every output still adds its nine taps in (ky, kx) order.
__global__ __launch_bounds__(256) void srf_nhwc_dwconv3x3s2_k(const float *__restrict__ x, long long x_ld, int N, int H, int W, int Cq, int Ho,
                                                              int Wo, const float *__restrict__ w, const float *__restrict__ scale,
                                                              const float *__restrict__ shift, int relu, float *__restrict__ y, long long y_ld,
                                                              const float *__restrict__ side, long long side_ld, int Csq,
                                                              float *__restrict__ side_out)
{
    const int Ho2 = (Ho + 1) >> 1, Wo2 = (Wo + 1) >> 1;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long conv_total = (long long)N * Ho2 * Wo2 * Cq;
    if (t >= conv_total) {
        const long long u = t - conv_total;
        if (u >= (long long)N * Ho * Wo * Csq) return;
        const long long pix = u / Csq;   // (n, yo, xo) linearised = the pixel row of both `side` and the output buffer
        const int sq = (int)(u - pix * Csq);
        *reinterpret_cast<f32x4n *>(side_out + pix * y_ld + sq * 4) = *reinterpret_cast<const f32x4n *>(side + pix * side_ld + sq * 4);
        return;
    }
    const int cq = (int)(t % Cq);
    long long r = t / Cq;
    const int xb = (int)(r % Wo2);
    r /= Wo2;
    const int yb = (int)(r % Ho2), n = (int)(r / Ho2);
    float k[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 9; ++j) k[c][j] = w[(cq * 4 + c) * 9 + j];
    const float *base = x + ((long long)n * H * W) * x_ld + cq * 4;
    const int yi0 = 4 * yb - 1, xi0 = 4 * xb - 1;
    f32x4n v[5][5];
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
        const int yi = yi0 + dy;
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
            const int xi = xi0 + dx;
            f32x4n z = {0.f, 0.f, 0.f, 0.f};
            if (yi >= 0 && yi < H && xi >= 0 && xi < W) z = *reinterpret_cast<const f32x4n *>(base + ((long long)yi * W + xi) * x_ld);
            v[dy][dx] = z;
        }
    }
#pragma unroll
    for (int oy = 0; oy < 2; ++oy) {
        const int yo = 2 * yb + oy;
        if (yo >= Ho) break;
#pragma unroll
        for (int ox = 0; ox < 2; ++ox) {
            const int xo = 2 * xb + ox;
            if (xo >= Wo) continue;
            f32x4n acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = __fmaf_rn(v[2 * oy + ky][2 * ox + kx][c], k[c][ky * 3 + kx], acc[c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float o = acc[c];
                if (scale) o = __fmaf_rn(o, scale[cq * 4 + c], shift ? shift[cq * 4 + c] : 0.f);
                else if (shift) o = __fadd_rn(o, shift[cq * 4 + c]);
                if (relu) o = fmaxf(o, 0.f);
                acc[c] = o;
            }
            *reinterpret_cast<f32x4n *>(y + (((long long)n * Ho + yo) * Wo + xo) * y_ld + cq * 4) = acc;
        }
    }
}

extern "C" int srf_nhwc_dwconv3x3s2(const float *x, long long x_ld, int N, int H, int W, int C, const float *w, const float *scale,
                                    const float *shift, int relu, float *y, long long y_ld, srf_stream_t stream)
{
    if (N < 0 || H < 1 || W < 1 || C <= 0 || x_ld < C || y_ld < C) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!x || !w || !y) return SRF_EINVAL;
    if ((C & 3) || (x_ld & 3) || (y_ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return SRF_EUNSUPPORTED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)N * ((Ho + 1) / 2) * ((Wo + 1) / 2) * (C / 4);
    hipLaunchKernelGGL(srf_nhwc_dwconv3x3s2_k, dim3(srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_ld, N, H, W, C / 4, Ho, Wo,
                       w, scale, shift, relu, y, y_ld, (const float *)nullptr, 0LL, 0, (float *)nullptr);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// The step of the proposal generator's stair (srfdet_head.py:525-536: x = conv(x); x = cat([level, x], 1)) as one launch:
// the depthwise convolution writes `y` and the pyramid level `side` (N, Ho, Wo, Cs) is copied to `side_out`; both outputs
// are slices of the same concat buffer (row stride y_ld).
extern "C" int srf_nhwc_dwconv3x3s2_cat(const float *x, long long x_ld, int N, int H, int W, int C, const float *w, const float *scale,
                                        const float *shift, int relu, float *y, long long y_ld, const float *side, long long side_ld,
                                        int Cs, float *side_out, srf_stream_t stream)
{
    if (N < 0 || H < 1 || W < 1 || C <= 0 || Cs <= 0 || x_ld < C || y_ld < C + Cs || side_ld < Cs) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!x || !w || !y || !side || !side_out) return SRF_EINVAL;
    if ((C & 3) || (Cs & 3) || (x_ld & 3) || (y_ld & 3) || (side_ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15) ||
        ((uintptr_t)side & 15) || ((uintptr_t)side_out & 15))
        return SRF_EUNSUPPORTED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)N * ((Ho + 1) / 2) * ((Wo + 1) / 2) * (C / 4) + (long long)N * Ho * Wo * (Cs / 4);
    hipLaunchKernelGGL(srf_nhwc_dwconv3x3s2_k, dim3(srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_ld, N, H, W, C / 4, Ho, Wo,
                       w, scale, shift, relu, y, y_ld, side, side_ld, Cs / 4, side_out);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// srf_nhwc_pool_sum: the reduction between the stair and fc1 of the proposal generator (srfdet_head.py:537 for the BEV
// pyramid; :548-552 for the cameras: F.interpolate(nearest) to (Ho, Wo), sum over the cameras, sum over the channels):
//   out[b][p] = sum over cam < n_cam, c < C of x[b * n_cam + cam][sy(p)][sx(p)][c],   p = yo * Wo + xo,
// sy / sx the `nearest` source index of torch (floor(dst * in / out) in float32, clamped); Ho = H, Wo = W, n_cam = 1 is the
// plain channel sum.  One wave per output: lanes take float4 strides over (cam, c), then an xor-shuffle tree -- a fixed
// order.  Row b of `out` has out_ld >= Ho * Wo floats; the columns past Ho * Wo are written as zeros (the K padding of fc1).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void srf_nhwc_pool_sum_k(const float *__restrict__ x, long long x_ld, int B, int n_cam, int H, int W, int Cq,
                                                         int Ho, int Wo, float sy_scale, float sx_scale, float *__restrict__ out, int out_ld)
{
    const int lane = threadIdx.x & 63;
    const long long o = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= (long long)B * out_ld) return;
    const int b = (int)(o / out_ld), p = (int)(o % out_ld);
    if (p >= Ho * Wo) {
        if (lane == 0) out[o] = 0.0f;
        return;
    }
    const int yo = p / Wo, xo = p - yo * Wo;
    int sy = Ho == H ? yo : (int)floorf(__fmul_rn((float)yo, sy_scale));
    int sx = Wo == W ? xo : (int)floorf(__fmul_rn((float)xo, sx_scale));
    sy = sy < H - 1 ? sy : H - 1;
    sx = sx < W - 1 ? sx : W - 1;
    float acc = 0.0f;
    for (int e = lane; e < n_cam * Cq; e += 64) {
        const int cam = e / Cq, cq = e - cam * Cq;
        const f32x4n v = *reinterpret_cast<const f32x4n *>(x + ((((long long)b * n_cam + cam) * H + sy) * W + sx) * x_ld + cq * 4);
        acc = __fadd_rn(acc, __fadd_rn(__fadd_rn(v[0], v[1]), __fadd_rn(v[2], v[3])));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc = __fadd_rn(acc, __shfl_xor(acc, d, 64));
    if (lane == 0) out[o] = acc;
}

extern "C" int srf_nhwc_pool_sum(const float *x, long long x_ld, int B, int n_cam, int H, int W, int C, int Ho, int Wo, float *out, int out_ld,
                                 srf_stream_t stream)
{
    if (B < 0 || n_cam <= 0 || H < 1 || W < 1 || C <= 0 || Ho < 1 || Wo < 1 || x_ld < C || out_ld < Ho * Wo) return SRF_EINVAL;
    if (B == 0) return SRF_OK;
    if (!x || !out) return SRF_EINVAL;
    if ((C & 3) || (x_ld & 3) || ((uintptr_t)x & 15)) return SRF_EUNSUPPORTED;
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;  // torch: compute_scales_value<float>(input / output)
    hipLaunchKernelGGL(srf_nhwc_pool_sum_k, dim3(srf_ceil_div((long long)B * out_ld, 4)), dim3(256), 0, (hipStream_t)stream, x, x_ld, B, n_cam,
                       H, W, C / 4, Ho, Wo, sy, sx, out, out_ld);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// srf_nhwc_affine_relu_bwd: the backward pass of  y = relu(z * s[c] + t[c])  (a convolution's eval-mode BatchNorm + ReLU, run as the
// convolution kernel's epilogue in the forward pass) in ONE streaming pass over channels-last tensors:
//     gu = relu ? (y > 0 ? gy : 0) : gy          gz = gu * s[c]          sum_gu[c] = sum_p gu          sum_guy[c] = sum_p gu * y
// from which the caller forms  d beta = sum_gu  and  d s = sum_p gu * z = (sum_guy - t * sum_gu) / s  (z = (y - t) / s wherever
// gu != 0) without ever having stored z.  As torch ops this is threshold_backward, a multiply and two strided column sums
// (four passes over the tensor).  Column sums are deterministic: per-block partials (256 rows each), added in block order.
// ---------------------------------------------------------------------------------------------------------------------
#define SRF_ARB_ROWS 256

__global__ __launch_bounds__(256) void srf_affine_relu_bwd_k(const float *__restrict__ gy, long long gy_ld, const float *__restrict__ y,
                                                           long long y_ld, long long M, int Cq, const float *__restrict__ scale, int relu,
                                                           float *__restrict__ gz, long long gz_ld, float *__restrict__ partial,
                                                           const float *__restrict__ gy2, long long gy2_ld)
{
    __shared__ float s_red[256 * 8];
    const int rpp = 256 / Cq;  // rows per pass (Cq <= 256)
    const int r_local = threadIdx.x / Cq, cq = threadIdx.x - r_local * Cq;
    const long long row0 = (long long)blockIdx.x * SRF_ARB_ROWS;
    const long long row1 = row0 + SRF_ARB_ROWS < M ? row0 + SRF_ARB_ROWS : M;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r_local < rpp) {
        f32x4n s = {1.f, 1.f, 1.f, 1.f};
        if (scale) s = *reinterpret_cast<const f32x4n *>(scale + cq * 4);
        for (long long row = row0 + r_local; row < row1; row += rpp) {
            f32x4n g = *reinterpret_cast<const f32x4n *>(gy + row * gy_ld + cq * 4);
            const f32x4n v = *reinterpret_cast<const f32x4n *>(y + row * y_ld + cq * 4);
            if (gy2) {   // a second gradient of the same output (its other consumer): gy + gy2, the add autograd would make as a pass of its own
                const f32x4n g2 = *reinterpret_cast<const f32x4n *>(gy2 + row * gy2_ld + cq * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = __fadd_rn(g[j], g2[j]);
            }
            f32x4n o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gu = (relu && !(v[j] > 0.f)) ? 0.f : g[j];
                o[j] = __fmul_rn(gu, s[j]);
                acc[j] = __fadd_rn(acc[j], gu);
                acc[4 + j] = __fmaf_rn(gu, v[j], acc[4 + j]);
            }
            *reinterpret_cast<f32x4n *>(gz + row * gz_ld + cq * 4) = o;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s_red[threadIdx.x * 8 + j] = acc[j];
    __syncthreads();
    if (r_local == 0) {
        const int C = Cq * 4;
        float tot[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < rpp; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) tot[j] = __fadd_rn(tot[j], s_red[(r * Cq + cq) * 8 + j]);
        float *p = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p[cq * 4 + j] = tot[j];
            p[C + cq * 4 + j] = tot[4 + j];
        }
    }
}

// 16 columns x 16 segments per workgroup: segment g adds the partials of blocks [g nb / 16, (g + 1) nb / 16) in order, then the
// 16 segment sums are added in order -- a fixed tree (one thread walking all ~270 blocks of a stage-4 layer took 50 us)
__global__ __launch_bounds__(256) void srf_affine_relu_bwd_finish_k(const float *__restrict__ partial, int nblocks, int C2,
                                                                  float *__restrict__ sums)
{
    __shared__ float s_seg[16][17];
    const int col = threadIdx.x & 15, seg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + col;
    const int b0 = (int)((long long)nblocks * seg / 16), b1 = (int)((long long)nblocks * (seg + 1) / 16);
    float acc = 0.f;
    if (c < C2)
        for (int b = b0; b < b1; ++b) acc = __fadd_rn(acc, partial[(size_t)b * C2 + c]);
    s_seg[seg][col] = acc;
    __syncthreads();
    if (seg == 0 && c < C2) {
        float tot = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) tot = __fadd_rn(tot, s_seg[g][col]);
        sums[c] = tot;
    }
}

// The per-channel arithmetic around an eval-mode BatchNorm under training (norm_eval=True, vovnet.py:371), one launch each instead of
// five and seven element-wise torch launches per layer and step (76 layers: ~900 launches of ~4 us):
//   srf_bn_eval_fold_k:   inv = rsqrt(var + eps), s = gamma inv, t0 = beta - mean s          (the folded affine map y = z s + t0)
//   srf_bn_eval_grads_k:  d beta = sum gu,  d gamma = ((s != 0 ? (sum gu y - t0 sum gu) / s : 0) - mean sum gu) inv
// -- the operations of train_conv._ConvAffineRelu's torch expressions in their order, one f32 rounding each.
__global__ __launch_bounds__(256) void srf_bn_eval_fold_k(const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        const float *__restrict__ mean, const float *__restrict__ var, float eps, int C,
                                                        float *__restrict__ out /* [3][C]: s, t0, inv */)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float inv = rsqrtf(__fadd_rn(var[c], eps));
    const float sc = __fmul_rn(gamma[c], inv);
    out[c] = sc;
    out[C + c] = __fsub_rn(beta[c], __fmul_rn(mean[c], sc));
    out[2 * C + c] = inv;
}

__global__ __launch_bounds__(256) void srf_bn_eval_grads_k(const float *__restrict__ sums /* [2][C] */, const float *__restrict__ fold /* [3][C] */,
                                                         const float *__restrict__ mean, int C, float *__restrict__ out /* [2][C]: d gamma, d beta */)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float s0 = sums[c], s1 = sums[C + c], sc = fold[c], t0 = fold[C + c], inv = fold[2 * C + c];
    const float z = sc != 0.f ? __fdiv_rn(__fsub_rn(s1, __fmul_rn(t0, s0)), sc) : 0.f;
    out[c] = __fmul_rn(__fsub_rn(z, __fmul_rn(mean[c], s0)), inv);
    out[C + c] = s0;
}

extern "C" int srf_bn_eval_fold(const float *gamma, const float *beta, const float *mean, const float *var, float eps, int C, float *out,
                                srf_stream_t stream)
{
    if (C <= 0 || !gamma || !beta || !mean || !var || !out) return SRF_EINVAL;
    hipLaunchKernelGGL(srf_bn_eval_fold_k, dim3(srf_ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, eps, C, out);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_bn_eval_grads(const float *sums, const float *fold, const float *mean, int C, float *out, srf_stream_t stream)
{
    if (C <= 0 || !sums || !fold || !mean || !out) return SRF_EINVAL;
    hipLaunchKernelGGL(srf_bn_eval_grads_k, dim3(srf_ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, fold, mean, C, out);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" size_t srf_nhwc_affine_relu_bwd_workspace_bytes(long long M, int C)
{
    if (M <= 0 || C <= 0) return 0;
    return (size_t)((M + SRF_ARB_ROWS - 1) / SRF_ARB_ROWS) * 2 * C * sizeof(float);
}

extern "C" int srf_nhwc_affine_relu_bwd2(const float *gy, long long gy_ld, const float *gy2, long long gy2_ld, const float *y, long long y_ld, long long M, int C,
                                        const float *scale, int relu, float *gz, long long gz_ld, float *sums, void *workspace,
                                        size_t workspace_bytes, srf_stream_t stream)
{
    if (M < 0 || C <= 0 || gy_ld < C || y_ld < C || gz_ld < C || (gy2 && gy2_ld < C)) return SRF_EINVAL;
    if ((C & 3) || C > 1024 || (gy_ld & 3) || (y_ld & 3) || (gz_ld & 3) || ((uintptr_t)gy & 15) || ((uintptr_t)y & 15) || ((uintptr_t)gz & 15) ||
        (scale && ((uintptr_t)scale & 15)) || (gy2 && ((gy2_ld & 3) || ((uintptr_t)gy2 & 15))))
        return SRF_EUNSUPPORTED;
    if (!sums) return SRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        SRF_HIP_TRY(srf_fill_bytes(sums, 0, sizeof(float) * 2 * C, st));
        return SRF_OK;
    }
    if (!gy || !y || !gz || !workspace) return SRF_EINVAL;
    if (workspace_bytes < srf_nhwc_affine_relu_bwd_workspace_bytes(M, C)) return SRF_EWORKSPACE;
    const long long nb = (M + SRF_ARB_ROWS - 1) / SRF_ARB_ROWS;
    if (nb > 0x7FFFFFFF) return SRF_EUNSUPPORTED;
    hipLaunchKernelGGL(srf_affine_relu_bwd_k, dim3((unsigned)nb), dim3(256), 0, st, gy, gy_ld, y, y_ld, M, C / 4, scale, relu, gz, gz_ld,
                       (float *)workspace, gy2, gy2_ld);
    hipLaunchKernelGGL(srf_affine_relu_bwd_finish_k, dim3(srf_ceil_div(2 * C, 16)), dim3(256), 0, st, (const float *)workspace, (int)nb, 2 * C,
                       sums);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_nhwc_affine_relu_bwd(const float *gy, long long gy_ld, const float *y, long long y_ld, long long M, int C,
                                        const float *scale, int relu, float *gz, long long gz_ld, float *sums, void *workspace,
                                        size_t workspace_bytes, srf_stream_t stream)
{
    return srf_nhwc_affine_relu_bwd2(gy, gy_ld, nullptr, 0, y, y_ld, M, C, scale, relu, gz, gz_ld, sums, workspace, workspace_bytes, stream);
}
